/*
 * vitlora.h -- C ABI of libvitlora_hip.so: the MI355X (gfx950) implementation of the
 * ViT + LoRA + FGSM/PGD hot path of
 *   rneddojr/Adapting-Pretrained-Vision-Transformers-with-LoRA-against-Attack-Vectors.
 *
 * The reference has no FFI: the path is reached through Python callables.  Each
 * entry point below names the reference callable (file:line in /root/reference) it
 * stands in for; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - every launch function takes a hipStream_t (as void*) and only enqueues work:
 *     no allocation, no host synchronisation after vl_plan()/vl_set_workspace();
 *   - return value 0 = OK, negative = error; vl_last_error() gives the message;
 *   - one handle per device, driven by one host thread (the reference's model:
 *     a single Python thread drives one device, whitebox_attacks.py:67).
 */
#ifndef VITLORA_H
#define VITLORA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VL_OK 0
#define VL_ERR_ARG (-1)
#define VL_ERR_HIP (-2)
#define VL_ERR_STATE (-3)
#define VL_ERR_UNSUPPORTED (-4)
#define VL_ERR_NONFINITE (-5)   /* an EARLIER call produced a non-finite gradient (fp16 range exceeded): that step must be skipped / redone in f32 */

/* LoRA target bits.  peft matches module-name suffixes: the reference's
 * ["query","key","value","output.dense"] (train_loras.py:81) = Q|K|V|O|FC2. */
#define VL_T_Q 1u
#define VL_T_K 2u
#define VL_T_V 4u
#define VL_T_O 8u      /* attention.output.dense */
#define VL_T_FC1 16u   /* intermediate.dense     */
#define VL_T_FC2 32u   /* output.dense           */

#define VL_PREC_F16 0
#define VL_PREC_F32 1
#define VL_PREC_BF16 2     /* bf16 operands, fp32 accumulation: the same kernels as VL_PREC_F16 instantiated on __bf16 (no gradient-range
                              cliff, 8 mantissa bits); BASELINE config 5 / north_star name this type.
                              DOCUMENTED DEVIATION from north_star's "1e-2 bf16": against the fp32 reference this mode is held to
                              logits 1.5e-2, dLoss/dx 2.2e-2, LoRA gradients 3.5e-2 on ViT-B/16 (measured 7.9e-3 / 1.5e-2 / 2.5e-2;
                              tests/test_hip_bf16.py) and 2e-2 / 3e-2 on the 24-layer ViT-L/16 (tests/test_hip_patch.py).  Cause, priced on
                              the CPU oracle with no kernel involved (tools/error_budget_mixed.py, profiles/r05_bf16_mixed_mode_cpu.txt):
                              the bf16 MFMA OPERANDS alone (weights, LayerNorm output, q/k/v, probabilities, context, gelu, dz) cost
                              8.9e-3 of the ViT-B and 1.02e-2 of the ViT-L input gradient with EVERY other site in fp32 -- no storage
                              choice brings ViT-L under 1e-2 with bf16 products.  VL_PREC_F16 meets 1e-2 at both depths (1.8e-3 / 2.4e-3)
                              at the same speed and is the default everywhere, including patch_attack.py (config 5); bf16 is for range:
                              it cannot raise VL_ERR_NONFINITE. */

typedef struct vl_config {
    /* architecture: HF ViTConfig as built by create_vit_model, Utils.py:84-90 */
    int32_t image_size;   /* 224 */
    int32_t patch_size;   /* 16  */
    int32_t hidden;       /* 768  (multiple of 128; head_dim must be 64) */
    int32_t layers;       /* 12  */
    int32_t heads;        /* 12  */
    int32_t mlp;          /* 3072 (multiple of 128) */
    int32_t num_labels;   /* #lines of class_mappings.txt, whitebox_attacks.py:88-90 */
    float   ln_eps;       /* 1e-12 */
    /* LoRA: LoraConfig(r, lora_alpha, lora_dropout, target_modules), train_loras.py:83-90 */
    int32_t lora_r;       /* 0 = no adapters */
    float   lora_alpha;
    float   lora_dropout; /* applied to the LoRA branch input in train mode only */
    uint32_t lora_targets;/* VL_T_* bits */
    int32_t lora_merged;  /* 0: rank-r update fused into the GEMMs as extra K tiles;
                             1: W' = W + s*B*A folded at vl_lora_commit (merge_and_unload,
                                eval_compose.py:110) */
    int32_t precision;    /* VL_PREC_F16 (default): fp16 operands and 16-bit residual streams, fp32 accumulation (MFMA rate);
                             VL_PREC_BF16: the same kernels on bf16 operands (fp32's range: no VL_ERR_NONFINITE; 8 mantissa bits);
                             VL_PREC_F32: every operand and activation fp32 (v_mfma_f32_16x16x4_f32), the
                             parity mode held to 1e-3 against the reference's fp32 CPU path */
    int32_t reserved[3];
} vl_config;

typedef struct vl_model vl_model;

const char* vl_version(void);
const char* vl_last_error(void);

/* create_vit_model(num_classes) + setup_peft_lora(model, rank, alpha, dropout, targets)
 * (Utils.py:84-90, train_loras.py:79-95).  Allocates the packed frozen weights. */
int vl_create(const vl_config* cfg, vl_model** out);
int vl_destroy(vl_model* m);

/* model.load_state_dict(torch.load(path)) (whitebox_attacks.py:94): one call per
 * state-dict entry, HF-4.55.2 key names ("vit.encoder.layer.3.attention.attention.query.weight",
 * "classifier.bias", ...).  src = device fp32, contiguous, numel elements. */
int vl_load_tensor(vl_model* m, const char* name, const float* src, int64_t numel, void* stream);

/* LoRA / classifier master parameters live in ONE flat fp32 buffer (so that the
 * gradient all-reduce and Adam are single flat operations).
 * which: 0 = lora_A [r,in], 1 = lora_B [out,r].  target = one VL_T_* bit.
 * Classifier: layer = -1, which 0 = weight [C,D], 1 = bias [C]. */
int vl_param_tensor(vl_model* m, int layer, uint32_t target, int which, float** ptr, int64_t* numel);
int vl_param_flat(vl_model* m, float** ptr, int64_t* numel);
/* Re-derive the 16-bit GEMM operands from the flat master buffer (after loading an
 * adapter, after an optimiser step, or to merge).  PeftModel.from_pretrained /
 * merge_and_unload (train_loras.py:419, eval_compose.py:108-110). */
int vl_lora_commit(vl_model* m, void* stream);
/* The library tracks whether the flat parameters changed since the last commit: vl_param_tensor /
 * vl_param_flat hand out writable pointers and mark the handle dirty, vl_adam_step marks the model
 * whose flat buffer it updates, and a caller that writes through a pointer it kept calls
 * vl_params_changed.  vl_forward / vl_pgd_attack commit by themselves when the handle is dirty.  The contract is
 * therefore: writes the library can see (its own entry points) are never stale; a write through a KEPT pointer is the
 * caller's to announce (vl_params_changed).  The Python facade announces in-place torch operations on the flat
 * Parameter itself (it compares the Parameter's version counter before each forward / attack), so the reference's
 * unmodified torch.optim.Adam and parameter.copy_ are covered; writes through parameter.data or through a tensor obtained
 * from vl_param_tensor / vl_param_flat do not move that counter and are the caller's to announce (mark_dirty()). */
int vl_params_changed(vl_model* m);

/* merge_and_unload for one adapted module (eval_compose.py:102-114): W_out = W_in + (alpha/r) B A,
 * fp32 [out,in] device buffers of the caller (may alias); A, B are the module's current adapters. */
int vl_merge_weight(vl_model* m, int layer, uint32_t target, const float* W_in, float* W_out, void* stream);

/* mean/std used when normalise != 0 (default: ImageNet constants of get_normalization,
 * Utils.py:92-93; torchattacks' set_normalization_used(mean, std), whitebox_attacks.py:169). */
int vl_set_normalization(vl_model* m, const float mean[3], const float std[3]);

/* Workspace: vl_plan returns the bytes needed for batches up to max_batch;
 * train != 0 additionally keeps what the LoRA weight-gradient needs (per-layer activations, and the per-chunk copies of the LoRA
 * gradient that make it bit-reproducible: at most 32 x the LoRA parameters x 4 B, 123 MB at r = 8).  The bytes handed to
 * vl_set_workspace are the ONLY scratch memory the library touches (guard-band tested) and belong to it until the next
 * vl_set_workspace / vl_destroy; results do not depend on what they held before (tested with 0x00 against 0xFF fills).
 * What the bytes contain on the 16-bit path: the main activation workspace for max_batch images PLUS two half-batch eval
 * workspaces of ceil(min(max_batch, 191) / 2) images each for the two-chain forms of vl_pgd_attack / vl_forward(train = 0)
 * ("pgd_chains", "api_chains" below) -- for every plan, train-mode ones included (config 3's step attacks, then trains, through one
 * handle).  At max_batch = 256 that is about +75 % (ViT-B/16: 12 GB + 9 GB; the chains serve calls with 2 .. 191 images of that
 * plan); vl_debug_set_option(m, "pgd_chains", 1) BEFORE vl_plan leaves them out (round-4 ADVICE). */
int vl_plan(vl_model* m, int max_batch, int train, size_t* bytes);
int vl_set_workspace(vl_model* m, void* ws, size_t bytes);

/* model(x) -> logits [B, C] fp32  (LogitsModel.forward / get_model_output,
 * whitebox_attacks.py:13-19,41-48; peft_model.base_model(pixel_values=x).logits,
 * train_loras.py:310-311).  x = [B,3,S,S] fp32 NCHW.  normalise != 0 applies
 * (x-mean)/std with the ImageNet constants of get_normalization (Utils.py:92-93)
 * inside the patch gather (whitebox_attacks.py:26).  train != 0: LoRA dropout on,
 * activations for vl_backward_lora kept. */
int vl_forward(vl_model* m, const float* x, int batch, int normalise, int train,
               float* logits_out, void* stream);

/* F.cross_entropy(logits, labels) mean reduction (whitebox_attacks.py:29,
 * train_loras.py:313) on the logits of the last vl_forward.  labels int64 [B].
 * loss_out: 1 float (device). Also stages dLoss/dlogits for the backward calls. */
int vl_loss_ce(vl_model* m, const int64_t* labels, float* loss_out, void* stream);

/* Supply dLoss/dlogits [B,C] fp32 computed by the caller instead of vl_loss_ce (lets
 * `criterion(logits, labels).backward()` of train_loras.py:313-314 drive the chain). */
int vl_set_dlogits(vl_model* m, const float* dlogits, void* stream);

/* One backward pass producing both gradients; either output may be NULL. */
int vl_backward(vl_model* m, float* grad_x_out, float* flat_grad_out, void* stream);

/* LoRA dropout (LoraConfig.lora_dropout, train_loras.py:88; active only in vl_forward(train=1)):
 * the keep-mask of a LoRA branch input is a pure function of (seed, layer, projection, element),
 * regenerated by the dgrad and wgrad kernels instead of being stored; q, k, v share the mask of
 * the fused QKV projection's input.  The seed advances by one per train-mode forward.
 * vl_dropout_mask writes the mask (0 or 1/(1-p)) the LAST train-mode forward used for projection
 * proj (0 fused qkv, 1 attention out, 2 fc1, 3 fc2) of `layer`: [B*T, in] fp32 -- parity tests
 * feed it to the oracle. */
int vl_set_dropout_seed(vl_model* m, uint64_t seed);
int vl_dropout_mask(vl_model* m, int layer, int proj, float* out, void* stream);

/* loss.backward() restricted to the input: dLoss/dx [B,3,S,S] fp32 in the space of
 * the x given to vl_forward (perturbed.grad, whitebox_attacks.py:30-32). */
int vl_backward_input(vl_model* m, float* grad_x_out, void* stream);

/* loss.backward() restricted to the trainable parameters (train_loras.py:314):
 * flat_grad_out has the layout of vl_param_flat.  Requires vl_forward(train=1). */
int vl_backward_lora(vl_model* m, float* flat_grad_out, void* stream);

/* Fused FGSM/PGD update (K10): adv <- clamp(x0 + clamp(adv + alpha*sign(g) - x0, -eps, eps), lo, hi)
 * FGSM (whitebox_attacks.py:32-36) = one call with x0 == adv, alpha = eps. n elements. */
int vl_pgd_step(float* adv, const float* x0, const float* grad, float eps, float alpha,
                float lo, float hi, int64_t n, void* stream);
/* PGD random start (K11): adv = clamp(x0 + U(-eps,eps), lo, hi), counter-based RNG. */
int vl_pgd_init(float* adv, const float* x0, float eps, float lo, float hi, uint64_t seed,
                int64_t n, void* stream);

/* torchattacks.PGD(model, eps, alpha, steps, random_start)(images, labels)
 * (whitebox_attacks.py:112-113,170).  One iteration (forward, CE, backward-to-input,
 * fused step) is captured once into a hipGraph and replayed `steps` times (batches of 2 .. 191 images: one graph per
 * half batch, replayed on `stream` and on an internal stream that is joined before the result is copied out).
 * x0, adv_out: [B,3,S,S] fp32 in [0,1]; adv_out may not alias x0. */
int vl_pgd_attack(vl_model* m, const float* x0, const int64_t* labels, int batch,
                  float eps, float alpha, int steps, int random_start, uint64_t seed,
                  float* adv_out, void* stream);

/* optimizer.step() of torch.optim.Adam(lr, betas, eps) (train_loras.py:284,315) on a
 * flat buffer; t = 1-based step count. */
int vl_adam_step(float* param, const float* grad, float* m1, float* m2, float lr, float b1,
                 float b2, float eps, int t, int64_t n, void* stream);

/* save_images quantisation (Utils.py:108-112): clamp(0,1)*255 -> uint8 truncation,
 * CHW float -> HWC bytes. */
int vl_quantize_u8(const float* images, uint8_t* out_hwc, int batch, int channels, int height,
                   int width, void* stream);

/* dst[b,c,:,:] = src[b,c,:,:] * scale[c] + shift[c]  (torchattacks' normalize /
 * inverse_normalize around the attack when set_normalization_used was called). dst may alias src. */
int vl_channel_affine(float* dst, const float* src, const float scale[3], const float shift[3],
                      int batch, int64_t hw, void* stream);

/* Adversarial patch with expectation over transformations: the overlay ART's AdversarialPatchPyTorch._random_overlay
 * computes for patch_attack.py:47-75,193-208 (generate / apply_patch).  patch [3,ps,ps] fp32 in [0,1]; inv_affine [B,6]
 * fp32 device = per-image INVERSE affine matrix of torchvision's affine(angle, translate, scale) (the host samples scale /
 * rotation / location and builds it: patch.py); patch_type 0 = square, 1 = circle.  out may not alias images.
 *   out = clamp(images * (1 - M') + P' * M', 0, 1),  P' / M' = the resized (bilinear, ps -> S) patch / mask warped by the affine
 *   (bilinear / nearest, zero fill).
 * vl_patch_grad: d(loss)/d(patch) [3,ps,ps] from d(loss)/d(out) of the same overlay (the input gradient of vl_backward_input). */
int vl_patch_apply(const float* images, const float* patch, const float* inv_affine, int batch, int image_size,
                   int patch_size, int patch_type, float* out, void* stream);
int vl_patch_grad(const float* grad_out, const float* inv_affine, int batch, int image_size, int patch_size, int patch_type,
                  float* patch_grad, void* stream);
/* The same overlay with ART's perspective distortion (patch_attack.py:95 --distortion_scale_max > 0): both canvases go through
 * torchvision's perspective(startpoints, endpoints; bilinear, zero fill) BEFORE the affine.  persp [B,8] fp32 device = the
 * per-image coefficients (a..h) of torchvision's _get_perspective_coeffs: canvas pixel (x, y) of the warped canvas reads the
 * unwarped one at ((a X + b Y + c) / (g X + h Y + 1) - .5, (d X + e Y + f) / (g X + h Y + 1) - .5), X = x + .5, Y = y + .5
 * (the host draws the corner displacements and solves for the coefficients: patch.py).  persp == NULL: no distortion,
 * bit-identical to vl_patch_apply / vl_patch_grad. */
int vl_patch_apply_persp(const float* images, const float* patch, const float* inv_affine, const float* persp, int batch,
                         int image_size, int patch_size, int patch_type, float* out, void* stream);
int vl_patch_grad_persp(const float* grad_out, const float* inv_affine, const float* persp, int batch, int image_size,
                        int patch_size, int patch_type, float* patch_grad, void* stream);
/* x <- clamp(x, lo, hi): the patch is clipped to the classifier's clip_values after every optimiser step. */
int vl_clamp(float* x, float lo, float hi, int64_t n, void* stream);

/* ---- Swin Transformer + LoRA (BASELINE config 4: Swin-T + LoRA r = 16, PGD-40) -----------------------------------------
 * The reference lists Swin-T (README.md:53) without code; the model is HF's SwinForImageClassification (modeling_swin.py):
 * 4x4 patch embedding + LayerNorm, four stages of shifted-window attention blocks (window 7, relative position bias,
 * -100 shift mask), patch merging between stages, final LayerNorm + mean pool + classifier.  fp32 on the exact-f32 MFMA.
 * State-dict keys are HF-4.55.2 names ("swin.encoder.layers.S.blocks.B.attention.self.query.weight", ...).  LoRA adapters
 * (eval mode: no dropout, no weight gradients on this path) on the VL_T_* targets; A / B live in one flat fp32 buffer. */
typedef struct vl_swin_config {
    int32_t image_size;   /* 224 */
    int32_t patch_size;   /* 4   */
    int32_t embed_dim;    /* 96  (stage i has embed_dim << i channels and heads[i] heads of dimension 32) */
    int32_t depths[4];    /* 2, 2, 6, 2 */
    int32_t heads[4];     /* 3, 6, 12, 24 */
    int32_t window;       /* 7 */
    int32_t num_labels;
    float   ln_eps;       /* 1e-5 */
    int32_t lora_r;       /* 0, 4, 8, 12 or 16 */
    float   lora_alpha;
    uint32_t lora_targets;
    int32_t reserved[4];
} vl_swin_config;
typedef struct vl_swin vl_swin;
int vl_swin_create(const vl_swin_config* cfg, vl_swin** out);
int vl_swin_destroy(vl_swin* m);
int vl_swin_load_tensor(vl_swin* m, const char* name, const float* src, int64_t numel, void* stream);
int vl_swin_param_flat(vl_swin* m, float** ptr, int64_t* numel);
/* which: 0 = lora_A [r, in], 1 = lora_B [out, r] of `target` in block `block` of stage `stage` */
int vl_swin_param_tensor(vl_swin* m, int stage, int block, uint32_t target, int which, float** ptr, int64_t* numel);
int vl_swin_set_normalization(vl_swin* m, const float mean[3], const float std[3]);
int vl_swin_plan(vl_swin* m, int max_batch, size_t* bytes);
/* as vl_set_workspace (the planned bytes are zeroed once, synchronously).  Since round 5 the results do NOT depend on what the
 * bytes hold between calls (round 4 required them to stay zero where the channel-padding columns were): the 16-bit activations of
 * stages 1-2 are dense (no pad columns), pad columns that remain are written by the kernel that owns the row, and the K-tail heads
 * a GEMM reads past the last valid row are zeroed at the start of every forward (csrc/swin.hip: zero_tails_kernel); tested by
 * filling every planned byte with 0xFF between two runs (tests/test_hip_swin.py). */
int vl_swin_set_workspace(vl_swin* m, void* ws, size_t bytes);
int vl_swin_forward(vl_swin* m, const float* x, int batch, int normalise, float* logits_out, void* stream);
int vl_swin_loss_ce(vl_swin* m, const int64_t* labels, float* loss_out, void* stream);
int vl_swin_backward_input(vl_swin* m, float* grad_x_out, void* stream);
/* Batches of >= 32 images run as TWO half-batch chains on two streams (the caller's and an internal one that forks after the staging
 * copies and joins before the result is copied out); the chains' workspaces share the planned bytes with the main workspace, so a
 * forward held for vl_swin_backward_input does not survive an attack (the handle then reports "backward before loss").  Bit-identical
 * to one chain; VITLORA_SWIN_CHAINS=0 (environment, read at vl_swin_create) restores the single chain. */
int vl_swin_pgd_attack(vl_swin* m, const float* x0, const int64_t* labels, int batch, float eps, float alpha, int steps,
                       int random_start, uint64_t seed, float* adv_out, void* stream);
/* vl_check_errors for a Swin handle (same codes: VL_ERR_ARG bad label, VL_ERR_NONFINITE fp16 gradient out of range). */
int vl_swin_check_errors(vl_swin* m, void* stream);

/* Per-launch timing with HIP events on the launch stream, for bench.py's roofline object.
 * Between begin and report every kernel launch of this library is bracketed by an event
 * pair (PGD runs eagerly, not as a graph, while active).  report synchronises the device and
 * writes a JSON object {"<kernel>": {"n": launches, "ms": total, "flops": algorithmic,
 * "bytes": algorithmic}, ...} into buf. */
/* Synchronises `stream` and returns the error kernels have flagged since the last check: VL_ERR_ARG (label outside
 * [0, num_labels)), VL_ERR_NONFINITE (fp16 mode: a gradient left the fp16 range or is NaN; the caller skips that step or
 * redoes the batch with precision = f32 -- what an AMP skip-step does; torch autograd in the reference is fp32 and has no
 * counterpart).  Every other entry point reports the same condition at its next call, without synchronising. */
int vl_check_errors(vl_model* m, void* stream);

int vl_profile_begin(void);
int vl_profile_report(char* buf, size_t cap);

/* GEMM micro-benchmark (tools/gemm_sweep.py): random bf16 operands allocated internally, `iters`
 * launches timed with HIP events; epi = GemmEpilogue of csrc/gemm.h (+100: all rows stored to row 0). */
int vl_bench_gemm(int M, int N, int K1, int K2, int epi, int bn, int iters, float* ms_out);

/* GEMM self-check: one random GEMM (M % 128 == 0, N % 256 == 0) through the 128-row kernel and through the kernel
 * selected with pp_mode (0 = 256-row persistent kernel, 1 = ping-pong kernel, 2 / 3 = ping-pong kernel with the LoRA down
 * projection of 16 / 32 columns computed inside it, checked against the separate skinny GEMM); *max_diff = largest |difference|. */
int vl_check_gemm(int M, int N, int K1, int K2, int epi, int pp_mode, float* max_diff);
/* A/B switch of the ping-pong GEMM (same values as the environment variable VITLORA_GEMM_PP); returns the old mode. */
int vl_debug_set_gemm_pp(int mode);
/* A/B switch of the streaming GEMM (tall, shallow products: the Swin stages 1-2; csrc/gemm_stream.hip): bit 0 = kernel on,
 * bit 1 = LoRA down projections computed inside it (same values as VITLORA_GEMM_STREAM / VITLORA_GEMM_STREAM_DOWN);
 * returns the old mode. */
int vl_debug_set_gemm_stream(int mode);
/* Diagnostic switches of one handle: "dead_rows" (1: eval-mode forwards compute the last encoder layer on the CLS rows only,
 * exact; 0: every row) "fuse_pgd" (1: vl_pgd_attack applies K10 in the patch-gradient epilogue; 0: separate launch) and "pgd_chains" (0, default:
 * vl_pgd_attack runs batches of 2 .. 191 images as two half-batch chains -- one captured iteration each, own activation workspaces
 * carved behind the main one by vl_plan, two streams that meet at the start and the end of the attack; 1: one chain always, set
 * BEFORE vl_plan to save those workspaces; 2: two chains whenever the halves fit) and "api_chains" (0, default; 1: vl_forward(train = 0)
 * and the backward after it run such batches as the same two chains -- the adversarial-patch EoT step goes through these calls;
 * vl_debug_tensor then does not see the activations).  Results are bit-identical either way. */
int vl_debug_set_option(vl_model* m, const char* name, int value);
int vl_debug_set_cus(vl_model* m, int cus);   /* persistent GEMM grid size (diagnostic; default = the device's CU count) */

/* Introspection for tests / profiling.  vl_debug_counter: "graph_captures" = PGD graphs captured so far,
 * "commits" = vl_lora_commit executions, "dirty" = 1 if parameters changed since the last commit, "lds_poisons" = launches of the
 * "poison_lds" test hook (vl_debug_set_option: 1 = every profiled kernel launch is preceded by a kernel that fills every CU's
 * LDS with NaN patterns; results must not change -- process-wide, attacks then run without the graph). */
int vl_debug_counter(vl_model* m, const char* what, int64_t* value);
int vl_debug_tensor(vl_model* m, const char* what, int layer, void** ptr, int64_t* numel, int* dtype);

#ifdef __cplusplus
}
#endif
#endif /* VITLORA_H */
