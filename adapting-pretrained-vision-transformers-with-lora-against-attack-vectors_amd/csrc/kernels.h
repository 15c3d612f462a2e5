// Host-callable launchers of every kernel (all enqueue on the given stream, no sync).
#pragma once
#include "common.h"
#include "gemm.h"

namespace VLNS {

// elementwise.hip
void k_patch_gather(const float* x, h16* out, int B, int S, int P, int normalise, const float* mean,
                    const float* std, hipStream_t s);
void k_cls_rows(float* x, const float* cls, const float* pos, int B, int T, int D, hipStream_t s);
// h = LN(x) (h16) + row statistics.  With `delta`: first x_out = x + delta (fp32 residual stream + the h16
// output of the projection before it), then the LN of x_out; h == nullptr: the add alone.
// P / ng / t: optional fused LoRA down-projection of the row of h (8*ng rows of P, ng <= 3; t has 64 columns).
void k_layernorm_fwd(const float* x, h16* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                     float eps, const h16* delta, float* xout, const h16* P, int ng, h16* t, hipStream_t s,
                     int ldh = 0);   // ldh: row stride of h / delta (0 = D; > D for zero-padded h16 operands)
// P / ng / u: optional fused LoRA down-projection of the h16 output row (u[row][0..63] = dx_h[row] . P[j], 8*ng rows
// of P [>= 8*ng, D]; ng in {1, 2}; u has 64 columns), see lora_down_row in elementwise.hip
void k_layernorm_bwd(const h16* dh, const float* x, const float* mean, const float* rstd, const float* g,
                     const float* dres, float* dx, h16* dx_h, int M, int D, const h16* P, int ng, h16* u, hipStream_t s,
                     int* err = nullptr, int ldh = 0);   // err: mapped host word, set to 2 when a gradient leaves the fp16 range / is NaN; ldh: row stride of dh / dx_h
void k_head_fwd(const float* x, int B, int T, int D, int C, float eps, const float* g, const float* b, const float* Wc,
                const float* bc, float* xhat, float* xf, float* rstd, float* logits, hipStream_t s);
// ---- 16-bit residual streams (the ViT path's 16-bit precision, round 4): x / x_out / dres are h16 [M, D] ----
// k_layernorm_fwd16: x_out = round16(x + delta) (delta optional), h = LN(x_out); h == nullptr: the add alone;
//   err: set to 2 when x + delta leaves the fp16 range.  k_layernorm_bwd16: dres is read and overwritten IN PLACE
//   (it is the residual-gradient stream and the A operand of the next dgrad GEMM).  P / ng / t|u as above (P staged in LDS).
void k_cls_rows16(h16* x, const float* cls, const float* pos, int B, int T, int D, hipStream_t s);
void k_layernorm_fwd16(const h16* x, h16* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                       float eps, const h16* delta, h16* xout, const h16* P, int ng, h16* t, hipStream_t s, int* err = nullptr);
void k_layernorm_bwd16(const h16* dh, const h16* x, const float* mean, const float* rstd, const float* g, h16* dres, int M, int D,
                       const h16* P, int ng, h16* u, hipStream_t s, int* err = nullptr);
void k_head_fwd16(const h16* x, int B, int T, int D, int C, float eps, const float* g, const float* b, const float* Wc,
                  const float* bc, float* xhat, float* xf, float* rstd, float* logits, hipStream_t s);
// err: mapped host word; set to 1 when a label is outside [0, C) (that image's loss / dlogits become NaN)
void k_ce_loss(const float* logits, const int64_t* labels, int B, int C, float* dlogits, float* loss_img, float* loss,
               int* err, hipStream_t s);
// per-image power-of-two gradient scale of the fp16 backward: gscale[b] = 2^(10 - e), max|dlogits[b]| = f * 2^e with
// f in [0.5, 1); uniform != 0: one scale for the whole batch (parameter gradients sum over images)
void k_grad_scale(const float* dlogits, int B, int C, int uniform, float* gscale, float* inv_gscale, hipStream_t s);
// gscale: per-image factor applied to dlogits (nullptr = 1); dx and dx_h each optional
void k_head_bwd(const float* dlogits, const float* gscale, const float* Wc, const float* g, const float* xhat,
                const float* rstd, int B, int T, int D, int C, float* dx, h16* dx_h, hipStream_t s);
void k_classifier_grad(const float* dlogits, const float* xf, int B, int D, int C, float* dW, float* db, hipStream_t s);
// err (optional): mapped host word, set to 2 when a gradient element is not finite
void k_pgd_step(float* adv, const float* x0, const float* grad, float eps, float alpha, float lo, float hi, int64_t n,
                hipStream_t s, int* err = nullptr);
void k_zero(void* p, size_t bytes, hipStream_t s);   // bytes % 16 == 0; a kernel node, not a memset node, under capture
void k_pgd_init(float* adv, const float* x0, float eps, float lo, float hi, uint64_t seed, int64_t n, hipStream_t s);
// err (optional): set to 3 when a gradient element is not finite (that element is skipped)
void k_adam(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, int t, int64_t n,
            hipStream_t s, int* err = nullptr);
void k_channel_affine(float* dst, const float* src, const float* scale, const float* shift, int B, int64_t hw,
                      hipStream_t s);
void k_fill_random_h16(h16* dst, size_t n, uint64_t seed, hipStream_t s);   // U(-1,1), benchmarks only
void k_quantize(const float* img, uint8_t* out, int B, int C, int H, int W, hipStream_t s);
void k_pack_h16(const float* src, h16* dst, int rows, int cols, int ldd, int coff, float scale, hipStream_t s);
void k_pack_h16_t(const float* src, h16* dst, int rows, int cols, int ldd, int roff, float scale, hipStream_t s);
void k_merge_lora(const float* W, const float* A, const float* B, int out, int in, int r, float sc, h16* dst, int ldd,
                  int roff, h16* dstT, int lddT, int coffT, hipStream_t s);

void k_dropout(const h16* x, h16* xd, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s);
void k_dropout_mask(float* out, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s);
void k_merge_f32(const float* W, const float* A, const float* B, int out, int in, int r, float sc, float* dst,
                 hipStream_t s);


// attention32.hip (32x32x16 MFMA)
void attention32_set_ring(int on);   // 1: single-pass (ring) per-image backward, 0: two-phase form (process-wide)
int attention32_init(int device);   // per-device kernel attributes (outside any stream capture); 0 = ok
int k_attention32_fwd(const h16* qkv, h16* ctx, float* lse2, int B, int T, int H, int D, hipStream_t s);
int k_attention32_bwd(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T,
                      int H, int D, hipStream_t s);

// per-image persistent forms with the LoRA down projection fused in (attention32.hip); Ad / Bd == nullptr: attention only
int k_attention_img_fwd(const h16* qkv, h16* ctx, float* lse2, int B, int T, int H, int D, const h16* Ad, h16* t, int r,
                        hipStream_t s);
int k_attention_img_bwd(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T, int H,
                        int D, const h16* Bd, h16* u, int r, unsigned mods, hipStream_t s);

// cls_path.hip: the last encoder layer on CLS rows only (compact [B, D] buffers)
int k_attn_cls_fwd(const h16* qkv, h16* ctx_c, float* lse_c, int B, int T, int H, int D, hipStream_t s);
int k_attn_cls_bwd(const h16* qkv, const h16* ctx_c, const h16* dctx_c, const float* lse_c, h16* dqkv, int B, int T, int H, int D,
                   hipStream_t s);
// rows of the 16-bit residual streams <-> compact fp32 rows
void k_gather_rows(const h16* src, float* dst, int B, int D, int64_t stride, hipStream_t s);    // dst[b] = src[b * stride]
void k_scatter_rows(const float* src, h16* dst, int B, int D, int64_t stride, hipStream_t s);   // dst[b * stride] = src[b] (saturating)

// patch.hip: adversarial-patch overlay (warp-and-paste) and its gradient w.r.t. the patch
void k_patch_overlay(const float* img, const float* patch, const float* mats, const float* persp, float* out, int B, int S,
                     int ps, int circle, hipStream_t s);
void k_patch_overlay_bwd(const float* g, const float* mats, const float* persp, float* dpatch, int B, int S, int ps, int circle,
                         hipStream_t s);
void k_clamp(float* x, float lo, float hi, int64_t n, hipStream_t s);

// lora_grad.hip
// dB[n][j] = sum_m dy[m][n] * t[m][j] ; dA[j][k] = sum_m u[m][j] * x[m][k]  (fp32 outputs), deterministic (round 5):
// the token rows are cut into lora_wgrad_chunks(M) chunks; chunk c STORES its partial block at out + c * chunk_stride
// (a slab of per-chunk copies of the flat gradient) and k_reduce_chunks sums the copies in chunk order -- no atomics.
// inv_gscale: device pointer to the factor that undoes the fp16 gradient scale (element 0 is used; nullptr = 1)
int lora_wgrad_chunk(int M);      // token rows per chunk: >= 512, at most 32 chunks
int lora_wgrad_chunks(int M);
void k_lora_wgrad(const h16* L, int ldl, int ncols_l, const h16* Rm, int ldr, int ncols_r, int M, float scale,
                  float* out, int ldo, int transpose_out, const float* inv_gscale, int64_t chunk_stride, hipStream_t s);
void k_reduce_chunks(const float* slab, float* out, int64_t n, int chunks, int64_t stride, hipStream_t s);

}  // namespace VLNS
