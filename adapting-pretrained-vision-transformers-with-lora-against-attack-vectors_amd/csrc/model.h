// Model handle, packed weights and workspace of libvitlora_hip.so, shared by the API file
// (vitlora.hip: 16-bit operand path) and the fp32 parity path (vitlora_f32.hip).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "common.h"
// The handle type exists once per build of the 16-bit path (h16 = _Float16 / __bf16 pointer members): the tag itself is renamed
// (vl_model_f16 / vl_model_bf16) and every h16-typed struct below lives in the build's namespace, so no two translation units
// define one name differently (round-4 ADVICE: one-definition rule).  api_dispatch.cpp sees the opaque ABI type only and reads
// the precision through the vl_config the handle starts with (static_assert below).
#define vl_model VL_API(vl_model)
#include "../../include/vitlora.h"
#include "kernels.h"

enum { LQKV = 0, LO = 1, LFC1 = 2, LFC2 = 3 };
static const uint32_t kTargetBits[6] = {VL_T_Q, VL_T_K, VL_T_V, VL_T_O, VL_T_FC1, VL_T_FC2};

namespace VLNS {

struct Slot {           // one adapted module inside a fused projection
    int target_idx;     // 0..5 (q,k,v,o,fc1,fc2)
    int row_off;        // first output row of the module inside the fused projection
    int out, in;
    int ext_off;        // first column of its r slots inside the K extension
    int64_t a_off, b_off;  // offsets into the flat parameter buffer
};

struct Linear {
    int out = 0, in = 0;
    h16* W = nullptr;     // [out, in]
    h16* WT = nullptr;    // [in, out]
    float* bias = nullptr;
    float* Wf32 = nullptr;   // fp32 master: kept when adapters may be merged into W, and in fp32 mode
    float* Wrun = nullptr;   // fp32 mode: the operand the GEMMs read (= Wf32, or the merged copy W + s B A)
    int kext = 0;
    std::vector<Slot> slots;
    h16* Ad = nullptr;    // [kext, in]   t = x Ad^T
    h16* Bu = nullptr;    // [out, kext]  y += t Bu^T        (scaling folded in)
    h16* Bd = nullptr;    // [kext, out]  u = dy Bd^T
    h16* Au = nullptr;    // [in, kext]   dx += u Au^T       (scaling folded in)
};

struct Layer {
    Linear lin[4];
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};

struct Workspace {
    char* base = nullptr;
    size_t bytes = 0;
    int max_batch = 0, train = 0;
    int64_t Mpad = 0, Mppad = 0;
    h16* patches;
    std::vector<float*> xs;          // 2L+1 residual-stream snapshots [Mpad, D] (fp32 parity mode)
    std::vector<h16*> xs16;          // the same on the 16-bit path: an h16 residual stream (round 4)
    std::vector<float*> mean, rstd;  // 2L
    std::vector<h16*> h1, h2, a;     // per layer in train mode, shared otherwise
    std::vector<h16*> qkv, ctx, z;
    std::vector<float*> lse;
    std::vector<h16*> t[4];          // LoRA down outputs (per layer in train mode)
    float *xhat, *xf, *rstd_f, *logits, *dlogits, *loss, *loss_img;
    float *gscale, *inv_gscale;      // per-image power-of-two gradient scale of the 16-bit backward (and its inverse)
    float* dres[2];                  // fp32 mode: residual-gradient stream (ping-pong)
    h16 *dres_h;                     // 16-bit path: THE residual-gradient stream, updated in place, A operand of the dgrad GEMMs
    h16 *dh, *dctx, *dqkv, *dz, *u;
    h16* xd;                         // train mode: dropout(x) of the current LoRA branch / dgrad temporary [Mpad, MLP]
    float* wg_slab;                  // train mode: [lora_wgrad_chunks(max rows)][flat LoRA elements] per-chunk copies of the LoRA gradient (lora_grad.hip)
    float* grad_img;                 // [max_batch, 3, S, S] for vl_pgd_attack
    // last layer on CLS rows only (cls_path.hip): compact [Bc = round_up(B, 128), .] buffers, 16-bit operand path
    struct Cls {
        int64_t Bc = 0;
        float *x0, *x1, *x2, *mean, *rstd, *lse;     // residual stream before attention / after attention / after the MLP
        h16 *ctx, *delta, *h2, *a, *z, *t;
        float* dres[2];
        h16 *dres_h, *dh, *dctx, *dz, *u;
    } c;
    // persistent staging of vl_pgd_attack: the captured graph only ever sees these addresses
    float *stage_x0, *stage_adv;
    int64_t* stage_labels;
    // ---- fp32 parity mode (vitlora_f32.hip): every activation fp32 ----
    float* f_patches;
    std::vector<float*> f_h1, f_h2, f_a, f_qkv, f_ctx, f_z;
    std::vector<float*> f_t[4];
    float *f_dh, *f_dctx, *f_dqkv, *f_dz, *f_u, *f_xd, *f_tmp;
};

struct GraphEntry {
    int B; float eps, alpha;
    hipGraphExec_t exec;
    int chains;
    hipGraphExec_t exec1;     // chains == 2: the second half-batch's iteration (launched on the side stream)
};

}  // namespace VLNS

// what the handle-less entry points (vl_adam_step: fp16 build only) need to know about a live handle of EITHER build
struct VlFlatRecord { void* model; float* flat; int64_t flat_n; int* dirty; int** err_flag; };

struct vl_model {
    vl_config cfg;
    int D, L, H, MLP, S, P, G, NP, T, C, PK;   // PK = 3*P*P
    int r = 0;
    float scaling = 0.f;
    int f32 = 0;                                // cfg.precision == VL_PREC_F32
    int device = 0;
    // embeddings / head
    h16 *Wpe = nullptr, *WpeT = nullptr;
    float* Wpe_f32 = nullptr;                   // fp32 mode
    float *bpe = nullptr, *cls = nullptr, *pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr;
    std::vector<VLNS::Layer> layers;
    std::vector<void*> allocs;
    // flat trainable parameters: [layer][target]{A,B} ..., classifier W, classifier b
    float* flat = nullptr;
    int64_t flat_n = 0, cls_w_off = 0, cls_b_off = 0;
    int dirty = 1;                              // flat parameters changed since the last vl_lora_commit
    VLNS::Workspace ws;
    // state of the last forward
    int cur_B = 0, cur_M = 0, cur_norm = 0, cur_train = 0, have_loss = 0;
    uint64_t drop_seed = 0x5eed, drop_base = 0x5eed, drop_calls = 0;   // LoRA dropout: seed of the last train-mode forward
    // PGD graph cache (one executable graph per (batch, eps, alpha); staging buffers make it pointer-independent)
    std::vector<VLNS::GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;
    // vl_pgd_attack at small batches (round 4): the batch runs as TWO independent half-batch chains, captured as parallel branches
    // of the one graph (fork / join by events on cap_stream / side_stream) -- a GEMM launch costs about one round more than its
    // tiles (pipeline fill + exposed epilogue), and one chain's ends then meet the other's main loops.  Each chain has its own
    // activation workspace (carved behind the main one for batches up to CHAIN_MAX_BATCH / 2).
    static constexpr int CHAIN_MAX_BATCH = 191;     // below the batch at which the per-image attention kernels take over (3/4 of 256 CUs): both forms of a batch then use the same kernels
    VLNS::Workspace chain_ws[2];
    int chain_batch = 0;          // images each chain workspace holds (0: none planned)
    int pgd_chains = 0;           // "pgd_chains" / VITLORA_PGD_CHAINS: 0 = by batch size (2 for 2 <= batch <= 191), 1 = never, 2 = whenever it fits
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the same two chains behind vl_forward(train = 0) / vl_backward_input (opt-in, "api_chains": the adversarial-patch EoT step runs
    // through these calls): logits are gathered into the main workspace, vl_loss_ce works on the whole batch, the backward splits again
    int api_chains = 0;
    int fwd_chains = 0;           // how the last vl_forward ran (0: one chain in the main workspace; 2: two chains)
    int chain_B[2] = {0, 0}, chain_cls[2] = {0, 0};
    int64_t n_captures = 0, n_commits = 0;
    int use_graph = 1;
    int resid_epi = 2;    // residual add of the 16-bit stream in the GEMM epilogue (EPI_RESID_H16): 1 = attention output projection, 2 = + fc2, 0 = LayerNorm-side
    int plan_batch = 0, plan_train = 0;
    int attn_img_mode = -1;   // VITLORA_ATTN_IMG: 1 / 0 force the per-image attention kernels on / off, -1 = by batch size
    int num_cus = 256;
    int dead_rows = 1;        // eval-mode forward / backward: last layer on CLS rows only (VITLORA_DEAD_ROWS=0 or vl_debug_set_dead_rows: off)
    int cur_cls_only = 0;     // the last forward took that route
    int fuse_down_min_k = 2048;   // LoRA down projection inside the ping-pong GEMM for projections at least this deep (VITLORA_FUSE_DOWN_MIN_K)
    int small_m_rows = 16384;     // token rows up to which the "small batch" forms apply (VITLORA_SMALL_M_ROWS; batch <= 83 at T = 197)
    int fuse_pgd = 1;         // vl_pgd_attack: PGD step inside the patch-gradient epilogue (VITLORA_FUSE_PGD=0: separate K10 launch)
    int* err_flag = nullptr;                    // pinned host word written by kernels (bad label, ...), read at API entry
    float mean[3] = {0.485f, 0.456f, 0.406f};   // get_normalization, Utils.py:92-93
    float stdv[3] = {0.229f, 0.224f, 0.225f};
};

static_assert(offsetof(vl_model, cfg) == 0, "api_dispatch.cpp reads the precision through the vl_config the handle starts with");

// helpers defined in vitlora.hip
int vl_fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return vl_fail(VL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
namespace VLNS {
bool vl_drop_on(const vl_model* m);

// fp32 parity path (vitlora_f32.hip): reached through the fp16 build only (vl_create_bf16 refuses VL_PREC_F32)
#ifdef VL_BF16
static inline size_t f32_carve(vl_model*, int, int, char*, size_t off0) { return off0; }
static inline int f32_forward(vl_model*, const float*, int, int, int, hipStream_t) { return VL_ERR_STATE; }
static inline int f32_backward(vl_model*, float*, float*, hipStream_t) { return VL_ERR_STATE; }
static inline int f32_init(int) { return 0; }
#else
size_t f32_carve(vl_model* m, int B, int train, char* base, size_t off0);
int f32_forward(vl_model* m, const float* x, int B, int normalise, int train, hipStream_t s);
int f32_backward(vl_model* m, float* grad_x, float* flat_grad, hipStream_t s);
int f32_init(int device);     // kernel attributes; 0 = ok
#endif
}  // namespace VLNS
