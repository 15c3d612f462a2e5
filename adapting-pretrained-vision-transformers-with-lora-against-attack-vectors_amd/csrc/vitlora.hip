// C ABI of libvitlora_hip.so (include/vitlora.h): model handle, packed weights, workspace
// plan and the launch sequences for forward, backward-to-input, LoRA backward and the PGD
// loop (one hipGraph per iteration).  Host-side orchestration only: all arithmetic is in
// gemm.hip / attention.hip / elementwise.hip / lora_grad.hip.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "api_rename.h"      // handle-taking entry points are declared and defined as vl_*_f16 / vl_*_bf16 (api_dispatch.cpp exports the ABI names)
#include "model.h"
#include "prof.h"

// This file is compiled twice (common.h): process-wide state and the entry points without a handle exist once, in the fp16 build.
extern std::vector<VlFlatRecord> g_vl_models;      // live handles of BOTH builds (vl_adam_step finds the model a flat buffer belongs to)
extern long long g_poison_count;
#ifndef VL_BF16
Profiler* g_prof = nullptr;
int g_poison_lds = 0;
long long g_poison_count = 0;
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds_all[];
    constexpr int N = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < N; i += 256) lds_all[i] = pattern;
    __syncthreads();
    if (sink && lds_all[(threadIdx.x * 97) % N] != pattern) *sink = 1;      // keeps the stores alive
}
}  // namespace
void vl_poison_lds(hipStream_t s) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    // one workgroup owns a CU's whole LDS, so 2 x #CUs workgroups reach every CU of an otherwise idle stream
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2 * cus), dim3(256), 160 * 1024, s, 0xFFFFFFFFu, (unsigned*)nullptr);
    if (hipGetLastError() == hipSuccess) ++g_poison_count;
}
std::vector<VlFlatRecord> g_vl_models;
namespace {
thread_local std::string g_err;
}

int vl_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#endif
#define fail vl_fail
#define g_models g_vl_models

namespace VLNS {

namespace {

template <typename Tp>
int dev_alloc(vl_model* m, Tp** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(Tp) + 256) != hipSuccess) return fail(VL_ERR_HIP, "hipMalloc(%zu) failed", n * sizeof(Tp));
    if (hipMemset(q, 0, n * sizeof(Tp) + 256) != hipSuccess) return fail(VL_ERR_HIP, "hipMemset failed");
    m->allocs.push_back(q);
    *p = (Tp*)q;
    return VL_OK;
}

GemmArgs gemm_args(const h16* A, int lda, const h16* W, int ldw, int K, int M, int N) {
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A1 = A; g.lda1 = lda; g.W1 = W; g.ldw1 = ldw; g.K1 = K;
    g.M = M; g.Mvalid = M; g.N = N;
    return g;
}

void add_ext(GemmArgs& g, const h16* A2, int lda2, const h16* W2, int ldw2, int K2) {
    g.A2 = A2; g.lda2 = lda2; g.W2 = W2; g.ldw2 = ldw2; g.K2 = K2;
}

// 8-row groups of Bd the LayerNorm-backward kernel can project its output row onto (0: not fusable, the
// skinny GEMM runs): a 64-column u, r * modules <= 16, dy as wide as the residual stream.
// columns of the K extension in use: q, k, v keep fixed positions 0, r, 2r inside the fused qkv projection
int ext_cols(const vl_model* m, const Linear& ln) {
    int n = 0;
    for (const Slot& sl : ln.slots) n = sl.ext_off + m->r > n ? sl.ext_off + m->r : n;
    return n;
}
int fused_down(const vl_model* m, const Linear& ln) {
    if (m->cfg.lora_merged || !ln.kext || ln.slots.empty() || ln.kext != 64 || ln.out != m->D) return 0;
    const int nc = ext_cols(m, ln);
    return nc <= 8 ? 1 : nc <= 16 ? 2 : 0;
}

// same for the forward: t = h Ad^T out of the LayerNorm that writes h (not with LoRA dropout: the branch then reads
// dropout(h)).  Only r * modules <= 8 (e.g. r = 4 on q, v): with 24 columns (r = 8 on q, k, v) the 144 registers of
// P cost the LayerNorm more (+1.1 ms per PGD iteration) than the skinny GEMM it replaces (0.3 ms) -- measured.
int fused_down_fwd(const vl_model* m, const Linear& ln) {
    if (m->cfg.lora_merged || !ln.kext || ln.slots.empty() || ln.kext != 64 || ln.in != m->D || vl_drop_on(m)) return 0;
    const int nc = ext_cols(m, ln);
    return nc <= 8 ? 1 : 0;
}



// per-image persistent attention kernels: when the batch fills the chip (one workgroup per image), or forced by
// VITLORA_ATTN_IMG = 1 / 0 (tests exercise both forms at small batches)
bool attn_img(const vl_model* m, int B) {
    if (m->attn_img_mode >= 0) return m->attn_img_mode != 0;
    return B * 4 >= m->num_cus * 3;
}
// LoRA down projection that the per-image attention kernels can sum over heads in registers: rank <= 8, 64-column t / u
bool down_fusable(const vl_model* m, const Linear& ln) {
    return !m->cfg.lora_merged && !ln.slots.empty() && ln.kext == 64 && m->r <= 8;
}

// smallest K from which a LoRA down projection rides inside the ping-pong GEMM instead of a skinny launch of its own: deep
// products at any size (VITLORA_FUSE_DOWN_MIN_K, default 2048: fc2); every supported depth (K >= 768) at small batches
int down_min_k(const vl_model* m, int Mpad) { return Mpad <= m->small_m_rows ? 768 : m->fuse_down_min_k; }

// y = x W^T (+ LoRA) with epilogue; `t` receives the LoRA down projection when fused.
// stream_id = layer*4 + projection: names the dropout mask of this projection's LoRA branch input.
// t_ready: the LayerNorm that produced x already wrote t (fused_down_fwd below).
void linear_fwd(vl_model* m, const Linear& ln, const h16* x, h16* t, int Mpad, GemmArgs g, int epi, hipStream_t s,
                uint32_t stream_id, bool t_ready = false) {
    g.A1 = x; g.lda1 = ln.in; g.W1 = ln.W; g.ldw1 = ln.in; g.K1 = ln.in;
    g.M = Mpad; g.N = ln.out; g.bias = ln.bias;
    g.Mvalid = m->cur_M;
    if (ln.kext && !m->cfg.lora_merged) {
        const h16* xb = x;                  // LoRA branch input: dropout(x) in train mode (peft Linear.forward)
        if (vl_drop_on(m)) {
            k_dropout(x, m->ws.xd, (int64_t)m->cur_M * ln.in, m->drop_seed, stream_id, m->cfg.lora_dropout, s);
            xb = m->ws.xd;
        }
        GemmArgs d = gemm_args(xb, ln.in, ln.Ad, ln.in, ln.in, Mpad, ln.kext);
        d.Mvalid = m->cur_M; d.n_algo = m->r * (int)ln.slots.size();
        g.k2_algo = m->r;      // each output column sees r LoRA columns
        g.k2_used = ext_cols(m, ln);
        d.C = t; d.ldc = ln.kext;
        add_ext(g, t, ln.kext, ln.Bu, ln.kext, ln.kext);
        // long-K projections (fc2: K = 3072, t would cost a pass over the whole GELU output): the ping-pong GEMM computes t
        // itself from the A tiles it streams through LDS anyway (gemm_pp.hip, ND > 0) and still writes it out for wgrad
        bool fused = false;
        // (the small-batch extension only with the plain store: with the residual-add epilogue the fused form takes its bias from
        //  the LoRA K tile in fp16 -- results would then depend on the batch SIZE, and shards of a batch must reproduce it bit for bit)
        if (!t_ready && xb == x && ln.kext == 64 && ln.in >= (epi == EPI_STORE_H16 ? down_min_k(m, Mpad) : m->fuse_down_min_k)) {
            GemmArgs f = g;
            f.down_W = ln.Ad; f.down_ldw = ln.in; f.down_out = t; f.down_ld = ln.kext;
            f.down_groups = ext_cols(m, ln) <= 16 ? 1 : ext_cols(m, ln) <= 32 ? 2 : 0;
            if (epi == EPI_RESID_H16) { f.ones_col = 1; f.bias = nullptr; }      // bias through column 63 of the LoRA tile (vl_lora_commit)
            if (f.down_groups && gemm_pp_fuses_down(f, epi)) { g = f; g.A2 = nullptr; fused = true; }
        }
        if (!t_ready && !fused) launch_gemm(d, EPI_STORE_H16, 64, s);
    }
    launch_gemm(g, epi, 128, s);
}

// dx = dy W (+ LoRA) with epilogue; `u` receives dy B.
// u_ready: the kernel that produced dy already wrote u (fused_down below).
void linear_dgrad(vl_model* m, const Linear& ln, const h16* dy, h16* u, int Mpad, GemmArgs g, int epi, hipStream_t s,
                  uint32_t stream_id, bool u_ready = false) {
    g.A1 = dy; g.lda1 = ln.out; g.W1 = ln.WT; g.ldw1 = ln.out; g.K1 = ln.out;
    g.M = Mpad; g.N = ln.in; g.bias = nullptr;
    g.Mvalid = m->cur_M;
    if (ln.kext && !m->cfg.lora_merged) {
        GemmArgs d = gemm_args(dy, ln.out, ln.Bd, ln.out, ln.out, Mpad, ln.kext);
        // u = dy B: each of the r*slots columns sums over its own module's `out` rows only
        d.Mvalid = m->cur_M; d.n_algo = m->r; 
        d.C = u; d.ldc = ln.kext;
        // u inside the dgrad GEMM itself (the ping-pong kernel computes it from the dy tiles it streams through LDS: gemm_pp.hip,
        // ND > 0) when no producer of dy delivered it: deep products always, shallow ones at small batches, where a separate
        // skinny launch is pure latency (17 - 22 us for 6 - 12 k rows)
        bool fused = false;
        if (!u_ready && !vl_drop_on(m) && ln.out >= down_min_k(m, Mpad) && ln.kext == 64 && epi == EPI_STORE_H16) {
            GemmArgs f = g;
            f.k2_algo = m->r * (int)ln.slots.size();
            f.k2_used = ext_cols(m, ln);
            add_ext(f, u, ln.kext, ln.Au, ln.kext, ln.kext);
            f.down_W = ln.Bd; f.down_ldw = ln.out; f.down_out = u; f.down_ld = ln.kext;
            f.down_groups = ext_cols(m, ln) <= 16 ? 1 : ext_cols(m, ln) <= 32 ? 2 : 0;
            if (f.down_groups && gemm_pp_fuses_down(f, epi)) { f.A2 = nullptr; launch_gemm(f, epi, 128, s); return; }
        }
        (void)fused;
        if (!u_ready) launch_gemm(d, EPI_STORE_H16, 64, s);
        if (vl_drop_on(m)) {
            // the LoRA branch saw dropout(x): d(x) = dy W + mask * (u (sA)), then the caller's epilogue factor.
            // two launches: the frozen part into a temporary, then the masked rank-r part on top of it.
            GemmArgs main = g;
            main.C = m->ws.xd; main.ldc = ln.in; main.R = nullptr; main.C2 = nullptr;
            launch_gemm(main, EPI_STORE_H16, 128, s);
            GemmArgs lo = gemm_args(u, ln.kext, ln.Au, ln.kext, ln.kext, Mpad, ln.in);
            lo.Mvalid = m->cur_M; lo.k2_algo = 0;
            lo.C = g.C; lo.ldc = g.ldc; lo.R = m->ws.xd; lo.ldr = ln.in;
            if (epi == EPI_GELU_BWD) { lo.G = (const h16*)g.R; lo.ldg = g.ldr; }
            lo.drop_seed = m->drop_seed; lo.drop_stream = stream_id; lo.drop_p = m->cfg.lora_dropout;
            lo.drop_inv_keep = 1.f / (1.f - m->cfg.lora_dropout);
            launch_gemm(lo, EPI_DROP_ACC, 128, s);
            return;
        }
        g.k2_algo = m->r * (int)ln.slots.size();
        g.k2_used = ext_cols(m, ln);
        add_ext(g, u, ln.kext, ln.Au, ln.kext, ln.kext);
    }
    launch_gemm(g, epi, 128, s);
}

int parse_layer(const char* name, const char** rest) {
    // "vit.encoder.layer.<i>.<rest>"
    const char* pfx = "vit.encoder.layer.";
    const size_t n = strlen(pfx);
    if (strncmp(name, pfx, n) != 0) return -1;
    char* end = nullptr;
    long i = strtol(name + n, &end, 10);
    if (end == name + n || *end != '.') return -1;
    *rest = end + 1;
    return (int)i;
}

// errors a kernel reported through the pinned host word (label out of range): surfaced by the next API call
int check_async(vl_model* m) {
    if (m->err_flag && *m->err_flag) {
        const int code = *m->err_flag;
        *m->err_flag = 0;
        if (code == 1) return fail(VL_ERR_ARG, "a label passed to an earlier vl_loss_ce / vl_pgd_attack was outside [0, num_labels)");
        if (code == 2) return fail(VL_ERR_NONFINITE, "an earlier backward pass produced a non-finite input gradient (fp16 range exceeded): "
                                   "redo that batch with precision = f32");
        if (code == 3) return fail(VL_ERR_NONFINITE, "an earlier vl_adam_step saw a non-finite parameter gradient (those elements were "
                                   "skipped): drop that step or redo it with precision = f32");
        if (code == 4) return fail(VL_ERR_NONFINITE, "an earlier FORWARD pass left the fp16 range (the 16-bit residual stream exceeded 65504): "
                                   "redo that batch with precision = f32 or bf16");
        return fail(VL_ERR_HIP, "device-side error flag %d", code);
    }
    return VL_OK;
}

// launch failures (bad grid, LDS attribute not applied, ...) are sticky in the runtime: read them after a launch
// sequence that ran OUTSIDE stream capture
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VL_ERR_HIP, "%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return VL_OK;
}

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

}  // namespace

bool vl_drop_on(const vl_model* m) { return m->cur_train && m->r && m->cfg.lora_dropout > 0.f; }

extern "C" {

#ifndef VL_BF16
const char* vl_version(void) { return "vitlora-hip 0.3 (gfx950; fp16 / bf16 operands or fp32)"; }
#endif
#ifndef VL_BF16
const char* vl_last_error(void) { return g_err.c_str(); }
#endif

int vl_create(const vl_config* cfg, vl_model** out) {
    if (!cfg || !out) return fail(VL_ERR_ARG, "null argument");
    if (cfg->heads <= 0 || cfg->hidden != cfg->heads * 64)
        return fail(VL_ERR_UNSUPPORTED, "head_dim must be 64 (hidden %d, heads %d)", cfg->hidden, cfg->heads);
    if (cfg->hidden % 128 || cfg->mlp % 128) return fail(VL_ERR_UNSUPPORTED, "hidden and mlp must be multiples of 128");
    if (cfg->image_size % cfg->patch_size || cfg->patch_size % 8)
        return fail(VL_ERR_UNSUPPORTED, "image_size %% patch_size != 0 or patch_size %% 8 != 0");
    if ((3 * cfg->patch_size * cfg->patch_size) % 128) return fail(VL_ERR_UNSUPPORTED, "3*patch^2 must be a multiple of 128");
    const int G = cfg->image_size / cfg->patch_size;
    if (G * G + 1 > 224) return fail(VL_ERR_UNSUPPORTED, "at most 224 tokens");
    if (cfg->lora_r < 0 || cfg->lora_r > 64) return fail(VL_ERR_UNSUPPORTED, "lora_r must be in [0,64]");
    if (cfg->num_labels <= 0) return fail(VL_ERR_ARG, "num_labels must be positive");

#ifdef VL_BF16
    if (cfg->precision != VL_PREC_BF16) return fail(VL_ERR_ARG, "precision %d reached the bf16 build", cfg->precision);
#else
    if (cfg->precision != VL_PREC_F16 && cfg->precision != VL_PREC_F32) return fail(VL_ERR_ARG, "unknown precision %d", cfg->precision);
#endif
    if (cfg->precision != VL_PREC_F32 && cfg->hidden > 1024)      // k_layernorm_fwd16 / bwd16 hold a row in registers: D <= 1024 (ViT-B / ViT-L); round-4 ADVICE: refuse here, not by abort() at the first forward
        return fail(VL_ERR_UNSUPPORTED, "hidden %d: the 16-bit path supports hidden <= 1024 (ViT-B/16, ViT-L/16); use precision = f32", cfg->hidden);
    if (cfg->precision == VL_PREC_F32 && cfg->lora_targets && cfg->lora_r % 4)
        return fail(VL_ERR_UNSUPPORTED, "fp32 mode needs lora_r %% 4 == 0");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    // kernel attributes (dynamic LDS sizes) are per device: a rejected one would make every later launch fail
    if (int e = gemm_init(dev)) return fail(VL_ERR_HIP, "gemm_init: hipFuncSetAttribute failed (%d) on device %d", e, dev);
    if (int e = attention32_init(dev)) return fail(VL_ERR_HIP, "attention32_init: hipFuncSetAttribute failed (%d) on device %d", e, dev);
    if (int e = f32_init(dev)) return fail(VL_ERR_HIP, "f32_init: hipFuncSetAttribute failed (%d) on device %d", e, dev);
    vl_model* m = new vl_model();
    m->cfg = *cfg;
    m->device = dev;
    m->f32 = cfg->precision == VL_PREC_F32;
    m->D = cfg->hidden; m->L = cfg->layers; m->H = cfg->heads; m->MLP = cfg->mlp;
    m->S = cfg->image_size; m->P = cfg->patch_size; m->G = G; m->NP = G * G; m->T = G * G + 1;
    m->C = cfg->num_labels; m->PK = 3 * cfg->patch_size * cfg->patch_size;
    m->r = cfg->lora_targets ? cfg->lora_r : 0;
    m->scaling = m->r ? cfg->lora_alpha / (float)m->r : 0.f;
    const char* ng = getenv("VITLORA_NO_GRAPH");
    m->use_graph = !(ng && ng[0] == '1');
    // residual add of the 16-bit stream: 1 (default) = in the epilogue of the o / fc2 projection (EPI_RESID_H16: the stream row is
    // read two K steps ahead and x' = round16(x + acc + bias) stored -- the LayerNorm after it then moves 4 B per element instead
    // of 8); 0 (VITLORA_RESID=ln) = the projection stores a 16-bit delta and the LayerNorm adds it.  Same arithmetic, same rounding.
    // 2 (default) = attention output projection and fc2, 1 (VITLORA_RESID=o) = the former only, 0 (VITLORA_RESID=ln) = neither.
    // Measured on one box: 502.5 / 505.5 / 506.8 img/s for ln / o / both -- the row read in the epilogue is nearly as exposed as
    // the LayerNorm bytes it saves; what it buys is ONE rounding per residual add instead of two (the delta, then the sum): the
    // reference-driven PGD-20 trajectory on ViT-B keeps 98.5 % of its pixels instead of 97.6 % (CPU storage simulation: 98.9 / 98.0).
    { const char* re = getenv("VITLORA_RESID"); m->resid_epi = !re ? 2 : !strcmp(re, "ln") ? 0 : !strcmp(re, "o") ? 1 : 2; }
    { const char* dr = getenv("VITLORA_DEAD_ROWS"); m->dead_rows = !(dr && dr[0] == '0'); }
    { const char* fp = getenv("VITLORA_FUSE_PGD"); m->fuse_pgd = !(fp && fp[0] == '0'); }
    if (const char* pc = getenv("VITLORA_PGD_CHAINS")) { const int v = atoi(pc); m->pgd_chains = v < 0 ? 0 : v > 2 ? 2 : v; }
    { const char* fk = getenv("VITLORA_FUSE_DOWN_MIN_K"); if (fk) m->fuse_down_min_k = atoi(fk); }
    { const char* sm = getenv("VITLORA_SMALL_M_ROWS"); if (sm) m->small_m_rows = atoi(sm); }
    { const char* ai = getenv("VITLORA_ATTN_IMG"); m->attn_img_mode = ai ? (ai[0] == '1' ? 1 : 0) : -1; }
    { hipDeviceProp_t prop; m->num_cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
    const int D = m->D, MLP = m->MLP, r = m->r;
    int rc;
#define A_(p, n) if ((rc = dev_alloc(m, &(p), (size_t)(n))) != VL_OK) { vl_destroy(m); return rc; }
    const bool f32 = m->f32;
    if (f32) { A_(m->Wpe_f32, (size_t)D * m->PK); } else { A_(m->Wpe, (size_t)D * m->PK); A_(m->WpeT, (size_t)m->PK * D); }
    A_(m->bpe, D); A_(m->cls, D);
    A_(m->pos, (size_t)m->T * D); A_(m->lnf_g, D); A_(m->lnf_b, D);
    // flat parameter layout
    int64_t off = 0;
    m->layers.resize(m->L);
    for (int l = 0; l < m->L; ++l) {
        Layer& ly = m->layers[l];
        const int outs[4] = {3 * D, D, MLP, D}, ins[4] = {D, D, D, MLP};
        for (int k = 0; k < 4; ++k) {
            Linear& ln = ly.lin[k];
            ln.out = outs[k]; ln.in = ins[k];
            if (f32) { A_(ln.Wf32, (size_t)ln.out * ln.in); ln.Wrun = ln.Wf32; }
            else { A_(ln.W, (size_t)ln.out * ln.in); A_(ln.WT, (size_t)ln.out * ln.in); }
            A_(ln.bias, ln.out);
        }
        A_(ly.ln1_g, D); A_(ly.ln1_b, D); A_(ly.ln2_g, D); A_(ly.ln2_b, D);
        if (r) {
            // slots: q,k,v inside the fused qkv projection; o, fc1, fc2 alone
            const int lin_of[6] = {LQKV, LQKV, LQKV, LO, LFC1, LFC2};
            const int row_of[6] = {0, D, 2 * D, 0, 0, 0};
            const int out_of[6] = {D, D, D, D, MLP, D}, in_of[6] = {D, D, D, D, D, MLP};
            const int ext_of[6] = {0, r, 2 * r, 0, 0, 0};
            for (int ti = 0; ti < 6; ++ti) {
                if (!(cfg->lora_targets & kTargetBits[ti])) continue;
                Slot sl;
                sl.target_idx = ti; sl.row_off = row_of[ti]; sl.out = out_of[ti]; sl.in = in_of[ti];
                sl.ext_off = ext_of[ti];
                sl.a_off = off; off += (int64_t)r * sl.in;
                sl.b_off = off; off += (int64_t)sl.out * r;
                ly.lin[lin_of[ti]].slots.push_back(sl);
            }
            for (int k = 0; k < 4; ++k) {
                Linear& ln = ly.lin[k];
                if (ln.slots.empty()) continue;
                ln.kext = (int)round_up(k == LQKV ? 3 * r : r, 64);
                if (f32) { if (cfg->lora_merged) { A_(ln.Wrun, (size_t)ln.out * ln.in); } }
                else if (cfg->lora_merged) { A_(ln.Wf32, (size_t)ln.out * ln.in); }
                else {
                    A_(ln.Ad, (size_t)ln.kext * ln.in); A_(ln.Bu, (size_t)ln.out * ln.kext);
                    A_(ln.Bd, (size_t)ln.kext * ln.out); A_(ln.Au, (size_t)ln.in * ln.kext);
                }
            }
        }
    }
    m->cls_w_off = off; off += (int64_t)m->C * D;
    m->cls_b_off = off; off += m->C;
    m->flat_n = off;
    A_(m->flat, (size_t)off);
#undef A_
    // pinned host word that kernels write error codes to (mapped: the device writes through PCIe, the host reads it
    // at API entry without synchronising)
    if (hipHostMalloc((void**)&m->err_flag, 64, hipHostMallocMapped) != hipSuccess) { vl_destroy(m); return fail(VL_ERR_HIP, "hipHostMalloc failed"); }
    *m->err_flag = 0;
    g_models.push_back(VlFlatRecord{m, m->flat, m->flat_n, &m->dirty, &m->err_flag});
    *out = m;
    return VL_OK;
}

int vl_destroy(vl_model* m) {
    if (!m) return VL_OK;
    for (GraphEntry& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); if (g.exec1) (void)hipGraphExecDestroy(g.exec1); }
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->err_flag) (void)hipHostFree(m->err_flag);
    for (size_t i = 0; i < g_models.size(); ++i) if (g_models[i].model == (void*)m) { g_models.erase(g_models.begin() + i); break; }
    delete m;
    return VL_OK;
}

int vl_load_tensor(vl_model* m, const char* name, const float* src, int64_t numel, void* stream) {
    if (!m || !name || !src) return fail(VL_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int D = m->D;
    auto copyf = [&](float* dst, int64_t n) -> int {
        if (n != numel) return fail(VL_ERR_ARG, "%s: expected %lld elements, got %lld", name, (long long)n, (long long)numel);
        HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return VL_OK;
    };
    auto need = [&](int64_t n) -> int {
        return n == numel ? VL_OK : fail(VL_ERR_ARG, "%s: expected %lld elements, got %lld", name, (long long)n, (long long)numel);
    };
    int rc;
    if (!strcmp(name, "vit.embeddings.cls_token")) return copyf(m->cls, D);
    if (!strcmp(name, "vit.embeddings.position_embeddings")) return copyf(m->pos, (int64_t)m->T * D);
    if (!strcmp(name, "vit.embeddings.patch_embeddings.projection.weight")) {
        if ((rc = need((int64_t)D * m->PK))) return rc;
        if (m->f32) { HIPCHK(hipMemcpyAsync(m->Wpe_f32, src, (size_t)D * m->PK * sizeof(float), hipMemcpyDeviceToDevice, s)); return VL_OK; }
        k_pack_h16(src, m->Wpe, D, m->PK, m->PK, 0, 1.f, s);
        k_pack_h16_t(src, m->WpeT, D, m->PK, D, 0, 1.f, s);
        return VL_OK;
    }
    if (!strcmp(name, "vit.embeddings.patch_embeddings.projection.bias")) return copyf(m->bpe, D);
    if (!strcmp(name, "vit.layernorm.weight")) return copyf(m->lnf_g, D);
    if (!strcmp(name, "vit.layernorm.bias")) return copyf(m->lnf_b, D);
    if (!strcmp(name, "classifier.weight")) { m->dirty = 1; return copyf(m->flat + m->cls_w_off, (int64_t)m->C * D); }
    if (!strcmp(name, "classifier.bias")) { m->dirty = 1; return copyf(m->flat + m->cls_b_off, m->C); }
    const char* rest = nullptr;
    const int li = parse_layer(name, &rest);
    if (li < 0 || li >= m->L) return fail(VL_ERR_ARG, "unknown tensor name: %s", name);
    Layer& ly = m->layers[li];
    if (!strcmp(rest, "layernorm_before.weight")) return copyf(ly.ln1_g, D);
    if (!strcmp(rest, "layernorm_before.bias")) return copyf(ly.ln1_b, D);
    if (!strcmp(rest, "layernorm_after.weight")) return copyf(ly.ln2_g, D);
    if (!strcmp(rest, "layernorm_after.bias")) return copyf(ly.ln2_b, D);
    struct { const char* mod; int lin; int row_off; } mods[6] = {
        {"attention.attention.query.", LQKV, 0}, {"attention.attention.key.", LQKV, D},
        {"attention.attention.value.", LQKV, 2 * D}, {"attention.output.dense.", LO, 0},
        {"intermediate.dense.", LFC1, 0}, {"output.dense.", LFC2, 0}};
    for (auto& md : mods) {
        const size_t n = strlen(md.mod);
        if (strncmp(rest, md.mod, n) != 0) continue;
        Linear& ln = ly.lin[md.lin];
        const int rows = md.lin == LQKV ? D : ln.out;
        if (!strcmp(rest + n, "weight")) {
            if ((rc = need((int64_t)rows * ln.in))) return rc;
            m->dirty = 1;                    // a merged operand (W + s B A) has to be re-derived from the new master
            if (m->f32) {
                HIPCHK(hipMemcpyAsync(ln.Wf32 + (size_t)md.row_off * ln.in, src, (size_t)rows * ln.in * sizeof(float),
                                      hipMemcpyDeviceToDevice, s));
                return VL_OK;
            }
            k_pack_h16(src, ln.W + (size_t)md.row_off * ln.in, rows, ln.in, ln.in, 0, 1.f, s);
            k_pack_h16_t(src, ln.WT, rows, ln.in, ln.out, md.row_off, 1.f, s);
            if (ln.Wf32)
                HIPCHK(hipMemcpyAsync(ln.Wf32 + (size_t)md.row_off * ln.in, src, (size_t)rows * ln.in * sizeof(float),
                                      hipMemcpyDeviceToDevice, s));
            return VL_OK;
        }
        if (!strcmp(rest + n, "bias")) {
            if ((rc = need(rows))) return rc;
            HIPCHK(hipMemcpyAsync(ln.bias + md.row_off, src, rows * sizeof(float), hipMemcpyDeviceToDevice, s));
            m->dirty = 1;           // the bias also lives in column 63 of the LoRA up operand (vl_lora_commit)
            return VL_OK;
        }
    }
    return fail(VL_ERR_ARG, "unknown tensor name: %s", name);
}

int vl_param_flat(vl_model* m, float** ptr, int64_t* numel) {
    if (!m) return fail(VL_ERR_ARG, "null model");
    if (ptr) { *ptr = m->flat; m->dirty = 1; }     // a writable pointer leaves the library
    if (numel) *numel = m->flat_n;
    return VL_OK;
}

int vl_param_tensor(vl_model* m, int layer, uint32_t target, int which, float** ptr, int64_t* numel) {
    if (!m || !ptr || !numel) return fail(VL_ERR_ARG, "null argument");
    m->dirty = 1;                                  // a writable pointer leaves the library
    if (layer < 0) {
        *ptr = m->flat + (which == 0 ? m->cls_w_off : m->cls_b_off);
        *numel = which == 0 ? (int64_t)m->C * m->D : m->C;
        return VL_OK;
    }
    if (layer >= m->L) return fail(VL_ERR_ARG, "layer %d out of range", layer);
    for (int k = 0; k < 4; ++k)
        for (const Slot& sl : m->layers[layer].lin[k].slots)
            if (kTargetBits[sl.target_idx] == target) {
                *ptr = m->flat + (which == 0 ? sl.a_off : sl.b_off);
                *numel = which == 0 ? (int64_t)m->r * sl.in : (int64_t)sl.out * m->r;
                return VL_OK;
            }
    return fail(VL_ERR_ARG, "target 0x%x has no adapter in layer %d", target, layer);
}

int vl_lora_commit(vl_model* m, void* stream) {
    if (!m) return fail(VL_ERR_ARG, "null model");
    hipStream_t s = (hipStream_t)stream;
    const int r = m->r;
    m->dirty = 0;
    m->n_commits++;
    if (!r) return VL_OK;
    for (Layer& ly : m->layers)
        for (int k = 0; k < 4; ++k) {
            Linear& ln = ly.lin[k];
            if (m->f32) {
                // fp32 mode reads A / B straight from the flat master; only a merged operand is derived
                if (!m->cfg.lora_merged || ln.slots.empty()) continue;
                HIPCHK(hipMemcpyAsync(ln.Wrun, ln.Wf32, (size_t)ln.out * ln.in * sizeof(float), hipMemcpyDeviceToDevice, s));
                for (const Slot& sl : ln.slots)
                    k_merge_f32(ln.Wf32 + (size_t)sl.row_off * ln.in, m->flat + sl.a_off, m->flat + sl.b_off, sl.out, sl.in, r,
                                m->scaling, ln.Wrun + (size_t)sl.row_off * ln.in, s);
                continue;
            }
            for (const Slot& sl : ln.slots) {
                const float* A = m->flat + sl.a_off;   // [r, in]
                const float* B = m->flat + sl.b_off;   // [out, r]
                if (m->cfg.lora_merged) {
                    k_merge_lora(ln.Wf32 + (size_t)sl.row_off * ln.in, A, B, sl.out, sl.in, r, m->scaling, ln.W, ln.in,
                                 sl.row_off, ln.WT, ln.out, sl.row_off, s);
                } else {
                    k_pack_h16(A, ln.Ad + (size_t)sl.ext_off * ln.in, r, sl.in, ln.in, 0, 1.f, s);
                    k_pack_h16(B, ln.Bu + (size_t)sl.row_off * ln.kext, sl.out, r, ln.kext, sl.ext_off, m->scaling, s);
                    k_pack_h16_t(B, ln.Bd + (size_t)sl.ext_off * ln.out, sl.out, r, ln.out, sl.row_off, 1.f, s);
                    k_pack_h16_t(A, ln.Au, r, sl.in, ln.kext, sl.ext_off, m->scaling, s);
                }
            }
            // the projection's bias as column 63 of the LoRA up operand [out, 64]: inert for every producer of t (their column 63 is
            // zero); the ping-pong GEMM with the down projection inside puts a 1 there instead of loading the bias (gemm_pp.hip, BC)
            if (!m->cfg.lora_merged && !ln.slots.empty() && ln.kext == 64 && ext_cols(m, ln) <= 56 && ln.bias)
                k_pack_h16(ln.bias, ln.Bu, ln.out, 1, ln.kext, 63, 1.f, s);
        }
    if (!capturing(s)) return check_launch("vl_lora_commit");
    return VL_OK;
}

int vl_params_changed(vl_model* m) {
    if (!m) return fail(VL_ERR_ARG, "null model");
    m->dirty = 1;
    return VL_OK;
}

// merge_and_unload for one module (eval_compose.py:110): W_out = W_in + (alpha/r) * B A in fp32 on
// the device.  W_in / W_out: [out, in] fp32 device buffers of the caller (may alias).
int vl_merge_weight(vl_model* m, int layer, uint32_t target, const float* W_in, float* W_out, void* stream) {
    if (!m || !W_in || !W_out) return fail(VL_ERR_ARG, "null argument");
    if (layer < 0 || layer >= m->L) return fail(VL_ERR_ARG, "layer out of range");
    for (int k = 0; k < 4; ++k)
        for (const Slot& sl : m->layers[layer].lin[k].slots)
            if (kTargetBits[sl.target_idx] == target) {
                k_merge_f32(W_in, m->flat + sl.a_off, m->flat + sl.b_off, sl.out, sl.in, m->r, m->scaling, W_out,
                            (hipStream_t)stream);
                return VL_OK;
            }
    return fail(VL_ERR_ARG, "target 0x%x has no adapter in layer %d", target, layer);
}

// ---- workspace -------------------------------------------------------------------------------
static size_t carve(vl_model* m, int B, int train, char* base) {
    Workspace& w = m->ws;
    const int D = m->D, L = m->L, MLP = m->MLP;
    // token rows are padded to a multiple of 256 on the 16-bit path (round 4): the 256-row GEMM then never needs a second, 128-row
    // launch for a leftover half tile -- at batch 64 / 32 (SURVEY 8e's per-GPU shares) that launch was one full tile time for 3 tiles
    const int64_t Mpad = round_up((int64_t)B * m->T, m->f32 ? 128 : 256), Mppad = round_up((int64_t)B * m->NP, 128);
    size_t off = 0;
    auto take = [&](size_t bytes) -> char* {
        char* p = base ? base + off : nullptr;
        off += (size_t)round_up((int64_t)bytes, 256);
        return p;
    };
    w.Mpad = Mpad; w.Mppad = Mppad;
    // ---- common to both precisions: statistics, head, attack staging; the residual stream and its gradient stream are
    // fp32 in the fp32 parity mode and h16 on the 16-bit path (round 4: a LayerNorm pass moves 8 B per element, not 12 / 16) ----
    w.xs.assign(2 * L + 1, nullptr); w.xs16.assign(2 * L + 1, nullptr);
    if (m->f32) for (auto& p : w.xs) p = (float*)take((size_t)Mpad * D * 4);
    else for (auto& p : w.xs16) p = (h16*)take((size_t)Mpad * D * 2);
    w.mean.resize(2 * L); w.rstd.resize(2 * L);
    for (int i = 0; i < 2 * L; ++i) { w.mean[i] = (float*)take(Mpad * 4); w.rstd[i] = (float*)take(Mpad * 4); }
    w.lse.resize(L);
    for (int l = 0; l < L; ++l) w.lse[l] = (float*)take((size_t)B * m->H * m->T * 4);
    w.xhat = (float*)take((size_t)B * D * 4); w.xf = (float*)take((size_t)B * D * 4);
    w.rstd_f = (float*)take((size_t)B * 4);
    w.logits = (float*)take((size_t)B * m->C * 4); w.dlogits = (float*)take((size_t)B * m->C * 4);
    w.loss = (float*)take(256);
    w.loss_img = (float*)take((size_t)B * 4);
    w.gscale = (float*)take((size_t)B * 4); w.inv_gscale = (float*)take((size_t)B * 4);
    w.dres[0] = w.dres[1] = nullptr;
    if (m->f32) { w.dres[0] = (float*)take((size_t)Mpad * D * 4); w.dres[1] = (float*)take((size_t)Mpad * D * 4); }
    const size_t img = (size_t)B * 3 * m->S * m->S * 4;
    w.grad_img = (float*)take(img);
    w.stage_x0 = (float*)take(img); w.stage_adv = (float*)take(img);
    w.stage_labels = (int64_t*)take((size_t)B * 8);
    if (m->f32) return f32_carve(m, B, train, base, off);
    // ---- 16-bit operand path ----
    w.patches = (h16*)take((size_t)Mppad * m->PK * 2);
    w.h1.resize(L); w.h2.resize(L); w.a.resize(L); w.qkv.resize(L); w.ctx.resize(L); w.z.resize(L);
    int kext_max = 64;
    for (int k = 0; k < 4; ++k) { w.t[k].resize(L); if (m->layers[0].lin[k].kext > kext_max) kext_max = m->layers[0].lin[k].kext; }
    h16* sh_h = train ? nullptr : (h16*)take((size_t)Mpad * D * 2);
    h16* sh_a = train ? nullptr : (h16*)take((size_t)Mpad * MLP * 2);
    h16* sh_t = train ? nullptr : (h16*)take((size_t)Mpad * kext_max * 2);
    for (int l = 0; l < L; ++l) {
        w.h1[l] = train ? (h16*)take((size_t)Mpad * D * 2) : sh_h;
        w.h2[l] = train ? (h16*)take((size_t)Mpad * D * 2) : sh_h;
        w.a[l] = train ? (h16*)take((size_t)Mpad * MLP * 2) : sh_a;
        w.qkv[l] = (h16*)take((size_t)Mpad * 3 * D * 2);
        w.ctx[l] = (h16*)take((size_t)Mpad * D * 2);
        w.z[l] = (h16*)take((size_t)Mpad * MLP * 2);
        for (int k = 0; k < 4; ++k)
            w.t[k][l] = train ? (h16*)take((size_t)Mpad * kext_max * 2) : sh_t;
    }
    {   // compact CLS-row buffers of the last layer
        Workspace::Cls& c = w.c;
        const int64_t Bc = round_up(B, 128);
        c.Bc = Bc;
        c.x0 = (float*)take((size_t)Bc * D * 4); c.x1 = (float*)take((size_t)Bc * D * 4); c.x2 = (float*)take((size_t)Bc * D * 4);
        c.mean = (float*)take((size_t)Bc * 4); c.rstd = (float*)take((size_t)Bc * 4);
        c.lse = (float*)take((size_t)B * m->H * 4);
        c.ctx = (h16*)take((size_t)Bc * D * 2); c.delta = (h16*)take((size_t)Bc * D * 2); c.h2 = (h16*)take((size_t)Bc * D * 2);
        c.a = (h16*)take((size_t)Bc * MLP * 2); c.z = (h16*)take((size_t)Bc * MLP * 2);
        c.t = (h16*)take((size_t)Bc * kext_max * 2);
        c.dres[0] = (float*)take((size_t)Bc * D * 4); c.dres[1] = (float*)take((size_t)Bc * D * 4);
        c.dres_h = (h16*)take((size_t)Bc * D * 2); c.dh = (h16*)take((size_t)Bc * D * 2); c.dctx = (h16*)take((size_t)Bc * D * 2);
        c.dz = (h16*)take((size_t)Bc * MLP * 2); c.u = (h16*)take((size_t)Bc * kext_max * 2);
    }
    w.dres_h = (h16*)take((size_t)Mpad * D * 2);
    w.dh = (h16*)take((size_t)Mpad * D * 2);
    w.dctx = (h16*)take((size_t)Mpad * D * 2);
    w.dqkv = (h16*)take((size_t)Mpad * 3 * D * 2);
    w.dz = (h16*)take((size_t)Mpad * MLP * 2);
    w.u = (h16*)take((size_t)Mpad * kext_max * 2);
    w.xd = train ? (h16*)take((size_t)Mpad * MLP * 2) : nullptr;
    // per-chunk copies of the LoRA gradient (deterministic weight gradients: lora_grad.hip); the LoRA parameters are the
    // first cls_w_off elements of the flat buffer
    w.wg_slab = train && m->cls_w_off > 0 ? (float*)take((size_t)lora_wgrad_chunks((int)((int64_t)B * m->T)) * (size_t)m->cls_w_off * 4) : nullptr;
    return off;
}

static void drop_graphs(vl_model* m) {
    for (GraphEntry& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); if (g.exec1) (void)hipGraphExecDestroy(g.exec1); }
    m->graphs.clear();
}

static int chain_resources(vl_model* m) {
    if (!m->side_stream) HIPCHK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
    if (!m->ev_fork) HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    if (!m->ev_join) HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    return VL_OK;
}

// Two-chain PGD (model.h): each chain's activations for up to `cb` images, carved behind the main workspace.  carve() fills
// m->ws, so the chain structs are swapped in while it runs.
static int chain_images(const vl_model* m, int max_batch) {
    if (m->f32 || m->pgd_chains == 1 || max_batch < 2) return 0;
    const int top = max_batch < vl_model::CHAIN_MAX_BATCH ? max_batch : vl_model::CHAIN_MAX_BATCH;
    return (top + 1) / 2;
}
static size_t chain_bytes(vl_model* m, int max_batch) {
    const int cb = chain_images(m, max_batch);
    if (!cb) return 0;
    std::swap(m->ws, m->chain_ws[0]);
    const size_t one = carve(m, cb, 0, nullptr);
    std::swap(m->ws, m->chain_ws[0]);
    return 2 * one;
}

int vl_plan(vl_model* m, int max_batch, int train, size_t* bytes) {
    if (!m || !bytes || max_batch <= 0) return fail(VL_ERR_ARG, "bad argument");
    if (!m->f32) {
        // the 16-bit GEMM / attention kernels address an operand with 32-bit byte offsets from its base: the widest activation
        // ([token rows, max(3 D, MLP, 3 P^2)] h16) must stay below 4 GiB (ViT-B: 3 547 images per call, ViT-L: 2 660)
        const int64_t rows = round_up((int64_t)max_batch * m->T, 256);
        const int64_t wide = std::max<int64_t>(std::max<int64_t>(3 * m->D, m->MLP), m->PK);
        if (rows * wide >= ((int64_t)1 << 31))
            return fail(VL_ERR_UNSUPPORTED, "max_batch %d: an activation of %lld x %lld 16-bit elements exceeds the 4 GiB the kernels' 32-bit "
                        "operand offsets reach; split the batch (at most %lld images per call for this architecture)", max_batch,
                        (long long)rows, (long long)wide, (long long)((((int64_t)1 << 31) / wide - 255) / m->T));
    }
    *bytes = carve(m, max_batch, train, nullptr) + chain_bytes(m, max_batch);
    m->ws.max_batch = 0;   // a plan alone does not arm the workspace
    m->plan_batch = max_batch; m->plan_train = train;
    return VL_OK;
}

// arms the workspace for the (max_batch, train) of the last vl_plan
int vl_set_workspace(vl_model* m, void* wsp, size_t bytes) {
    if (!m || !wsp) return fail(VL_ERR_ARG, "null argument");
    if (m->plan_batch <= 0) return fail(VL_ERR_STATE, "vl_set_workspace before vl_plan");
    const int max_batch = m->plan_batch, train = m->plan_train;
    if (((uintptr_t)wsp) & 255) return fail(VL_ERR_ARG, "workspace must be 256-byte aligned");
    const size_t main_need = carve(m, max_batch, train, nullptr);
    const size_t need = main_need + chain_bytes(m, max_batch);
    if (bytes < need) return fail(VL_ERR_ARG, "workspace too small: %zu < %zu", bytes, need);
    carve(m, max_batch, train, (char*)wsp);
    m->ws.base = (char*)wsp; m->ws.bytes = bytes; m->ws.max_batch = max_batch; m->ws.train = train;
    m->chain_batch = chain_images(m, max_batch);
    for (int c = 0; c < 2 && m->chain_batch; ++c) {
        char* cbase = (char*)wsp + main_need + (size_t)c * ((need - main_need) / 2);
        std::swap(m->ws, m->chain_ws[c]);
        carve(m, m->chain_batch, 0, cbase);
        m->ws.base = cbase; m->ws.bytes = (need - main_need) / 2; m->ws.max_batch = m->chain_batch; m->ws.train = 0;
        std::swap(m->ws, m->chain_ws[c]);
    }
    if (hipMemset(wsp, 0, need) != hipSuccess) return fail(VL_ERR_HIP, "hipMemset(workspace) failed");
    drop_graphs(m);
    m->cur_B = 0;
    return VL_OK;
}

// ---- forward ---------------------------------------------------------------------------------
static int forward_impl(vl_model* m, const float* x, int B, int normalise, int train, hipStream_t s) {
    Workspace& w = m->ws;
    if (B <= 0 || B > w.max_batch) return fail(VL_ERR_STATE, "batch %d exceeds planned workspace (%d)", B, w.max_batch);
    if (train && !w.train) return fail(VL_ERR_STATE, "workspace was not planned for training");
    if (train && m->cfg.lora_merged) return fail(VL_ERR_STATE, "training needs lora_merged = 0");
    if (train && m->r && m->cfg.lora_dropout >= 1.f) return fail(VL_ERR_ARG, "lora_dropout must be < 1");
    const int D = m->D, L = m->L, T = m->T;
    const int Mpad = (int)round_up((int64_t)B * T, m->f32 ? 128 : 256), Mppad = (int)round_up((int64_t)B * m->NP, 128);
    const int M = B * T;
    m->cur_M = M;
    m->cur_train = train;
    if (train) m->drop_seed = m->drop_base + (++m->drop_calls);
    if (m->f32) {
        int rc = f32_forward(m, x, B, normalise, train, s);
        if (rc) return rc;
        m->cur_B = B; m->cur_norm = normalise; m->cur_train = train; m->have_loss = 0;
        return VL_OK;
    }
    k_patch_gather(x, w.patches, B, m->S, m->P, normalise, m->mean, m->stdv, s);
    {
        GemmArgs g = gemm_args(w.patches, m->PK, m->Wpe, m->PK, m->PK, Mppad, D);
        g.Mvalid = B * m->NP; g.bias = m->bpe; g.C = w.xs16[0]; g.ldc = D;
        g.pos = m->pos; g.tokens = T; g.patches = m->NP;
        launch_gemm(g, EPI_PATCH_FWD, 128, s);
    }
    k_cls_rows16(w.xs16[0], m->cls, m->pos, B, T, D, s);
    // residual stream (h16): the o / fc2 projections store their output (bias and LoRA included) as
    // h16 and the LayerNorm that follows adds it to the stream while it normalises: x' = round16(x + delta), h = LN(x')
    h16* delta = w.dres_h;                         // backward scratch, idle during the forward
    int* const ef = m->err_flag;
    const bool re = m->resid_epi >= 1, re2 = m->resid_epi >= 2;      // o projection / fc2
    // eval-mode forwards only feed logits and input gradients: the last layer runs on the CLS rows alone
    const bool cls_only = m->dead_rows && !train;
    m->cur_cls_only = 0;
    for (int l = 0; l < L; ++l) {
        Layer& ly = m->layers[l];
        GemmArgs g;
        const int n1 = fused_down_fwd(m, ly.lin[LQKV]);      // t of the qkv projection comes out of LN1
        const h16* P1 = n1 ? ly.lin[LQKV].Ad : nullptr;
        if (l == 0 || re2) k_layernorm_fwd16(w.xs16[2 * l], w.h1[l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, ly.ln1_b, M, D, m->cfg.ln_eps, nullptr, nullptr, P1, n1, w.t[LQKV][l], s, ef);
        else k_layernorm_fwd16(w.xs16[2 * l - 1], w.h1[l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, ly.ln1_b, M, D, m->cfg.ln_eps, delta, w.xs16[2 * l], P1, n1, w.t[LQKV][l], s, ef);
        memset(&g, 0, sizeof g); g.C = w.qkv[l]; g.ldc = 3 * D;
        linear_fwd(m, ly.lin[LQKV], w.h1[l], w.t[LQKV][l], Mpad, g, EPI_STORE_H16, s, l * 4 + LQKV, n1 > 0);
        if (cls_only && l == L - 1) {
            // ---- last layer, CLS rows only (cls_path.hip): B rows from here to the classifier ----
            Workspace::Cls& c = w.c;
            const int Bc = (int)c.Bc;
            if (k_attn_cls_fwd(w.qkv[l], c.ctx, c.lse, B, T, m->H, D, s)) return fail(VL_ERR_UNSUPPORTED, "attention: T > 256");
            m->cur_M = B;                                          // rows the compact GEMMs may store
            memset(&g, 0, sizeof g); g.C = c.delta; g.ldc = D;
            linear_fwd(m, ly.lin[LO], c.ctx, c.t, Bc, g, EPI_STORE_H16, s, l * 4 + LO);
            k_gather_rows(w.xs16[2 * l], c.x0, B, D, (int64_t)T * D, s);
            const int n2c = fused_down_fwd(m, ly.lin[LFC1]);
            k_layernorm_fwd(c.x0, c.h2, c.mean, c.rstd, ly.ln2_g, ly.ln2_b, B, D, m->cfg.ln_eps, c.delta, c.x1,
                            n2c ? ly.lin[LFC1].Ad : nullptr, n2c, c.t, s);
            memset(&g, 0, sizeof g); g.C = c.a; g.ldc = m->MLP; g.C2 = c.z; g.ldc2 = m->MLP;
            linear_fwd(m, ly.lin[LFC1], c.h2, c.t, Bc, g, EPI_GELU, s, l * 4 + LFC1, n2c > 0);
            memset(&g, 0, sizeof g); g.C = c.delta; g.ldc = D;
            linear_fwd(m, ly.lin[LFC2], c.a, c.t, Bc, g, EPI_STORE_H16, s, l * 4 + LFC2);
            k_layernorm_fwd(c.x1, nullptr, nullptr, nullptr, nullptr, nullptr, B, D, m->cfg.ln_eps, c.delta, c.x2, nullptr, 0, nullptr, s);
            m->cur_M = M;
            k_head_fwd(c.x2, B, 1, D, m->C, m->cfg.ln_eps, m->lnf_g, m->lnf_b, m->flat + m->cls_w_off,
                       m->flat + m->cls_b_off, w.xhat, w.xf, w.rstd_f, w.logits, s);
            m->cur_B = B; m->cur_norm = normalise; m->cur_train = train; m->have_loss = 0; m->cur_cls_only = 1;
            return VL_OK;
        }
        // large batches: one persistent workgroup per image walks the heads; the LoRA down projection of the output
        // projection (t = ctx Ad^T) is summed over heads inside it, the skinny GEMM over ctx disappears
        const bool img = attn_img(m, B);
        const bool t_o = img && down_fusable(m, ly.lin[LO]) && !vl_drop_on(m);
        if (img) {
            if (k_attention_img_fwd(w.qkv[l], w.ctx[l], w.lse[l], B, T, m->H, D, t_o ? ly.lin[LO].Ad : nullptr, w.t[LO][l], m->r, s))
                return fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        } else if (k_attention32_fwd(w.qkv[l], w.ctx[l], w.lse[l], B, T, m->H, D, s)) return fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        memset(&g, 0, sizeof g); g.C = delta; g.ldc = D;
        if (re) { g.C = w.xs16[2 * l + 1]; g.R = w.xs16[2 * l]; g.ldr = D; }
        linear_fwd(m, ly.lin[LO], w.ctx[l], w.t[LO][l], Mpad, g, re ? EPI_RESID_H16 : EPI_STORE_H16, s, l * 4 + LO, t_o);
        const int n2 = fused_down_fwd(m, ly.lin[LFC1]);      // t of fc1 (r columns) comes out of LN2
        const h16* P2 = n2 ? ly.lin[LFC1].Ad : nullptr;
        if (re) k_layernorm_fwd16(w.xs16[2 * l + 1], w.h2[l], w.mean[2 * l + 1], w.rstd[2 * l + 1], ly.ln2_g, ly.ln2_b, M, D, m->cfg.ln_eps, nullptr, nullptr, P2, n2, w.t[LFC1][l], s, ef);
        else k_layernorm_fwd16(w.xs16[2 * l], w.h2[l], w.mean[2 * l + 1], w.rstd[2 * l + 1], ly.ln2_g, ly.ln2_b, M, D, m->cfg.ln_eps, delta, w.xs16[2 * l + 1], P2, n2, w.t[LFC1][l], s, ef);
        memset(&g, 0, sizeof g); g.C = w.a[l]; g.ldc = m->MLP; g.C2 = w.z[l]; g.ldc2 = m->MLP;
        linear_fwd(m, ly.lin[LFC1], w.h2[l], w.t[LFC1][l], Mpad, g, EPI_GELU, s, l * 4 + LFC1, n2 > 0);
        memset(&g, 0, sizeof g); g.C = delta; g.ldc = D;
        if (re2) { g.C = w.xs16[2 * l + 2]; g.R = w.xs16[2 * l + 1]; g.ldr = D; }
        linear_fwd(m, ly.lin[LFC2], w.a[l], w.t[LFC2][l], Mpad, g, re2 ? EPI_RESID_H16 : EPI_STORE_H16, s, l * 4 + LFC2);
    }
    if (!re2) k_layernorm_fwd16(w.xs16[2 * L - 1], nullptr, nullptr, nullptr, nullptr, nullptr, M, D, m->cfg.ln_eps, delta, w.xs16[2 * L], nullptr, 0, nullptr, s, ef);
    k_head_fwd16(w.xs16[2 * L], B, T, D, m->C, m->cfg.ln_eps, m->lnf_g, m->lnf_b, m->flat + m->cls_w_off,
               m->flat + m->cls_b_off, w.xhat, w.xf, w.rstd_f, w.logits, s);
    m->cur_B = B; m->cur_norm = normalise; m->cur_train = train; m->have_loss = 0;
    return VL_OK;
}

int vl_forward(vl_model* m, const float* x, int batch, int normalise, int train, float* logits_out, void* stream) {
    if (!m || !x) return fail(VL_ERR_ARG, "null argument");
    if (!m->ws.max_batch) return fail(VL_ERR_STATE, "no workspace: call vl_plan + vl_set_workspace first");
    hipStream_t s = (hipStream_t)stream;
    int rc = check_async(m);
    if (rc) return rc;
    if (m->dirty && (rc = vl_lora_commit(m, stream))) return rc;     // never run on stale adapter operands
    m->fwd_chains = 0;
    const int b0 = (batch + 1) / 2, b1 = batch - b0;
    if (m->api_chains && !train && !m->f32 && m->chain_batch > 0 && batch >= 2 && b0 <= m->chain_batch &&
        batch <= vl_model::CHAIN_MAX_BATCH && batch <= m->ws.max_batch && !capturing(s) && !g_prof) {
        // two half-batch chains (model.h): chain 0 on the caller's stream, chain 1 on the side stream, logits gathered in the main workspace
        if ((rc = chain_resources(m))) return rc;
        float* const main_logits = m->ws.logits;
        const int64_t img = (int64_t)3 * m->S * m->S;
        HIPCHK(hipEventRecord(m->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
        int rcs[2];
        for (int c = 0; c < 2; ++c) {
            hipStream_t sc = c ? m->side_stream : s;
            const int bc = c ? b1 : b0;
            std::swap(m->ws, m->chain_ws[c]);
            rcs[c] = forward_impl(m, x + (c ? b0 * img : 0), bc, normalise, 0, sc);
            float* const cl = m->ws.logits;
            std::swap(m->ws, m->chain_ws[c]);
            m->chain_B[c] = bc; m->chain_cls[c] = m->cur_cls_only;
            if (!rcs[c]) (void)hipMemcpyAsync(main_logits + (size_t)(c ? b0 : 0) * m->C, cl, (size_t)bc * m->C * sizeof(float), hipMemcpyDeviceToDevice, sc);
        }
        HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
        HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0));
        if (rcs[0] || rcs[1]) { m->cur_B = 0; return rcs[0] ? rcs[0] : rcs[1]; }
        m->cur_B = batch; m->cur_M = batch * m->T; m->cur_norm = normalise; m->cur_train = 0; m->have_loss = 0; m->fwd_chains = 2;
    } else {
        rc = forward_impl(m, x, batch, normalise, train, s);
        if (rc) return rc;
    }
    if (logits_out)
        HIPCHK(hipMemcpyAsync(logits_out, m->ws.logits, (size_t)batch * m->C * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!capturing(s)) return check_launch("vl_forward");
    return VL_OK;
}

int vl_loss_ce(vl_model* m, const int64_t* labels, float* loss_out, void* stream) {
    if (!m || !labels) return fail(VL_ERR_ARG, "null argument");
    if (!m->cur_B) return fail(VL_ERR_STATE, "vl_loss_ce before vl_forward");
    hipStream_t s = (hipStream_t)stream;
    int rc = check_async(m);
    if (rc) return rc;
    k_ce_loss(m->ws.logits, labels, m->cur_B, m->C, m->ws.dlogits, m->ws.loss_img, m->ws.loss, m->err_flag, s);
    if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, m->ws.loss, sizeof(float), hipMemcpyDeviceToDevice, s));
    m->have_loss = 1;
    return VL_OK;
}

// ---- backward --------------------------------------------------------------------------------
// shared dgrad chain; flat_grad != null additionally produces the LoRA / classifier gradients,
// grad_x != null the input gradient.
// K10 fused into the patch-gradient epilogue: the pixel gradient never goes to HBM (vl_pgd_attack; vl_pgd_step stays the ABI entry)
struct PgdFuse { float* adv; const float* x0; float eps, alpha; };

static int backward_impl(vl_model* m, float* grad_x, float* flat_grad, hipStream_t s, const PgdFuse* pf = nullptr) {
    Workspace& w = m->ws;
    if (!m->have_loss) return fail(VL_ERR_STATE, "backward before vl_loss_ce");
    const int B = m->cur_B, D = m->D, L = m->L, T = m->T, MLP = m->MLP, r = m->r;
    const int Mpad = (int)round_up((int64_t)B * T, m->f32 ? 128 : 256), Mppad = (int)round_up((int64_t)B * m->NP, 128);
    const int M = B * T;
    const float sc = m->scaling;
    if (flat_grad && !m->cur_train) return fail(VL_ERR_STATE, "vl_backward_lora needs vl_forward(train=1)");
    if (m->f32) return f32_backward(m, grad_x, flat_grad, s);
    // fp16 gradients: every image's dLoss/dlogits row is scaled by a power of two S_b so that its largest entry lands
    // in [2^9, 2^10) -- the backward chain is linear and per image, so results are exact up to under/overflow and
    // the input gradient of a confidently classified image (dlogits ~ 1e-6) keeps its precision.  With parameter
    // gradients (sums over images) one common scale is used.  Undone in the patch epilogue / the wgrad scale.
    k_grad_scale(w.dlogits, B, m->C, flat_grad ? 1 : 0, w.gscale, w.inv_gscale, s);
    if (flat_grad) {
        // every element is written: the classifier's here, the LoRA matrices' by k_reduce_chunks at the end (modules without an
        // adapter have no elements in the flat buffer)
        k_classifier_grad(w.dlogits, w.xf, B, D, m->C, flat_grad + m->cls_w_off, flat_grad + m->cls_b_off, s);
    }
    const bool cls_only = m->cur_cls_only != 0;
    if (cls_only && flat_grad) return fail(VL_ERR_STATE, "parameter gradients need a train-mode forward");
    if (!cls_only) {
        // the residual-gradient stream (h16, updated in place by every LayerNorm backward; also the A operand of the dgrad GEMMs)
        k_zero(w.dres_h, (size_t)Mpad * D * sizeof(h16), s);       // a kernel node, not a memset node: section 3.3 of DESIGN.md
        k_head_bwd(w.dlogits, w.gscale, m->flat + m->cls_w_off, m->lnf_g, w.xhat, w.rstd_f, B, T, D, m->C, nullptr, w.dres_h, s);
    }

    // LoRA weight gradients of one projection: dy [M,out], x [M,in], t/u [M,kext]
    auto wgrad = [&](const Linear& ln, const h16* dy, const h16* x, const h16* t, const h16* u, uint32_t stream_id) {
        if (!flat_grad || ln.slots.empty()) return;
        if (vl_drop_on(m)) {       // dA sees the dropped branch input: regenerate it (same seed / stream as the forward)
            k_dropout(x, w.xd, (int64_t)M * ln.in, m->drop_seed, stream_id, m->cfg.lora_dropout, s);
            x = w.xd;
        }
        for (const Slot& sl : ln.slots) {
            // dB[n][j] = s * sum_m dy[m][row_off+n] * t[m][ext_off+j]
            k_lora_wgrad(dy + sl.row_off, ln.out, sl.out, t + sl.ext_off, ln.kext, r, M, sc, w.wg_slab + sl.b_off, r, 0,
                         w.inv_gscale, m->cls_w_off, s);
            // dA[j][k] = s * sum_m u[m][ext_off+j] * x[m][k]   (computed transposed: L = x)
            k_lora_wgrad(x, ln.in, sl.in, u + sl.ext_off, ln.kext, r, M, sc, w.wg_slab + sl.a_off, sl.in, 1, w.inv_gscale, m->cls_w_off, s);
        }
    };

    for (int l = L - 1; l >= 0; --l) {
        Layer& ly = m->layers[l];
        GemmArgs g;
        if (cls_only && l == L - 1) {
            // ---- last layer: the gradient lives on the CLS rows until dK / dV of the attention spread it over every token ----
            Workspace::Cls& c = w.c;
            const int Bc = (int)c.Bc;
            m->cur_M = B;
            k_head_bwd(w.dlogits, w.gscale, m->flat + m->cls_w_off, m->lnf_g, w.xhat, w.rstd_f, B, 1, D, m->C, c.dres[0], c.dres_h, s);
            memset(&g, 0, sizeof g); g.C = c.dz; g.ldc = MLP; g.R = c.z; g.ldr = MLP;
            linear_dgrad(m, ly.lin[LFC2], c.dres_h, c.u, Bc, g, EPI_GELU_BWD, s, l * 4 + LFC2);
            memset(&g, 0, sizeof g); g.C = c.dh; g.ldc = D;
            linear_dgrad(m, ly.lin[LFC1], c.dz, c.u, Bc, g, EPI_STORE_H16, s, l * 4 + LFC1);
            const int foc = fused_down(m, ly.lin[LO]);
            k_layernorm_bwd(c.dh, c.x1, c.mean, c.rstd, ly.ln2_g, c.dres[0], c.dres[1], c.dres_h, B, D, ly.lin[LO].Bd, foc, c.u, s, m->err_flag);
            memset(&g, 0, sizeof g); g.C = c.dctx; g.ldc = D;
            linear_dgrad(m, ly.lin[LO], c.dres_h, c.u, Bc, g, EPI_STORE_H16, s, l * 4 + LO, foc > 0);
            m->cur_M = M;
            if (k_attn_cls_bwd(w.qkv[l], c.ctx, c.dctx, c.lse, w.dqkv, B, T, m->H, D, s)) return fail(VL_ERR_UNSUPPORTED, "attention: T > 256");
            memset(&g, 0, sizeof g); g.C = w.dh; g.ldc = D;
            linear_dgrad(m, ly.lin[LQKV], w.dqkv, w.u, Mpad, g, EPI_STORE_H16, s, l * 4 + LQKV);
            // residual gradient entering LN1: zero except the CLS rows
            k_zero(w.dres_h, (size_t)Mpad * D * sizeof(h16), s);
            k_scatter_rows(c.dres[1], w.dres_h, B, D, (int64_t)T * D, s);
            const int ffc = l > 0 ? fused_down(m, m->layers[l - 1].lin[LFC2]) : 0;
            k_layernorm_bwd16(w.dh, w.xs16[2 * l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, w.dres_h,
                              M, D, ffc ? m->layers[l - 1].lin[LFC2].Bd : nullptr, ffc, w.u, s, m->err_flag);
            continue;
        }
        // MLP: dz = (dx2 Wfc2 (+LoRA)) * gelu'(z)
        memset(&g, 0, sizeof g); g.C = w.dz; g.ldc = MLP; g.R = w.z[l]; g.ldr = MLP;
        // u of this dgrad came with dres_h from the LayerNorm backward of the layer above (not for the top layer)
        linear_dgrad(m, ly.lin[LFC2], w.dres_h, w.u, Mpad, g, EPI_GELU_BWD, s, l * 4 + LFC2, l < L - 1 && fused_down(m, ly.lin[LFC2]) > 0);
        wgrad(ly.lin[LFC2], w.dres_h, w.a[l], w.t[LFC2][l], w.u, l * 4 + LFC2);
        memset(&g, 0, sizeof g); g.C = w.dh; g.ldc = D;
        linear_dgrad(m, ly.lin[LFC1], w.dz, w.u, Mpad, g, EPI_STORE_H16, s, l * 4 + LFC1);
        wgrad(ly.lin[LFC1], w.dz, w.h2[l], w.t[LFC1][l], w.u, l * 4 + LFC1);
        const int fo = fused_down(m, ly.lin[LO]);
        k_layernorm_bwd16(w.dh, w.xs16[2 * l + 1], w.mean[2 * l + 1], w.rstd[2 * l + 1], ly.ln2_g, w.dres_h, M, D, ly.lin[LO].Bd, fo, w.u, s,
                          m->err_flag);
        // attention block
        memset(&g, 0, sizeof g); g.C = w.dctx; g.ldc = D;
        linear_dgrad(m, ly.lin[LO], w.dres_h, w.u, Mpad, g, EPI_STORE_H16, s, l * 4 + LO, fo > 0);
        wgrad(ly.lin[LO], w.dres_h, w.ctx[l], w.t[LO][l], w.u, l * 4 + LO);
        const bool img = attn_img(m, B);
        const bool u_qkv = img && down_fusable(m, ly.lin[LQKV]);
        if (img) {
            unsigned mods = 0;
            for (const Slot& sl : ly.lin[LQKV].slots) mods |= 1u << sl.target_idx;     // q, k, v = target 0, 1, 2
            if (k_attention_img_bwd(w.qkv[l], w.ctx[l], w.dctx, w.lse[l], w.dqkv, B, T, m->H, D, u_qkv ? ly.lin[LQKV].Bd : nullptr,
                                    w.u, m->r, mods, s))
                return fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        } else if (k_attention32_bwd(w.qkv[l], w.ctx[l], w.dctx, w.lse[l], w.dqkv, B, T, m->H, D, s))
            return fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        memset(&g, 0, sizeof g); g.C = w.dh; g.ldc = D;
        linear_dgrad(m, ly.lin[LQKV], w.dqkv, w.u, Mpad, g, EPI_STORE_H16, s, l * 4 + LQKV, u_qkv);
        wgrad(ly.lin[LQKV], w.dqkv, w.h1[l], w.t[LQKV][l], w.u, l * 4 + LQKV);
        const int ff = l > 0 ? fused_down(m, m->layers[l - 1].lin[LFC2]) : 0;     // next consumer: fc2 dgrad of the layer below
        k_layernorm_bwd16(w.dh, w.xs16[2 * l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, w.dres_h, M, D,
                          ff ? m->layers[l - 1].lin[LFC2].Bd : nullptr, ff, w.u, s, m->err_flag);
    }
    if (flat_grad && m->cls_w_off > 0)      // the chunk copies summed in chunk order: bit-reproducible LoRA gradients
        k_reduce_chunks(w.wg_slab, flat_grad, m->cls_w_off, lora_wgrad_chunks(M), m->cls_w_off, s);
    if (grad_x || pf) {
        // d(pixels): patch rows of d(x0) times Wpe, scattered back to NCHW, chain rule of (x-mean)/std
        GemmArgs g = gemm_args(w.dres_h, D, m->WpeT, D, D, Mppad, m->PK);
        g.Mvalid = B * m->NP; g.C = grad_x; g.a_gather = 1;
        g.tokens = T; g.patches = m->NP; g.grid = m->G; g.psize = m->P; g.img = m->S;
        for (int c = 0; c < 3; ++c) g.inv_std[c] = m->cur_norm ? 1.f / m->stdv[c] : 1.f;
        g.row_scale = w.inv_gscale;          // per image: undoes the gradient scale
        g.err_flag = m->err_flag;
        if (pf) {                            // sign -> alpha step -> eps projection -> clamp on the gradient in registers
            g.C = pf->adv; g.R = pf->x0; g.pgd_eps = pf->eps; g.pgd_alpha = pf->alpha; g.pgd_lo = 0.f; g.pgd_hi = 1.f;
            launch_gemm(g, EPI_PATCH_PGD, 128, s);
        } else launch_gemm(g, EPI_PATCH_BWD, 128, s);
    }
    return VL_OK;
}

// Synchronises `stream` and reports what the kernels enqueued so far flagged (bad label: VL_ERR_ARG; a gradient that left the
// fp16 range or is NaN: VL_ERR_NONFINITE).  Other entry points report the same at their NEXT call without synchronising.
int vl_check_errors(vl_model* m, void* stream) {
    if (!m) return fail(VL_ERR_ARG, "null model");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(VL_ERR_HIP, "hipStreamSynchronize failed");
    return check_async(m);
}

int vl_set_dlogits(vl_model* m, const float* dlogits, void* stream) {
    if (!m || !dlogits) return fail(VL_ERR_ARG, "null argument");
    if (!m->cur_B) return fail(VL_ERR_STATE, "vl_set_dlogits before vl_forward");
    HIPCHK(hipMemcpyAsync(m->ws.dlogits, dlogits, (size_t)m->cur_B * m->C * sizeof(float), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    m->have_loss = 1;
    return VL_OK;
}

static int backward_api(vl_model* m, float* grad_x, float* flat_grad, hipStream_t s) {
    int rc = check_async(m);
    if (rc) return rc;
    if (m->fwd_chains == 2) {
        // the forward ran as two chains: so does the backward, each from its slice of the whole batch's dLoss/dlogits
        if (flat_grad) return fail(VL_ERR_STATE, "vl_backward_lora needs vl_forward(train=1)");
        if (!m->have_loss) return fail(VL_ERR_STATE, "backward before vl_loss_ce");
        if ((rc = chain_resources(m))) return rc;
        const int batch = m->cur_B, b0 = m->chain_B[0];
        const float* const main_dlogits = m->ws.dlogits;
        const int64_t img = (int64_t)3 * m->S * m->S;
        HIPCHK(hipEventRecord(m->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
        int rcs[2];
        for (int c = 0; c < 2; ++c) {
            hipStream_t sc = c ? m->side_stream : s;
            std::swap(m->ws, m->chain_ws[c]);
            (void)hipMemcpyAsync(m->ws.dlogits, main_dlogits + (size_t)(c ? b0 : 0) * m->C, (size_t)m->chain_B[c] * m->C * sizeof(float),
                                 hipMemcpyDeviceToDevice, sc);
            m->cur_B = m->chain_B[c]; m->cur_cls_only = m->chain_cls[c]; m->have_loss = 1;
            m->cur_M = m->chain_B[c] * m->T;          // Mvalid of this chain's dgrad GEMMs: its own rows (round-4 ADVICE: pad rows of the chain workspace were stored with dead_rows = 0)
            rcs[c] = backward_impl(m, grad_x ? grad_x + (c ? b0 * img : 0) : nullptr, nullptr, sc);
            std::swap(m->ws, m->chain_ws[c]);
        }
        m->cur_B = batch; m->cur_M = batch * m->T;
        HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
        HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0));
        if (rcs[0] || rcs[1]) return rcs[0] ? rcs[0] : rcs[1];
        if (!capturing(s)) return check_launch("vl_backward");
        return VL_OK;
    }
    rc = backward_impl(m, grad_x, flat_grad, s);
    if (rc) return rc;
    if (!capturing(s)) return check_launch("vl_backward");
    return VL_OK;
}

int vl_backward(vl_model* m, float* grad_x_out, float* flat_grad_out, void* stream) {
    if (!m || (!grad_x_out && !flat_grad_out)) return fail(VL_ERR_ARG, "null argument");
    return backward_api(m, grad_x_out, flat_grad_out, (hipStream_t)stream);
}

int vl_set_normalization(vl_model* m, const float mean[3], const float stdv[3]) {
    if (!m || !mean || !stdv) return fail(VL_ERR_ARG, "null argument");
    for (int c = 0; c < 3; ++c) {
        if (!(stdv[c] > 0.f)) return fail(VL_ERR_ARG, "std must be positive");
        m->mean[c] = mean[c]; m->stdv[c] = stdv[c];
    }
    drop_graphs(m);      // mean / std are kernel arguments baked into the captured iteration
    return VL_OK;
}

#ifndef VL_BF16
int vl_channel_affine(float* dst, const float* src, const float scale[3], const float shift[3], int batch, int64_t hw,
                      void* stream) {
    if (!dst || !src || !scale || !shift || batch <= 0 || hw <= 0) return fail(VL_ERR_ARG, "bad argument");
    k_channel_affine(dst, src, scale, shift, batch, hw, (hipStream_t)stream);
    return VL_OK;
}
#endif

int vl_set_dropout_seed(vl_model* m, uint64_t seed) {
    if (!m) return fail(VL_ERR_ARG, "null model");
    m->drop_base = seed; m->drop_calls = 0;
    return VL_OK;
}

// keep-mask (0 or 1/(1-p)) that the last train-mode forward applied to the LoRA branch input of
// projection `proj` (0 = fused qkv, 1 = attention out, 2 = fc1, 3 = fc2) of `layer`: [B*T, in] fp32.
int vl_dropout_mask(vl_model* m, int layer, int proj, float* out, void* stream) {
    if (!m || !out || layer < 0 || layer >= m->L || proj < 0 || proj > 3) return fail(VL_ERR_ARG, "bad argument");
    if (!m->cur_train || !(m->cfg.lora_dropout > 0.f)) return fail(VL_ERR_STATE, "no train-mode forward with dropout");
    k_dropout_mask(out, (int64_t)m->cur_M * m->layers[layer].lin[proj].in, m->drop_seed, (uint32_t)(layer * 4 + proj),
                   m->cfg.lora_dropout, (hipStream_t)stream);
    return VL_OK;
}

int vl_backward_input(vl_model* m, float* grad_x_out, void* stream) {
    if (!m || !grad_x_out) return fail(VL_ERR_ARG, "null argument");
    return backward_api(m, grad_x_out, nullptr, (hipStream_t)stream);
}

int vl_backward_lora(vl_model* m, float* flat_grad_out, void* stream) {
    if (!m || !flat_grad_out) return fail(VL_ERR_ARG, "null argument");
    return backward_api(m, nullptr, flat_grad_out, (hipStream_t)stream);
}

// ---- attacks ---------------------------------------------------------------------------------
#ifndef VL_BF16
int vl_pgd_step(float* adv, const float* x0, const float* grad, float eps, float alpha, float lo, float hi, int64_t n,
                void* stream) {
    if (!adv || !x0 || !grad || n <= 0) return fail(VL_ERR_ARG, "bad argument");
    k_pgd_step(adv, x0, grad, eps, alpha, lo, hi, n, (hipStream_t)stream);
    return VL_OK;
}
#endif

#ifndef VL_BF16
int vl_pgd_init(float* adv, const float* x0, float eps, float lo, float hi, uint64_t seed, int64_t n, void* stream) {
    if (!adv || !x0 || n <= 0) return fail(VL_ERR_ARG, "bad argument");
    k_pgd_init(adv, x0, eps, lo, hi, seed, n, (hipStream_t)stream);
    return VL_OK;
}
#endif

// VITLORA_GRAPH_DUMP=<file>: append the node list of a captured iteration (type, kernel name, grid, block, dynamic LDS)
static void dump_graph(hipGraph_t graph) {
    const char* path = getenv("VITLORA_GRAPH_DUMP");
    if (!path || !path[0]) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) { fclose(f); return; }
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { fclose(f); return; }
    fprintf(f, "graph %zu nodes\n", n);
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) { fprintf(f, "%zu ?\n", i); continue; }
        if (ty == hipGraphNodeTypeKernel) {
            hipKernelNodeParams kp;
            memset(&kp, 0, sizeof kp);
            if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) { fprintf(f, "%zu kernel ?\n", i); continue; }
            const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
            fprintf(f, "%zu kernel %s grid %u,%u,%u block %u lds %u\n", i, nm ? nm : "?", kp.gridDim.x, kp.gridDim.y, kp.gridDim.z,
                    kp.blockDim.x, kp.sharedMemBytes);
        } else if (ty == hipGraphNodeTypeMemset) {
            hipMemsetParams mp;
            memset(&mp, 0, sizeof mp);
            if (hipGraphMemsetNodeGetParams(nodes[i], &mp) != hipSuccess) { fprintf(f, "%zu memset ?\n", i); continue; }
            fprintf(f, "%zu memset dst %p value %u elem %u width %zu height %zu\n", i, mp.dst, mp.value, mp.elementSize, mp.width, mp.height);
        } else {
            fprintf(f, "%zu type %d\n", i, (int)ty);
        }
    }
    {   // dependency edges as index pairs
        size_t ne = 0;
        if (hipGraphGetEdges(graph, nullptr, nullptr, &ne) == hipSuccess && ne) {
            std::vector<hipGraphNode_t> from(ne), to(ne);
            if (hipGraphGetEdges(graph, from.data(), to.data(), &ne) == hipSuccess) {
                auto idx = [&](hipGraphNode_t x) -> long { for (size_t i = 0; i < n; ++i) if (nodes[i] == x) return (long)i; return -1; };
                fprintf(f, "edges %zu:", ne);
                for (size_t e = 0; e < ne; ++e) fprintf(f, " %ld>%ld", idx(from[e]), idx(to[e]));
                fprintf(f, "\n");
            }
        } else fprintf(f, "edges 0\n");
    }
    (void)hipGetLastError();
    fclose(f);
}

static int pgd_iteration(vl_model* m, const float* x0, const int64_t* labels, int B, float eps, float alpha, float* adv,
                         hipStream_t s) {
    int rc = forward_impl(m, adv, B, 1, 0, s);
    if (rc) return rc;
    k_ce_loss(m->ws.logits, labels, B, m->C, m->ws.dlogits, m->ws.loss_img, m->ws.loss, m->err_flag, s);
    m->have_loss = 1;
    if (m->fuse_pgd && !m->f32) {
        const PgdFuse pf = {adv, x0, eps, alpha};
        return backward_impl(m, nullptr, nullptr, s, &pf);
    }
    rc = backward_impl(m, m->ws.grad_img, nullptr, s);
    if (rc) return rc;
    k_pgd_step(adv, x0, m->ws.grad_img, eps, alpha, 0.f, 1.f, (int64_t)B * 3 * m->S * m->S, s, m->err_flag);
    return VL_OK;
}

int vl_pgd_attack(vl_model* m, const float* x0, const int64_t* labels, int batch, float eps, float alpha, int steps,
                  int random_start, uint64_t seed, float* adv_out, void* stream) {
    if (!m || !x0 || !labels || !adv_out) return fail(VL_ERR_ARG, "bad argument");
    if (!m->ws.max_batch) return fail(VL_ERR_STATE, "no workspace");
    if (batch <= 0 || batch > m->ws.max_batch) return fail(VL_ERR_STATE, "batch exceeds planned workspace");
    hipStream_t s = (hipStream_t)stream;
    int rc = check_async(m);
    if (rc) return rc;
    if (m->dirty && (rc = vl_lora_commit(m, stream))) return rc;     // the attack sees the CURRENT adapters
    Workspace& w = m->ws;
    const int64_t n = (int64_t)batch * 3 * m->S * m->S;
    // Persistent staging: the captured iteration only ever reads / writes the workspace's x0 / labels / adv buffers,
    // so ONE executable graph per (batch, eps, alpha) serves every batch of a run whatever tensors the caller passes
    // (3 device copies of the image batch per attack: < 0.1 % of a PGD-20 attack).
    HIPCHK(hipMemcpyAsync(w.stage_x0, x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(w.stage_labels, labels, (size_t)batch * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (random_start) k_pgd_init(w.stage_adv, w.stage_x0, eps, 0.f, 1.f, seed, n, s);
    else HIPCHK(hipMemcpyAsync(w.stage_adv, w.stage_x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    // two half-batch chains as parallel branches of the captured iteration (model.h): by batch size, or as "pgd_chains" says
    float* const sx0 = w.stage_x0; float* const sadv = w.stage_adv; int64_t* const slab = w.stage_labels;    // the MAIN staging buffers
    const int b0 = (batch + 1) / 2, b1 = batch - b0;
    const bool chain_fits = m->chain_batch > 0 && batch >= 2 && b0 <= m->chain_batch;
    const int chains = (chain_fits && (m->pgd_chains == 2 || (m->pgd_chains == 0 && batch <= vl_model::CHAIN_MAX_BATCH))) ? 2 : 1;
    // one PGD iteration of the whole batch, enqueued on s0 (chains == 2: second half on s1 between a fork and a join event)
    auto iteration = [&](hipStream_t s0, hipStream_t s1) -> int {
        if (chains == 1) return pgd_iteration(m, sx0, slab, batch, eps, alpha, sadv, s0);
        const int64_t off = (int64_t)b0 * 3 * m->S * m->S;
        HIPCHK(hipEventRecord(m->ev_fork, s0));
        HIPCHK(hipStreamWaitEvent(s1, m->ev_fork, 0));
        int rc0, rc1;
        std::swap(m->ws, m->chain_ws[0]);            // chain 0's activations become "the" workspace while its kernels are enqueued
        rc0 = pgd_iteration(m, sx0, slab, b0, eps, alpha, sadv, s0);
        std::swap(m->ws, m->chain_ws[0]);
        std::swap(m->ws, m->chain_ws[1]);
        rc1 = pgd_iteration(m, sx0 + off, slab + b0, b1, eps, alpha, sadv + off, s1);
        std::swap(m->ws, m->chain_ws[1]);
        HIPCHK(hipEventRecord(m->ev_join, s1));
        HIPCHK(hipStreamWaitEvent(s0, m->ev_join, 0));
        m->cur_B = 0;                 // the handle holds no forward of the whole batch: a backward call needs its own forward
        return rc0 ? rc0 : rc1;
    };
    m->fwd_chains = 0;
    if (chains == 2 && (rc = chain_resources(m))) return rc;
    if (steps > 0) {
        if (!m->use_graph || g_prof || g_poison_lds) {
            for (int i = 0; i < steps; ++i)
                if ((rc = iteration(s, m->side_stream))) return rc;
            if ((rc = check_launch("vl_pgd_attack"))) return rc;
        } else {
            hipGraphExec_t exec = nullptr, exec1 = nullptr;
            for (GraphEntry& g : m->graphs)
                if (g.B == batch && g.eps == eps && g.alpha == alpha && g.chains == chains) { exec = g.exec; exec1 = g.exec1; break; }
            if (!exec) {
                // Captured cold: nothing has to run eagerly first.  (Round 2 ran the first iteration eagerly because replays of
                // a graph captured in a fresh process differed from the eager result.  Cause, established in round 3 with
                // tools/cold_capture_diag.py: the two hipMemsetAsync nodes of the backward did not take effect on replays when
                // they were captured before the runtime's fill kernel had ever run; the first launch passed only because the
                // workspace was still zero.  They are k_zero kernel nodes now: profiles/r03_cold_capture_*.txt.)
                // capture on a private stream (the caller's may be the legacy default stream, which cannot
                // capture); nothing executes during capture, the graph is launched on the caller's stream.
                if (!m->cap_stream) HIPCHK(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
                // one captured iteration per chain: chains == 2 gives two graphs, each over its half of the staged batch and its
                // own activation workspace, replayed on two streams that meet only at the start and the end of the attack
                auto capture = [&](int c, hipGraphExec_t* out) -> int {
                    hipGraph_t graph = nullptr;
                    (void)hipGetLastError();
                    HIPCHK(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
                    int rcc;
                    if (chains == 1) rcc = pgd_iteration(m, sx0, slab, batch, eps, alpha, sadv, m->cap_stream);
                    else {
                        const int64_t off = c ? (int64_t)b0 * 3 * m->S * m->S : 0;
                        std::swap(m->ws, m->chain_ws[c]);
                        rcc = pgd_iteration(m, sx0 + off, slab + (c ? b0 : 0), c ? b1 : b0, eps, alpha, sadv + off, m->cap_stream);
                        std::swap(m->ws, m->chain_ws[c]);
                        m->cur_B = 0;         // the handle holds no forward of the whole batch
                    }
                    hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
                    if (rcc) { if (graph) (void)hipGraphDestroy(graph); return rcc; }
                    if (e != hipSuccess) return fail(VL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
                    dump_graph(graph);
                    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(graph);
                    if (e != hipSuccess) return fail(VL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
                    return VL_OK;
                };
                if ((rc = capture(0, &exec))) return rc;
                if (chains == 2 && (rc = capture(1, &exec1))) { (void)hipGraphExecDestroy(exec); return rc; }
                if (m->graphs.size() >= 8) {
                    (void)hipGraphExecDestroy(m->graphs.front().exec);
                    if (m->graphs.front().exec1) (void)hipGraphExecDestroy(m->graphs.front().exec1);
                    m->graphs.erase(m->graphs.begin());
                }
                m->graphs.push_back({batch, eps, alpha, exec, chains, exec1});
                m->n_captures++;
            }
            if (chains == 2) {
                HIPCHK(hipEventRecord(m->ev_fork, s));
                HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
            }
            for (int i = 0; i < steps; ++i) {
                HIPCHK(hipGraphLaunch(exec, s));
                if (chains == 2) HIPCHK(hipGraphLaunch(exec1, m->side_stream));
            }
            if (chains == 2) {
                HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
                HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0));
            }
        }
    }
    HIPCHK(hipMemcpyAsync(adv_out, w.stage_adv, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return VL_OK;
}

#ifndef VL_BF16
int vl_adam_step(float* param, const float* grad, float* m1, float* m2, float lr, float b1, float b2, float eps, int t,
                 int64_t n, void* stream) {
    if (!param || !grad || !m1 || !m2 || n <= 0 || t <= 0) return fail(VL_ERR_ARG, "bad argument");
    int* err = nullptr;
    for (const VlFlatRecord& mm : g_models)      // optimizer.step() on a model's flat parameters: its operands are stale now
        if (param < mm.flat + mm.flat_n && param + n > mm.flat) { *mm.dirty = 1; err = *mm.err_flag; }
    k_adam(param, grad, m1, m2, lr, b1, b2, eps, t, n, (hipStream_t)stream, err);
    return VL_OK;
}
#endif

#ifndef VL_BF16
int vl_quantize_u8(const float* images, uint8_t* out_hwc, int batch, int channels, int height, int width, void* stream) {
    if (!images || !out_hwc) return fail(VL_ERR_ARG, "null argument");
    k_quantize(images, out_hwc, batch, channels, height, width, (hipStream_t)stream);
    return VL_OK;
}
#endif

// ---- adversarial patch (patch_attack.py: ART AdversarialPatchPyTorch) ------------------------------------
#ifndef VL_BF16
static int patch_args_ok(int batch, int image_size, int patch_size, int patch_type) {
    if (batch <= 0 || image_size <= 0 || patch_size <= 0 || patch_size > 64 || patch_size > image_size)
        return fail(VL_ERR_UNSUPPORTED, "patch_size must be in [1, min(64, image_size)]");
    if (patch_type != 0 && patch_type != 1) return fail(VL_ERR_ARG, "patch_type: 0 = square, 1 = circle");
    return VL_OK;
}
int vl_patch_apply_persp(const float* images, const float* patch, const float* inv_affine, const float* persp, int batch,
                         int image_size, int patch_size, int patch_type, float* out, void* stream) {
    if (!images || !patch || !inv_affine || !out || out == images) return fail(VL_ERR_ARG, "bad argument");
    if (int rc = patch_args_ok(batch, image_size, patch_size, patch_type)) return rc;
    k_patch_overlay(images, patch, inv_affine, persp, out, batch, image_size, patch_size, patch_type, (hipStream_t)stream);
    if (!capturing((hipStream_t)stream)) return check_launch("vl_patch_apply");
    return VL_OK;
}
int vl_patch_apply(const float* images, const float* patch, const float* inv_affine, int batch, int image_size, int patch_size,
                   int patch_type, float* out, void* stream) {
    return vl_patch_apply_persp(images, patch, inv_affine, nullptr, batch, image_size, patch_size, patch_type, out, stream);
}
int vl_patch_grad_persp(const float* grad_out, const float* inv_affine, const float* persp, int batch, int image_size,
                        int patch_size, int patch_type, float* patch_grad, void* stream) {
    if (!grad_out || !inv_affine || !patch_grad) return fail(VL_ERR_ARG, "bad argument");
    if (int rc = patch_args_ok(batch, image_size, patch_size, patch_type)) return rc;
    k_patch_overlay_bwd(grad_out, inv_affine, persp, patch_grad, batch, image_size, patch_size, patch_type, (hipStream_t)stream);
    if (!capturing((hipStream_t)stream)) return check_launch("vl_patch_grad");
    return VL_OK;
}
int vl_patch_grad(const float* grad_out, const float* inv_affine, int batch, int image_size, int patch_size, int patch_type,
                  float* patch_grad, void* stream) {
    return vl_patch_grad_persp(grad_out, inv_affine, nullptr, batch, image_size, patch_size, patch_type, patch_grad, stream);
}
#endif

#ifndef VL_BF16
int vl_clamp(float* x, float lo, float hi, int64_t n, void* stream) {
    if (!x || n <= 0 || !(lo <= hi)) return fail(VL_ERR_ARG, "bad argument");
    k_clamp(x, lo, hi, n, (hipStream_t)stream);
    return VL_OK;
}
#endif

// ---- GEMM micro-benchmark (tools/gemm_sweep.py): random h16 operands, HIP-event timing -----
#ifndef VL_BF16
int vl_bench_gemm(int M, int N, int K1, int K2, int epi, int bn, int iters, float* ms_out) {
    if (M % 128 || N % 64 || K1 % 64 || K2 % 64 || iters <= 0 || !ms_out) return fail(VL_ERR_ARG, "bad argument");
    { int dev = 0; HIPCHK(hipGetDevice(&dev)); if (gemm_init(dev)) return fail(VL_ERR_HIP, "gemm_init failed"); }
    h16 *A = nullptr, *W = nullptr, *A2 = nullptr, *W2 = nullptr, *C = nullptr, *C2 = nullptr;
    float *R = nullptr, *bias = nullptr;
    const size_t nA = (size_t)M * K1, nW = (size_t)N * K1, nC = (size_t)M * N;
    HIPCHK(hipMalloc(&A, nA * 2)); HIPCHK(hipMalloc(&W, nW * 2));
    HIPCHK(hipMalloc(&A2, (size_t)M * 64 * 2 + 256)); HIPCHK(hipMalloc(&W2, (size_t)N * 64 * 2 + 256));
    HIPCHK(hipMalloc(&C, nC * 4)); HIPCHK(hipMalloc(&C2, nC * 2)); HIPCHK(hipMalloc(&R, nC * 4));
    HIPCHK(hipMalloc(&bias, (size_t)N * 4));
    k_fill_random_h16(A, nA, 1, 0); k_fill_random_h16(W, nW, 2, 0);
    k_fill_random_h16(A2, (size_t)M * 64, 3, 0); k_fill_random_h16(W2, (size_t)N * 64, 4, 0);
    k_fill_random_h16((h16*)R, nC * 2, 5, 0); k_fill_random_h16((h16*)bias, (size_t)N * 2, 6, 0);
    GemmArgs g = gemm_args(A, K1, W, K1, K1, M, N);
    if (K2) add_ext(g, A2, K2, W2, K2, K2);
    g.bias = bias; g.C = C; g.ldc = N; g.C2 = C2; g.ldc2 = N; g.R = R; g.ldr = N;
    h16* Wd = nullptr;
    if (epi == 200 || epi == 300) {        // store_h16 with the LoRA down projection (16 / 32 columns) inside the ping-pong kernel
        const int nd = epi / 100 - 1;
        epi = EPI_STORE_H16;
        HIPCHK(hipMalloc(&Wd, (size_t)64 * K1 * 2 + 256));
        k_fill_random_h16(Wd, (size_t)64 * K1, 7, 0);
        g.A2 = nullptr; g.down_W = Wd; g.down_ldw = K1; g.down_out = A2; g.down_ld = 64; g.down_groups = nd;
        if (!gemm_pp_fuses_down(g, epi)) return fail(VL_ERR_UNSUPPORTED, "shape not fusable");
    }
    if (epi >= 100) { epi -= 100; g.ldc = 0; g.ldc2 = 0; }     // diagnostic: every row stored to row 0 (no HBM write stream)
    if (epi == EPI_RESID_F32) g.R = C;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_gemm(g, epi, bn, 0);
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_gemm(g, epi, bn, 0);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(A2); (void)hipFree(W2); (void)hipFree(C); (void)hipFree(C2);
    (void)hipFree(R); (void)hipFree(bias); (void)hipFree(Wd);
    return VL_OK;
}
#endif

// ---- GEMM self-check (tests/test_hip_engine.py): the same random GEMM through the 128-row kernel (the oldest, simplest
// form) and through whichever kernel launch_gemm selects for `pp_mode`; returns the largest |difference| over C (and C2) --
#ifndef VL_BF16
int vl_check_gemm(int M, int N, int K1, int K2, int epi, int pp_mode, float* max_diff) {
    if (M % 128 || N % 256 || K1 % 64 || K2 % 64 || !max_diff || pp_mode < 0 || pp_mode > 3) return fail(VL_ERR_ARG, "bad argument");
    const int nd = pp_mode >= 2 ? pp_mode - 1 : 0;           // fused LoRA down projection with 16 nd columns
    if (nd && (K2 != 64 || !(epi == EPI_STORE_H16 || epi == EPI_RESID_H16))) return fail(VL_ERR_ARG, "fused down: K2 = 64, epi = store_h16 or resid_h16");
    const bool bias_col = nd && epi == EPI_RESID_H16;        // the fused kernel takes the bias from column 63 of W2 (gemm_pp.hip, BC)
    { int dev = 0; HIPCHK(hipGetDevice(&dev)); if (gemm_init(dev)) return fail(VL_ERR_HIP, "gemm_init failed"); }
    h16 *A = nullptr, *W = nullptr, *A2 = nullptr, *W2 = nullptr, *Wd = nullptr, *T2 = nullptr;
    float *R = nullptr, *bias = nullptr;
    char *C[2] = {nullptr, nullptr}, *C2[2] = {nullptr, nullptr};
    struct Guard {                                   // frees the scratch operands and restores the ping-pong switch on every path
        std::vector<void**> ptrs; int pp_keep;
        ~Guard() { for (void** q : ptrs) if (*q) (void)hipFree(*q); gemm_pp_set_mode(pp_keep); }
    } guard{{(void**)&A, (void**)&W, (void**)&A2, (void**)&W2, (void**)&Wd, (void**)&T2, (void**)&R, (void**)&bias,
             (void**)&C[0], (void**)&C[1], (void**)&C2[0], (void**)&C2[1]}, gemm_pp_mode()};
    const size_t nA = (size_t)M * K1, nW = (size_t)N * K1, nC = (size_t)M * N;
    HIPCHK(hipMalloc(&A, nA * 2)); HIPCHK(hipMalloc(&W, nW * 2));
    HIPCHK(hipMalloc(&A2, (size_t)M * 64 * 2 + 256)); HIPCHK(hipMalloc(&W2, (size_t)N * 64 * 2 + 256));
    HIPCHK(hipMalloc(&T2, (size_t)M * 64 * 2 + 256)); HIPCHK(hipMalloc(&Wd, (size_t)64 * K1 * 2 + 256));
    HIPCHK(hipMalloc(&R, nC * 4)); HIPCHK(hipMalloc(&bias, (size_t)N * 4));
    for (int i = 0; i < 2; ++i) { HIPCHK(hipMalloc(&C[i], nC * 4)); HIPCHK(hipMalloc(&C2[i], nC * 2)); }
    k_fill_random_h16(A, nA, 1, 0); k_fill_random_h16(W, nW, 2, 0);
    k_fill_random_h16(A2, (size_t)M * 64, 3, 0); k_fill_random_h16(W2, (size_t)N * 64, 4, 0);
    k_fill_random_h16((h16*)R, nC * 2, 5, 0); k_fill_random_h16((h16*)bias, (size_t)N * 2, 6, 0);
    HIPCHK(hipMemset(Wd, 0, (size_t)64 * K1 * 2)); HIPCHK(hipMemset(T2, 0, (size_t)M * 64 * 2));
    if (bias_col) {
        // a bias that fp16 holds exactly, in both places: the fp32 vector the reference route adds and column 63 of W2
        std::vector<float> hb(N);
        std::vector<h16> hw((size_t)N * 64);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(hw.data(), W2, hw.size() * 2, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) { const h16 v = (h16)(0.03125f * (float)((n * 7) % 37 - 18)); hb[n] = (float)v; hw[(size_t)n * 64 + 63] = v; }
        HIPCHK(hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(W2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    if (nd) {
        k_fill_random_h16(Wd, (size_t)(16 * nd - 8) * K1, 7, 0);          // 8 / 24 rows in use, like r = 8 on one / three modules
        GemmArgs d = gemm_args(A, K1, Wd, K1, K1, M, 64);                   // reference t through the skinny GEMM
        d.C = A2; d.ldc = 64;
        launch_gemm(d, EPI_STORE_H16, 64, 0);
    }
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipMemset(C[i], 0, nC * 4)); HIPCHK(hipMemset(C2[i], 0, nC * 2));
        GemmArgs g = gemm_args(A, K1, W, K1, K1, M, N);
        if (K2) add_ext(g, A2, K2, W2, K2, K2);
        g.bias = bias; g.C = C[i]; g.ldc = N; g.C2 = C2[i]; g.ldc2 = N; g.R = R; g.ldr = N;
        if (epi == EPI_RESID_F32) { HIPCHK(hipMemcpy(C[i], R, nC * 4, hipMemcpyDeviceToDevice)); g.R = C[i]; }
        gemm_pp_set_mode(i == 0 ? 0 : (pp_mode ? 1 : 0));
        if (i == 1 && nd) {
            g.A2 = nullptr; g.down_W = Wd; g.down_ldw = K1; g.down_out = T2; g.down_ld = 64; g.down_groups = nd;
            if (bias_col) { g.ones_col = 1; g.bias = nullptr; }
            if (!gemm_pp_fuses_down(g, epi)) return fail(VL_ERR_UNSUPPORTED, "shape not fusable");
        }
        const int keep_small = gemm_force_small(i == 0 ? 1 : 0);
        launch_gemm(g, epi, 128, 0);
        gemm_force_small(keep_small);
    }
    HIPCHK(hipDeviceSynchronize());
    const bool f32out = epi == EPI_RESID_F32 || epi == EPI_STORE_F32;
    const size_t bytes = nC * (f32out ? 4 : 2);
    std::vector<char> h0(bytes), h1(bytes);
    HIPCHK(hipMemcpy(h0.data(), C[0], bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h1.data(), C[1], bytes, hipMemcpyDeviceToHost));
    double md = 0.0;
    auto cmp = [&](const char* a, const char* b, bool f32, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            const double x = f32 ? ((const float*)a)[i] : (double)(float)((const h16*)a)[i];
            const double y = f32 ? ((const float*)b)[i] : (double)(float)((const h16*)b)[i];
            const double d = x > y ? x - y : y - x;
            if (!(d <= md)) md = d == d ? d : 1e30;
        }
    };
    cmp(h0.data(), h1.data(), f32out, nC);
    if (epi == EPI_GELU) {
        HIPCHK(hipMemcpy(h0.data(), C2[0], nC * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h1.data(), C2[1], nC * 2, hipMemcpyDeviceToHost));
        cmp(h0.data(), h1.data(), false, nC);
    }
    if (nd) {          // the t the fused kernel wrote against the skinny GEMM's
        std::vector<char> t0((size_t)M * 64 * 2), t1((size_t)M * 64 * 2);
        HIPCHK(hipMemcpy(t0.data(), A2, t0.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(t1.data(), T2, t1.size(), hipMemcpyDeviceToHost));
        cmp(t0.data(), t1.data(), false, (size_t)M * 64);
    }
    *max_diff = (float)md;
    return VL_OK;
}
#endif

// persistent-grid size of the GEMM kernels (and the batch threshold of the per-image attention form of `m`, if given):
// experiments with two half-batch chains on disjoint halves of the chip (tools/dual_chain_probe.py)
int vl_debug_set_cus(vl_model* m, int cus) {
    if (cus <= 0) return fail(VL_ERR_ARG, "cus must be positive");
    gemm256_set_cus(cus); gemm_pp_set_cus(cus);
    if (m) m->num_cus = cus;
    return VL_OK;
}

// Diagnostic switches of a handle (tests, A/B timing); every change drops the cached PGD graphs.
//   "dead_rows" 1 (default): eval-mode forwards run the last layer on the CLS rows only; 0: every row of every layer
//   "fuse_pgd"  1 (default): vl_pgd_attack applies the PGD step inside the patch-gradient GEMM epilogue; 0: gradient to HBM + K10
//   "resid_epi" 1 (default): the residual add of the 16-bit stream sits in the o / fc2 GEMM epilogue; 0: in the LayerNorm after it
//   "attn_ring" 1 (default): single-pass per-image attention backward; 0: the two-phase form (process-wide switch)
//   "pgd_chains" 0 (default): vl_pgd_attack runs batches of 2 .. 191 images as two half-batch chains in parallel branches of the
//               captured iteration; 1: one chain always; 2: two chains whenever the chain workspaces hold the halves
//   "api_chains" 0 (default); 1: vl_forward(train = 0) and the backward after it run batches of 2 .. 191 images as the same two
//               chains (the adversarial-patch EoT step uses these calls); vl_debug_tensor then does not see the activations
//   "poison_lds" 0 (default); 1: every profiled launch is preceded by a kernel that fills every CU's LDS with NaN patterns (test hook)
int vl_debug_set_option(vl_model* m, const char* name, int value) {
    if (!m || !name) return fail(VL_ERR_ARG, "null argument");
    if (!strcmp(name, "dead_rows")) m->dead_rows = value ? 1 : 0;
    else if (!strcmp(name, "fuse_pgd")) m->fuse_pgd = value ? 1 : 0;
    else if (!strcmp(name, "resid_epi")) m->resid_epi = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(name, "attn_ring")) attention32_set_ring(value);        // process-wide
    else if (!strcmp(name, "poison_lds")) g_poison_lds = value ? 1 : 0;      // process-wide test hook (prof.h); attacks then run eagerly
    else if (!strcmp(name, "pgd_chains")) m->pgd_chains = value < 0 ? 0 : value > 2 ? 2 : value;    // 1: set BEFORE vl_plan to save the chain workspaces
    else if (!strcmp(name, "api_chains")) m->api_chains = value ? 1 : 0;      // vl_forward(train = 0) / vl_backward_input as two half-batch chains (2 .. 191 images)
    else return fail(VL_ERR_ARG, "unknown option %s", name);
    drop_graphs(m);
    return VL_OK;
}

#ifndef VL_BF16
int vl_debug_set_gemm_pp(int mode) { const int old = gemm_pp_mode(); gemm_pp_set_mode(mode); return old; }
#endif
#ifndef VL_BF16
int vl_debug_set_gemm_stream(int mode) { return gemm_stream_set_mode(mode); }
#endif

// ---- profiling -------------------------------------------------------------------------------
#ifndef VL_BF16
int vl_profile_begin(void) {
    if (g_prof) return fail(VL_ERR_STATE, "profile already active");
    g_prof = new Profiler();
    return VL_OK;
}
#endif

// Synchronises the device, aggregates per kernel name and writes one JSON object:
// {"name": {"n": launches, "ms": total_ms, "flops": total, "bytes": total, "exec_flops": total}, ...}
// flops / bytes are ALGORITHMIC; exec_flops are the FLOPs the launches issue to the matrix pipes (padded rows,
// padded attention tiles, the whole LoRA K tile) -- bench.py's roofline.path.executed_frac
#ifndef VL_BF16
int vl_profile_report(char* buf, size_t cap) {
    if (!g_prof) return fail(VL_ERR_STATE, "profile not active");
    Profiler* p = g_prof;
    g_prof = nullptr;
    if (hipDeviceSynchronize() != hipSuccess) { delete p; return fail(VL_ERR_HIP, "hipDeviceSynchronize failed"); }
    struct Agg { std::string name; int n; double ms, flops, bytes, exec; };
    std::vector<Agg> agg;
    for (ProfRecord& r : p->recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
        Agg* a = nullptr;
        for (Agg& x : agg) if (x.name == r.name) { a = &x; break; }
        if (!a) { agg.push_back({r.name, 0, 0, 0, 0, 0}); a = &agg.back(); }
        a->n++; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes; a->exec += r.exec_flops;
    }
    delete p;
    std::string out = "{";
    for (size_t i = 0; i < agg.size(); ++i) {
        char line[320];
        snprintf(line, sizeof line, "%s\"%s\": {\"n\": %d, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"exec_flops\": %.6e}",
                 i ? ", " : "", agg[i].name.c_str(), agg[i].n, agg[i].ms, agg[i].flops, agg[i].bytes, agg[i].exec);
        out += line;
    }
    out += "}";
    if (!buf || cap < out.size() + 1) return fail(VL_ERR_ARG, "buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return VL_OK;
}
#endif

int vl_debug_counter(vl_model* m, const char* what, int64_t* value) {
    if (!m || !what || !value) return fail(VL_ERR_ARG, "null argument");
    if (!strcmp(what, "graph_captures")) { *value = m->n_captures; return VL_OK; }
    if (!strcmp(what, "commits")) { *value = m->n_commits; return VL_OK; }
    if (!strcmp(what, "dirty")) { *value = m->dirty; return VL_OK; }
    if (!strcmp(what, "lds_poisons")) { *value = g_poison_count; return VL_OK; }       // launches of the "poison_lds" test hook (process-wide)
    return fail(VL_ERR_ARG, "unknown counter %s", what);
}

int vl_debug_tensor(vl_model* m, const char* what, int layer, void** ptr, int64_t* numel, int* dtype) {
    // dtype: 0 = f32, 1 = fp16, 2 = bf16
    if (!m || !what || !ptr || !numel || !dtype) return fail(VL_ERR_ARG, "null argument");
    Workspace& w = m->ws;
    if (!w.max_batch) return fail(VL_ERR_STATE, "no workspace");
    const int64_t MD = (int64_t)m->cur_B * m->T * m->D;
    {   // backward-side buffers (shared across layers: they hold what the LAST executed kernel sequence left) and the head
        const int64_t Mp = w.Mpad, img = (int64_t)m->cur_B * 3 * m->S * m->S;
        struct { const char* name; void* p; int64_t n; int dt; } extra[] = {
            {"dres0", w.dres[0], Mp * m->D, 0}, {"dres1", w.dres[1], Mp * m->D, 0}, {"grad_img", w.grad_img, img, 0},   // fp32 mode only
            {"stage_adv", w.stage_adv, img, 0}, {"stage_x0", w.stage_x0, img, 0},
            {"logits", w.logits, (int64_t)m->cur_B * m->C, 0}, {"dlogits", w.dlogits, (int64_t)m->cur_B * m->C, 0},
            {"gscale", w.gscale, m->cur_B, 0}, {"inv_gscale", w.inv_gscale, m->cur_B, 0}, {"loss_img", w.loss_img, m->cur_B, 0},
            {"xhat", w.xhat, (int64_t)m->cur_B * m->D, 0}, {"rstd_f", w.rstd_f, m->cur_B, 0},
            {"dres_h", m->f32 ? nullptr : w.dres_h, Mp * m->D, VL_DT16}, {"dh", m->f32 ? nullptr : w.dh, Mp * m->D, VL_DT16},
            {"dctx", m->f32 ? nullptr : w.dctx, Mp * m->D, VL_DT16}, {"dqkv", m->f32 ? nullptr : w.dqkv, Mp * 3 * m->D, VL_DT16},
            {"dz", m->f32 ? nullptr : w.dz, Mp * m->MLP, VL_DT16}, {"u", m->f32 ? nullptr : w.u, Mp * 64, VL_DT16},
            {"cls_x1", m->f32 ? nullptr : w.c.x1, (int64_t)m->cur_B * m->D, 0}, {"cls_x2", m->f32 ? nullptr : w.c.x2, (int64_t)m->cur_B * m->D, 0},
            {"cls_ctx", m->f32 ? nullptr : w.c.ctx, (int64_t)m->cur_B * m->D, VL_DT16}};
        for (auto& e : extra)
            if (!strcmp(what, e.name)) {
                if (!e.p) return fail(VL_ERR_UNSUPPORTED, "%s: not in this precision mode", what);
                *ptr = e.p; *numel = e.n; *dtype = e.dt; return VL_OK;
            }
    }
    if (!strcmp(what, "xs")) {       // residual-stream snapshots: fp32 in the fp32 mode, h16 on the 16-bit path
        if (layer < 0 || layer > 2 * m->L) return fail(VL_ERR_ARG, "index");
        *ptr = m->f32 ? (void*)w.xs[layer] : (void*)w.xs16[layer]; *numel = MD; *dtype = m->f32 ? 0 : VL_DT16; return VL_OK;
    }
    if (layer < 0 || layer >= m->L) return fail(VL_ERR_ARG, "layer out of range");
    if (m->f32) {
        if (!strcmp(what, "qkv")) { *ptr = w.f_qkv[layer]; *numel = 3 * MD; *dtype = 0; return VL_OK; }
        if (!strcmp(what, "ctx")) { *ptr = w.f_ctx[layer]; *numel = MD; *dtype = 0; return VL_OK; }
        if (!strcmp(what, "z")) { *ptr = w.f_z[layer]; *numel = (int64_t)m->cur_B * m->T * m->MLP; *dtype = 0; return VL_OK; }
    } else {
        if (!strcmp(what, "qkv")) { *ptr = w.qkv[layer]; *numel = 3 * MD; *dtype = VL_DT16; return VL_OK; }
        if (!strcmp(what, "ctx")) { *ptr = w.ctx[layer]; *numel = MD; *dtype = VL_DT16; return VL_OK; }
        if (!strcmp(what, "z")) { *ptr = w.z[layer]; *numel = (int64_t)m->cur_B * m->T * m->MLP; *dtype = VL_DT16; return VL_OK; }
    }
    if (!strcmp(what, "lse")) { *ptr = w.lse[layer]; *numel = (int64_t)m->cur_B * m->H * m->T; *dtype = 0; return VL_OK; }
    return fail(VL_ERR_ARG, "unknown debug tensor %s", what);
}

}  // extern "C"

}  // namespace VLNS
