// h16 MFMA GEMM for gfx950 with fused epilogues and the LoRA rank-r update appended as
// extra K tiles.  Stands in for the cuBLAS/MKL matmuls that HF ViT + peft issue on the
// reference's hot path (whitebox_attacks.py:27, train_loras.py:310; SURVEY.md K2,K4,K6-K8).
//
// This file: the 128 x BN x 64 kernel (4 waves, 2x2), used for the skinny LoRA-down GEMMs
// (BN = 64), for small problems and as the fallback of the 256-row kernel in gemm256.hip.
// Direct-to-LDS 16-byte loads (global_load_lds_dwordx4) into a double-buffered, XOR-swizzled
// row-major image (rows of 128 B; 16-byte chunk c of row r stored at chunk c ^ (r & 7):
// conflict-free ds_read_b128 fragment reads), one barrier per K step, 16x16x32 MFMA with the
// operands swapped so that each lane owns 4 consecutive output columns (8/16-byte stores).
#include <cstdio>
#include <cstdlib>

#include "gemm_epi.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

void launch_gemm256(const GemmArgs& a, int epi, hipStream_t s);   // gemm256.hip
bool gemm256_supports(const GemmArgs& a, int epi);
int gemm256_init();
void launch_gemm_pp(const GemmArgs& a, int epi, hipStream_t s);    // gemm_pp.hip
bool gemm_pp_supports(const GemmArgs& a, int epi);
int gemm_pp_init();
void launch_gemm_stream(const GemmArgs& a, int epi, int bn, hipStream_t s);   // gemm_stream.hip
bool gemm_stream_supports(const GemmArgs& a, int epi, int bn);
int gemm_stream_init();

namespace {

constexpr int BK = 64;

// STAGES = LDS ring depth: 2 (one K step ahead, drained every step) or 4 (three K steps of
// direct-to-LDS loads in flight behind a counted vmcnt: the skinny LoRA-down products are pure
// HBM streams and need the bytes in flight, not the MFMA rate).
template <int BM, int BN, int EPI, int STAGES = 2>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sA = (h16*)smem;                   // [STAGES][BM][BK]
    h16* sW = sA + STAGES * BM * BK;         // [STAGES][BN][BK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesN = p.N / BN;
    const int ntiles = (p.M / BM) * tilesN;
    const int id = xcd_remap(blockIdx.x, ntiles);
    const int bm = id / tilesN, bn = id - bm * tilesN;
    const int wm = w >> 1, wn = w & 1;
    constexpr int NJ = BN / 32;          // 16-wide column tiles per wave
    constexpr int MI = BM / 32;          // 16-high row tiles per wave
    constexpr int WG = BN / 32;          // 8-row load groups of the W tile per wave
    constexpr int AG = BM / 32;          // 8-row load groups of the A tile per wave

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk1 = p.K1 / BK;
    const int nk = nk1 + p.K2 / BK;
    const int lr = lane >> 3;            // row inside an 8-row load group
    const int lc = lane & 7;             // 16-byte chunk position inside the 128-byte row

    // per-lane A rows of the 4 load groups (fixed over the K loop)
    size_t arow[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        int m = bm * BM + (w * AG + i) * 8 + lr;
        if (p.a_gather) m = m < p.Mvalid ? m + m / p.patches + 1 : 0;
        arow[i] = (size_t)m;
    }

    auto issue = [&](int kt, int buf) {
        const h16* Ap; const h16* Wp; int lda, ldw, k0;
        if (kt < nk1) { Ap = p.A1; Wp = p.W1; lda = p.lda1; ldw = p.ldw1; k0 = kt * BK; }
        else          { Ap = p.A2; Wp = p.W2; lda = p.lda2; ldw = p.ldw2; k0 = (kt - nk1) * BK; }
        h16* dA = sA + buf * BM * BK;
        h16* dW = sW + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int g = w * AG + i;
            const int r = g * 8 + lr;
            const int c = lc ^ (r & 7);
            glds16(Ap + arow[i] * lda + k0 + c * 8, dA + g * 8 * BK);
        }
#pragma unroll
        for (int i = 0; i < WG; ++i) {
            const int g = w * WG + i;
            const int r = g * 8 + lr;
            const int c = lc ^ (r & 7);
            glds16(Wp + (size_t)(bn * BN + r) * ldw + k0 + c * 8, dW + g * 8 * BK);
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    constexpr int LPS = AG + WG;              // load instructions per wave per stage
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) issue(st, st);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt landed (in every wave after the barrier); later stages stay in flight
        if constexpr (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int ahead = min(nk - 1, kt + STAGES - 2) - kt;     // stages issued after stage kt
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        const h16* cA = sA + (kt % STAGES) * BM * BK;
        const h16* cW = sW + (kt % STAGES) * BN * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 af[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * (BM / 2) + i * 16 + fr;
                const int c = (ks * 4 + fg) ^ (r & 7);
                af[i] = *(const h16x8*)(cA + r * BK + c * 8);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                const int c = (ks * 4 + fg) ^ (r & 7);
                wf[j] = *(const h16x8*)(cW + r * BK + c * 8);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);   // D[n][m]: lane owns 4 consecutive n
        }
    }

    // bias: 4 consecutive columns per lane and column tile, loaded once
    f32x4 bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            bv[j] = *(const f32x4*)(p.bias + bn * BN + wn * (BN / 2) + j * 16 + fg * 4);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = bm * BM + wm * (BM / 2) + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = bn * BN + wn * (BN / 2) + j * 16 + fg * 4;
            if (p.n_store && n >= p.n_store) continue;          // columns of the tile grid beyond the (narrower) result rows
            epilogue_store<EPI>(p, m, n, acc[i][j] + bv[j]);
        }
    }
}

template <int BN, int EPI, int BM = 128, int STAGES = 2>
void launch_t(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.M / BM) * (a.N / BN);
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(h16);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, EPI, STAGES>), dim3(ntiles), dim3(256), lds, s, a);
}

int g_attr_err = 0;
template <int BN, int EPI, int BM = 128, int STAGES = 2>
void set_attr() {
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(h16);
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, BN, EPI, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) g_attr_err = (int)e;
}

int g_force_small = -1;
int g_dephase = -1;
int g_tile_group = -1;
int g_skinny_bm = 128;      // VITLORA_SKINNY_BM=64: 64-row tiles (4 LDS stages) for the skinny LoRA-down GEMM

}  // namespace

int gemm_init(int device) {
    static bool done[64] = {};
    if (device < 0 || device >= 64) return -1;
    if (done[device]) return 0;
    g_attr_err = 0;
    set_attr<64, EPI_STORE_H16>();
    set_attr<64, EPI_STORE_H16, 64, 4>();
    set_attr<64, EPI_STORE_H16, 128, 2>();
    set_attr<128, EPI_STORE_H16>();
    set_attr<128, EPI_RESID_F32>();
    set_attr<128, EPI_GELU>();
    set_attr<128, EPI_GELU_BWD>();
    set_attr<128, EPI_RESID_H16>();
    set_attr<128, EPI_PATCH_FWD>();
    set_attr<128, EPI_PATCH_BWD>();
    set_attr<128, EPI_PATCH_PGD>();
    set_attr<128, EPI_STORE_F32>();
    set_attr<128, EPI_DROP_ACC>();
    if (int e2 = gemm256_init()) g_attr_err = e2;
    if (int e3 = gemm_pp_init()) g_attr_err = e3;
    if (int e4 = gemm_stream_init()) g_attr_err = e4;
    const char* e = getenv("VITLORA_GEMM128");      // A/B switch: force the 128-row kernel
    g_force_small = (e && e[0] == '1') ? 1 : 0;
    const char* dp = getenv("VITLORA_DEPHASE");      // experiment knob: start-offset unit of gemm256
    if (dp) g_dephase = atoi(dp);
    if (const char* tg = getenv("VITLORA_TILE_GROUP")) g_tile_group = atoi(tg);
    if (const char* sb = getenv("VITLORA_SKINNY_BM")) g_skinny_bm = atoi(sb);     // experiment knob: gemm256 tile walk
    if (g_attr_err) return g_attr_err;
    done[device] = true;
    return 0;
}

int gemm_force_small(int v) { const int old = g_force_small; g_force_small = v; return old; }
bool gemm_small_forced() { return g_force_small == 1; }

void launch_gemm(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    const double mv = a.Mvalid ? a.Mvalid : a.M;
    const double flops = 2.0 * mv * (a.n_algo ? a.n_algo : a.N) * (a.K1 + (a.k2_algo ? a.k2_algo : a.K2));
    char name[64];
    if (g_force_small != 1 && a.down_W && gemm_stream_fuses_down(a, epi)) {     // tall, shallow, LoRA down projection inside
        launch_gemm_stream(a, epi, bn, s);
        return;
    }
    const bool narrow = a.n_store > 0 && a.n_store < a.N;      // only the 128-row kernel skips columns
    if (bn != 64 && g_force_small != 1 && !narrow && !a.no_pp && gemm_pp_supports(a, epi)) {
        launch_gemm_pp(a, epi, s);
        return;
    }
    if (bn != 64 && g_force_small != 1 && !narrow && gemm256_supports(a, epi)) {
        GemmArgs b = a;
        if (g_dephase >= 0) b.dephase = g_dephase;
        if (g_tile_group >= 0) b.tile_group = g_tile_group;
        launch_gemm256(b, epi, s);
        return;
    }
    if (g_force_small != 1 && gemm_stream_supports(a, epi, bn)) {        // tall and shallow: an HBM stream (gemm_stream.hip)
        launch_gemm_stream(a, epi, bn, s);
        return;
    }
    if (a.down_W || (a.K2 && !a.A2)) {       // a fused LoRA-down argument set: only gemm_pp / gemm_stream can run it (A2 is null)
        fprintf(stderr, "vitlora: GEMM with the LoRA down projection inside reached the 128-row kernel (M %d N %d K %d)\n", a.M, a.N, a.K1);
        abort();
    }
    snprintf(name, sizeof name, "gemm_nt_kernel<128, %d, %d>", bn == 64 ? 64 : 128, epi);
    ProfScope prof_(name, flops, gemm_algo_bytes(a, epi, mv), s, 2.0 * a.M * a.N * (a.K1 + a.K2));
    if (bn == 64) {
        switch (epi) {
            // skinny LoRA-down product (HBM-bound on A): 64-row tiles -> 3x more workgroups in flight
            case EPI_STORE_H16:
                if (g_skinny_bm == 64 && a.M % 64 == 0) launch_t<64, EPI_STORE_H16, 64, 4>(a, s);
                else launch_t<64, EPI_STORE_H16, 128, 2>(a, s);
                return;
            default: break;
        }
    }
    switch (epi) {
        case EPI_STORE_H16: launch_t<128, EPI_STORE_H16>(a, s); break;
        case EPI_RESID_F32: launch_t<128, EPI_RESID_F32>(a, s); break;
        case EPI_GELU: launch_t<128, EPI_GELU>(a, s); break;
        case EPI_GELU_BWD: launch_t<128, EPI_GELU_BWD>(a, s); break;
        case EPI_RESID_H16: launch_t<128, EPI_RESID_H16>(a, s); break;
        case EPI_PATCH_FWD: launch_t<128, EPI_PATCH_FWD>(a, s); break;
        case EPI_PATCH_BWD: launch_t<128, EPI_PATCH_BWD>(a, s); break;
        case EPI_PATCH_PGD: launch_t<128, EPI_PATCH_PGD>(a, s); break;
        case EPI_STORE_F32: launch_t<128, EPI_STORE_F32>(a, s); break;
        case EPI_DROP_ACC: launch_t<128, EPI_DROP_ACC>(a, s); break;
    }
}

}  // namespace VLNS
