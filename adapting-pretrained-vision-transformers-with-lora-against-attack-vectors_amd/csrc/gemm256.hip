// Main GEMM of the path: BM x 256 x 64 tiles (BM = 256 or 128), 8 waves (2 x 4), one
// PERSISTENT workgroup per CU, 16x16x32 bf16 MFMA, fused epilogues (gemm_epi.h) and the LoRA
// update as extra K tiles.  C[M,N] = A1 W1^T + A2 W2^T, all operands K-contiguous bf16.
//
// Pipeline (one barrier per phase, four phases per 64-deep K tile, 4 half-tiles in flight):
//   * LDS holds two K tiles (2 x (BM + 256) rows of 128 B, XOR-swizzled 16-byte chunks).
//     Each tile is cut in four "half-tiles": A0/A1 = the rows of every wave's upper / lower
//     output half, W0/W1 = the rows of every wave's left / right output half.
//   * phase p of K tile T computes one output quadrant (16 or 8 MFMAs per wave):
//       P1 reads A0,W0 -> (0,0)   P2 reads W1 -> (0,1)   P3 reads A1 -> (1,1)   P4 -> (1,0)
//     and issues ONE half-tile of direct-to-LDS loads (global_load_lds_dwordx4):
//       P1: W1(T+1)   P2: A1(T+1)   P3: A0(T+2)   P4: W0(T+2)
//     i.e. a region is overwritten two phases after its last ds_read, so the single barrier
//     in between orders the write after every wave's reads (WAR), and a half-tile is first
//     read one phase after the counted `s_waitcnt vmcnt(N)` + barrier that retires it (RAW).
//   * N = the loads of the 4 half-tiles issued after the one needed next (2 A + 2 W halves):
//     never 0 in the steady state, so HBM/L2 latency spans ~4 phases of MFMA work.
//   * the last two K tiles use their own exact counts (nothing left to prefetch).
//   * persistent: a workgroup walks output tiles g, g + G, ...; the first 6 half-tiles of the
//     NEXT output tile are issued before the epilogue of the current one, so their latency
//     hides under the epilogue's conversions and stores (K is only 12-48 tiles deep here).
//   * W rows are permuted on the way into LDS so that a lane ends up with 16 ADJACENT output
//     columns: the epilogue moves 16 bytes per lane per instruction.
#include <type_traits>

#include "gemm_epi.h"

namespace {

constexpr int BN = 256;
constexpr int BK = 64;

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define BARRIER()                                   \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        asm volatile("s_barrier" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

template <int BM, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = (BM + BN) * BK;     // elements per K-tile buffer: A rows then W rows
    constexpr int MI = BM / 64;             // 16-row MFMA tiles per output quadrant (m)
    constexpr int NA = BM / 128;            // load instructions per wave per A half-tile
    constexpr int NW = 2;                   // ... per W half-tile
    constexpr int STEADY = 2 * NA + 2 * NW;
    bf16* sm = (bf16*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int lr = lane >> 3, lc = lane & 7;
    const int tilesN = p.N / BN;
    const int ntiles = (p.M / BM) * tilesN;
    const int nk1 = p.K1 / BK;
    const int nk = nk1 + p.K2 / BK;
    const int G = gridDim.x;

    // tile walked in round `it` by this workgroup.  Workgroups that share an XCD (blockIdx % 8
    // under round-robin placement; a speed assumption only) take 1/8 of each round's tiles as a
    // contiguous run, so neighbouring tiles (same A row panel, all of W) meet in one L2.
    auto tile_of = [&](int it) -> int {
        const int bid = (int)blockIdx.x;                   // signed: cnt may be <= 0 past the last round
        const int base = it * G;
        const int cnt = min(G, ntiles - base);             // tiles in this round
        if (bid >= cnt) return -1;
        return base + xcd_remap(bid, cnt);
    };

    // ---- per-lane load bookkeeping ------------------------------------------------------------
    // A half h, instruction i: 8-row group g = w*NA + i of the half; the half's rows are
    // wm'*(BM/2) + h*(BM/4) + [0, BM/4) for wm' = 0,1.
    // W rows are PERMUTED on the way into LDS: LDS row (wn, nh, j', r') of the tile holds output
    // column wn*64 + 16*(r'>>2) + 4*(2*nh + j') + (r'&3), so that the four column tiles of a lane
    // (C/D layout: column 4*fg + reg of tile jj) are the 16 adjacent columns wn*64 + 16*fg + 4*jj + reg.
    unsigned a_m[2][NA], w_n[2][NW];
    int a_lds[2][NA], w_lds[2][NW];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int g = w * NA + i;
            const int chunk = g / (BM / 32), gi = g % (BM / 32);
            a_lds[h][i] = (chunk * (BM / 2) + h * (BM / 4) + gi * 8) * BK;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int g = w * NW + i;
            const int chunk = g >> 2, gi = g & 3;
            w_lds[h][i] = BM * BK + (chunk * 64 + h * 32 + gi * 8) * BK;
        }
    }
    auto set_tile = [&](int bm, int bn) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int m = bm * BM + a_lds[h][i] / BK + lr;
                if (p.a_gather) m = m < p.Mvalid ? m + m / p.patches + 1 : 0;
                a_m[h][i] = (unsigned)m;
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int g = w * NW + i;
                const int chunk = g >> 2, gi = g & 3;
                const int jp = gi >> 1, rp = (gi & 1) * 8 + lr;          // j', r' of this lane's LDS row
                w_n[h][i] = (unsigned)(bn * BN + chunk * 64 + 16 * (rp >> 2) + 4 * (2 * h + jp) + (rp & 3));
            }
        }
    };
    const unsigned csw = (unsigned)((lc ^ lr) * 8);   // swizzled source chunk (row & 7 == lr for every group)

    // addresses = wave-uniform base + 32-bit per-lane byte offset (operands are < 4 GiB)
    auto issueA = [&](int h, int T) {
        const char* Ap; unsigned lda, k0;
        if (T < nk1) { Ap = (const char*)p.A1; lda = p.lda1; k0 = T * BK; } else { Ap = (const char*)p.A2; lda = p.lda2; k0 = (T - nk1) * BK; }
        bf16* dst = sm + (T & 1) * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(Ap + (size_t)((a_m[h][i] * lda + k0 + csw) * 2u), dst + a_lds[h][i]);
    };
    auto issueW = [&](int h, int T) {
        const char* Wp; unsigned ldw, k0;
        if (T < nk1) { Wp = (const char*)p.W1; ldw = p.ldw1; k0 = T * BK; } else { Wp = (const char*)p.W2; ldw = p.ldw2; k0 = (T - nk1) * BK; }
        bf16* dst = sm + (T & 1) * BUF;
#pragma unroll
        for (int i = 0; i < NW; ++i) glds16(Wp + (size_t)((w_n[h][i] * ldw + k0 + csw) * 2u), dst + w_lds[h][i]);
    };
    auto prologue = [&]() {     // K tile 0 complete, A0/W0 of K tile 1
        issueA(0, 0); issueW(0, 0); issueW(1, 0); issueA(1, 0);
        issueA(0, 1); issueW(0, 1);
    };

    // ---- fragment addressing ------------------------------------------------------------------
    const int xo0 = ((0 + fg) ^ (fr & 7)) * 8, xo1 = ((4 + fg) ^ (fr & 7)) * 8;   // k-step 0 / 1 chunk
    const int a_base = (wm * (BM / 2) + fr) * BK;                // + mh*(BM/4)*BK + i*16*BK
    const int w_base = BM * BK + (wn * 64 + fr) * BK;            // + nh*32*BK + j*16*BK

    f32x4 acc[2 * MI][4];
    bf16x8 af[2][MI], wf[2][2][2];

    auto readA = [&](const bf16* buf, int mh) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const bf16* r = buf + a_base + (mh * (BM / 4) + i * 16) * BK;
            af[0][i] = *(const bf16x8*)(r + xo0);
            af[1][i] = *(const bf16x8*)(r + xo1);
        }
    };
    auto readW = [&](const bf16* buf, int nh) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16* r = buf + w_base + (nh * 32 + j * 16) * BK;
            wf[nh][0][j] = *(const bf16x8*)(r + xo0);
            wf[nh][1][j] = *(const bf16x8*)(r + xo1);
        }
    };
    auto mma = [&](int mh, int nh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * MI + i][nh * 2 + j] = mfma16(wf[nh][ks][j], af[ks][i], acc[mh * MI + i][nh * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // MODE 0: steady state (K tiles T+1 and T+2 exist); 1: T = nk-2; 2: T = nk-1
    auto ktile = [&](int T, auto mode) {
        constexpr int MODE = decltype(mode)::value;
        const bf16* buf = sm + (T & 1) * BUF;
        // P1
        readA(buf, 0);
        readW(buf, 0);
        if constexpr (MODE <= 1) { issueW(1, T + 1); VMCNT(STEADY); } else { VMCNT(NA); }
        BARRIER();
        mma(0, 0);
        // P2
        readW(buf, 1);
        if constexpr (MODE <= 1) { issueA(1, T + 1); VMCNT(STEADY); } else { VMCNT(0); }
        BARRIER();
        mma(0, 1);
        // P3
        readA(buf, 1);
        if constexpr (MODE == 0) issueA(0, T + 2);
        BARRIER();
        mma(1, 1);
        // P4
        if constexpr (MODE == 0) { issueW(0, T + 2); VMCNT(STEADY); }
        if constexpr (MODE == 1) { VMCNT(NA + NW); }
        BARRIER();
        mma(1, 0);
    };

    int cur = tile_of(0);
    if (cur < 0) return;
    int bm = cur / tilesN, bn = cur - bm * tilesN;
    set_tile(bm, bn);
    prologue();

    for (int it = 0;; ++it) {
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // everything issued so far (prologue loads, previous epilogue's stores) retired by every wave
        VMCNT(0);
        BARRIER();

        for (int T = 0; T + 2 < nk; ++T) ktile(T, std::integral_constant<int, 0>{});
        ktile(nk - 2, std::integral_constant<int, 1>{});
        ktile(nk - 1, std::integral_constant<int, 2>{});

        // next output tile: its first loads go out before this tile's epilogue
        const int nxt = tile_of(it + 1);
        const int cbm = bm, cbn = bn;
        // (every wave passed the last phase's barrier only after its final ds_reads completed,
        //  so LDS may be refilled from here on without another barrier)
        if (nxt >= 0) {
            bm = nxt / tilesN; bn = nxt - bm * tilesN;
            set_tile(bm, bn);
            prologue();
        }

        // ---- epilogue of tile (cbm, cbn) ----------------------------------------------------
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n0 = cbn * BN + wn * 64 + fg * 16;      // this lane's 16 adjacent columns
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(p.bias + n0 + 4 * j);
        }
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) {
            const int m = cbm * BM + wm * (BM / 2) + i * 16 + fr;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + bv[j];
            epilogue_row16<EPI>(p, m, n0, v);
        }
        if (nxt < 0) break;
    }
}

int g_num_cus = 0;

template <int BM, int EPI>
void launch_t(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.M / BM) * (a.N / BN);
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16);
    const int grid = ntiles < g_num_cus ? ntiles : g_num_cus;
    hipLaunchKernelGGL((gemm256_kernel<BM, EPI>), dim3(grid), dim3(512), lds, s, a);
}

template <int BM, int EPI>
void set_attr() {
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16);
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<BM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int EPI>
void launch_bm(const GemmArgs& a, hipStream_t s) {
    // 256-row tiles unless the last, partly filled round of workgroups costs more than the
    // (slightly less efficient) 128-row tiles would: rounds are over the CUs.
    const long c = g_num_cus;
    const long t256 = (long)(a.M / 256) * (a.N / BN), t128 = (long)(a.M / 128) * (a.N / BN);
    const double r256 = (double)((t256 + c - 1) / c), r128 = 0.5 * (double)((t128 + c - 1) / c);
    const bool use256 = (a.M % 256 == 0) && r256 <= 1.08 * r128;
    if (use256) launch_t<256, EPI>(a, s); else launch_t<128, EPI>(a, s);
}

}  // namespace

bool gemm256_supports(const GemmArgs& a, int epi) {
    if (a.N % BN || a.M % 128 || a.K1 % BK || a.K2 % BK) return false;
    if ((a.K1 + a.K2) / BK < 2) return false;
    (void)epi;
    return true;
}

void gemm256_init() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
    set_attr<256, EPI_STORE_BF16>(); set_attr<128, EPI_STORE_BF16>();
    set_attr<256, EPI_RESID_F32>(); set_attr<128, EPI_RESID_F32>();
    set_attr<256, EPI_GELU>(); set_attr<128, EPI_GELU>();
    set_attr<256, EPI_GELU_BWD>(); set_attr<128, EPI_GELU_BWD>();
    set_attr<256, EPI_PATCH_FWD>(); set_attr<128, EPI_PATCH_FWD>();
    set_attr<256, EPI_PATCH_BWD>(); set_attr<128, EPI_PATCH_BWD>();
    set_attr<256, EPI_STORE_F32>(); set_attr<128, EPI_STORE_F32>();
}

void launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_STORE_BF16: launch_bm<EPI_STORE_BF16>(a, s); break;
        case EPI_RESID_F32: launch_bm<EPI_RESID_F32>(a, s); break;
        case EPI_GELU: launch_bm<EPI_GELU>(a, s); break;
        case EPI_GELU_BWD: launch_bm<EPI_GELU_BWD>(a, s); break;
        case EPI_PATCH_FWD: launch_bm<EPI_PATCH_FWD>(a, s); break;
        case EPI_PATCH_BWD: launch_bm<EPI_PATCH_BWD>(a, s); break;
        case EPI_STORE_F32: launch_bm<EPI_STORE_F32>(a, s); break;
    }
}
