// Main GEMM of the path: 256 x 256 x 64 tiles, 8 waves (2 x 4), one PERSISTENT workgroup per
// CU, 16x16x32 h16 MFMA, fused epilogues (gemm_epi.h) and the LoRA update as extra K tiles.
// C[M,N] = A1 W1^T + A2 W2^T, all operands K-contiguous h16.
//
// Pipeline (four phases per 64-deep K tile, 4 half-tiles of loads in flight):
//   * LDS holds two K tiles (2 x (BM + 256) rows of 128 B, XOR-swizzled 16-byte chunks).
//     Each tile is cut in four "half-tiles": A0/A1 = the rows of every wave's upper / lower
//     output half, W0/W1 = the rows of every wave's left / right output half.
//   * phase p of K tile T computes one output quadrant (16 MFMAs per wave; 8 for BM = 128):
//       P1 reads A0,W0 -> (0,0)   P2 reads W1 -> (0,1)   P3 reads A1 -> (1,1)   P4 -> (1,0)
//     and issues ONE half-tile of direct-to-LDS loads (global_load_lds_dwordx4):
//       P1: W1(T+1)   P2: A1(T+1)   P3: A0(T+2)   P4: W0(T+2)
//     A region is overwritten two phases after its last ds_read (WAR) and a half-tile is first
//     read one phase after the counted `s_waitcnt vmcnt(N)` + barrier that retires it (RAW).
//   * N = the loads of the 4 half-tiles issued after the one needed next (2 A + 2 W halves):
//     never 0 in the steady state; the last two K tiles use their own exact counts.
//   * every phase is {ds_reads, loads, wait} BARRIER {MFMAs} BARRIER and waves 4-7 (the SIMD
//     partners of waves 0-3) trail by ONE barrier: while one half of the workgroup runs its
//     MFMA segment the other half runs its read/load segment, so each SIMD's matrix pipe always
//     has a wave to run.  The WAR/RAW distances above hold under that half-phase stagger.
//   * persistent: a workgroup walks output tiles g, g + G, ...; the first 6 half-tiles of the
//     NEXT output tile are issued before the epilogue of the current one.
//   * tail: the rows of the last, partly filled round are cut into 128-row tiles when that
//     lets them finish in half a round (same kernel, BM = 128 code path).
//   * W rows are permuted on the way into LDS so that a lane ends up with 16 ADJACENT output
//     columns: the epilogue moves 16 bytes per lane per instruction.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_epi.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

#ifdef VITLORA_GEMM_STAMPS   // diagnostic build only (tools/gemm_stamp.hip): per-wave s_memtime stamps, 4 per output tile
__device__ unsigned long long g_gemm_stamps[256 * 8 * 16 * 4];
#define GSTAMP(it, k) do { if (blockIdx.x < 256 && (it) < 16 && (threadIdx.x & 63) == 0) g_gemm_stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (it)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GSTAMP(it, k) do { } while (0)
#endif

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

// VMEM instructions per lane that the epilogue of one output tile is GUARANTEED to issue per
// 16-row block (a lower bound: conditional ones are not counted).  They sit between the next
// tile's first loads and its main loop in the in-order vmcnt queue.
template <int EPI>
constexpr int epilogue_vmem_per_row() {
    return EPI == EPI_STORE_H16 ? 2 : EPI == EPI_STORE_F32 ? 4 : EPI == EPI_RESID_F32 ? 8
         : EPI == EPI_GELU ? 4 : EPI == EPI_GELU_BWD ? 4 : EPI == EPI_RESID_H16 ? 4 : 0;
}
constexpr int clamp63(int v) { return v > 63 ? 63 : v; }

template <int V> using IC = std::integral_constant<int, V>;

// LDS-DMA with explicit operands: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset,
// LDS destination (wave-uniform) through M0, which is saved and restored around the instruction.
// hipcc does not count these in its own s_waitcnt bookkeeping; every wait on them below is explicit.
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, const void* lds_dst) {
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds4_sv(const void* sbase, unsigned voff, const void* lds_dst) {
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// One launch walks `ntiles` BM-row tiles (row-major over (bm, bn), tile rows starting at bm0)
// round-robin over the G workgroups.  A GEMM is one BM = 256 launch over full rounds plus, when
// the tail pays, one BM = 128 launch over the remaining rows (plan_tiles below).
template <int BM, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs p, int ntiles, int bm0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sm = (h16*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int lr = lane >> 3, lc = lane & 7;
    const unsigned csw = (unsigned)((lc ^ lr) * 8);   // swizzled source chunk (row & 7 == lr for every load group)
    // fp32 outputs keep the natural MFMA column order (a lane quad then writes 64 contiguous bytes per row and
    // instruction); 16-bit outputs permute the W rows so that a lane owns 16 adjacent columns (16-byte pieces)
    constexpr bool PERM = !(EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32);
    const unsigned wl = PERM ? (unsigned)(16 * (lr >> 2) + (lr & 3)) : (unsigned)lr;   // lane part of the W row
    const int tilesN = p.N / BN;
    const int nk1 = p.K1 / BK;
    const int nk = nk1 + p.K2 / BK;
    const int G = gridDim.x, bid = (int)blockIdx.x;

    // ---- direct-to-LDS loaders ----------------------------------------------------------------
    // A half h, instruction i: 8-row group g = w*NA + i; the half's rows are wm'*(BM/2) + h*(BM/4) + [0, BM/4).
    // W rows are PERMUTED on the way into LDS: LDS row (wn, nh, j', r') holds output column
    // wn*64 + 16*(r'>>2) + 4*(2*nh + j') + (r'&3), so that the four column tiles of a lane (C/D
    // layout: column 4*fg + reg of tile jj) are the 16 adjacent columns wn*64 + 16*fg + 4*jj + reg.
    // addresses = wave-uniform base + 32-bit per-lane byte offset (operands are < 4 GiB);
    // nothing per-lane is kept between calls: row = uniform part + lr, 2 VALU per load.
    // per-lane byte offsets (kernel-invariant): row part (lr or the permuted W row) * ld + chunk
    const unsigned voA1 = ((unsigned)lr * (unsigned)p.lda1 + csw) * 2u, voA2 = ((unsigned)lr * (unsigned)p.lda2 + csw) * 2u;
    const unsigned voW1 = (wl * (unsigned)p.ldw1 + csw) * 2u, voW2 = (wl * (unsigned)p.ldw2 + csw) * 2u;
    const float gather_rcp = p.patches > 0 ? 1.f / (float)p.patches : 0.f;
    auto issueA = [&](int bm, int h, int T) {
        constexpr int NA = BM / 128, BUF = (BM + BN) * BK;
        const bool ext = T >= nk1;
        const char* Ap = (const char*)(ext ? p.A2 : p.A1);
        const unsigned lda = ext ? p.lda2 : p.lda1, k0 = (ext ? T - nk1 : T) * BK, vo = ext ? voA2 : voA1;
        h16* dst = sm + (T & 1) * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int g = w * NA + i;
            const int r0 = (g / (BM / 32)) * (BM / 2) + h * (BM / 4) + (g % (BM / 32)) * 8;
            if constexpr (EPI == EPI_PATCH_BWD || EPI == EPI_PATCH_PGD) {
                if (p.a_gather && !ext) {
                    // patch-embedding backward: GEMM row b*patches + pi reads token row b*tokens + 1 + pi of the gradient
                    // stream (the CLS rows are skipped); rows past Mvalid read row 0.  m / patches through the reciprocal:
                    // exact for m < 2^22 (the error of (m + .5) * rcp is far below the .5 / patches margin).
                    const int mrow = bm * BM + r0 + lr;
                    const int src = mrow < p.Mvalid ? mrow + (int)(((float)mrow + 0.5f) * gather_rcp) + 1 : 0;
                    glds16(Ap + (size_t)(((unsigned)src * lda + k0 + csw) * 2u), dst + r0 * BK);
                    continue;
                }
            }
            const unsigned so = ((unsigned)(bm * BM + r0) * lda + k0) * 2u;      // wave-uniform
            glds16(Ap + (size_t)(vo + so), dst + r0 * BK);
        }
    };
    auto issueW = [&](int bn, int h, int T) {
        constexpr int BUF = (BM + BN) * BK;
        const bool ext = T >= nk1;
        const char* Wp = (const char*)(ext ? p.W2 : p.W1);
        const unsigned ldw = ext ? p.ldw2 : p.ldw1, k0 = (ext ? T - nk1 : T) * BK, vo = ext ? voW2 : voW1;
        h16* dst = sm + (T & 1) * BUF + BM * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = w * 2 + i;
            const int chunk = g >> 2, gi = g & 3;
            // LDS row (chunk, h, j' = gi>>1, r' = (gi&1)*8 + lr)  <-  column chunk*64 + 16*(r'>>2) + 4*(2h+j') + (r'&3)
            const unsigned col = PERM ? chunk * 64 + 4 * (2 * h + (gi >> 1)) + 32 * (gi & 1) : chunk * 64 + h * 32 + gi * 8;
            const unsigned so = ((unsigned)(bn * BN) + col) * ldw * 2u + k0 * 2u;
            glds16(Wp + (size_t)(vo + so), dst + (chunk * 64 + h * 32 + gi * 8) * BK);
        }
    };
    auto prologue = [&](int bm, int bn) {     // K tile 0 complete, A0/W0 of K tile 1
        issueA(bm, 0, 0); issueW(bn, 0, 0); issueW(bn, 1, 0); issueA(bm, 1, 0);
        issueA(bm, 0, 1); issueW(bn, 0, 1);
    };

    // A tile that is not the workgroup's first is entered with at least EX "older" VMEM
    // instructions (the previous tile's epilogue) queued behind its first loads.  They may still
    // be in flight when the tile starts: the waits of K tile 0 count them instead of draining
    // them, so the stores of one tile retire under the MFMAs of the next.
    constexpr int EX = (BM / 32) * epilogue_vmem_per_row<EPI>();   // BM/32 16-row blocks per lane

    // One output tile: main loop over K (its first 6 half-tiles are already issued), then
    // `issue_next()` (starts the next tile's loads), then the epilogue.
    auto run_tile = [&](int bm, int bn, bool first, int it_, auto&& issue_next) {
        GSTAMP(it_, 0);
        constexpr int BUF = (BM + BN) * BK;
        constexpr int MI = BM / 64;             // 16-row MFMA tiles per output quadrant (m)
        constexpr int NA = BM / 128, NW = 2;
        constexpr int STEADY = 2 * NA + 2 * NW;
        const int xo0 = ((0 + fg) ^ (fr & 7)) * 8, xo1 = ((4 + fg) ^ (fr & 7)) * 8;   // k-step 0 / 1 chunk
        const int a_base = (wm * (BM / 2) + fr) * BK;            // + mh*(BM/4)*BK + i*16*BK
        const int w_base = BM * BK + (wn * 64 + fr) * BK;        // + nh*32*BK + j*16*BK

        f32x4 acc[2 * MI][4];
        h16x8 af[2][MI], wf[2][2][2];
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto readA = [&](const h16* buf, int mh) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const h16* r = buf + a_base + (mh * (BM / 4) + i * 16) * BK;
                af[0][i] = *(const h16x8*)(r + xo0);
                af[1][i] = *(const h16x8*)(r + xo1);
            }
        };
        auto readW = [&](const h16* buf, int nh) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const h16* r = buf + w_base + (nh * 32 + j * 16) * BK;
                wf[nh][0][j] = *(const h16x8*)(r + xo0);
                wf[nh][1][j] = *(const h16x8*)(r + xo1);
            }
        };
        // lower_only: the K tile's upper 32 columns are zero on both sides (LoRA tile with r * modules <= 32)
        auto mma = [&](int mh, int nh, bool lower_only) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1 && lower_only) break;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mh * MI + i][nh * 2 + j] = mfma16(wf[nh][ks][j], af[ks][i], acc[mh * MI + i][nh * 2 + j]);
            }
            __builtin_amdgcn_s_setprio(0);
        };
        const bool ext_half = p.K2 == BK && p.k2_used > 0 && p.k2_used <= 32;   // the LoRA tile is the last K tile
        // MODE 0: steady state (K tiles T+1 and T+2 exist); 1: T = nk-2; 2: T = nk-1
        // EXC: older in-flight epilogue instructions to tolerate (K tile 0 only)
        auto ktile = [&](int T, auto mode, auto extra) {
            constexpr int MODE = decltype(mode)::value;
            constexpr int EXC = decltype(extra)::value;
            const h16* buf = sm + (T & 1) * BUF;
            // P1
            readA(buf, 0);
            readW(buf, 0);
            if constexpr (MODE <= 1) { issueW(bn, 1, T + 1); VMCNT(clamp63(STEADY + EXC)); } else { VMCNT(NA); }
            BARRIER();
            mma(0, 0, MODE == 2 && ext_half);
            BARRIER();
            // P2
            readW(buf, 1);
            if constexpr (MODE <= 1) { issueA(bm, 1, T + 1); VMCNT(clamp63(STEADY + EXC)); } else { VMCNT(0); }
            BARRIER();
            mma(0, 1, MODE == 2 && ext_half);
            BARRIER();
            // P3
            readA(buf, 1);
            if constexpr (MODE == 0) issueA(bm, 0, T + 2);
            BARRIER();
            mma(1, 1, MODE == 2 && ext_half);
            BARRIER();
            // P4
            if constexpr (MODE == 0) { issueW(bn, 0, T + 2); VMCNT(clamp63(STEADY + EXC)); }
            if constexpr (MODE == 1) { VMCNT(clamp63(NA + NW + EXC)); }
            BARRIER();
            mma(1, 0, MODE == 2 && ext_half);
            BARRIER();
        };

        // this tile's first loads (A0, W0 of K tile 0) retired by every wave; the EX younger
        // epilogue instructions of the previous tile and the 4 later half-tiles may stay in flight
        // (a workgroup's first tile has no older epilogue in flight: drain, which also satisfies
        //  every wait of K tile 0, all of them on loads issued before this point)
        if (first) VMCNT(0);
        VMCNT(clamp63(STEADY + EX));
        BARRIER();
        if (wm == 1) BARRIER();                      // stagger: waves 4-7 trail by one barrier
        if (nk > 2) {
            ktile(0, IC<0>{}, IC<EX>{});
            for (int T = 1; T + 2 < nk; ++T) ktile(T, IC<0>{}, IC<0>{});
            ktile(nk - 2, IC<1>{}, IC<0>{});
        } else {
            ktile(0, IC<1>{}, IC<EX>{});
        }
        ktile(nk - 1, IC<2>{}, IC<0>{});
        if (wm == 0) BARRIER();                      // waves 0-3 rejoin (equal barrier counts)
        // every wave passed the last barrier only after its final ds_reads completed: LDS may be refilled
        GSTAMP(it_, 1);
        issue_next();
        GSTAMP(it_, 2);

        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n0 = bn * BN + wn * 64 + fg * 16;  // this lane's 16 adjacent columns (PERM)
        const int nq = bn * BN + wn * 64 + fg * 4;   // natural order: column tile j of this lane starts at nq + 16 j
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(p.bias + (PERM ? n0 + 4 * j : nq + 16 * j));
        }
        if constexpr (EPI == EPI_GELU_BWD || EPI == EPI_RESID_H16) {
            // the saved gelu'(z) (RESID_H16: the residual-stream rows) of a whole half of the wave's rows is requested before any of it is used or any
            // result is stored (the stores may alias as far as the compiler knows, so it would not hoist the loads
            // itself): two memory round trips per output tile instead of eight
            h16x8 rz[2 * MI][2];
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) {
                const h16* zs = (const h16*)p.R + (size_t)(bm * BM + wm * (BM / 2) + i * 16 + fr) * p.ldr + n0;
                rz[i][0] = ld_once((const h16x8*)zs);
                rz[i][1] = ld_once((const h16x8*)(zs + 8));
            }
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) {
                const int m = bm * BM + wm * (BM / 2) + i * 16 + fr;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + bv[j];
                if constexpr (EPI == EPI_GELU_BWD) epilogue_gelu_bwd16(p, m, n0, v, rz[i][0], rz[i][1]);
                else epilogue_resid16(p, m, n0, v, rz[i][0], rz[i][1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) {
                const int m = bm * BM + wm * (BM / 2) + i * 16 + fr;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + bv[j];
                if constexpr (PERM) {
                    epilogue_row16<EPI>(p, m, n0, v);
                } else {
                    float* dst = (float*)p.C + (size_t)m * p.ldc + nq;
                    if constexpr (EPI == EPI_RESID_F32) {
                        const float* r = (const float*)p.R + (size_t)m * p.ldr + nq;
                        f32x4 rv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[q] = *(const f32x4*)(r + 16 * q);
#pragma unroll
                        for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 16 * q) = v[q] + rv[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 16 * q) = v[q];
                    }
                }
            }
        }
        GSTAMP(it_, 3);
    };

    // Workgroups that share an XCD (blockIdx % 8 under round-robin placement; a speed assumption
    // only) take 1/8 of each round's tiles as a contiguous run, so neighbouring tiles (same A row
    // panel, all of W) meet in one L2.
    auto tile_of = [&](int it) -> int {
        const int base = it * G;
        const int cnt = min(G, ntiles - base);
        if (bid >= cnt) return -1;
        return base + xcd_remap(bid, cnt);
    };
    // tile id -> (bm, bn): groups of GM tile rows walked column by column (bm fastest inside a group): the 32 consecutive ids
    // of an XCD's workgroups then cover GM row panels x 32/GM W panels instead of 2-3 row panels x every W panel, and W
    // (4.7 MB for N = 3072, more than one XCD's L2) is not re-streamed by every XCD in every round
    // (wide outputs only: with 3 W panels the row-major walk already keeps them resident and grouping re-reads A; measured
    //  L2-miss bytes per launch, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: N = 3072 fc1 1060 -> 961 MB, fc2 dgrad 1081 -> 978 MB;
    //  N = 768 / K = 3072 would go 428 -> 538 MB; launch times equal within noise either way)
    const int GM = p.tile_group > 0 ? p.tile_group : (tilesN >= 8 ? 8 : 1);
    const int tilesM = ntiles / tilesN;
    auto bm_of = [&](int tile, int& bn) -> int {
        const int per_group = GM * tilesN;
        const int g = tile / per_group, r = tile - g * per_group;
        const int rows = min(GM, tilesM - g * GM);
        bn = r / rows;
        return bm0 + g * GM + (r - bn * rows);
    };
    int cur = tile_of(0);
    if (cur < 0) return;
    // De-phase the workgroups: without this every CU computes, then every CU stores its tile at
    // the same moment (HBM idle, then a 32 MB write burst).  Quarter-tile start offsets spread
    // the epilogue traffic of neighbouring CUs over the whole round.
    if (p.dephase > 0) {
        for (int i = (bid >> 3) & 3; i > 0; --i)
            for (int j = 0; j < p.dephase; ++j) __builtin_amdgcn_s_sleep(127);
    }
    int cbn;
    int cbm = bm_of(cur, cbn);
    prologue(cbm, cbn);
    for (int it = 0; cur >= 0; ++it) {
        const int nxt = tile_of(it + 1);
        int nbn = 0;
        const int nbm = nxt >= 0 ? bm_of(nxt, nbn) : 0;
        run_tile(cbm, cbn, it == 0, it, [&]() {
            if (nxt >= 0) prologue(nbm, nbn);
        });
        cur = nxt; cbm = nbm; cbn = nbn;
    }
}

int g_num_cus = 0;

// Choose how many 256-row tile rows stay "big"; the rest of the rows become 128-row tiles that
// must fit one round (<= G workgroups).  cost = big rounds + 0.55 per small round.
double g_tail_cost = 0.55;      // VITLORA_TAIL_COST: price of the 128-row tail launch in units of one round of 256-row tiles
bool plan_tiles(int M, int N, int G, int* nbig, int* nsmall) {
    const int tilesN = N / BN, rows256 = M / 256, rows128 = M / 128;
    double best = 1e30;
    bool ok = false;
    for (int give = 0; give <= rows256; ++give) {
        const int big_rows = rows256 - give;
        const long nb = (long)big_rows * tilesN, ns = (long)(rows128 - 2 * big_rows) * tilesN;
        if (ns > G) break;
        const double cost = (double)((nb + G - 1) / G) + (ns > 0 ? g_tail_cost : 0.0);
        if (cost < best - 1e-9) { best = cost; *nbig = (int)nb; *nsmall = (int)ns; ok = true; }
    }
    return ok;
}

int g_tail_sibling = 0;                 // VITLORA_TAIL_SIBLING=1 / 2: tail launch as a parallel graph branch (2: enqueued before the main launch)
hipStream_t g_tail_stream = nullptr;
hipEvent_t g_tail_fork = nullptr, g_tail_join = nullptr;

template <int BM, int EPI>
void launch_one(const GemmArgs& a, int ntiles, int bm0, hipStream_t s) {
    if (ntiles <= 0) return;
    // one profiling record per actual launch, named like the kernel symbol rocprofv3 reports;
    // algorithmic FLOPs of this launch = its share of the GEMM's rows
    char name[64];
    snprintf(name, sizeof name, "gemm256_kernel<%d, %d>", BM, EPI);
    const double rows = (double)(ntiles / (a.N / BN)) * BM;
    const double valid = a.Mvalid ? (double)a.Mvalid / a.M : 1.0;
    const double k2x = a.K2 ? (a.k2_used > 0 && a.k2_used <= 32 ? 32.0 : (double)a.K2) : 0.0;     // the zero upper half of the LoRA tile is skipped
    ProfScope prof_(name, 2.0 * rows * valid * (a.n_algo ? a.n_algo : a.N) * (a.K1 + (a.k2_algo ? a.k2_algo : a.K2)),
                    gemm_algo_bytes(a, EPI, rows * valid), s, 2.0 * rows * a.N * (a.K1 + k2x));
    const int grid = ntiles < g_num_cus ? ntiles : g_num_cus;
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(h16) + 1024;
    hipLaunchKernelGGL((gemm256_kernel<BM, EPI>), dim3(grid), dim3(512), lds, s, a, ntiles, bm0);
}

template <int EPI>
void launch_t(const GemmArgs& a, hipStream_t s) {
    if constexpr (EPI == EPI_PATCH_BWD || EPI == EPI_PATCH_PGD) {
        // the image-layout scatter (and the PGD step on top of it) does not fit the 256-row tile's register budget (26 spilled
        // VGPRs, every reload a vmcnt(0) drain of the LDS-DMA queue): all rows as 128-row tiles, as many rounds as it takes
        launch_one<128, EPI>(a, (a.M / 128) * (a.N / BN), 0, s);
        return;
    }
    int nbig = 0, nsmall = 0;
    plan_tiles(a.M, a.N, g_num_cus, &nbig, &nsmall);
    if (g_tail_sibling && nbig > 0 && nsmall > 0 && g_tail_stream && !g_prof) {
        // A/B (round-4 verdict 1a): the 128-row tail as a SIBLING of the main launch -- a parallel branch on a second stream
        // between a fork and a join -- instead of its successor (the rows are disjoint)
        (void)hipEventRecord(g_tail_fork, s);
        (void)hipStreamWaitEvent(g_tail_stream, g_tail_fork, 0);
        if (g_tail_sibling == 2) launch_one<128, EPI>(a, nsmall, (nbig / (a.N / BN)) * 2, g_tail_stream);      // tail enqueued first
        launch_one<256, EPI>(a, nbig, 0, s);
        if (g_tail_sibling != 2) launch_one<128, EPI>(a, nsmall, (nbig / (a.N / BN)) * 2, g_tail_stream);
        (void)hipEventRecord(g_tail_join, g_tail_stream);
        (void)hipStreamWaitEvent(s, g_tail_join, 0);
        return;
    }
    launch_one<256, EPI>(a, nbig, 0, s);
    launch_one<128, EPI>(a, nsmall, (nbig / (a.N / BN)) * 2, s);
}

int g_attr_err256 = 0;
template <int BM, int EPI>
void set_attr1() {
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(h16) + 1024;
    const hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<BM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) g_attr_err256 = (int)e;
}
template <int EPI>
void set_attr() { set_attr1<256, EPI>(); set_attr1<128, EPI>(); }

}  // namespace

bool gemm256_supports(const GemmArgs& a, int epi) {
    if (epi == EPI_DROP_ACC) return false;                 // masked accumulate stays on the 128-row kernel
    if (a.a_gather && !(epi == EPI_PATCH_BWD || epi == EPI_PATCH_PGD)) return false;     // row-gathered A: the patch-gradient epilogues only
    if (a.N % BN || a.M % 128 || a.K1 % BK || a.K2 % BK) return false;
    if ((a.K1 + a.K2) / BK < 2) return false;
    int nb, ns;
    if (epi == EPI_PATCH_BWD || epi == EPI_PATCH_PGD) return true;
    return plan_tiles(a.M, a.N, g_num_cus ? g_num_cus : 256, &nb, &ns);
}

void gemm256_set_cus(int n) { g_num_cus = n; }

int gemm256_init() {
    g_attr_err256 = 0;
    if (const char* tc = getenv("VITLORA_TAIL_COST")) g_tail_cost = atof(tc);
    if (const char* ts = getenv("VITLORA_TAIL_SIBLING")) g_tail_sibling = atoi(ts);
    if (g_tail_sibling && !g_tail_stream) {
        if (hipStreamCreateWithFlags(&g_tail_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g_tail_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g_tail_join, hipEventDisableTiming) != hipSuccess) { g_tail_stream = nullptr; g_tail_sibling = 0; }
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
    set_attr<EPI_STORE_H16>(); set_attr<EPI_RESID_F32>(); set_attr<EPI_GELU>(); set_attr<EPI_GELU_BWD>(); set_attr<EPI_RESID_H16>();
    set_attr<EPI_PATCH_FWD>(); set_attr<EPI_PATCH_BWD>(); set_attr<EPI_PATCH_PGD>(); set_attr<EPI_STORE_F32>(); set_attr<EPI_NONE>();
    return g_attr_err256;
}

void launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_STORE_H16: launch_t<EPI_STORE_H16>(a, s); break;
        case EPI_RESID_F32: launch_t<EPI_RESID_F32>(a, s); break;
        case EPI_GELU: launch_t<EPI_GELU>(a, s); break;
        case EPI_GELU_BWD: launch_t<EPI_GELU_BWD>(a, s); break;
        case EPI_RESID_H16: launch_t<EPI_RESID_H16>(a, s); break;
        case EPI_PATCH_FWD: launch_t<EPI_PATCH_FWD>(a, s); break;
        case EPI_PATCH_BWD: launch_t<EPI_PATCH_BWD>(a, s); break;
        case EPI_PATCH_PGD: launch_t<EPI_PATCH_PGD>(a, s); break;
        case EPI_STORE_F32: launch_t<EPI_STORE_F32>(a, s); break;
        case EPI_NONE: launch_t<EPI_NONE>(a, s); break;
    }
}

}  // namespace VLNS
