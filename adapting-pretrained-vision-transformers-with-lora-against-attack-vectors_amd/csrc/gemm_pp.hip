// "Ping-pong" form of the main GEMM: the epilogue of one output tile runs UNDER the main loop of the next one.
//
// gemm256.hip computes a 256 x 256 tile with all 8 waves and then all 8 waves run the epilogue: for the epilogue-heavy
// shapes (GELU forward: erf + exp per element and two stores; GELU backward; the fp32 residual) the matrix pipes idle
// for 40 - 65 % of a tile's life.  Here the workgroup's two wave groups (waves 0-3 / 4-7, one wave of each per SIMD)
// own DIFFERENT 128 x 256 output tiles and alternate roles per tile "slot":
//
//     slot n   : group n&1 = COMPUTE: per 64-deep K tile, 24 ds_read_b128 + 64 MFMA 16x16x32 (128 x 64 outputs per wave)
//                group 1-(n&1) = HELPER: (a) the epilogue of ITS previous tile, one 16-row block per step,
//                                         (b) all direct-to-LDS loads (12 x 1 KiB per wave and step) for the compute
//                                             group -- K tile s+2 of this slot, or the first K tiles of the next slot
//     one s_barrier per K tile ("step"); the roles swap at the slot boundary.
//
// LDS: three stages of (128 A rows + 256 W rows) x 128 B = 144 KiB, XOR-swizzled 16-byte chunks, W rows permuted as in
// gemm256.hip (a lane ends up with 16 adjacent output columns).  K tile q of the stream lives in stage q % 3, is issued
// in step q-2 and waited for (counted vmcnt by the issuing wave) before the barrier that ends step q-1; the stage it
// overwrites was last read in step q-3 (WAR safe behind that step's barrier).
// Epilogue operands that must be READ (the saved gelu') and the bias are requested with explicit global_load
// instructions two steps before their use and waited for with counted vmcnt, so the helper never stalls on the loads it
// has just issued for the compute group (hipcc does not see LDS-DMA in its own vmcnt bookkeeping and would otherwise wait
// for everything).  Registers that receive such a load must not be spilled or copied before the counted wait: the
// kernels are built with ZERO spilled VGPRs and tests/test_host_cpu.py holds the build to that.
//
// Measured (tools/gemm_pp_check.py): 7 - 12 % ahead of gemm256 on the plain 16-bit-store shapes with K <= 2304, level at
// K = 3072, behind on the GELU epilogues: the helper group is 4 waves, so the erf VALU of a tile takes twice as long as
// with 8, and its stores share the wave's in-order VMEM queue with the 12 DMA pieces per step.  launch_gemm therefore
// sends only those shapes here (VITLORA_GEMM_PP=1: every supported shape, 0: none).
//
// Tiles are 128 rows, walked round-robin: full rounds give every workgroup a PAIR (2u, 2u+1); a last partial round of
// at most G tiles is handed out as single tiles (group 0 only), so the tail quantum is one 128-row tile.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_epi.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

constexpr int BM = 128;
constexpr int BN = 256;
constexpr int BK = 64;
constexpr int STG0 = (BM + BN) * BK;          // h16 elements per stage
constexpr int DROWS = 32;                    // extra rows per stage for the LoRA-down operand (ND > 0)
constexpr int stg_of(int nd) { return (BM + BN + (nd ? DROWS : 0)) * BK; }
constexpr int NSTAGE = 3;
constexpr int ndma_of(int nd) { return nd ? 13 : 12; }   // LDS-DMA instructions per helper wave and K tile (4 A + 8 W [+ 1 Ad])
constexpr int STATIC_STEPS = 10;         // helper steps with a statically known epilogue share (needs nk >= 12)

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define BARRIER()                                   \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        asm volatile("s_barrier" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

template <int V> using IC = std::integral_constant<int, V>;

// LDS-DMA with explicit operands: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset, LDS destination
// (wave-uniform) through M0, saved and restored around the instruction.
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, const void* lds_dst) {
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// explicit 16-byte global load (not tracked by the compiler's waitcnt insertion: every use sits behind wait_dep)
__device__ __forceinline__ f32x4 gload16(const void* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(ptr) : "memory");
    return v;
}
// the same for an operand that is read exactly once (saved gelu'(z), residual-stream rows): non-temporal (gemm_epi.h, VL_EPI_NT)
__device__ __forceinline__ f32x4 gload16_once(const void* ptr) {
    f32x4 v;
#if VL_EPI_NT
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(v) : "v"(ptr) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(ptr) : "memory");
#endif
    return v;
}
// the same for a block's registers alone (BC: no bias registers)
template <int N, int NB>
__device__ __forceinline__ void wait_dep1(f32x4 (&b)[4]) {
    if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// counted wait that the listed registers depend on (keeps their uses below it); NB of the block's registers are real
template <int N, int NB>
__device__ __forceinline__ void wait_dep(f32x4 (&a)[4], f32x4 (&b)[4]) {
    if constexpr (NB == 4)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                     : "n"(N) : "memory");
    else if constexpr (NB == 2)
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])
                     : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}

// VMEM instructions of one 16-row epilogue block: loads requested ahead / stores
template <int EPI> constexpr int epi_loads() { return (EPI == EPI_GELU_BWD || EPI == EPI_RESID_H16) ? 2 : 0; }
template <int EPI> constexpr int epi_stores() {
    return EPI == EPI_STORE_H16 ? 2 : EPI == EPI_STORE_F32 ? 4 : EPI == EPI_GELU ? 4 : (EPI == EPI_GELU_BWD || EPI == EPI_RESID_H16) ? 2 : 0;
}

// ND > 0: the LoRA down projection t = A1 Ad^T (16 ND columns) is computed by the HELPER group from the same LDS stages the
// compute group reads, and written as the A operand of the LoRA K tile straight into LDS (optionally also to p.down_out):
// no separate skinny GEMM over A1, no t round trip through HBM.
// BC (with ND > 0 only): the bias rides in column 63 of the LoRA K tile -- the helper writes a 1 there beside t, W2 column 63 holds
// the bias (vl_lora_commit) -- so the epilogue needs neither the 16 bias registers nor their four loads per tile (the residual-add
// epilogue with its two row operands in flight does not fit the register budget otherwise: a spilled register is a vmcnt(0) drain)
template <int EPI, int ND, bool BC = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs p, int ntiles) {
    constexpr int STG = stg_of(ND);
    constexpr int NDMA = ndma_of(ND);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sm = (h16*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, wn = w & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int lr = lane >> 3, lc = lane & 7;
    const unsigned csw = (unsigned)((lc ^ lr) * 8);                 // swizzled source chunk
    // fp32 outputs keep the natural MFMA column order (64 contiguous bytes per row, lane quad and instruction); 16-bit
    // outputs permute the W rows so that a lane owns 16 adjacent columns
    constexpr bool PERM = EPI != EPI_STORE_F32;
    const unsigned wl = PERM ? (unsigned)(16 * (lr >> 2) + (lr & 3)) : (unsigned)lr;      // lane part of the W row
    const int tilesN = p.N / BN;
    // tile id -> (bm, bn): groups of GM tile rows are walked column by column (bm fastest inside a group), so that the 32
    // workgroups of an XCD, which hold consecutive ids, cover a compact block (16 row panels x 2-4 W panels fits the 4 MB L2)
    // and the two tiles of a pair share their W panel.  Row-major order re-read the operands 3x from HBM (rocprofv3 FETCH_SIZE).
    constexpr int GM = 16;
    const int tilesM = ntiles / tilesN;
    auto bm_of = [&](int tile, int& bn) -> int {
        const int per_group = GM * tilesN;
        const int g = tile / per_group, r = tile - g * per_group;
        const int rows = min(GM, tilesM - g * GM);
        bn = r / rows;
        return g * GM + (r - bn * rows);
    };
    const int nk1 = p.K1 / BK;
    const int nk = nk1 + p.K2 / BK;
    const int G = gridDim.x, bid = (int)blockIdx.x;

    const unsigned voA1 = ((unsigned)lr * (unsigned)p.lda1 + csw) * 2u, voA2 = ((unsigned)lr * (unsigned)p.lda2 + csw) * 2u;
    const unsigned voW1 = (wl * (unsigned)p.ldw1 + csw) * 2u, voW2 = (wl * (unsigned)p.ldw2 + csw) * 2u;
    const unsigned voD = ((unsigned)lr * (unsigned)p.down_ldw + csw) * 2u;

    // all loads of K tile T of output tile `tile` into `stage`, spread over the 4 waves of the helper group.
    // MAIN = 1: T is known to lie in the first operand pair (no LoRA-tile selects for the compiler to hoist).
    // address = wave-uniform 64-bit base (SGPRs) + kernel-invariant 32-bit per-lane offset
    auto issue_ktile = [&](auto main_only, int tile, int T, int stage) {
        constexpr bool MAIN = decltype(main_only)::value != 0;
        int bn; const int bm = bm_of(tile, bn);
        const bool ext = !MAIN && T >= nk1;
        const char* Ap = (const char*)(ext ? p.A2 : p.A1);
        const char* Wp = (const char*)(ext ? p.W2 : p.W1);
        const unsigned lda = ext ? p.lda2 : p.lda1, ldw = ext ? p.ldw2 : p.ldw1, k0 = (ext ? T - nk1 : T) * BK;
        const unsigned voa = ext ? voA2 : voA1, vow = ext ? voW2 : voW1;
        h16* dA = sm + stage * STG;
        h16* dW = dA + BM * BK;
        const char* Wb = Wp + ((size_t)(bn * BN) * ldw + k0) * 2u;
        if (ND == 0 || !ext) {          // with ND > 0 the A side of the LoRA tile is written by the helper itself
            const char* Ab = Ap + ((size_t)(bm * BM + wn * 32) * lda + k0) * 2u;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_sv(Ab + (size_t)(i * 8) * lda * 2u, voa, dA + (wn * 32 + i * 8) * BK);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gw = wn * 8 + i;
            const int chunk = gw >> 3, h = (gw >> 2) & 1, gi = gw & 3;
            // LDS row (chunk, h, j' = gi>>1, r' = (gi&1)*8 + lr)  <-  column chunk*64 + 16*(r'>>2) + 4*(2h+j') + (r'&3)
            const int col = PERM ? chunk * 64 + 4 * (2 * h + (gi >> 1)) + 32 * (gi & 1) : gw * 8;
            glds16_sv(Wb + (size_t)col * ldw * 2u, vow, dW + gw * 8 * BK);
        }
        if constexpr (ND > 0) {
            if (!ext) {                 // one 8-row piece of Ad per wave (ND = 1: waves 2, 3 repeat pieces 0, 1)
                const int piece = ND == 1 ? (wn & 1) : wn;
                const char* Db = (const char*)p.down_W + ((size_t)(piece * 8) * p.down_ldw + k0) * 2u;
                glds16_sv(Db, voD, dW + (BN + piece * 8) * BK);
            }
        }
    };

    // slot n -> output tile (-1: none)
    auto tile_of_slot = [&](int n) -> int {
        const int it = n >> 1, h = n & 1;
        const int base = it * 2 * G;
        const int rem = ntiles - base;
        if (rem <= 0) return -1;
        if (rem >= 2 * G) return base + 2 * xcd_remap(bid, G) + h;
        if (rem <= G) {                                   // single tiles: group 0 only
            if (h || bid >= rem) return -1;
            return base + xcd_remap(bid, rem);
        }
        const int cnt = (rem + 1) >> 1;
        if (bid >= cnt) return -1;
        const int t = base + 2 * xcd_remap(bid, cnt) + h;
        return t < ntiles ? t : -1;
    };

    f32x4 acc[8][4];
    const int xo0 = ((0 + fg) ^ (fr & 7)) * 8, xo1 = ((4 + fg) ^ (fr & 7)) * 8;
    const int a_base = fr * BK;                                  // + i*16*BK
    const int w_base = BM * BK + (wn * 64 + fr) * BK;            // + jj*16*BK

    // COMPUTE role, one whole output tile.  Fragment reads run ahead of the MFMAs that use them (64 fragment registers);
    // the step's barrier sits in the MIDDLE of its MFMAs -- after the last LDS read of the stage has returned, with 16
    // MFMAs still queued in the matrix pipe and 32 more to issue behind it -- and the first reads of the NEXT stage follow
    // it directly, so neither the barrier nor the LDS latency drains the pipe.
    auto compute_tile = [&](int st) {
        h16x8 wf0[4], wf1[4], afA[4], afB[4];
        auto rdW = [&](h16x8 (&f)[4], int stage, int xo) {
            const h16* bufW = sm + stage * STG + w_base;
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = *(const h16x8*)(bufW + j * 16 * BK + xo);
        };
        auto rdA = [&](h16x8 (&f)[4], int stage, int half, int xo) {
            const h16* bufA = sm + stage * STG + a_base;
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = *(const h16x8*)(bufA + (half * 4 + i) * 16 * BK + xo);
        };
        auto mm = [&](const h16x8 (&wf)[4], const h16x8 (&af)[4], int half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = mfma16(wf[j], af[i], acc[half * 4 + i][j]);
        };
#define SB() __builtin_amdgcn_sched_barrier(0)
        rdW(wf0, st, xo0); rdA(afA, st, 0, xo0); rdA(afB, st, 1, xo0); SB();
        __builtin_amdgcn_s_setprio(1);
        for (int s = 0; s < nk; ++s) {
            const int nx = st == NSTAGE - 1 ? 0 : st + 1;
            mm(wf0, afA, 0); SB();
            rdW(wf1, st, xo1); rdA(afA, st, 0, xo1); SB();
            mm(wf0, afB, 1); SB();
            rdA(afB, st, 1, xo1); SB();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of this stage has returned
            VMCNT(0);                                                // loads this wave issued as a helper before the role swap
            BARRIER();
            mm(wf1, afA, 0); SB();
            // next stage (certified by the barrier above); after the last step these reads are unused
            rdW(wf0, nx, xo0); rdA(afA, nx, 0, xo0); SB();
            mm(wf1, afB, 1); SB();
            rdA(afB, nx, 1, xo0); SB();
            st = nx;
        }
        __builtin_amdgcn_s_setprio(0);
#undef SB
    };

    // ---- epilogue pieces (helper role): row block c of tile `tile`, this lane's row m and 16 columns from n0 ----
    constexpr int L = epi_loads<EPI>(), ST = epi_stores<EPI>();
    f32x4 bv[BC ? 1 : 4];             // bias of the lane's 16 columns
    // operands requested ahead (gelu' of GELU_BWD, the stream rows of RESID_H16): block c is requested in step c and applied in
    // step c + 2, AFTER which the same step requests block c + 2 into the slot it just read -- two slots suffice
    f32x4 pre[2][4];
    auto request_bias = [&](int tile) {
        int bn; (void)bm_of(tile, bn);
        const int n0 = PERM ? bn * BN + wn * 64 + fg * 16 : bn * BN + wn * 64 + fg * 4;
        if constexpr (!BC) {
            const float* bp = p.bias ? p.bias + n0 : (const float*)p.A1;      // always 4 loads (static vmcnt counts)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = gload16(bp + (PERM ? 4 : 16) * j);
        }
    };
    auto request_block = [&](int tile, auto cc) {
        constexpr int c = decltype(cc)::value;
        int bn; const int bm = bm_of(tile, bn);
        const int m = bm * BM + c * 16 + fr, n0 = bn * BN + wn * 64 + fg * 16;
        if constexpr (EPI == EPI_GELU_BWD || EPI == EPI_RESID_H16) {
            const h16* zs = (const h16*)p.R + (size_t)m * p.ldr + n0;
            pre[c & 1][0] = gload16_once(zs);
            pre[c & 1][1] = gload16_once(zs + 8);
        }
    };
    auto apply_block = [&](int tile, auto cc) {
        constexpr int c = decltype(cc)::value;
        int bn; const int bm = bm_of(tile, bn);
        const int m = bm * BM + c * 16 + fr, n0 = bn * BN + wn * 64 + fg * 16;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (!BC && p.bias) ? acc[c][j] + bv[BC ? 0 : j] : acc[c][j];
        if constexpr (EPI == EPI_STORE_F32) {
            float* dst = (float*)p.C + (size_t)m * p.ldc + bn * BN + wn * 64 + fg * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 16 * q) = v[q];
        } else if constexpr (EPI == EPI_GELU_BWD) {
            epilogue_gelu_bwd16(p, m, n0, v, __builtin_bit_cast(h16x8, pre[c & 1][0]), __builtin_bit_cast(h16x8, pre[c & 1][1]));
        } else if constexpr (EPI == EPI_RESID_H16) {
            epilogue_resid16(p, m, n0, v, __builtin_bit_cast(h16x8, pre[c & 1][0]), __builtin_bit_cast(h16x8, pre[c & 1][1]));
        } else {
            epilogue_row16<EPI>(p, m, n0, v);
        }
    };
    // ---- fused LoRA down projection (helper role, ND > 0): rows 32 wn .. 32 wn + 31 of the compute group's tile ----
    f32x4 tacc[2][ND > 0 ? ND : 1];
    auto t_zero = [&]() {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int d = 0; d < (ND > 0 ? ND : 1); ++d) tacc[ii][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto t_accumulate = [&](int stage) {          // K tile in `stage` (landed: certified by the previous barrier)
        if constexpr (ND > 0) {
            const h16* bufA = sm + stage * STG + a_base + (wn * 32) * BK;
            const h16* bufD = sm + stage * STG + (BM + BN) * BK + fr * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int xo = ks ? xo1 : xo0;
                h16x8 adf[ND], af2[2];
#pragma unroll
                for (int d = 0; d < ND; ++d) adf[d] = *(const h16x8*)(bufD + d * 16 * BK + xo);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) af2[ii] = *(const h16x8*)(bufA + ii * 16 * BK + xo);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int d = 0; d < ND; ++d) tacc[ii][d] = mfma16(adf[d], af2[ii], tacc[ii][d]);
            }
        }
    };
    // t (fp16) becomes the A side of the LoRA K tile in `stage`: lane (fr, fg) holds row fr, columns 16 d + 4 fg .. + 3 of
    // each of its two 16-row tiles; the remaining 16-byte chunks of the 128-byte rows are zeroed (the tile is 64 deep)
    auto t_finalize = [&](int tile, int stage) {
        if constexpr (ND > 0) {
            h16* dA = sm + stage * STG + (wn * 32) * BK;
            int bn_; const int bm = bm_of(tile, bn_);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int row = ii * 16 + fr;
                h16* rp = dA + row * BK;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const h16x4 o = {f2h(tacc[ii][d][0]), f2h(tacc[ii][d][1]), f2h(tacc[ii][d][2]), f2h(tacc[ii][d][3])};
                    const int chunk = 2 * d + (fg >> 1);
                    *(h16x4*)(rp + ((chunk ^ (fr & 7)) * 8) + (fg & 1) * 4) = o;
                    if (p.down_out)
                        *(h16x4*)(p.down_out + (size_t)(bm * BM + wn * 32 + row) * p.down_ld + d * 16 + 4 * fg) = o;
                }
                const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                h16x8 z1 = z;
                z1[7] = (h16)1.0f;                        // BC: column 63 of the tile multiplies the bias column of W2
                *(h16x8*)(rp + (((2 * ND + fg) ^ (fr & 7)) * 8)) = (BC && 2 * ND + fg == 7) ? z1 : z;
                if (2 * ND + 4 + fg < 8) *(h16x8*)(rp + (((2 * ND + 4 + fg) ^ (fr & 7)) * 8)) = (BC && 2 * ND + 4 + fg == 7) ? z1 : z;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    // counted wait with a run-time (wave-uniform) count
    auto vmcnt_rt = [&](int n) {
        switch (n) {
            case 8: VMCNT(8); break;    case 10: VMCNT(10); break;  case 12: VMCNT(12); break;  case 13: VMCNT(13); break;
            case 14: VMCNT(14); break;  case 15: VMCNT(15); break;  case 16: VMCNT(16); break;  case 17: VMCNT(17); break;
            default: VMCNT(0); break;
        }
    };
    // VMEM instructions a helper step issues AFTER its DMA (only when it has an epilogue to run)
    auto post_ops = [](int S) constexpr { return (S == 0 && !BC ? 4 : 0) + ((S >= 2 && S < 10) ? ST : 0) + ((S >= 0 && S < 8) ? L : 0); };
    // helper step S < STATIC_STEPS.  Order inside the step: DMA for K tile S+2, apply block S-2, request block S -- the
    // stores of a step are younger than its DMA, so the wait for that DMA one step later does not wait for them.
    auto helper_static = [&](auto ss, int tile_prev, int t_cur, int stage) {
        constexpr int S = decltype(ss)::value;
        const bool have_prev = tile_prev >= 0;
        issue_ktile(IC<1>{}, t_cur, S + 2, (stage + 2) % NSTAGE);
        if constexpr (S == 0) t_zero();
        t_accumulate(stage);
        if (have_prev) {
            if constexpr (S == 0) request_bias(tile_prev);
            if constexpr (S >= 2 && S < 10) {
                // younger than block (S-2)'s requests: everything of step S-1 and this step's DMA
                constexpr int younger = 2 * NDMA + post_ops(S - 1);
                if constexpr (BC) wait_dep1<younger, L>(pre[(S - 2) & 1]);
                else wait_dep<younger, L>(bv, pre[(S - 2) & 1]);
                apply_block(tile_prev, IC<S - 2>{});
            }
            if constexpr (S < 8) request_block(tile_prev, IC<S>{});
        }
        // the DMA of the previous step has landed; what was issued after it may stay in flight
        constexpr int mine = post_ops(S - 1) + NDMA + post_ops(S);
        if (have_prev) VMCNT(mine); else VMCNT(NDMA);
        BARRIER();
    };

    int t_cur = tile_of_slot(0);
    if (t_cur < 0) return;
    if (grp == 1) {
        issue_ktile(IC<1>{}, t_cur, 0, 0);
        issue_ktile(IC<1>{}, t_cur, 1, 1);
        VMCNT(NDMA);
    }
    BARRIER();

    int pending = -1;        // tile whose accumulators this group still holds
    int stage = 0;
    for (int n = 0; t_cur >= 0; ++n) {
        const int t_next = tile_of_slot(n + 1);
        if ((n & 1) == grp) {
            // ---------------- COMPUTE ----------------
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            compute_tile(stage);
            pending = t_cur;
        } else {
            // ---------------- HELPER ----------------
            int st = stage;
            auto adv = [&]() { st = st == NSTAGE - 1 ? 0 : st + 1; };
            helper_static(IC<0>{}, pending, t_cur, st); adv();
            helper_static(IC<1>{}, pending, t_cur, st); adv();
            helper_static(IC<2>{}, pending, t_cur, st); adv();
            helper_static(IC<3>{}, pending, t_cur, st); adv();
            helper_static(IC<4>{}, pending, t_cur, st); adv();
            helper_static(IC<5>{}, pending, t_cur, st); adv();
            helper_static(IC<6>{}, pending, t_cur, st); adv();
            helper_static(IC<7>{}, pending, t_cur, st); adv();
            helper_static(IC<8>{}, pending, t_cur, st); adv();
            helper_static(IC<9>{}, pending, t_cur, st); adv();
            for (int s = STATIC_STEPS; s < nk; ++s) {
                const int T = s + 2;
                const int tgt = T < nk ? t_cur : t_next;
                int issued = 0;
                if (tgt >= 0) {
                    issue_ktile(IC<0>{}, tgt, T < nk ? T : T - nk, (st + 2) % NSTAGE);
                    issued = (ND > 0 && T < nk && T >= nk1) ? 8 : NDMA;
                }
                int extra = (s == STATIC_STEPS && pending >= 0) ? ST : 0;      // the stores of step 9 are younger than its DMA
                if (ND > 0 && s < nk1) t_accumulate(st);
                if (ND > 0 && s == nk1 - 1) {
                    t_finalize(t_cur, st == NSTAGE - 1 ? 0 : st + 1);
                    if (p.down_out) extra += 2 * ND;
                }
                vmcnt_rt(issued ? issued + extra : 0);
                BARRIER();
                adv();
            }
            pending = -1;
        }
        stage = (stage + nk) % NSTAGE;
        t_cur = t_next;
    }
    // the group that computed the last slot still owes its epilogue
    if (pending >= 0) {
        int bn; const int bm = bm_of(pending, bn);
        const int n0 = PERM ? bn * BN + wn * 64 + fg * 16 : bn * BN + wn * 64 + fg * 4;
        f32x4 b2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b2[j] = (!BC && p.bias) ? *(const f32x4*)(p.bias + n0 + (PERM ? 4 : 16) * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = bm * BM + i * 16 + fr;
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + b2[j];
            if constexpr (PERM) {
                epilogue_row16<EPI>(p, m, n0, v);
            } else {
                float* dst = (float*)p.C + (size_t)m * p.ldc + n0;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 16 * q) = v[q];
            }
        }
    }
}

int g_pp_cus = 0;
int g_pp_mode = 2;          // VITLORA_GEMM_PP: 0 = off, 1 = every supported GEMM, 2 = the shapes it wins on (plain 16-bit stores, K <= 2304)
int g_pp_attr_err = 0;
int g_pp_max_k = 2304;      // VITLORA_GEMM_PP_MAXK: deepest plain-store product sent here in mode 2

template <int EPI, int ND, bool BC = false>
void launch_pp(const GemmArgs& a, hipStream_t s) {
    const int ntiles = (a.M / BM) * (a.N / BN);
    char name[64];
    if (ND) snprintf(name, sizeof name, "gemm_pp_kernel<%d, down %d>", EPI, ND);
    else snprintf(name, sizeof name, "gemm_pp_kernel<%d>", EPI);
    const double valid = a.Mvalid ? (double)a.Mvalid / a.M : 1.0;
    ProfScope prof_(name, 2.0 * a.M * valid * (a.n_algo ? a.n_algo : a.N) * (a.K1 + (a.k2_algo ? a.k2_algo : a.K2)),
                    gemm_algo_bytes(a, EPI, a.M * valid), s, 2.0 * a.M * a.N * (a.K1 + a.K2 + (ND ? 64 : 0)) + (ND ? 2.0 * a.M * 16.0 * ND * a.K1 : 0.0));
    const int units = (ntiles + 1) / 2;
    const int grid = units < g_pp_cus ? (ntiles < g_pp_cus ? ntiles : g_pp_cus) : g_pp_cus;
    const size_t lds = (size_t)NSTAGE * stg_of(ND) * sizeof(h16) + 1024;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, ND, BC>), dim3(grid), dim3(512), lds, s, a, ntiles);
}
template <int EPI, int ND, bool BC = false>
void set_attr_pp() {
    const size_t lds = (size_t)NSTAGE * stg_of(ND) * sizeof(h16) + 1024;
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, ND, BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) g_pp_attr_err = (int)e;
}

bool shape_ok(const GemmArgs& a) {
    if (a.a_gather) return false;
    if (a.N % BN || a.M % BM || a.K1 % BK || a.K2 % BK) return false;
    return a.K1 / BK >= STATIC_STEPS + 2;      // the statically unrolled helper steps load from the first operand pair only
}

}  // namespace

void gemm_pp_set_cus(int n) { g_pp_cus = n; }
int gemm_pp_mode() { return g_pp_mode; }
void gemm_pp_set_mode(int m) { g_pp_mode = m; }

bool gemm_pp_fuses_down(const GemmArgs& a, int epi) {
    // the residual-add epilogue combines with the fused down projection only with the bias in the LoRA K tile (BC): its two row
    // operands requested ahead leave no room for 16 bias registers
    if (g_pp_mode == 0 || !a.down_W || gemm_small_forced()) return false;
    if (!(epi == EPI_STORE_H16 || (epi == EPI_RESID_H16 && a.ones_col && !a.bias))) return false;
    if (a.ones_col && epi != EPI_RESID_H16) return false;
    if (a.down_groups < 1 || a.down_groups > 2 || a.K2 != BK || !a.W2) return false;
    return shape_ok(a);
}

bool gemm_pp_supports(const GemmArgs& a, int epi) {
    if (g_pp_mode == 0) return false;
    if (a.down_W) return gemm_pp_fuses_down(a, epi);
    // epilogues that READ a second operand keep it in flight in registers across steps (explicit loads); the fp32 residual
    // form does not fit the 256-register budget next to the accumulators and stays on gemm256
    if (!(epi == EPI_STORE_H16 || epi == EPI_GELU || epi == EPI_GELU_BWD || epi == EPI_STORE_F32 || epi == EPI_NONE || epi == EPI_RESID_H16)) return false;
    // measured (tools/gemm_pp_check.py, MI355X): ahead of gemm256 by 7 - 12 % on the plain 16-bit-store shapes with K <= 2304
    // (qkv forward, o / qkv dgrad), level at K = 3072, behind on the GELU epilogues (4 helper waves carry the erf VALU)
    if (g_pp_mode == 2 && !((epi == EPI_STORE_H16 || epi == EPI_RESID_H16) && a.K1 <= g_pp_max_k)) return false;
    return shape_ok(a);
}

int gemm_pp_init() {
    g_pp_attr_err = 0;
    if (const char* e = getenv("VITLORA_GEMM_PP")) g_pp_mode = atoi(e);
    if (const char* e = getenv("VITLORA_GEMM_PP_MAXK")) g_pp_max_k = atoi(e);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_pp_cus = prop.multiProcessorCount;
    if (g_pp_cus <= 0) g_pp_cus = 256;
    set_attr_pp<EPI_STORE_H16, 0>(); set_attr_pp<EPI_GELU, 0>(); set_attr_pp<EPI_GELU_BWD, 0>();
    set_attr_pp<EPI_STORE_F32, 0>(); set_attr_pp<EPI_NONE, 0>();
    set_attr_pp<EPI_STORE_H16, 1>(); set_attr_pp<EPI_STORE_H16, 2>();
    set_attr_pp<EPI_RESID_H16, 0>(); set_attr_pp<EPI_RESID_H16, 1, true>(); set_attr_pp<EPI_RESID_H16, 2, true>();
    return g_pp_attr_err;
}

void launch_gemm_pp(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.down_W) {
        if (epi == EPI_RESID_H16) { if (a.down_groups == 1) launch_pp<EPI_RESID_H16, 1, true>(a, s); else launch_pp<EPI_RESID_H16, 2, true>(a, s); }
        else if (a.down_groups == 1) launch_pp<EPI_STORE_H16, 1>(a, s); else launch_pp<EPI_STORE_H16, 2>(a, s);
        return;
    }
    switch (epi) {
        case EPI_STORE_H16: launch_pp<EPI_STORE_H16, 0>(a, s); break;
        case EPI_GELU: launch_pp<EPI_GELU, 0>(a, s); break;
        case EPI_GELU_BWD: launch_pp<EPI_GELU_BWD, 0>(a, s); break;
        case EPI_RESID_H16: launch_pp<EPI_RESID_H16, 0>(a, s); break;
        case EPI_STORE_F32: launch_pp<EPI_STORE_F32, 0>(a, s); break;
        case EPI_NONE: launch_pp<EPI_NONE, 0>(a, s); break;
    }
}

}  // namespace VLNS
