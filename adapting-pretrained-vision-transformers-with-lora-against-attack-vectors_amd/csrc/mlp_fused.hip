// Fused MLP of a NARROW transformer block (Swin-T stage 1: C = 96, hidden 384, 802 816 token rows at batch 256) -- round 5.
//
// On such a block the two MLP products are HBM streams whose bytes are almost all the HIDDEN activation: fc1 writes gelu(z)
// and gelu'(z) (2 x 616 MB), fc2 reads gelu(z) again; the backward writes and re-reads d(z) the same way.  Here the hidden
// activation never leaves the CU: one persistent workgroup per CU walks 128-row tiles and, per tile, 128-column CHUNKS of the
// hidden dimension --
//
//     forward   z_c = x W1_c^T + b1_c          (GEMM 1: K = C)        a_c = gelu(z_c) -> LDS,  gelu'(z_c) -> HBM (the backward needs it)
//               y  += a_c W2[:, c]^T           (GEMM 2: K = chunk)    (+ LoRA of fc2: t += a_c Ad[:, c]^T, then y += t (sB)^T;  + b2)
//     backward  da_c = g W2[:, c] (+ LoRA: u = g Bd^T once per tile, da_c += u (sA)_c^T)        dz_c = da_c * gelu'(z_c) -> LDS
//               dx  += dz_c W1_c               (GEMM 2: K = chunk)
//
// -- so a row tile moves x in, gelu' out (or in) and y out: 0.92 GB per launch at batch 256 instead of 2.0 + 0.9 GB for the two
// separate launches.  Skeleton of gemm_stream.hip (csrc): eight multiplier waves (4 x 2, 32 x 64 outputs each, 16x16x32 MFMA,
// operands swapped so that a lane owns 4 adjacent columns), two loader waves that issue every LDS-DMA three steps ahead through a
// three-stage ring and wait with counted vmcnt, one s_barrier per step.  A "step" is one 64-deep K tile of GEMM 1, of GEMM 2, or a
// LoRA K tile; the chunk activation is written by the multipliers in the A-stage layout and read back as GEMM 2's A operand (as
// gemm_stream's fused LoRA down projection does with t).  Same MFMA order per output element as the separate launches for
// GEMM 1; GEMM 2 sums its K in the same order too (chunks in order), so results agree with the unfused path to the rounding of
// the LoRA tile (tested against it and against the oracle through the Swin tests).
#include <cstdio>
#include <cstdlib>

#include "gemm_epi.h"
#include "mlp_fused.h"
#include "prof.h"

namespace VLNS {

namespace {

constexpr int BK = 64, BM = 128, BN = 128, CW = 8, STAGES = 3;
constexpr int DN = 16;            // rows of the LoRA down matrix staged per step (one adapted module, r <= 16)
constexpr int TK = 32;            // depth of the LoRA K tile as the multipliers read it (r <= 16 of it non-zero)
constexpr size_t mlp_lds() {
    return ((size_t)STAGES * (BM + BN + DN) * BK + 2 * BM * BK + BM * TK + (size_t)CW * 16 * (BN / 2)) * sizeof(h16);
}

template <bool BWD>
__global__ __launch_bounds__(64 * (CW + 2)) void mlp_fused_kernel(const MlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sA = (h16*)smem;                       // [STAGES][BM][BK]   x (forward) / g (backward) K tiles
    h16* sW = sA + STAGES * BM * BK;            // [STAGES][BN][BK]   weight K tiles of the step
    h16* sD = sW + STAGES * BN * BK;            // [STAGES][DN][BK]   LoRA down matrix K tiles
    h16* sAct = sD + STAGES * DN * BK;          // [2][BM][BK]        the chunk activation (a_c / dz_c), two A stages
    h16* sT = sAct + 2 * BM * BK;               // [BM][TK]           t / u of the row tile (columns r .. TK-1 stay zero)
    h16* sImg = sT + BM * TK;                   // [CW][16][BN / 2]   per-wave staging of result rows (full-line stores)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = p.M / BM, G = gridDim.x, g0 = blockIdx.x;
    const int nk1 = p.K1 / BK, nch = p.HID / BN;
    // LoRA sits on GEMM 2 in the forward (fc2's down projection reads the activation) and on GEMM 1 in the backward
    const bool lora = p.lora != 0;
    const int x1 = (BWD && lora) ? 1 : 0;                    // LoRA K step after GEMM 1's K steps (per chunk)
    const int per_chunk = nk1 + x1 + 2;
    const int per_tile = nch * per_chunk + ((!BWD && lora) ? 1 : 0);
    const int my_rows = g0 < tilesM ? (tilesM - g0 + G - 1) / G : 0;
    const int S = my_rows * per_tile;
    const int lr = lane >> 3, lc = lane & 7;
    constexpr int LA = BM / 8, LW = BN / 8;

    // step t of a row tile: chunk j, position q inside the chunk; kinds: 0 = GEMM 1 K tile q, 1 = GEMM 1 LoRA tile, 2 = GEMM 2 K tile
    // (q - nk1 - x1), 3 = GEMM 2 LoRA tile (after the last chunk)
    struct Step { int kind, j, k; };
    auto decode = [&](int t) -> Step {
        if (t >= nch * per_chunk) return {3, nch - 1, 0};
        const int j = t / per_chunk, q = t - j * per_chunk;
        if (q < nk1) return {0, j, q};
        if (q < nk1 + x1) return {1, j, 0};
        return {2, j, q - nk1 - x1};
    };

    if (w >= CW) {
        // ---------------------------------------------------------------- loaders ----------------------------------------
        const bool isA = w == CW;
        const int c8 = (lc ^ lr) * 8;
        int it = 0, ibm = g0;                       // next step to issue: step `it` of row tile ibm
        auto a_loads = [&](int t) { return decode(t).kind == 0 ? LA : 0; };      // the A loader has work in GEMM 1's K steps only
        auto issue = [&](int s) {
            const Step st = decode(it);
            const int bm = ibm;
            if (++it == per_tile) { it = 0; ibm += G; }
            const int buf = s % STAGES;
            if (isA) {
                if (st.kind != 0) return;
                const h16* src = p.X + (size_t)(bm * BM + lr) * p.ldx + st.k * BK + c8;
                h16* dst = sA + buf * BM * BK;
#pragma unroll
                for (int i = 0; i < LA; ++i) glds16(src + (size_t)i * 8 * p.ldx, dst + i * 8 * BK);
            } else {
                const h16* Wp; int ldw, row0, k0;
                if (st.kind == 0) { Wp = p.Wa; ldw = p.ldwa; row0 = st.j * BN; k0 = st.k * BK; }
                else if (st.kind == 1) { Wp = p.Lup; ldw = BK; row0 = st.j * BN; k0 = 0; }           // (s A)^T rows of the chunk
                else if (st.kind == 2) { Wp = p.Wb; ldw = p.ldwb; row0 = 0; k0 = st.j * BN + st.k * BK; }
                else { Wp = p.Lup; ldw = BK; row0 = 0; k0 = 0; }                                   // (s B) rows
                const h16* src = Wp + (size_t)(row0 + lr) * ldw + k0 + c8;
                h16* dst = sW + buf * BN * BK;
#pragma unroll
                for (int i = 0; i < LW; ++i) glds16(src + (size_t)i * 8 * ldw, dst + i * 8 * BK);
                // LoRA down matrix K tile (16 rows): backward with GEMM 1's K tile, forward with GEMM 2's; loaded every step so
                // that every stage of this loader is the same number of instructions for the counted waits
                const int dk0 = lora ? (BWD ? (st.kind == 0 ? st.k * BK : 0) : (st.kind == 2 ? st.j * BN + st.k * BK : 0)) : 0;
                const h16* dsrc = (lora ? p.Ldown : p.Wa) + (size_t)lr * (lora ? p.ldd : p.ldwa) + dk0 + c8;
                h16* ddst = sD + buf * DN * BK;
#pragma unroll
                for (int i = 0; i < DN / 8; ++i) glds16(dsrc + (size_t)i * 8 * (lora ? p.ldd : p.ldwa), ddst + i * 8 * BK);
            }
        };
        // the A loader's loads per step vary (GEMM 1's K steps only): its waits count the loads of the steps issued after step s
        int la_hist[STAGES];                                   // loads issued for the step that went into ring slot (step % STAGES)
        int issued = 0;
        auto issue_counted = [&](int s) {
            if (isA) { const int t = it; la_hist[s % STAGES] = a_loads(t); }
            issue(s);
            ++issued;
        };
        for (int s = 0; s < STAGES - 1 && s < S; ++s) issue_counted(s);
        constexpr int LPS_W = LW + DN / 8;
        for (int s = 0; s < S; ++s) {
            const int ahead = min(S - 1, s + STAGES - 2) - s;
            if (isA) {
                int pend = 0;
                for (int a = 1; a <= ahead; ++a) pend += la_hist[(s + a) % STAGES];
                if (pend >= 2 * LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LA) : "memory");
                else if (pend == LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS_W) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS_W) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (s + STAGES - 1 < S) issue_counted(s + STAGES - 1);
        }
        return;
    }

    // -------------------------------------------------------------------- multipliers ------------------------------------
    const int wm = w >> 1, wn = w & 1;
    constexpr int NJ = BN / 32, WROWS = BM / (CW / 2), MI = WROWS / 16;
    const int fr = lane & 15, fg = lane >> 4;
    // sT's columns beyond the LoRA rank are never written: zero them once (published by the first step's barrier)
    for (int i = tid; i < BM * TK / 8; i += 64 * CW) ((h16x8*)sT)[i] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc1[MI][NJ], acc2[MI][NJ], tacc[MI];
    char* img = (char*)(sImg + w * 16 * (BN / 2));
    constexpr int WCOLS = BN / 2, RB = WCOLS * 2, CPR = RB / 16;
    // full-line stores of a 32 x 64 wave tile through the wave's LDS image (gemm_stream.hip); columns n0 + [0, 64), stored where < nlim
    // one 16-row group of the wave tile (16 x 64) out through the wave's LDS image: full 128-byte rows
    auto flush16 = [&](const h16x4 (&o)[NJ], h16* Cp, int ldcp, int row0, int n0, int nlim) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int chunk = j * 2 + (fg >> 1);
            *(h16x4*)(img + fr * RB + ((chunk ^ (fr & (CPR - 1))) << 4) + (fg & 1) * 8) = o[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int RPI = 64 / CPR;
#pragma unroll
        for (int it2 = 0; it2 < 16 / RPI; ++it2) {
            const int rl = it2 * RPI + lane / CPR, chunk = lane % CPR;
            const h16x8 v = *(const h16x8*)(img + rl * RB + ((chunk ^ (rl & (CPR - 1))) << 4));
            const int n = n0 + chunk * 8;
            if (n < nlim) *(h16x8*)(Cp + (size_t)(row0 + rl) * ldcp + n) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // one 16-row group into the chunk activation's A stage `wn` (row r: 16-byte chunk c at c ^ (r & 7))
    auto to_act16 = [&](const h16x4 (&o)[NJ], int i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = wm * WROWS + i * 16 + fr, chunk = j * 2 + (fg >> 1);
            *(h16x4*)(sAct + wn * BM * BK + r * BK + ((chunk ^ (r & 7)) << 3) + (fg & 1) * 4) = o[j];
        }
    };
    int t = 0, bm = g0;
    for (int s = 0; s < S; ++s) {
        const Step st = decode(t);
        const int cbm = bm, ct = t;
        if (++t == per_tile) { t = 0; bm += G; }
        if (ct == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                tacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (st.kind == 0 && st.k == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS writes (activation, t) are done before the barrier publishes them
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const h16* cW = sW + (s % STAGES) * BN * BK;
        const h16* cD = sD + (s % STAGES) * DN * BK;
        // (separate straight-line bodies per accumulator: with the accumulator chosen inside the loops the compiler keeps both
        //  sets live through selects and spills)
        auto lora_step = [&](f32x4 (&acc)[MI][NJ]) {
            // ---- LoRA K tile: A operand = t / u in LDS (TK deep), W = the scaled up matrix's rows ----
            h16x8 af[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WROWS + i * 16 + fr;
                af[i] = *(const h16x8*)(sT + r * TK + ((fg ^ (r & 3)) << 3));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fr;
                wf[j] = *(const h16x8*)(cW + r * BK + ((fg ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        };
        // the down projection accumulates from the A operand of GEMM 1 (backward: u = g Bd^T, first chunk only) or GEMM 2
        // (forward: t = a Ad^T, every chunk); one 16-column tile, computed by the wn == 0 waves for their 32 rows
        auto gemm_step = [&](f32x4 (&acc)[MI][NJ], const h16* cA, bool t_step) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h16x8 af[MI], wf[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = wm * WROWS + i * 16 + fr;
                    af[i] = *(const h16x8*)(cA + r * BK + (((ks * 4 + fg) ^ (r & 7)) << 3));
                }
                if (t_step) {
                    const h16x8 df = *(const h16x8*)(cD + fr * BK + (((ks * 4 + fg) ^ (fr & 7)) << 3));
#pragma unroll
                    for (int i = 0; i < MI; ++i) tacc[i] = mfma16(df, af[i], tacc[i]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r = wn * (BN / 2) + j * 16 + fr;
                    wf[j] = *(const h16x8*)(cW + r * BK + (((ks * 4 + fg) ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
            }
        };
        if (st.kind == 0) gemm_step(acc1, sA + (s % STAGES) * BM * BK, lora && BWD && wn == 0 && st.j == 0);
        else if (st.kind == 2) gemm_step(acc2, sAct + st.k * BM * BK, lora && !BWD && wn == 0);
        else if (st.kind == 1) lora_step(acc1);
        else lora_step(acc2);
        // ---- t / u complete: rounded once into sT (row r: 16-byte chunk c at c ^ (r & 3); this lane's 4 columns 4 fg ..) ----
        const bool t_done = lora && wn == 0 && (BWD ? (st.kind == 0 && st.j == 0 && st.k == nk1 - 1) : (st.kind == 2 && st.j == nch - 1 && st.k == 1));
        if (t_done) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WROWS + i * 16 + fr, chunk = fg >> 1;
                h16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = f2h(tacc[i][k]);
                *(h16x4*)(sT + r * TK + ((chunk ^ (r & 3)) << 3) + (fg & 1) * 4) = o;
            }
        }
        // ---- GEMM 1 of chunk j complete: the elementwise stage, result into the activation stages ----
        if ((st.kind == 0 && st.k == nk1 - 1 && !x1) || st.kind == 1) {
            const int nb = st.j * BN + wn * (BN / 2) + fg * 4;          // this lane's hidden column of column tile jj: nb + 16 jj
            const int mb = cbm * BM + wm * WROWS + fr;
            if constexpr (!BWD) {
                f32x4 bv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = *(const f32x4*)(p.bias1 + nb + 16 * j);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(bv[j][k]));
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    h16x4 oa[NJ], og[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f32x4 v = acc1[i][j] + bv[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const GeluParts gp = gelu_parts(v[k]);
                            oa[j][k] = f2h(v[k] * gp.cdf);
                            og[j][k] = f2h(fmaf(v[k], gp.pdf, gp.cdf));      // gelu'(z): stored for the backward
                        }
                    }
                    to_act16(oa, i);
                    flush16(og, p.S, p.lds_, cbm * BM + wm * WROWS + i * 16, st.j * BN + wn * WCOLS, p.HID);
                }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    h16x4 rz[NJ], od[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) rz[j] = *(const h16x4*)(p.S + (size_t)(mb + 16 * i) * p.lds_ + nb + 16 * j);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) od[j][k] = f2h(acc1[i][j][k] * h2f(rz[j][k]));
                    to_act16(od, i);
                }
            }
        }
        // ---- row tile complete: bias, store y ----
        if (ct == per_tile - 1) {
            const int nb = wn * (BN / 2) + fg * 4;
            f32x4 bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = p.bias2 ? *(const f32x4*)(p.bias2 + nb + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(bv[j][k]));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                h16x4 o2[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 v = acc2[i][j] + bv[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o2[j][k] = f2h(v[k]);
                }
                flush16(o2, p.Y, p.ldy, cbm * BM + wm * WROWS + i * 16, wn * WCOLS, p.n_store);
            }
        }
    }
}

int g_mlp_cus = 0, g_mlp_err = 0, g_mlp_on = 1;
int g_mlp_min_rows = 65536;      // tall products only (VITLORA_MLP_FUSED_MIN_ROWS: tests lower it to run the kernel at small batches)

}  // namespace

int mlp_fused_init() {
    g_mlp_err = 0;
    for (const void* f : {(const void*)mlp_fused_kernel<false>, (const void*)mlp_fused_kernel<true>})
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_lds()) != hipSuccess) g_mlp_err = 1;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_mlp_cus = cus;
    if (const char* e = getenv("VITLORA_MLP_FUSED")) g_mlp_on = e[0] != '0';
    g_mlp_min_rows = 65536;
    if (const char* e = getenv("VITLORA_MLP_FUSED_MIN_ROWS")) g_mlp_min_rows = atoi(e);
    return g_mlp_err;
}

// shapes the kernel covers: output width and GEMM 1 depth of at most 128 (one column tile, K1 a multiple of 64), hidden a multiple
// of 128, row count a multiple of 128, at most one adapted module with r <= 16 on the projection that carries the LoRA
bool mlp_fused_supports(const MlpArgs& a) {
    if (!g_mlp_on || g_mlp_err) return false;
    if (a.M % BM || a.M < g_mlp_min_rows || a.K1 % BK || a.K1 < BK || a.K1 > 128 || a.HID % BN || a.n_store > BN || a.n_store <= 0) return false;
    if (a.lora && (!a.Ldown || !a.Lup)) return false;
    return true;
}

void launch_mlp_fused(const MlpArgs& a, int backward, hipStream_t s) {
    const int tilesM = a.M / BM;
    const int G = tilesM < g_mlp_cus ? tilesM : g_mlp_cus;
    const double rows = a.Mvalid ? a.Mvalid : a.M;
    const double flops = 2.0 * rows * a.HID * (double)(a.K1_algo + a.n_store) + (a.lora ? 2.0 * rows * a.r_algo * (a.HID + (backward ? a.K1_algo : a.n_store)) : 0.0);
    const double bytes = rows * 2.0 * (a.K1_algo + a.HID + a.n_store);
    ProfScope prof_(backward ? "mlp_fused_kernel<true>" : "mlp_fused_kernel<false>", flops, bytes, s, flops);
    if (backward) hipLaunchKernelGGL((mlp_fused_kernel<true>), dim3(G), dim3(64 * (CW + 2)), mlp_lds(), s, a);
    else hipLaunchKernelGGL((mlp_fused_kernel<false>), dim3(G), dim3(64 * (CW + 2)), mlp_lds(), s, a);
}

}  // namespace VLNS
