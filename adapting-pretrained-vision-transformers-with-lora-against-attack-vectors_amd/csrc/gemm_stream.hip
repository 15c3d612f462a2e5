// Streaming GEMM for tall, shallow products (M >> N, K <= a few hundred: the Swin stage-1 / stage-2 linears at batch 256,
// M = 802 816 rows, K = 128 .. 512): such a product is an HBM stream -- A in, C out, W resident in L2 -- and what decides its
// time is how many bytes a CU keeps in flight, not its MFMA rate.  gemm_nt_kernel runs it as independent 128-row tiles
// (load -> wait -> multiply -> store, two workgroups per CU, one 32 KB stage of loads in flight each): 1.6 - 2.7x the HBM
// time of its bytes.  Here:
//   * ONE persistent workgroup per CU walks row tiles bm = g, g + G, ... and, inside a row tile, every 128-column tile;
//   * waves 0-3 multiply and store (2 x 2 waves, 64 x 64 outputs each, 16x16x32 h16 MFMA, the epilogues of gemm_epi.h);
//     waves 4 and 5 are LOADERS (A rows / W rows): they issue every LDS-DMA of the workgroup, run three 64-deep K steps
//     ahead of the multipliers through a four-stage LDS ring, ACROSS tile boundaries, and wait with counted vmcnt on queues
//     that hold nothing but their own loads (a wave that also stores cannot count: its stores sit in the same queue);
//   * one s_barrier per K step hands a landed stage to the multipliers and a consumed one back to the loaders.
// The multipliers' epilogue (bias, GELU, stores) of tile t therefore runs under the loads of tile t + 1.
// Same arithmetic as gemm_nt_kernel (same MFMA order per output element): results are bit-identical.
#include <cstdio>
#include <cstdlib>

#include "gemm_epi.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

constexpr int BK = 64, BM = 128, CW = 8;      // CW compute waves (CW / 2 x 2), then 2 loader waves
constexpr int DN = 64;                        // columns of the LoRA down projection computed in-kernel (DOWN): t = A1 down_W^T
template <bool DOWN> constexpr int stages_of() { return DOWN ? 3 : 4; }
template <int BN, bool DOWN>
constexpr size_t stream_lds() {
    return ((size_t)stages_of<DOWN>() * (BM + BN + (DOWN ? DN : 0)) * BK + (DOWN ? BM * DN : 0) + (size_t)CW * 16 * (BN / 2)) * sizeof(h16);
}

// DOWN: the LoRA down projection rides along.  t = A1 down_W^T (64 columns, zero padded) is accumulated by the multipliers from
// the very A stages of the row tile's FIRST column tile (down_W's K tile arrives with W's), written to LDS in the A-stage
// format, and every column tile's last K step (the LoRA K tile, W2 = sB) reads its A operand from there: the separate skinny
// GEMM over the activation and the t round trip through HBM disappear.  Three stages instead of four (40 KB each).
template <int BN, int EPI, bool DOWN>
__global__ __launch_bounds__(64 * (CW + 2)) void gemm_stream_kernel(const GemmArgs p) {
    constexpr int STAGES = stages_of<DOWN>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sA = (h16*)smem;                   // [STAGES][BM][BK]
    h16* sW = sA + STAGES * BM * BK;         // [STAGES][BN][BK]
    h16* sD = sW + STAGES * BN * BK;         // DOWN: [STAGES][DN][BK]  K tile of down_W
    h16* sT = sD + (DOWN ? STAGES * DN * BK : 0);      // DOWN: [BM][DN]  t of the current row tile, A-stage format
    h16* sImg = sT + (DOWN ? BM * DN : 0);   // [CW][16][BN / 2]: per-wave staging of the result rows (full-line stores)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesN = p.N / BN, tilesM = p.M / BM;
    const int G = gridDim.x, g0 = blockIdx.x;
    const int nk1 = p.K1 / BK, nk = nk1 + p.K2 / BK;
    const int my_rows = g0 < tilesM ? (tilesM - g0 + G - 1) / G : 0;     // row tiles of this workgroup
    const int S = my_rows * tilesN * nk;                                   // K steps of this workgroup
    const int lr = lane >> 3, lc = lane & 7;
    constexpr int LA = BM / 8, LW = BN / 8;        // 1 KiB load instructions per stage: A loader / W loader

    // the workgroup's steps in order: K steps of a tile, column tiles of a row tile, row tiles g0, g0 + G, ...
    struct Pos { int bm, bn, kt; };
    auto advance = [&](Pos& q) {
        if (++q.kt == nk) { q.kt = 0; if (++q.bn == tilesN) { q.bn = 0; q.bm += G; } }
    };

    if (w >= CW) {
        // ---------------------------------------------------------------- loaders ----------------------------------------
        const bool isA = w == CW;
        const int c8 = (lc ^ lr) * 8;                // swizzled source chunk: row & 7 == lr for every load group
        Pos pi = {g0, 0, 0};                          // next step to issue
        auto issue = [&](int s) {
            const int bm = pi.bm, bn = pi.bn, kt = pi.kt;
            advance(pi);
            const bool ext = kt >= nk1;
            const int k0 = (ext ? kt - nk1 : kt) * BK;
            const int buf = s % STAGES;
            if (isA) {
                // DOWN: the LoRA K step reads t from LDS; its A stage is filled with the tile's first K tile again (unused),
                // so that every stage is the same number of loads for the counted waits
                const h16* Ap = (ext && !DOWN) ? p.A2 : p.A1;
                const int lda = (ext && !DOWN) ? p.lda2 : p.lda1;
                const h16* src = Ap + (size_t)(bm * BM + lr) * lda + ((ext && DOWN) ? 0 : k0) + c8;
                h16* dst = sA + buf * BM * BK;
#pragma unroll
                for (int i = 0; i < LA; ++i) glds16(src + (size_t)i * 8 * lda, dst + i * 8 * BK);
            } else {
                const h16* Wp = ext ? p.W2 : p.W1;
                const int ldw = ext ? p.ldw2 : p.ldw1;
                const h16* src = Wp + (size_t)(bn * BN + lr) * ldw + k0 + c8;
                h16* dst = sW + buf * BN * BK;
#pragma unroll
                for (int i = 0; i < LW; ++i) glds16(src + (size_t)i * 8 * ldw, dst + i * 8 * BK);
                if constexpr (DOWN) {       // (read by the first column tile's main steps only; loaded every step: see above)
                    const h16* dsrc = p.down_W + (size_t)lr * p.down_ldw + (ext ? 0 : k0) + c8;
                    h16* ddst = sD + buf * DN * BK;
#pragma unroll
                    for (int i = 0; i < DN / 8; ++i) glds16(dsrc + (size_t)i * 8 * p.down_ldw, ddst + i * 8 * BK);
                }
            }
        };
        constexpr int LPS_A = LA, LPS_W = LW + (DOWN ? DN / 8 : 0);
        for (int s = 0; s < STAGES - 1 && s < S; ++s) issue(s);
        for (int s = 0; s < S; ++s) {
            // stage s landed; the (up to two) stages issued after it may stay in flight
            const int ahead = min(S - 1, s + STAGES - 2) - s;
            if (isA) {
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS_A) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS_A) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS_W) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS_W) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // every multiplier passed this barrier after its reads of stage s - 1: that buffer may be refilled
            if (s + STAGES - 1 < S) issue(s + STAGES - 1);
        }
        return;
    }

    // -------------------------------------------------------------------- multipliers ------------------------------------
    const int wm = w >> 1, wn = w & 1;
    constexpr int NJ = BN / 32;          // 16-wide column tiles per wave
    constexpr int WROWS = BM / (CW / 2);   // rows per wave
    constexpr int MI = WROWS / 16;         // 16-high row tiles per wave
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MI][NJ];
    f32x4 tacc[MI][2];                       // DOWN: this wave's 32 x 32 block of t (columns wn * 32 ..)
    Pos pc = {g0, 0, 0};
    for (int s = 0; s < S; ++s) {
        const int bm = pc.bm, bn = pc.bn, kt = pc.kt;
        advance(pc);
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool ext_step = DOWN && kt >= nk1;                  // the LoRA K step: A operand = t in LDS
        const bool t_step = DOWN && bn == 0 && kt < nk1;          // accumulate t from this A stage
        if (DOWN && bn == 0 && kt == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) { tacc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; tacc[i][1] = tacc[i][0]; }
        }
        const h16* cA = ext_step ? sT : sA + (s % STAGES) * BM * BK;
        const h16* cW = sW + (s % STAGES) * BN * BK;
        const h16* cD = sD + (s % STAGES) * DN * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 af[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wm * WROWS + i * 16 + fr;
                const int c = (ks * 4 + fg) ^ (r & 7);
                af[i] = *(const h16x8*)(cA + r * BK + c * 8);
            }
            if constexpr (DOWN) {
                if (t_step) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int r = wn * 32 + j2 * 16 + fr;
                        const h16x8 df = *(const h16x8*)(cD + r * BK + (((ks * 4 + fg) ^ (r & 7)) << 3));
#pragma unroll
                        for (int i = 0; i < MI; ++i) tacc[i][j2] = mfma16(df, af[i], tacc[i][j2]);
                    }
                }
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
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);   // D[n][m]: lane owns 4 consecutive n
        }
        if constexpr (DOWN) {
            if (bn == 0 && kt == nk1 - 1) {
                // t of this row tile, rounded once, in the A-stage layout (row r: 16-byte chunk c at c ^ (r & 7)); the next
                // step's barrier publishes it, and it is rewritten only after every wave has passed the row tile's last K step
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int r = wm * WROWS + i * 16 + fr, chunk = wn * 4 + j2 * 2 + (fg >> 1);
                        h16x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = f2h(tacc[i][j2][k]);
                        *(h16x4*)(sT + r * DN + ((chunk ^ (r & 7)) << 3) + (fg & 1) * 4) = o;
                    }
            }
        }
        if (kt == nk - 1) {
            // the fragment reads above are complete before the accumulators are final (the MFMAs consumed them), so the next
            // barrier may hand this stage back while the epilogue runs.
            // Two phases: every load (bias, saved gelu') and all arithmetic first, then nothing but stores -- with loads
            // pending between guarded stores hipcc falls back to s_waitcnt vmcnt(0) at every branch join, i.e. each store
            // waits for the previous one to be acknowledged.
            const int nb = bn * BN + wn * (BN / 2) + fg * 4;          // this lane's column of column tile j: nb + 16 j
            const int mb = bm * BM + wm * WROWS + fr;                 // row of row tile i: mb + 16 i
            bool live[NJ];                                            // wave-uniform: n_store is a multiple of 16
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                live[j] = !p.n_store || __builtin_amdgcn_readfirstlane(bn * BN + wn * (BN / 2) + j * 16) < p.n_store;
            f32x4 bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = *(const f32x4*)(p.bias + nb + 16 * j);
            }
            // the loaded values are "used" here, so the compiler's wait for them sits here and not in front of each store
            // (there it reads "at most k operations outstanding", which also drains the stores issued so far)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(bv[j][k]));
            h16x4 o1[MI][NJ];
            if constexpr (EPI == EPI_GELU_BWD) {
                h16x4 rz[MI][NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        // (dead column tiles re-read tile 0's columns: straight-line loads, no branch for the waitcnt pass to join)
                        rz[i][j] = *(const h16x4*)((const h16*)p.R + (size_t)(mb + 16 * i) * p.ldr + nb + (live[j] ? 16 * j : 0));
                    }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        u32x2 t = __builtin_bit_cast(u32x2, rz[i][j]);
                        asm volatile("" : "+v"(t[0]), "+v"(t[1]));
                        rz[i][j] = __builtin_bit_cast(h16x4, t);
                    }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f32x4 v = acc[i][j] + bv[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) o1[i][j][k] = f2h(v[k] * h2f(rz[i][j][k]));
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f32x4 v = acc[i][j] + bv[j];
                        if constexpr (EPI == EPI_GELU) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const GeluParts gp = gelu_parts(v[k]);
                                o1[i][j][k] = f2h(v[k] * gp.cdf);
                                acc[i][j][k] = fmaf(v[k], gp.pdf, gp.cdf);        // gelu'(z): second output, staged after the first
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) o1[i][j][k] = f2h(v[k]);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            // Results leave as FULL row segments: a lane's MFMA result is 4 columns (8 B) and a wave instruction would write 16
            // rows x 32 B; through a wave-private LDS image (32 rows x BN/2 columns, 16-byte chunks XOR-swizzled by the row) a
            // wave instruction writes 8 rows (BN = 128: 128 B each, 16 B per lane) instead.
            constexpr int WCOLS = BN / 2, RB = WCOLS * 2, CPR = RB / 16;       // columns / bytes / 16-byte chunks per image row
            char* img = (char*)(sImg + w * 16 * WCOLS);
            auto flush = [&](const h16x4 (&o)[MI][NJ], h16* Cp, int ldcp) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int chunk = j * 2 + (fg >> 1);
                        *(h16x4*)(img + fr * RB + ((chunk ^ (fr & (CPR - 1))) << 4) + (fg & 1) * 8) = o[i][j];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int rl = it * RPI + lane / CPR, chunk = lane % CPR;
                        const h16x8 v = *(const h16x8*)(img + rl * RB + ((chunk ^ (rl & (CPR - 1))) << 4));
                        const int n = bn * BN + wn * WCOLS + chunk * 8;
                        if (!p.n_store || n < p.n_store)
                            *(h16x8*)(Cp + (size_t)(bm * BM + wm * WROWS + i * 16 + rl) * ldcp + n) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is rewritten by the next row group
                }
            };
            flush(o1, (h16*)p.C, p.ldc);
            if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) o1[i][j][k] = f2h(acc[i][j][k]);
                flush(o1, (h16*)p.C2, p.ldc2);
            }
        }
    }
}

int g_stream_cus = 0;
int g_stream_on = 1;            // VITLORA_GEMM_STREAM=0: every such product on gemm_nt_kernel
int g_stream_min_rows = 65536;  // tall ...
int g_stream_min_rows64 = 8192;  // (the 64-column form: the LoRA down products that stay separate launches)
int g_stream_max_k = 832;       // ... and shallow (round 5: 640 -> 832 takes Swin stage 2's K = 768 + LoRA tile products from the 128 x 128 kernel: -0.2 ms per step)
int g_stream_down = 1;          // VITLORA_GEMM_STREAM_DOWN=0: LoRA down projections stay separate launches

template <int BN, int EPI, bool DOWN = false>
void launch_s(const GemmArgs& a, hipStream_t s) {
    const size_t lds = stream_lds<BN, DOWN>();
    const int tilesM = a.M / BM;
    const int G = tilesM < g_stream_cus ? tilesM : g_stream_cus;
    hipLaunchKernelGGL((gemm_stream_kernel<BN, EPI, DOWN>), dim3(G), dim3(64 * (CW + 2)), lds, s, a);
}
int g_stream_err = 0;
template <int BN, int EPI, bool DOWN = false>
void set_attr_s() {
    const size_t lds = stream_lds<BN, DOWN>();
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_stream_kernel<BN, EPI, DOWN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) g_stream_err = (int)e;
}

}  // namespace

int gemm_stream_init() {
    g_stream_err = 0;
    set_attr_s<128, EPI_STORE_H16>();
    set_attr_s<128, EPI_GELU>();
    set_attr_s<128, EPI_GELU_BWD>();
    set_attr_s<64, EPI_STORE_H16>();
    set_attr_s<128, EPI_STORE_H16, true>();
    set_attr_s<128, EPI_GELU_BWD, true>();
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_stream_cus = cus;
    if (const char* e = getenv("VITLORA_GEMM_STREAM")) g_stream_on = e[0] != '0';
    if (const char* e = getenv("VITLORA_GEMM_STREAM_MIN_ROWS")) g_stream_min_rows = atoi(e);
    if (const char* e = getenv("VITLORA_GEMM_STREAM_MIN_ROWS64")) g_stream_min_rows64 = atoi(e);
    if (const char* e = getenv("VITLORA_GEMM_STREAM_MAX_K")) g_stream_max_k = atoi(e);
    if (const char* e = getenv("VITLORA_GEMM_STREAM_DOWN")) g_stream_down = e[0] != '0';
    return g_stream_err;
}

int gemm_stream_set_mode(int mode) {
    const int old = (g_stream_on ? 1 : 0) | (g_stream_down ? 2 : 0);
    g_stream_on = mode & 1; g_stream_down = (mode >> 1) & 1;
    return old;
}

// bn: 128 or 64 (the caller's column tile, as for gemm_nt_kernel)
// a.down_W set (down_groups = 1: 64 rows, down_ldw), W2 / ldw2 / K2 = 64 the LoRA K tile, A2 unused: can the down projection
// run inside this GEMM?
bool gemm_stream_fuses_down(const GemmArgs& a, int epi) {
    if (gemm_small_forced()) return false;
    if (!g_stream_on || !g_stream_down || a.a_gather || !a.down_W || a.down_out) return false;
    if (a.M < g_stream_min_rows || a.M % BM || a.K1 + a.K2 > g_stream_max_k || a.K1 % BK || a.K1 < BK || a.K2 != BK || a.N % 128) return false;
    return epi == EPI_STORE_H16 || epi == EPI_GELU_BWD;
}
bool gemm_stream_supports(const GemmArgs& a, int epi, int bn) {
    if (a.down_W) return gemm_stream_fuses_down(a, epi);
    if (!g_stream_on || a.a_gather) return false;
    if (bn == 64) return a.M >= g_stream_min_rows64 && a.M % BM == 0 && a.K1 % BK == 0 && a.K2 == 0 && epi == EPI_STORE_H16 && a.N % 64 == 0;
    if (a.M < g_stream_min_rows || a.M % BM || a.K1 + a.K2 > g_stream_max_k || a.K1 % BK || a.K2 % BK) return false;
    if (a.N % 128) return false;
    return epi == EPI_STORE_H16 || epi == EPI_GELU || epi == EPI_GELU_BWD;
}

void launch_gemm_stream(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    const double mv = a.Mvalid ? a.Mvalid : a.M;
    const double flops = 2.0 * mv * (a.n_algo ? a.n_algo : a.N) * (a.K1 + (a.k2_algo ? a.k2_algo : a.K2));
    char name[64];
    snprintf(name, sizeof name, "gemm_stream_kernel<%d, %d, %s>", bn == 64 ? 64 : 128, epi, a.down_W ? "true" : "false");    // as rocprofv3 prints it
    ProfScope prof_(name, flops, gemm_algo_bytes(a, epi, mv), s, 2.0 * a.M * a.N * (a.K1 + a.K2 + (a.down_W ? 64 : 0)) + (a.down_W ? 2.0 * a.M * 64.0 * a.K1 : 0.0));
    if (a.down_W) {
        if (epi == EPI_STORE_H16) launch_s<128, EPI_STORE_H16, true>(a, s);
        else launch_s<128, EPI_GELU_BWD, true>(a, s);
        return;
    }
    if (bn == 64) { launch_s<64, EPI_STORE_H16>(a, s); return; }
    switch (epi) {
        case EPI_STORE_H16: launch_s<128, EPI_STORE_H16>(a, s); break;
        case EPI_GELU: launch_s<128, EPI_GELU>(a, s); break;
        case EPI_GELU_BWD: launch_s<128, EPI_GELU_BWD>(a, s); break;
        default: break;
    }
}

}  // namespace VLNS
