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

namespace {

constexpr int BK = 64, BM = 128, STAGES = 4, CW = 4;      // CW compute waves, then 2 loader waves

template <int BN, int EPI>
__global__ __launch_bounds__(64 * (CW + 2)) void gemm_stream_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sA = (h16*)smem;                   // [STAGES][BM][BK]
    h16* sW = sA + STAGES * BM * BK;         // [STAGES][BN][BK]
    h16* sImg = sW + STAGES * BN * BK;       // [CW][32][BN / 2]: per-wave staging of the result rows (full-line stores)
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
            const int buf = s & (STAGES - 1);
            if (isA) {
                const h16* Ap = ext ? p.A2 : p.A1;
                const int lda = ext ? p.lda2 : p.lda1;
                const h16* src = Ap + (size_t)(bm * BM + lr) * lda + k0 + c8;
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
            }
        };
        constexpr int LPS_A = LA, LPS_W = LW;
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
    constexpr int MI = BM / 32;          // 16-high row tiles per wave
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MI][NJ];
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
        const h16* cA = sA + (s & (STAGES - 1)) * BM * BK;
        const h16* cW = sW + (s & (STAGES - 1)) * BN * BK;
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
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);   // D[n][m]: lane owns 4 consecutive n
        }
        if (kt == nk - 1) {
            // the fragment reads above are complete before the accumulators are final (the MFMAs consumed them), so the next
            // barrier may hand this stage back while the epilogue runs.
            // Two phases: every load (bias, saved gelu') and all arithmetic first, then nothing but stores -- with loads
            // pending between guarded stores hipcc falls back to s_waitcnt vmcnt(0) at every branch join, i.e. each store
            // waits for the previous one to be acknowledged.
            const int nb = bn * BN + wn * (BN / 2) + fg * 4;          // this lane's column of column tile j: nb + 16 j
            const int mb = bm * BM + wm * (BM / 2) + fr;              // row of row tile i: mb + 16 i
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
            char* img = (char*)(sImg + w * 32 * WCOLS);
            auto flush = [&](const h16x4 (&o)[MI][NJ], h16* Cp, int ldcp) {
#pragma unroll
                for (int hh = 0; hh < MI / 2; ++hh) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int rl = ii * 16 + fr, chunk = j * 2 + (fg >> 1);
                            *(h16x4*)(img + rl * RB + ((chunk ^ (rl & (CPR - 1))) << 4) + (fg & 1) * 8) = o[hh * 2 + ii][j];
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int rl = it * RPI + lane / CPR, chunk = lane % CPR;
                        const h16x8 v = *(const h16x8*)(img + rl * RB + ((chunk ^ (rl & (CPR - 1))) << 4));
                        const int n = bn * BN + wn * WCOLS + chunk * 8;
                        if (!p.n_store || n < p.n_store)
                            *(h16x8*)(Cp + (size_t)(bm * BM + wm * (BM / 2) + hh * 32 + rl) * ldcp + n) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is rewritten by the next half
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
int g_stream_max_k = 640;       // ... and shallow

template <int BN, int EPI>
void launch_s(const GemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(h16) + (size_t)CW * 32 * (BN / 2) * sizeof(h16);
    const int tilesM = a.M / BM;
    const int G = tilesM < g_stream_cus ? tilesM : g_stream_cus;
    hipLaunchKernelGGL((gemm_stream_kernel<BN, EPI>), dim3(G), dim3(64 * (CW + 2)), lds, s, a);
}
int g_stream_err = 0;
template <int BN, int EPI>
void set_attr_s() {
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(h16) + (size_t)CW * 32 * (BN / 2) * sizeof(h16);
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_stream_kernel<BN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) g_stream_err = (int)e;
}

}  // namespace

int gemm_stream_init() {
    g_stream_err = 0;
    set_attr_s<128, EPI_STORE_H16>();
    set_attr_s<128, EPI_GELU>();
    set_attr_s<128, EPI_GELU_BWD>();
    set_attr_s<64, EPI_STORE_H16>();
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    g_stream_cus = cus;
    if (const char* e = getenv("VITLORA_GEMM_STREAM")) g_stream_on = e[0] != '0';
    if (const char* e = getenv("VITLORA_GEMM_STREAM_MIN_ROWS")) g_stream_min_rows = atoi(e);
    if (const char* e = getenv("VITLORA_GEMM_STREAM_MAX_K")) g_stream_max_k = atoi(e);
    return g_stream_err;
}

// bn: 128 or 64 (the caller's column tile, as for gemm_nt_kernel)
bool gemm_stream_supports(const GemmArgs& a, int epi, int bn) {
    if (!g_stream_on || a.a_gather || a.down_W) return false;
    if (a.M < g_stream_min_rows || a.M % BM || a.K1 + a.K2 > g_stream_max_k || a.K1 % BK || a.K2 % BK) return false;
    if (bn == 64) return epi == EPI_STORE_H16 && a.N % 64 == 0;
    if (a.N % 128) return false;
    return epi == EPI_STORE_H16 || epi == EPI_GELU || epi == EPI_GELU_BWD;
}

void launch_gemm_stream(const GemmArgs& a, int epi, int bn, hipStream_t s) {
    const double mv = a.Mvalid ? a.Mvalid : a.M;
    const double flops = 2.0 * mv * (a.n_algo ? a.n_algo : a.N) * (a.K1 + (a.k2_algo ? a.k2_algo : a.K2));
    char name[64];
    snprintf(name, sizeof name, "gemm_stream_kernel<%d, %d>", bn == 64 ? 64 : 128, epi);
    ProfScope prof_(name, flops, gemm_algo_bytes(a, epi, mv), s);
    if (bn == 64) { launch_s<64, EPI_STORE_H16>(a, s); return; }
    switch (epi) {
        case EPI_STORE_H16: launch_s<128, EPI_STORE_H16>(a, s); break;
        case EPI_GELU: launch_s<128, EPI_GELU>(a, s); break;
        case EPI_GELU_BWD: launch_s<128, EPI_GELU_BWD>(a, s); break;
        default: break;
    }
}
