// Attention forward / backward, second generation: 32x32x16 h16 MFMA (half the LDS bytes per
// FLOP of the 16x16x32 version), K/V/Q/dO images filled by direct-to-LDS loads, 32-query /
// 32-key work items.  Same math and same interfaces as attention.hip (HF eager attention,
// modeling_vit.py:164-189, and its backward); T <= 224 tokens, head_dim 64.
//
// LDS image of one [rows][64] h16 operand: rows of 128 B, 16-byte chunk c of row r stored at
// chunk c ^ f(r), f(r) = ((r>>1)&1)<<2 | ((r>>2)&3).  With that f both access kinds used here
// are bank-conflict free:
//   * row fragments  (ds_read_b128, lane = row of a 32-row tile, all lanes of a half-wave the
//     same chunk): operands whose MFMA k index is the head dimension d;
//   * transposed fragments (ds_read_b64_tr_b16, 4 rows x 16 columns per 16-lane group):
//     operands whose k index is the token index.
// Score tiles are computed with the token index that is NOT summed next on the LANE (queries
// for S^T = K Q^T, keys for S = Q K^T): the accumulator is then directly the B operand of the
// following product (k order inside a 16-deep step: 8*(j>>2) + 4*(lane>>5) + (j&3), applied
// to both operands), and softmax statistics / LSE / delta are per-lane scalars.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

#ifdef VITLORA_ATTN_STAMPS   // diagnostic build only (tools/attn_stamp.hip): per-wave s_memtime stamps
__device__ unsigned long long g_attn_stamps[8192 * 8 * 8];
#define STAMP(k) do { if (blockIdx.x < 8192 && (threadIdx.x & 63) == 0) g_attn_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ISTAMP(hd, k) do { if ((hd) == 3) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_attn_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#define ISTAMP(hd, k) do { } while (0)
#endif
#ifdef VITLORA_ATTN_STAMPS_FWD   // per-head stamps of the per-image forward (tools/attn_fwd_stamp.hip); the backward's are off
#undef ISTAMP
#define ISTAMP(hd, k) do { } while (0)
#define FSTAMP(hd, k) do { if ((hd) == 3) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_attn_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define FSTAMP(hd, k) do { } while (0)
#endif
#ifdef VITLORA_ATTN_STAMPS2   // intra-step stamps of the ring backward (step 3 of head 3), instead of the per-head ones
#undef ISTAMP
#define ISTAMP(hd, k) do { } while (0)
#define JSTAMP(hd, i, k) do { if ((hd) == 3 && (i) == 3) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_attn_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define JSTAMP(hd, i, k) do { } while (0)
#endif

// Workgroup barrier that orders LDS only.  __syncthreads() also drains every wave's GLOBAL loads and stores
// (s_waitcnt vmcnt(0)) -- in the per-image kernels that made each head wait out its own output stores and the prefetch of
// the next head's operands twice (a quarter of the backward: tools/attn_img_stamp.hip).  Nothing in those kernels is
// exchanged between waves through global memory; the loader wave waits for its LDS-DMA itself before it arrives.
#define LDS_BARRIER()                                                        \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)

namespace {

// VL_ATTN_NT=1: q / k / v / dO / O are streamed with non-temporal loads (LDS-DMA aux bit 1 and the row-fragment loads); an
// A/B switch (tools/build_variant.sh).  Measured in round 5 and not kept: backward 2.22 -> 2.51 ms, forward 0.85 -> 1.05 ms per
// iteration (the operands were written by the GEMM before and are Infinity-Cache hits; profiles/r05_nt_loads_ab.txt)
#ifndef VL_ATTN_NT
#define VL_ATTN_NT 0
#endif
__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, VL_ATTN_NT ? 2 : 0);
}
__device__ __forceinline__ h16x8 ld_attn(const h16* p) {
#if VL_ATTN_NT
    return __builtin_nontemporal_load((const h16x8*)p);
#else
    return *(const h16x8*)p;
#endif
}



constexpr int HD = 64;
#define TPAD(t) ((double)(((t) + 31) / 32 * 32))      // executed-FLOP accounting: the kernels work on 32-token tiles
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma32(h16x8 a, h16x8 b, f32x16 c) {
#ifdef VL_BF16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// raw v_exp_f32 (no denormal-range fix-up: results below 2^-126 flush to 0, exp2(-inf) = 0)
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// row fragment of a 32-row tile: lane (r = lane&31, h = lane>>5) gets img[row0 + r][16*ks + 8*h .. +7]
__device__ __forceinline__ h16x8 row_frag32(const h16* img, int row, int ks, int h) {
    return *(const h16x8*)(img + row * HD + (((2 * ks + h) ^ swz(row)) << 3));
}
// transposed fragment: element j of lane (c = lane&31, h = lane>>5) = img[t0 + 8*(j>>2) + 4*h + (j&3)][c0 + c]
// (t0 multiple of 16, c0 multiple of 32).  16-lane group g = lane>>4: columns c0 + 16*(g&1) + [0,16), h = g>>1;
// lane i = 4q + p of the group supplies the address of row q, columns 4p..4p+3 of the 4 x 16 block.
__device__ __forceinline__ h16x8 tr_frag32(const h16* img, int t0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int h = g >> 1, q = i >> 2, p = i & 3;
    const int col = c0 + 16 * (g & 1) + 4 * p;
    const int r0 = t0 + 4 * h + q, r1 = r0 + 8;
    const h16* a0 = img + r0 * HD + (((col >> 3) ^ swz(r0)) << 3) + (col & 4);
    const h16* a1 = img + r1 * HD + (((col >> 3) ^ swz(r1)) << 3) + (col & 4);
    return cat4(lds_read_tr16(a0), lds_read_tr16(a1));
}
__device__ __forceinline__ h16x8 pack8(const f32x16& v, int s) {   // registers 8s .. 8s+7
    h16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2h(v[8 * s + j]);
    return o;
}
// Per-lane element offsets of the two fragment kinds inside ONE 32-row tile of a swizzled image, computed once
// per kernel.  swz() only looks at row bits 1..3, so they are the same for every 32-row tile and 16-row half:
// inside the (fully unrolled) tile loops a fragment address is  image + constant + one of these 8 registers,
// i.e. the ds_read's immediate offset -- no per-step address arithmetic (it was a quarter of the loop's VALU work).
struct FragOffs {
    int row[4];        // row fragment, k-step ks:  + tile * 2048
    int tr[2][2];      // transposed fragment, feature half dt, the two 4-row reads:  + tile * 2048 + st * 1024
};
__device__ __forceinline__ FragOffs frag_offs(int lane) {
    FragOffs o;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o.row[ks] = c * HD + (((2 * ks + h) ^ swz(c)) << 3);
    const int g = lane >> 4, i = lane & 15;
    const int th = g >> 1, q = i >> 2, p = i & 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 4 * th + q + 8 * e;
            o.tr[dt][e] = r * HD + (((dt * 4 + 2 * (g & 1) + (p >> 1)) ^ swz(r)) << 3) + ((p & 1) << 2);
        }
    return o;
}
__device__ __forceinline__ h16x8 row_frag_o(const h16* img, int tile, int off) {
    return *(const h16x8*)(img + tile * 32 * HD + off);
}
__device__ __forceinline__ h16x8 tr_frag_o(const h16* img, int tile, int st, const int (&off)[2]) {
    const h16* b = img + tile * 32 * HD + st * 16 * HD;
    return cat4(lds_read_tr16(b + off[0]), lds_read_tr16(b + off[1]));
}
// fill rows [0, ROWS) of a swizzled image from global rows min(r, T-1) (row stride ld elements):
// one 1 KiB direct-to-LDS load per 8 rows, groups dealt round-robin to NW waves.
template <int ROWS, int NW>
__device__ __forceinline__ void stage_glds(h16* img, const h16* src, int ld, int T, int w, int lane) {
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll 1
    for (int g = w; g < ROWS / 8; g += NW) {
        const int r = g * 8 + lr;
        const int rs = r < T ? r : T - 1;
        glds16a(src + (size_t)rs * ld + ((lc ^ swz(r)) << 3), img + g * 8 * HD);
    }
}

// Store a wave's 32 x 64 result tile (token on the lane: acc[dt][4*rq + k] = value of token c, feature
// dt*32 + 8*rq + 4*h + k) as full 128-byte rows: through a wave-private 2 KiB LDS image (16 rows per pass,
// 16-byte chunk ch of row r at ch ^ (r & 7)), then wave stores of 8 rows x 128 B each.  Row-per-lane 8-byte
// stores touch 32 lines per instruction and run at a third of the rate (tools/store_bench.hip).  Measured:
// forward -10 %; the backward (148 KB of LDS with 4 KiB images) did not gain and keeps its direct stores.
__device__ __forceinline__ void store_rows32_half(char* img, const f32x16 (&acc)[2], float mul, h16* dst0, int ld, int tok0,
                                                  int T, int lane) {
    const int c = lane & 31, h = lane >> 5;
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if ((c >> 4) == p) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    h16x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = f2h(acc[dt][4 * rq + k] * mul);
                    *(h16x4*)(img + (c & 15) * 128 + (((dt * 4 + rq) ^ (c & 7)) << 4) + 8 * h) = o;
                }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 8 + lr;
            const h16x8 v = *(const h16x8*)(img + row * 128 + ((lc ^ lr) << 4));
            const int tok = tok0 + p * 16 + row;
            if (tok < T) *(h16x8*)(dst0 + (size_t)tok * ld + lc * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward: one workgroup (8 waves) per (image, head); each wave owns 32-query blocks.
// NT = number of 32-key tiles (ROWS = 32*NT >= T).
// ------------------------------------------------------------------------------------------
constexpr int FWD_WAVES = 8;
template <int NT>
__global__ __launch_bounds__(64 * FWD_WAVES) void attn_fwd32_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx,
                                                            float* __restrict__ lse2, int T, int H, int D, float scale_log2e) {
    constexpr int ROWS = NT * 32;
    __shared__ __attribute__((aligned(16))) h16 sm[2 * ROWS * HD];
    __shared__ __attribute__((aligned(16))) char wimg[FWD_WAVES * 2048];     // per-wave output images (store_rows32_half)
    h16* sK = sm;
    h16* sV = sm + ROWS * HD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld + hd * HD;
    stage_glds<ROWS, FWD_WAVES>(sK, base + D, ld, T, w, lane);
    stage_glds<ROWS, FWD_WAVES>(sV, base + 2 * D, ld, T, w, lane);

    const int nqb = (T + 31) >> 5;
    int qb = w;
    h16x8 qf[4];
    auto load_q = [&](int blk) {
        const int q = blk * 32 + c;
        const h16* qr = base + (size_t)(q < T ? q : T - 1) * ld + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const h16x8*)(qr + 16 * ks);
    };
    if (qb < nqb) load_q(qb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const FragOffs fo = frag_offs(lane);
    const int tcut = T - 4 * h;                     // key < T  <=>  kt*32 + (r&3) + 8*(r>>2) < tcut
    for (; qb < nqb; qb += FWD_WAVES) {
        // online softmax over 32-key tiles (running max m, running sum l per query = per lane pair)
        float m = -INFINITY, l = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        // one 32-key tile; MASKED: the last, partly filled one (keys >= T get -inf)
        auto step = [&](int kt, auto masked) {
            constexpr bool MASKED = decltype(masked)::value;
            // S^T tile: rows = keys, column = query c
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = mfma32(row_frag_o(sK, kt, fo.row[ks]), qf[ks], s);
            // m, tmax are in RAW score units (the softmax scale is positive); p = exp2(s*c - m*c)
            float tmax = -INFINITY;
            if constexpr (MASKED) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((kt * 32 + (r & 3) + 8 * (r >> 2)) >= tcut) s[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m)) {                  // some row's running max grew: rescale (first tile: always)
                const float mn = fmaxf(m, tmax);    // finite: key 0 of tile 0 is always valid
                const float alpha = fexp2((m - mn) * scale_log2e);   // first tile: exp2(-inf) = 0, l = 0, o = 0
                m = mn;
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            const float mc = -m * scale_log2e;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fexp2(fmaf(s[r], scale_log2e, mc)); ps += s[r]; }
            l += ps;
            // O^T += V^T P^T: rows = d, column = query c
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const h16x8 pb = pack8(s, st);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(tr_frag_o(sV, kt, st, fo.tr[dt]), pb, o[dt]);
            }
        };
        const int nfull = T >> 5;
#pragma unroll 1
        for (int kt = 0; kt < nfull; ++kt) step(kt, std::false_type{});
        if (T & 31) step(nfull, std::true_type{});
        l += __shfl_xor(l, 32, 64);
        const int q = qb * 32 + c;
        const float inv = 1.f / l;
        store_rows32_half(wimg + w * 2048, o, inv, ctx + (size_t)b * T * D + hd * HD, D, qb * 32, T, lane);
        if (q < T && h == 0) lse2[((size_t)b * H + hd) * T + q] = m * scale_log2e + log2f(l);
        if (qb + FWD_WAVES < nqb) load_q(qb + FWD_WAVES);
    }
}

// ------------------------------------------------------------------------------------------
// backward: one workgroup (8 waves) per (image, head).  Work items: NB key blocks (phase B:
// dK, dV; key on the lane) and NB query blocks (phase A: dQ; query on the lane), 32 tokens each.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void attn_bwd32_kernel(const h16* __restrict__ qkv, const h16* __restrict__ ctx,
                                                         const h16* __restrict__ dctx, const float* __restrict__ lse2,
                                                         h16* __restrict__ dqkv, int T, int H, int D, float scale,
                                                         float scale_log2e) {
    constexpr int ROWS = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // image order: the pairs read together (Q with dO in phase B, K with V in phase A) sit one image (28 KiB) apart,
    // inside the 64 KiB immediate-offset range of one address register
    h16* sQ = (h16*)smem;
    h16* sdO = sQ + ROWS * HD;
    h16* sK = sdO + ROWS * HD;
    h16* sV = sK + ROWS * HD;
    float* sLse = (float*)(sV + ROWS * HD);
    float* sDelta = sLse + ROWS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld + hd * HD;
    const h16* dobase = dctx + (size_t)b * T * D + hd * HD;
    const h16* obase = ctx + (size_t)b * T * D + hd * HD;
    STAMP(0);
    stage_glds<ROWS, 8>(sQ, base, ld, T, w, lane);
    stage_glds<ROWS, 8>(sK, base + D, ld, T, w, lane);
    stage_glds<ROWS, 8>(sV, base + 2 * D, ld, T, w, lane);
    stage_glds<ROWS, 8>(sdO, dobase, D, T, w, lane);
    // delta[r] = sum_d dO[r][d] * O[r][d]; LSE (rows >= T: +inf so that P = 0 there).  All of a thread's global
    // loads go out before the first one is consumed: one memory round trip for the whole prologue, not one per
    // pass (the passes used to serialise behind the in-order vmcnt: 4 round trips, a third of the kernel's time;
    // measured with tools/attn_stamp.hip).
    constexpr int NPASS = (ROWS * 8 + 511) / 512;
    h16x8 pdv[NPASS], pov[NPASS];
    float plse[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int idx = tid + j * 512, r = idx >> 3, cc = idx & 7;
        const int rc = r < T ? r : T - 1;                  // clamped: every load is valid, masked below
        pdv[j] = ld_attn(dobase + (size_t)rc * D + cc * 8);
        pov[j] = ld_attn(obase + (size_t)rc * D + cc * 8);
        plse[j] = lse2[((size_t)b * H + hd) * T + rc];
    }
    STAMP(7);
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int idx = tid + j * 512, r = idx >> 3, cc = idx & 7;
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) part += h2f(pdv[j][k]) * h2f(pov[j][k]);
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (cc == 0 && r < ROWS) {
            sDelta[r] = r < T ? part : 0.f;
            sLse[r] = r < T ? plse[j] : INFINITY;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    STAMP(1);
    const int nb = (T + 31) >> 5;
    const FragOffs fo = frag_offs(lane);
    const int tcut = T - 4 * h;                      // key index held in register r of a tile: kt*32 + (r&3) + 8*(r>>2) + 4*h
    for (int item = w; item < 2 * nb; item += 8) {
        if (item < nb) {
            // ---------------- phase B: key block on the lane ----------------
            const int key = item * 32 + c;
            h16x8 kf[4], vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kf[ks] = row_frag32(sK, key, ks, h); vf[ks] = row_frag32(sV, key, ks, h); }
            f32x16 dv[2], dk[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
#pragma unroll 1
            for (int qt = 0; qt < nb; ++qt) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(sQ, qt, fo.row[ks]), kf[ks], s);                     // S[q][key]
                    dp = mfma32(row_frag_o(sQ + ROWS * HD, qt, fo.row[ks]), vf[ks], dp);       // dP[q][key]  (sdO = sQ + one image)
                }
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 lq = *(const f32x4*)(sLse + qt * 32 + 8 * rq + 4 * h);
                    const f32x4 dl = *(const f32x4*)(sDelta + qt * 32 + 8 * rq + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float p = fexp2(fmaf(s[4 * rq + k], scale_log2e, -lq[k]));   // rows >= T: 0
                        s[4 * rq + k] = p;
                        dp[4 * rq + k] = p * (dp[4 * rq + k] - dl[k]);     // dS / scale (folded into dK below)
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const h16x8 pb = pack8(s, st), dsb = pack8(dp, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = mfma32(tr_frag_o(sQ + ROWS * HD, qt, st, fo.tr[dt]), pb, dv[dt]);   // dV^T[d][key]
                        dk[dt] = mfma32(tr_frag_o(sQ, qt, st, fo.tr[dt]), dsb, dk[dt]);   // dK^T[d][key]
                    }
                }
            }
            STAMP(2);
            if (key < T) {
                h16* dst = dqkv + ((size_t)b * T + key) * ld + hd * HD + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        h16x4 kv, vv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { kv[k] = f2h(dk[dt][4 * rq + k] * scale); vv[k] = f2h(dv[dt][4 * rq + k]); }
                        *(h16x4*)(dst + D + dt * 32 + 8 * rq) = kv;
                        *(h16x4*)(dst + 2 * D + dt * 32 + 8 * rq) = vv;
                    }
            }
            STAMP(3);
        } else {
            // ---------------- phase A: query block on the lane ----------------
            STAMP(4);
            const int q = (item - nb) * 32 + c;
            h16x8 qf[4], dof[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag32(sQ, q, ks, h); dof[ks] = row_frag32(sdO, q, ks, h); }
            const float lse_q = sLse[q], delta_q = sDelta[q];
            f32x16 dq[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
            auto stepA = [&](int kt, auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(sK, kt, fo.row[ks]), qf[ks], s);                     // S^T[key][q]
                    dp = mfma32(row_frag_o(sK + ROWS * HD, kt, fo.row[ks]), dof[ks], dp);      // dP^T[key][q]  (sV = sK + one image)
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fexp2(fmaf(s[r], scale_log2e, -lse_q));
                    if constexpr (MASKED) { if (kt * 32 + (r & 3) + 8 * (r >> 2) >= tcut) p = 0.f; }   // keys >= T
                    dp[r] = p * (dp[r] - delta_q);                                            // dS / scale (folded into dQ below)
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const h16x8 dsb = pack8(dp, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(tr_frag_o(sK, kt, st, fo.tr[dt]), dsb, dq[dt]);
                }
            };
            const int nfull = T >> 5;
#pragma unroll 1
            for (int kt = 0; kt < nfull; ++kt) stepA(kt, std::false_type{});
            if (T & 31) stepA(nfull, std::true_type{});       // the one partly filled key tile
            STAMP(5);
            if (q < T) {
                h16* dst = dqkv + ((size_t)b * T + q) * ld + hd * HD + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        h16x4 ov;
#pragma unroll
                        for (int k = 0; k < 4; ++k) ov[k] = f2h(dq[dt][4 * rq + k] * scale);
                        *(h16x4*)(dst + dt * 32 + 8 * rq) = ov;
                    }
            }
        }
    }
    STAMP(6);
}


// ==========================================================================================
// Per-IMAGE persistent kernels (large batches: one workgroup per image walks its H heads).
//   * waves 0 .. NT-1 each own one 32-token block of the image for every head; the last wave is the
//     LOADER: it streams the next operands into an LDS ring by LDS-DMA while the others compute (the
//     per-(image, head) kernels above spent half of a workgroup's life in an un-overlapped prologue),
//     and prepares the per-row constants (backward: LSE and delta = rowsum(dO * O)).
//   * compute waves issue no LDS-DMA: their own-block fragments come straight from global memory
//     (requested one phase ahead), so the compiler's own s_waitcnt bookkeeping stays valid for them;
//     only the loader wave has DMA in flight and it drains it (vmcnt(0)) before every barrier.
//   * because all heads of an image meet in one workgroup, the LoRA DOWN projections that read the
//     attention outputs are summed over heads in registers -- t = ctx Ad^T (forward, attention.output.dense)
//     and u = dqkv Bd^T (backward, query / key / value) -- and the separate skinny GEMM passes over ctx /
//     dqkv disappear.  The accumulator tile of the previous product is the B operand ("token on the lane").
// ==========================================================================================
struct LoraDown {            // rank-r down projection fused into the attention kernels (nullptr W = off)
    const h16* W;            // forward: Ad [kext, D]; backward: Bd [kext, 3D]   (row j = LoRA column j)
    h16* out;                // t / u  [B*T (padded), 64]
    int r;                   // rank (<= 8); backward: q, k, v rows at 0, r, 2r
    unsigned mods;           // backward: bit 0 / 1 / 2 = query / key / value adapted
};

// A operand of the down product Y[j][token] = sum_d W[j][d] X[d][token] whose B operand is an accumulator tile X
// (pack8 of registers 8*st .. 8*st+7): element jj of lane half h must carry k = 16*st + 8*(jj>>2) + 4*h + (jj&3)
// (the accumulator's row order, see tr_frag32).  The head's slice of W ([8 rows j][64 features], rows >= r zero) was
// staged in LDS by the loader wave; lanes j >= 8 supply zeros.  dt = feature half of the tile.
__device__ __forceinline__ h16x8 down_frag_lds(const h16* sW, int dt, int st, int lane) {
    const int j = lane & 31, h = lane >> 5;
    h16x4 lo4, hi4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo4[k] = (h16)0.f; hi4[k] = (h16)0.f; }
    if (j < 8) {
        const h16* p = sW + j * HD + dt * 32 + 16 * st + 4 * h;
        lo4 = *(const h16x4*)p;
        hi4 = *(const h16x4*)(p + 8);
    }
    return cat4(lo4, hi4);
}
// loader wave: rows row0 .. row0 + r - 1 of W (row stride ldw), 64 features from col0 -> sW [8][64], zero rows >= r
__device__ __forceinline__ void stage_down(h16* sW, const h16* W, int ldw, int row0, int r, int col0, int lane) {
    const int j = lane >> 3, c8 = (lane & 7) * 8;
    h16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (h16)0.f;
    if (j < r) v = *(const h16x8*)(W + (size_t)(row0 + j) * ldw + col0 + c8);
    *(h16x8*)(sW + j * HD + c8) = v;
}
// Y (registers 0..3 = rows j = 4h .. 4h+3 of token c) added to the per-token fp32 sums kept in LDS
__device__ __forceinline__ void down_accumulate(float* acc, const f32x16& y, int tok, int h) {
    f32x4* p = (f32x4*)(acc + tok * 8 + 4 * h);
    f32x4 v = *p;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += y[k];
    *p = v;
}

constexpr int IMG_WAVES = 8;

template <int NT>
size_t fwd_img_lds() {
    return (size_t)4 * NT * 32 * HD * sizeof(h16) + (size_t)(IMG_WAVES - 1) * 2048 + (size_t)NT * 32 * 8 * sizeof(float) +
           (size_t)2 * 8 * HD * sizeof(h16);
}

template <int NT>
__global__ __launch_bounds__(64 * IMG_WAVES) void attn_fwd_img_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx,
                                                                      float* __restrict__ lse2, int T, int H, int D,
                                                                      float scale_log2e, const LoraDown lo) {
    constexpr int ROWS = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* ring = (h16*)smem;                                   // 4 images: {K, V} of even heads, {K, V} of odd heads
    char* wimg = smem + (size_t)4 * ROWS * HD * sizeof(h16);  // per-wave output staging (store_rows32_half)
    float* tsum = (float*)(wimg + (IMG_WAVES - 1) * 2048);    // [ROWS][8] fp32: t summed over heads
    h16* sAd = (h16*)(tsum + ROWS * 8);                       // [2][8][64]: the head's slice of Ad (head parity)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    const bool loader = w == IMG_WAVES - 1;
    const bool active = w < ((T + 31) >> 5);
    const FragOffs fo = frag_offs(lane);
    const int tcut = T - 4 * h;
    const int q = w * 32 + c, qc = q < T ? q : T - 1;

    for (int i = tid; i < ROWS * 8; i += 64 * IMG_WAVES) tsum[i] = 0.f;
    if (loader) {
        stage_glds<ROWS, 1>(ring, base + D, ld, T, 0, lane);
        stage_glds<ROWS, 1>(ring + ROWS * HD, base + 2 * D, ld, T, 0, lane);
        if (lo.W) {                                            // zero the image's t rows once (columns >= r stay zero)
            h16x8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (h16)0.f;
            for (int i = lane; i < T * 8; i += 64) *(h16x8*)(lo.out + ((size_t)b * T + (i >> 3)) * 64 + (i & 7) * 8) = z;
            stage_down(sAd, lo.W, D, 0, lo.r, 0, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    h16x8 qf[4];
    if (!loader && active) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_attn(base + (size_t)qc * ld + 8 * h + 16 * ks);
    }
    __syncthreads();

    for (int hd = 0; hd < H; ++hd) {
        const h16* sK = ring + (hd & 1) * 2 * ROWS * HD;
        const h16* sV = sK + ROWS * HD;
        FSTAMP(hd, 0);
        if (loader) {
            if (hd + 1 < H) {
                h16* nK = ring + ((hd + 1) & 1) * 2 * ROWS * HD;
                stage_glds<ROWS, 1>(nK, base + D + (hd + 1) * HD, ld, T, 0, lane);
                stage_glds<ROWS, 1>(nK + ROWS * HD, base + 2 * D + (hd + 1) * HD, ld, T, 0, lane);
                if (lo.W) stage_down(sAd + ((hd + 1) & 1) * 8 * HD, lo.W, D, 0, lo.r, (hd + 1) * HD, lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FSTAMP(hd, 5);
        } else if (active) {
            float m = -INFINITY, l = 0.f;
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            // software pipeline over the key tiles: the S MFMAs of tile kt+1 are issued before the softmax VALU of tile kt
            // (independent registers), so one wave keeps both the matrix pipe and the VALU busy
            auto scores = [&](int kt) {
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mfma32(row_frag_o(sK, kt, fo.row[ks]), qf[ks], s);
                return s;
            };
            auto step = [&](int kt, f32x16 s, auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
                float tmax = -INFINITY;
                if constexpr (MASKED) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((kt * 32 + (r & 3) + 8 * (r >> 2)) >= tcut) s[r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                if (__any(tmax > m)) {
                    const float mn = fmaxf(m, tmax);
                    const float alpha = fexp2((m - mn) * scale_log2e);
                    m = mn;
                    l *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
                const float mc = -m * scale_log2e;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = fexp2(fmaf(s[r], scale_log2e, mc)); ps += s[r]; }
                l += ps;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const h16x8 pb = pack8(s, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(tr_frag_o(sV, kt, st, fo.tr[dt]), pb, o[dt]);
                }
            };
            const int nfull = T >> 5;
            const int ntile = nfull + ((T & 31) ? 1 : 0);
            if (ntile == NT) {
                // every key tile in use (T = 197: the path's case): whole score rows in registers (16 NT values per lane),
                // ONE row maximum, no running rescale, and a branch-free body the scheduler can interleave -- the exp2 of
                // tile kt+1 issues under the P V MFMAs of tile kt
                f32x16 S[NT];
                {
                    // K row fragments one tile ahead of the MFMAs that use them (the compiler otherwise reloads one
                    // register quad per MFMA and waits out the LDS latency 4 NT times)
                    h16x8 kf[2][4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) kf[0][ks] = row_frag_o(sK, 0, fo.row[ks]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        if (kt + 1 < NT) {
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) kf[(kt + 1) & 1][ks] = row_frag_o(sK, kt + 1, fo.row[ks]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) S[kt] = mfma32(kf[kt & 1][ks], qf[ks], S[kt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                FSTAMP(hd, 1);
                if (T & 31) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (((NT - 1) * 32 + (r & 3) + 8 * (r >> 2)) >= tcut) S[NT - 1][r] = -INFINITY;
                }
                // T = 197 leaves 5 keys in the last tile: only its registers 0..3 (keys 0..3 / 4..7 of the tile) can be live, the
                // other 12 are padding in every lane -- no maximum, no exp2, and no P V MFMA for the tile's upper 16 keys
                const bool short_tail = (T & 31) != 0 && (T & 31) <= 8;
                float tmax = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT - 1; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, S[kt][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, S[NT - 1][r]);
                if (!short_tail) {
#pragma unroll
                    for (int r = 4; r < 16; ++r) tmax = fmaxf(tmax, S[NT - 1][r]);
                }
                m = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mc = -m * scale_log2e;
                FSTAMP(hd, 2);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    if (kt == NT - 1 && short_tail) break;
                    // this tile's V fragments are requested before its exp2 block and consumed after it
                    h16x8 vfr[2][2];
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) vfr[st][dt] = tr_frag_o(sV, kt, st, fo.tr[dt]);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x2 ps2 = {0.f, 0.f};                 // packed fp32 fma / add: two elements per VALU instruction
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 t = f32x2{S[kt][r], S[kt][r + 1]} * f32x2{scale_log2e, scale_log2e} + f32x2{mc, mc};
                        const f32x2 p = {fexp2(t[0]), fexp2(t[1])};
                        S[kt][r] = p[0]; S[kt][r + 1] = p[1];
                        ps2 += p;
                    }
                    l += ps2[0] + ps2[1];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const h16x8 pb = pack8(S[kt], st);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vfr[st][dt], pb, o[dt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (short_tail) {
                    const h16x8 vt0 = tr_frag_o(sV, NT - 1, 0, fo.tr[0]), vt1 = tr_frag_o(sV, NT - 1, 0, fo.tr[1]);
                    f32x16& Sl = S[NT - 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { Sl[r] = fexp2(fmaf(Sl[r], scale_log2e, mc)); l += Sl[r]; }
#pragma unroll
                    for (int r = 4; r < 16; ++r) Sl[r] = 0.f;
                    const h16x8 pb = pack8(Sl, 0);
                    o[0] = mfma32(vt0, pb, o[0]);
                    o[1] = mfma32(vt1, pb, o[1]);
                }
            } else {
                f32x16 s_cur = scores(0);
#pragma unroll 1
                for (int kt = 0; kt < nfull; ++kt) {
                    f32x16 s_nxt = s_cur;
                    if (kt + 1 < ntile) s_nxt = scores(kt + 1);
                    step(kt, s_cur, std::false_type{});
                    s_cur = s_nxt;
                }
                if (T & 31) step(nfull, s_cur, std::true_type{});
            }
            FSTAMP(hd, 3);
            if (hd + 1 < H) {                                  // next head's query fragments fly under the epilogue
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_attn(base + (size_t)qc * ld + (hd + 1) * HD + 8 * h + 16 * ks);
            }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.f / l;
            if (lo.W) {
                // t^T[j][token] += Ad_h[j][d] * ctx^T[d][token]: the normalised output tile is the B operand
                f32x16 y;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = 0.f;
                const h16* sA = sAd + (hd & 1) * 8 * HD;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= inv;
#pragma unroll
                    for (int st = 0; st < 2; ++st) y = mfma32(down_frag_lds(sA, dt, st, lane), pack8(o[dt], st), y);
                }
                down_accumulate(tsum, y, q, h);
                store_rows32_half(wimg + w * 2048, o, 1.f, ctx + (size_t)b * T * D + hd * HD, D, w * 32, T, lane);
            } else {
                store_rows32_half(wimg + w * 2048, o, inv, ctx + (size_t)b * T * D + hd * HD, D, w * 32, T, lane);
            }
            if (q < T && h == 0) lse2[((size_t)b * H + hd) * T + q] = m * scale_log2e + log2f(l);
            FSTAMP(hd, 4);
        }
        FSTAMP(hd, 6);
        LDS_BARRIER();
        FSTAMP(hd, 7);
    }
    if (lo.W && !loader && active && q < T && 4 * h < lo.r) {
        const f32x4 t4 = *(const f32x4*)(tsum + q * 8 + 4 * h);      // LoRA columns 4h .. 4h+3 of this token
        h16x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = f2h(t4[k]);
        *(h16x4*)(lo.out + ((size_t)b * T + q) * 64 + 4 * h) = v;
    }
}

template <int NT>
size_t bwd_img_lds() {
    return (size_t)4 * NT * 32 * HD * sizeof(h16) + (size_t)2 * NT * 32 * sizeof(float) + (size_t)3 * NT * 32 * 8 * sizeof(float) +
           (size_t)2 * 3 * 8 * HD * sizeof(h16) + (size_t)(IMG_WAVES - 1) * 2048;
}

template <int NT>
__global__ __launch_bounds__(64 * IMG_WAVES) void attn_bwd_img_kernel(const h16* __restrict__ qkv, const h16* __restrict__ ctx,
                                                                      const h16* __restrict__ dctx, const float* __restrict__ lse2,
                                                                      h16* __restrict__ dqkv, int T, int H, int D, float scale,
                                                                      float scale_log2e, const LoraDown lo) {
    constexpr int ROWS = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sK = (h16*)smem;                    // slots 0, 1: K, V of the current head (phase A operands)
    h16* sV = sK + ROWS * HD;
    h16* sQ = sV + ROWS * HD;                // slots 2, 3: Q, dO of the current head (phase B operands)
    h16* sdO = sQ + ROWS * HD;
    float* sLse = (float*)(sdO + ROWS * HD); // [ROWS]  written by the phase A waves, read by phase B
    float* sDelta = sLse + ROWS;             // [ROWS]
    float* usum = sDelta + ROWS;             // [3][ROWS][8] fp32: u of query / key / value summed over heads
    h16* sBd = (h16*)(usum + 3 * ROWS * 8);  // [2][3][8][64]: the head's slices of Bd (head parity; q, k, v)
    char* wimg = (char*)(sBd + 2 * 3 * 8 * HD);   // per-wave 2 KiB images: results leave as full 128-byte rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    const h16* dobase = dctx + (size_t)b * T * D;
    const h16* obase = ctx + (size_t)b * T * D;
    const bool loader = w == IMG_WAVES - 1;
    const int nb = (T + 31) >> 5;
    const bool active = w < nb;
    const FragOffs fo = frag_offs(lane);
    const int tcut = T - 4 * h;
    const int tok = w * 32 + c, tokc = tok < T ? tok : T - 1;      // this lane's query (phase A) / key (phase B)

    auto stage_bd = [&](int hd) {            // loader: this head's [8][64] slices of Bd for query / key / value
        h16* dst = sBd + (hd & 1) * 3 * 8 * HD;
#pragma unroll
        for (int md = 0; md < 3; ++md) stage_down(dst + md * 8 * HD, lo.W, ld, md * lo.r, (lo.mods >> md) & 1u ? lo.r : 0, md * D + hd * HD, lane);
    };
    for (int i = tid; i < 3 * ROWS * 8; i += 64 * IMG_WAVES) usum[i] = 0.f;
    if (loader) {
        stage_glds<ROWS, 1>(sK, base + D, ld, T, 0, lane);
        stage_glds<ROWS, 1>(sV, base + 2 * D, ld, T, 0, lane);
        if (lo.W) {
            h16x8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (h16)0.f;
            for (int i = lane; i < T * 8; i += 64) *(h16x8*)(lo.out + ((size_t)b * T + (i >> 3)) * 64 + (i & 7) * 8) = z;
            stage_bd(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    h16x8 fa[4], fb[4], fo_[4];              // own-block fragments: phase A (q, dO rows; O rows for delta), phase B (k, v rows)
    float lse_own = 0.f;
    auto fetch_a = [&](int hd) {             // phase A operands of head hd for this lane's query
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa[ks] = ld_attn(base + (size_t)tokc * ld + hd * HD + 8 * h + 16 * ks);
            fb[ks] = ld_attn(dobase + (size_t)tokc * D + hd * HD + 8 * h + 16 * ks);
            fo_[ks] = ld_attn(obase + (size_t)tokc * D + hd * HD + 8 * h + 16 * ks);
        }
        lse_own = lse2[((size_t)b * H + hd) * T + tokc];
    };
    if (!loader && active) fetch_a(0);
    if (w >= 4 && !loader) __builtin_amdgcn_s_setprio(1);     // the second-dispatched half loses VALU arbitration otherwise
    __syncthreads();

    for (int hd = 0; hd < H; ++hd) {
        const float* L = sLse;
        const float* Dl = sDelta;
        ISTAMP(hd, 0);
        // ------------------------------ phase A: dQ (query on the lane; K, V images) ------------------------------
        if (loader) {
            stage_glds<ROWS, 1>(sQ, base + hd * HD, ld, T, 0, lane);
            stage_glds<ROWS, 1>(sdO, dobase + hd * HD, D, T, 0, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (active) {
            // delta = rowsum(dO * O) of the own query from its fragments (each lane half holds 32 of the 64 features);
            // LSE (rows >= T: +inf -> P = 0) and delta go to LDS for phase B, where every query's constants are needed
            float delta_q = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int k = 0; k < 8; ++k) delta_q = fmaf(h2f(fb[ks][k]), h2f(fo_[ks][k]), delta_q);
            delta_q += __shfl_xor(delta_q, 32, 64);
            const float lse_q = tok < T ? lse_own : INFINITY;
            if (tok >= T) delta_q = 0.f;
            if (h == 0) { sLse[tok] = lse_q; sDelta[tok] = delta_q; }
            f32x16 dq[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
            auto stepA = [&](int kt, auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(sK, kt, fo.row[ks]), fa[ks], s);                     // S^T[key][q]
                    dp = mfma32(row_frag_o(sV, kt, fo.row[ks]), fb[ks], dp);                   // dP^T[key][q]
                }
                // two elements per VALU instruction where the ISA has a packed fp32 form (fma, sub, mul); exp2 stays scalar
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = f32x2{s[r], s[r + 1]} * f32x2{scale_log2e, scale_log2e} - f32x2{lse_q, lse_q};
                    f32x2 p = {fexp2(t[0]), fexp2(t[1])};
                    if constexpr (MASKED) {
                        if (kt * 32 + (r & 3) + 8 * (r >> 2) >= tcut) p[0] = 0.f;
                        if (kt * 32 + ((r + 1) & 3) + 8 * ((r + 1) >> 2) >= tcut) p[1] = 0.f;
                    }
                    const f32x2 d = p * (f32x2{dp[r], dp[r + 1]} - f32x2{delta_q, delta_q});
                    dp[r] = d[0]; dp[r + 1] = d[1];
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const h16x8 dsb = pack8(dp, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(tr_frag_o(sK, kt, st, fo.tr[dt]), dsb, dq[dt]);
                }
            };
            const int nfull = T >> 5;
#pragma unroll 1
            for (int kt = 0; kt < nfull; ++kt) stepA(kt, std::false_type{});
            if (T & 31) stepA(nfull, std::true_type{});
            ISTAMP(hd, 1);
            // phase B fragments of this head: the k, v rows of the own block are still in the K / V images (the loader
            // overwrites them only after the barrier below)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { fa[ks] = row_frag32(sK, tok, ks, h); fb[ks] = row_frag32(sV, tok, ks, h); }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[dt][r] *= scale;
            if (lo.W && (lo.mods & 1u)) {
                f32x16 y;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = 0.f;
                const h16* sB = sBd + (hd & 1) * 3 * 8 * HD;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int st = 0; st < 2; ++st) y = mfma32(down_frag_lds(sB, dt, st, lane), pack8(dq[dt], st), y);
                down_accumulate(usum, y, tok, h);
            }
            // row-per-lane 8-byte stores touch 32 lines per instruction and were a quarter of the head's time here
            // (tools/attn_img_stamp.hip): the tile leaves through the wave's LDS image as full 128-byte rows
            store_rows32_half(wimg + w * 2048, dq, 1.f, dqkv + (size_t)b * T * ld + hd * HD, ld, w * 32, T, lane);
        }
        ISTAMP(hd, 2);
        LDS_BARRIER();
        ISTAMP(hd, 3);
        // ------------------------------ phase B: dK, dV (key on the lane; Q, dO images) ------------------------------
        if (loader) {
            if (hd + 1 < H) {
                stage_glds<ROWS, 1>(sK, base + D + (hd + 1) * HD, ld, T, 0, lane);
                stage_glds<ROWS, 1>(sV, base + 2 * D + (hd + 1) * HD, ld, T, 0, lane);
                if (lo.W) stage_bd(hd + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (active) {
            f32x16 dv[2], dk[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
#pragma unroll 1
            for (int qt = 0; qt < nb; ++qt) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(sQ, qt, fo.row[ks]), fa[ks], s);                     // S[q][key]
                    dp = mfma32(row_frag_o(sdO, qt, fo.row[ks]), fb[ks], dp);                  // dP[q][key]
                }
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 lq = *(const f32x4*)(L + qt * 32 + 8 * rq + 4 * h);
                    const f32x4 dl = *(const f32x4*)(Dl + qt * 32 + 8 * rq + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float p = fexp2(fmaf(s[4 * rq + k], scale_log2e, -lq[k]));   // rows >= T: 0
                        s[4 * rq + k] = p;
                        dp[4 * rq + k] = p * (dp[4 * rq + k] - dl[k]);
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const h16x8 pb = pack8(s, st), dsb = pack8(dp, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = mfma32(tr_frag_o(sdO, qt, st, fo.tr[dt]), pb, dv[dt]);        // dV^T[d][key]
                        dk[dt] = mfma32(tr_frag_o(sQ, qt, st, fo.tr[dt]), dsb, dk[dt]);        // dK^T[d][key]
                    }
                }
            }
            ISTAMP(hd, 4);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[dt][r] *= scale;
            if (lo.W) {
                const h16* sB = sBd + (hd & 1) * 3 * 8 * HD;
#pragma unroll
                for (int md = 1; md < 3; ++md) {
                    if (!((lo.mods >> md) & 1u)) continue;
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int st = 0; st < 2; ++st)
                            y = mfma32(down_frag_lds(sB + md * 8 * HD, dt, st, lane), pack8(md == 1 ? dk[dt] : dv[dt], st), y);
                    down_accumulate(usum + md * ROWS * 8, y, tok, h);
                }
            }
            store_rows32_half(wimg + w * 2048, dk, 1.f, dqkv + (size_t)b * T * ld + D + hd * HD, ld, w * 32, T, lane);
            store_rows32_half(wimg + w * 2048, dv, 1.f, dqkv + (size_t)b * T * ld + 2 * D + hd * HD, ld, w * 32, T, lane);
            ISTAMP(hd, 7);
            // next head's phase A operands (q, dO, O rows of the own block, LSE): the registers are free now and the
            // latency overlaps the wait at the barrier
            if (hd + 1 < H) fetch_a(hd + 1);
        }
        ISTAMP(hd, 5);
        LDS_BARRIER();
        ISTAMP(hd, 6);
    }
    if (lo.W && !loader && active && tok < T && 4 * h < lo.r) {
        h16* dst = lo.out + ((size_t)b * T + tok) * 64 + 4 * h;
#pragma unroll
        for (int md = 0; md < 3; ++md) {
            if (!((lo.mods >> md) & 1u)) continue;
            const f32x4 u4 = *(const f32x4*)(usum + (md * ROWS + tok) * 8 + 4 * h);
            h16x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = f2h(u4[k]);
            *(h16x4*)(dst + md * lo.r) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward, single pass ("ring"): round 3.  attn_bwd_img_kernel above recomputes S, P, dP and dS twice per head (phase A with
// the query on the lane for dQ, phase B with the key on the lane for dK / dV): 28 MFMAs and two exp blocks per 32 x 32 tile
// pair.  Here each compute wave owns key block w AND query block w.  Step i of a head: the wave forms the tile
// (queries (w+i) mod nb, keys w) ONCE with the key on the lane, accumulates its dK / dV, and hands the 32 x 32 dS tile
// (h16, 2 KiB) to the owner of that query block through LDS; after the step's barrier it takes the tile another wave made for
// ITS queries (keys (w-i) mod nb) and accumulates dQ += dS K: 20 MFMAs and one exp block per tile pair.  In one step every wave
// writes a different slot (the ring), so the exchange needs no atomics, and two slot sets alternate so that one barrier per
// step orders both the hand-over and the reuse.
//   LDS: Q, dO of the head, K of this head and of the next (images of IR rows: T <= 200; reads past an image's end fall in
//   the next image, or in the exchange slots behind the last one, which only ever hold finite h16 -- the products they
//   enter are masked by P = 0 (queries >= T, LSE = +inf) or by zeroed dS (keys >= T)); exchange slots 2 x NT x 2 KiB;
//   LSE / delta of the head and of the next; u sums; Bd slices.
//   Loader wave: K of the next head during the steps, Q / dO of the next head under the compute waves' epilogue.
// ------------------------------------------------------------------------------------------
#ifndef RING_PRIO
#define RING_PRIO 0
#endif
template <int NT> constexpr int ring_img_rows() { return NT == 7 ? 200 : NT * 32; }
template <int NT>
size_t bwd_ring_lds() {
    return (size_t)4 * ring_img_rows<NT>() * HD * sizeof(h16) + (size_t)2 * NT * 2048 + (size_t)4 * NT * 32 * sizeof(float) +
           (size_t)3 * NT * 32 * 8 * sizeof(float) + (size_t)2 * 3 * 8 * HD * sizeof(h16);
}

template <int NT>
__global__ __launch_bounds__(64 * IMG_WAVES) void attn_bwd_ring_kernel(const h16* __restrict__ qkv, const h16* __restrict__ ctx,
                                                                       const h16* __restrict__ dctx, const float* __restrict__ lse2,
                                                                       h16* __restrict__ dqkv, int T, int H, int D, float scale,
                                                                       float scale_log2e, const LoraDown lo) {
    constexpr int ROWS = NT * 32, IR = ring_img_rows<NT>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* sQ = (h16*)smem;
    h16* sdO = sQ + IR * HD;
    h16* sK0 = sdO + IR * HD;                 // K of even / odd heads
    char* sE = (char*)(sK0 + 2 * IR * HD);    // [2][NT][2048]: dS tiles, [key][query] h16, 64-byte rows, 8-byte chunk ch at ch ^ ((row>>2)&3)
    float* sLD = (float*)(sE + 2 * NT * 2048);   // [2][2][ROWS]: {LSE, delta} by head parity
    float* usum = sLD + 4 * ROWS;             // [3][ROWS][8]
    h16* sBd = (h16*)(usum + 3 * ROWS * 8);   // [2][3][8][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    const h16* dobase = dctx + (size_t)b * T * D;
    const h16* obase = ctx + (size_t)b * T * D;
    const int nb = (T + 31) >> 5;

    for (int i = tid; i < 3 * ROWS * 8; i += 64 * IMG_WAVES) usum[i] = 0.f;
    for (int i = tid; i < 2 * NT * 2048 / 16; i += 64 * IMG_WAVES) ((f32x4*)sE)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tile reads past the end of the even-head K image fall into the first rows of the odd-head image, which is not loaded
    // before head 1: whatever the LDS held there enters dQ's products against ZEROED dS, and 0 x NaN is NaN -- make it zeros
    // (afterwards the region always holds K rows of some head: finite)
    if constexpr (ROWS > IR)
        for (int i = tid; i < (ROWS - IR) * HD * 2 / 16; i += 64 * IMG_WAVES) ((f32x4*)(sK0 + IR * HD))[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Every wave executes the same barrier sequence: one after the prologue, then per head one per step and one at its end.
    if (w == IMG_WAVES - 1) {
        // ================================ LOADER ================================
        // the head's [8][64] slices of Bd for query / key / value: requested early (bd_issue), stored to LDS a step later
        // (bd_commit) -- a load that is used at once costs the loader a full memory round trip, and every wave waits for it
        h16x8 bdv[3];
        auto bd_issue = [&](int hd) {
            const int j = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
            for (int md = 0; md < 3; ++md) {
#pragma unroll
                for (int k = 0; k < 8; ++k) bdv[md][k] = (h16)0.f;
                if (((lo.mods >> md) & 1u) && j < lo.r) bdv[md] = *(const h16x8*)(lo.W + (size_t)(md * lo.r + j) * ld + md * D + hd * HD + c8);
            }
        };
        auto bd_commit = [&](int hd) {
            const int j = lane >> 3, c8 = (lane & 7) * 8;
            h16* dst = sBd + (hd & 1) * 3 * 8 * HD;
#pragma unroll
            for (int md = 0; md < 3; ++md) *(h16x8*)(dst + md * 8 * HD + j * HD + c8) = bdv[md];
        };
        // row constants of head hd for the 32 queries of block blk: LSE (rows >= T: +inf -> P = 0) and
        // delta = rowsum(dO * O), each lane half sums 32 of the 64 features.  Split in two so that a block's loads are
        // requested one step BEFORE they are used: a dependent global load under this kernel's own traffic takes longer
        // than a whole step of the compute waves, and every wave meets the loader at the step's barrier.
        h16x8 rd8[4], ro8[4];
        float rlse = 0.f;
        auto consts_issue = [&](int hd, int blk) {
            const int t = blk * 32 + c, tc = t < T ? t : T - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                rd8[ks] = ld_attn(dobase + (size_t)tc * D + hd * HD + 8 * h + 16 * ks);
                ro8[ks] = ld_attn(obase + (size_t)tc * D + hd * HD + 8 * h + 16 * ks);
            }
            rlse = lse2[((size_t)b * H + hd) * T + tc];
        };
        auto consts_finish = [&](int hd, int blk) {
            const int t = blk * 32 + c;
            float delta_q = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int k = 0; k < 8; ++k) delta_q = fmaf(h2f(rd8[ks][k]), h2f(ro8[ks][k]), delta_q);
            delta_q += __shfl_xor(delta_q, 32, 64);
            if (h == 0) {
                float* Lw = sLD + (hd & 1) * 2 * ROWS;
                Lw[t] = t < T ? rlse : INFINITY;
                Lw[ROWS + t] = t < T ? delta_q : 0.f;
            }
        };
        stage_glds<IR, 1>(sQ, base, ld, T, 0, lane);
        stage_glds<IR, 1>(sdO, dobase, D, T, 0, lane);
        stage_glds<IR, 1>(sK0, base + D, ld, T, 0, lane);
        if (lo.W) {
            h16x8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (h16)0.f;
            for (int i = lane; i < T * 8; i += 64) *(h16x8*)(lo.out + ((size_t)b * T + (i >> 3)) * 64 + (i & 7) * 8) = z;
            bd_issue(0);
            bd_commit(0);
        }
        for (int blk = 0; blk < NT; ++blk) { consts_issue(0, blk); consts_finish(0, blk); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LDS_BARRIER();
#pragma unroll 1
        for (int hd = 0; hd < H; ++hd) {
            const bool more = hd + 1 < H;
            ISTAMP(hd, 0);
            if (more) {
                consts_issue(hd + 1, 0);
                if (lo.W) bd_issue(hd + 1);
            }
#pragma unroll 1
            for (int i = 0; i < nb; ++i) {
                if (i == 3) ISTAMP(hd, 1);
                if (more) {                                                        // next head's constants, one block per step
                    consts_finish(hd + 1, i);
                    if (i == 0 && lo.W) bd_commit(hd + 1);
                    if (i + 1 < NT) consts_issue(hd + 1, i + 1);
                    // K of the next head goes out in slices, one per step, AFTER the next block's loads: with LDS-DMA in flight the
                    // compiler drains the whole queue (vmcnt(0)) at the next use of an ordinary load, i.e. at the next step's
                    // consts_finish -- a slice has had a whole step to land by then, the whole image at once had not
                    {
                        constexpr int NG = IR / 8;
                        const int per = (NG + nb - 1) / nb;
                        const int lr = lane >> 3, lc = lane & 7;
                        h16* img = sK0 + ((hd + 1) & 1) * IR * HD;
                        const h16* src = base + D + (hd + 1) * HD;
                        for (int g = i * per; g < (i + 1) * per && g < NG; ++g) {
                            const int r = g * 8 + lr;
                            const int rs = r < T ? r : T - 1;
                            glds16a(src + (size_t)rs * ld + ((lc ^ swz(r)) << 3), img + g * 8 * HD);
                        }
                    }
                }
                if (i == nb - 1) {
                    if (more) for (int blk = nb; blk < NT; ++blk) { consts_finish(hd + 1, blk); if (blk + 1 < NT) consts_issue(hd + 1, blk + 1); }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // K of the next head has landed
                }
                if (i == 3) ISTAMP(hd, 2);
                LDS_BARRIER();
                if (i == 3) ISTAMP(hd, 3);
            }
            ISTAMP(hd, 4);
            if (more) {                                                            // Q / dO images are free after the last step's barrier
                stage_glds<IR, 1>(sQ, base + (hd + 1) * HD, ld, T, 0, lane);
                stage_glds<IR, 1>(sdO, dobase + (hd + 1) * HD, D, T, 0, lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ISTAMP(hd, 6);
            LDS_BARRIER();
            ISTAMP(hd, 7);
        }
        return;
    }
    if (w >= nb) {
        // ================================ idle wave (T < 32 * NT): barriers only ================================
        LDS_BARRIER();
        for (int hd = 0; hd < H; ++hd) {
            for (int i = 0; i < nb; ++i) LDS_BARRIER();
            LDS_BARRIER();
        }
        return;
    }
    // ================================ COMPUTE: key block w and query block w ================================
    const FragOffs fo = frag_offs(lane);
    const int tok = w * 32 + c, tokc = tok < T ? tok : T - 1;
    // exchange-slot offsets (bytes inside a 2 KiB slot).  Writer: own key = row c, queries 8*rq + 4*h + (0..3) = logical 8-byte
    // chunk 2*rq + h, stored at chunk ^ ((c >> 2) & 3): ewr[rq & 1] + 32 * (rq >> 1).  Reader: transposed 4 x 16 blocks,
    // rows 16*st + 4*(g>>1) + q4 + 8*e, logical chunk 4*(g&1) + p4: erd[e] + 1024 * st.
    int ewr[2], erd[2];
    {
        const int f = (c >> 2) & 3;
#pragma unroll
        for (int r1 = 0; r1 < 2; ++r1) ewr[r1] = c * 64 + (((2 * r1 + h) ^ f) << 3);      // (2*rq + h) ^ f = ((2*(rq&1) + h) ^ f) + 4*(rq>>1): f < 4
        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 4 * (g >> 1) + q4 + 8 * e;
            erd[e] = r * 64 + (((4 * (g & 1) + p4) ^ ((r >> 2) & 3)) << 3);
        }
    }
    h16x8 fb[4];                            // own v rows (B operand of dP); own k rows are re-read from the K image (registers are short)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = ld_attn(base + (size_t)tokc * ld + 2 * D + 8 * h + 16 * ks);
    if (w >= 4 && RING_PRIO) __builtin_amdgcn_s_setprio(1);
    LDS_BARRIER();

#pragma unroll 1
    for (int hd = 0; hd < H; ++hd) {
        const h16* sK = sK0 + (hd & 1) * IR * HD;
        const float* L = sLD + (hd & 1) * 2 * ROWS;
        const float* Dl = L + ROWS;
        f32x16 dq[2], dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dq[dt][r] = 0.f; dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        // Software pipeline inside the wave: the matrix products that CONSUME a step's P / dS (dV, dK of the own keys; dQ of the
        // own queries from the tile another wave handed over) are issued one step late, between the pieces of the next step's
        // exp block -- they are independent of it, so the matrix pipe runs under the VALU work instead of before / after it.
        // Step 0 runs them on zero operands.
        h16x8 pbk[2], dsk[2];                 // P and dS of the previous step, packed (B operands of its dV / dK products)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) { pbk[st][j] = (h16)0.f; dsk[st][j] = (h16)0.f; }
        int qp = w, kp = w;                   // previous step's query tile (own keys) / key tile (own queries)
        ISTAMP(hd, 0);
        auto take_ds = [&](int step, int kt_, int st) -> h16x8 {      // the dS tile of step `step` made for the own queries, keys 16*st ..
            const char* eslot = sE + ((step & 1) * NT + w) * 2048;
            h16x8 d = cat4(lds_read_tr16(eslot + 1024 * st + erd[0]), lds_read_tr16(eslot + 1024 * st + erd[1]));
            if (kt_ == nb - 1 && (T & 31)) {              // keys >= T of the partly filled tile contribute nothing
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (kt_ * 32 + 16 * st + 8 * (j >> 2) + 4 * h + (j & 3) >= T) d[j] = (h16)0.f;
            }
            return d;
        };
#pragma unroll 1
        for (int i = 0; i < nb; ++i) {
            int qt = w + i; if (qt >= nb) qt -= nb;
            int kt = w - i; if (kt < 0) kt += nb;
            if (i == 3) ISTAMP(hd, 1);
            JSTAMP(hd, i, 0);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = mfma32(row_frag_o(sQ, qt, fo.row[ks]), row_frag_o(sK, w, fo.row[ks]), s);     // S[q][key]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = mfma32(row_frag_o(sdO, qt, fo.row[ks]), fb[ks], dp);                         // dP[q][key]
            // previous step's products ride in the four pieces of this step's exp block: piece 0, 1 -> dV, dK (key half st = 0, 1),
            // piece 2, 3 -> dQ (st = 0, 1); the LDS fragments of piece n + 1 are requested before piece n's arithmetic
            h16x8 fv0[2], fk0[2], fv1[2], fk1[2], fq0[2], fq1[2], d0, d1;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { fv0[dt] = tr_frag_o(sdO, qp, 0, fo.tr[dt]); fk0[dt] = tr_frag_o(sQ, qp, 0, fo.tr[dt]); }
            __builtin_amdgcn_sched_barrier(0);
            JSTAMP(hd, i, 1);
            char* eslot_w = sE + ((i & 1) * NT + qt) * 2048;
            auto exp_piece = [&](int rq) {
                const f32x4 lq = *(const f32x4*)(L + qt * 32 + 8 * rq + 4 * h);
                const f32x4 dl = *(const f32x4*)(Dl + qt * 32 + 8 * rq + 4 * h);
                h16x4 ds4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float p = fexp2(fmaf(s[4 * rq + k], scale_log2e, -lq[k]));   // rows >= T: 0
                    const float d = p * (dp[4 * rq + k] - dl[k]);
                    // this piece's P and dS leave at once: packed for the own dV / dK products of the NEXT step (the previous
                    // step's packs of this half were consumed before this piece's arithmetic), and dS to the exchange slot --
                    // its LDS write completes under the following pieces instead of in front of the barrier
                    pbk[rq >> 1][4 * (rq & 1) + k] = f2h(p);
                    ds4[k] = f2h_sat(d);                                               // finite whatever happens (aliasing note above)
                    dsk[rq >> 1][4 * (rq & 1) + k] = ds4[k];
                }
                *(h16x4*)(eslot_w + ewr[rq & 1] + 32 * (rq >> 1)) = ds4;               // queries 8*rq + 4*h + (0..3) of key c
            };
            // piece 0
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { fv1[dt] = tr_frag_o(sdO, qp, 1, fo.tr[dt]); fk1[dt] = tr_frag_o(sQ, qp, 1, fo.tr[dt]); }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { dv[dt] = mfma32(fv0[dt], pbk[0], dv[dt]); dk[dt] = mfma32(fk0[dt], dsk[0], dk[dt]); }
            exp_piece(0);
            __builtin_amdgcn_sched_barrier(0);
            JSTAMP(hd, i, 2);
            // piece 1
            d0 = take_ds(i + 1, kp, 0);                                                           // slot set of step i - 1
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) fq0[dt] = tr_frag_o(sK, kp, 0, fo.tr[dt]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { dv[dt] = mfma32(fv1[dt], pbk[1], dv[dt]); dk[dt] = mfma32(fk1[dt], dsk[1], dk[dt]); }
            exp_piece(1);
            __builtin_amdgcn_sched_barrier(0);
            JSTAMP(hd, i, 3);
            // piece 2
            d1 = take_ds(i + 1, kp, 1);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) fq1[dt] = tr_frag_o(sK, kp, 1, fo.tr[dt]);
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { d0[j] = (h16)0.f; d1[j] = (h16)0.f; }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(fq0[dt], d0, dq[dt]);                   // dQ^T[d][q]
            exp_piece(2);
            __builtin_amdgcn_sched_barrier(0);
            JSTAMP(hd, i, 4);
            // piece 3
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(fq1[dt], d1, dq[dt]);
            exp_piece(3);
            __builtin_amdgcn_sched_barrier(0);
            JSTAMP(hd, i, 5);
            qp = qt; kp = kt;
            if (i == nb - 1) {
                // last step: its dV / dK cannot wait -- the loader overwrites the Q / dO images right after this step's barrier
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = mfma32(tr_frag_o(sdO, qp, st, fo.tr[dt]), pbk[st], dv[dt]);
                        dk[dt] = mfma32(tr_frag_o(sQ, qp, st, fo.tr[dt]), dsk[st], dk[dt]);
                    }
            }
            if (i == 3) ISTAMP(hd, 2);
            JSTAMP(hd, i, 6);
            LDS_BARRIER();
            if (i == 3) ISTAMP(hd, 3);
            JSTAMP(hd, i, 7);
        }
        ISTAMP(hd, 4);
        // drain the pipeline: the last step's dQ (K image and exchange slots stay valid until the end-of-head barrier)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const h16x8 d = take_ds(nb - 1, kp, st);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(tr_frag_o(sK, kp, st, fo.tr[dt]), d, dq[dt]);
        }
        // ---- head epilogue (the loader streams Q / dO of the next head meanwhile) ----
        {
            // (address arithmetic of this block derives from an opaque copy of the lane id: hoisted out of the head loop it
            //  would occupy registers the step loop needs and spill)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int c2 = ln & 31, h2 = ln >> 5, tok2 = w * 32 + c2, tokc2 = tok2 < T ? tok2 : T - 1;
            // next head's own v rows first: ahead of this head's stores in the wave's memory queue
            if (hd + 1 < H) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fb[ks] = ld_attn(base + (size_t)tokc2 * ld + 2 * D + (hd + 1) * HD + 8 * h2 + 16 * ks);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dq[dt][r] *= scale; dk[dt][r] *= scale; }
            if (lo.W) {
                const h16* sB = sBd + (hd & 1) * 3 * 8 * HD;
#pragma unroll
                for (int md = 0; md < 3; ++md) {
                    if (!((lo.mods >> md) & 1u)) continue;
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int st = 0; st < 2; ++st)
                            y = mfma32(down_frag_lds(sB + md * 8 * HD, dt, st, ln), pack8(md == 0 ? dq[dt] : md == 1 ? dk[dt] : dv[dt], st), y);
                    down_accumulate(usum + md * ROWS * 8, y, tok2, h2);
                }
            }
            ISTAMP(hd, 5);
            char* wimg = sE + ((nb & 1) * NT + w) * 2048;       // the slot set the last step did not use
            h16* drow = dqkv + (size_t)b * T * ld + hd * HD;
            store_rows32_half(wimg, dq, 1.f, drow, ld, w * 32, T, ln);
            store_rows32_half(wimg, dk, 1.f, drow + D, ld, w * 32, T, ln);
            store_rows32_half(wimg, dv, 1.f, drow + 2 * D, ld, w * 32, T, ln);
        }
        ISTAMP(hd, 6);
        LDS_BARRIER();
        ISTAMP(hd, 7);
    }
    if (lo.W && tok < T && 4 * h < lo.r) {
        h16* dst = lo.out + ((size_t)b * T + tok) * 64 + 4 * h;
#pragma unroll
        for (int md = 0; md < 3; ++md) {
            if (!((lo.mods >> md) & 1u)) continue;
            const f32x4 u4 = *(const f32x4*)(usum + (md * ROWS + tok) * 8 + 4 * h);
            h16x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = f2h(u4[k]);
            *(h16x4*)(dst + md * lo.r) = v;
        }
    }
}

template <int NT>
void launch_fwd_img(const h16* qkv, h16* ctx, float* lse2, int B, int T, int H, int D, const LoraDown& lo, hipStream_t s) {
    const float sl = 0.125f * 1.4426950408889634f;
    hipLaunchKernelGGL((attn_fwd_img_kernel<NT>), dim3(B), dim3(64 * IMG_WAVES), fwd_img_lds<NT>(), s, qkv, ctx, lse2, T, H, D, sl, lo);
}
int g_attn_ring = 1;      // VITLORA_ATTN_RING=0: the two-phase per-image backward (attn_bwd_img_kernel)
template <int NT>
void launch_bwd_img(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T, int H, int D,
                    const LoraDown& lo, hipStream_t s) {
    const float scale = 0.125f;
    if (g_attn_ring && T <= ring_img_rows<NT>()) {
        hipLaunchKernelGGL((attn_bwd_ring_kernel<NT>), dim3(B), dim3(64 * IMG_WAVES), bwd_ring_lds<NT>(), s, qkv, ctx, dctx, lse2, dqkv,
                           T, H, D, scale, scale * 1.4426950408889634f, lo);
        return;
    }
    hipLaunchKernelGGL((attn_bwd_img_kernel<NT>), dim3(B), dim3(64 * IMG_WAVES), bwd_img_lds<NT>(), s, qkv, ctx, dctx, lse2, dqkv,
                       T, H, D, scale, scale * 1.4426950408889634f, lo);
}

template <int NT>
size_t bwd_lds() { return (size_t)4 * NT * 32 * HD * sizeof(h16) + (size_t)2 * NT * 32 * sizeof(float); }

template <int NT>
void launch_bwd(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T, int H,
                int D, hipStream_t s) {
    const float scale = 0.125f;   // 64^-1/2
    hipLaunchKernelGGL((attn_bwd32_kernel<NT>), dim3(B * H), dim3(512), bwd_lds<NT>(), s, qkv, ctx, dctx, lse2, dqkv, T,
                       H, D, scale, scale * 1.4426950408889634f);
}

}  // namespace

void attention32_set_ring(int on) { g_attn_ring = on ? 1 : 0; }

int attention32_init(int device) {
    static bool done[64] = {};
    if (device < 0 || device >= 64) return -1;
    if (done[device]) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<1>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd32_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<7>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_fwd_img_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_img_lds<1>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_fwd_img_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_img_lds<7>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_img_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_img_lds<1>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_img_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_img_lds<7>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_ring_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_ring_lds<1>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_ring_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_ring_lds<7>());
    if (e != hipSuccess) return (int)e;
    if (const char* rg = getenv("VITLORA_ATTN_RING")) g_attn_ring = rg[0] != '0';
    done[device] = true;
    return 0;
}

int k_attention32_fwd(const h16* qkv, h16* ctx, float* lse2, int B, int T, int H, int D, hipStream_t s) {
    ProfScope prof_("attn_fwd32_kernel", 4.0 * B * H * (double)T * T * HD, (double)B * T * D * 8.0, s, 4.0 * B * H * (double)TPAD(T) * TPAD(T) * HD);
    const float sl = 0.125f * 1.4426950408889634f;
    if (T <= 32) hipLaunchKernelGGL((attn_fwd32_kernel<1>), dim3(B * H), dim3(64 * FWD_WAVES), 0, s, qkv, ctx, lse2, T, H, D, sl);
    else if (T <= 224) hipLaunchKernelGGL((attn_fwd32_kernel<7>), dim3(B * H), dim3(64 * FWD_WAVES), 0, s, qkv, ctx, lse2, T, H, D, sl);
    else return -1;
    return 0;
}

int k_attention32_bwd(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T,
                      int H, int D, hipStream_t s) {
    ProfScope prof_("attn_bwd32_kernel", 10.0 * B * H * (double)T * T * HD, (double)B * T * D * 16.0, s, 10.0 * B * H * (double)TPAD(T) * TPAD(T) * HD);
    if (T <= 32) launch_bwd<1>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, s);
    else if (T <= 224) launch_bwd<7>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, s);
    else return -1;
    return 0;
}

// Per-image persistent forms (one workgroup per image over all heads) with the LoRA down projection of the
// consuming linear fused in: W = Ad [64, D] of attention.output.dense (forward, t = ctx Ad^T) or Bd [64, 3D] of the
// fused q / k / v projection (backward, u = dqkv Bd^T); W == nullptr: attention only.  r <= 8.
int k_attention_img_fwd(const h16* qkv, h16* ctx, float* lse2, int B, int T, int H, int D, const h16* Ad, h16* t, int r,
                        hipStream_t s) {
    ProfScope prof_("attn_fwd_img_kernel", 4.0 * B * H * (double)T * T * HD, (double)B * T * D * 8.0, s, 4.0 * B * H * (double)TPAD(T) * TPAD(T) * HD);
    LoraDown lo;
    lo.W = Ad; lo.out = t; lo.r = r; lo.mods = 1u;
    if (!Ad || !t || r <= 0 || r > 8) { lo.W = nullptr; lo.out = nullptr; }
    if (T <= 32) launch_fwd_img<1>(qkv, ctx, lse2, B, T, H, D, lo, s);
    else if (T <= 224) launch_fwd_img<7>(qkv, ctx, lse2, B, T, H, D, lo, s);
    else return -1;
    return 0;
}
int k_attention_img_bwd(const h16* qkv, const h16* ctx, const h16* dctx, const float* lse2, h16* dqkv, int B, int T, int H, int D,
                        const h16* Bd, h16* u, int r, unsigned mods, hipStream_t s) {
    const bool ring = g_attn_ring && T <= (T <= 32 ? ring_img_rows<1>() : ring_img_rows<7>());      // the kernel launch_bwd_img picks
    // executed: the ring form issues 5 products per 32 x 32 tile pair (S, dP, dV, dK, dQ), the two-phase form 7 (S and dP twice)
    ProfScope prof_(ring ? "attn_bwd_ring_kernel" : "attn_bwd_img_kernel", 10.0 * B * H * (double)T * T * HD, (double)B * T * D * 16.0, s,
                    (ring ? 10.0 : 14.0) * B * H * (double)TPAD(T) * TPAD(T) * HD);
    LoraDown lo;
    lo.W = Bd; lo.out = u; lo.r = r; lo.mods = mods;
    if (!Bd || !u || r <= 0 || r > 8 || !mods) { lo.W = nullptr; lo.out = nullptr; }
    if (T <= 32) launch_bwd_img<1>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, lo, s);
    else if (T <= 224) launch_bwd_img<7>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, lo, s);
    else return -1;
    return 0;
}

}  // namespace VLNS
