// Last encoder layer, CLS rows only.  The classifier reads token 0 of the final LayerNorm (modeling_vit.py:560-561), so in
// the LAST layer everything after the key / value projection is needed for one query row per image: attention of the CLS
// query over all keys, then output projection, LayerNorm, MLP and the residual adds on B rows instead of B * T.  Backward
// likewise: the gradient enters at the CLS rows only, reaches every token again through dK / dV of the last attention.
// Exactly the reference's result (nothing that reaches the logits or the input gradient is dropped); the arithmetic of
// the kernels below mirrors the per-image MFMA kernels of attention32.hip (fp16 roundings at the same places).
#include "kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

constexpr int HD = 64;

// sum over the 8 lanes that share a row (lane & 7 = 16-byte chunk of the 128-byte head row)
__device__ __forceinline__ float row8_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v + __shfl_xor(v, 4, 64);
}
// sum / max over the 8 row slots of a wave (lane >> 3), same chunk
__device__ __forceinline__ float slot8_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float slot8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ void cvt8(const h16x8& t, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = h2f(t[k]);
}

// One wave per (image, head): scores of the CLS query against every key, softmax, context row.  A wave walks the keys EIGHT ROWS
// AT A TIME: lane = (row slot r = lane >> 3, chunk c = lane & 7), so a load instruction covers eight whole 128-byte head rows
// (round 4; lane = key row made every lane walk its own row, 16 bytes of 64 different lines per instruction: 50 / 155 us per
// launch forward / backward at batch 256 against the 28 / 70 us their bytes take).  Dot products are 8-element partial sums per
// lane + a 3-step butterfly over the row's lanes.
// ctx_c [B, D] h16 (compact), lse_c [B, H] fp32 (base-2 log-sum-exp of the scaled scores, as attention32.hip saves it)
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx_c,
                                                           float* __restrict__ lse_c, int B, int T, int H, int D,
                                                           float scale_log2e) {
    __shared__ float sp[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane >> 3, c = lane & 7;
    const int idx = blockIdx.x * 4 + w;
    if (idx >= B * H) return;
    const int b = idx / H, hd = idx - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    float q[8];
    cvt8(*(const h16x8*)(base + hd * HD + 8 * c), q);
    float mx = -INFINITY;
    for (int j0 = 0; j0 < T; j0 += 8) {
        const int j = j0 + r;
        float a = 0.f;
        if (j < T) {
            float k[8];
            cvt8(*(const h16x8*)(base + (size_t)j * ld + D + hd * HD + 8 * c), k);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(q[e], k[e], a);
        }
        a = row8_sum(a);
        if (j < T) { mx = fmaxf(mx, a); if (c == 0) sp[w][j] = a; }
    }
    mx = slot8_max(mx);
    const float mc = -mx * scale_log2e;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS writes, read below by the same wave
    float l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < T; j0 += 8) {
        const int j = j0 + r;
        if (j < T) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sp[w][j], scale_log2e, mc));
            l += p;
            const float p16 = h2f(f2h(p));         // P enters the P V product as h16 (attention32.hip: pack8 of the score tile)
            float v[8];
            cvt8(*(const h16x8*)(base + (size_t)j * ld + 2 * D + hd * HD + 8 * c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(p16, v[e], acc[e]);
        }
    }
    l = slot8_sum(l);                              // every row once: the 8 slots hold disjoint rows, the chunk lanes the same ones
    const float inv = 1.f / l;
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2h(slot8_sum(acc[e]) * inv);
    if (r == 0) *(h16x8*)(ctx_c + (size_t)b * D + hd * HD + 8 * c) = o;
    if (lane == 0) lse_c[idx] = mx * scale_log2e + log2f(l);
}

// backward of the same: dO = d(ctx row) (compact), O = ctx row (compact).  Writes the FULL dqkv [B*T, 3D]: dK, dV rows of every
// token, dQ of the CLS row, zeros in the dQ part of all other rows.  Same eight-rows-per-instruction walk; ONE pass over K and V.
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const h16* __restrict__ qkv, const h16* __restrict__ ctx_c,
                                                           const h16* __restrict__ dctx_c, const float* __restrict__ lse_c,
                                                           h16* __restrict__ dqkv, int B, int T, int H, int D, float scale,
                                                           float scale_log2e) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane >> 3, c = lane & 7;
    const int idx = blockIdx.x * 4 + w;
    if (idx >= B * H) return;
    const int b = idx / H, hd = idx - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld + hd * HD + 8 * c;
    h16* dbase = dqkv + (size_t)b * T * ld + hd * HD + 8 * c;
    float q[8], dO[8];
    cvt8(*(const h16x8*)base, q);
    cvt8(*(const h16x8*)(dctx_c + (size_t)b * D + hd * HD + 8 * c), dO);
    float delta = 0.f;
    {
        float o[8];
        cvt8(*(const h16x8*)(ctx_c + (size_t)b * D + hd * HD + 8 * c), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) delta = fmaf(dO[e], o[e], delta);
        delta = row8_sum(delta);
    }
    const float lse = lse_c[idx];
    h16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (h16)0.f;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < T; j0 += 8) {
        const int j = j0 + r;
        const bool live = j < T;
        float k[8], v[8];
        float sc = 0.f, dp = 0.f;
        if (live) {
            cvt8(*(const h16x8*)(base + (size_t)j * ld + D), k);
            cvt8(*(const h16x8*)(base + (size_t)j * ld + 2 * D), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc = fmaf(q[e], k[e], sc); dp = fmaf(dO[e], v[e], dp); }
        }
        sc = row8_sum(sc);
        dp = row8_sum(dp);
        if (live) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sc, scale_log2e, -lse));
            const float p16 = h2f(f2h(p));
            const float ds16 = h2f(f2h(p * (dp - delta)));       // dS enters its products as h16, the 1/sqrt(d) factor is applied to the sums
            h16x8 ok, ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ok[e] = f2h(ds16 * q[e] * scale);
                ov[e] = f2h(p16 * dO[e]);
                acc[e] = fmaf(ds16, k[e], acc[e]);
            }
            *(h16x8*)(dbase + (size_t)j * ld + D) = ok;
            *(h16x8*)(dbase + (size_t)j * ld + 2 * D) = ov;
            if (j > 0) *(h16x8*)(dbase + (size_t)j * ld) = z8;
        }
    }
    h16x8 dq;
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = f2h(slot8_sum(acc[e]) * scale);
    if (r == 0) *(h16x8*)dbase = dq;
}

// dst[b][0..D) = src[b * stride .. + D)   (fp32, D % 4 == 0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const h16* __restrict__ src, float* __restrict__ dst, int B, int D,
                                                          int64_t stride) {
    const int n4 = D / 4;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * n4) return;
    const int b = (int)(i / n4), c = (int)(i - (int64_t)b * n4);
    const h16x4 v = *(const h16x4*)(src + b * stride + 4 * c);
    *(f32x4*)(dst + (int64_t)b * D + 4 * c) = f32x4{h2f(v[0]), h2f(v[1]), h2f(v[2]), h2f(v[3])};
}
// dst[b * stride .. + D) = src[b][0..D)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, h16* __restrict__ dst, int B, int D,
                                                           int64_t stride) {
    const int n4 = D / 4;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * n4) return;
    const int b = (int)(i / n4), c = (int)(i - (int64_t)b * n4);
    const f32x4 v = *(const f32x4*)(src + (int64_t)b * D + 4 * c);
    *(h16x4*)(dst + b * stride + 4 * c) = h16x4{f2h_sat(v[0]), f2h_sat(v[1]), f2h_sat(v[2]), f2h_sat(v[3])};
}

}  // namespace

int k_attn_cls_fwd(const h16* qkv, h16* ctx_c, float* lse_c, int B, int T, int H, int D, hipStream_t s) {
    if (T > 256 || D != H * HD) return -1;
    ProfScope prof_("attn_cls_fwd_kernel", 4.0 * B * H * (double)T * HD, (double)B * T * D * 4.0, s);
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((B * H + 3) / 4), dim3(256), 0, s, qkv, ctx_c, lse_c, B, T, H, D,
                       0.125f * 1.4426950408889634f);
    return 0;
}
int k_attn_cls_bwd(const h16* qkv, const h16* ctx_c, const h16* dctx_c, const float* lse_c, h16* dqkv, int B, int T, int H, int D,
                   hipStream_t s) {
    if (T > 256 || D != H * HD) return -1;
    ProfScope prof_("attn_cls_bwd_kernel", 10.0 * B * H * (double)T * HD, (double)B * T * D * 10.0, s);
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((B * H + 3) / 4), dim3(256), 0, s, qkv, ctx_c, dctx_c, lse_c, dqkv, B, T, H, D,
                       0.125f, 0.125f * 1.4426950408889634f);
    return 0;
}
void k_gather_rows(const h16* src, float* dst, int B, int D, int64_t stride, hipStream_t s) {
    const int64_t n = (int64_t)B * (D / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, B, D, stride);
}
void k_scatter_rows(const float* src, h16* dst, int B, int D, int64_t stride, hipStream_t s) {
    const int64_t n = (int64_t)B * (D / 4);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, B, D, stride);
}

}  // namespace VLNS
