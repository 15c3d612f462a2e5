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

__device__ __forceinline__ void load_row64(const h16* p, float (&v)[HD]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const h16x8 t = *(const h16x8*)(p + 8 * c);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[8 * c + k] = h2f(t[k]);
    }
}

// one wave per (image, head): scores of the CLS query against every key, softmax, context row.
// ctx_c [B, D] h16 (compact), lse_c [B, H] fp32 (base-2 log-sum-exp of the scaled scores, as attention32.hip saves it)
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const h16* __restrict__ qkv, h16* __restrict__ ctx_c,
                                                           float* __restrict__ lse_c, int B, int T, int H, int D,
                                                           float scale_log2e) {
    __shared__ float sp[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + w;
    if (idx >= B * H) return;
    const int b = idx / H, hd = idx - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    float q[HD];
    load_row64(base + hd * HD, q);
    float s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = lane + 64 * jj;
        s[jj] = -INFINITY;
        if (j < T) {
            const h16* kr = base + (size_t)j * ld + D + hd * HD;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const h16x8 kc = *(const h16x8*)(kr + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf(q[8 * c + e], h2f(kc[e]), a);
            }
            s[jj] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    const float mc = -mx * scale_log2e;
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = lane + 64 * jj;
        float p = 0.f;
        if (j < T) p = __builtin_amdgcn_exp2f(fmaf(s[jj], scale_log2e, mc));
        l += p;
        sp[w][j] = h2f(f2h(p));            // P enters the P V product as h16 (attention32.hip: pack8 of the score tile)
    }
    l = wave_sum(l);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS writes, read below by the same wave
    float acc = 0.f;
    const h16* vb = base + 2 * D + hd * HD + lane;
#pragma unroll 8
    for (int j = 0; j < T; ++j) acc = fmaf(sp[w][j], h2f(vb[(size_t)j * ld]), acc);
    ctx_c[(size_t)b * D + hd * HD + lane] = f2h(acc * (1.f / l));
    if (lane == 0) lse_c[idx] = mx * scale_log2e + log2f(l);
}

// backward of the same: dO = d(ctx row) (compact), O = ctx row (compact).  Writes the FULL dqkv [B*T, 3D]: dK, dV rows of every
// token, dQ of the CLS row, zeros in the dQ part of all other rows.
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const h16* __restrict__ qkv, const h16* __restrict__ ctx_c,
                                                           const h16* __restrict__ dctx_c, const float* __restrict__ lse_c,
                                                           h16* __restrict__ dqkv, int B, int T, int H, int D, float scale,
                                                           float scale_log2e) {
    __shared__ float sds[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + w;
    if (idx >= B * H) return;
    const int b = idx / H, hd = idx - b * H;
    const int ld = 3 * D;
    const h16* base = qkv + (size_t)b * T * ld;
    h16* dbase = dqkv + (size_t)b * T * ld;
    float q[HD], dO[HD];
    load_row64(base + hd * HD, q);
    load_row64(dctx_c + (size_t)b * D + hd * HD, dO);
    float delta = 0.f;
    {
        float o[HD];
        load_row64(ctx_c + (size_t)b * D + hd * HD, o);
#pragma unroll
        for (int d = 0; d < HD; ++d) delta = fmaf(dO[d], o[d], delta);
    }
    const float lse = lse_c[idx];
    h16x8 z8;
#pragma unroll
    for (int k = 0; k < 8; ++k) z8[k] = (h16)0.f;
#pragma unroll 1
    for (int jj = 0; jj < 4; ++jj) {
        const int j = lane + 64 * jj;
        float ds16 = 0.f;
        if (j < T) {
            const h16* kr = base + (size_t)j * ld + D + hd * HD;
            const h16* vr = base + (size_t)j * ld + 2 * D + hd * HD;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const h16x8 kc = *(const h16x8*)(kr + 8 * c), vc = *(const h16x8*)(vr + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s = fmaf(q[8 * c + e], h2f(kc[e]), s); dp = fmaf(dO[8 * c + e], h2f(vc[e]), dp); }
            }
            const float p = __builtin_amdgcn_exp2f(fmaf(s, scale_log2e, -lse));
            const float p16 = h2f(f2h(p));
            ds16 = h2f(f2h(p * (dp - delta)));       // dS enters its products as h16, the 1/sqrt(d) factor is applied to the sums
            h16* rk = dbase + (size_t)j * ld + D + hd * HD;
            h16* rv = dbase + (size_t)j * ld + 2 * D + hd * HD;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                h16x8 ok, ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ok[e] = f2h(ds16 * q[8 * c + e] * scale);
                    ov[e] = f2h(p16 * dO[8 * c + e]);
                }
                *(h16x8*)(rk + 8 * c) = ok;
                *(h16x8*)(rv + 8 * c) = ov;
            }
            if (j > 0) {
                h16* rq = dbase + (size_t)j * ld + hd * HD;
#pragma unroll
                for (int c = 0; c < 8; ++c) *(h16x8*)(rq + 8 * c) = z8;
            }
        }
        sds[w][j] = ds16;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;
    const h16* kb = base + D + hd * HD + lane;
#pragma unroll 8
    for (int j = 0; j < T; ++j) acc = fmaf(sds[w][j], h2f(kb[(size_t)j * ld]), acc);
    dbase[hd * HD + lane] = f2h(acc * scale);
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
