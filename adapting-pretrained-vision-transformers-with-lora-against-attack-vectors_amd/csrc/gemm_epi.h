// Fused GEMM epilogues, shared by the 128-row (gemm.hip) and 256-row (gemm256.hip) kernels.
// A lane owns 4 consecutive output columns n..n+3 of row m (operand-swapped MFMA).
#pragma once
#include "gemm.h"

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): one v_rcp + one v_exp instead of the
// branchy libm erff; the shared exp(-x^2/2) also gives the Gaussian pdf for gelu'.
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float ax = fabsf(x) * 0.70710678118654752f;                 // |x| / sqrt(2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-(ax * ax) * 1.4426950408889634f);   // exp(-x^2/2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = 1.0f - poly * t * e;                         // erf(|x|/sqrt 2)
    const float erf_s = copysignf(erf_abs, x);
    return {0.5f * (1.0f + erf_s), 0.3989422804014327f * e};
}
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const GeluParts g = gelu_parts(x);
    return fmaf(x, g.pdf, g.cdf);
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int m, int n, f32x4 v) {
    if constexpr (EPI == EPI_STORE_BF16) {
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)((bf16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_STORE_F32) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == EPI_RESID_F32) {
        const f32x4 r = *(const f32x4*)((const float*)p.R + (size_t)m * p.ldr + n);
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v + r;
    } else if constexpr (EPI == EPI_GELU) {
        bf16x4 z = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        bf16x4 a = {f2bf(gelu_fast(v[0])), f2bf(gelu_fast(v[1])), f2bf(gelu_fast(v[2])), f2bf(gelu_fast(v[3]))};
        *(bf16x4*)((bf16*)p.C2 + (size_t)m * p.ldc2 + n) = z;
        *(bf16x4*)((bf16*)p.C + (size_t)m * p.ldc + n) = a;
    } else if constexpr (EPI == EPI_GELU_BWD) {
        const bf16x4 z = *(const bf16x4*)((const bf16*)p.R + (size_t)m * p.ldr + n);
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2bf(v[i] * gelu_grad_fast(bf2f(z[i])));
        *(bf16x4*)((bf16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_PATCH_FWD) {
        if (m < p.Mvalid) {
            const int b = m / p.patches, pi = m - b * p.patches;
            const f32x4 pe = *(const f32x4*)(p.pos + (size_t)(1 + pi) * p.ldc + n);
            *(f32x4*)((float*)p.C + ((size_t)b * p.tokens + 1 + pi) * p.ldc + n) = v + pe;
        }
    } else if constexpr (EPI == EPI_PATCH_BWD) {
        if (m < p.Mvalid) {
            const int b = m / p.patches, pi = m - b * p.patches;
            const int py = pi / p.grid, px = pi - py * p.grid;
            const int pp = p.psize * p.psize;
            const int c = n / pp, rem = n - c * pp;
            const int ph = rem / p.psize, pw = rem - ph * p.psize;
            const float s = p.inv_std[c];
            f32x4 o = {v[0] * s, v[1] * s, v[2] * s, v[3] * s};
            float* dst = (float*)p.C + (((size_t)b * 3 + c) * p.img + py * p.psize + ph) * p.img +
                         px * p.psize + pw;
            *(f32x4*)dst = o;
        }
    }
}
