// Fused GEMM epilogues, shared by the 128-row (gemm.hip) and 256-row (gemm256.hip) kernels.
// A lane owns 4 consecutive output columns n..n+3 of row m (operand-swapped MFMA).
#pragma once
#include "gemm.h"

// Operands an epilogue reads exactly once (the saved gelu'(z), the residual-stream row, adv / x0 of the fused PGD step):
// VL_EPI_NT=1 streams them with non-temporal loads.  Measured in round 5 (same box, alternating, profiles/r05_nt_loads_ab.txt)
// and NOT kept: inside the PGD iteration these operands are served from the Infinity Cache (adv / x0 stay resident between
// iterations, gelu'(z) and the stream rows were written a few kernels earlier) and the non-temporal form loses those hits
// (patch-gradient + PGD-step GEMM 0.19 -> 0.26 ms, GELU-backward GEMM +2 %, 527 -> 521 img/s).  The standalone K10, whose
// 617 MB do not fit the cache, gains 20 % from the same loads (elementwise.hip).
#ifndef VL_EPI_NT
#define VL_EPI_NT 0
#endif
namespace VLNS {
template <typename V> __device__ __forceinline__ V ld_once(const V* p) {
#if VL_EPI_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}


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
    if constexpr (EPI == EPI_STORE_H16) {
        h16x4 o = {f2h(v[0]), f2h(v[1]), f2h(v[2]), f2h(v[3])};
        *(h16x4*)((h16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_STORE_F32) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
    } else if constexpr (EPI == EPI_RESID_F32) {
        const f32x4 r = *(const f32x4*)((const float*)p.R + (size_t)m * p.ldr + n);
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v + r;
    } else if constexpr (EPI == EPI_GELU) {
        h16x4 g, a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const GeluParts gp = gelu_parts(v[i]);
            a[i] = f2h(v[i] * gp.cdf);
            g[i] = f2h(fmaf(v[i], gp.pdf, gp.cdf));
        }
        *(h16x4*)((h16*)p.C2 + (size_t)m * p.ldc2 + n) = g;
        *(h16x4*)((h16*)p.C + (size_t)m * p.ldc + n) = a;
    } else if constexpr (EPI == EPI_GELU_BWD) {
        const h16x4 z = *(const h16x4*)((const h16*)p.R + (size_t)m * p.ldr + n);
        h16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2h(v[i] * h2f(z[i]));
        *(h16x4*)((h16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_RESID_H16) {
        const h16x4 r = *(const h16x4*)((const h16*)p.R + (size_t)m * p.ldr + n);
        h16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2h(v[i] + h2f(r[i]));
        *(h16x4*)((h16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_DROP_ACC) {
        // d(input) = R + mask * (u (sA)) [* gelu'(z)]: the LoRA branch saw the dropped input
        const h16x4 r = *(const h16x4*)((const h16*)p.R + (size_t)m * p.ldr + n);
        h16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ms = drop_scale(p.drop_seed, p.drop_stream, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + i),
                                        p.drop_p, p.drop_inv_keep);
            float t = h2f(r[i]) + ms * v[i];
            if (p.G) t *= h2f(p.G[(size_t)m * p.ldg + n + i]);
            o[i] = f2h(t);
        }
        *(h16x4*)((h16*)p.C + (size_t)m * p.ldc + n) = o;
    } else if constexpr (EPI == EPI_PATCH_FWD) {
        if (m < p.Mvalid) {
            const int b = m / p.patches, pi = m - b * p.patches;
            // the token row of the 16-bit residual stream (round 4; the fp32 parity mode has its own kernels)
            const f32x4 pe = *(const f32x4*)(p.pos + (size_t)(1 + pi) * p.ldc + n);
            const f32x4 o = v + pe;
            *(h16x4*)((h16*)p.C + ((size_t)b * p.tokens + 1 + pi) * p.ldc + n) = h16x4{f2h(o[0]), f2h(o[1]), f2h(o[2]), f2h(o[3])};
        }
    } else if constexpr (EPI == EPI_PATCH_BWD || EPI == EPI_PATCH_PGD) {
        if (m < p.Mvalid) {
            const int b = m / p.patches, pi = m - b * p.patches;
            const int py = pi / p.grid, px = pi - py * p.grid;
            const int pp = p.psize * p.psize;
            const int c = n / pp, rem = n - c * pp;
            const int ph = rem / p.psize, pw = rem - ph * p.psize;
            const float s = p.inv_std[c] * (p.row_scale ? p.row_scale[b] : 1.f);
            f32x4 o = {v[0] * s, v[1] * s, v[2] * s, v[3] * s};
            // an fp16 backward that left its range shows up here as inf / NaN (the chain saturates or overflows upstream):
            // never silent -- the next API call reports VL_ERR_NONFINITE (sign(NaN) = 0: that pixel does not move)
            if (p.err_flag && !(fabsf(o[0]) < INFINITY && fabsf(o[1]) < INFINITY && fabsf(o[2]) < INFINITY && fabsf(o[3]) < INFINITY))
                *p.err_flag = 2;
            const size_t at = (((size_t)b * 3 + c) * p.img + py * p.psize + ph) * p.img + px * p.psize + pw;
            float* dst = (float*)p.C + at;
            if constexpr (EPI == EPI_PATCH_PGD) {
                // K10 on the gradient while it is still in registers (the same fp32 operations as pgd_step_kernel, in the
                // same order: bit-identical to the two-kernel form)
                const f32x4 a = ld_once((const f32x4*)dst);
                const f32x4 x = ld_once((const f32x4*)((const float*)p.R + at));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sg = (o[k] > 0.f) ? 1.f : ((o[k] < 0.f) ? -1.f : 0.f);
                    const float t = a[k] + p.pgd_alpha * sg;
                    const float d = fminf(fmaxf(t - x[k], -p.pgd_eps), p.pgd_eps);
                    o[k] = fminf(fmaxf(x[k] + d, p.pgd_lo), p.pgd_hi);
                }
            }
            *(f32x4*)dst = o;
        }
    }
}

// GELU_BWD on 16 columns with gelu'(z) already in registers (z0 = columns 0..7, z1 = 8..15)
__device__ __forceinline__ void epilogue_gelu_bwd16(const GemmArgs& p, int m, int n0, const f32x4 (&v)[4], h16x8 z0, h16x8 z1) {
    h16x8 lo, hi;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo[k] = f2h(v[0][k] * h2f(z0[k]));
        lo[4 + k] = f2h(v[1][k] * h2f(z0[4 + k]));
        hi[k] = f2h(v[2][k] * h2f(z1[k]));
        hi[4 + k] = f2h(v[3][k] * h2f(z1[4 + k]));
    }
    h16* dst = (h16*)p.C + (size_t)m * p.ldc + n0;
    *(h16x8*)dst = lo;
    *(h16x8*)(dst + 8) = hi;
}

// RESID_H16 on 16 columns with the stream row already in registers
__device__ __forceinline__ void epilogue_resid16(const GemmArgs& p, int m, int n0, const f32x4 (&v)[4], h16x8 r0, h16x8 r1) {
    h16x8 lo, hi;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo[k] = f2h(v[0][k] + h2f(r0[k]));
        lo[4 + k] = f2h(v[1][k] + h2f(r0[4 + k]));
        hi[k] = f2h(v[2][k] + h2f(r1[k]));
        hi[4 + k] = f2h(v[3][k] + h2f(r1[4 + k]));
    }
    h16* dst = (h16*)p.C + (size_t)m * p.ldc + n0;
    *(h16x8*)dst = lo;
    *(h16x8*)(dst + 8) = hi;
}

// 16 consecutive output columns n0..n0+15 of row m held by one lane as v[0..3] (the 256-row
// kernel permutes the W rows of its LDS image so that a lane's four column tiles are adjacent):
// 16-byte loads / stores, 64 B (h16) or 256 B (f32) contiguous per row and lane quad.
template <int EPI>
__device__ __forceinline__ void epilogue_row16(const GemmArgs& p, int m, int n0, const f32x4 (&v)[4]) {
    auto pack8 = [](const f32x4& a, const f32x4& b) {
        h16x8 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = f2h(a[k]); o[4 + k] = f2h(b[k]); }
        return o;
    };
    if constexpr (EPI == EPI_STORE_H16) {
        h16* dst = (h16*)p.C + (size_t)m * p.ldc + n0;
        *(h16x8*)dst = pack8(v[0], v[1]);
        *(h16x8*)(dst + 8) = pack8(v[2], v[3]);
    } else if constexpr (EPI == EPI_STORE_F32) {
        float* dst = (float*)p.C + (size_t)m * p.ldc + n0;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 4 * q) = v[q];
    } else if constexpr (EPI == EPI_RESID_F32) {
        const float* r = (const float*)p.R + (size_t)m * p.ldr + n0;
        float* dst = (float*)p.C + (size_t)m * p.ldc + n0;
        f32x4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = *(const f32x4*)(r + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 4 * q) = v[q] + rv[q];
    } else if constexpr (EPI == EPI_GELU) {
        // C = gelu(z); C2 = gelu'(z) = Phi(z) + z phi(z): the backward needs only this derivative,
        // and exp(-z^2/2) is already at hand here, so it costs two more FMAs instead of a second
        // erf + exp pass in the dgrad epilogue.
        f32x4 a[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const GeluParts gp = gelu_parts(v[q][k]);
                a[q][k] = v[q][k] * gp.cdf;
                g[q][k] = fmaf(v[q][k], gp.pdf, gp.cdf);
            }
        h16* gd = (h16*)p.C2 + (size_t)m * p.ldc2 + n0;
        h16* ad = (h16*)p.C + (size_t)m * p.ldc + n0;
        *(h16x8*)gd = pack8(g[0], g[1]);
        *(h16x8*)(gd + 8) = pack8(g[2], g[3]);
        *(h16x8*)ad = pack8(a[0], a[1]);
        *(h16x8*)(ad + 8) = pack8(a[2], a[3]);
    } else if constexpr (EPI == EPI_GELU_BWD) {
        // R = gelu'(z) saved by the forward epilogue
        const h16* zs = (const h16*)p.R + (size_t)m * p.ldr + n0;
        epilogue_gelu_bwd16(p, m, n0, v, ld_once((const h16x8*)zs), ld_once((const h16x8*)(zs + 8)));
    } else if constexpr (EPI == EPI_RESID_H16) {
        const h16* rs = (const h16*)p.R + (size_t)m * p.ldr + n0;
        epilogue_resid16(p, m, n0, v, ld_once((const h16x8*)rs), ld_once((const h16x8*)(rs + 8)));
    } else if constexpr (EPI == EPI_NONE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(v[q]));
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) epilogue_store<EPI>(p, m, n0 + 4 * q, v[q]);
    }
}

}  // namespace VLNS
