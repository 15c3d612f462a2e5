// Kernels of the fp32 PARITY mode (vl_config.precision = VL_PREC_F32): every operand, activation and
// gradient is fp32, matrix products run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: bit-for-bit a
// k-ordered fmaf chain).  This mode exists to be compared with the reference's fp32 CPU path at 1e-3
// (whitebox_attacks.py:22-38, train_loras.py:310-315 run in fp32, no autocast); it is written for
// clarity, not for the roofline -- the fp16-operand kernels (gemm256.hip, attention32.hip) are the
// performance path.
#include <cstdio>
#include <cstdlib>

#include "f32_kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// C[M,N] = alpha * A[M,K] * op(W) + bias + R      64 x 64 x 16 tiles, 4 waves (2 x 2), 16x16x4 f32 MFMA
// with the operands swapped (D[n][m]) so that a lane owns 4 consecutive output columns.
// ---------------------------------------------------------------------------------------------
constexpr int FBM = 64, FBN = 64, FBK = 16, FLD = FBK + 1;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32 p) {
    __shared__ float sA[FBM * FLD];
    __shared__ float sW[FBN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // non-transposed operand: thread -> (row = tid / 4, 4 consecutive k); transposed: (k = tid / 16, 4 consecutive rows)
    const int r_nt = tid >> 2, k_nt = (tid & 3) * 4;
    const int k_tr = tid >> 4, r_tr = (tid & 15) * 4;
    int64_t arow = -1;                       // source row of the A operand (non-transposed form)
    if (!p.transA) {
        int m = m0 + r_nt;
        if (m < p.M) {
            if (p.a_gather) m = m + m / p.patches + 1;     // GEMM row b*patches + pi reads token row b*tokens + 1 + pi
            arow = m;
        }
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
        f32x4_t av = zero, wv = zero;
        if (!p.transA) {
            if (arow >= 0 && k0 + k_nt < p.K) av = *(const f32x4_t*)(p.A + arow * p.lda + k0 + k_nt);
        } else {
            if (k0 + k_tr < p.K && m0 + r_tr < p.M) av = *(const f32x4_t*)(p.A + (int64_t)(k0 + k_tr) * p.lda + m0 + r_tr);
        }
        if (!p.transW) {
            if (n0 + r_nt < p.N && k0 + k_nt < p.K) wv = *(const f32x4_t*)(p.W + (int64_t)(n0 + r_nt) * p.ldw + k0 + k_nt);
        } else {
            if (k0 + k_tr < p.K && n0 + r_tr < p.N) wv = *(const f32x4_t*)(p.W + (int64_t)(k0 + k_tr) * p.ldw + n0 + r_tr);
        }
        __syncthreads();                                 // the previous step's fragment reads are done
        if (!p.transA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sA[r_nt * FLD + k_nt + i] = av[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) sA[(r_tr + i) * FLD + k_tr] = av[i];
        }
        if (!p.transW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sW[r_nt * FLD + k_nt + i] = wv[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) sW[(r_tr + i) * FLD + k_tr] = wv[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = sA[(wm * 32 + i * 16 + (lane & 15)) * FLD + kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = sW[(wn * 32 + j * 16 + (lane & 15)) * FLD + kk * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + (lane & 15);
        if (m >= p.Mstore) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + 4 * (lane >> 4);
            if (n >= p.N) continue;                      // N is a multiple of 4: a lane's 4 columns are in or out together
            f32x4_t v = acc[i][j] * p.alpha;
            if (p.bias) v += *(const f32x4_t*)(p.bias + n);
            if (p.R) v += *(const f32x4_t*)(p.R + (int64_t)m * p.ldr + n);
            *(f32x4_t*)(p.C + (int64_t)m * p.ldc + n) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same product on 128 x 128 x 16 tiles (round 3): 4 waves (2 x 2), 64 x 64 outputs per wave = 16 accumulators of the
// 16x16x4 f32 MFMA, i.e. 16 MFMAs per 8 fragment reads (the 64 x 64 kernel above: 4 per 4); the next K tile's global loads
// are requested before the current tile's products and land in the OTHER LDS buffer after them: one barrier per K tile, the
// memory latency under 64 MFMAs per wave.  Same arithmetic (k-ordered fmaf chain per output): bit-identical results.
// Used when the output is at least one tile in both directions; ragged edges are zero-filled / not stored.
// ---------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 128;

__global__ __launch_bounds__(256, 2) void gemm_f32_big_kernel(const GemmF32 p) {
    __shared__ float sA[2][GBM * FLD];
    __shared__ float sW[2][GBN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // non-transposed operand: thread -> (row = tid / 2, 8 consecutive k); transposed: (k = tid / 16, 8 consecutive rows)
    const int r_nt = tid >> 1, k_nt = (tid & 1) * 8;
    const int k_tr = tid >> 4, r_tr = (tid & 15) * 8;
    int64_t arow = -1;
    if (!p.transA) {
        int m = m0 + r_nt;
        if (m < p.M) {
            if (p.a_gather) m = m + m / p.patches + 1;
            arow = m;
        }
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t av[2], wv[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            av[q] = zero; wv[q] = zero;
            if (!p.transA) {
                if (arow >= 0 && k0 + k_nt + 4 * q < p.K) av[q] = *(const f32x4_t*)(p.A + arow * p.lda + k0 + k_nt + 4 * q);
            } else {
                if (k0 + k_tr < p.K && m0 + r_tr + 4 * q < p.M) av[q] = *(const f32x4_t*)(p.A + (int64_t)(k0 + k_tr) * p.lda + m0 + r_tr + 4 * q);
            }
            if (!p.transW) {
                if (n0 + r_nt < p.N && k0 + k_nt + 4 * q < p.K) wv[q] = *(const f32x4_t*)(p.W + (int64_t)(n0 + r_nt) * p.ldw + k0 + k_nt + 4 * q);
            } else {
                if (k0 + k_tr < p.K && n0 + r_tr + 4 * q < p.N) wv[q] = *(const f32x4_t*)(p.W + (int64_t)(k0 + k_tr) * p.ldw + n0 + r_tr + 4 * q);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!p.transA) sA[buf][r_nt * FLD + k_nt + 4 * q + i] = av[q][i];
                else sA[buf][(r_tr + 4 * q + i) * FLD + k_tr] = av[q][i];
                if (!p.transW) sW[buf][r_nt * FLD + k_nt + 4 * q + i] = wv[q][i];
                else sW[buf][(r_tr + 4 * q + i) * FLD + k_tr] = wv[q][i];
            }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
        const bool more = k0 + FBK < p.K;
        if (more) fetch(k0 + FBK);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = sA[buf][(wm * 64 + i * 16 + (lane & 15)) * FLD + kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = sW[buf][(wn * 64 + j * 16 + (lane & 15)) * FLD + kk * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.Mstore) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
            if (n >= p.N) continue;
            f32x4_t v = acc[i][j] * p.alpha;
            if (p.bias) v += *(const f32x4_t*)(p.bias + n);
            if (p.R) v += *(const f32x4_t*)(p.R + (int64_t)m * p.ldr + n);
            *(f32x4_t*)(p.C + (int64_t)m * p.ldc + n) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int LNV = 8;       // float4 chunks per lane: rows up to 8 * 64 * 4 = 2048 features
// LayerNorm (ViT eps 1e-12, configuration_vit.py:58; Swin 1e-5): one wave per row, D <= 2048
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float lane_group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NV float4 chunks per lane, G lanes per row (64 / G rows per wave), like ln_bwd_f32_kernel below: with one wave per row a
// 96-wide row (Swin patch embedding) used 24 lanes of 64 and carried the registers of a 2048-wide one
template <int NV, int G>
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ h,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int M, int D, float eps) {
    const int lane = threadIdx.x & 63, li = lane % G;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    const bool live = row < M;
    const int64_t off = (int64_t)(live ? row : M - 1) * D;
    const int nv = D >> 2;
    f32x4_t v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = li + i * G;
        v[i] = c < nv ? *(const f32x4_t*)(x + off + c * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = lane_group_sum<G>(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = li + i * G;
        if (c < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(lane_group_sum<G>(q) / D + eps);
    if (!live) return;
    if (li == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = li + i * G;
        if (c < nv) {
            const f32x4_t g = *(const f32x4_t*)(gamma + c * 4), b = *(const f32x4_t*)(beta + c * 4);
            f32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mean) * rstd * g[k] + b[k];
            *(f32x4_t*)(h + off + c * 4) = o;
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dh * gamma
// NV float4 chunks per lane, G lanes per row (64 / G rows per wave): narrow rows (Swin stage 1: D = 96) would otherwise leave
// 40 of 64 lanes idle and carry the registers of the widest row the kernel supports
template <int NV, int G>
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ gamma, const float* __restrict__ dres,
                                                         float* __restrict__ dx, int M, int D) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, li = lane & (G - 1);
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
    const bool live = row < M;
    const int rr = live ? row : M - 1;
    const float mean = mean_in[rr], rstd = rstd_in[rr];
    const int64_t off = (int64_t)rr * D;
    const int nv = D >> 2;
    f32x4_t g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = li + i * G;
        g[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        xh[i] = g[i];
        if (c < nv) {
            const f32x4_t d = *(const f32x4_t*)(dh + off + c * 4);
            const f32x4_t xv = *(const f32x4_t*)(x + off + c * 4);
            const f32x4_t gm = *(const f32x4_t*)(gamma + c * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[i][k] = d[k] * gm[k];
                xh[i][k] = (xv[k] - mean) * rstd;
                s1 += g[i][k];
                s2 += g[i][k] * xh[i][k];
            }
        }
    }
    const float c1 = lane_group_sum<G>(s1) / D, c2 = lane_group_sum<G>(s2) / D;
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = li + i * G;
        if (c < nv) {
            const f32x4_t r = dres ? *(const f32x4_t*)(dres + off + c * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            f32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = r[k] + rstd * (g[i][k] - c1 - xh[i][k] * c2);
            *(f32x4_t*)(dx + off + c * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// elementwise pieces
// ---------------------------------------------------------------------------------------------
__global__ void gelu_fwd_f32_kernel(const float* __restrict__ z, float* __restrict__ a, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) a[i] = gelu_f(z[i]);
}
__global__ void gelu_bwd_f32_kernel(float* __restrict__ dz, const float* __restrict__ z, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = z[i];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
        dz[i] *= cdf + x * pdf;
    }
}
// dst = src * mask (add == 0) or dst += src * mask (add != 0); mask = LoRA dropout keep-mask of (seed, stream)
__global__ void mask_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, int add, uint64_t seed,
                                uint32_t stream, float p, float inv_keep) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = src[i] * drop_scale(seed, stream, (uint64_t)i, p, inv_keep);
        dst[i] = add ? dst[i] + v : v;
    }
}
// pixels [B,3,S,S] -> patches [B*NP, 3*P*P] fp32, (x - mean) / std fused
__global__ void patch_gather_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int S, int P, int G,
                                        int normalise, float m0, float m1, float m2, float is0, float is1, float is2) {
    const int K = 3 * P * P;
    const int64_t total = (int64_t)B * G * G * K;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
        const int col = (int)(t % K);
        const int64_t m = t / K;
        const int c = col / (P * P), rem = col - c * P * P;
        const int ph = rem / P, pw = rem - ph * P;
        const int b = (int)(m / (G * G)), pi = (int)(m - (int64_t)b * G * G);
        const int py = pi / G, px = pi - py * G;
        float v = x[(((int64_t)b * 3 + c) * S + py * P + ph) * S + px * P + pw];
        if (normalise) v = (v - (c == 0 ? m0 : (c == 1 ? m1 : m2))) * (c == 0 ? is0 : (c == 1 ? is1 : is2));
        out[t] = v;
    }
}
// d(pixels)[b,c,y,x] = d(patches)[b*NP + pi][c*P*P + ph*P + pw] * inv_std[c]
__global__ void patch_scatter_f32_kernel(const float* __restrict__ dp, float* __restrict__ gx, int B, int S, int P, int G,
                                         float is0, float is1, float is2) {
    const int K = 3 * P * P;
    const int64_t total = (int64_t)B * 3 * S * S;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += stride) {
        const int xx = (int)(t % S);
        int64_t r = t / S;
        const int yy = (int)(r % S); r /= S;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        const int py = yy / P, ph = yy - py * P, px = xx / P, pw = xx - px * P;
        const int64_t m = (int64_t)b * G * G + py * G + px;
        gx[t] = dp[m * K + c * P * P + ph * P + pw] * (c == 0 ? is0 : (c == 1 ? is1 : is2));
    }
}

// ---------------------------------------------------------------------------------------------
// attention (HF eager attention, modeling_vit.py:164-189): softmax(Q K^T / 8) V per (image, head),
// head_dim 64 = one lane per feature.  K and V (backward phase B: Q and dO) of the head live in LDS
// as rows of 65 floats (conflict-free for both "lane = row" and "lane = feature" reads).
// ---------------------------------------------------------------------------------------------
constexpr int AHD = 64, ALD = 65, ATMAX = 224;

struct AttnLds {
    float* X;      // [T][ALD]
    float* Y;      // [T][ALD]
    float* vecA;   // [4][64]  per-wave broadcast row (q / k)
    float* vecB;   // [4][64]  per-wave broadcast row (dO / v)
    float* pa;     // [4][ATMAX]
    float* pb;     // [4][ATMAX]
    float* delta;  // [ATMAX]
};
__device__ __forceinline__ AttnLds attn_lds(float* sm, int T) {
    AttnLds l;
    l.X = sm; l.Y = l.X + T * ALD; l.vecA = l.Y + T * ALD; l.vecB = l.vecA + 4 * 64;
    l.pa = l.vecB + 4 * 64; l.pb = l.pa + 4 * ATMAX; l.delta = l.pb + 4 * ATMAX;
    return l;
}
size_t attn_lds_bytes(int T) { return ((size_t)2 * T * ALD + 8 * 64 + 8 * ATMAX + ATMAX) * sizeof(float); }

__device__ __forceinline__ void attn_stage(float* dst, const float* src, int ld, int T, int tid) {
    for (int i = tid; i < T * 16; i += 256) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const f32x4_t v = *(const f32x4_t*)(src + (int64_t)r * ld + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[r * ALD + c4 + k] = v[k];
    }
}

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                           float* __restrict__ lse, int T, int H, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const AttnLds L = attn_lds(sm, T);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)b * T * ld + hd * AHD;
    attn_stage(L.X, base + D, ld, T, tid);          // K
    attn_stage(L.Y, base + 2 * D, ld, T, tid);      // V
    __syncthreads();
    float* qs = L.vecA + w * 64;
    float* ps = L.pa + w * ATMAX;
    for (int q = w; q < T; q += 4) {
        qs[lane] = base[(int64_t)q * ld + lane];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float s[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = lane + 64 * j;
            s[j] = -INFINITY;
            if (key < T) {
                float a = 0.f;
#pragma unroll 16
                for (int d = 0; d < AHD; ++d) a = fmaf(qs[d], L.X[key * ALD + d], a);
                s[j] = a * 0.125f;
            }
            mx = fmaxf(mx, s[j]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] = (lane + 64 * j < T) ? expf(s[j] - mx) : 0.f; sum += s[j]; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < T) ps[lane + 64 * j] = s[j] * inv;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int key = 0; key < T; ++key) o = fmaf(ps[key], L.Y[key * ALD + lane], o);
        ctx[((int64_t)b * T + q) * D + hd * AHD + lane] = o;
        if (lane == 0) lse[((int64_t)b * H + hd) * T + q] = mx + logf(sum);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                           const float* __restrict__ dctx, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int T, int H, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const AttnLds L = attn_lds(sm, T);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)b * T * ld + hd * AHD;
    const float* dob = dctx + (int64_t)b * T * D + hd * AHD;
    const float* ob = ctx + (int64_t)b * T * D + hd * AHD;
    const float* lrow = lse + ((int64_t)b * H + hd) * T;
    float* dbase = dqkv + (int64_t)b * T * ld + hd * AHD;
    float* va = L.vecA + w * 64;
    float* vb = L.vecB + w * 64;
    float* pa = L.pa + w * ATMAX;
    float* pb = L.pb + w * ATMAX;
    // ---- phase A: dQ (K, V in LDS; one query per wave step) ----
    attn_stage(L.X, base + D, ld, T, tid);
    attn_stage(L.Y, base + 2 * D, ld, T, tid);
    __syncthreads();
    for (int q = w; q < T; q += 4) {
        const float qd = base[(int64_t)q * ld + lane], dod = dob[(int64_t)q * D + lane];
        va[lane] = qd; vb[lane] = dod;
        const float delta = wave_sum(dod * ob[(int64_t)q * D + lane]);
        if (lane == 0) L.delta[q] = delta;
        const float lq = lrow[q];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = lane + 64 * j;
            if (key < T) {
                float a = 0.f, dp = 0.f;
#pragma unroll 16
                for (int d = 0; d < AHD; ++d) {
                    a = fmaf(va[d], L.X[key * ALD + d], a);
                    dp = fmaf(vb[d], L.Y[key * ALD + d], dp);
                }
                const float p = expf(a * 0.125f - lq);
                pa[key] = p * (dp - delta) * 0.125f;        // dS * scale
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float dq = 0.f;
        for (int key = 0; key < T; ++key) dq = fmaf(pa[key], L.X[key * ALD + lane], dq);
        dbase[(int64_t)q * ld + lane] = dq;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- phase B: dK, dV (Q, dO in LDS; one key per wave step) ----
    attn_stage(L.X, base, ld, T, tid);
    attn_stage(L.Y, dob, D, T, tid);
    __syncthreads();
    for (int key = w; key < T; key += 4) {
        va[lane] = base[(int64_t)key * ld + D + lane];          // k_j
        vb[lane] = base[(int64_t)key * ld + 2 * D + lane];      // v_j
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + 64 * j;
            if (q < T) {
                float a = 0.f, dp = 0.f;
#pragma unroll 16
                for (int d = 0; d < AHD; ++d) {
                    a = fmaf(L.X[q * ALD + d], va[d], a);
                    dp = fmaf(L.Y[q * ALD + d], vb[d], dp);
                }
                const float p = expf(a * 0.125f - lrow[q]);
                pa[q] = p;
                pb[q] = p * (dp - L.delta[q]) * 0.125f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float dv = 0.f, dk = 0.f;
        for (int q = 0; q < T; ++q) {
            dv = fmaf(pa[q], L.Y[q * ALD + lane], dv);
            dk = fmaf(pb[q], L.X[q * ALD + lane], dk);
        }
        dbase[(int64_t)key * ld + D + lane] = dk;
        dbase[(int64_t)key * ld + 2 * D + lane] = dv;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// The same attention on the exact-f32 MFMA (round 3; the scalar kernels above stay as the A/B reference, VITLORA_F32_ATTN=0).
// 16 x 16 tiles of v_mfma_f32_16x16x4_f32, D[m][n] with lane (n = lane & 15, g = lane >> 4) holding rows m = 4g + i.
// Operand fragments from LDS matrices X[row][ALD]:
//   frow(X, r0, k0): lane (r, kq) = X[r0 + r][k0 + kq]   -- the MFMA index is the matrix ROW, the k index its column
//   fcol(X, k0, c0): lane (c, kq) = X[k0 + kq][c0 + c]   -- the MFMA index is the matrix COLUMN, the k index its row
// A score tile is computed with the summed-next index on the ROWS (S^T[key][q] forward / phase A, S[q][key] phase B): an
// accumulator then holds, per lane, 4 consecutive k values of ONE column -- it becomes the B operand of the next product
// after a 4 x 4 exchange between the four lane groups (through a 1 KiB per-wave LDS tile).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float frow(const float* X, int r0, int k0, int lane) { return X[(r0 + (lane & 15)) * ALD + k0 + (lane >> 4)]; }
__device__ __forceinline__ float fcol(const float* X, int k0, int c0, int lane) { return X[(k0 + (lane >> 4)) * ALD + c0 + (lane & 15)]; }
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// accumulator tile (lane (n, g), rows 4g + i) -> tile scratch [16 rows][17]
__device__ __forceinline__ void acc_to_scratch(float* scr, const f32x4_t& v, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) scr[(4 * (lane >> 4) + i) * 17 + (lane & 15)] = v[i];
}
// B operand of k-step s4 (rows 4*s4 .. 4*s4+3 of the tile): lane (n, kq) = tile[4*s4 + kq][n]
__device__ __forceinline__ float scratch_b(const float* scr, int s4, int lane) { return scr[(4 * s4 + (lane >> 4)) * 17 + (lane & 15)]; }
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// rows [0, T) of a [T][64] operand (row stride ld) -> X[row][ALD]; rows [T, Tp) zero
__device__ __forceinline__ void attn_stage_pad(float* dst, const float* src, int ld, int T, int Tp, int tid) {
    for (int i = tid; i < Tp * 16; i += 256) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (r < T) v = *(const f32x4_t*)(src + (int64_t)r * ld + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[r * ALD + c4 + k] = v[k];
    }
}
size_t attn_mfma_lds_bytes(int T) {
    const int Tp = (T + 15) / 16 * 16;
    return ((size_t)2 * Tp * ALD + 2 * Tp + 4 * 2 * 16 * 17) * sizeof(float);
}

// forward: flash-style over 16-key tiles, one 16-query tile per wave step
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                                float* __restrict__ lse, int T, int H, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Tp = (T + 15) / 16 * 16, nt = Tp / 16;
    float* sK = sm;
    float* sV = sK + Tp * ALD;
    float* scr = sV + Tp * ALD + 2 * Tp + (threadIdx.x >> 6) * 2 * 16 * 17;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)b * T * ld + hd * AHD;
    attn_stage_pad(sK, base + D, ld, T, Tp, tid);
    attn_stage_pad(sV, base + 2 * D, ld, T, Tp, tid);
    __syncthreads();
    for (int qt = w; qt < nt; qt += 4) {
        const int q = qt * 16 + n, qc = q < T ? q : T - 1;
        float qf[16];                                  // B operand (n = query, k = feature): lane (n, kq) = Q[q][4s + kq]
#pragma unroll
        for (int s = 0; s < 16; ++s) qf[s] = base[(int64_t)qc * ld + 4 * s + g];
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;         // two chains: a dependent f32 MFMA waits 40 cycles, an independent one 32
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                s0 = mfma4(frow(sK, kt * 16, 4 * s, lane), qf[s], s0);               // S^T[key][q]
                s1 = mfma4(frow(sK, kt * 16, 4 * s + 4, lane), qf[s + 1], s1);
            }
            f32x4_t st = (s0 + s1) * 0.125f;
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kt * 16 + 4 * g + i >= T) st[i] = -INFINITY;
                mx = fmaxf(mx, st[i]);
            }
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);            // first tile: exp(-inf) = 0
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { st[i] = expf(st[i] - m_new); ps += st[i]; }
            l_run = l_run * alpha + group_sum(ps);
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
            acc_to_scratch(scr, st, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float pb = scratch_b(scr, s4, lane);                           // P[key 4*s4 + kq][q]
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = mfma4(fcol(sV, kt * 16 + 4 * s4, dt * 16, lane), pb, o[dt]);   // O^T[d][q]
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (q < T) {
            const float inv = 1.f / l_run;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(f32x4_t*)(ctx + ((int64_t)b * T + q) * D + hd * AHD + dt * 16 + 4 * g) = o[dt] * inv;
            if (g == 0) lse[((int64_t)b * H + hd) * T + q] = m_run + logf(l_run);
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                                const float* __restrict__ dctx, const float* __restrict__ lse,
                                                                float* __restrict__ dqkv, int T, int H, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Tp = (T + 15) / 16 * 16, nt = Tp / 16;
    float* sX = sm;
    float* sY = sX + Tp * ALD;
    float* sDelta = sY + Tp * ALD;      // [Tp]
    float* sLse = sDelta + Tp;          // [Tp], rows >= T: +inf (P = 0)
    float* scr = sLse + Tp + (threadIdx.x >> 6) * 2 * 16 * 17;
    float* scr2 = scr + 16 * 17;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const int ld = 3 * D;
    const float* base = qkv + (int64_t)b * T * ld + hd * AHD;
    const float* dob = dctx + (int64_t)b * T * D + hd * AHD;
    const float* ob = ctx + (int64_t)b * T * D + hd * AHD;
    const float* lrow = lse + ((int64_t)b * H + hd) * T;
    float* dbase = dqkv + (int64_t)b * T * ld + hd * AHD;
    // ---- phase A: dQ (K, V in LDS; one 16-query tile per wave step) ----
    attn_stage_pad(sX, base + D, ld, T, Tp, tid);
    attn_stage_pad(sY, base + 2 * D, ld, T, Tp, tid);
    for (int i = tid; i < Tp; i += 256) sLse[i] = i < T ? lrow[i] : INFINITY;
    __syncthreads();
    for (int qt = w; qt < nt; qt += 4) {
        const int q = qt * 16 + n, qc = q < T ? q : T - 1;
        float qf[16], dof[16];
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            qf[s] = base[(int64_t)qc * ld + 4 * s + g];
            dof[s] = dob[(int64_t)qc * D + 4 * s + g];
            dsum = fmaf(dof[s], ob[(int64_t)qc * D + 4 * s + g], dsum);
        }
        const float delta = group_sum(dsum);                   // rowsum(dO * O) of query q, the same in its four lanes
        if (g == 0) sDelta[q] = q < T ? delta : 0.f;
        const float lq = sLse[q];
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nt; ++kt) {
            f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dpt = st;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                st = mfma4(frow(sX, kt * 16, 4 * s, lane), qf[s], st);                // S^T[key][q]
                dpt = mfma4(frow(sY, kt * 16, 4 * s, lane), dof[s], dpt);             // dP^T[key][q]
            }
            f32x4_t ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float p = expf(st[i] * 0.125f - lq);
                if (kt * 16 + 4 * g + i >= T) p = 0.f;
                ds[i] = p * (dpt[i] - delta) * 0.125f;
            }
            acc_to_scratch(scr, ds, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float bd = scratch_b(scr, s4, lane);                            // dS[key 4*s4 + kq][q]
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma4(fcol(sX, kt * 16 + 4 * s4, dt * 16, lane), bd, dq[dt]);   // dQ^T[d][q]
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (q < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4_t*)(dbase + (int64_t)q * ld + dt * 16 + 4 * g) = dq[dt];
        }
    }
    __syncthreads();
    // ---- phase B: dK, dV (Q, dO in LDS; one 16-key tile per wave step) ----
    attn_stage_pad(sX, base, ld, T, Tp, tid);
    attn_stage_pad(sY, dob, D, T, Tp, tid);
    __syncthreads();
    for (int kt = w; kt < nt; kt += 4) {
        const int key = kt * 16 + n, kc = key < T ? key : T - 1;
        float kf[16], vf[16];                           // B operands (n = key, k = feature)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            kf[s] = base[(int64_t)kc * ld + D + 4 * s + g];
            vf[s] = base[(int64_t)kc * ld + 2 * D + 4 * s + g];
        }
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
        for (int qt = 0; qt < nt; ++qt) {
            f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = sc;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                sc = mfma4(frow(sX, qt * 16, 4 * s, lane), kf[s], sc);                // S[q][key]
                dp = mfma4(frow(sY, qt * 16, 4 * s, lane), vf[s], dp);                // dP[q][key]
            }
            f32x4_t p, ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qi = qt * 16 + 4 * g + i;
                p[i] = expf(sc[i] * 0.125f - sLse[qi]);                               // queries >= T: exp(-inf) = 0
                ds[i] = p[i] * (dp[i] - sDelta[qi]) * 0.125f;
            }
            acc_to_scratch(scr, p, lane);
            acc_to_scratch(scr2, ds, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float bp = scratch_b(scr, s4, lane), bd = scratch_b(scr2, s4, lane);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma4(fcol(sY, qt * 16 + 4 * s4, dt * 16, lane), bp, dv[dt]);   // dV^T[d][key]
                    dk[dt] = mfma4(fcol(sX, qt * 16 + 4 * s4, dt * 16, lane), bd, dk[dt]);   // dK^T[d][key]
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (key < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4_t*)(dbase + (int64_t)key * ld + D + dt * 16 + 4 * g) = dk[dt];
                *(f32x4_t*)(dbase + (int64_t)key * ld + 2 * D + dt * 16 + 4 * g) = dv[dt];
            }
        }
    }
}

inline int nblk(int64_t n, int t, int cap) {
    int64_t b = (n + t - 1) / t;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

int g_f32_big = 1;       // VITLORA_F32_BIG=0: every fp32 GEMM on the 64 x 64 tile kernel (A/B)
int g_f32_attn_mfma = 1; // VITLORA_F32_ATTN=0: the scalar fp32 attention kernels (A/B)

int f32_init(int device) {
    if (const char* e = getenv("VITLORA_F32_BIG")) g_f32_big = e[0] != '0';
    static bool done[64] = {};
    if (device < 0 || device >= 64) return -1;
    if (done[device]) return 0;
    const int bytes = (int)attn_lds_bytes(ATMAX);
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_fwd_f32_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_mfma_lds_bytes(ATMAX));
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_f32_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_mfma_lds_bytes(ATMAX));
    if (e != hipSuccess) return (int)e;
    if (const char* a = getenv("VITLORA_F32_ATTN")) g_f32_attn_mfma = a[0] != '0';
    done[device] = true;
    return 0;
}

void k_gemm_f32(const GemmF32& g, hipStream_t s) {
    const double bytes = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * (g.R ? 2.0 : 1.0));
    if (g_f32_big && g.M >= GBM && g.N >= GBN) {
        ProfScope prof_("gemm_f32_big_kernel", 2.0 * g.M * (double)g.N * g.K, bytes, s);
        dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM);
        hipLaunchKernelGGL(gemm_f32_big_kernel, grid, dim3(256), 0, s, g);
        return;
    }
    ProfScope prof_("gemm_f32_kernel", 2.0 * g.M * (double)g.N * g.K, bytes, s);
    dim3 grid((g.N + FBN - 1) / FBN, (g.M + FBM - 1) / FBM);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
}
void k_ln_fwd_f32(const float* x, float* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                  float eps, hipStream_t s) {
    const int nv = D / 4;
#define LNF(NV_, G_) hipLaunchKernelGGL((ln_fwd_f32_kernel<NV_, G_>), dim3((M + 4 * (64 / G_) - 1) / (4 * (64 / G_))), dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps)
    if (nv <= 16) LNF(1, 16); else if (nv <= 32) LNF(1, 32); else if (nv <= 64) LNF(1, 64); else if (nv <= 128) LNF(2, 64);
    else if (nv <= 256) LNF(4, 64); else LNF(8, 64);
#undef LNF
}
void k_ln_bwd_f32(const float* dh, const float* x, const float* mean, const float* rstd, const float* g, const float* dres,
                  float* dx, int M, int D, hipStream_t s) {
    const int nv = D / 4;
#define LNB(NV_, G_) hipLaunchKernelGGL((ln_bwd_f32_kernel<NV_, G_>), dim3((M + 4 * (64 / G_) - 1) / (4 * (64 / G_))), dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, M, D)
    if (nv <= 16) LNB(1, 16); else if (nv <= 32) LNB(1, 32); else if (nv <= 64) LNB(1, 64); else if (nv <= 128) LNB(2, 64);
    else if (nv <= 256) LNB(4, 64); else LNB(8, 64);
#undef LNB
}
void k_gelu_fwd_f32(const float* z, float* a, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_fwd_f32_kernel, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, z, a, n);
}
void k_gelu_bwd_f32(float* dz, const float* z, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, dz, z, n);
}
void k_mask_f32(float* dst, const float* src, int64_t n, int add, uint64_t seed, uint32_t stream, float p, hipStream_t s) {
    hipLaunchKernelGGL(mask_f32_kernel, dim3(nblk(n, 256, 8192)), dim3(256), 0, s, dst, src, n, add, seed, stream, p,
                       1.f / (1.f - p));
}
void k_patch_gather_f32(const float* x, float* out, int B, int S, int P, int normalise, const float* mean, const float* std,
                        hipStream_t s) {
    const int G = S / P;
    hipLaunchKernelGGL(patch_gather_f32_kernel, dim3(nblk((int64_t)B * G * G * 3 * P * P, 256, 8192)), dim3(256), 0, s, x,
                       out, B, S, P, G, normalise, mean[0], mean[1], mean[2], 1.f / std[0], 1.f / std[1], 1.f / std[2]);
}
void k_patch_scatter_f32(const float* dp, float* gx, int B, int S, int P, const float inv_std[3], hipStream_t s) {
    const int G = S / P;
    hipLaunchKernelGGL(patch_scatter_f32_kernel, dim3(nblk((int64_t)B * 3 * S * S, 256, 8192)), dim3(256), 0, s, dp, gx, B,
                       S, P, G, inv_std[0], inv_std[1], inv_std[2]);
}
int k_attn_fwd_f32(const float* qkv, float* ctx, float* lse, int B, int T, int H, int D, hipStream_t s) {
    if (T > ATMAX) return -1;
    if (g_f32_attn_mfma) {
        ProfScope prof_("attn_fwd_f32_mfma_kernel", 4.0 * B * H * (double)T * T * AHD, (double)B * T * D * 16.0, s);
        hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel, dim3(B * H), dim3(256), attn_mfma_lds_bytes(T), s, qkv, ctx, lse, T, H, D);
        return 0;
    }
    ProfScope prof_("attn_fwd_f32_kernel", 4.0 * B * H * (double)T * T * AHD, 0.0, s);
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(B * H), dim3(256), attn_lds_bytes(T), s, qkv, ctx, lse, T, H, D);
    return 0;
}
int k_attn_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dqkv, int B, int T, int H,
                   int D, hipStream_t s) {
    if (T > ATMAX) return -1;
    if (g_f32_attn_mfma) {
        ProfScope prof_("attn_bwd_f32_mfma_kernel", 10.0 * B * H * (double)T * T * AHD, (double)B * T * D * 32.0, s);
        hipLaunchKernelGGL(attn_bwd_f32_mfma_kernel, dim3(B * H), dim3(256), attn_mfma_lds_bytes(T), s, qkv, ctx, dctx, lse, dqkv, T, H, D);
        return 0;
    }
    ProfScope prof_("attn_bwd_f32_kernel", 10.0 * B * H * (double)T * T * AHD, 0.0, s);
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(B * H), dim3(256), attn_lds_bytes(T), s, qkv, ctx, dctx, lse, dqkv, T, H, D);
    return 0;
}

}  // namespace VLNS
