// HBM-bound kernels of the path: patch gather (+normalise), LayerNorm fwd/bwd, the CLS
// head with cross-entropy, the fused FGSM/PGD update (K10), the PGD random start (K11),
// flat Adam (K12), the save_images quantiser and the weight packers.
// Every kernel moves 16 bytes per lane wherever the layout allows it.
#include "kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

// ---------------------------------------------------------------------------------
// K1+K2 gather: pixels [B,3,S,S] f32 -> patches [B*NP (padded), 3*P*P] h16, (x-mean)/std fused
// (whitebox_attacks.py:26 feeds (perturbed-mean)/std to the Conv2d patch projection).
// ---------------------------------------------------------------------------------
__global__ void patch_gather_kernel(const float* __restrict__ x, h16* __restrict__ out, int B, int S,
                                    int P, int G, int normalise, float m0, float m1, float m2,
                                    float is0, float is1, float is2) {
    const int K = 3 * P * P;
    const int64_t total = (int64_t)B * G * G * (K / 8);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % (K / 8));
        const int64_t m = t / (K / 8);
        const int col = c8 * 8;
        const int c = col / (P * P), rem = col - c * P * P;
        const int ph = rem / P, pw = rem - ph * P;
        const int b = (int)(m / (G * G)), pi = (int)(m - (int64_t)b * G * G);
        const int py = pi / G, px = pi - py * G;
        const float* src = x + (((int64_t)b * 3 + c) * S + py * P + ph) * S + px * P + pw;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        float mean = 0.f, is = 1.f;
        if (normalise) { mean = c == 0 ? m0 : (c == 1 ? m1 : m2); is = c == 0 ? is0 : (c == 1 ? is1 : is2); }
        h16x8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = f2h((v0[i] - mean) * is); o[4 + i] = f2h((v1[i] - mean) * is); }
        *(h16x8*)(out + m * K + col) = o;
    }
}

// x[b*T + 0][:] = cls + pos[0]   (ViTEmbeddings: cat(cls, patches) + position_embeddings)
template <typename XT>
__global__ void cls_rows_kernel(XT* __restrict__ x, const float* __restrict__ cls,
                                const float* __restrict__ pos, int B, int T, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    x[(int64_t)b * T * D + d] = (XT)(cls[d] + pos[d]);
}

// ---------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, two-pass statistics in registers.
// ---------------------------------------------------------------------------------
// LoRA down-projection of ONE row fused into the kernel that produces the row (saves the separate skinny GEMM's
// pass over the activation): out[j] = sum_c v[c] * P[j][c] for the 8*NG rows of P (h16 [>= 8*NG, D], zero rows
// past r * modules), written as a full 64-column h16 row (zeros past 8*NG).  v = this lane's 4*NV values of the
// row (the h16 values the GEMM this replaces would read), columns (lane + 64 i)*4 + k.
// Reduction: halving butterfly (4 + 2 + 1 exchanges leave lane L with column (L>>3)&7) + 3 xor steps.
template <int NV, int NG>
struct LoraDownP { h16x4 p[NG][NV][8]; };
// this lane's slice of P, fetched EARLY (before the row's own loads are consumed) so that its L2 latency hides
template <int NV, int NG>
__device__ __forceinline__ void lora_down_load(LoraDownP<NV, NG>& r, int nv, int lane, const h16* __restrict__ P, int D) {
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r.p[gq][i][j] = c < nv ? *(const h16x4*)(P + (size_t)(gq * 8 + j) * D + c * 4) : h16x4{0, 0, 0, 0};
        }
}
template <int NV, int NG>
__device__ __forceinline__ void lora_down_row(const h16x4 (&v)[NV], const LoraDownP<NV, NG>& r, int nv, int lane,
                                              h16* __restrict__ out_row) {
    typedef h16x2 h2;
    float outv = 0.f;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const h2 vlo = {v[i][0], v[i][1]}, vhi = {v[i][2], v[i][3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const h16x4 p = r.p[gq][i][j];
                    const h2 plo = {p[0], p[1]}, phi = {p[2], p[3]};
                    acc[j] = dot2_acc(vlo, plo, acc[j]);     // v_dot2c_f32_f16 / _bf16: no conversions
                    acc[j] = dot2_acc(vhi, phi, acc[j]);
                }
            }
        }
        const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
        float a4[4], a2[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = (b5 ? acc[4 + q] : acc[q]) + __shfl_xor(b5 ? acc[q] : acc[4 + q], 32, 64);
#pragma unroll
        for (int q = 0; q < 2; ++q) a2[q] = (b4 ? a4[2 + q] : a4[q]) + __shfl_xor(b4 ? a4[q] : a4[2 + q], 16, 64);
        float a1 = (b3 ? a2[1] : a2[0]) + __shfl_xor(b3 ? a2[0] : a2[1], 8, 64);
        a1 += __shfl_xor(a1, 4, 64);
        a1 += __shfl_xor(a1, 2, 64);
        a1 += __shfl_xor(a1, 1, 64);
        const float routed = __shfl(a1, (lane & 7) << 3, 64);       // column j sits in lanes 8j .. 8j+7
        if ((lane >> 3) == gq) outv = routed;
    }
    out_row[lane] = f2h(lane < 8 * NG ? outv : 0.f);
}

template <int NV, int NG>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, h16* __restrict__ h,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int M, int D, float eps, const h16* __restrict__ delta,
                                                            float* __restrict__ xout, const h16* __restrict__ P,
                                                            h16* __restrict__ t, int ldh) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    LoraDownP<NV, NG ? NG : 1> pr;                  // fused t = h Ad^T of the projection that reads h next (see layernorm_bwd_kernel)
    if constexpr (NG > 0) lora_down_load<NV, NG>(pr, nv, lane, P, D);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        const float* xr = x + (int64_t)row * D;
        f32x4 v[NV], gam[NV], bet[NV];
        float s = 0.f;
        if (h) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {          // requested with the row, not after its two reductions
                const int c = lane + i * 64;
                gam[i] = c < nv ? *(const f32x4*)(gamma + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                bet[i] = c < nv ? *(const f32x4*)(beta + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nv ? *(const f32x4*)(xr + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (delta && c < nv) {
                // residual add fused in: x_out = x + delta (the h16 output of the projection before it)
                const h16x4 dl = *(const h16x4*)(delta + (int64_t)row * ldh + c * 4);   // ldh: row stride of the h16 operands (>= D: padded)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] += h2f(dl[k]);
                *(f32x4*)(xout + (int64_t)row * D + c * 4) = v[i];
            }
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        if (!h) continue;                            // add only (last layer: the head normalises the CLS rows)
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        h16* hr = h + (int64_t)row * ldh;
        h16x4 vb[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const f32x4 g = gam[i], b = bet[i];
                h16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = f2h((v[i][k] - mean) * rstd * g[k] + b[k]);
                *(h16x4*)(hr + c * 4) = o;
                vb[i] = o;
            }
        }
        if constexpr (NG > 0) lora_down_row<NV, NG>(vb, pr, nv, lane, t + (int64_t)row * 64);
    }
}

// LayerNorm backward fused with the residual-gradient add:
//   dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dh * gamma
// writes dx as f32 (residual-gradient stream) and as h16 (A operand of the next dgrad GEMM).
template <int NV, int NG, bool PAD>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const h16* __restrict__ dh, const float* __restrict__ x,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, const float* __restrict__ dres,
                                                            float* __restrict__ dx, h16* __restrict__ dx_h, int M, int D,
                                                            const h16* __restrict__ P, h16* __restrict__ u, int* __restrict__ err, int ldh) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    bool sat = false;          // a gradient that left the fp16 range (clamped by f2h_sat below) or is NaN: reported, never silent
    // with the fused projection a wave keeps its slice of P in registers and walks rows (grid-stride): P is read
    // once per wave, not once per row (per row it would double the kernel's L1 requests)
    LoraDownP<NV, NG ? NG : 1> pr;
    if constexpr (NG > 0) lora_down_load<NV, NG>(pr, nv, lane, P, D);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    // fp32 rows / h16 rows; PAD: the h16 rows have their own (padded) stride -- a second 64-bit offset, which costs the
    // D = 768 fused form its fourth wave per SIMD (132 VGPRs), so the unpadded form shares one
    const int64_t off = (int64_t)row * D, offh = PAD ? (int64_t)row * ldh : off;
    f32x4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        xh[i] = g[i];
        if (c < nv) {
            const h16x4 d = *(const h16x4*)(dh + offh + c * 4);
            const f32x4 xv = *(const f32x4*)(x + off + c * 4);
            const f32x4 gm = *(const f32x4*)(gamma + c * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[i][k] = h2f(d[k]) * gm[k];
                xh[i][k] = (xv[k] - mean) * rstd;
                s1 += g[i][k];
                s2 += g[i][k] * xh[i][k];
            }
        }
    }
    const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
    h16x4 vb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 r = *(const f32x4*)(dres + off + c * 4);     // (requesting it with the row's first loads costs a wave per SIMD: +30 %)
            f32x4 o; h16x4 ob;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = r[k] + rstd * (g[i][k] - c1 - xh[i][k] * c2);
                ob[k] = f2h_sat(o[k]);
                sat |= !(fabsf(o[k]) <= H16_MAX);
            }
            // dx == nullptr: nobody reads the fp32 stream below (LN1 of layer 0, which never carries a fused projection;
            // the test in the fused forms would cost them two VGPRs and with that their fourth wave per SIMD)
            if (NG > 0 || dx) *(f32x4*)(dx + off + c * 4) = o;
            *(h16x4*)(dx_h + offh + c * 4) = ob;
            vb[i] = ob;                                              // the row as the next dgrad GEMM reads it
        }
    }
    // u = dx_h B of the projection whose dgrad consumes dx_h next (linear_dgrad skips its down GEMM)
    if constexpr (NG > 0) lora_down_row<NV, NG>(vb, pr, nv, lane, u + (int64_t)row * 64);
    }
    if (sat && err) *err = 2;
}

// ---------------------------------------------------------------------------------
// Narrow rows (D <= 128: Swin stage 1 has D = 96): 32 lanes per row, two rows per wave -- with one wave per row 40 of
// the 64 lanes idle and the kernel runs at half the HBM rate.  Same arithmetic as the kernels above, no LoRA fusion.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void layernorm_fwd_small_kernel(const float* __restrict__ x, h16* __restrict__ h,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int M, int D, float eps, const h16* __restrict__ delta,
                                                                  float* __restrict__ xout, int ldh) {
    const int lane = threadIdx.x & 63, li = lane & 31;
    const int nv = D >> 2;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = row < M;
    const int rr = live ? row : M - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (li < nv) {
        v = *(const f32x4*)(x + (int64_t)rr * D + li * 4);
        if (delta) {
            const h16x4 dl = *(const h16x4*)(delta + (int64_t)rr * ldh + li * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += h2f(dl[k]);
            if (live) *(f32x4*)(xout + (int64_t)rr * D + li * 4) = v;
        }
    }
    if (!h) return;
    const float mean = half_wave_sum(v[0] + v[1] + v[2] + v[3]) / D;
    float q = 0.f;
    if (li < nv) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = v[k] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(half_wave_sum(q) / D + eps);
    if (!live) return;
    if (li == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    if (li < nv) {
        const f32x4 g = *(const f32x4*)(gamma + li * 4), b = *(const f32x4*)(beta + li * 4);
        h16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f2h((v[k] - mean) * rstd * g[k] + b[k]);
        *(h16x4*)(h + (int64_t)row * ldh + li * 4) = o;
    }
}
__global__ __launch_bounds__(256) void layernorm_bwd_small_kernel(const h16* __restrict__ dh, const float* __restrict__ x,
                                                                  const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                  const float* __restrict__ gamma, const float* __restrict__ dres,
                                                                  float* __restrict__ dx, h16* __restrict__ dx_h, int M, int D,
                                                                  int* __restrict__ err, int ldh) {
    const int lane = threadIdx.x & 63, li = lane & 31;
    const int nv = D >> 2;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = row < M;
    const int rr = live ? row : M - 1;
    const float mean = mean_in[rr], rstd = rstd_in[rr];
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, xh = g;
    float s1 = 0.f, s2 = 0.f;
    if (li < nv) {
        const h16x4 d = *(const h16x4*)(dh + (int64_t)rr * ldh + li * 4);
        const f32x4 xv = *(const f32x4*)(x + (int64_t)rr * D + li * 4);
        const f32x4 gm = *(const f32x4*)(gamma + li * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = h2f(d[k]) * gm[k];
            xh[k] = (xv[k] - mean) * rstd;
            s1 += g[k];
            s2 += g[k] * xh[k];
        }
    }
    const float c1 = half_wave_sum(s1) / D, c2 = half_wave_sum(s2) / D;
    if (!live || li >= nv) return;
    const f32x4 r = *(const f32x4*)(dres + (int64_t)row * D + li * 4);
    f32x4 o; h16x4 ob;
    bool sat = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[k] = r[k] + rstd * (g[k] - c1 - xh[k] * c2);
        ob[k] = f2h_sat(o[k]);
        sat |= !(fabsf(o[k]) <= H16_MAX);
    }
    if (dx) *(f32x4*)(dx + (int64_t)row * D + li * 4) = o;
    *(h16x4*)(dx_h + (int64_t)row * ldh + li * 4) = ob;
    if (sat && err) *err = 2;
}

// ---------------------------------------------------------------------------------
// 16-bit residual streams (round 4; the ViT path's 16-bit precision).  The residual stream x and the residual-gradient
// stream are stored as h16 -- HF runs this model with an fp16 residual stream -- so a LayerNorm pass moves 8 B per
// element instead of 12 (forward) / 16 (backward): forward reads x, delta and writes x' = round16(x + delta) and h;
// backward reads dh, x', the incoming stream gradient and writes the outgoing one IN PLACE, which is also the A operand
// of the next dgrad GEMM (the fp32 stream had a separate h16 shadow for that).  Statistics and all arithmetic stay fp32;
// the forward normalises the ROUNDED x', so the backward (which re-derives xhat from the saved x') differentiates exactly
// the function the forward computed.  tools/error_budget_streams.py prices the two roundings on ViT-B: logits 6.7e-4 ->
// 1.0e-3, input gradient 1.4e-3 -> 1.7e-3 against the fp32 reference (north_star: 1e-2).
// Layout: 32 lanes per row (two rows per wave), 16-byte chunk c = li + 32 i of the row per lane: every access is a
// global_load/store_dwordx4 (8-byte accesses reach 0.54-0.70 of that rate, MI355X_MICROARCH.md).
// Fused LoRA down projection (NG > 0): out[j] = sum_k v[k] P[j][k] for the 8 NG rows of P, P staged ONCE per workgroup in
// LDS (in registers it would cost this layout 96 VGPRs per group), blocks walk the rows grid-stride.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float dot8(h16x8 a, h16x8 b, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const h16x2 x = {a[2 * k], a[2 * k + 1]}, y = {b[2 * k], b[2 * k + 1]};
        acc = dot2_acc(x, y, acc);
    }
    return acc;
}
template <int NV, int NG>
__device__ __forceinline__ void lora_down_row16(const h16x8 (&v)[NV], const h16* sP, int D, int nc, int li, int half,
                                                h16* __restrict__ out_row, bool live) {
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = li + i * 32;
            if (c < nc) {
                // all eight rows' chunks requested before the first is used (the compiler otherwise keeps two LDS reads in
                // flight and exposes a round trip per pair)
                h16x8 pj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pj[j] = *(const h16x8*)(sP + (size_t)(gq * 8 + j) * D + c * 8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = dot8(v[i], pj[j], acc[j]);
            }
        }
        // halving butterfly inside the half wave: 4 + 2 + 1 exchanges leave lane li with column (li >> 2) & 7, two xor steps finish
        const bool b4 = li & 16, b3 = li & 8, b2 = li & 4;
        float a4[4], a2[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = (b4 ? acc[4 + q] : acc[q]) + __shfl_xor(b4 ? acc[q] : acc[4 + q], 16, 64);
#pragma unroll
        for (int q = 0; q < 2; ++q) a2[q] = (b3 ? a4[2 + q] : a4[q]) + __shfl_xor(b3 ? a4[q] : a4[2 + q], 8, 64);
        float a1 = (b2 ? a2[1] : a2[0]) + __shfl_xor(b2 ? a2[0] : a2[1], 4, 64);
        a1 += __shfl_xor(a1, 2, 64);
        a1 += __shfl_xor(a1, 1, 64);
        // lane li stores columns 2 li, 2 li + 1 of the 64-column row: group li >> 2, columns 2 (li & 3) and the next one
        const int src = half * 32 + ((li & 3) << 3);
        const float v0 = __shfl(a1, src, 64), v1 = __shfl(a1, src + 4, 64);
        if ((li >> 2) == gq) { o0 = v0; o1 = v1; }
    }
    if (live) *(h16x2*)(out_row + 2 * li) = h16x2{f2h(o0), f2h(o1)};
}

// Streamed 16-byte loads of the 16-bit LayerNorm kernels (every byte of x / dh / the gradient stream is used once per pass).
// VL_LN_NT: 1 = non-temporal loads (round 5: K10 gained 20 % from them, tools/k10_sweep.hip), 0 = plain loads.
#ifndef VL_LN_NT
#define VL_LN_NT 1
#endif
__device__ __forceinline__ h16x8 ld_stream8(const h16* p) {
#if VL_LN_NT
    return __builtin_nontemporal_load((const h16x8*)p);
#else
    return *(const h16x8*)p;
#endif
}

template <int NV, int NG>
__global__ __launch_bounds__(256) void layernorm_fwd16_kernel(const h16* __restrict__ x, h16* __restrict__ h,
                                                              float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int M, int D, float eps, const h16* __restrict__ delta,
                                                              h16* __restrict__ xout, const h16* __restrict__ P,
                                                              h16* __restrict__ t, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    h16* sP = (h16*)ln_smem;
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int nc = D >> 3;
    if constexpr (NG > 0) {
        for (int i = threadIdx.x; i < 8 * NG * nc; i += 256) ((h16x8*)sP)[i] = ((const h16x8*)P)[i];
        __syncthreads();
    }
    bool sat = false;
    const int pairs = (M + 1) >> 1;
    for (int pr = blockIdx.x * 4 + (threadIdx.x >> 6); pr < pairs; pr += gridDim.x * 4) {
        const int row = pr * 2 + half;
        const bool live = row < M;
        const int64_t off = (int64_t)(live ? row : M - 1) * D;
        f32x4 v[NV][2];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = li + i * 32;
            v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nc) {
                const h16x8 xv = ld_stream8(x + off + c * 8);
                if (delta) {
                    // residual add fused in: x' = round16(x + delta) (delta = the h16 output of the projection before it)
                    const h16x8 dl = ld_stream8(delta + off + c * 8);
                    h16x8 xr;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float a = h2f(xv[k]) + h2f(dl[k]);
                        sat |= !(fabsf(a) <= H16_MAX);
                        xr[k] = f2h(a);
                        v[i][k >> 2][k & 3] = h2f(xr[k]);
                    }
                    if (live) *(h16x8*)(xout + off + c * 8) = xr;
                } else {
                    // (the add happened in the o / fc2 GEMM epilogue: an out-of-range sum arrives here as inf)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[i][k >> 2][k & 3] = h2f(xv[k]); sat |= !(fabsf(h2f(xv[k])) <= H16_MAX); }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) s += v[i][0][k] + v[i][1][k];
            }
        }
        if (!h) continue;                            // add only (the head normalises the CLS rows)
        const float mean = half_wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (li + i * 32 < nc) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d0 = v[i][0][k] - mean, d1 = v[i][1][k] - mean;
                    q += d0 * d0 + d1 * d1;
                }
            }
        }
        const float rstd = rsqrtf(half_wave_sum(q) / D + eps);
        if (li == 0 && live) { mean_out[row] = mean; rstd_out[row] = rstd; }
        h16x8 vb[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = li + i * 32;
            vb[i] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nc) {
                const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
                const f32x4 b0 = *(const f32x4*)(beta + c * 8), b1 = *(const f32x4*)(beta + c * 8 + 4);
                h16x8 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = f2h((v[i][0][k] - mean) * rstd * g0[k] + b0[k]);
                    o[4 + k] = f2h((v[i][1][k] - mean) * rstd * g1[k] + b1[k]);
                }
                if (live) *(h16x8*)(h + off + c * 8) = o;
                vb[i] = o;
            }
        }
        if constexpr (NG > 0) lora_down_row16<NV, NG>(vb, sP, D, nc, li, half, t + (int64_t)(live ? row : 0) * 64, live);
    }
    if (sat && err) *err = 4;      // a FORWARD overflow of the 16-bit residual stream (its own message: round-4 ADVICE)
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dh * gamma, xhat = (x - mean) * rstd; `dres` is read and
// overwritten in place (each row by the half wave that read it)
template <int NV, int NG>
__global__ __launch_bounds__(256) void layernorm_bwd16_kernel(const h16* __restrict__ dh, const h16* __restrict__ x,
                                                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                              const float* __restrict__ gamma, h16* dres, int M, int D,
                                                              const h16* __restrict__ P, h16* __restrict__ u, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    h16* sP = (h16*)ln_smem;
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int nc = D >> 3;
    if constexpr (NG > 0) {
        for (int i = threadIdx.x; i < 8 * NG * nc; i += 256) ((h16x8*)sP)[i] = ((const h16x8*)P)[i];
        __syncthreads();
    }
    bool sat = false;          // a gradient that left the fp16 range (clamped by f2h_sat below) or is NaN: reported, never silent
    const int pairs = (M + 1) >> 1;
    for (int pr = blockIdx.x * 4 + (threadIdx.x >> 6); pr < pairs; pr += gridDim.x * 4) {
        const int row = pr * 2 + half;
        const bool live = row < M;
        const int rr = live ? row : M - 1;
        const int64_t off = (int64_t)rr * D;
        const float mean = mean_in[rr], rstd = rstd_in[rr];
        h16x8 rv[NV];
        f32x4 g[NV][2], xh[NV][2];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = li + i * 32;
            g[i][0] = g[i][1] = xh[i][0] = xh[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nc) {
                const h16x8 d = ld_stream8(dh + off + c * 8);
                const h16x8 xv = ld_stream8(x + off + c * 8);
                rv[i] = ld_stream8(dres + off + c * 8);
                const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g[i][0][k] = h2f(d[k]) * g0[k];
                    g[i][1][k] = h2f(d[4 + k]) * g1[k];
                    xh[i][0][k] = (h2f(xv[k]) - mean) * rstd;
                    xh[i][1][k] = (h2f(xv[4 + k]) - mean) * rstd;
                    s1 += g[i][0][k] + g[i][1][k];
                    s2 += g[i][0][k] * xh[i][0][k] + g[i][1][k] * xh[i][1][k];
                }
            }
        }
        const float c1 = half_wave_sum(s1) / D, c2 = half_wave_sum(s2) / D;
        h16x8 vb[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = li + i * 32;
            vb[i] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (c < nc) {
                h16x8 ob;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float o = h2f(rv[i][k]) + rstd * (g[i][k >> 2][k & 3] - c1 - xh[i][k >> 2][k & 3] * c2);
                    ob[k] = f2h_sat(o);
                    sat |= !(fabsf(o) <= H16_MAX);
                }
                if (live) *(h16x8*)(dres + off + c * 8) = ob;
                vb[i] = ob;                                              // the row as the next dgrad GEMM reads it
            }
        }
        // u = dx B of the projection whose dgrad consumes dx next (linear_dgrad skips its down GEMM)
        if constexpr (NG > 0) lora_down_row16<NV, NG>(vb, sP, D, nc, li, half, u + (int64_t)rr * 64, live);
    }
    if (sat && err) *err = 2;
}

// ---------------------------------------------------------------------------------
// CLS head: final LayerNorm on token 0 + classifier (fp32) -- modeling_vit.py:385,560-561.
// one block (256 threads) per image.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename XT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const XT* __restrict__ x, int T, int D, int C, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ Wc, const float* __restrict__ bc,
                                                       float* __restrict__ xhat_out, float* __restrict__ xf_out,
                                                       float* __restrict__ rstd_out, float* __restrict__ logits) {
    extern __shared__ float sm[];      // [D] normalised+affine CLS row, then 4 floats of reduction scratch
    float* xf = sm;
    float* red = sm + D;
    const int b = blockIdx.x;
    const XT* xr = x + (int64_t)b * T * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += (float)xr[d];
    const float mean = block_sum256(s, red) / D;
    float q = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float t = (float)xr[d] - mean; q += t * t; }
    const float rstd = rsqrtf(block_sum256(q, red) / D + eps);
    for (int d = threadIdx.x; d < D; d += 256) {
        const float xh = ((float)xr[d] - mean) * rstd;
        const float v = xh * gamma[d] + beta[d];
        xf[d] = v;
        xhat_out[(int64_t)b * D + d] = xh;
        xf_out[(int64_t)b * D + d] = v;
    }
    if (threadIdx.x == 0) rstd_out[b] = rstd;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = w; c < C; c += 4) {
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a += xf[d] * Wc[(int64_t)c * D + d];
        a = wave_sum(a);
        if (lane == 0) logits[(int64_t)b * C + c] = a + bc[c];
    }
}

// F.cross_entropy(logits, labels), mean reduction; dlogits = (softmax - onehot)/B.
// one wave per image (4 per block) writes the per-image loss; a second tiny launch sums them in a
// fixed order (bitwise reproducible loss).
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      int B, int C, float* __restrict__ dlogits, float* __restrict__ loss_img,
                                                      int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t yl = labels[b];
    if (yl < 0 || yl >= C) {               // refuse loudly: NaN loss for this image + the error word the next API call reports
        for (int c = lane; c < C; c += 64) dlogits[(int64_t)b * C + c] = __builtin_nanf("");
        if (lane == 0) { loss_img[b] = __builtin_nanf(""); if (err) *err = 1; }
        return;
    }
    const float* lr = logits + (int64_t)b * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(lr[c] - mx);
    se = wave_sum(se);
    const int y = (int)yl;
    const float lse = mx + logf(se);
    for (int c = lane; c < C; c += 64)
        dlogits[(int64_t)b * C + c] = (expf(lr[c] - lse) - (c == y ? 1.f : 0.f)) / B;
    if (lane == 0) loss_img[b] = lse - lr[y];
}
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_img, int B, float* __restrict__ loss_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += loss_img[b];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss_out = (red[0] + red[1] + red[2] + red[3]) / B;
}

// head backward: d(xf) = dlogits * Wc ; LN backward on the CLS row; writes the CLS row of the
// residual-gradient stream (all other rows of that stream are zero: memset by the driver).
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ gscale,
                                                       const float* __restrict__ Wc,
                                                       const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                       const float* __restrict__ rstd_in, int T, int D, int C,
                                                       float* __restrict__ dx, h16* __restrict__ dx_h) {
    extern __shared__ float sm[];
    float* g = sm;           // [D]
    float* red = sm + D;
    const int b = blockIdx.x;
    const float gs = gscale ? gscale[b] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += dlogits[(int64_t)b * C + c] * Wc[(int64_t)c * D + d];
        a *= gamma[d] * gs;
        g[d] = a;
        s1 += a;
        s2 += a * xhat[(int64_t)b * D + d];
    }
    const float c1 = block_sum256(s1, red) / D;
    const float c2 = block_sum256(s2, red) / D;
    const float rstd = rstd_in[b];
    for (int d = threadIdx.x; d < D; d += 256) {
        const float o = rstd * (g[d] - c1 - xhat[(int64_t)b * D + d] * c2);
        if (dx) dx[(int64_t)b * T * D + d] = o;
        if (dx_h) dx_h[(int64_t)b * T * D + d] = f2h_sat(o);
    }
}

// classifier gradient (train): dW[c][d] = sum_b dlogits[b][c] * xf[b][d]; db[c] = sum_b dlogits[b][c]
__global__ void classifier_grad_kernel(const float* __restrict__ dlogits, const float* __restrict__ xf, int B, int D,
                                       int C, float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * D) {
        const int c = i / D, d = i - c * D;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dlogits[(int64_t)b * C + c] * xf[(int64_t)b * D + d];
        dW[i] = a;
    }
    if (i < C) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dlogits[(int64_t)b * C + i];
        db[i] = a;
    }
}

// ---------------------------------------------------------------------------------
// K10: fused input-grad -> sign -> eps-project -> clamp.  16 B/elem algorithmic traffic
// (reads g, adv, x0; writes adv).  whitebox_attacks.py:32-36 (FGSM) / torchattacks PGD step.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float g) { return (g > 0.f) ? 1.f : ((g < 0.f) ? -1.f : 0.f); }

// Access shape (round 5, tools/k10_sweep.hip on MI355X, 616.6 MB per launch): the three input streams are read with
// NON-TEMPORAL 16-byte loads, two vectors per thread in flight, plain stores: 92 us = 6.7 TB/s = 0.84 of the 8 TB/s peak
// (plain loads, rounds 1-4: 112-119 us = 0.65-0.69; a float4 copy reaches 5.5 TB/s on the same box; non-temporal STORES lose
// the gain again: 106 us).  Every byte is used once, so nothing is lost by not keeping the lines.
__global__ __launch_bounds__(256) void pgd_step_kernel(float* __restrict__ adv, const float* __restrict__ x0,
                                                       const float* __restrict__ grad, float eps, float alpha,
                                                       float lo, float hi, int64_t n4, int64_t n, int* __restrict__ err) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const int64_t j = i + stride;
        const bool two = j < n4;
        const f32x4 a0 = __builtin_nontemporal_load((const f32x4*)adv + i);
        const f32x4 x0v = __builtin_nontemporal_load((const f32x4*)x0 + i);
        const f32x4 g0 = __builtin_nontemporal_load((const f32x4*)grad + i);
        f32x4 a1 = a0, x1v = x0v, g1 = g0;
        if (two) {
            a1 = __builtin_nontemporal_load((const f32x4*)adv + j);
            x1v = __builtin_nontemporal_load((const f32x4*)x0 + j);
            g1 = __builtin_nontemporal_load((const f32x4*)grad + j);
        }
        bool bad = false;
        f32x4 o0, o1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bad |= !(fabsf(g0[k]) < INFINITY) | !(fabsf(g1[k]) < INFINITY);
            const float t0 = a0[k] + alpha * sgn(g0[k]);
            const float d0 = fminf(fmaxf(t0 - x0v[k], -eps), eps);
            o0[k] = fminf(fmaxf(x0v[k] + d0, lo), hi);
            const float t1 = a1[k] + alpha * sgn(g1[k]);
            const float d1 = fminf(fmaxf(t1 - x1v[k], -eps), eps);
            o1[k] = fminf(fmaxf(x1v[k] + d1, lo), hi);
        }
        if (err && bad) *err = 2;
        *((f32x4*)adv + i) = o0;
        if (two) *((f32x4*)adv + j) = o1;
    }
    // tail (n not a multiple of 4)
    const int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const float t = adv[i] + alpha * sgn(grad[i]);
        const float d = fminf(fmaxf(t - x0[i], -eps), eps);
        adv[i] = fminf(fmaxf(x0[i] + d, lo), hi);
    }
}

// Zero fill (bytes a multiple of 16, 16-byte aligned).  Used INSIDE captured sequences instead of hipMemsetAsync: a memset
// node captured before the runtime's own fill kernel had ever run did not take effect on graph replays (ROCm 7.2;
// tools/cold_capture_diag.py), a kernel node of the library's own code object does.
__global__ __launch_bounds__(256) void zero_kernel(f32x4* __restrict__ p, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = z;
}

// K11: counter-based uniform noise (splitmix64 finaliser on (seed, index)); not torch's stream
// (random_start=True, whitebox_attacks.py:113, needs a seeded deterministic start, not that stream).
// (mix64 lives in common.h: the LoRA dropout mask uses the same generator)
__global__ void pgd_init_kernel(float* __restrict__ adv, const float* __restrict__ x0, float eps, float lo, float hi,
                                uint64_t seed, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = mix64(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);       // [0,1)
        const float v = x0[i] + (2.f * u - 1.f) * eps;
        adv[i] = fminf(fmaxf(v, lo), hi);
    }
}

__global__ void fill_random_h16_kernel(h16* __restrict__ dst, size_t n, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = mix64(seed * 0xD1342543DE82EF95ull + i);
        dst[i] = f2h((float)(r >> 40) * (2.0f / 16777216.0f) - 1.0f);
    }
}

// K12: torch.optim.Adam single flat update (train_loras.py:284,315)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1,
                            float sqrt_bc2, int64_t n, int* __restrict__ err) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    if (!(fabsf(gi) < INFINITY)) {           // non-finite gradient: leave the parameter and its moments alone, report it
        if (err) *err = 3;
        return;
    }
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// save_images (Utils.py:108-112): clamp(0,1) -> *255 -> uint8 truncation, CHW -> HWC
__global__ void quantize_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int B, int Cn, int H, int W) {
    const int64_t n = (int64_t)B * Cn * H * W;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // i indexes the OUTPUT (b, y, x, c)
    const int c = (int)(i % Cn);
    int64_t r = i / Cn;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    float v = img[(((int64_t)b * Cn + c) * H + yy) * W + xx];
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = (uint8_t)(v * 255.f);
}

__global__ void channel_affine_kernel(float* __restrict__ dst, const float* __restrict__ src, float s0, float s1,
                                      float s2, float t0, float t1, float t2, int64_t hw, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / hw) % 3);
        const float sc = c == 0 ? s0 : (c == 1 ? s1 : s2), sh = c == 0 ? t0 : (c == 1 ? t1 : t2);
        dst[i] = src[i] * sc + sh;
    }
}

// ---------------------------------------------------------------------------------
// packers
// ---------------------------------------------------------------------------------
// dst[r*ldd + coff + c] = h16(scale * src[r*cols + c])
__global__ void pack_h16_kernel(const float* __restrict__ src, h16* __restrict__ dst, int rows, int cols, int ldd,
                                 int coff, float scale) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[(int64_t)r * ldd + coff + c] = f2h(scale * src[i]);
}
// dst[c*ldd + roff + r] = h16(scale * src[r*cols + c])   (transpose)
__global__ void pack_h16_t_kernel(const float* __restrict__ src, h16* __restrict__ dst, int rows, int cols, int ldd,
                                   int roff, float scale) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = by + k, c = bx + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = bx + k, r = by + tx;
        if (r < rows && c < cols) dst[(int64_t)c * ldd + roff + r] = f2h(scale * tile[tx][k]);
    }
}
// W'[o][k] = h16(W[o][k] + s * sum_j B[o][j] A[j][k])  (merge_and_unload), optionally transposed
__global__ void merge_lora_kernel(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ Bm,
                                  int out, int in, int r, float s, h16* __restrict__ dst, int ldd, int roff,
                                  h16* __restrict__ dstT, int lddT, int coffT) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)out * in) return;
    const int o = (int)(i / in), k = (int)(i - (int64_t)o * in);
    float a = 0.f;
    for (int j = 0; j < r; ++j) a += Bm[(int64_t)o * r + j] * A[(int64_t)j * in + k];
    const h16 v = f2h(W[i] + s * a);
    dst[(int64_t)(roff + o) * ldd + k] = v;
    dstT[(int64_t)k * lddT + coffT + o] = v;
}
// fp32 form of the same fold (adapter composition: the merged weight becomes the next base)
__global__ void merge_f32_kernel(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ Bm,
                                 int out, int in, int r, float s, float* __restrict__ dst) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)out * in) return;
    const int o = (int)(i / in), k = (int)(i - (int64_t)o * in);
    float a = 0.f;
    for (int j = 0; j < r; ++j) a += Bm[(int64_t)o * r + j] * A[(int64_t)j * in + k];
    dst[i] = W[i] + s * a;
}

}  // namespace

void k_merge_f32(const float* W, const float* A, const float* B, int out, int in, int r, float sc, float* dst,
                 hipStream_t s) {
    const int64_t n = (int64_t)out * in;
    hipLaunchKernelGGL(merge_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, A, B, out, in, r, sc, dst);
}

// ---------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------
static inline int nblk(int64_t n, int t, int cap = 1 << 20) {
    int64_t b = (n + t - 1) / t;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

void k_patch_gather(const float* x, h16* out, int B, int S, int P, int normalise, const float* mean,
                    const float* std, hipStream_t s) {
    ProfScope prof_("patch_gather_kernel", 0.0, (double)B * 3 * S * S * 6.0, s);
    const int G = S / P;
    const int64_t total = (int64_t)B * G * G * (3 * P * P / 8);
    hipLaunchKernelGGL(patch_gather_kernel, dim3(nblk(total, 256, 8192)), dim3(256), 0, s, x, out, B, S, P, G,
                       normalise, mean[0], mean[1], mean[2], 1.f / std[0], 1.f / std[1], 1.f / std[2]);
}
void k_cls_rows(float* x, const float* cls, const float* pos, int B, int T, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel<float>, dim3(nblk((int64_t)B * D, 256)), dim3(256), 0, s, x, cls, pos, B, T, D);
}
void k_cls_rows16(h16* x, const float* cls, const float* pos, int B, int T, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel<h16>, dim3(nblk((int64_t)B * D, 256)), dim3(256), 0, s, x, cls, pos, B, T, D);
}
template <int NV>
static void launch_ln_fwd(dim3 grid, hipStream_t s, const float* x, h16* h, float* mean, float* rstd, const float* g,
                          const float* b, int M, int D, float eps, const h16* delta, float* xout, const h16* P, int ng, h16* t, int ldh) {
    if (ng == 1) hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 1>), grid, dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, ldh);
    else if (ng == 2) hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 2>), grid, dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, ldh);
    else if (ng == 3) hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 3>), grid, dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, ldh);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<NV, 0>), grid, dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, ldh);
}
void k_layernorm_fwd(const float* x, h16* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                     float eps, const h16* delta, float* xout, const h16* P, int ng, h16* t, hipStream_t s, int ldh) {
    if (ldh <= 0) ldh = D;
    if (D <= 128 && !(P && t && ng > 0)) {          // narrow rows: two per wave
        ProfScope prof_("layernorm_fwd_kernel", 0.0, (double)M * D * (delta ? (h ? 12.0 : 10.0) : 6.0), s);
        hipLaunchKernelGGL(layernorm_fwd_small_kernel, dim3((M + 7) / 8), dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, ldh);
        return;
    }
    ProfScope prof_("layernorm_fwd_kernel", 0.0, (double)M * D * (delta ? (h ? 12.0 : 10.0) : 6.0), s);
    const int nv = (D / 4 + 63) / 64;
    if (!P || !t || !h || ng < 0 || ng > 3) ng = 0;
    dim3 grid((M + 3) / 4);
    if (ng && grid.x > 1024) grid.x = 1024;          // resident blocks walk the rows, P stays in registers
    switch (nv) {
        case 1: launch_ln_fwd<1>(grid, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, ldh); break;
        case 2: launch_ln_fwd<2>(grid, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, ldh); break;
        case 3: launch_ln_fwd<3>(grid, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, ldh); break;
        default: launch_ln_fwd<4>(grid, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, ldh); break;
    }
}
template <int NV>
static void launch_ln_bwd(dim3 grid, hipStream_t s, const h16* dh, const float* x, const float* mean, const float* rstd,
                          const float* g, const float* dres, float* dx, h16* dx_h, int M, int D, const h16* P, int ng, h16* u, int* err, int ldh) {
    if (ng > 0 && (ldh != D || !dx)) { fprintf(stderr, "vitlora: LayerNorm backward with a fused LoRA projection needs unpadded h16 rows and an fp32 output\n"); abort(); }
    if (ng == 1) hipLaunchKernelGGL((layernorm_bwd_kernel<NV, 1, false>), grid, dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, u, err, ldh);
    else if (ng == 2) hipLaunchKernelGGL((layernorm_bwd_kernel<NV, 2, false>), grid, dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, u, err, ldh);
    else if (ldh != D) hipLaunchKernelGGL((layernorm_bwd_kernel<NV, 0, true>), grid, dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, u, err, ldh);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<NV, 0, false>), grid, dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, u, err, ldh);
}
void k_layernorm_bwd(const h16* dh, const float* x, const float* mean, const float* rstd, const float* g,
                     const float* dres, float* dx, h16* dx_h, int M, int D, const h16* P, int ng, h16* u, hipStream_t s, int* err, int ldh) {
    if (ldh <= 0) ldh = D;
    if (D <= 128 && !(P && u && ng > 0)) {
        ProfScope prof_("layernorm_bwd_kernel", 0.0, (double)M * D * (dx ? 16.0 : 12.0), s);
        hipLaunchKernelGGL(layernorm_bwd_small_kernel, dim3((M + 7) / 8), dim3(256), 0, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, err, ldh);
        return;
    }
    ProfScope prof_("layernorm_bwd_kernel", 0.0, (double)M * D * (dx ? 16.0 : 12.0), s);
    const int nv = (D / 4 + 63) / 64;
    if (!P || !u || ng < 0 || ng > 2) ng = 0;
    dim3 grid((M + 3) / 4);
    if (ng && grid.x > 1024) grid.x = 1024;          // 4 resident blocks per CU walk the rows
    switch (nv) {
        case 1: launch_ln_bwd<1>(grid, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, ng, u, err, ldh); break;
        case 2: launch_ln_bwd<2>(grid, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, ng, u, err, ldh); break;
        case 3: launch_ln_bwd<3>(grid, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, ng, u, err, ldh); break;
        default: launch_ln_bwd<4>(grid, s, dh, x, mean, rstd, g, dres, dx, dx_h, M, D, P, ng, u, err, ldh); break;
    }
}
void k_head_fwd(const float* x, int B, int T, int D, int C, float eps, const float* g, const float* b, const float* Wc,
                const float* bc, float* xhat, float* xf, float* rstd, float* logits, hipStream_t s) {
    ProfScope prof_("head_fwd_kernel", 0.0, (double)B * D * 4.0, s);
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(B), dim3(256), (D + 4) * sizeof(float), s, x, T, D, C, eps, g, b, Wc, bc,
                       xhat, xf, rstd, logits);
}
void k_head_fwd16(const h16* x, int B, int T, int D, int C, float eps, const float* g, const float* b, const float* Wc,
                  const float* bc, float* xhat, float* xf, float* rstd, float* logits, hipStream_t s) {
    ProfScope prof_("head_fwd_kernel", 0.0, (double)B * D * 2.0, s);
    hipLaunchKernelGGL(head_fwd_kernel<h16>, dim3(B), dim3(256), (D + 4) * sizeof(float), s, x, T, D, C, eps, g, b, Wc, bc,
                       xhat, xf, rstd, logits);
}
// ---- 16-bit residual streams ----
template <int NV>
static void launch_ln_fwd16(dim3 grid, size_t lds, hipStream_t s, const h16* x, h16* h, float* mean, float* rstd, const float* g,
                            const float* b, int M, int D, float eps, const h16* delta, h16* xout, const h16* P, int ng, h16* t, int* err) {
    if (ng == 1) hipLaunchKernelGGL((layernorm_fwd16_kernel<NV, 1>), grid, dim3(256), lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, err);
    else if (ng == 2) hipLaunchKernelGGL((layernorm_fwd16_kernel<NV, 2>), grid, dim3(256), lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, err);
    else if (ng == 3) hipLaunchKernelGGL((layernorm_fwd16_kernel<NV, 3>), grid, dim3(256), lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, err);
    else hipLaunchKernelGGL((layernorm_fwd16_kernel<NV, 0>), grid, dim3(256), 0, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, t, err);
}
void k_layernorm_fwd16(const h16* x, h16* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                       float eps, const h16* delta, h16* xout, const h16* P, int ng, h16* t, hipStream_t s, int* err) {
    if (D % 8 || D > 1024) { fprintf(stderr, "vitlora: LayerNorm (16-bit stream) needs D %% 8 == 0 and D <= 1024\n"); abort(); }
    ProfScope prof_("layernorm_fwd16_kernel", 0.0, (double)M * D * (delta ? (h ? 8.0 : 6.0) : 4.0), s);
    const int nv = (D / 8 + 31) / 32;
    if (!P || !t || !h || ng < 0 || ng > 3) ng = 0;
    dim3 grid(((M + 1) / 2 + 3) / 4);
    if (ng && grid.x > 1024) grid.x = 1024;          // resident blocks walk the rows, P stays in LDS
    const size_t lds = (size_t)8 * ng * D * sizeof(h16);
    switch (nv) {
        case 1: launch_ln_fwd16<1>(grid, lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, err); break;
        case 2: launch_ln_fwd16<2>(grid, lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, err); break;
        case 3: launch_ln_fwd16<3>(grid, lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, err); break;
        default: launch_ln_fwd16<4>(grid, lds, s, x, h, mean, rstd, g, b, M, D, eps, delta, xout, P, ng, t, err); break;
    }
}
template <int NV>
static void launch_ln_bwd16(dim3 grid, size_t lds, hipStream_t s, const h16* dh, const h16* x, const float* mean, const float* rstd,
                            const float* g, h16* dres, int M, int D, const h16* P, int ng, h16* u, int* err) {
    if (ng == 1) hipLaunchKernelGGL((layernorm_bwd16_kernel<NV, 1>), grid, dim3(256), lds, s, dh, x, mean, rstd, g, dres, M, D, P, u, err);
    else if (ng == 2) hipLaunchKernelGGL((layernorm_bwd16_kernel<NV, 2>), grid, dim3(256), lds, s, dh, x, mean, rstd, g, dres, M, D, P, u, err);
    else hipLaunchKernelGGL((layernorm_bwd16_kernel<NV, 0>), grid, dim3(256), 0, s, dh, x, mean, rstd, g, dres, M, D, P, u, err);
}
void k_layernorm_bwd16(const h16* dh, const h16* x, const float* mean, const float* rstd, const float* g, h16* dres, int M, int D,
                       const h16* P, int ng, h16* u, hipStream_t s, int* err) {
    if (D % 8 || D > 1024) { fprintf(stderr, "vitlora: LayerNorm (16-bit stream) needs D %% 8 == 0 and D <= 1024\n"); abort(); }
    ProfScope prof_("layernorm_bwd16_kernel", 0.0, (double)M * D * 8.0, s);
    const int nv = (D / 8 + 31) / 32;
    if (!P || !u || ng < 0 || ng > 2) ng = 0;
    dim3 grid(((M + 1) / 2 + 3) / 4);
    if (ng && grid.x > 1024) grid.x = 1024;          // 4 resident blocks per CU walk the rows
    const size_t lds = (size_t)8 * ng * D * sizeof(h16);
    switch (nv) {
        case 1: launch_ln_bwd16<1>(grid, lds, s, dh, x, mean, rstd, g, dres, M, D, P, ng, u, err); break;
        case 2: launch_ln_bwd16<2>(grid, lds, s, dh, x, mean, rstd, g, dres, M, D, P, ng, u, err); break;
        case 3: launch_ln_bwd16<3>(grid, lds, s, dh, x, mean, rstd, g, dres, M, D, P, ng, u, err); break;
        default: launch_ln_bwd16<4>(grid, lds, s, dh, x, mean, rstd, g, dres, M, D, P, ng, u, err); break;
    }
}
void k_ce_loss(const float* logits, const int64_t* labels, int B, int C, float* dlogits, float* loss_img, float* loss,
               int* err, hipStream_t s) {
    hipLaunchKernelGGL(ce_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, labels, B, C, dlogits, loss_img, err);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, loss_img, B, loss);
}
void k_head_bwd(const float* dlogits, const float* gscale, const float* Wc, const float* g, const float* xhat,
                const float* rstd, int B, int T, int D, int C, float* dx, h16* dx_h, hipStream_t s) {
    ProfScope prof_("head_bwd_kernel", 0.0, (double)B * D * 6.0, s);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(256), (D + 4) * sizeof(float), s, dlogits, gscale, Wc, g, xhat, rstd,
                       T, D, C, dx, dx_h);
}

namespace {
// per-image (one wave per image, four per block) or whole-batch (one block) max |dlogits| -> power-of-two scale that puts it
// in [2^9, 2^10)
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ dlogits, int B, int C, int uniform,
                                                         float* __restrict__ gscale, float* __restrict__ inv_gscale) {
    __shared__ float red[4];
    auto scale_of = [](float mx) -> float {
        if (!(mx > 0.f) || !(mx < INFINITY)) return 1.f;          // all-zero or non-finite row: leave it alone
        int e;
        (void)frexpf(mx, &e);                                     // mx = f * 2^e, f in [0.5, 1)
        e = 10 - e;
        e = e < -60 ? -60 : (e > 60 ? 60 : e);
        return ldexpf(1.f, e);
    };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (uniform) {
        float mx = 0.f;
        for (int i = threadIdx.x; i < B * C; i += 256) mx = fmaxf(mx, fabsf(dlogits[i]));
        mx = wave_max(mx);
        if (lane == 0) red[w] = mx;
        __syncthreads();
        const float sc = scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        for (int b = threadIdx.x; b < B; b += 256) { gscale[b] = sc; inv_gscale[b] = 1.f / sc; }
        return;
    }
    // (a single block walking 64 images per wave was 38 us of dependent load latency per PGD iteration at batch 256)
    for (int b = blockIdx.x * 4 + w; b < B; b += 4 * gridDim.x) {
        float mx = 0.f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, fabsf(dlogits[(int64_t)b * C + c]));
        mx = wave_max(mx);
        if (lane == 0) { const float sc = scale_of(mx); gscale[b] = sc; inv_gscale[b] = 1.f / sc; }
    }
}
}  // namespace
void k_grad_scale(const float* dlogits, int B, int C, int uniform, float* gscale, float* inv_gscale, hipStream_t s) {
    hipLaunchKernelGGL(grad_scale_kernel, dim3(uniform ? 1 : (B + 3) / 4), dim3(256), 0, s, dlogits, B, C, uniform, gscale, inv_gscale);
}
void k_classifier_grad(const float* dlogits, const float* xf, int B, int D, int C, float* dW, float* db, hipStream_t s) {
    hipLaunchKernelGGL(classifier_grad_kernel, dim3(nblk((int64_t)C * D, 256)), dim3(256), 0, s, dlogits, xf, B, D, C,
                       dW, db);
}
void k_pgd_step(float* adv, const float* x0, const float* grad, float eps, float alpha, float lo, float hi, int64_t n,
                hipStream_t s, int* err) {
    ProfScope prof_("pgd_step_kernel", 0.0, (double)n * 16.0, s);
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(pgd_step_kernel, dim3(nblk(n4 > 0 ? (n4 + 1) / 2 : 1, 256, 8192)), dim3(256), 0, s, adv, x0, grad, eps,
                       alpha, lo, hi, n4, n, err);
}
void k_zero(void* p, size_t bytes, hipStream_t s) {
    ProfScope prof_("zero_kernel", 0.0, (double)bytes, s);
    const int64_t n16 = (int64_t)(bytes / 16);
    if (n16 > 0) hipLaunchKernelGGL(zero_kernel, dim3(nblk(n16, 256, 4096)), dim3(256), 0, s, (f32x4*)p, n16);
}
void k_pgd_init(float* adv, const float* x0, float eps, float lo, float hi, uint64_t seed, int64_t n, hipStream_t s) {
    ProfScope prof_("pgd_init_kernel", 0.0, (double)n * 8.0, s);
    hipLaunchKernelGGL(pgd_init_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, s, adv, x0, eps, lo, hi, seed, n);
}
void k_adam(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, int t, int64_t n,
            hipStream_t s, int* err) {
    ProfScope prof_("adam_kernel", 0.0, (double)n * 28.0, s);
    const float bc1 = 1.f - powf(b1, (float)t);
    const float sqrt_bc2 = sqrtf(1.f - powf(b2, (float)t));
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, p, g, m, v, lr, b1, b2, eps, bc1, sqrt_bc2, n, err);
}
void k_channel_affine(float* dst, const float* src, const float* scale, const float* shift, int B, int64_t hw,
                      hipStream_t s) {
    const int64_t n = (int64_t)B * 3 * hw;
    hipLaunchKernelGGL(channel_affine_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, s, dst, src, scale[0], scale[1],
                       scale[2], shift[0], shift[1], shift[2], hw, n);
}
void k_fill_random_h16(h16* dst, size_t n, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_random_h16_kernel, dim3(4096), dim3(256), 0, s, dst, n, seed);
}
void k_quantize(const float* img, uint8_t* out, int B, int C, int H, int W, hipStream_t s) {
    const int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(quantize_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, img, out, B, C, H, W);
}
void k_pack_h16(const float* src, h16* dst, int rows, int cols, int ldd, int coff, float scale, hipStream_t s) {
    hipLaunchKernelGGL(pack_h16_kernel, dim3(nblk((int64_t)rows * cols, 256)), dim3(256), 0, s, src, dst, rows, cols,
                       ldd, coff, scale);
}
void k_pack_h16_t(const float* src, h16* dst, int rows, int cols, int ldd, int roff, float scale, hipStream_t s) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(pack_h16_t_kernel, grid, dim3(256), 0, s, src, dst, rows, cols, ldd, roff, scale);
}
void k_merge_lora(const float* W, const float* A, const float* B, int out, int in, int r, float sc, h16* dst, int ldd,
                  int roff, h16* dstT, int lddT, int coffT, hipStream_t s) {
    hipLaunchKernelGGL(merge_lora_kernel, dim3(nblk((int64_t)out * in, 256)), dim3(256), 0, s, W, A, B, out, in, r, sc,
                       dst, ldd, roff, dstT, lddT, coffT);
}

// ---------------------------------------------------------------------------------
// LoRA dropout (train mode): xd = x * mask/(1-p), mask = drop_scale(seed, stream, m*cols + c)
// ---------------------------------------------------------------------------------
namespace {
__global__ void dropout_kernel(const h16* __restrict__ x, h16* __restrict__ xd, int64_t n8, uint64_t seed,
                               uint32_t stream, float p, float inv_keep) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += stride) {
        const h16x8 v = *(const h16x8*)(x + i * 8);
        h16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = f2h(h2f(v[k]) * drop_scale(seed, stream, (uint64_t)(i * 8 + k), p, inv_keep));
        *(h16x8*)(xd + i * 8) = o;
    }
}
__global__ void dropout_mask_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint32_t stream, float p,
                                    float inv_keep) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = drop_scale(seed, stream, (uint64_t)i, p, inv_keep);
}
}  // namespace

void k_dropout(const h16* x, h16* xd, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s) {
    ProfScope prof_("dropout_kernel", 0.0, (double)n * 4.0, s);
    hipLaunchKernelGGL(dropout_kernel, dim3(nblk(n / 8, 256, 4096)), dim3(256), 0, s, x, xd, n / 8, seed, stream, p,
                       1.f / (1.f - p));
}
void k_dropout_mask(float* out, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(nblk(n, 256, 4096)), dim3(256), 0, s, out, n, seed, stream, p,
                       1.f / (1.f - p));
}

}  // namespace VLNS
