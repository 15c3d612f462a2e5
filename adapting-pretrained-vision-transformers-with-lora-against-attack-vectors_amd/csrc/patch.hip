// Adversarial-patch overlay (expectation over transformations) -- the differentiable warp-and-paste step that the
// reference reaches through ART's AdversarialPatchPyTorch._random_overlay (patch_attack.py:47-75, 193-208):
//
//     padded_patch = resize(patch  -> S x S, bilinear)            mask = resize(circle / square mask -> S x S, bilinear)
//     per image:  P' = affine(padded_patch, angle, translate, scale; bilinear, zero fill)
//                 M' = affine(mask,         angle, translate, scale; nearest,  zero fill)
//     out = clamp(image * (1 - M') + P' * M', 0, 1)
//
// With distortion_scale_max > 0 (patch_attack.py:95) ART warps both canvases with torchvision's `perspective` (bilinear,
// zero fill) BEFORE the affine: the kernels then evaluate affine o perspective o resize tap by tap (4 x 4 x 4 taps for the
// patch, 4 x 4 behind the mask's nearest tap) from 8 homography coefficients per image; no intermediate canvas in HBM.
//
// restated from ART 1.20.1 / torchvision's tensor affine (grid_sample, align_corners = False); neither package is
// installable here, so the oracle (oracle/patch_oracle.py) is a torch restatement of the same formulas and these
// kernels are held to it ("parity unpinned").  The host passes the INVERSE affine matrix per image (6 floats,
// torchvision's _get_inverse_affine_matrix, computed in double); both kernels are HBM-bound elementwise passes.
#include "kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

struct Tap { int i0, i1; float w0, w1; };

// torch bilinear resize in -> out (align_corners = False): source position of destination index d
__device__ __forceinline__ Tap resize_tap(int d, int in, float ratio) {
    float src = ((float)d + 0.5f) * ratio - 0.5f;
    src = src < 0.f ? 0.f : src;
    int i0 = (int)src;
    i0 = i0 > in - 1 ? in - 1 : i0;
    const int i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    const float l = src - (float)i0;
    return {i0, i1, 1.f - l, l};
}

// ART's circular mask on a ps x ps grid: 1 - clip((x^2 + y^2)^40, -1, 1), x, y = linspace(-1, 1, ps); square: ones
__device__ __forceinline__ float base_mask(int r, int c, int ps, int circle) {
    if (!circle) return 1.f;
    const float x = ps > 1 ? -1.f + 2.f * c / (float)(ps - 1) : 0.f;
    const float y = ps > 1 ? -1.f + 2.f * r / (float)(ps - 1) : 0.f;
    const float z = powf(x * x + y * y, 40.f);
    return 1.f - fminf(fmaxf(z, -1.f), 1.f);
}

// resized mask at integer position (i, j) of the S x S canvas
__device__ __forceinline__ float mask_at(int j, int i, int ps, int S, int circle) {
    if (!circle) return 1.f;
    const float ratio = (float)ps / (float)S;
    const Tap ty = resize_tap(j, ps, ratio), tx = resize_tap(i, ps, ratio);
    return ty.w0 * (tx.w0 * base_mask(ty.i0, tx.i0, ps, 1) + tx.w1 * base_mask(ty.i0, tx.i1, ps, 1)) +
           ty.w1 * (tx.w0 * base_mask(ty.i1, tx.i0, ps, 1) + tx.w1 * base_mask(ty.i1, tx.i1, ps, 1));
}

struct Warp {          // per output pixel: source position in the S x S canvas
    float u, v;        // continuous source pixel coordinates
};
__device__ __forceinline__ Warp warp_of(const float* m, int x, int y, int S) {
    const float xb = (float)x - 0.5f * S + 0.5f, yb = (float)y - 0.5f * S + 0.5f;
    return {m[0] * xb + m[1] * yb + m[2] + 0.5f * S - 0.5f, m[3] * xb + m[4] * yb + m[5] + 0.5f * S - 0.5f};
}

// torchvision `_perspective_grid` + grid_sample(align_corners = False): source position in the canvas of pixel (x1, y1) of
// the perspective-warped canvas, u = (a (x1 + .5) + b (y1 + .5) + c) / (g (x1 + .5) + h (y1 + .5) + 1) - .5 (v alike).
// Returns up to four in-canvas bilinear taps; q == nullptr (no distortion): the pixel itself with weight 1, which keeps the
// undistorted path bit-identical to the two-stage form.
struct Taps4 { int u[4], v[4]; float w[4]; int n; };
__device__ __forceinline__ Taps4 persp_taps(const float* q, int x1, int y1, int S) {
    Taps4 t;
    if (q == nullptr) {
        t.n = 1; t.u[0] = x1; t.v[0] = y1; t.w[0] = 1.f;
        return t;
    }
    const float xb = (float)x1 + 0.5f, yb = (float)y1 + 0.5f;
    const float den = q[6] * xb + q[7] * yb + 1.f;
    const float u = (q[0] * xb + q[1] * yb + q[2]) / den - 0.5f, v = (q[3] * xb + q[4] * yb + q[5]) / den - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const int u0 = (int)fu, v0 = (int)fv;
    const float au = u - fu, av = v - fv;
    t.n = 0;
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
        for (int du = 0; du < 2; ++du) {
            const int cu = u0 + du, cv = v0 + dv;
            if (cu < 0 || cu >= S || cv < 0 || cv >= S) continue;
            t.u[t.n] = cu; t.v[t.n] = cv; t.w[t.n] = (du ? au : 1.f - au) * (dv ? av : 1.f - av);
            ++t.n;
        }
    return t;
}

// mask canvas after the perspective warp at integer position (i, j)
__device__ __forceinline__ float mask1_at(const float* q, int j, int i, int ps, int S, int circle) {
    if (q == nullptr) return mask_at(j, i, ps, S, circle);
    const Taps4 t = persp_taps(q, i, j, S);
    float m = 0.f;
    for (int k = 0; k < t.n; ++k) m += t.w[k] * mask_at(t.v[k], t.u[k], ps, S, circle);
    return m;
}

__global__ __launch_bounds__(256) void patch_overlay_kernel(const float* __restrict__ img, const float* __restrict__ patch,
                                                            const float* __restrict__ mats, const float* __restrict__ persp,
                                                            float* __restrict__ out, int B, int S, int ps, int circle) {
    const int64_t total = (int64_t)B * S * S;
    const float ratio = (float)ps / (float)S;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % S);
        const int y = (int)((t / S) % S);
        const int b = (int)(t / ((int64_t)S * S));
        const float* m = mats + b * 6;
        const float* q = persp ? persp + b * 8 : nullptr;
        const Warp wp = warp_of(m, x, y, S);
        // mask: nearest neighbour (round half to even, as grid_sample's nearbyint), zero outside the canvas
        const int nu = (int)nearbyintf(wp.u), nv = (int)nearbyintf(wp.v);
        float mk = 0.f;
        if (nu >= 0 && nu < S && nv >= 0 && nv < S) mk = mask1_at(q, nv, nu, ps, S, circle);
        float pv[3] = {0.f, 0.f, 0.f};
        if (mk != 0.f) {
            // patch: bilinear over the resized canvas, each canvas pixel itself bilinear over the ps x ps patch
            const float fu = floorf(wp.u), fv = floorf(wp.v);
            const int u0 = (int)fu, v0 = (int)fv;
            const float au = wp.u - fu, av = wp.v - fv;
#pragma unroll
            for (int dv = 0; dv < 2; ++dv)
#pragma unroll
                for (int du = 0; du < 2; ++du) {
                    const int cu = u0 + du, cv = v0 + dv;
                    if (cu < 0 || cu >= S || cv < 0 || cv >= S) continue;
                    const float wa = (du ? au : 1.f - au) * (dv ? av : 1.f - av);
                    const Taps4 pt = persp_taps(q, cu, cv, S);
                    for (int k = 0; k < pt.n; ++k) {
                        const float wk = wa * pt.w[k];
                        const Tap ty = resize_tap(pt.v[k], ps, ratio), tx = resize_tap(pt.u[k], ps, ratio);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float* pc = patch + c * ps * ps;
                            pv[c] += wk * (ty.w0 * (tx.w0 * pc[ty.i0 * ps + tx.i0] + tx.w1 * pc[ty.i0 * ps + tx.i1]) +
                                           ty.w1 * (tx.w0 * pc[ty.i1 * ps + tx.i0] + tx.w1 * pc[ty.i1 * ps + tx.i1]));
                        }
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t idx = (((int64_t)b * 3 + c) * S + y) * S + x;
            const float v = img[idx] * (1.f - mk) + pv[c] * mk;
            out[idx] = fminf(fmaxf(v, 0.f), 1.f);
        }
    }
}

// d(patch)[c][r][s] = sum over pixels g[b][c][y][x] * M' * (affine bilinear weight) * (resize bilinear weight).
// DETERMINISTIC (round 5; rounds 1-4 summed with float atomics, whose order -- and so the last bits of a few entries -- changed
// from run to run).  Two launches:
//   1. patch_grad_max_kernel: the largest |g * M'| over the pixels the patch covers (atomicMax on the bit pattern of a
//      non-negative float: order-independent);
//   2. patch_overlay_bwd_kernel, one workgroup per (image, band of rows): every contribution is converted to 64-bit FIXED POINT
//      whose unit is chosen from that maximum so that even the sum of ALL contributions of the call cannot leave 63 bits
//      (|contribution| <= max < 2^e; at most 2^cb contributions; unit 2^(e + cb - 62): >= 2^-35 of the largest pixel gradient at
//      batch 128, whatever the gradient's magnitude) and summed with INTEGER atomics -- in an LDS copy of the patch gradient,
//      then one 64-bit global atomic per touched entry into a scratch image.  Integer addition is associative: the result does
//      not depend on the order in which lanes and workgroups arrive.  The workgroup that arrives last converts the scratch image
//      to fp32, writes the result and leaves the scratch zeroed for the next call.
// A non-finite pixel gradient turns the whole result into NaN (the backward's non-finite flag has fired upstream in that case).
constexpr int PS_MAX = 64;                               // patch_args_ok (vitlora.hip)
__device__ unsigned long long g_patch_acc[3 * PS_MAX * PS_MAX];     // zero between calls; calls on one device are stream-ordered by the caller
__device__ unsigned g_patch_arrived, g_patch_poison, g_patch_max_bits;

__global__ __launch_bounds__(256) void patch_grad_max_kernel(const float* __restrict__ g, const float* __restrict__ mats,
                                                             const float* __restrict__ persp, int S, int ps, int circle, int bands) {
    const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
    const int rows = (S + bands - 1) / bands;
    const int y0 = band * rows, y1 = min(S, y0 + rows);
    const float* m = mats + b * 6;
    const float* q = persp ? persp + b * 8 : nullptr;
    float mx = 0.f;
    unsigned bad = 0u;
    for (int t = y0 * S + threadIdx.x; t < y1 * S; t += 256) {
        const int x = t % S, y = t / S;
        const Warp wp = warp_of(m, x, y, S);
        const int nu = (int)nearbyintf(wp.u), nv = (int)nearbyintf(wp.v);
        if (nu < 0 || nu >= S || nv < 0 || nv >= S) continue;
        const float mk = mask1_at(q, nv, nu, ps, S, circle);
        if (mk == 0.f) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = fabsf(g[(((int64_t)b * 3 + c) * S + y) * S + x] * mk);
            if (!(v < INFINITY)) bad = 1u; else mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&g_patch_max_bits, __float_as_uint(mx));
    if (bad) atomicOr(&g_patch_poison, 1u);
}

__global__ __launch_bounds__(256) void patch_overlay_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mats,
                                                                const float* __restrict__ persp, float* __restrict__ dpatch,
                                                                int S, int ps, int circle, int bands, int count_bits) {
    extern __shared__ unsigned long long acc[];        // [3][ps][ps] fixed point
    __shared__ unsigned s_last;
    const int n = 3 * ps * ps;
    for (int i = threadIdx.x; i < n; i += 256) acc[i] = 0ull;
    // the call's fixed-point unit: max < 2^e (frexp exponent), every contribution is at most max in magnitude
    const float gmax = __uint_as_float(__hip_atomic_load(&g_patch_max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int e = 0;
    (void)frexpf(gmax, &e);
    e = max(e, -90);                                     // keeps 2^(62 - cb - e) inside fp32
    const float to_fix = ldexpf(1.f, 62 - count_bits - e);
    __syncthreads();
    const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
    const int rows = (S + bands - 1) / bands;
    const int y0 = band * rows, y1 = min(S, y0 + rows);
    const float* m = mats + b * 6;
    const float* q = persp ? persp + b * 8 : nullptr;
    const float ratio = (float)ps / (float)S;
    auto fix_add = [&](unsigned long long* p, float v) {
        atomicAdd(p, (unsigned long long)__float2ll_rn(v * to_fix));      // two's complement: signed sums wrap correctly
    };
    for (int t = y0 * S + threadIdx.x; gmax > 0.f && t < y1 * S; t += 256) {
        const int x = t % S, y = t / S;
        const Warp wp = warp_of(m, x, y, S);
        const int nu = (int)nearbyintf(wp.u), nv = (int)nearbyintf(wp.v);
        if (nu < 0 || nu >= S || nv < 0 || nv >= S) continue;
        const float mk = mask1_at(q, nv, nu, ps, S, circle);
        if (mk == 0.f) continue;
        float gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gv[c] = g[(((int64_t)b * 3 + c) * S + y) * S + x] * mk;
            if (!(fabsf(gv[c]) < INFINITY)) gv[c] = 0.f;          // counted by the first launch: the result is NaN anyway
        }
        const float fu = floorf(wp.u), fv = floorf(wp.v);
        const int u0 = (int)fu, v0 = (int)fv;
        const float au = wp.u - fu, av = wp.v - fv;
#pragma unroll
        for (int dv = 0; dv < 2; ++dv)
#pragma unroll
            for (int du = 0; du < 2; ++du) {
                const int cu = u0 + du, cv = v0 + dv;
                if (cu < 0 || cu >= S || cv < 0 || cv >= S) continue;
                const float wa = (du ? au : 1.f - au) * (dv ? av : 1.f - av);
                const Taps4 pt = persp_taps(q, cu, cv, S);
                for (int k = 0; k < pt.n; ++k) {
                    const float wk = wa * pt.w[k];
                    const Tap ty = resize_tap(pt.v[k], ps, ratio), tx = resize_tap(pt.u[k], ps, ratio);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        unsigned long long* pc = acc + c * ps * ps;
                        const float gw = gv[c] * wk;
                        fix_add(pc + ty.i0 * ps + tx.i0, gw * ty.w0 * tx.w0);
                        fix_add(pc + ty.i0 * ps + tx.i1, gw * ty.w0 * tx.w1);
                        fix_add(pc + ty.i1 * ps + tx.i0, gw * ty.w1 * tx.w0);
                        fix_add(pc + ty.i1 * ps + tx.i1, gw * ty.w1 * tx.w1);
                    }
                }
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if (acc[i] != 0ull) atomicAdd(g_patch_acc + i, acc[i]);
    // arrival ticket: this workgroup's adds are performed device-wide before its ticket is taken
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&g_patch_arrived, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const unsigned poison = atomicExch(&g_patch_poison, 0u);
    const double from_fix = ldexp(1.0, -(62 - count_bits - e));
    for (int i = threadIdx.x; i < n; i += 256) {
        const long long v = (long long)atomicExch(g_patch_acc + i, 0ull);       // read at the coherence point and re-arm
        dpatch[i] = poison ? __uint_as_float(0x7fc00000u) : (float)((double)v * from_fix);
    }
    if (threadIdx.x == 0) { atomicExch(&g_patch_arrived, 0u); atomicExch(&g_patch_max_bits, 0u); }
}

__global__ void clamp_kernel(float* __restrict__ x, float lo, float hi, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) x[i] = fminf(fmaxf(x[i], lo), hi);
}

}  // namespace

void k_patch_overlay(const float* img, const float* patch, const float* mats, const float* persp, float* out, int B, int S,
                     int ps, int circle, hipStream_t s) {
    ProfScope prof_("patch_overlay_kernel", 0.0, (double)B * 3 * S * S * 8.0, s);
    const int64_t total = (int64_t)B * S * S;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(patch_overlay_kernel, dim3((unsigned)blocks), dim3(256), 0, s, img, patch, mats, persp, out, B, S, ps, circle);
}
void k_patch_overlay_bwd(const float* g, const float* mats, const float* persp, float* dpatch, int B, int S, int ps, int circle,
                         hipStream_t s) {
    ProfScope prof_("patch_overlay_bwd_kernel", 0.0, (double)B * 3 * S * S * 4.0, s);
    const int bands = 8;
    const size_t lds = (size_t)3 * ps * ps * sizeof(unsigned long long);
    if (lds > 48 * 1024)       // ps > 45: above the default dynamic-LDS limit (not a stream operation; legal under capture)
        (void)hipFuncSetAttribute((const void*)patch_overlay_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // contributions of the whole call: pixels x 3 channels are separate entries, so per entry at most B * S * S pixels x 4 affine
    // taps x (4 perspective taps) x 4 resize taps
    double cnt = (double)B * S * S * 16.0 * (persp ? 4.0 : 1.0);
    int count_bits = 1;
    while (ldexp(1.0, count_bits) < cnt) ++count_bits;
    hipLaunchKernelGGL(patch_grad_max_kernel, dim3(B * bands), dim3(256), 0, s, g, mats, persp, S, ps, circle, bands);
    hipLaunchKernelGGL(patch_overlay_bwd_kernel, dim3(B * bands), dim3(256), lds, s, g, mats, persp, dpatch, S, ps, circle, bands, count_bits);
}
void k_clamp(float* x, float lo, float hi, int64_t n, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, x, lo, hi, n);
}

}  // namespace VLNS
