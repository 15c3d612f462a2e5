// K10 (fused sign / project / clamp step, whitebox_attacks.py:32-36) launch-shape sweep: which access pattern gets the
// four fp32 streams (read g, adv, x0; write adv: 16 B per element) closest to the HBM rate on MI355X.
// Standalone (no library): hipcc --offload-arch=gfx950 -O3 tools/k10_sweep.hip -o tools/k10_sweep ; ./tools/k10_sweep
// Every variant computes the same function; results are compared bit for bit with variant 0.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sgn(float g) { return (g > 0.f) ? 1.f : ((g < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ f32x4 step4(f32x4 a, f32x4 x, f32x4 g, float eps, float alpha) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = a[k] + alpha * sgn(g[k]);
        const float d = fminf(fmaxf(t - x[k], -eps), eps);
        o[k] = fminf(fmaxf(x[k] + d, 0.f), 1.f);
    }
    return o;
}

// U vectors per thread per trip, grid-stride over trips; NT bit 0: non-temporal stores, bit 1: non-temporal loads
template <int U, int NT>
__global__ __launch_bounds__(256) void k10_strided(float* __restrict__ adv, const float* __restrict__ x0, const float* __restrict__ grad,
                                                   float eps, float alpha, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += stride * U) {
        f32x4 a[U], x[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = i + u * stride;
            if (j < n4) {
                if (NT & 2) {
                    a[u] = __builtin_nontemporal_load((const f32x4*)adv + j);
                    x[u] = __builtin_nontemporal_load((const f32x4*)x0 + j);
                    g[u] = __builtin_nontemporal_load((const f32x4*)grad + j);
                } else {
                    a[u] = ((const f32x4*)adv)[j]; x[u] = ((const f32x4*)x0)[j]; g[u] = ((const f32x4*)grad)[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = i + u * stride;
            if (j < n4) {
                const f32x4 o = step4(a[u], x[u], g[u], eps, alpha);
                if (NT & 1) __builtin_nontemporal_store(o, (f32x4*)adv + j);
                else ((f32x4*)adv)[j] = o;
            }
        }
    }
}

// each workgroup owns contiguous chunks of CH vectors (CH * 16 B per stream), U vectors per thread in flight
template <int U, int NT>
__global__ __launch_bounds__(256) void k10_chunked(float* __restrict__ adv, const float* __restrict__ x0, const float* __restrict__ grad,
                                                   float eps, float alpha, long long n4) {
    constexpr long long CH = 256ll * U;
    const long long nchunks = (n4 + CH - 1) / CH;
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long base = c * CH + threadIdx.x;
        f32x4 a[U], x[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = base + u * 256;
            if (j < n4) {
                if (NT & 2) {
                    a[u] = __builtin_nontemporal_load((const f32x4*)adv + j);
                    x[u] = __builtin_nontemporal_load((const f32x4*)x0 + j);
                    g[u] = __builtin_nontemporal_load((const f32x4*)grad + j);
                } else {
                    a[u] = ((const f32x4*)adv)[j]; x[u] = ((const f32x4*)x0)[j]; g[u] = ((const f32x4*)grad)[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = base + u * 256;
            if (j < n4) {
                const f32x4 o = step4(a[u], x[u], g[u], eps, alpha);
                if (NT & 1) __builtin_nontemporal_store(o, (f32x4*)adv + j);
                else ((f32x4*)adv)[j] = o;
            }
        }
    }
}

__global__ void fill(float* p, long long n, unsigned seed, float lo, float hi) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned z = (unsigned)i * 2654435761u + seed;
        z ^= z >> 16; z *= 2246822519u; z ^= z >> 13;
        p[i] = lo + (hi - lo) * (float)(z >> 8) * (1.f / 16777216.f);
    }
}
__global__ void copy4(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) d[i] = s[i];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

struct Variant { const char* name; void (*launch)(float*, const float*, const float*, long long, int, hipStream_t); };

template <int U, int NT> void l_strided(float* a, const float* x, const float* g, long long n4, int grid, hipStream_t s) {
    hipLaunchKernelGGL((k10_strided<U, NT>), dim3(grid), dim3(256), 0, s, a, x, g, 8.f / 255, 2.f / 255, n4);
}
template <int U, int NT> void l_chunked(float* a, const float* x, const float* g, long long n4, int grid, hipStream_t s) {
    hipLaunchKernelGGL((k10_chunked<U, NT>), dim3(grid), dim3(256), 0, s, a, x, g, 8.f / 255, 2.f / 255, n4);
}

int main(int argc, char** argv) {
    const long long B = argc > 1 ? atoll(argv[1]) : 256;
    const long long n = B * 3 * 224 * 224, n4 = n / 4;
    float *adv, *adv0, *x0, *g, *ref;
    CK(hipMalloc(&adv, n * 4)); CK(hipMalloc(&adv0, n * 4)); CK(hipMalloc(&x0, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&ref, n * 4));
    fill<<<4096, 256>>>(x0, n, 1, 0.f, 1.f);
    fill<<<4096, 256>>>(adv0, n, 2, 0.f, 1.f);
    fill<<<4096, 256>>>(g, n, 3, -1.f, 1.f);
    CK(hipDeviceSynchronize());
    std::vector<Variant> vs = {
        {"strided U1", l_strided<1, 0>}, {"strided U2", l_strided<2, 0>}, {"strided U4", l_strided<4, 0>},
        {"strided U2 nt-store", l_strided<2, 1>}, {"strided U2 nt-load", l_strided<2, 2>}, {"strided U2 nt-both", l_strided<2, 3>},
        {"strided U4 nt-both", l_strided<4, 3>},
        {"chunked U1", l_chunked<1, 0>}, {"chunked U2", l_chunked<2, 0>}, {"chunked U4", l_chunked<4, 0>},
        {"chunked U2 nt-store", l_chunked<2, 1>}, {"chunked U2 nt-both", l_chunked<2, 3>}, {"chunked U4 nt-both", l_chunked<4, 3>},
    };
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h_ref(1 << 16), h_out(1 << 16);
    printf("K10 sweep, batch %lld (%.1f MB per stream, %.1f MB per launch)\n", B, n * 4 / 1e6, n * 16 / 1e6);
    // yardstick: float4 copy of the same footprint (2 streams)
    {
        for (int it = 0; it < 3; ++it) copy4<<<4096, 256>>>((const f32x4*)adv0, (f32x4*)adv, n4);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; ++it) { copy4<<<4096, 256>>>((const f32x4*)adv0, (f32x4*)adv, n4); copy4<<<4096, 256>>>((const f32x4*)x0, (f32x4*)ref, n4); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("float4 copy: %.1f us per %.1f MB = %.2f TB/s\n", ms / 40 * 1e3, n * 8 / 1e6, n * 8.0 / (ms / 40 * 1e-3) / 1e12);
    }
    bool first = true;
    for (const Variant& v : vs)
        for (int grid : grids) {
            // correctness: one step from adv0
            CK(hipMemcpy(adv, adv0, n * 4, hipMemcpyDeviceToDevice));
            v.launch(adv, x0, g, n4, grid, 0);
            CK(hipDeviceSynchronize());
            if (first) { CK(hipMemcpy(ref, adv, n * 4, hipMemcpyDeviceToDevice)); first = false; }
            // compare a strided sample + the tail
            bool same = true;
            for (long long off : {0ll, n / 3, n - (1ll << 16)}) {
                CK(hipMemcpy(h_ref.data(), ref + off, h_ref.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h_out.data(), adv + off, h_out.size() * 4, hipMemcpyDeviceToHost));
                same &= memcmp(h_ref.data(), h_out.data(), h_ref.size() * 4) == 0;
            }
            for (int it = 0; it < 3; ++it) v.launch(adv, x0, g, n4, grid, 0);
            CK(hipEventRecord(e0));
            const int iters = 20;
            for (int it = 0; it < iters; ++it) v.launch(adv, x0, g, n4, grid, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms / iters * 1e3;
            printf("%-22s grid %5d: %7.1f us  %.2f TB/s  (%.3f of 8)  %s\n", v.name, grid, us, n * 16.0 / (us * 1e-6) / 1e12,
                   n * 16.0 / (us * 1e-6) / 8e12, same ? "bit-equal" : "MISMATCH");
        }
    return 0;
}
