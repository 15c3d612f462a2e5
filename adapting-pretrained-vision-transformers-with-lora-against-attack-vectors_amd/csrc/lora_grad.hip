// LoRA weight gradients (train_loras.py:314 loss.backward(), restricted to lora_A / lora_B):
//   dB[n][j] = s * sum_m dy[m][n] * t[m][j]      (t = x A^T kept from the forward)
//   dA[j][k] = s * sum_m u[m][j]  * x[m][k]      (u = dy B  kept from the dgrad)
// Generic form: Out[i][j] = scale * sum_m L[m][i] * R[m][j] over the M token rows.
// v1: LDS-staged VALU kernel, M split over workgroups, fp32 atomics into a zeroed output.
#include "kernels.h"
#include "prof.h"

namespace {

constexpr int MC = 64;    // token rows per workgroup

__global__ __launch_bounds__(256) void lora_wgrad_kernel(const bf16* __restrict__ L, int ldl, int ncl,
                                                         const bf16* __restrict__ R, int ldr, int ncr, int M,
                                                         float scale, float* __restrict__ out, int ldo, int transpose_out) {
    __shared__ float sL[MC][65];
    __shared__ float sR[MC][64];
    const int m0 = blockIdx.x * MC;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < MC * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        const int m = m0 + r;
        sL[r][c] = (m < M && i0 + c < ncl) ? bf2f(L[(size_t)m * ldl + i0 + c]) : 0.f;
        sR[r][c] = (m < M && j0 + c < ncr) ? bf2f(R[(size_t)m * ldr + j0 + c]) : 0.f;
    }
    __syncthreads();
    const int il = tid & 63, jq = tid >> 6;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int r = 0; r < MC; ++r) {
        const float lv = sL[r][il];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += lv * sR[r][jq * 16 + k];
    }
    const int i = i0 + il;
    if (i < ncl) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = j0 + jq * 16 + k;
            if (j < ncr) {
                float* dst = transpose_out ? out + (size_t)j * ldo + i : out + (size_t)i * ldo + j;
                atomicAdd(dst, scale * acc[k]);
            }
        }
    }
}

}  // namespace

void k_lora_wgrad(const bf16* L, int ldl, int ncl, const bf16* R, int ldr, int ncr, int M, float scale, float* out,
                  int ldo, int transpose_out, float* /*scratch*/, hipStream_t s) {
    ProfScope prof_("lora_wgrad_kernel", 2.0 * M * (double)ncl * ncr, 0.0, s);
    dim3 grid((M + MC - 1) / MC, (ncl + 63) / 64, (ncr + 63) / 64);
    hipLaunchKernelGGL(lora_wgrad_kernel, grid, dim3(256), 0, s, L, ldl, ncl, R, ldr, ncr, M, scale, out, ldo,
                       transpose_out);
}
