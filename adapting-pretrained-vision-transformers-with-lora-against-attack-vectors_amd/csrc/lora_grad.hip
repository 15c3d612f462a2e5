// LoRA weight gradients (train_loras.py:314 loss.backward(), restricted to lora_A / lora_B):
//   dB[n][j] = s * sum_m dy[m][n] * t[m][j]      (t = dropout(x) A^T kept from the forward)
//   dA[j][k] = s * sum_m u[m][j]  * x[m][k]      (u = dy B  kept from the dgrad)
// Generic form: Out[i][j] = scale * sum_m L[m][i] * R[m][j] over the M token rows, L = the wide
// operand (768..3072 columns), R = the rank-r operand (<= 64 columns).
//
// Both operands are token-major, i.e. the summed index m is the ROW of both: each 32-token step
// is staged into LDS as two swizzled [32][64] h16 images and BOTH MFMA operands are read with the
// hardware-transposed ds_read_b64_tr_b16 (same k order on both sides).  A workgroup owns 64 columns
// of L and a chunk of tokens and writes its partial result with plain stores into ITS chunk's copy of the
// flat gradient (a slab [chunks][flat LoRA elements] in the train workspace); k_reduce_chunks then sums the
// copies in chunk order.  No atomics: two runs of the same step give the same bits (round 5; rounds 1-4
// added the partials to the output with float atomics, whose order changed from run to run).  HBM-bound on L.
#include "kernels.h"
#include "prof.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

constexpr int HD = 64;          // image row length (elements)
constexpr int WG_MAX_CHUNKS = 32;   // bounds the slab: chunks x flat LoRA elements x 4 B (r = 8: <= 123 MB)
constexpr int MCHUNK_MIN = 512;     // tokens per workgroup: at least this many, and at most WG_MAX_CHUNKS chunks per launch

// transposed fragment of a swizzled [rows][64] image (chunk c of row r at c ^ (r & 7)):
// element j of lane (fr, fg) = img[r0 + 16*(j>>2) + 4*fg + (j&3)][c0 + fr]
__device__ __forceinline__ h16x8 tr_frag16(const h16* img, int r0, int c0, int fr, int fg) {
    const int p = fr & 3;
    const int row = r0 + 4 * fg + (fr >> 2);
    const int chunk = (c0 >> 3) + (p >> 1);
    const h16* a0 = img + row * HD + ((chunk ^ (row & 7)) << 3) + ((p & 1) << 2);
    const int row1 = row + 16;
    const h16* a1 = img + row1 * HD + ((chunk ^ (row1 & 7)) << 3) + ((p & 1) << 2);
    return cat4(lds_read_tr16(a0), lds_read_tr16(a1));
}

// NRT = 16-column tiles of R covered (ncr <= 16*NRT)
template <int NRT>
__global__ __launch_bounds__(256) void lora_wgrad_mfma_kernel(const h16* __restrict__ L, int ldl, int ncl,
                                                              const h16* __restrict__ R, int ldr, int ncr, int M,
                                                              float scale, float* __restrict__ out, int ldo,
                                                              int transpose_out, const float* __restrict__ inv_gscale,
                                                              int mchunk, long long chunk_stride) {
    __shared__ __attribute__((aligned(16))) h16 sL[32 * HD];
    __shared__ __attribute__((aligned(16))) h16 sR[32 * HD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int i0 = blockIdx.x * 64;                 // first L column of this workgroup
    const int m_begin = blockIdx.y * mchunk;
    const int m_end = min(M, m_begin + mchunk);
    out += (size_t)blockIdx.y * (size_t)chunk_stride;      // this chunk's copy of the flat gradient
    const int srow = tid >> 3, sc = tid & 7;        // staging: 32 rows x 8 chunks of 16 B
    const bool l_ok = i0 + sc * 8 < ncl;            // ncl is a multiple of 8 (module widths)
    const bool r_ok = sc * 8 < ncr;
    // 16-byte loads of R need r % 8 == 0 and an aligned slot; other ranks (e.g. r = 4) go element-wise
    const bool r_vec = (ncr % 8 == 0) && (ldr % 8 == 0) && (((size_t)R & 15) == 0);

    f32x4 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 zero;
#pragma unroll
    for (int k = 0; k < 8; ++k) zero[k] = (h16)0.f;

    auto load = [&](int m0, h16x8& lv, h16x8& rv) {
        const int m = m0 + srow;
        lv = zero; rv = zero;
        if (m < m_end) {
            if (l_ok) lv = *(const h16x8*)(L + (size_t)m * ldl + i0 + sc * 8);
            if (r_ok) {
                if (r_vec) rv = *(const h16x8*)(R + (size_t)m * ldr + sc * 8);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (sc * 8 + k < ncr) rv[k] = R[(size_t)m * ldr + sc * 8 + k];
                }
            }
        }
    };
    h16x8 lv, rv;
    load(m_begin, lv, rv);
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        __syncthreads();                            // previous step's fragment reads are done
        *(h16x8*)(sL + srow * HD + ((sc ^ (srow & 7)) << 3)) = lv;
        *(h16x8*)(sR + srow * HD + ((sc ^ (srow & 7)) << 3)) = rv;
        __syncthreads();
        if (m0 + 32 < m_end) load(m0 + 32, lv, rv); // next step's global loads fly under the MFMAs
        // D[row = j (R column)][col = i (L column)] += sum_m R[m][j] * L[m][i]
        const h16x8 lb = tr_frag16(sL, 0, w * 16, fr, fg);
#pragma unroll
        for (int t = 0; t < NRT; ++t) acc[t] = mfma16(tr_frag16(sR, 0, t * 16, fr, fg), lb, acc[t]);
    }
    const int i = i0 + w * 16 + fr;
    if (inv_gscale) scale *= inv_gscale[0];        // undo the (batch-uniform) fp16 gradient scale
    if (i < ncl) {
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = t * 16 + 4 * fg + k;
                if (j < ncr) {
                    float* dst = transpose_out ? out + (size_t)j * ldo + i : out + (size_t)i * ldo + j;
                    *dst = scale * acc[t][k];
                }
            }
    }
}

__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n,
                                                            int chunks, int64_t stride) {
    const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (i >= n) return;
    float a = slab[i];
    for (int c = 1; c < chunks; ++c) a += slab[(size_t)c * stride + i];      // chunk order: fixed
    out[i] = a;
}

}  // namespace

int lora_wgrad_chunk(int M) {
    int c = (M + WG_MAX_CHUNKS - 1) / WG_MAX_CHUNKS;
    c = (c + 31) / 32 * 32;
    return c < MCHUNK_MIN ? MCHUNK_MIN : c;
}
int lora_wgrad_chunks(int M) { const int c = lora_wgrad_chunk(M); return (M + c - 1) / c; }

// L: wide operand [M, ncl]; Rm: rank operand [M, ncr <= 64]; chunk c of the token rows writes scale * L_c^T Rm_c (or its
// transpose) to out + c * chunk_stride (every element of the [ncl, ncr] block, plain stores)
void k_lora_wgrad(const h16* L, int ldl, int ncl, const h16* Rm, int ldr, int ncr, int M, float scale, float* out,
                  int ldo, int transpose_out, const float* inv_gscale, int64_t chunk_stride, hipStream_t s) {
    ProfScope prof_("lora_wgrad_mfma_kernel", 2.0 * M * (double)ncl * ncr, (double)M * ncl * 2.0, s);
    const int mchunk = lora_wgrad_chunk(M);
    dim3 grid((ncl + 63) / 64, (M + mchunk - 1) / mchunk);
    if (ncr <= 16) hipLaunchKernelGGL((lora_wgrad_mfma_kernel<1>), grid, dim3(256), 0, s, L, ldl, ncl, Rm, ldr, ncr, M, scale, out, ldo, transpose_out, inv_gscale, mchunk, (long long)chunk_stride);
    else if (ncr <= 32) hipLaunchKernelGGL((lora_wgrad_mfma_kernel<2>), grid, dim3(256), 0, s, L, ldl, ncl, Rm, ldr, ncr, M, scale, out, ldo, transpose_out, inv_gscale, mchunk, (long long)chunk_stride);
    else hipLaunchKernelGGL((lora_wgrad_mfma_kernel<4>), grid, dim3(256), 0, s, L, ldl, ncl, Rm, ldr, ncr, M, scale, out, ldo, transpose_out, inv_gscale, mchunk, (long long)chunk_stride);
}

// out[i] = sum over chunks (in chunk order) of slab[c * stride + i]
void k_reduce_chunks(const float* slab, float* out, int64_t n, int chunks, int64_t stride, hipStream_t s) {
    ProfScope prof_("reduce_chunks_kernel", 0.0, (double)n * 4.0 * (chunks + 1), s);
    if (n <= 0) return;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, out, n, chunks, stride);
}

}  // namespace VLNS
