// Launchers of the fp32 parity-mode kernels (f32_kernels.hip).
#pragma once
#include "common.h"

namespace VLNS {

// C[M,N] = alpha * A * op(W) + bias + R.  transA: element (m, k) of A at A[k*lda + m] (else A[m*lda + k]);
// transW: element (n, k) of W at W[k*ldw + n] (else W[n*ldw + k]).  K, N and (when transposed) M multiples of 4,
// every base pointer / leading dimension 16-byte aligned.  R may alias C (accumulate).
struct GemmF32 {
    const float* A; int lda; int transA;
    const float* W; int ldw; int transW;
    int M, N, K;
    int Mstore;            // rows >= Mstore are computed but not stored
    float alpha;
    const float* bias;
    const float* R; int ldr;
    float* C; int ldc;
    int a_gather, patches; // A row remap of the patch-embedding backward: row m reads token row m + m / patches + 1
};

void k_gemm_f32(const GemmF32& g, hipStream_t s);
void k_ln_fwd_f32(const float* x, float* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                  float eps, hipStream_t s);
void k_ln_bwd_f32(const float* dh, const float* x, const float* mean, const float* rstd, const float* g, const float* dres,
                  float* dx, int M, int D, hipStream_t s);
void k_gelu_fwd_f32(const float* z, float* a, int64_t n, hipStream_t s);
void k_gelu_bwd_f32(float* dz, const float* z, int64_t n, hipStream_t s);
void k_mask_f32(float* dst, const float* src, int64_t n, int add, uint64_t seed, uint32_t stream, float p, hipStream_t s);
void k_patch_gather_f32(const float* x, float* out, int B, int S, int P, int normalise, const float* mean, const float* std,
                        hipStream_t s);
void k_patch_scatter_f32(const float* dp, float* gx, int B, int S, int P, const float inv_std[3], hipStream_t s);
int k_attn_fwd_f32(const float* qkv, float* ctx, float* lse, int B, int T, int H, int D, hipStream_t s);
int k_attn_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const float* lse, float* dqkv, int B, int T, int H,
                   int D, hipStream_t s);

}  // namespace VLNS
