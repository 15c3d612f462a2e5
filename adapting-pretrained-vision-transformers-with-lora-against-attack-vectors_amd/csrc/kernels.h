// Host-callable launchers of every kernel (all enqueue on the given stream, no sync).
#pragma once
#include "common.h"
#include "gemm.h"

// elementwise.hip
void k_patch_gather(const float* x, bf16* out, int B, int S, int P, int normalise, const float* mean,
                    const float* std, hipStream_t s);
void k_cls_rows(float* x, const float* cls, const float* pos, int B, int T, int D, hipStream_t s);
// h = LN(x) (bf16) + row statistics.  With `delta`: first x_out = x + delta (fp32 residual stream + the bf16
// output of the projection before it), then the LN of x_out; h == nullptr: the add alone.
// P / ng / t: optional fused LoRA down-projection of the row of h (8*ng rows of P, ng <= 3; t has 64 columns).
void k_layernorm_fwd(const float* x, bf16* h, float* mean, float* rstd, const float* g, const float* b, int M, int D,
                     float eps, const bf16* delta, float* xout, const bf16* P, int ng, bf16* t, hipStream_t s);
// P / ng / u: optional fused LoRA down-projection of the bf16 output row (u[row][0..63] = dx_bf[row] . P[j], 8*ng rows
// of P [>= 8*ng, D]; ng in {1, 2}; u has 64 columns), see lora_down_row in elementwise.hip
void k_layernorm_bwd(const bf16* dh, const float* x, const float* mean, const float* rstd, const float* g,
                     const float* dres, float* dx, bf16* dx_bf, int M, int D, const bf16* P, int ng, bf16* u, hipStream_t s);
void k_head_fwd(const float* x, int B, int T, int D, int C, float eps, const float* g, const float* b, const float* Wc,
                const float* bc, float* xhat, float* xf, float* rstd, float* logits, hipStream_t s);
void k_ce_loss(const float* logits, const int64_t* labels, int B, int C, float* dlogits, float* loss_img, float* loss,
               hipStream_t s);
void k_head_bwd(const float* dlogits, const float* Wc, const float* g, const float* xhat, const float* rstd, int B,
                int T, int D, int C, float* dx, bf16* dx_bf, hipStream_t s);
void k_classifier_grad(const float* dlogits, const float* xf, int B, int D, int C, float* dW, float* db, hipStream_t s);
void k_pgd_step(float* adv, const float* x0, const float* grad, float eps, float alpha, float lo, float hi, int64_t n,
                hipStream_t s);
void k_pgd_init(float* adv, const float* x0, float eps, float lo, float hi, uint64_t seed, int64_t n, hipStream_t s);
void k_adam(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, int t, int64_t n,
            hipStream_t s);
void k_channel_affine(float* dst, const float* src, const float* scale, const float* shift, int B, int64_t hw,
                      hipStream_t s);
void k_fill_random_bf16(bf16* dst, size_t n, uint64_t seed, hipStream_t s);   // U(-1,1), benchmarks only
void k_quantize(const float* img, uint8_t* out, int B, int C, int H, int W, hipStream_t s);
void k_pack_bf16(const float* src, bf16* dst, int rows, int cols, int ldd, int coff, float scale, hipStream_t s);
void k_pack_bf16_t(const float* src, bf16* dst, int rows, int cols, int ldd, int roff, float scale, hipStream_t s);
void k_merge_lora(const float* W, const float* A, const float* B, int out, int in, int r, float sc, bf16* dst, int ldd,
                  int roff, bf16* dstT, int lddT, int coffT, hipStream_t s);

void k_dropout(const bf16* x, bf16* xd, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s);
void k_dropout_mask(float* out, int64_t n, uint64_t seed, uint32_t stream, float p, hipStream_t s);
void k_merge_f32(const float* W, const float* A, const float* B, int out, int in, int r, float sc, float* dst,
                 hipStream_t s);

// attention.hip
void attention_init();   // one-time kernel attributes (outside any stream capture)
int k_attention_fwd(const bf16* qkv, bf16* ctx, float* lse2, int B, int T, int H, int D, hipStream_t s);
int k_attention_bwd(const bf16* qkv, const bf16* ctx, const bf16* dctx, const float* lse2, bf16* dqkv, int B, int T,
                    int H, int D, hipStream_t s);

// attention32.hip (32x32x16 MFMA generation; same interfaces)
void attention32_init();
int k_attention32_fwd(const bf16* qkv, bf16* ctx, float* lse2, int B, int T, int H, int D, hipStream_t s);
int k_attention32_bwd(const bf16* qkv, const bf16* ctx, const bf16* dctx, const float* lse2, bf16* dqkv, int B, int T,
                      int H, int D, hipStream_t s);

// lora_grad.hip
// dB[n][j] (+)= sum_m dy[m][n] * t[m][j] ; dA[j][k] (+)= sum_m u[m][j] * x[m][k]  (fp32 outputs)
void k_lora_wgrad(const bf16* L, int ldl, int ncols_l, const bf16* Rm, int ldr, int ncols_r, int M, float scale,
                  float* out, int ldo, int transpose_out, float* scratch, hipStream_t s);
