// Multi-head self-attention for ViT sequence lengths (T <= 224 tokens, head_dim 64) on gfx950.
// Stands in for HF eager_attention_forward (modeling_vit.py:164-189: softmax(QK^T * d^-1/2) V,
// softmax in fp32) and its autograd backward on the reference's hot path.
//
// One workgroup per (image, head): the whole K and V (and in backward Q, dO) of the head sit in
// LDS as XOR-swizzled row-major [keys][64] bf16 images (128-byte rows, 16-byte chunk c of row r
// stored at chunk c ^ (r & 7)).  The same image serves row reads (ds_read_b128, MFMA operands
// that sum over d) and hardware-transposed reads (ds_read_b64_tr_b16, operands that sum over
// the token index) with no bank conflicts and no second copy.
//
// All products are 16x16x32 bf16 MFMAs arranged so that the query (or key) index of a score
// tile sits on the LANE and the summed index in the registers: the score accumulators are then
// directly the B operand of the next product (k order permuted identically on both operands),
// and every per-row quantity (max, sum, LSE, delta) is lane-local.
#include "kernels.h"
#include "prof.h"

namespace {

constexpr int HD = 64;

// byte-free helpers on the swizzled [rows][64] bf16 image -------------------------------------
__device__ __forceinline__ bf16x8 row_frag(const bf16* img, int row, int chunk) {
    return *(const bf16x8*)(img + row * HD + ((chunk ^ (row & 7)) << 3));
}
// transposed fragment: 8 token rows {r0 + 4*fg + jj (jj<4), r0 + 16 + 4*fg + (jj-4)} of column
// (c0 + lane&15); lane (fr = lane&15, fg = lane>>4).  r0 multiple of 32, c0 multiple of 16.
__device__ __forceinline__ bf16x8 tr_frag(const bf16* img, int r0, int c0, int fr, int fg) {
    const int p = fr & 3;
    const int row = r0 + 4 * fg + (fr >> 2);
    const int chunk = (c0 >> 3) + (p >> 1);
    const bf16* a0 = img + row * HD + ((chunk ^ (row & 7)) << 3) + ((p & 1) << 2);
    const int row1 = row + 16;
    const bf16* a1 = img + row1 * HD + ((chunk ^ (row1 & 7)) << 3) + ((p & 1) << 2);
    return cat4(lds_read_tr16(a0), lds_read_tr16(a1));
}
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = f2bf(a[j]); o[4 + j] = f2bf(b[j]); }
    return o;
}
// stage rows [0, ROWS) x 64 of one head slice (row stride ld elements) into a swizzled image;
// rows >= T are zero.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_image(bf16* img, const bf16* src, int ld, int T, int tid) {
    for (int idx = tid; idx < ROWS * 8; idx += NT) {
        const int r = idx >> 3, c = idx & 7;
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16)0.f;
        if (r < T) v = *(const bf16x8*)(src + (size_t)r * ld + c * 8);
        *(bf16x8*)(img + r * HD + ((c ^ (r & 7)) << 3)) = v;
    }
}

// ------------------------------------------------------------------------------------------
// forward.  qkv [rows, 3*D] bf16 (q | k | v, head h at columns h*64), ctx [rows, D] bf16,
// lse2 [B*H*T] f32 = log2-sum-exp2 of the scaled scores (base-2 units).
// NKT = number of 16-key tiles (even), NKT*16 >= T.
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ ctx,
                                                          float* __restrict__ lse2, int T, int H, int D,
                                                          float scale_log2e) {
    constexpr int ROWS = NKT * 16;
    __shared__ __attribute__((aligned(16))) bf16 sK[ROWS * HD];
    __shared__ __attribute__((aligned(16))) bf16 sV[ROWS * HD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int ld = 3 * D;
    const bf16* base = qkv + (size_t)b * T * ld + h * HD;
    stage_image<ROWS, 256>(sK, base + D, ld, T, tid);
    stage_image<ROWS, 256>(sV, base + 2 * D, ld, T, tid);
    __syncthreads();

    const int nqb = (T + 15) >> 4;
    for (int qb = w; qb < nqb; qb += 4) {
        const int q = qb * 16 + fr;
        const int qc = q < T ? q : T - 1;
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + (ks * 4 + fg) * 8);

        // S^T tiles: rows = keys (4*fg + j inside tile kt), column = query fr
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma16(row_frag(sK, kt * 16 + fr, ks * 4 + fg), qf[ks], s[kt]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kt * 16 + 4 * fg + j;
                const float v = key < T ? s[kt][j] * scale_log2e : -INFINITY;
                s[kt][j] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = exp2f(s[kt][j] - mx);
                s[kt][j] = p;
                l += p;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);

        // O^T = V^T P^T : rows = d (nt*16 + 4*fg + j), column = query fr
        f32x4 o[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKT / 2; ++kk) {
            const bf16x8 pb = pack_pair(s[2 * kk], s[2 * kk + 1]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) o[nt] = mfma16(tr_frag(sV, kk * 32, nt * 16, fr, fg), pb, o[nt]);
        }
        const float inv = 1.f / l;
        if (q < T) {
            bf16* dst = ctx + ((size_t)b * T + q) * D + h * HD + 4 * fg;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bf16x4 ov = {f2bf(o[nt][0] * inv), f2bf(o[nt][1] * inv), f2bf(o[nt][2] * inv), f2bf(o[nt][3] * inv)};
                *(bf16x4*)(dst + nt * 16) = ov;
            }
            if (fg == 0) lse2[((size_t)b * H + h) * T + q] = mx + log2f(l);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward.  dctx [rows, D] bf16 -> dqkv [rows, 3*D] bf16.  P is recomputed from Q, K and the
// forward's LSE; delta = rowsum(dO * O) from the saved forward output.
// phase A (query on the lane): dQ^T = K^T dS^T ; phase B (key on the lane): dV^T = dO^T P,
// dK^T = Q^T dS.  26 work items (13 + 13 at T = 197) are dealt round-robin to 8 waves.
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(512) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ ctx,
                                                       const bf16* __restrict__ dctx, const float* __restrict__ lse2,
                                                       bf16* __restrict__ dqkv, int T, int H, int D, float scale,
                                                       float scale_log2e) {
    constexpr int ROWS = NKT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sQ = (bf16*)smem;
    bf16* sK = sQ + ROWS * HD;
    bf16* sV = sK + ROWS * HD;
    bf16* sdO = sV + ROWS * HD;
    float* sLse = (float*)(sdO + ROWS * HD);
    float* sDelta = sLse + ROWS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int ld = 3 * D;
    const bf16* base = qkv + (size_t)b * T * ld + h * HD;
    const bf16* dobase = dctx + (size_t)b * T * D + h * HD;
    const bf16* obase = ctx + (size_t)b * T * D + h * HD;
    stage_image<ROWS, 512>(sQ, base, ld, T, tid);
    stage_image<ROWS, 512>(sK, base + D, ld, T, tid);
    stage_image<ROWS, 512>(sV, base + 2 * D, ld, T, tid);
    // dO image + delta[r] = sum_d dO[r][d] * O[r][d]  (8 consecutive lanes share a row)
    for (int idx = tid; idx < ROWS * 8; idx += 512) {
        const int r = idx >> 3, c = idx & 7;
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16)0.f;
        float part = 0.f;
        if (r < T) {
            v = *(const bf16x8*)(dobase + (size_t)r * D + c * 8);
            const bf16x8 ov = *(const bf16x8*)(obase + (size_t)r * D + c * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) part += bf2f(v[k]) * bf2f(ov[k]);
        }
        *(bf16x8*)(sdO + r * HD + ((c ^ (r & 7)) << 3)) = v;
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (c == 0) {
            sDelta[r] = part;
            sLse[r] = r < T ? lse2[((size_t)b * H + h) * T + r] : INFINITY;
        }
    }
    __syncthreads();

    const int nblk = (T + 15) >> 4;
    // work items: [0, nblk) = phase B key blocks (heavier, dealt first), [nblk, 2*nblk) = phase A
    for (int item = w; item < 2 * nblk; item += 8) {
        if (item >= nblk) {
            // ---------------- phase A: query block on the lane ----------------
            const int qb = item - nblk;
            const int q = qb * 16 + fr;
            bf16x8 qf[2], dof[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[ks] = row_frag(sQ, q, ks * 4 + fg);
                dof[ks] = row_frag(sdO, q, ks * 4 + fg);
            }
            const float lse_q = sLse[q], delta_q = sDelta[q];
            f32x4 dq[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) dq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKT / 2; ++kk) {
                f32x4 ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int kt = 2 * kk + t;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        s = mfma16(row_frag(sK, kt * 16 + fr, ks * 4 + fg), qf[ks], s);      // S^T[key][q]
                        dp = mfma16(row_frag(sV, kt * 16 + fr, ks * 4 + fg), dof[ks], dp);   // dP^T[key][q]
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = kt * 16 + 4 * fg + j;
                        const float p = key < T ? exp2f(s[j] * scale_log2e - lse_q) : 0.f;
                        ds[t][j] = p * (dp[j] - delta_q) * scale;
                    }
                }
                const bf16x8 dsb = pack_pair(ds[0], ds[1]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) dq[nt] = mfma16(tr_frag(sK, kk * 32, nt * 16, fr, fg), dsb, dq[nt]);
            }
            if (q < T) {
                bf16* dst = dqkv + ((size_t)b * T + q) * ld + h * HD + 4 * fg;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bf16x4 ov = {f2bf(dq[nt][0]), f2bf(dq[nt][1]), f2bf(dq[nt][2]), f2bf(dq[nt][3])};
                    *(bf16x4*)(dst + nt * 16) = ov;
                }
            }
        } else {
            // ---------------- phase B: key block on the lane ----------------
            const int kb = item;
            const int key = kb * 16 + fr;
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[ks] = row_frag(sK, key, ks * 4 + fg);
                vf[ks] = row_frag(sV, key, ks * 4 + fg);
            }
            f32x4 dv[4], dk[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { dv[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[nt] = dv[nt]; }
#pragma unroll
            for (int kk = 0; kk < NKT / 2; ++kk) {
                f32x4 pt[2], ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int qt = 2 * kk + t;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        s = mfma16(row_frag(sQ, qt * 16 + fr, ks * 4 + fg), kf[ks], s);      // S[q][key]
                        dp = mfma16(row_frag(sdO, qt * 16 + fr, ks * 4 + fg), vf[ks], dp);   // dP[q][key]
                    }
                    const f32x4 lq = *(const f32x4*)(sLse + qt * 16 + 4 * fg);
                    const f32x4 dl = *(const f32x4*)(sDelta + qt * 16 + 4 * fg);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p = exp2f(s[j] * scale_log2e - lq[j]);   // rows >= T: lse = +inf -> 0
                        pt[t][j] = p;
                        ds[t][j] = p * (dp[j] - dl[j]) * scale;
                    }
                }
                const bf16x8 pb = pack_pair(pt[0], pt[1]);
                const bf16x8 dsb = pack_pair(ds[0], ds[1]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    dv[nt] = mfma16(tr_frag(sdO, kk * 32, nt * 16, fr, fg), pb, dv[nt]);   // dV^T[d][key]
                    dk[nt] = mfma16(tr_frag(sQ, kk * 32, nt * 16, fr, fg), dsb, dk[nt]);   // dK^T[d][key]
                }
            }
            if (key < T) {
                bf16* dst = dqkv + ((size_t)b * T + key) * ld + h * HD + 4 * fg;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bf16x4 kv = {f2bf(dk[nt][0]), f2bf(dk[nt][1]), f2bf(dk[nt][2]), f2bf(dk[nt][3])};
                    bf16x4 vv = {f2bf(dv[nt][0]), f2bf(dv[nt][1]), f2bf(dv[nt][2]), f2bf(dv[nt][3])};
                    *(bf16x4*)(dst + D + nt * 16) = kv;
                    *(bf16x4*)(dst + 2 * D + nt * 16) = vv;
                }
            }
        }
    }
}

template <int NKT>
void launch_bwd(const bf16* qkv, const bf16* ctx, const bf16* dctx, const float* lse2, bf16* dqkv, int B, int T, int H,
                int D, hipStream_t s) {
    const size_t lds = (size_t)4 * NKT * 16 * HD * sizeof(bf16) + (size_t)2 * NKT * 16 * sizeof(float);
    const float scale = 0.125f;   // 64^-1/2
    hipLaunchKernelGGL((attn_bwd_kernel<NKT>), dim3(B * H), dim3(512), lds, s, qkv, ctx, dctx, lse2, dqkv, T, H, D,
                       scale, scale * 1.4426950408889634f);
}

template <int NKT>
void bwd_attr() {
    const size_t lds = (size_t)4 * NKT * 16 * HD * sizeof(bf16) + (size_t)2 * NKT * 16 * sizeof(float);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

void attention_init() {
    static bool done = false;
    if (done) return;
    bwd_attr<2>();
    bwd_attr<14>();
    done = true;
}

int k_attention_fwd(const bf16* qkv, bf16* ctx, float* lse2, int B, int T, int H, int D, hipStream_t s) {
    ProfScope prof_("attn_fwd_kernel", 4.0 * B * H * (double)T * T * HD, (double)B * T * D * 8.0, s);
    const float sl = 0.125f * 1.4426950408889634f;
    if (T <= 32) hipLaunchKernelGGL((attn_fwd_kernel<2>), dim3(B * H), dim3(256), 0, s, qkv, ctx, lse2, T, H, D, sl);
    else if (T <= 224) hipLaunchKernelGGL((attn_fwd_kernel<14>), dim3(B * H), dim3(256), 0, s, qkv, ctx, lse2, T, H, D, sl);
    else return -1;
    return 0;
}

int k_attention_bwd(const bf16* qkv, const bf16* ctx, const bf16* dctx, const float* lse2, bf16* dqkv, int B, int T,
                    int H, int D, hipStream_t s) {
    ProfScope prof_("attn_bwd_kernel", 10.0 * B * H * (double)T * T * HD, (double)B * T * D * 16.0, s);
    if (T <= 32) launch_bwd<2>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, s);
    else if (T <= 224) launch_bwd<14>(qkv, ctx, dctx, lse2, dqkv, B, T, H, D, s);
    else return -1;
    return 0;
}
