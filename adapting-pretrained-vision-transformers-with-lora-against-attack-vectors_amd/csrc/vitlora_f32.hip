// fp32 parity mode: forward / backward launch sequences with every activation in fp32 (f32_kernels.hip).
// Same model handle, same C ABI (vl_config.precision = VL_PREC_F32); the LoRA branch is computed as peft
// writes it -- t = dropout(x) A^T, y += (alpha / r) t B^T (train_loras.py:79-95) -- from the fp32 master
// parameters, unfused.  Reference arithmetic: HF modeling_vit.py:146-157 (embeddings), :164-189 (attention),
// :241-254 (MLP), :257-286 (layer), :385, :560-561 (final LN + head).
#include <algorithm>
#include <cstring>

#include "f32_kernels.h"
#include "model.h"

namespace VLNS {      // vl_f16 / vl_bf16: the 16-bit path is compiled once per operand type (common.h)

namespace {

GemmF32 gm(const float* A, int lda, const float* W, int ldw, int transW, int M, int N, int K, float* C, int ldc) {
    GemmF32 g;
    memset(&g, 0, sizeof g);
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.transW = transW;
    g.M = M; g.Mstore = M; g.N = N; g.K = K; g.alpha = 1.f; g.C = C; g.ldc = ldc;
    return g;
}

// y (+)= x W^T + b (+ R), then the module's LoRA updates  y[:, rows of slot] += s * (xb A^T) B^T
// t: [M, kext] buffer receiving the down projections (columns ext_off .. ext_off + r of each slot)
void linear_fwd(vl_model* m, const Linear& ln, const float* x, float* t, int M, float* y, int ldy, const float* R, int ldr,
                uint32_t stream_id, hipStream_t s) {
    GemmF32 g = gm(x, ln.in, ln.Wrun, ln.in, 0, M, ln.out, ln.in, y, ldy);
    g.bias = ln.bias; g.R = R; g.ldr = ldr;
    k_gemm_f32(g, s);
    if (ln.slots.empty() || m->cfg.lora_merged) return;
    const float* xb = x;
    if (vl_drop_on(m)) {
        k_mask_f32(m->ws.f_xd, x, (int64_t)M * ln.in, 0, m->drop_seed, stream_id, m->cfg.lora_dropout, s);
        xb = m->ws.f_xd;
    }
    for (const Slot& sl : ln.slots) {
        GemmF32 d = gm(xb, ln.in, m->flat + sl.a_off, sl.in, 0, M, m->r, sl.in, t + sl.ext_off, ln.kext);     // t = xb A^T
        k_gemm_f32(d, s);
        GemmF32 u = gm(t + sl.ext_off, ln.kext, m->flat + sl.b_off, m->r, 0, M, sl.out, m->r, y + sl.row_off, ldy);
        u.alpha = m->scaling; u.R = y + sl.row_off; u.ldr = ldy;                                                // y += s t B^T
        k_gemm_f32(u, s);
    }
}

// dx = dy W (+ LoRA: u = dy B, dx += s * mask * (u A)); u: [M, kext]
void linear_dgrad(vl_model* m, const Linear& ln, const float* dy, float* u, int M, float* dx, uint32_t stream_id,
                  hipStream_t s) {
    GemmF32 g = gm(dy, ln.out, ln.Wrun, ln.in, 1, M, ln.in, ln.out, dx, ln.in);      // W [out, in] read as [k][n]
    k_gemm_f32(g, s);
    if (ln.slots.empty() || m->cfg.lora_merged) return;
    for (const Slot& sl : ln.slots) {
        GemmF32 d = gm(dy + sl.row_off, ln.out, m->flat + sl.b_off, m->r, 1, M, m->r, sl.out, u + sl.ext_off, ln.kext);   // u = dy B
        k_gemm_f32(d, s);
        if (vl_drop_on(m)) {
            GemmF32 a = gm(u + sl.ext_off, ln.kext, m->flat + sl.a_off, sl.in, 1, M, sl.in, m->r, m->ws.f_tmp, sl.in);
            a.alpha = m->scaling;
            k_gemm_f32(a, s);
            k_mask_f32(dx, m->ws.f_tmp, (int64_t)M * sl.in, 1, m->drop_seed, stream_id, m->cfg.lora_dropout, s);
        } else {
            GemmF32 a = gm(u + sl.ext_off, ln.kext, m->flat + sl.a_off, sl.in, 1, M, sl.in, m->r, dx, ln.in);              // dx += s u A
            a.alpha = m->scaling; a.R = dx; a.ldr = ln.in;
            k_gemm_f32(a, s);
        }
    }
}

// dB[out, r] = s * dy^T t ; dA[r, in] = s * u^T xb      (sums over the M token rows, one writer per element)
void lora_wgrad(vl_model* m, const Linear& ln, const float* dy, const float* x, const float* t, const float* u, int M,
                float* flat_grad, uint32_t stream_id, hipStream_t s) {
    if (!flat_grad || ln.slots.empty()) return;
    const float* xb = x;
    if (vl_drop_on(m)) {
        k_mask_f32(m->ws.f_xd, x, (int64_t)M * ln.in, 0, m->drop_seed, stream_id, m->cfg.lora_dropout, s);
        xb = m->ws.f_xd;
    }
    for (const Slot& sl : ln.slots) {
        GemmF32 gb;
        memset(&gb, 0, sizeof gb);
        gb.A = dy + sl.row_off; gb.lda = ln.out; gb.transA = 1; gb.W = t + sl.ext_off; gb.ldw = ln.kext; gb.transW = 1;
        gb.M = sl.out; gb.Mstore = sl.out; gb.N = m->r; gb.K = M; gb.alpha = m->scaling; gb.C = flat_grad + sl.b_off; gb.ldc = m->r;
        k_gemm_f32(gb, s);
        GemmF32 ga;
        memset(&ga, 0, sizeof ga);
        ga.A = u + sl.ext_off; ga.lda = ln.kext; ga.transA = 1; ga.W = xb; ga.ldw = ln.in; ga.transW = 1;
        ga.M = m->r; ga.Mstore = m->r; ga.N = sl.in; ga.K = M; ga.alpha = m->scaling; ga.C = flat_grad + sl.a_off; ga.ldc = sl.in;
        k_gemm_f32(ga, s);
    }
}

}  // namespace

size_t f32_carve(vl_model* m, int B, int train, char* base, size_t off0) {
    Workspace& w = m->ws;
    const int D = m->D, L = m->L, MLP = m->MLP;
    const int64_t Mpad = w.Mpad, Mppad = w.Mppad;
    size_t off = off0;
    auto take = [&](size_t bytes) -> float* {
        char* p = base ? base + off : nullptr;
        off += (size_t)round_up((int64_t)bytes, 256);
        return (float*)p;
    };
    int kext_max = 64;
    for (int k = 0; k < 4; ++k) if (m->layers[0].lin[k].kext > kext_max) kext_max = m->layers[0].lin[k].kext;
    w.f_patches = take((size_t)Mppad * m->PK * 4);
    w.f_h1.resize(L); w.f_h2.resize(L); w.f_a.resize(L); w.f_qkv.resize(L); w.f_ctx.resize(L); w.f_z.resize(L);
    for (int k = 0; k < 4; ++k) w.f_t[k].resize(L);
    float* sh_h = train ? nullptr : take((size_t)Mpad * D * 4);
    float* sh_a = train ? nullptr : take((size_t)Mpad * MLP * 4);
    float* sh_t = train ? nullptr : take((size_t)Mpad * kext_max * 4);
    for (int l = 0; l < L; ++l) {
        w.f_h1[l] = train ? take((size_t)Mpad * D * 4) : sh_h;
        w.f_h2[l] = train ? take((size_t)Mpad * D * 4) : sh_h;
        w.f_a[l] = train ? take((size_t)Mpad * MLP * 4) : sh_a;
        w.f_qkv[l] = take((size_t)Mpad * 3 * D * 4);
        w.f_ctx[l] = take((size_t)Mpad * D * 4);
        w.f_z[l] = take((size_t)Mpad * MLP * 4);
        for (int k = 0; k < 4; ++k) w.f_t[k][l] = train ? take((size_t)Mpad * kext_max * 4) : sh_t;
    }
    w.f_dh = take((size_t)Mpad * D * 4);
    w.f_dctx = take((size_t)Mpad * D * 4);
    w.f_dqkv = take((size_t)Mpad * 3 * D * 4);
    w.f_dz = take((size_t)Mpad * MLP * 4);
    w.f_u = take((size_t)Mpad * kext_max * 4);
    // f_tmp holds a masked LoRA product [M, in <= MLP] AND the patch-embedding backward's [B * NP, PK] product before its
    // scatter: with an MLP narrower than PK = 3 P^2 (small test architectures) the second is the larger one (sized for the MLP
    // alone until round 4, the patch gradient then ran past the end of the workspace)
    w.f_tmp = take(std::max((size_t)Mpad * MLP, (size_t)Mppad * m->PK) * 4);
    w.f_xd = train ? take((size_t)Mpad * MLP * 4) : nullptr;
    return off;
}

int f32_forward(vl_model* m, const float* x, int B, int normalise, int train, hipStream_t s) {
    (void)train;
    Workspace& w = m->ws;
    const int D = m->D, L = m->L, T = m->T, MLP = m->MLP;
    const int M = B * T, Mp = B * m->NP;
    k_patch_gather_f32(x, w.f_patches, B, m->S, m->P, normalise, m->mean, m->stdv, s);
    // patch embedding: rows b*NP + pi -> token rows b*T + 1 + pi, + bias + position embedding.  Done per image
    // (NP rows each) so that the plain GEMM epilogue can add pos[1..] as its R operand.
    for (int b = 0; b < B; ++b) {
        GemmF32 g = gm(w.f_patches + (size_t)b * m->NP * m->PK, m->PK, m->Wpe_f32, m->PK, 0, m->NP, D, m->PK,
                       w.xs[0] + ((size_t)b * T + 1) * D, D);
        g.bias = m->bpe; g.R = m->pos + D; g.ldr = D;
        k_gemm_f32(g, s);
    }
    (void)Mp;
    k_cls_rows(w.xs[0], m->cls, m->pos, B, T, D, s);
    for (int l = 0; l < L; ++l) {
        Layer& ly = m->layers[l];
        k_ln_fwd_f32(w.xs[2 * l], w.f_h1[l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, ly.ln1_b, M, D, m->cfg.ln_eps, s);
        linear_fwd(m, ly.lin[LQKV], w.f_h1[l], w.f_t[LQKV][l], M, w.f_qkv[l], 3 * D, nullptr, 0, l * 4 + LQKV, s);
        if (k_attn_fwd_f32(w.f_qkv[l], w.f_ctx[l], w.lse[l], B, T, m->H, D, s)) return vl_fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        linear_fwd(m, ly.lin[LO], w.f_ctx[l], w.f_t[LO][l], M, w.xs[2 * l + 1], D, w.xs[2 * l], D, l * 4 + LO, s);
        k_ln_fwd_f32(w.xs[2 * l + 1], w.f_h2[l], w.mean[2 * l + 1], w.rstd[2 * l + 1], ly.ln2_g, ly.ln2_b, M, D, m->cfg.ln_eps, s);
        linear_fwd(m, ly.lin[LFC1], w.f_h2[l], w.f_t[LFC1][l], M, w.f_z[l], MLP, nullptr, 0, l * 4 + LFC1, s);
        k_gelu_fwd_f32(w.f_z[l], w.f_a[l], (int64_t)M * MLP, s);
        linear_fwd(m, ly.lin[LFC2], w.f_a[l], w.f_t[LFC2][l], M, w.xs[2 * l + 2], D, w.xs[2 * l + 1], D, l * 4 + LFC2, s);
    }
    k_head_fwd(w.xs[2 * L], B, T, D, m->C, m->cfg.ln_eps, m->lnf_g, m->lnf_b, m->flat + m->cls_w_off,
               m->flat + m->cls_b_off, w.xhat, w.xf, w.rstd_f, w.logits, s);
    return VL_OK;
}

int f32_backward(vl_model* m, float* grad_x, float* flat_grad, hipStream_t s) {
    Workspace& w = m->ws;
    const int B = m->cur_B, D = m->D, L = m->L, T = m->T, MLP = m->MLP;
    const int M = B * T;
    if (flat_grad) {
        HIPCHK(hipMemsetAsync(flat_grad, 0, (size_t)m->flat_n * sizeof(float), s));
        k_classifier_grad(w.dlogits, w.xf, B, D, m->C, flat_grad + m->cls_w_off, flat_grad + m->cls_b_off, s);
    }
    int cur = 0;
    k_zero(w.dres[0], (size_t)w.Mpad * D * sizeof(float), s);
    k_head_bwd(w.dlogits, nullptr, m->flat + m->cls_w_off, m->lnf_g, w.xhat, w.rstd_f, B, T, D, m->C, w.dres[0], nullptr, s);
    for (int l = L - 1; l >= 0; --l) {
        Layer& ly = m->layers[l];
        // MLP
        linear_dgrad(m, ly.lin[LFC2], w.dres[cur], w.f_u, M, w.f_dz, l * 4 + LFC2, s);                  // d(a)
        lora_wgrad(m, ly.lin[LFC2], w.dres[cur], w.f_a[l], w.f_t[LFC2][l], w.f_u, M, flat_grad, l * 4 + LFC2, s);
        k_gelu_bwd_f32(w.f_dz, w.f_z[l], (int64_t)M * MLP, s);                                         // d(z) = d(a) gelu'(z)
        linear_dgrad(m, ly.lin[LFC1], w.f_dz, w.f_u, M, w.f_dh, l * 4 + LFC1, s);
        lora_wgrad(m, ly.lin[LFC1], w.f_dz, w.f_h2[l], w.f_t[LFC1][l], w.f_u, M, flat_grad, l * 4 + LFC1, s);
        k_ln_bwd_f32(w.f_dh, w.xs[2 * l + 1], w.mean[2 * l + 1], w.rstd[2 * l + 1], ly.ln2_g, w.dres[cur], w.dres[cur ^ 1], M, D, s);
        cur ^= 1;
        // attention block
        linear_dgrad(m, ly.lin[LO], w.dres[cur], w.f_u, M, w.f_dctx, l * 4 + LO, s);
        lora_wgrad(m, ly.lin[LO], w.dres[cur], w.f_ctx[l], w.f_t[LO][l], w.f_u, M, flat_grad, l * 4 + LO, s);
        if (k_attn_bwd_f32(w.f_qkv[l], w.f_ctx[l], w.f_dctx, w.lse[l], w.f_dqkv, B, T, m->H, D, s))
            return vl_fail(VL_ERR_UNSUPPORTED, "attention: T > 224");
        linear_dgrad(m, ly.lin[LQKV], w.f_dqkv, w.f_u, M, w.f_dh, l * 4 + LQKV, s);
        lora_wgrad(m, ly.lin[LQKV], w.f_dqkv, w.f_h1[l], w.f_t[LQKV][l], w.f_u, M, flat_grad, l * 4 + LQKV, s);
        k_ln_bwd_f32(w.f_dh, w.xs[2 * l], w.mean[2 * l], w.rstd[2 * l], ly.ln1_g, w.dres[cur], w.dres[cur ^ 1], M, D, s);
        cur ^= 1;
    }
    if (grad_x) {
        // d(patches) = d(x0)[patch rows] Wpe, scattered back to NCHW with the chain rule of (x - mean) / std
        GemmF32 g = gm(w.dres[cur], D, m->Wpe_f32, m->PK, 1, B * m->NP, m->PK, D, w.f_tmp, m->PK);
        g.a_gather = 1; g.patches = m->NP;
        k_gemm_f32(g, s);
        float is[3];
        for (int c = 0; c < 3; ++c) is[c] = m->cur_norm ? 1.f / m->stdv[c] : 1.f;
        k_patch_scatter_f32(w.f_tmp, grad_x, B, m->S, m->P, is, s);
    }
    return VL_OK;
}

}  // namespace VLNS
