#!/usr/bin/env python3
"""CPU pricing (round 5, verdict item 1c) of folding the LayerNorm FORWARD into the consuming projection:

    y = rstd_r * (x' (gamma . W)^T - mu_r * c) + d,     c_n = sum_k round16(gamma_k W_nk),  d = b + W beta

i.e. the qkv / fc1 GEMMs read the 16-bit residual stream x' directly (operand W' = round16(gamma . W)) and apply the row statistics
in their epilogue; the normalised activation h is never formed or rounded.  Question: does the cancellation (x' carries the row mean
that the epilogue subtracts again) cost accuracy?  Everything else as the kernels store it today (oracle sim16 + 16-bit streams).

    python tools/error_budget_lnfold.py [vitb]        (12 layers, 2 images; ~1 min on 8 threads)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import vit_lora_oracle as O  # noqa: E402
import error_budget_mixed as EM  # noqa: E402

HF, F32 = torch.float16, torch.float32


def forward(w, cfg, x_norm, lora, fold, gain=1.0, shift=0.0):
    """fp16 storage at every site as the kernels have it; fold: LayerNorm forward inside the qkv / fc1 projections.
    gain / shift scale the LayerNorm weights and add a per-row offset to the stream (stress: large |mu / sigma|)."""
    B = x_norm.shape[0]
    D, H, dh, N, P = cfg.hidden, cfg.heads, cfg.head_dim, cfg.tokens, cfg.patch_size
    g = cfg.image_size // P
    r = lambda t: EM.Round.apply(t, HF, HF)
    patches = r(x_norm.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P))
    Wpe = w["vit.embeddings.patch_embeddings.projection.weight"].reshape(D, 3 * P * P)
    emb = F.linear(patches, EM.rt(Wpe, HF), w["vit.embeddings.patch_embeddings.projection.bias"])
    x = torch.cat([w["vit.embeddings.cls_token"].expand(B, -1, -1), emb], dim=1) + w["vit.embeddings.position_embeddings"] + shift
    sc = lora.scaling if lora is not None else 0.0
    x = r(x)
    for i in range(cfg.layers):
        p = f"vit.encoder.layer.{i}."

        def lin(short, inp):
            k = p + dict(O.LINEAR_MODULES)[short]
            y = F.linear(inp, EM.rt(w[k + ".weight"], HF), w[k + ".bias"])
            ab = lora.ab.get((i, short)) if lora is not None else None
            if ab is not None:
                A, Bm = ab
                y = y + F.linear(r(F.linear(inp, EM.rt(A, HF))), EM.rt(Bm * sc, HF))
            return y

        def lin_fold(short, xs, gam, bet, eps):
            """the folded form on the un-normalised stream xs (fp16 values): statistics in fp32, operand W' = round16(gamma . W)"""
            k = p + dict(O.LINEAR_MODULES)[short]
            W, b = w[k + ".weight"], w[k + ".bias"]
            mu = xs.mean(-1, keepdim=True)
            rstd = torch.rsqrt(((xs - mu) ** 2).mean(-1, keepdim=True) + eps)
            Wg = EM.rt(W * gam, HF)
            y = rstd * (F.linear(xs, Wg) - mu * Wg.sum(-1)) + (b + W @ bet)
            ab = lora.ab.get((i, short)) if lora is not None else None
            if ab is not None:
                A, Bm = ab
                Ag = EM.rt(A * gam, HF)
                t = r(rstd * (F.linear(xs, Ag) - mu * Ag.sum(-1)) + A @ bet)
                y = y + F.linear(t, EM.rt(Bm * sc, HF))
            return y

        g1, b1 = w[p + "layernorm_before.weight"] * gain, w[p + "layernorm_before.bias"]
        if fold:
            q, k_, v = (r(lin_fold(s, x, g1, b1, cfg.ln_eps)).view(B, N, H, dh).transpose(1, 2) for s in ("q", "k", "v"))
        else:
            h = r(F.layer_norm(x, (D,), g1, b1, cfg.ln_eps))
            q, k_, v = (r(lin(s, h)).view(B, N, H, dh).transpose(1, 2) for s in ("q", "k", "v"))
        pr = r(torch.softmax(torch.matmul(q, k_.transpose(2, 3)) * (dh ** -0.5), dim=-1))
        ctx = r(torch.matmul(pr, v).transpose(1, 2).reshape(B, N, D))
        x = r(x + lin("o", ctx))
        g2, b2 = w[p + "layernorm_after.weight"] * gain, w[p + "layernorm_after.bias"]
        if fold:
            a = EM.Gelu.apply(lin_fold("fc1", x, g2, b2, cfg.ln_eps), HF, HF, HF)
        else:
            a = EM.Gelu.apply(lin("fc1", r(F.layer_norm(x, (D,), g2, b2, cfg.ln_eps))), HF, HF, HF)
        x = r(x + lin("fc2", a))
    xf = F.layer_norm(x[:, 0], (D,), w["vit.layernorm.weight"], w["vit.layernorm.bias"], cfg.ln_eps)
    return F.linear(xf, w["classifier.weight"], w["classifier.bias"])


def ref_forward(w, cfg, x_norm, lora, gain, shift):
    """fp32 arithmetic of the same (stressed) network"""
    w2 = dict(w)
    for i in range(cfg.layers):
        for nm in ("layernorm_before.weight", "layernorm_after.weight"):
            w2[f"vit.encoder.layer.{i}.{nm}"] = w[f"vit.encoder.layer.{i}.{nm}"] * gain
    if shift:
        w2["vit.embeddings.position_embeddings"] = w["vit.embeddings.position_embeddings"] + shift
    return O.vit_forward(w2, cfg, x_norm, lora)


def run(w, cfg, x, y, lora, gain, shift):
    def lg(fwd):
        xx = x.clone().requires_grad_(True)
        logits = fwd(O.normalise(xx))
        (gx,) = torch.autograd.grad(F.cross_entropy(logits, y), xx)
        return logits.detach(), gx
    lo0, g0 = lg(lambda z: ref_forward(w, cfg, z, lora, gain, shift))
    for tag, fold in (("as built (LayerNorm kernel, h rounded to fp16)", False), ("folded into qkv / fc1 (no h)", True)):
        lo, gx = lg(lambda z: forward(w, cfg, z, lora, fold, gain, shift))
        print(f"  {tag:48s} logits {EM.rel(lo, lo0):.2e}  dL/dx {EM.rel(gx, g0):.2e}", flush=True)


def main():
    torch.set_num_threads(8)
    cfg = O.OracleConfig(num_labels=21)
    w = O.init_weights(cfg, seed=31)
    lora = O.init_lora(cfg, r=8, targets=("q", "k", "v", "o", "fc2"), seed=32, b_std=0.02)
    g = torch.Generator().manual_seed(33)
    x = torch.rand(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 21, (2,), generator=g)
    for gain, shift in ((1.0, 0.0), (1.0, 0.5), (1.0, 4.0)):
        print(f"ViT-B/16 + LoRA r=8, LayerNorm gain x{gain}, stream offset {shift} (|mu / sigma| of the first layers ~ {shift / 0.03:.0f}):")
        run(w, cfg, x, y, lora, gain, shift)


if __name__ == "__main__":
    main()
