#!/usr/bin/env python3
"""CPU experiment (round 4): what the LayerNorm-side storage choices cost against the fp32 reference, on top of
the 16-bit sites the kernels already have (oracle sim16):

  resid:f   the forward residual stream stored in 16 bits (rounded after every residual add)
  resid:b   the residual-GRADIENT stream stored in 16 bits (rounded after every LayerNorm backward)
  fold      LayerNorm gain / bias folded into the consuming projection (W' = W diag(gamma), b' = b + W beta);
            the kernel then stores the normalised row xn in 16 bits ONCE -- GEMM operand and, in the backward,
            the xhat of the LayerNorm gradient (instead of re-deriving it from the fp32 stream)

The backward runs on a per-image power-of-two scaled loss like the kernels (largest |dlogits| of an image in
[2^9, 2^10)), so fp16 underflow does not pollute the numbers.

    python tools/error_budget_streams.py [vitb|tiny197] [f16|bf16]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vit_lora_oracle as O  # noqa: E402
from test_oracle_golden import load_case  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def r16(x):
    return x.to(O.SIM_DTYPE).to(torch.float32)


class RoundF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class RoundB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return r16(g)


class LNFold(torch.autograd.Function):
    """xn = round16((x - mean) * rstd); backward with the ROUNDED xn (what the kernel saved)."""

    @staticmethod
    def forward(ctx, x, eps):
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        xn = r16((x - mean) * rstd)
        ctx.save_for_backward(xn, rstd)
        return xn

    @staticmethod
    def backward(ctx, g):
        xn, rstd = ctx.saved_tensors
        c1 = g.mean(-1, keepdim=True)
        c2 = (g * xn).mean(-1, keepdim=True)
        return rstd * (g - c1 - xn * c2), None


def forward(w, cfg, x_norm, lora, sim, resid_f, resid_b, fold):
    B = x_norm.shape[0]
    D, H, dh, N, P = cfg.hidden, cfg.heads, cfg.head_dim, cfg.tokens, cfg.patch_size
    g = cfg.image_size // P
    patches = x_norm.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    patches = O._rb(patches, sim, "patches")
    Wpe = w["vit.embeddings.patch_embeddings.projection.weight"].reshape(D, 3 * P * P)
    emb = F.linear(patches, O._wq(Wpe, sim), w["vit.embeddings.patch_embeddings.projection.bias"])
    x = torch.cat([w["vit.embeddings.cls_token"].expand(B, -1, -1), emb], dim=1)
    x = x + w["vit.embeddings.position_embeddings"]
    sc = lora.scaling if lora is not None else 0.0

    def stream(x):
        if resid_f:
            x = RoundF.apply(x)
        if resid_b:
            x = RoundB.apply(x)
        return x

    x = stream(x)
    for i in range(cfg.layers):
        p = f"vit.encoder.layer.{i}."

        def lin(short, inp, gam=None, bet=None):
            k = p + dict(O.LINEAR_MODULES)[short]
            W, b = w[k + ".weight"], w[k + ".bias"]
            ab = lora.ab.get((i, short)) if lora is not None else None
            if gam is None:
                return O.lora_linear(inp, W, b, ab, sc, sim)
            # folded: operands are rounded AFTER the fold; bias terms stay fp32
            y = F.linear(inp, r16(W * gam), b + W @ bet)
            if ab is not None:
                A, Bm = ab
                t = O._rb(F.linear(inp, r16(A * gam)) + A @ bet, sim, "t")
                y = y + F.linear(t, r16(Bm * sc))
            return y

        g1, b1 = w[p + "layernorm_before.weight"], w[p + "layernorm_before.bias"]
        if fold:
            h = O._rb(LNFold.apply(x, cfg.ln_eps), {"h:b"} if sim else False, "h")
            q, k_, v = (O._rb(lin(s, h, g1, b1), sim, "qkv").view(B, N, H, dh).transpose(1, 2) for s in ("q", "k", "v"))
        else:
            h = O._rb(F.layer_norm(x, (D,), g1, b1, cfg.ln_eps), sim, "h")
            q, k_, v = (O._rb(lin(s, h), sim, "qkv").view(B, N, H, dh).transpose(1, 2) for s in ("q", "k", "v"))
        s = torch.matmul(q, k_.transpose(2, 3)) * (dh ** -0.5)
        pr = O._rb(torch.softmax(s, dim=-1), sim, "probs")
        ctx = O._rb(torch.matmul(pr, v).transpose(1, 2).reshape(B, N, D), sim, "ctx")
        x = stream(x + O._rb(lin("o", ctx), sim, "delta"))
        g2, b2 = w[p + "layernorm_after.weight"], w[p + "layernorm_after.bias"]
        if fold:
            h2 = O._rb(LNFold.apply(x, cfg.ln_eps), {"h:b"} if sim else False, "h")
            a = O.gelu_sim(lin("fc1", h2, g2, b2), sim)
        else:
            h2 = O._rb(F.layer_norm(x, (D,), g2, b2, cfg.ln_eps), sim, "h")
            a = O.gelu_sim(lin("fc1", h2), sim)
        x = stream(x + O._rb(lin("fc2", a), sim, "delta"))
    xf = F.layer_norm(x[:, 0], (D,), w["vit.layernorm.weight"], w["vit.layernorm.bias"], cfg.ln_eps)
    return F.linear(xf, w["classifier.weight"], w["classifier.bias"])


def loss_grad(w, cfg, x01, y, lora, sim, resid_f=False, resid_b=False, fold=False, scale=True):
    x = x01.clone().requires_grad_(True)
    logits = forward(w, cfg, O.normalise(x), lora, sim, resid_f, resid_b, fold)
    with torch.no_grad():
        dl = (torch.softmax(logits, -1) - F.one_hot(y, logits.shape[-1])) / len(y)
        mx = dl.abs().amax(-1)
        s = torch.exp2(9 - torch.floor(torch.log2(mx))) if scale else torch.ones_like(mx)
    per = F.cross_entropy(logits, y, reduction="none") / len(y)
    (g,) = torch.autograd.grad((per * s).sum(), x)
    return g / s.view(-1, 1, 1, 1), logits.detach()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    if len(sys.argv) > 2 and sys.argv[2] == "bf16":
        O.SIM_DTYPE = torch.bfloat16
    torch.set_num_threads(8)
    cfg, w, x, y, z = load_case(name)
    lora = O.init_lora(cfg, r=8, seed=3, b_std=0.02)
    g0, lo0 = loss_grad(w, cfg, x, y, lora, False, scale=False)
    print(f"case {name}, storage {O.SIM_DTYPE}")
    for tag, kw in (("sim16 as the kernels are today", {}),
                    ("+ resid:b (16-bit gradient stream)", dict(resid_b=True)),
                    ("+ fold (xn saved in 16 bits)", dict(fold=True)),
                    ("+ resid:b + fold", dict(resid_b=True, fold=True)),
                    ("+ resid:f (16-bit forward stream)", dict(resid_f=True)),
                    ("+ resid:f + resid:b + fold", dict(resid_f=True, resid_b=True, fold=True))):
        g, lo = loss_grad(w, cfg, x, y, lora, True, **kw)
        print(f"{tag:42s} logits {rel(lo, lo0):.2e}  dL/dx {rel(g, g0):.2e}", flush=True)
    for tag, kw in (("ONLY resid:b", dict(resid_b=True)), ("ONLY fold", dict(fold=True)), ("ONLY resid:f", dict(resid_f=True))):
        g, lo = loss_grad(w, cfg, x, y, lora, False, **kw)
        print(f"{tag:42s} logits {rel(lo, lo0):.2e}  dL/dx {rel(g, g0):.2e}", flush=True)


if __name__ == "__main__":
    main()
