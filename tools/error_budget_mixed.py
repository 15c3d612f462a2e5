#!/usr/bin/env python3
"""CPU experiment (round 5, verdict item 4 ii): can a MIXED bf16 mode -- bf16 only where an MFMA needs a bf16 operand, a wider
16-bit (fp16) or 32-bit type at the storage sites that are NOT MFMA operands -- bring the ViT-B / ViT-L input gradient under
north_star's 1e-2?  No kernel involved: the oracle's arithmetic with a round trip of the named type at every site where the HIP
path stores a tensor.

sites and who reads them on the HIP path
  MFMA operands (bf16 by definition of the mode): weights, lora_w, patches, h (LayerNorm output), qkv, probs, ctx, act (gelu),
      dz, t, and -- in the backward -- every gradient tensor a dgrad GEMM or the attention backward reads (dh, dqkv, dctx and the
      residual-gradient stream, which is the A operand of the o / fc2 dgrad);
  NOT operands: the forward residual stream (read by LayerNorm / the residual-add epilogue only), gelu'(z) (multiplied in the
      fc2-dgrad epilogue), LayerNorm statistics (fp32 already).

    python tools/error_budget_mixed.py [vitb|vitl]        (vitl: seeded random ViT-L/16, 24 layers, 2 images)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vit_lora_oracle as O  # noqa: E402

BF, HF, F32 = torch.bfloat16, torch.float16, torch.float32


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def rt(x, dt):
    return x if dt is F32 else x.to(dt).to(F32)


class Round(torch.autograd.Function):
    """value rounded to `fdt` in the forward, gradient rounded to `bdt` in the backward (F32 = untouched)"""

    @staticmethod
    def forward(ctx, x, fdt, bdt):
        ctx.bdt = bdt
        return rt(x, fdt)

    @staticmethod
    def backward(ctx, g):
        return rt(g, ctx.bdt), None, None


class Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, d_act, d_gp, d_dz):
        cdf = 0.5 * (1.0 + torch.erf(z * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * z * z)
        ctx.save_for_backward(rt(cdf + z * pdf, d_gp))
        ctx.d_dz = d_dz
        return rt(z * cdf, d_act)

    @staticmethod
    def backward(ctx, da):
        (gp,) = ctx.saved_tensors
        return rt(da * gp, ctx.d_dz), None, None, None


def forward(w, cfg, x_norm, lora, dt):
    """dt: site -> dtype.  `op` = MFMA operand sites (value and gradient), `gp` = saved gelu', `xf` = forward residual stream,
    `xb` = residual-gradient stream."""
    op, gp, xf, xb = dt["op"], dt["gp"], dt["xf"], dt["xb"]
    B = x_norm.shape[0]
    D, H, dh, N, P = cfg.hidden, cfg.heads, cfg.head_dim, cfg.tokens, cfg.patch_size
    g = cfg.image_size // P
    r = lambda t: Round.apply(t, op, op)
    patches = r(x_norm.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P))
    Wpe = w["vit.embeddings.patch_embeddings.projection.weight"].reshape(D, 3 * P * P)
    emb = F.linear(patches, rt(Wpe, op), w["vit.embeddings.patch_embeddings.projection.bias"])
    x = torch.cat([w["vit.embeddings.cls_token"].expand(B, -1, -1), emb], dim=1) + w["vit.embeddings.position_embeddings"]
    sc = lora.scaling if lora is not None else 0.0
    stream = lambda t: Round.apply(t, xf, xb)
    x = stream(x)
    for i in range(cfg.layers):
        p = f"vit.encoder.layer.{i}."

        def lin(short, inp):
            k = p + dict(O.LINEAR_MODULES)[short]
            y = F.linear(inp, rt(w[k + ".weight"], op), w[k + ".bias"])
            ab = lora.ab.get((i, short)) if lora is not None else None
            if ab is not None:
                A, Bm = ab
                y = y + F.linear(r(F.linear(inp, rt(A, op))), rt(Bm * sc, op))
            return y

        h = r(F.layer_norm(x, (D,), w[p + "layernorm_before.weight"], w[p + "layernorm_before.bias"], cfg.ln_eps))
        q, k_, v = (r(lin(s, h)).view(B, N, H, dh).transpose(1, 2) for s in ("q", "k", "v"))
        pr = r(torch.softmax(torch.matmul(q, k_.transpose(2, 3)) * (dh ** -0.5), dim=-1))
        ctx = r(torch.matmul(pr, v).transpose(1, 2).reshape(B, N, D))
        x = stream(x + lin("o", ctx))                       # the add happens in the GEMM epilogue: ONE rounding, of the stream
        h2 = r(F.layer_norm(x, (D,), w[p + "layernorm_after.weight"], w[p + "layernorm_after.bias"], cfg.ln_eps))
        a = Gelu.apply(lin("fc1", h2), op, gp, op)
        x = stream(x + lin("fc2", a))
    xfin = F.layer_norm(x[:, 0], (D,), w["vit.layernorm.weight"], w["vit.layernorm.bias"], cfg.ln_eps)
    return F.linear(xfin, w["classifier.weight"], w["classifier.bias"])


def loss_grad(w, cfg, x01, y, lora, dt, want_lora=False):
    x = x01.clone().requires_grad_(True)
    params = []
    if want_lora:
        for ab in lora.ab.values():
            for t in ab:
                t.requires_grad_(True)
                params.append(t)
    logits = forward(w, cfg, O.normalise(x), lora, dt)
    loss = F.cross_entropy(logits, y)
    gs = torch.autograd.grad(loss, [x] + params)
    for t in params:
        t.requires_grad_(False)
    ga = torch.cat([g.flatten() for g in gs[1:]]) if params else None
    return gs[0], logits.detach(), ga


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    torch.set_num_threads(8)
    if name == "vitl":
        cfg = O.OracleConfig(hidden=1024, layers=24, heads=16, mlp=4096, num_labels=21)
        r = 16
    else:
        cfg = O.OracleConfig(num_labels=21)
        r = 8
    w = O.init_weights(cfg, seed=31)
    lora = O.init_lora(cfg, r=r, targets=("q", "k", "v", "o", "fc2"), seed=32, b_std=0.02)
    g = torch.Generator().manual_seed(33)
    x = torch.rand(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 21, (2,), generator=g)
    ref = dict(op=F32, gp=F32, xf=F32, xb=F32)
    g0, lo0, a0 = loss_grad(w, cfg, x, y, lora, ref, True)
    print(f"case {name}: {cfg.layers} layers, hidden {cfg.hidden}, LoRA r = {r}; error against fp32 arithmetic (rel. L2)")
    rows = (
        ("fp16 mode as built (all 16-bit sites fp16)", dict(op=HF, gp=HF, xf=HF, xb=HF)),
        ("bf16 mode as built (all 16-bit sites bf16)", dict(op=BF, gp=BF, xf=BF, xb=BF)),
        ("mixed A: bf16 operands; gelu' + forward stream fp16", dict(op=BF, gp=HF, xf=HF, xb=BF)),
        ("mixed B: A + gradient stream fp32 (+ a bf16 shadow)", dict(op=BF, gp=HF, xf=HF, xb=F32)),
        ("mixed C: bf16 operands, every other site fp32", dict(op=BF, gp=F32, xf=F32, xb=F32)),
        ("only the MFMA operand sites bf16 == C", None),
    )
    for tag, dt in rows:
        if dt is None:
            continue
        gx, lo, ga = loss_grad(w, cfg, x, y, lora, dt, True)
        print(f"{tag:56s} logits {rel(lo, lo0):.2e}  dL/dx {rel(gx, g0):.2e}  LoRA grads {rel(ga, a0):.2e}", flush=True)


if __name__ == "__main__":
    main()
