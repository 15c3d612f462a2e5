#!/usr/bin/env python3
"""Error report of the HIP path (all three precisions: python tools/parity_report.py [f16] [bf16] [f32]) against (a) the reference's own outputs (tests/golden/*.npz:
HF ViT logits and input gradient) and (b) the fp32 oracle on seeded cases incl. LoRA gradients.
Run on the GPU box: python tools/parity_report.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import O, make_case, make_engine, rel_l2  # noqa: E402
from test_oracle_golden import load_case  # noqa: E402


def lora_grad_errors(eng, gp, grads, cfg, lora):
    base = eng.flat.data_ptr()
    ea = eb = 0.0
    for i in range(cfg.layers):
        for t in lora.targets:
            for which in ("A", "B"):
                v = eng.param(i, t, which)
                off = (v.data_ptr() - base) // 4
                got = gp[off:off + v.numel()].view(v.shape).cpu()
                e = rel_l2(got, grads[(which, i, t)])
                if which == "A":
                    ea = max(ea, e)
                else:
                    eb = max(eb, e)
    return ea, eb


def main():
    torch.set_num_threads(16)
    precs = sys.argv[1:] or ["f16", "bf16", "f32"]
    for prec in precs:
        for name in ("tiny17", "tiny197", "vitb"):
            cfg, w, x, y, z = load_case(name)
            eng = make_engine(cfg, w, None, precision=prec)
            logits = eng.forward(x.cuda(), normalise=True).cpu()
            loss = eng.loss_ce(y.cuda()).item()
            gx, _ = eng.backward(True, False, tuple(x.shape))
            g_ref = torch.from_numpy(z["grad"])
            big = g_ref.abs() > 0.1 * g_ref.abs().mean()
            agree = (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item()
            agree_all = (torch.sign(gx.cpu()) == torch.sign(g_ref)).float().mean().item()
            print(f"[{prec}] golden {name:8s} logits {rel_l2(logits, torch.from_numpy(z['logits'])):.2e}  loss {abs(loss - float(z['loss'])):.1e}  "
                  f"input-grad {rel_l2(gx.cpu(), g_ref):.2e}  sign agree big {agree:.5f} all {agree_all:.5f}", flush=True)
            del eng
        for image_size, batch, r, kw in ((64, 4, 8, {}), (224, 3, 8, {}), (224, 2, 8, dict(hidden=768, heads=12, mlp=3072, layers=12, std=0.02, b_std=0.02))):
            cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=r, targets=("q", "k", "v", "o", "fc2"), **kw)
            eng = make_engine(cfg, w, lora, precision=prec)
            xn = O.normalise(x)
            logits = eng.forward(xn.cuda(), normalise=False, train=True).cpu()
            eng.loss_ce(y.cuda())
            gx, gp = eng.backward(True, True, tuple(x.shape))
            l_ref, lg_ref, grads = O.lora_train_grads(w, cfg, xn, y, lora)
            _, g_ref, _ = O.loss_and_input_grad(w, cfg, xn, y, lora, normalised=True)
            ea, eb = lora_grad_errors(eng, gp, grads, cfg, lora)
            print(f"[{prec}] seeded img{image_size} L{cfg.layers} D{cfg.hidden} b{batch} r{r}: logits {rel_l2(logits, lg_ref):.2e}  "
                  f"dL/dx {rel_l2(gx.cpu(), g_ref):.2e}  max dA {ea:.2e}  max dB {eb:.2e}", flush=True)
            del eng


def pgd_agreement(precs):
    """G4: the reference-driven ViT-B PGD trajectories (tests/golden/pgd_vitb.npz): pixels identical after k steps."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "pgd_vitb.npz"))
    cfg, w, x, y, _ = load_case("vitb")
    for prec in precs:
        eng = make_engine(cfg, w, None, precision=prec)
        out = []
        for k in [int(v) for v in z["steps"]]:
            adv = eng.pgd_attack(x.cuda(), y.cuda(), float(z["eps"]), float(z["alpha"]), k, random_start=False).cpu()
            same = ((adv - (x + torch.from_numpy(z[f"delta_x0_{k}"]))).abs() < 1e-6).float().mean().item()
            out.append(f"PGD-{k} {same:.4f}")
        print(f"[{prec}] G4 ViT-B pixels identical to the reference-driven trajectory: " + "  ".join(out), flush=True)
        del eng


if __name__ == "__main__":
    main()
    pgd_agreement(sys.argv[1:] or ["f16", "bf16", "f32"])
