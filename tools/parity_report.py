#!/usr/bin/env python3
"""Error report of the HIP path against (a) the reference's own outputs (tests/golden/*.npz: HF ViT logits and
input gradient) and (b) the fp32 oracle on seeded cases.  Run on the GPU box: python tools/parity_report.py"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import O, make_case, make_engine, rel_l2  # noqa: E402
import test_hip_facade as F  # noqa: E402


def main():
    for name in ("tiny17", "tiny197", "vitb"):
        cfg, w, x, y, z = F.load_case(name)
        lora = None
        eng = make_engine(cfg, w, lora)
        logits = eng.forward(x.cuda(), normalise=True).cpu()
        eng.loss_ce(y.cuda())
        gx, _ = eng.backward(True, False, tuple(x.shape))
        g_ref = torch.from_numpy(z["grad"])
        big = g_ref.abs() > 0.1 * g_ref.abs().mean()
        agree = (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item()
        print(f"golden {name:8s} logits rel_l2 {rel_l2(logits, torch.from_numpy(z['logits'])):.2e}   "
              f"input-grad rel_l2 {rel_l2(gx.cpu(), g_ref):.2e}   sign agreement (|g| > 0.1 mean) {agree:.4f}")
    for image_size, batch, r in ((64, 4, 8), (224, 3, 8), (224, 3, 0)):
        cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=r)
        eng = make_engine(cfg, w, lora)
        logits = eng.forward(x.cuda(), normalise=True).cpu()
        eng.loss_ce(y.cuda())
        gx, _ = eng.backward(True, False, tuple(x.shape))
        _, g_sim, l_sim = O.loss_and_input_grad(w, cfg, x, y, lora, sim_bf16=True)
        _, g_ref, l_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
        print(f"seeded img{image_size} b{batch} r{r}: logits vs fp32 {rel_l2(logits, l_ref):.2e} vs sim {rel_l2(logits, l_sim):.2e}   "
              f"grad vs fp32 {rel_l2(gx.cpu(), g_ref):.2e} vs sim {rel_l2(gx.cpu(), g_sim):.2e}")


if __name__ == "__main__":
    main()
