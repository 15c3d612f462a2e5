#!/usr/bin/env python3
"""Ring (single-pass) attention backward against the two-phase form on one engine: input gradient and LoRA gradients of a
train-mode forward, repeated; prints relative differences per repeat (both are fp16 kernels: expect ~1e-3)."""
import os, sys, importlib
os.environ["VITLORA_ATTN_IMG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import O, make_case, make_engine, rel_l2
for image_size, batch, layers in ((224, 3, 2), (64, 4, 2), (224, 64, 2)):
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8, layers=layers)
    eng = make_engine(cfg, w, lora)
    xn = O.normalise(x)
    outs = {}
    for rep in range(3):
        for ring in (0, 1):
            eng.set_option("attn_ring", ring)
            eng.forward(xn.cuda(), normalise=False, train=True)
            eng.loss_ce(y.cuda())
            gx, gp = eng.backward(True, True, tuple(x.shape))
            torch.cuda.synchronize()
            outs[ring] = (gx.cpu(), gp.cpu())
            u = eng.debug_tensor("u", 0).float().cpu()
            dq = eng.debug_tensor("dqkv", 0).float().cpu()
            outs[ring] += (u, dq)
        print(f"T={cfg.tokens} B={batch} rep {rep}: gx rel {rel_l2(outs[1][0], outs[0][0]):.2e}  lora-grad rel {rel_l2(outs[1][1], outs[0][1]):.2e}  "
              f"u(layer0) rel {rel_l2(outs[1][2], outs[0][2]):.2e}  dqkv(layer0) rel {rel_l2(outs[1][3], outs[0][3]):.2e}  finite {bool(torch.isfinite(outs[1][0]).all())}", flush=True)
    _, g_ref, _ = O.loss_and_input_grad(w, cfg, xn, y, lora, normalised=True)
    print("   vs oracle: ring", rel_l2(outs[1][0], g_ref), "two-phase", rel_l2(outs[0][0], g_ref), flush=True)
