#!/usr/bin/env python3
"""Bitwise run-to-run determinism of the forward / backward chain at full size (diagnostic)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import PKG, pkg  # noqa: E402

P = pkg()
syn = importlib.import_module(PKG + ".synthetic")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
TARGETS = ("q", "k", "v", "o", "fc2")
arch = P.ArchConfig(num_labels=21)
eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS))
eng.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
    eng.param(i, t, "A").copy_(A)
    eng.param(i, t, "B").copy_(Bm)
x, y = syn.random_batch(arch, B, seed=100)
x, y = x.cuda(), y.cuda()


def snap():
    logits = eng.forward(x, normalise=True).clone()
    fw = {}
    for l in range(arch.layers):
        for what in ("qkv", "ctx", "z"):
            fw[(what, l)] = eng.debug_tensor(what, l).clone()
    for i in range(2 * arch.layers + 1):
        fw[("xs", i)] = eng.debug_tensor("xs", i).clone()
    eng.loss_ce(y)
    gx, _ = eng.backward(True, False, tuple(x.shape))
    return logits, fw, gx.clone()


ref = snap()
for it in range(4):
    cur = snap()
    bad = [k for k in ref[1] if not torch.equal(ref[1][k], cur[1][k])]
    print(f"run {it}: logits equal {torch.equal(ref[0], cur[0])}  fwd tensors differing {bad[:6]} ({len(bad)})  "
          f"grad equal {torch.equal(ref[2], cur[2])}  grad diff frac {(ref[2] != cur[2]).float().mean().item():.2e}", flush=True)
a = eng.pgd_attack(x, y, 8 / 255, 2 / 255, 3, random_start=False).clone()
for it in range(3):
    b = eng.pgd_attack(x, y, 8 / 255, 2 / 255, 3, random_start=False)
    print("pgd-3 equal", torch.equal(a, b), "diff frac", (a != b).float().mean().item(), flush=True)
os.environ["VITLORA_NO_GRAPH"] = "1"
