#!/usr/bin/env python3
"""Is the PGD iteration bound by time or by (XCD) energy?  Runs single PGD iterations (one hipGraph launch each) with an
idle gap of g ms between them (a one-thread spin kernel: no power) and reports the iteration time NET of the gap.  If the
net time falls as the gap grows, the chip is spending a power budget, and idle time buys clock."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
P = importlib.import_module(PKG)
syn = importlib.import_module(PKG + ".synthetic")
dev = torch.device("cuda", 0)
arch = P.ArchConfig(num_labels=21)
TARGETS = ("q", "k", "v", "o", "fc2")
EPS, ALPHA = 8 / 255, 2 / 255
eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev)
eng.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
    eng.param(i, t, "A").copy_(A)
    eng.param(i, t, "B").copy_(Bm)
eng.commit()
x, y = syn.random_batch(arch, 256, seed=100)
x, y = x.to(dev), y.to(dev)
adv = torch.empty_like(x)
eng.pgd_attack(x, y, EPS, ALPHA, 3, seed=1, out=adv)
torch.cuda.synchronize()

# calibrate the spin kernel
torch.cuda._sleep(1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(100_000_000)
torch.cuda.synchronize()
cyc_per_ms = 100_000_000 / ((time.perf_counter() - t0) * 1e3)
print(f"spin kernel: {cyc_per_ms:.0f} cycles per ms", flush=True)

N = 60
for rnd in range(2):
    for gap in (0.0, 1.0, 3.0, 6.0, 12.0, 30.0):
        cyc = int(gap * cyc_per_ms)
        # warm into the regime
        for _ in range(10):
            eng.pgd_attack(x, y, EPS, ALPHA, 1, random_start=False, out=adv)
            if cyc:
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            eng.pgd_attack(x, y, EPS, ALPHA, 1, random_start=False, out=adv)
            if cyc:
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / N
        print(f"round {rnd} gap {gap:5.1f} ms: {dt:7.3f} ms per iteration incl. gap, {dt - gap:7.3f} ms net", flush=True)
