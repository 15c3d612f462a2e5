#!/usr/bin/env python3
"""Schedule experiment (round 3): does the PGD iteration gain from running as TWO independent half-batch chains
that overlap each other's HBM-bound phases (LayerNorm, GEMM epilogues) with MFMA main loops?

Arms, interleaved in one process on one box (cdna_hip_programming.md rule 24):
  full      one engine, batch 256, persistent GEMM grids of 256 workgroups            (the shipped schedule)
  dual128   two engines, batch 128 each, GEMM grids of 128, on two streams concurrently
  dual256   two engines, batch 128 each, GEMM grids of 256 (oversubscribed), concurrently
  half128   one of the dual128 engines alone (what half the chip does with half the batch)
  half256   one of the dual256 engines alone
Prints img/s per arm and round.
"""
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
EPS, ALPHA, STEPS = 8 / 255, 2 / 255, int(os.environ.get("PGD_STEPS", 20))
TARGETS = ("q", "k", "v", "o", "fc2")
dev = torch.device("cuda", 0)
arch = P.ArchConfig(num_labels=21)


def make_engine():
    eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    return eng


def set_cus(eng, n):
    P.check(eng.lib.vl_debug_set_cus(eng.h, n), "vl_debug_set_cus")


x, y = syn.random_batch(arch, 256, seed=100)
x, y = x.to(dev), y.to(dev)
xa, ya, xb, yb = x[:128].contiguous(), y[:128].contiguous(), x[128:].contiguous(), y[128:].contiguous()
adv = torch.empty_like(x)
adva, advb = torch.empty_like(xa), torch.empty_like(xb)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

full = make_engine()
set_cus(full, 256)
full.pgd_attack(x, y, EPS, ALPHA, 2, seed=1, out=adv)
torch.cuda.synchronize()

d128 = [make_engine(), make_engine()]
for e in d128:
    set_cus(e, 128)
for e, s, (xx, yy, aa) in zip(d128, (sA, sB), ((xa, ya, adva), (xb, yb, advb))):
    with torch.cuda.stream(s):
        e.pgd_attack(xx, yy, EPS, ALPHA, 2, seed=1, out=aa)      # captures the graph with grids of 128
torch.cuda.synchronize()

d256 = [make_engine(), make_engine()]
for e in d256:
    set_cus(e, 256)
    e.num_cus_forced = True
for e, s, (xx, yy, aa) in zip(d256, (sA, sB), ((xa, ya, adva), (xb, yb, advb))):
    # per-image attention form as in the 128 arm (the batch threshold looks at the handle's CU count)
    P.check(e.lib.vl_debug_set_cus(e.h, 128), "set")
    P.check(e.lib.vl_debug_set_cus(None, 256), "set")
    with torch.cuda.stream(s):
        e.pgd_attack(xx, yy, EPS, ALPHA, 2, seed=1, out=aa)
torch.cuda.synchronize()


def t_full():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full.pgd_attack(x, y, EPS, ALPHA, STEPS, seed=2, out=adv)
    torch.cuda.synchronize()
    return 256 / (time.perf_counter() - t0)


def t_dual(engs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sA):
        engs[0].pgd_attack(xa, ya, EPS, ALPHA, STEPS, seed=2, out=adva)
    with torch.cuda.stream(sB):
        engs[1].pgd_attack(xb, yb, EPS, ALPHA, STEPS, seed=2, out=advb)
    torch.cuda.synchronize()
    return 256 / (time.perf_counter() - t0)


def t_half(eng):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sA):
        eng.pgd_attack(xa, ya, EPS, ALPHA, STEPS, seed=2, out=adva)
    torch.cuda.synchronize()
    return 128 / (time.perf_counter() - t0)


for rnd in range(int(os.environ.get("ROUNDS", 4))):
    r = {"full": t_full(), "dual128": t_dual(d128), "dual256": t_dual(d256), "half128": t_half(d128[0]),
         "half256": t_half(d256[0])}
    print("round", rnd, " ".join(f"{k}={v:.1f}" for k, v in r.items()), flush=True)

# same-result check: the two halves of the dual arm against the full-batch attack (per-image results do not depend on the
# batch they ride in, up to the 1/B rounding; report the fraction of identical pixels)
full.pgd_attack(x, y, EPS, ALPHA, STEPS, seed=2, random_start=False, out=adv)
with torch.cuda.stream(sA):
    d128[0].pgd_attack(xa, ya, EPS, ALPHA, STEPS, seed=2, random_start=False, out=adva)
torch.cuda.synchronize()
print("identical pixels full[:128] vs half:", float((adv[:128] == adva).float().mean()))
