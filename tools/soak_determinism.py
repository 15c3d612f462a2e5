#!/usr/bin/env python3
"""Soak: the same attack N times in ONE process must give the same pixels every time (races, uninitialised LDS and
mis-counted waits show up as rare mismatches).  Swin-T fp16 (persistent window kernels, streaming GEMM with the fused LoRA
down projection, one-pass merging) and ViT-B/16 fp16 (ring attention backward, fused PGD step), batch 256.
    python tools/soak_determinism.py [repeats]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P = importlib.import_module(PKG)
syn = importlib.import_module(PKG + ".synthetic")
swin = importlib.import_module(PKG + ".swin")
T = ("q", "k", "v", "o", "fc2")
g = torch.Generator().manual_seed(5)
x = torch.rand(256, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 21, (256,), generator=g).cuda()
bad = 0
# ---- Swin-T
from transformers import SwinConfig, SwinForImageClassification
torch.manual_seed(0)
hf = SwinForImageClassification(SwinConfig(num_labels=21))
se = swin.SwinEngine(swin.SwinArch(num_labels=21), lora_r=16, lora_alpha=16.0, lora_targets=T, device="cuda:0", precision="f16")
se.load_state_dict(hf.state_dict())
for si, d in enumerate((2, 2, 6, 2)):
    for bi in range(d):
        for t in T:
            A, Bm = se.param(si, bi, t, "A"), se.param(si, bi, t, "B")
            A.copy_((torch.rand(A.shape, generator=g) * 2 - 1) / A.shape[1] ** 0.5)
            Bm.copy_(torch.randn(Bm.shape, generator=g) * 0.02)
ref = se.pgd_attack(x, y, 8 / 255, 2 / 255, 3, random_start=True, seed=1).clone()
for i in range(N):
    out = se.pgd_attack(x, y, 8 / 255, 2 / 255, 3, random_start=True, seed=1)
    if not torch.equal(out, ref):
        bad += 1
        print(f"swin repeat {i}: {(out != ref).float().mean().item():.3e} of pixels differ", flush=True)
print(f"swin-t f16 PGD-3 x {N}: {bad} mismatches", flush=True)
del se, hf
# ---- ViT-B/16
arch = P.ArchConfig(num_labels=21)
eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=T), device="cuda:0")
eng.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, T, seed=1).items():
    eng.param(i, t, "A").copy_(A)
    eng.param(i, t, "B").copy_(Bm)
eng.commit()
ref = eng.pgd_attack(x, y, 8 / 255, 2 / 255, 5, True, seed=2).clone()
bad2 = 0
for i in range(N):
    out = eng.pgd_attack(x, y, 8 / 255, 2 / 255, 5, True, seed=2)
    if not torch.equal(out, ref):
        bad2 += 1
        print(f"vit repeat {i}: {(out != ref).float().mean().item():.3e} of pixels differ", flush=True)
print(f"vit-b/16 f16 PGD-5 x {N}: {bad2} mismatches", flush=True)
sys.exit(1 if bad + bad2 else 0)
