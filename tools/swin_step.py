#!/usr/bin/env python3
"""One Swin-T + LoRA r=16 PGD attack at batch 256 (BASELINE config 4) for profiling: python tools/swin_step.py [steps] [precision]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
swin = importlib.import_module(PKG + ".swin")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
TARGETS = ("q", "k", "v", "o", "fc2")
kw = {"precision": prec} if prec != "f32" else {}
se = swin.SwinEngine(swin.SwinArch(num_labels=21), lora_r=16, lora_alpha=16.0, lora_targets=TARGETS, device="cuda:0", **kw)
from transformers import SwinConfig, SwinForImageClassification
torch.manual_seed(0)
hf = SwinForImageClassification(SwinConfig(num_labels=21))
se.load_state_dict(hf.state_dict())
g = torch.Generator().manual_seed(5)
for si, d in enumerate((2, 2, 6, 2)):
    for bi in range(d):
        for t in TARGETS:
            A, Bm = se.param(si, bi, t, "A"), se.param(si, bi, t, "B")
            A.copy_((torch.rand(A.shape, generator=g) * 2 - 1) / A.shape[1] ** 0.5)
            Bm.copy_(torch.randn(Bm.shape, generator=g) * 0.02)
x = torch.rand(256, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 21, (256,), generator=g).cuda()
se.pgd_attack(x, y, 8 / 255, 2 / 255, 1, random_start=True, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
se.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=True, seed=2)
torch.cuda.synchronize()
print(f"swin-t pgd step {1e3 * (time.perf_counter() - t0) / steps:.2f} ms at batch 256 ({prec})")
