#!/usr/bin/env python3
"""Soak (round 5): the two gradients that used float atomics until round 4 -- the LoRA / classifier gradient of a train step
(csrc/lora_grad.hip: per-chunk partial blocks summed in chunk order) and the adversarial-patch gradient (csrc/patch.hip: 64-bit
fixed point, integer atomics) -- must come out bit-identical N times in one process, in fp16 and bf16, at BASELINE sizes
(ViT-B/16 + LoRA r = 8, 64 images: config 3's per-GPU batch; 32 x 32 circular patch on 64 images with random transforms).
    python tools/soak_gradients.py [repeats]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P = importlib.import_module(PKG)
syn = importlib.import_module(PKG + ".synthetic")
patch_mod = importlib.import_module(PKG + ".patch")
T = ("q", "k", "v", "o", "fc2")
g = torch.Generator().manual_seed(5)
x = torch.rand(64, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 21, (64,), generator=g).cuda()
bad = 0
for prec in ("f16", "bf16"):
    arch = P.ArchConfig(num_labels=21)
    eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.1, targets=T), device="cuda:0", precision=prec)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, 8, T, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    eng.lib.vl_set_dropout_seed(eng.h, 77)
    ref = None
    for i in range(N):
        eng.lib.vl_set_dropout_seed(eng.h, 77)              # same LoRA-dropout masks every repeat
        eng.forward(x, normalise=True, train=True)
        eng.loss_ce(y)
        _, gp = eng.backward(False, True)
        if ref is None:
            ref = gp.clone()
        elif not torch.equal(gp, ref):
            bad += 1
            print(f"{prec} LoRA gradient repeat {i}: {(gp != ref).float().mean().item():.3e} of elements differ", flush=True)
    print(f"vit-b/16 {prec} LoRA / classifier gradient x {N} (batch 64, dropout 0.1, fixed seed): " + ("bit-identical" if not bad else "MISMATCH"), flush=True)
    # patch gradient: forward through the eval path, backward to pixels, pull back onto the patch
    params = [(0.3 + 0.01 * k, -20.0 + 0.7 * k, 30.0 - k, -25.0 + 0.9 * k) for k in range(64)]
    mats = torch.tensor([patch_mod.inverse_affine_matrix(p[1], (p[2], p[3]), p[0]) for p in params], dtype=torch.float32, device="cuda:0")
    patch = torch.rand(3, 32, 32, generator=g).cuda()
    refp = None
    for i in range(N):
        patched = eng.patch_apply(x, patch, mats, 1)
        eng.forward(patched, normalise=True, train=False)
        eng.loss_ce(y)
        gx, _ = eng.backward(True, False, (64, 3, 224, 224))
        dp = eng.patch_grad(gx, mats, 32, 1)
        if refp is None:
            refp = dp.clone()
            assert torch.isfinite(refp).all() and refp.abs().max().item() > 0
        elif not torch.equal(dp, refp):
            bad += 1
            print(f"{prec} patch gradient repeat {i}: {(dp != refp).float().mean().item():.3e} of elements differ", flush=True)
    print(f"vit-b/16 {prec} patch gradient x {N} (64 images, 32 x 32 circle, per-image transforms): " + ("bit-identical" if not bad else "MISMATCH"), flush=True)
    del eng
sys.exit(1 if bad else 0)
