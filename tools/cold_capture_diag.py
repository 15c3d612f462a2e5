#!/usr/bin/env python3
"""Cold-capture divergence (round-2 VERDICT item 6): in ONE fresh process, capture the PGD iteration WITHOUT the eager first
iteration (VITLORA_COLD_CAPTURE=1), dump the captured graph's node list (VITLORA_GRAPH_DUMP), run the attack a few times
and snapshot every saved activation after each run; then force a second (warm) capture and diff the two node lists.

    python tools/cold_capture_diag.py <steps> <out_prefix>
"""
import importlib
import os
import sys

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prefix = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/cold"
os.environ["VITLORA_COLD_CAPTURE"] = "1"
os.environ["VITLORA_GRAPH_DUMP"] = prefix + "_nodes.txt"
if os.path.exists(prefix + "_nodes.txt"):
    os.remove(prefix + "_nodes.txt")

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
P = importlib.import_module(PKG)
syn = importlib.import_module(PKG + ".synthetic")
TARGETS = ("q", "k", "v", "o", "fc2")
arch = P.ArchConfig(num_labels=21)
eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS))
eng.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
    eng.param(i, t, "A").copy_(A)
    eng.param(i, t, "B").copy_(Bm)
eng.commit()
torch.cuda.synchronize()
x, y = syn.random_batch(arch, 256, seed=100)
x, y = x.cuda(), y.cuda()
L = arch.layers


def snap():
    """Saved activations of the LAST executed iteration, in forward order."""
    out = []
    out.append(("xs0", eng.debug_tensor("xs", 0).clone()))
    for l in range(L):
        out.append((f"qkv{l}", eng.debug_tensor("qkv", l).clone()))
        out.append((f"lse{l}", eng.debug_tensor("lse", l).clone()))
        out.append((f"ctx{l}", eng.debug_tensor("ctx", l).clone()))
        out.append((f"xs{2 * l + 1}", eng.debug_tensor("xs", 2 * l + 1).clone()))
        out.append((f"z{l}", eng.debug_tensor("z", l).clone()))
        out.append((f"xs{2 * l + 2}", eng.debug_tensor("xs", 2 * l + 2).clone()))
    # head and backward side (shared buffers: what the last kernels left, i.e. layer 0's backward)
    for n in ("logits", "dlogits", "loss_img", "gscale", "inv_gscale", "xhat", "rstd_f", "dz", "dh", "dctx", "dqkv", "u",
              "dres_h", "grad_img", "stage_adv"):        # (dres0 / dres1 exist in the fp32 mode only since the 16-bit gradient stream)
        out.append((n, eng.debug_tensor(n, 0).clone()))
    return out


def diff(a, b):
    bad = []
    for (n, u), (_, v) in zip(a, b):
        if not torch.equal(u.view(torch.int16 if u.dtype == torch.float16 else torch.int32),
                           v.view(torch.int16 if v.dtype == torch.float16 else torch.int32)):
            bad.append((n, float((u != v).float().mean())))
    return bad


advs, snaps = [], []
for it in range(4):
    a = eng.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
    torch.cuda.synchronize()
    advs.append(a)
    snaps.append(snap())
    print(f"attack {it}: captures {eng.counter('graph_captures')}  differs from attack 0 in {float((a != advs[0]).float().mean()):.4%} "
          f"of pixels, from previous in {float((a != advs[-2]).float().mean()) if it else 0.0:.4%}", flush=True)
for it in range(1, 4):
    d = diff(snaps[it - 1], snaps[it])
    print(f"saved activations, attack {it - 1} vs {it}: {len(d)} differ; first: {d[:6]}", flush=True)
d = diff(snaps[0], snaps[3])
print(f"saved activations, attack 0 (cold) vs 3: {len(d)} differ; first: {d[:6]}", flush=True)

# second capture, warm: dropping the graph cache (a changed normalisation drops it; set it back)
eng.set_normalization((0.5, 0.5, 0.5), (0.25, 0.25, 0.25))
eng.set_normalization((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
a = eng.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
torch.cuda.synchronize()
a2 = eng.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
torch.cuda.synchronize()
print(f"warm re-captured graph, second run vs its first: {float((a2 != a).float().mean()):.4%}", flush=True)
print(f"warm re-capture: captures {eng.counter('graph_captures')}  differs from attack 3 in {float((a != advs[3]).float().mean()):.4%}, "
      f"from attack 0 in {float((a != advs[0]).float().mean()):.4%}", flush=True)
# eager reference in the same process
os.environ["VITLORA_NO_GRAPH"] = "1"
e2 = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS))
e2.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
    e2.param(i, t, "A").copy_(A)
    e2.param(i, t, "B").copy_(Bm)
e2.commit()
ae = e2.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
torch.cuda.synchronize()
for it in range(4):
    print(f"eager vs attack {it}: {float((ae != advs[it]).float().mean()):.4%} of pixels differ", flush=True)

# node lists of the two captures
txt = open(prefix + "_nodes.txt").read().split("graph ")
lists = [[ln.split(" ", 1)[1] if " " in ln else ln for ln in g.strip().split("\n")[1:]] for g in txt if g.strip()]
print("captures dumped:", len(lists), "node counts:", [len(v) for v in lists])
if len(lists) >= 2:
    a, b = lists[0], lists[1]
    if a == b:
        print("cold and warm node lists are IDENTICAL (type, kernel name, grid, block, LDS)")
    else:
        import difflib
        for ln in difflib.unified_diff(a, b, "cold", "warm", lineterm="", n=1):
            print(ln)
