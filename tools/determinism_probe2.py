import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import PKG, pkg
P = pkg(); syn = importlib.import_module(PKG + ".synthetic")
TARGETS = ("q", "k", "v", "o", "fc2")
arch = P.ArchConfig(num_labels=21)
eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS))
eng.load_state_dict(syn.random_state_dict(arch, seed=0))
for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
    eng.param(i, t, "A").copy_(A); eng.param(i, t, "B").copy_(Bm)
eng.commit()
x, y = syn.random_batch(arch, 256, seed=100); x, y = x.cuda(), y.cuda()
outs = []
for it in range(4):
    outs.append(eng.pgd_attack(x, y, 8 / 255, 2 / 255, 3, random_start=False).clone())
    torch.cuda.synchronize()
    print(it, "captures", eng.counter("graph_captures"), "commits", eng.counter("commits"), "equal to first", torch.equal(outs[0], outs[-1]),
          "equal to prev", torch.equal(outs[-2], outs[-1]) if it else None, (outs[0] != outs[-1]).float().mean().item(), flush=True)
