import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import PKG, pkg
P = pkg(); syn = importlib.import_module(PKG + ".synthetic")
TARGETS = ("q", "k", "v", "o", "fc2")
arch = P.ArchConfig(num_labels=21)
def mk():
    eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS))
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A); eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    return eng
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x, y = syn.random_batch(arch, Bn, seed=100); x, y = x.cuda(), y.cuda()
os.environ["VITLORA_NO_GRAPH"] = "1"
e = mk()
E = e.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
E2 = e.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
print("eager repeat equal", torch.equal(E, E2))
del e
os.environ.pop("VITLORA_NO_GRAPH")
g = mk()
for it in range(3):
    o = g.pgd_attack(x, y, 8 / 255, 2 / 255, steps, random_start=False).clone()
    torch.cuda.synchronize()
    print(it, "graph == eager", torch.equal(o, E), (o != E).float().mean().item(), "per-image diff frac (first 8):",
          [(o[b] != E[b]).float().mean().item() for b in range(min(8, Bn))], flush=True)
