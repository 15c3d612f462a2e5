#!/usr/bin/env python3
"""CPU experiment: how much of the bf16 path's error against the fp32 reference each bf16 storage
site costs (oracle sim16 with ONE site on, and with all sites but one).  Guides which hops of the
HIP path stay fp32.  python tools/error_budget.py [vitb|tiny197] [lora]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vit_lora_oracle as O  # noqa: E402
from test_oracle_golden import load_case  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    with_lora = len(sys.argv) > 2
    torch.set_num_threads(8)
    cfg, w, x, y, z = load_case(name)
    lora = O.init_lora(cfg, r=8, seed=3, b_std=0.02) if with_lora else None
    if with_lora:
        l0, lg0, gr0 = O.lora_train_grads(w, cfg, O.normalise(x), y, lora)
    _, g0, lo0 = O.loss_and_input_grad(w, cfg, x, y, lora)

    def run(sim, tag):
        _, g, lo = O.loss_and_input_grad(w, cfg, x, y, lora, sim16=sim)
        msg = f"{tag:34s} logits {rel(lo, lo0):.2e}  dL/dx {rel(g, g0):.2e}"
        if with_lora:
            _, _, gr = O.lora_train_grads(w, cfg, O.normalise(x), y, lora, sim16=sim)
            ea = max(rel(gr[k], gr0[k]) for k in gr if k[0] == "A")
            eb = max(rel(gr[k], gr0[k]) for k in gr if k[0] == "B")
            msg += f"  max dA {ea:.2e}  max dB {eb:.2e}"
        print(msg, flush=True)

    run(True, "all sites")
    sites = []
    for s in O.SITES:
        if s in ("weights", "lora_w", "act", "gelu_prime", "dz", "patches"):
            sites.append(s)
        else:
            sites += [s + ":f", s + ":b"]
    for s in sites:
        run({s}, "only " + s)
    full = set(O.SITES)
    for s in ("delta:f", "delta:b", "h:b", "gelu_prime", "dz", "probs:b", "qkv:b", "ctx:b"):
        base = s.split(":")[0]
        keep = set(full) - {base}
        if ":" in s:
            keep.add(base + (":b" if s.endswith(":f") else ":f"))
        run(keep, "all but " + s)


if __name__ == "__main__":
    main()
