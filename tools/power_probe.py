#!/usr/bin/env python3
"""Is the PGD iteration power / clock limited?  Samples the GPU's hwmon power and clock files (sysfs, readable by an
ordinary user) every 20 ms from a thread while one process runs: the full-batch attack, a half-batch attack on half the
chip's CUs, and bare GEMM loops.  Prints mean / max power and mean sclk per phase.  (MI355X_MICROARCH.md 'DVFS
give-back': the in-kernel clock is the real test; this is the cheap first look.)"""
import ctypes
import glob
import importlib
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
P = importlib.import_module(PKG)
syn = importlib.import_module(PKG + ".synthetic")


def find_hwmon():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            files = {os.path.basename(f) for f in glob.glob(hw + "/*")}
            out.append((card, hw, files))
    return out


hw = find_hwmon()
for card, h, files in hw:
    print(card, h, sorted(f for f in files if f.startswith(("power", "freq", "temp", "in")))[:40], flush=True)
for card, h, files in hw:
    for f in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "freq2_input"):
        try:
            print(card, f, open(os.path.join(h, f)).read().strip())
        except Exception as e:
            print(card, f, "unreadable", type(e).__name__)
    for f in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
        try:
            print(card, f, open(os.path.join(card, f)).read().strip().replace("\n", " | "))
        except Exception as e:
            print(card, f, "unreadable", type(e).__name__)

# the visible GPU: the card whose PCI address is HIP device 0's
PW = FR = None
props = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
print("HIP device 0 PCI", bus, flush=True)
for card, h, files in hw:
    if bus not in os.path.realpath(card):
        continue
    for f in ("power1_average", "power1_input"):
        if f in files:
            PW = os.path.join(h, f)
            break
    if "freq1_input" in files:
        FR = os.path.join(h, "freq1_input")
print("sampling", PW, FR, flush=True)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.on = True
        self.rows = []

    def run(self):
        while self.on:
            try:
                p = int(open(PW).read()) / 1e6 if PW else 0.0
                f = int(open(FR).read()) / 1e6 if FR else 0.0
            except Exception:
                p = f = -1.0
            self.rows.append((time.perf_counter(), p, f))
            time.sleep(0.02)


def measure(name, fn, seconds=6.0):
    s = Sampler()
    torch.cuda.synchronize()
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize()
        n += 1
    dt = time.perf_counter() - t0
    s.on = False
    s.join()
    rows = [r for r in s.rows if r[0] - t0 > 1.0]       # skip the ramp
    pw = [r[1] for r in rows]
    fr = [r[2] for r in rows]
    print(f"{name:28s} {n / dt:8.3f} calls/s  power mean {sum(pw) / max(1, len(pw)):7.1f} W max {max(pw or [0]):7.1f}  "
          f"sclk mean {sum(fr) / max(1, len(fr)):7.1f} MHz min {min(fr or [0]):7.1f}", flush=True)
    return n / dt


dev = torch.device("cuda", 0)
arch = P.ArchConfig(num_labels=21)
TARGETS = ("q", "k", "v", "o", "fc2")
EPS, ALPHA = 8 / 255, 2 / 255


def make_engine():
    eng = P.Engine(arch, P.LoraSpec(r=8, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, 8, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    return eng


x, y = syn.random_batch(arch, 256, seed=100)
x, y = x.to(dev), y.to(dev)
adv = torch.empty_like(x)
full = make_engine()
full.pgd_attack(x, y, EPS, ALPHA, 2, seed=1, out=adv)
half = make_engine()
P.check(half.lib.vl_debug_set_cus(half.h, 128), "cus")
xa, ya = x[:128].contiguous(), y[:128].contiguous()
adva = torch.empty_like(xa)
half.pgd_attack(xa, ya, EPS, ALPHA, 2, seed=1, out=adva)
P.check(half.lib.vl_debug_set_cus(None, 256), "cus")
torch.cuda.synchronize()

measure("idle (sleep)", lambda: time.sleep(0.05), 3.0)
measure("full PGD-5, batch 256", lambda: full.pgd_attack(x, y, EPS, ALPHA, 5, seed=2, out=adv))
measure("half PGD-5, b128 on 128 CUs", lambda: half.pgd_attack(xa, ya, EPS, ALPHA, 5, seed=2, out=adva))

big = torch.empty(256 * 3 * 224 * 224 * 8, device=dev)
b2, b3 = torch.rand_like(big), torch.rand_like(big)
big.copy_(b2)
measure("pgd_step stream (HBM-bound)", lambda: [full.pgd_step(big, b2, b3, EPS, ALPHA) for _ in range(20)])
lib = full.lib
ms = ctypes.c_float()


def gemm(M, N, K1, K2, epi, iters=40):
    def f():
        P.check(lib.vl_bench_gemm(M, N, K1, K2, epi, 128, iters, ctypes.byref(ms)), "bench_gemm")
    return f


M = 50432
for name, (N, K1, K2, epi) in {"qkv fwd store": (2304, 768, 64, 0), "fc1 gelu": (3072, 768, 0, 2), "fc1 no store": (3072, 768, 0, 7),
                                "fc2 dgrad gelu'": (3072, 768, 64, 3), "fc1 dgrad": (768, 3072, 0, 0),
                                "4096^3 no store": (0, 0, 0, 0)}.items():
    if name == "4096^3 no store":
        r = measure(name, gemm(4096, 4096, 4096, 0, 7, 60))
        print(f"   -> {2 * 4096 ** 3 / (ms.value * 1e-3) / 1e12:.0f} TFLOP/s ({ms.value * 1e3:.1f} us)")
        continue
    measure(name, gemm(M, N, K1, K2, epi))
    print(f"   -> {2.0 * M * N * (K1 + K2) / (ms.value * 1e-3) / 1e12:.0f} TFLOP/s ({ms.value * 1e3:.1f} us)", flush=True)
for cus in (128, 64):
    P.check(lib.vl_debug_set_cus(None, cus), "cus")
    measure(f"fc1 no store, M/{256 // cus}, {cus} CUs", gemm(M // (256 // cus) // 256 * 256, 3072, 768, 0, 7))
    mm = M // (256 // cus) // 256 * 256
    print(f"   -> {2.0 * mm * 3072 * 768 / (ms.value * 1e-3) / 1e12:.0f} TFLOP/s ({ms.value * 1e3:.1f} us) on {cus} CUs", flush=True)
P.check(lib.vl_debug_set_cus(None, 256), "cus")
