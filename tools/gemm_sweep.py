#!/usr/bin/env python3
"""GEMM micro-benchmark at the path's shapes (random bf16 operands, HIP-event timing through
vl_bench_gemm).  Usage on the GPU box:  python tools/gemm_sweep.py [batch]"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
lib = importlib.import_module(PKG + "._lib").load()
lib.vl_bench_gemm.restype = C.c_int
lib.vl_bench_gemm.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None     # e.g. "qkv fwd,plain 3072x768"
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = B * 197
M = (M + 255) // 256 * 256
EPI = {0: "store_bf16", 1: "resid_f32", 2: "gelu", 3: "gelu_bwd", 6: "store_f32", 7: "none", 100: "bf16->row0", 102: "gelu->row0"}
shapes = [  # (name, N, K1, K2, epi)
    ("qkv fwd", 2304, 768, 64, 0), ("o fwd", 768, 768, 64, 1), ("fc1 fwd", 3072, 768, 0, 2),
    ("fc2 fwd", 768, 3072, 64, 1), ("fc2 dgrad", 3072, 768, 64, 3), ("fc1 dgrad", 768, 3072, 0, 0),
    ("o dgrad", 768, 768, 64, 0), ("qkv dgrad", 768, 2304, 64, 0),
    ("nostore 3072x768", 3072, 768, 0, 7), ("nostore 768x3072", 768, 3072, 0, 7), ("nostore 2304x768", 2304, 768, 0, 7),
    ("row0 3072x768", 3072, 768, 0, 100), ("row0gelu 3072x768", 3072, 768, 0, 102),
    ("plain 3072x768", 3072, 768, 0, 0), ("plain 768x3072", 768, 3072, 0, 0), ("square 4096^3", 4096, 4096, 0, 0),
]
tot_ms = tot_fl = 0.0
for name, N, K1, K2, epi in shapes:
    if ONLY and name not in ONLY:
        continue
    m = 4096 if name.startswith("square") else M
    ms = C.c_float()
    rc = lib.vl_bench_gemm(m, N, K1, K2, epi, 128, ITERS, C.byref(ms))
    if rc:
        print(name, "error", lib.vl_last_error().decode())
        continue
    fl = 2.0 * m * N * (K1 + K2)
    print(f"{name:16s} M={m} N={N:5d} K={K1}+{K2:2d} {EPI[epi]:10s} {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:7.1f} TFLOP/s")
    if not name.startswith(("plain", "square", "nostore", "row0")):
        tot_ms += ms.value
        tot_fl += fl
print(f"layer total (8 GEMMs): {tot_ms * 1e3:.1f} us, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s (padded-K flops)")
