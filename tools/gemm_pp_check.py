#!/usr/bin/env python3
"""Ping-pong GEMM (csrc/gemm_pp.hip): self-check against the 128-row kernel, then a timing sweep against the 256-row
kernel at the path's shapes.  Usage on the GPU box:  python tools/gemm_pp_check.py [batch]"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
lib = importlib.import_module(PKG + "._lib").load()

EPI = {0: "store_h16", 1: "resid_f32", 2: "gelu", 3: "gelu_bwd", 6: "store_f32", 7: "none"}
if "--time-only" not in sys.argv:
    for (M, N, K1, K2) in [(128 * 50, 3072, 768, 64), (128 * 100, 768, 3072, 64), (128 * 110, 3072, 768, 0), (128 * 3, 768, 768, 64),
                           (128 * 37, 2304, 768, 64)]:
        for epi in (0, 1, 2, 3, 6):
            for mode in (0, 1, 2, 3):
                if mode >= 2 and (epi != 0 or K2 != 64):
                    continue
                d = C.c_float()
                rc = lib.vl_check_gemm(M, N, K1, K2, epi, mode, C.byref(d))
                print(f"check M={M} N={N} K={K1}+{K2} {EPI[epi]:10s} mode={mode} rc={rc} max|diff|={d.value:.3e}", flush=True)

B = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 256
M = (B * 197 + 255) // 256 * 256
shapes = [("qkv fwd", 2304, 768, 64, 0), ("o fwd", 768, 768, 64, 1), ("fc1 fwd", 3072, 768, 0, 2), ("fc2 fwd", 768, 3072, 64, 1),
          ("fc2 dgrad", 3072, 768, 64, 3), ("fc1 dgrad", 768, 3072, 0, 0), ("o dgrad", 768, 768, 64, 0), ("qkv dgrad", 768, 2304, 64, 0),
          ("nostore 3072x768", 3072, 768, 0, 7), ("nostore 768x3072", 768, 3072, 0, 7)]
tot = [0.0, 0.0]
for name, N, K1, K2, epi in shapes:
    row = []
    for mode in (0, 1):
        os.environ["VITLORA_GEMM_PP"] = str(mode)
        lib.vl_debug_set_gemm_pp(mode)
        ms = C.c_float()
        rc = lib.vl_bench_gemm(M, N, K1, K2, epi, 128, 20, C.byref(ms))
        row.append(ms.value)
        if not name.startswith("nostore"):
            tot[mode] += ms.value
    fl = 2.0 * M * N * (K1 + K2)
    print(f"{name:18s} N={N:5d} K={K1}+{K2:2d} {EPI[epi]:10s}  gemm256 {row[0] * 1e3:7.1f} us {fl / row[0] / 1e9:7.1f} TF   pp {row[1] * 1e3:7.1f} us {fl / row[1] / 1e9:7.1f} TF", flush=True)
print(f"layer total: gemm256 {tot[0] * 1e3:.1f} us   pp {tot[1] * 1e3:.1f} us")
lib.vl_debug_set_gemm_pp(2)
for name, N, K1, epi in [("fc2 fwd + down (fused)", 768, 3072, 200), ("qkv fwd + down (fused)", 2304, 768, 300)]:
    ms, ms_sk, ms_main = C.c_float(), C.c_float(), C.c_float()
    lib.vl_bench_gemm(M, N, K1, 64, epi, 128, 20, C.byref(ms))
    lib.vl_bench_gemm(M, 64, K1, 0, 0, 64, 20, C.byref(ms_sk))
    lib.vl_bench_gemm(M, N, K1, 64, 0, 128, 20, C.byref(ms_main))
    print(f"{name:24s} fused {ms.value * 1e3:7.1f} us   separate: skinny {ms_sk.value * 1e3:6.1f} + main {ms_main.value * 1e3:6.1f} = {(ms_sk.value + ms_main.value) * 1e3:7.1f} us", flush=True)
