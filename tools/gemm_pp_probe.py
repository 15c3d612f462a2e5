#!/usr/bin/env python3
"""Timing experiments on the ping-pong GEMM (VITLORA_DEPHASE=101: no LDS-DMA after the prologue, 102: no waits on it;
results are invalid in both).  One process per setting because the knob is read at init."""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
lib = importlib.import_module(PKG + "._lib").load()
M = 50432
lib.vl_debug_set_gemm_pp(int(os.environ.get("PP", "1")))
SAFE = os.environ.get("VITLORA_DEPHASE", "0") in ("0", "102")     # 101 breaks the counted waits of the requested operands
for name, N, K1, K2, epi in [("nostore 3072x768", 3072, 768, 0, 7), ("nostore 768x3072", 768, 3072, 0, 7)] + ([
        ("gelu 3072x768", 3072, 768, 0, 2), ("store 3072x768", 3072, 768, 0, 0), ("store 2304x768", 2304, 768, 64, 0),
        ("store 768x3072", 768, 3072, 0, 0), ("resid 768x3072", 768, 3072, 64, 1), ("gelu_bwd", 3072, 768, 64, 3)] if SAFE else []):
    ms = C.c_float()
    lib.vl_bench_gemm(M, N, K1, K2, epi, 128, 20, C.byref(ms))
    fl = 2.0 * M * N * (K1 + K2)
    print(f"DEPHASE={os.environ.get('VITLORA_DEPHASE', '-'):4s} {name:18s} {ms.value * 1e3:7.1f} us {fl / ms.value / 1e9:7.1f} TF", flush=True)
