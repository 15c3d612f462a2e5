#!/usr/bin/env python3
"""LoRA-down (N = 64) GEMM timings: HBM-bound streams of A [M,K]."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd._lib").load()
M = 50432
for K in (768, 2304, 3072):
    ms = C.c_float()
    rc = lib.vl_bench_gemm(M, 64, K, 0, 0, 64, 20, C.byref(ms))
    print(f"K={K:5d}: {ms.value*1e3:7.1f} us  {M*K*2/ms.value/1e9:6.2f} TB/s (A bytes only)", "" if rc == 0 else lib.vl_last_error())
