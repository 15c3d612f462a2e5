#!/usr/bin/env python3
"""Soak test of the ping-pong GEMM's synchronisation: the same self-check (full output compared with the 128-row kernel)
repeated many times over the path's shapes, with and without the fused LoRA down projection.  Any lost wait or barrier
shows as a nonzero difference.  Usage on the GPU box:  python tools/gemm_pp_soak.py [rounds]"""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
lib = importlib.import_module(PKG + "._lib").load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [(50432, 2304, 768, 64, 0, 1), (50432, 768, 768, 64, 0, 1), (50432, 768, 2304, 64, 0, 1), (50432, 768, 3072, 64, 0, 2),
         (50432, 2304, 768, 64, 0, 3), (50432, 3072, 768, 0, 2, 1), (50432, 3072, 768, 64, 3, 1), (25216, 768, 3072, 64, 0, 2),
         (128 * 5, 768, 768, 64, 0, 1), (128 * 513, 768, 768, 64, 0, 2),
         # round 4: the residual-add epilogue (two-slot operand ring) plain and with the LoRA down projection + bias column inside
         (50432, 768, 768, 64, 10, 1), (50432, 768, 3072, 64, 10, 2), (50432, 3072, 768, 64, 3, 1), (12800, 2304, 768, 64, 0, 3)]
bad = 0
t0 = time.time()
for r in range(rounds):
    for (M, N, K1, K2, epi, mode) in cases:
        d = C.c_float(-1.0)
        rc = lib.vl_check_gemm(M, N, K1, K2, epi, mode, C.byref(d))
        lim = 4e-3 if epi == 2 else (0.04 if (epi == 10 and mode >= 2) else 0.0)      # bias inside the last MFMA step: <= 1 fp16 ulp
        if rc or not (d.value <= lim):
            bad += 1
            print(f"MISMATCH round {r} M={M} N={N} K={K1}+{K2} epi={epi} mode={mode} rc={rc} diff={d.value}", flush=True)
    print(f"round {r} done, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("soak:", "FAILED" if bad else "ok", rounds * len(cases), "checks")
sys.exit(1 if bad else 0)
