#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt through torch.matmul, fp16, fp32 accumulate) does on the path's shapes, next to this
library's kernels on the same box, same power cap: a yardstick for "how much matrix rate is left", not a product path.
    python tools/gemm_vs_library.py [batch] [iters]
Both sides: `iters` back-to-back launches between two HIP events after a warm-up; C = A W^T, A [M, K], W [N, K] K-contiguous
(the library GEMM carries no bias / epilogue, so ours is timed with the plain h16 store as well as with the fused epilogue)."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"
lib = importlib.import_module(PKG + "._lib").load()
lib.vl_bench_gemm.restype = C.c_int
lib.vl_bench_gemm.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M = (B * 197 + 255) // 256 * 256
shapes = [("qkv fwd", 2304, 768, 0), ("o fwd / o dgrad", 768, 768, 0), ("fc1 fwd", 3072, 768, 2), ("fc2 fwd / fc1 dgrad", 768, 3072, 0),
          ("fc2 dgrad", 3072, 768, 3), ("qkv dgrad", 768, 2304, 0)]
dev = torch.device("cuda:0")
print(f"M = {M} rows (batch {B} x 197, padded); {ITERS} launches per timing")
print(f"{'shape':22s} {'N':>5s} {'K':>5s} | {'hipBLASLt':>10s} | {'ours plain':>10s} | {'ours fused':>10s}   (TFLOP/s)")
for name, N, K, epi in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(ITERS):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * M * N * K
    t_lib = e0.elapsed_time(e1) / ITERS
    ms = C.c_float()
    ours = []
    for e in (0, epi):
        rc = lib.vl_bench_gemm(M, N, K, 0, e, 128, ITERS, C.byref(ms))
        ours.append(fl / ms.value / 1e9 if rc == 0 else float("nan"))
    print(f"{name:22s} {N:5d} {K:5d} | {fl / t_lib / 1e9:10.1f} | {ours[0]:10.1f} | {ours[1]:10.1f}")
    del a, w, out
