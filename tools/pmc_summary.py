#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from its results.db (rocpd sqlite), printed as
fractions of SQ_WAVE_CYCLES.  Usage: python tools/pmc_summary.py <results.db> [out.json]

out.json (profiles/pmc_sq.json, read by bench.py for roofline.kernels[*].mfma_busy): per kernel
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES) * (32 / 1024) ... i.e. matrix-pipe busy cycles per SIMD over the
  kernel's duration: SQ_BUSY_CYCLES sums the busy cycles of the chip's 32 shader engines (8 XCDs x 4), so the duration is
  BUSY / 32 cycles, and there are 1024 SIMDs: mfma_busy = MFMA_BUSY / (BUSY / 32 * 1024) = MFMA_BUSY / (32 * BUSY).
  Cross-check with the per-wave form for kernels with W resident waves per SIMD: MFMA_BUSY / SQ_WAVE_CYCLES * W / 4
  (WAVE_CYCLES are quad-cycles, MI355X_MICROARCH.md) -- gemm_pp (W = 2): 1.044 * 2 / 4 = 0.52 vs 0.49 from the formula above."""
import collections
import json
import os
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for name, counter, value, disp in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*$", "", k)
    k = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)
    k = re.sub(r"^_ZN\d+vl_b?f16\d+_GLOBAL__N_1\d+", "", k)      # kernels of the two 16-bit builds (namespaces vl_f16 / vl_bf16)
    k = k.replace("vl_f16::", "").replace("vl_bf16::", "")
    k = re.sub(r"ILi(\d+)ELi(\d+)ELb(\d)EEEv.*$", r"<\1, \2, \3>", k)          # <int, int, bool>
    k = re.sub(r"ILi(\d+)EEEv.*$", r"<\1>", k)
    k = re.sub(r"ILi(\d+)ELi(\d+)EE.*$", r"<\1, \2>", k)
    k = re.sub(r"(_kernel)E[Pv].*$", r"\1", k)
    acc[k][counter] += value
    n[k].add(disp)
names = sorted({x for v in acc.values() for x in v})
base = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in names else names[0]
print("kernel".ljust(30), "n".rjust(4), *[x.replace("SQ_", "")[:14].rjust(14) for x in names], base.rjust(12))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get(base, 0))[:24]:
    wc = v[base] or 1.0
    print(k[:30].ljust(30), str(len(n[k])).rjust(4), *[f"{v.get(x, 0) / wc:14.3f}" for x in names], f"{wc:12.3g}")

if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pmc_traffic import kernel_source_sha16
    out = {"_meta": {"kernel_source_sha16": kernel_source_sha16(),
                     "mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): matrix-pipe busy fraction per SIMD"}}
    for k, v in acc.items():
        if v.get("SQ_BUSY_CYCLES"):
            out[k] = {"launches": len(n[k]), "mfma_busy": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * v["SQ_BUSY_CYCLES"]),
                      "wait_any": v.get("SQ_WAIT_ANY", 0.0) / (v.get("SQ_WAVE_CYCLES") or 1.0),
                      "active_inst_valu": v.get("SQ_ACTIVE_INST_VALU", 0.0) / (v.get("SQ_WAVE_CYCLES") or 1.0)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
