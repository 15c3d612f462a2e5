#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs as
MI355X_MICROARCH.md prescribes), with the guide's gfx950 correction: FETCH_SIZE counts wide coalesced reads at
half their bytes -> x2; WRITE_SIZE is exact for 16-byte-per-lane stores.  Counter unit: KiB-like 'kilobytes' of
the rocprofv3 derived metric (value * 1024 bytes).
Usage: python tools/pmc_traffic.py <fetch results.db> <write results.db> <out.json> <out.txt>"""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for name, cn, value, disp in sqlite3.connect(db).execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cn != counter:
            continue
        k = name.replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)
        k = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)
        k = re.sub(r"ILi(\d+)ELi(\d+)EE.*$", r"<\1, \2>", k)
        acc[k] += value
        n[k].add(disp)
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out, lines = {}, []
for k in sorted(fetch, key=lambda k: -(fetch[k][0] * fetch[k][1])):
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    tot = (2.0 * f_kb + w_kb) * 1024.0
    out[k] = {"launches": n, "fetch_bytes_raw": f_kb * 1024.0, "fetch_bytes_x2": 2 * f_kb * 1024.0,
              "write_bytes": w_kb * 1024.0, "hbm_bytes_per_launch": tot}
    lines.append(f"{k:40s} launches {n:4d}  FETCH_SIZE {f_kb / 1024:9.1f} MB (x2 gfx950 correction = {2 * f_kb / 1024:9.1f} MB)"
                 f"  WRITE_SIZE {w_kb / 1024:9.1f} MB  -> HBM traffic/launch {tot / 1e6:9.1f} MB")
json.dump(out, open(sys.argv[3], "w"), indent=1)
open(sys.argv[4], "w").write(
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0 --pgd-steps 2 "
    "--no-cpu-baseline --no-roofline --no-extras` (batch 256)\nFETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide "
    "coalesced reads); averages per launch\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines[:14]))
