#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from its results.db (rocpd sqlite), printed as
fractions of SQ_WAVE_CYCLES.  Usage: python tools/pmc_summary.py <results.db>"""
import collections
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
