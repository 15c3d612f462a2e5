#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs as
MI355X_MICROARCH.md prescribes), with the guide's gfx950 correction: FETCH_SIZE counts wide coalesced reads at
half their bytes -> x2; WRITE_SIZE is exact for 16-byte-per-lane stores.  Counter unit: KiB-like 'kilobytes' of
the rocprofv3 derived metric (value * 1024 bytes).
Usage: python tools/pmc_traffic.py <fetch results.db> <write results.db> <out.json> <out.txt>"""
import collections
import hashlib
import json
import os
import re
import sqlite3
import sys


def kernel_source_sha16():
    """Fingerprint of the kernel sources the counters were taken on: bench.py reports `traffic` only while the
    sources still hash to this value (a stale profile is omitted, never quoted)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) or fn == "build.sh":        # build.sh: per-file compiler flags are part of the kernels
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]



def per_kernel(db, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for name, cn, value, disp in sqlite3.connect(db).execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cn != counter:
            continue
        k = name.replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)
        k = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)
        k = re.sub(r"^_ZN\d+vl_b?f16\d+_GLOBAL__N_1\d+", "", k)      # kernels of the two 16-bit builds (namespaces vl_f16 / vl_bf16)
        k = k.replace("vl_f16::", "").replace("vl_bf16::", "")
        k = re.sub(r"ILi(\d+)ELi(\d+)ELb(\d)EEEv.*$", r"<\1, \2, \3>", k)          # <int, int, bool>
        k = re.sub(r"ILi(\d+)EEEv.*$", r"<\1>", k)
        k = re.sub(r"(_kernel)E[Pv].*$", r"\1", k)
        k = re.sub(r"ILi(\d+)ELi(\d+)EE.*$", r"<\1, \2>", k)
        acc[k] += value
        n[k].add(disp)
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE") if __name__ == "__main__" else {}
write = per_kernel(sys.argv[2], "WRITE_SIZE") if __name__ == "__main__" else {}
out, lines = {"_meta": {"kernel_source_sha16": kernel_source_sha16()}}, []
for k in sorted(fetch, key=lambda k: -(fetch[k][0] * fetch[k][1])):
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    tot = (2.0 * f_kb + w_kb) * 1024.0
    out[k] = {"launches": n, "fetch_bytes_raw": f_kb * 1024.0, "fetch_bytes_x2": 2 * f_kb * 1024.0,
              "write_bytes": w_kb * 1024.0, "hbm_bytes_per_launch": tot}
    lines.append(f"{k:40s} launches {n:4d}  FETCH_SIZE {f_kb / 1024:9.1f} MB (x2 gfx950 correction = {2 * f_kb / 1024:9.1f} MB)"
                 f"  WRITE_SIZE {w_kb / 1024:9.1f} MB  -> HBM traffic/launch {tot / 1e6:9.1f} MB")
if __name__ == "__main__":
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    open(sys.argv[4], "w").write(
        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0 --pgd-steps 2 "
        "--no-cpu-baseline --no-roofline --no-extras` (batch 256); kernel sources sha16 " + out["_meta"]["kernel_source_sha16"] +
        "\nFETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide "
        "coalesced reads); averages per launch\n\n" + "\n".join(lines) + "\n")
    print("\n".join(lines[:14]))
