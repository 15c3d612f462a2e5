#!/bin/bash
# kernel trace of Swin-T fp16 PGD steps at batch 256: per-kernel totals per step.  usage (inside gpurun): tools/swin_trace.sh [steps] [stats.csv out]
set -e
ST=${1:-4}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/swin_trace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/swin_step.py $ST f16 > $OUT/run.txt 2> $OUT/log.txt
if [ -n "$2" ]; then cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $2; fi
python3 - $OUT $ST <<'PY'
import csv, glob, sys, collections
out, st = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the timed attack = the launches after the LAST pgd_init_kernel (the warm-up attack has its own)
last_init = max(i for i, r in enumerate(rows) if "pgd_init" in r[2])
seg = [r for r in rows[last_init + 1:] if "copyBuffer" not in r[2] and "fillBuffer" not in r[2]]
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("vl_f16::", "").split("(")[0][-70:]
    by[k][0] += 1; by[k][1] += e - s
tot = sum(t for c, t in by.values())
print(open(out + "/run.txt").read().strip())
print(f"kernels of the timed attack: {len(seg)} ({len(seg)/st:.0f} per step), busy {tot/1e6/st:.2f} ms per step, wall {(seg[-1][1]-seg[0][0])/1e6/st:.2f} ms per step")
for k, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:30]:
    print(f"  {t/1e6/st:7.3f} ms/step {c/st:6.1f} x {t/c/1e3:7.1f} us  {k}")
PY
rm -rf $OUT/kt
