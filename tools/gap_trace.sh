#!/bin/bash
# kernel trace of the attack at a small batch, graph replay as in production: sum of kernel durations against the wall
# time between the first and the last kernel of the timed iterations.  usage (inside gpurun): tools/gap_trace.sh <batch>
set -e
B=${1:-32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gap_b$B
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > $OUT/bench.json 2> $OUT/log.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the last 40 % of the trace is steady-state graph replay
n = len(rows)
seg = rows[int(n * 0.6):]
busy = sum(e - s for s, e, _ in seg)
wall = seg[-1][1] - seg[0][0]
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
pos = [g for g in gaps if g > 0]
print(f"kernels {len(seg)}  wall {wall/1e6:.3f} ms  busy {busy/1e6:.3f} ms  ({busy/wall:.3f})  mean gap {sum(pos)/max(1,len(pos))/1e3:.2f} us over {len(pos)} gaps, overlapping pairs {len(gaps)-len(pos)}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("vl_f16::", "").split("(")[0][-60:]
    by[k][0] += 1; by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:25]:
    print(f"  {t/1e6:8.3f} ms {c:6d} x {t/c/1e3:7.1f} us  {k}")
PY
rm -rf $OUT/kt
