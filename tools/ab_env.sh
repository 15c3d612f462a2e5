#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh "VAR=value" [pairs]   (inside gpurun; prints img/s and the kernel table diff)
set -e
SW="$1"; N=${2:-2}
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
for i in $(seq 1 $N); do
  timeout -k 10 200 $B > gpurun_out/abenv_A$i.json 2> /dev/null
  env $SW timeout -k 10 200 $B > gpurun_out/abenv_B$i.json 2> /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/abenv_[AB]*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels_ms_per_pgd_iteration"]
    print(f[-9:-5], round(d["value"], 1), {n: v for n, v in k.items() if v > 0.25})
PY
