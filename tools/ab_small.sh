#!/bin/bash
# small-batch A/B of environment switches on one box (inside gpurun): tools/ab_small.sh "32 64" "" "VAR=1" "VAR2=x VAR3=y" ...
# prints img/s per (batch, switch set), two passes so that drift shows
BATCHES="$1"; shift
for pass in 1 2; do
  for b in $BATCHES; do
    for sw in "$@"; do
      v=$(env $sw timeout -k 10 200 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))") || exit 1
      echo "pass $pass batch $b [$sw] $v" | tee -a gpurun_out/ab_small.log
    done
  done
done
