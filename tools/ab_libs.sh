#!/bin/bash
# same-box A/B of library variants: tools/ab_libs.sh <pairs> <lib1.so> [lib2.so ...]   (inside gpurun; libs relative to csrc/build/)
# prints img/s and every kernel line >= 0.15 ms of the kernel table per run; the working-tree library is arm "cur"
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
N=$1; shift
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rm -f gpurun_out/abl_*.json
for i in $(seq 1 $N); do
  timeout -k 10 200 $B > gpurun_out/abl_cur_$i.json 2>/dev/null || exit 1
  for l in "$@"; do
    VITLORA_LIB=$PWD/$P/csrc/build/$l timeout -k 10 200 $B > gpurun_out/abl_${l%.so}_$i.json 2>/dev/null || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abl_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels_ms_per_pgd_iteration"]
    print(f[15:-5].ljust(16), round(d["value"], 1), "K10 frac", round(d["roofline"]["pgd_step"]["frac"], 3),
          {n.replace("_kernel", ""): round(v, 3) for n, v in k.items() if v >= 0.15})
PY
