#!/bin/bash
# same-box A/B of library variants: tools/ab_libs.sh <pairs> <lib1.so> [lib2.so ...]   (inside gpurun; libs relative to csrc/build/)
# prints img/s and the attention / GEMM lines of the kernel table per run; the working-tree library is arm "cur"
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
N=$1; shift
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
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
    print(f[15:-5].ljust(14), round(d["value"], 1), {n.replace("_kernel", ""): v for n, v in k.items() if "attn" in n and v > 0.3},
          "gemm", round(sum(v for n, v in k.items() if n.startswith("gemm")), 3))
PY
