#!/bin/bash
# same-box A/B of the bf16 build: working-tree library (B) against csrc/build/libA.so (A).  usage (inside gpurun): tools/ab_bf16.sh [pairs]
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
A=$PWD/$P/csrc/build/libA.so
N=${1:-2}
B="python bench.py --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for i in $(seq 1 $N); do
  timeout -k 10 200 $B > gpurun_out/abbf_B$i.json 2>/dev/null || exit 1
  VITLORA_LIB=$A timeout -k 10 200 $B > gpurun_out/abbf_A$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abbf_[AB]*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels_ms_per_pgd_iteration"]
    print(f[-9:-5], round(d["value"], 1), {n: v for n, v in k.items() if n.startswith("gemm256")})
PY
