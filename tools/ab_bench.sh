#!/bin/bash
# A/B on one GPU box: working-tree library (B) against csrc/build/libA.so (A), alternating runs.
# usage (inside gpurun): tools/ab_bench.sh [pairs]     -> gpurun_out/ab_{A,B}{i}.log
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
A=$PWD/$P/csrc/build/libA.so
N=${1:-2}
for i in $(seq 1 $N); do
  timeout -k 10 200 python bench.py --steps 1 --warmup 1 --pgd-steps 4 --no-cpu-baseline --no-extras > gpurun_out/ab_B$i.log 2>/dev/null || exit 1
  VITLORA_LIB=$A timeout -k 10 200 python bench.py --steps 1 --warmup 1 --pgd-steps 4 --no-cpu-baseline --no-extras > gpurun_out/ab_A$i.log 2>/dev/null || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_[AB]*.log")):
    d = json.loads(open(f).read())
    k = d["roofline"]["kernels_ms_per_pgd_iteration"]
    g = sum(v for n, v in k.items() if n.startswith("gemm256"))
    print(f[-10:-4], round(d["ms_per_step"], 2), "gemm256", round(g, 3), "skinny", k.get("gemm_nt_kernel<128, 64, 0>"),
          "attn", k.get("attn_fwd32_kernel"), k.get("attn_bwd32_kernel"), "ln", k.get("layernorm_fwd_kernel"), k.get("layernorm_bwd_kernel"))
PY
