set -e
cd /root/repo
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for mode in ln o both; do
VITLORA_RESID=$mode timeout -k 10 200 $B > gpurun_out/r4_ab_${mode}_$rep.json 2> gpurun_out/r4_ab_${mode}_$rep.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab_${mode}_$rep.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels_ms_per_pgd_iteration']
print("$mode $rep", round(d['value'],1), {n:v for n,v in k.items() if 'layernorm' in n or 'pp_kernel' in n or '256, 10' in n})
PY
done
done
