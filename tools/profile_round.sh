#!/bin/bash
# Every profile the round's numbers come from, on the GPU box (inside gpurun):  tools/profile_round.sh r02
#   kernel trace + stats of the bench command, one SQ counter pass, FETCH_SIZE and WRITE_SIZE in passes of their own
#   (MI355X_MICROARCH.md: the two TCC counters do not fit one pass; FETCH_SIZE x2 on gfx950).
# Summaries land in gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
set -e
TAG=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
PMCB="python3 bench.py --steps 1 --warmup 0 --pgd-steps 2 --no-cpu-baseline --no-roofline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/ktrace.log
cp $(find $OUT/ktrace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
echo "kernel stats done"
export VITLORA_NO_GRAPH=1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -d $OUT/pmc_sq -- $PMCB > /dev/null 2> $OUT/pmc_sq.log
python3 tools/pmc_summary.py $(find $OUT/pmc_sq -name "*results.db" | head -1) $OUT/pmc_sq.json > $OUT/pmc_sq_summary.txt
echo "SQ pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $PMCB > /dev/null 2> $OUT/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $PMCB > /dev/null 2> $OUT/pmc_write.log
python3 tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*results.db" | head -1) $(find $OUT/pmc_write -name "*results.db" | head -1) $OUT/pmc_hbm_traffic.json $OUT/pmc_hbm_traffic.txt > /dev/null
echo "traffic passes done"
rm -rf $OUT/ktrace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
