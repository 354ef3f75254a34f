#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: kernel-trace stats of the Jacobi bench and of the full
# step, and separate --pmc passes for the Jacobi bench (never combined with trace domains other than
# --kernel-trace).  Run on the GPU box from the repo root:  bash tools/profile_round.sh gpurun_out/prof
set -e
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-step"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/jacobi_stats" -o jacobi -- $BENCH > "$ROOT/$OUT/jacobi_stats.log" 2>&1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$tag" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_$tag.log" 2>&1
    echo "pmc $c done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/step_stats" -o step -- python3 $ROOT/tools/full_step_run.py 512 10 > "$ROOT/$OUT/step_stats.log" 2>&1
echo "step stats done"
