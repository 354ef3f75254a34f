#!/bin/bash
# HBM-side bytes of whole steps of the full tank (tools/full_fluid_step.py): FETCH_SIZE and WRITE_SIZE over
# every kernel of the run, separate --pmc passes.   bash tools/pmc_step_traffic.sh gpurun_out/x [grid=512] [iters=200]
set -e
OUT=${1:-gpurun_out/pmc_step}; GRID=${2:-512}; ITERS=${3:-200}
ROOT=$(pwd); mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/full_fluid_step.py $GRID $ITERS"
i=0
for c in "TCC_EA0_RDREQ_sum" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$i" -o pmc -- $CMD > "$ROOT/$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($c) failed"
    echo "pmc pass $i done"
done
