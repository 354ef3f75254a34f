#!/bin/bash
# Counter passes over whole steps of the full tank (tools/full_fluid_step.py): what bounds 07, 01, 14.
#   bash tools/pmc_dense_step.sh gpurun_out/pmc_dense [grid=256] [iters=10]
set -e
OUT=${1:-gpurun_out/pmc_dense}; GRID=${2:-256}; ITERS=${3:-10}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/full_fluid_step.py $GRID $ITERS"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$i" -o pmc -- $CMD > "$ROOT/$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($c) failed"
    echo "pmc pass $i done"
done
python3 $ROOT/tools/pmc_summary.py "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt"
grep -E "k07_advect|k14_binned|k01_binned|k_pbin" "$ROOT/$OUT/summary.txt" || true
