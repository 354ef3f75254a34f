#!/bin/bash
# Counter evidence for the Jacobi loop kernel (profiles/roundNN): which resource bounds it and how many
# bytes it really moves.  Separate --pmc passes only (never with trace domains other than --kernel-trace).
#   bash tools/pmc_fused.sh gpurun_out/pmc [extra bench.py flags, e.g. --grid 1024 1024 64]
set -e
OUT=${1:-gpurun_out/pmc}
shift || true
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -o jacobi -- $BENCH > "$ROOT/$OUT/stats.log" 2>&1
echo "stats done"
i=0
if [ -n "$PMC_ONLY_TRAFFIC" ]; then   # large grids: the two passes the traffic record needs
for c in "TCC_EA0_RDREQ_sum" "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$i" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($c) failed"
    echo "pmc pass $i done: $c"
done
python3 $ROOT/tools/pmc_summary.py "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt"
exit 0
fi
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$i" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_$i.log" 2>&1 || echo "pass $i ($c) failed"
    echo "pmc pass $i done: $c"
done
python3 $ROOT/tools/pmc_summary.py "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt"
grep -E "k12_canon|k_fill_u32x4" "$ROOT/$OUT/summary.txt" || true
