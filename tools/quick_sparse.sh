ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-qs}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_engine_parity_gpu.py -x -q -k "pressure or quiet or moving_blob" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
for t in 2 3; do
  FLUID_FUSED_T=$t python3 tools/full_step_run.py 512 20 2>&1 | tail -1
  FLUID_FUSED_T=$t python3 tools/full_step_run.py 256 20 2>&1 | tail -1
  FLUID_FUSED_T=$t python3 tools/full_step_run.py 128 20 80 2>&1 | tail -1
done
for g in "512" "256"; do python3 bench.py --grid $g --steps 4 --warmup 2 --no-cpu-baseline --no-full-step 2> $OUT/b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g', round(d['value'],1), 'it/s', d['roofline']['kernel'], d['roofline']['sweeps_per_launch'], round(d['roofline']['frac'],3))"; done
FLUID_FUSED_RG=1 python3 bench.py --grid 256 --steps 4 --warmup 2 --no-cpu-baseline --no-full-step 2> $OUT/b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid 256 RG=1', round(d['value'],1), 'it/s')"
