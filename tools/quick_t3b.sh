# dev: parity of the pressure kernels, then two against three sweeps per pass at a few shapes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-t3b}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_engine_parity_gpu.py -x -q -k "pressure or c5 or moving_blob or quiet" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -4 $OUT/tests.log
grep -q "rc=0" $OUT/tests.log || exit 1
for t in 2 3; do for g in "512" "256" "512 512 64" "512 512 128"; do
  FLUID_FUSED_T=$t python3 bench.py --grid $g --steps 5 --warmup 2 --no-cpu-baseline --no-full-step 2> $OUT/b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T=$t grid $g:', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep')"
done; done
for t in 2 3; do
  FLUID_FUSED_T=$t python3 tools/full_step_run.py 512 20 2>&1 | tail -1
  FLUID_FUSED_T=$t python3 tools/full_step_run.py 256 20 2>&1 | tail -1
done
