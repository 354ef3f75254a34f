ROOT=$(pwd); OUT=$ROOT/gpurun_out/rg; mkdir -p $OUT
for rg in 1 2 3; do
  export FLUID_FUSED_RG=$rg
  python -m pytest tests/test_engine_parity_gpu.py -x -q -k "pressure or c5 or moving_blob" > $OUT/tests_rg$rg.log 2>&1; echo "RG=$rg tests rc=$? $(tail -1 $OUT/tests_rg$rg.log)"
  for g in "512" "1024 1024 64" "256"; do
    python3 bench.py --grid $g --steps 3 --warmup 1 --no-cpu-baseline --no-full-step > $OUT/b.json 2> $OUT/b.err
    python3 -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('RG=$rg grid $g:', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep')"
  done
done
