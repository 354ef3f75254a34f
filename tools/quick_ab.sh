ROOT=$(pwd); OUT=$ROOT/gpurun_out/ab; mkdir -p $OUT
for x in 0 1; do for g in "512" "256" "1024 1024 64"; do
  FLUID_FUSED_XCD=$x python3 bench.py --grid $g --steps 3 --warmup 1 --no-cpu-baseline --no-full-step > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('XCD=$x grid $g:', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep')"
done; done
