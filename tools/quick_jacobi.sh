# dev: parity of the pressure kernels, then the Jacobi loop numbers at the three shapes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-quick}; mkdir -p $OUT
python -m pytest tests/test_engine_parity_gpu.py -x -q -k "pressure or c5 or moving_blob" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -4 $OUT/tests.log
for g in "512" "1024 1024 64" "256" "1024 1024 512"; do
  python3 bench.py --grid $g --steps 3 --warmup 1 --no-cpu-baseline --no-full-step > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('grid $g:', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep alg frac', round(d['roofline']['frac'],3))"
done
