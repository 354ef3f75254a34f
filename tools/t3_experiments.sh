# dev: variants of the three-sweep kernel at 512^3 / 256^3: rebuild pressure_fused3.o with a flag set each
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-t3e}; mkdir -p $OUT
q() { for w in 3 10; do echo -n "$1 warmup $w: "; python3 tools/full_step_run.py 512 20 200 $w 2>&1 | tail -1; done; echo -n "$1 256: "; python3 tools/full_step_run.py 256 20 200 5 2>&1 | tail -1; for G in 256; do python3 bench.py --grid $G --steps 4 --warmup 2 --no-cpu-baseline --no-full-step 2> $OUT/b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 grid $G', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep')"; done; }
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for x in "" $T3_FLAGS; do   # T3_FLAGS="-DSOMETHING -DOTHER": one rebuild per flag
  (cd vulkan-3d-fluid-simulation_amd/csrc && touch pressure_fused3.hip && make HIPFLAGS="$FL $x" > $OUT/make.log 2>&1 || tail -3 $OUT/make.log)
  q "[$x]"
done

