# dev: where the three-sweep launch spends its time at 512^3 (launch shape, counters, chunk length, load point)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-t3x}; mkdir -p $OUT
q() { python3 bench.py --grid $G --steps 4 --warmup 2 --no-cpu-baseline --no-full-step 2> $OUT/b.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'it/s', round(d['roofline']['ms_per_sweep'],4), 'ms/sweep')"; }
G=512
FLUID_FUSED_DEBUG=1 python3 bench.py --grid 512 --steps 1 --warmup 0 --no-cpu-baseline --no-full-step > /dev/null 2> $OUT/debug.err; grep -m4 xcd_plan $OUT/debug.err
q "default T=3"
for z in 24 32 48 64 96 128; do FLUID_FUSED_ZCHUNK=$z q "zchunk $z"; done
FLUID_FUSED_XCD=0 q "xcd off"
FLUID_FUSED_NT=0 q "cached stores"
bash tools/pmc_fused.sh gpurun_out/${1:-t3x}/pmc512 > $OUT/pmc512.log 2>&1; grep -E "k12_canon" $OUT/pmc512/summary.txt | head -5
(cd vulkan-3d-fluid-simulation_amd/csrc && touch pressure_fused3.hip && make HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DFT3_LOAD_AFTER=1" > $OUT/make.log 2>&1; tail -1 $OUT/make.log)
q "load after stage 1"
G=256; q "256 load after stage 1"
