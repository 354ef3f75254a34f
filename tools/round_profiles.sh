#!/bin/bash
# Everything profiles/roundNN needs from one GPU call: the default bench line, the C2 / C3 lines, counter
# passes for the Jacobi kernel at 512^3 and at the C5 per-GPU slab shape, kernel stats of the dense and the
# sparse full step, the slab rehearsals.   bash tools/round_profiles.sh gpurun_out/round
OUT=${1:-gpurun_out/round}; ROOT=$(pwd); mkdir -p $OUT
# counters first: bench.py quotes roofline.traffic only from a record stamped with the current kernel sources
bash tools/pmc_fused.sh $OUT/pmc512 > $OUT/pmc512.log 2>&1
bash tools/pmc_fused.sh $OUT/pmc1024x64 --grid 1024 1024 64 > $OUT/pmc1024x64.log 2>&1
mkdir -p profiles/round02
python3 tools/make_pmc_traffic.py $OUT/pmc512 profiles/round02 512 512 512 > /dev/null
python3 tools/make_pmc_traffic.py $OUT/pmc1024x64 profiles/round02 1024 1024 64 > /dev/null
cp profiles/round02/pmc_traffic_k12_canon2_*.json $OUT/
python3 bench.py > $OUT/bench_512_default.json 2> $OUT/bench_512_default.err; echo "default bench rc=$?"
python3 bench.py --grid 256 --no-surface > $OUT/bench_256_c3.json 2>/dev/null
python3 bench.py --grid 128 --iters 80 --no-surface > $OUT/bench_128_c2.json 2>/dev/null
python3 bench.py --grid 1024 1024 512 --iters 400 --steps 3 --warmup 1 --no-cpu-baseline --no-full-step > $OUT/bench_1024x1024x512_c5.json 2>/dev/null
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/dense_stats -o dense -- python3 $ROOT/tools/full_fluid_step.py 512 200 > $ROOT/$OUT/dense_stats.log 2>&1)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/sparse_stats -o sparse -- python3 $ROOT/tools/full_step_run.py 512 10 > $ROOT/$OUT/sparse_stats.log 2>&1)
python3 tools/slab_rank_sim.py --ranks 2 4 8 --halo 8 > $OUT/slab_rank_rehearsal.txt 2>&1
python3 tools/slab_one_rank_step.py > $OUT/slab_one_rank_step.txt 2>&1
python3 tools/slab_dense_rank_step.py 512 8 > $OUT/slab_dense_rank_step.txt 2>&1
python3 tools/sor_time.py > $OUT/sor_time.txt 2>&1
python3 tools/particle_sort_ab.py 512 20 > $OUT/particle_sort_ab.txt 2>&1
python3 tools/particle_sort_longrun.py 512 400 50 > $OUT/particle_sort_longrun.txt 2>&1
(python3 tools/surface_time.py 128 0; echo "--- one dispatch per pass (FLUID_OPT_SURFACE_KERNEL = 1 is the v4 kernel; the z-march single dispatches: 0.77 ms each) ---"; python3 tools/surface_time.py 128 1) > $OUT/surface_time.txt 2>&1
tail -3 $OUT/dense_stats.log; tail -2 $OUT/sparse_stats.log; cat $OUT/slab_one_rank_step.txt
