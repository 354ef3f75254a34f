#!/bin/bash
# Everything profiles/roundNN needs from one GPU call: counter passes for the Jacobi kernels (512^3, 256^3, the C5
# per-GPU slab shape, the whole C5 grid) and for whole steps of the full tank, the default bench line and the
# C2 / C3 / C5 lines, kernel stats of the dense and the sparse full step, the slab rehearsals.
#   ROUND=round03 bash tools/round_profiles.sh gpurun_out/round03a
OUT=${1:-gpurun_out/round}; ROOT=$(pwd); mkdir -p $OUT; P=profiles/${ROUND:-round03}; mkdir -p $P
# counters first: bench.py quotes roofline.traffic only from a record stamped with the current kernel sources
# (PART=A: the counter passes; PART=B: everything else, once A's records are in profiles/; default: both)
if [ "$PART" != "B" ]; then
bash tools/pmc_fused.sh $OUT/pmc512 > $OUT/pmc512.log 2>&1; echo "pmc512 done"
bash tools/pmc_fused.sh $OUT/pmc256 --grid 256 > $OUT/pmc256.log 2>&1; echo "pmc256 done"
bash tools/pmc_fused.sh $OUT/pmc1024x64 --grid 1024 1024 64 > $OUT/pmc1024x64.log 2>&1; echo "pmc1024x64 done"
PMC_ONLY_TRAFFIC=1 bash tools/pmc_fused.sh $OUT/pmc1024x512 --grid 1024 1024 512 --iters 400 > $OUT/pmc1024x512.log 2>&1; echo "pmc1024x512 done"
python3 tools/make_pmc_traffic.py $OUT/pmc512 $P 512 512 512 > /dev/null
python3 tools/make_pmc_traffic.py $OUT/pmc256 $P 256 256 256 > /dev/null
python3 tools/make_pmc_traffic.py $OUT/pmc1024x64 $P 1024 1024 64 > /dev/null
python3 tools/make_pmc_traffic.py $OUT/pmc1024x512 $P 1024 1024 512 > /dev/null
bash tools/pmc_step_traffic.sh $OUT/pmc_dense_step 512 200 > $OUT/pmc_dense_step.log 2>&1
python3 tools/make_step_traffic.py $OUT/pmc_dense_step $P 512 200 > /dev/null; echo "dense step traffic done"
cp $P/pmc_traffic_*.json $OUT/
fi
[ "$PART" = "A" ] && exit 0
python3 bench.py > $OUT/bench_512_default.json 2> $OUT/bench_512_default.err; echo "default bench rc=$?"
python3 bench.py --grid 256 --no-surface > $OUT/bench_256_c3.json 2>/dev/null
python3 bench.py --grid 128 --iters 80 --no-surface > $OUT/bench_128_c2.json 2>/dev/null
python3 bench.py --grid 1024 1024 512 --iters 400 --steps 3 --warmup 1 --no-cpu-baseline --no-full-step > $OUT/bench_1024x1024x512_c5.json 2>/dev/null
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/dense_stats -o dense -- python3 $ROOT/tools/full_fluid_step.py 512 200 > $ROOT/$OUT/dense_stats.log 2>&1)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/sparse_stats -o sparse -- python3 $ROOT/tools/full_step_run.py 512 10 > $ROOT/$OUT/sparse_stats.log 2>&1)
python3 tools/slab_rank_sim.py --ranks 2 4 8 --halo 3 6 8 > $OUT/slab_rank_rehearsal.txt 2>&1
python3 tools/slab_one_rank_step.py > $OUT/slab_one_rank_step.txt 2>&1
python3 tools/slab_dense_rank_step.py 512 8 > $OUT/slab_dense_rank_step.txt 2>&1
tail -3 $OUT/dense_stats.log; tail -2 $OUT/sparse_stats.log; cat $OUT/slab_one_rank_step.txt
