set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# calibration: pure streaming kernels with known byte counts
for c in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" ; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$tag -o pmc -- $ROOT/tools/stream_ceiling > $OUT/calib_$tag.log 2>&1 || echo "fail $c"
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/calib_summary.txt; cat $OUT/calib_summary.txt
# single-sweep kernel (no temporal blocking) under the same counters
cd $ROOT
for c in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $OUT/nofuse_$tag -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-full-step --no-fuse > $OUT/nofuse_$tag.log 2>&1) || echo "fail $c"
done
python3 tools/pmc_summary.py $OUT > $OUT/all_summary.txt; grep "k12_canon<" $OUT/all_summary.txt
# W = 1024 per-GPU slab shape of C5 and the whole C5 grid
python3 bench.py --grid 1024 1024 64 --steps 5 --warmup 2 --no-cpu-baseline --no-full-step > $OUT/b1024x64.json 2> $OUT/b1024x64.err; python3 -c "
import json; d=json.loads(open('$OUT/b1024x64.json').read().strip().splitlines()[-1]); print('1024x1024x64', d['value'], d['roofline']['frac'], d['roofline']['ms_per_sweep'])"
