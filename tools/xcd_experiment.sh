# dev: does dealing the fused kernel's row tiles to the XCDs in bands cut the fabric reads, and the time?
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2e; mkdir -p $OUT
B="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-step"
for x in 0 1; do
  FLUID_FUSED_XCD=$x $B > $OUT/time_xcd$x.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$OUT/time_xcd$x.json').read().strip().splitlines()[-1]); print('XCD=$x', round(d['value'],1), 'it/s', d['roofline']['launch_ms'], 'ms/launch')"
  (cd /tmp && TMPDIR=/tmp FLUID_FUSED_XCD=$x rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_xcd$x -o pmc -- $B > $OUT/pmc_xcd$x.log 2>&1)
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_xcd$x | grep canon2
done
for z in 16 32 64 128; do
  FLUID_FUSED_ZCHUNK=$z $B > $OUT/time_z$z.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$OUT/time_z$z.json').read().strip().splitlines()[-1]); print('ZCHUNK=$z', round(d['value'],1), 'it/s', d['roofline']['launch_ms'], 'ms/launch')"
done
