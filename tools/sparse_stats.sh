# dev: kernel statistics of the sparse (dam-break) step at 512^3 and 256^3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-sp}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 512 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$n -o sparse -- python3 $ROOT/tools/full_step_run.py $n 20 > $OUT/s$n.log 2>&1
tail -1 $OUT/s$n.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/s$n/sparse_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.3f ms over 23 steps -> %.3f ms/step" % (tot / 1e6, tot / 1e6 / 23))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-90s calls %6s avg %9.1f us  total %8.3f ms" % (r["Name"][:90].replace("void fluid::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
