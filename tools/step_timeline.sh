# dev: kernel timeline of one sparse (dam-break) step: start, duration and the gap to the previous kernel's end
#   bash tools/step_timeline.sh [out=tl] [grid=512]          DENSE=1: the full tank (tools/full_fluid_step.py) instead;
#   SLAB=1: the middle rank of its 8-way split (tools/slab_dense_rank_step.py); SCRIPT=tools/x.py: that program
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-tl}; N=${2:-512}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$SCRIPT" ]; then
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o tl -- python3 $ROOT/$SCRIPT $N > $OUT/run.log 2>&1
elif [ -n "$SLAB" ]; then
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o tl -- python3 $ROOT/tools/slab_dense_rank_step.py $N 8 > $OUT/run.log 2>&1
elif [ -n "$DENSE" ]; then
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o tl -- python3 $ROOT/tools/full_fluid_step.py $N 200 > $OUT/run.log 2>&1
else
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o tl -- python3 $ROOT/tools/full_step_run.py $N 6 200 4 > $OUT/run.log 2>&1
fi
tail -1 $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/tl_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last k01 / psort kernel on
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k01_" in n and "strays" not in n]
lo = starts[-1]
out = open("$OUT/timeline.txt", "w")
prev_end = None
tot_k = tot_gap = 0
loop_k = loop_gap = 0; nloop = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    nm = r["Kernel_Name"].replace("void fluid::", "")[:70]
    out.write("%9.1f us gap %7.1f  dur %8.1f  %s  grid %s/%s/%s\n" % ((s - int(rows[lo]["Start_Timestamp"])) / 1e3, gap, (e - s) / 1e3, nm, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")))
    tot_k += (e - s) / 1e3; tot_gap += gap
    if "k12_canon" in nm:
        loop_k += (e - s) / 1e3; loop_gap += gap; nloop += 1
    prev_end = e
print("last step: kernels %.1f us, gaps %.1f us; loop launches %d: kernels %.1f us, gaps in front of them %.1f us" % (tot_k, tot_gap, nloop, loop_k, loop_gap))
PY
