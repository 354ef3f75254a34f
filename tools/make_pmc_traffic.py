"""Turn the counter passes of tools/pmc_fused.sh into the record bench.py reads as roofline.traffic:
profiles/<round>/pmc_traffic_k12_canon2_<W>x<H>x<D>.json, stamped with the hash of the kernel sources.

    python tools/make_pmc_traffic.py gpurun_out/r2g/pmc512 profiles/round02 512 512 512

Bytes: the L2s' read requests to the fabric are 128 B each for this kernel's loads (calibrated on streaming
kernels of known size, profiles/round02/README.md: k_copy / k_shape, TCC_EA0_RDREQ x 128 B = bytes read);
FETCH_SIZE counts them as 64 B, hence the factor of the guide.  WRITE_SIZE is exact (64-B requests).
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # kernel_sources_sha16

src, dst = sys.argv[1], sys.argv[2]
grid = [int(v) for v in sys.argv[3:6]]
KERNEL = sys.argv[6] if len(sys.argv) > 6 else ("k12_canon_t" if grid[0] <= 512 else "k12_canon2")
vals = {}
for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL + "<" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
avg_ns = None   # over all template variants of the kernel (the last pair of a loop is the KEEP one), like the counters
tot_ns = calls = 0.0
for path in glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")):
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL + "<" in r["Name"]:
                tot_ns += float(r["TotalDurationNs"])
                calls += float(r["Calls"])
if calls:
    avg_ns = tot_ns / calls
rd = mean["TCC_EA0_RDREQ_sum"] * 128.0
wr = mean["WRITE_SIZE"] * 1024.0
rec = {"kernel": KERNEL, "grid": grid, "sweeps_per_launch": 3 if KERNEL == "k12_canon_t" else 2,
       "command": "tools/pmc_fused.sh (python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step)",
       "collected": "separate rocprofv3 --pmc passes, mean over the launches of the run",
       "TCC_EA0_RDREQ_sum": mean["TCC_EA0_RDREQ_sum"], "FETCH_SIZE_KiB": mean.get("FETCH_SIZE"),
       "WRITE_SIZE_KiB": mean["WRITE_SIZE"],
       "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
       "traffic_bytes_per_launch": rd + wr,
       "single_pass_min_bytes": 13.0 * grid[0] * grid[1] * grid[2],
       "avg_launch_ns_kernel_trace": avg_ns,
       "kernel_sources_sha16": bench.kernel_sources_sha16()}
os.makedirs(dst, exist_ok=True)
out = os.path.join(dst, f"pmc_traffic_{KERNEL}_{grid[0]}x{grid[1]}x{grid[2]}.json")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(out, json.dumps(rec))
