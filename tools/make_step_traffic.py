"""Turn the two counter passes of tools/pmc_step_traffic.sh into the record bench.py reads as
full_step_dense.traffic: profiles/<round>/pmc_traffic_full_step_dense_<n>.json, stamped with the hash of ALL
kernel sources.  The run is tools/full_fluid_step.py: run_init + 8 steps; the bytes of every kernel of the run
are divided by 8 (the init list's clears and 00_init_particles are about 1 % of that).
    python tools/make_step_traffic.py gpurun_out/x profiles/round03 512 200"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, dst, n, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
tot = {}
per_kernel = {}
for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            k = r["Kernel_Name"].split("(")[0].replace("void fluid::", "").replace("fluid::", "")
            d = per_kernel.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
STEPS = 8
rd = tot["TCC_EA0_RDREQ_sum"] * 128.0 / STEPS
wr = tot["WRITE_SIZE"] * 1024.0 / STEPS
top = sorted(per_kernel.items(), key=lambda kv: -(kv[1].get("TCC_EA0_RDREQ_sum", 0) * 128 + kv[1].get("WRITE_SIZE", 0) * 1024))[:10]
rec = {"workload": f"full tank {n}^3, 8 particles per cell, {iters} Jacobi iterations (tools/full_fluid_step.py)",
       "grid": [n, n, n], "jacobi_iterations": iters, "steps_in_run": STEPS,
       "collected": "rocprofv3 --pmc TCC_EA0_RDREQ_sum / WRITE_SIZE, one pass each, summed over every kernel of the run",
       "read_bytes_per_step": rd, "write_bytes_per_step": wr, "traffic_bytes_per_step": rd + wr,
       "largest_kernels_bytes_per_step": {k: (v.get("TCC_EA0_RDREQ_sum", 0) * 128 + v.get("WRITE_SIZE", 0) * 1024) / STEPS
                                          for k, v in top},
       "all_kernel_sources_sha16": bench.all_kernel_sources_sha16()}
os.makedirs(dst, exist_ok=True)
out = os.path.join(dst, f"pmc_traffic_full_step_dense_{n}.json")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(out, json.dumps(rec))
