#!/bin/bash
# Copy what tools/round_profiles.sh collected (gpurun_out/<name>, scratch) into profiles/$ROUND (tracked) under the
# names its README.md lists.   ROUND=round03 bash tools/install_round_profiles.sh gpurun_out/round03c
set -e
R=${1:?collection directory}; P=profiles/${ROUND:-round03}; mkdir -p $P
cp $R/bench_512_default.json $R/bench_256_c3.json $R/bench_128_c2.json $R/bench_1024x1024x512_c5.json $P/
cp $R/pmc_traffic_*.json $P/
cp $R/pmc512/summary.txt $P/bench_512_jacobi_pmc_summary.txt
cp $R/pmc256/summary.txt $P/bench_256_jacobi_pmc_summary.txt
cp $R/pmc1024x64/summary.txt $P/bench_1024x1024x64_jacobi_pmc_summary.txt
cp $R/pmc512/stats/jacobi_kernel_stats.csv $P/bench_512_jacobi_kernel_stats.csv
cp $R/pmc256/stats/jacobi_kernel_stats.csv $P/bench_256_jacobi_kernel_stats.csv
cp $R/pmc1024x64/stats/jacobi_kernel_stats.csv $P/bench_1024x1024x64_jacobi_kernel_stats.csv
cp $R/dense_stats/dense_kernel_stats.csv $P/full_step_dense_512_kernel_stats.csv
cp $R/sparse_stats/sparse_kernel_stats.csv $P/full_step_512_kernel_stats.csv
cp $R/slab_rank_rehearsal.txt $R/slab_one_rank_step.txt $R/slab_dense_rank_step.txt $P/
grep -v "^[WE]2026" $R/dense_stats.log > $P/full_step_dense_512_sections.txt
grep -v "^[WE]2026" $R/sparse_stats.log > $P/full_step_512_sections.txt
python3 tools/kernel_resources.py > $P/kernel_resources.txt 2>/dev/null
python3 - <<PY
import json
R = '$P/'
for f in ['bench_512_default.json', 'bench_256_c3.json', 'bench_128_c2.json', 'bench_1024x1024x512_c5.json']:
    d = json.load(open(R + f)); r = d['roofline']
    print(f, round(d['value'], 1), 'frac', round(r['frac'], 3), 'launch_ms', round(r['launch_ms'], 4),
          'frac_traffic', r.get('frac_traffic'), r.get('traffic_over_single_pass_min'),
          round(r.get('measured_copy_GBps') or 0), r.get('kernel_sources_sha16'))
    for k in ('full_step', 'full_step_dense', 'surface_prep'):
        if k in d:
            x = d[k]
            print('  ', k, {kk: (round(x[kk], 3) if isinstance(x[kk], float) else x[kk]) for kk in x
                            if kk in ('steps_per_sec', 'ms_per_step', 'algorithmic_GBps', 'frac_of_hbm_peak',
                                      'surface_tail_ms', 'frac_traffic')})
PY
