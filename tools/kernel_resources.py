"""Dev tool: one line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage (what `make resources`
prints): SGPRs, VGPRs, scratch bytes per lane, wavefronts per SIMD, LDS bytes per workgroup.
    python tools/kernel_resources.py > profiles/roundNN/kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vulkan-3d-fluid-simulation_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
KEYS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"),
        (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "lds"))
order, d = [], {}
for unit in ("engine", "pressure_sweep", "pressure_fused", "pressure_fused_win", "pressure_fused_stream", "pressure_fused3",
             "pressure_passes"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", "-o",
                          "/dev/null", unit + ".hip"], cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    for line in out.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            if cur not in d:
                d[cur] = {}
                order.append(cur)
        for k, short in KEYS:
            m = re.search(r"(?<![A-Za-z])" + k + r": (\d+)", line)
            if m and cur:
                d[cur][short] = int(m.group(1))
names = subprocess.run(["c++filt", *order], capture_output=True, text=True).stdout.split("\n")
print("# kernel | SGPRs | VGPRs | scratch B/lane | wavefronts/SIMD | LDS B/workgroup   (hipcc, gfx950)")
for n, dm in zip(order, names):
    v = d[n]
    dm = re.sub(r"\(.*", "", dm).replace("void ", "").replace("fluid::", "")
    print(f"{dm:62s} {v.get('sgpr', -1):4d} {v.get('vgpr', -1):4d} {v.get('scratch', -1):5d} {v.get('occ', -1):2d} "
          f"{v.get('lds', -1):6d}")
