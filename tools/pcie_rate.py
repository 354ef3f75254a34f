"""Dev tool: host <-> device rate of fluid_upload_image / fluid_download_image (pageable numpy buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluid_amd
from fluid_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = fluid_amd.default_params(n, n, n, 0)
with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
    v = np.zeros((n, n, n, 4), np.float32)
    eng.upload_image(E.VELOCITIES_1, v)
    t0 = time.perf_counter(); eng.upload_image(E.VELOCITIES_1, v); up = time.perf_counter() - t0
    t0 = time.perf_counter(); w = eng.download_image(E.VELOCITIES_1); down = time.perf_counter() - t0
    print(f"VELOCITIES_1 {v.nbytes / 1e9:.2f} GB: upload {v.nbytes / up / 1e9:.1f} GB/s, download {v.nbytes / down / 1e9:.1f} GB/s")
