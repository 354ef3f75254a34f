# dev: parity of the sampler kernels, then per-section times of the dense tank and the dam break
python -m pytest tests/test_engine_parity_gpu.py tests/test_surface_gpu.py -x -q -k "section or advect or full_step or c1 or c2 or golden or blob or 256cubed" > gpurun_out/dense_tests.log 2>&1; tail -3 gpurun_out/dense_tests.log
python tools/full_fluid_step.py 512 50 2>&1 | tail -16
python tools/full_fluid_step.py 256 50 2>&1 | head -1
