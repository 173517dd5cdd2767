#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
cd $R
python - <<'PY' > gpurun_out/r02h_startup.log 2>&1
import sys, time
sys.path[:0] = ['.', 'frame-interpolation_amd']
import numpy as np
from film_hip import weights as W
from film_hip.options import PUBLISHED
w = W.make_synthetic_weights(PUBLISHED, seed=0)
from eval.interpolator import Interpolator
import torch; torch.cuda.init()
t0 = time.time(); it = Interpolator('', align=64, block_shape=[2, 2], weights=w); t1 = time.time()
print(f'Interpolator(...) construction (137.7 MB of parameters in memory -> engine ready): {t1 - t0:.2f} s; layout blob {it.engine.export_layouts().size * 4 / 1e6:.0f} MB')
x = np.random.default_rng(0).random((1, 1080, 1920, 3), dtype=np.float32)
t0 = time.time(); it(x, x, None); t1 = time.time(); it(x, x, None); t2 = time.time()
print(f'first 1080p 2x2 call (plan + autotune + graph capture): {t1 - t0:.2f} s; second call {t2 - t1:.3f} s')
PY
cat gpurun_out/r02h_startup.log | grep -v amdgpu
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02h_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r02h_gpu_tests.log
timeout 900 python bench.py --profile-out gpurun_out/r02h_per_op_profile.json > gpurun_out/r02h_bench_1gpu.json 2> gpurun_out/r02h_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r02h_bench_1gpu.json; tail -2 gpurun_out/r02h_bench_1gpu.err
