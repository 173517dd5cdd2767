#!/usr/bin/env python
"""Start-up numbers of the drop-in: Interpolator(...) construction, first 1080p 2x2-tiled call (plan + autotune + graph
capture), the same in a second engine that reads $FILM_TUNE_CACHE, steady-state call.  -> profiles/r02_startup.log"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'frame-interpolation_amd')]
from film_hip import weights as W  # noqa: E402
from film_hip.options import PUBLISHED  # noqa: E402
from eval.interpolator import Interpolator  # noqa: E402

w = W.make_synthetic_weights(PUBLISHED, seed=0)
rng = np.random.default_rng(0)
x0 = rng.random((1, 1080, 1920, 3), dtype=np.float32)
x1 = rng.random((1, 1080, 1920, 3), dtype=np.float32)
dt = np.full((1,), 0.5, np.float32)
os.environ['FILM_TUNE_CACHE'] = os.path.join(tempfile.mkdtemp(), 'tune.txt')
for label in ('no tune cache yet', 'tune cache of the first engine'):
    t0 = time.perf_counter()
    it = Interpolator('', align=64, block_shape=[2, 2], weights=w)
    t1 = time.perf_counter()
    it(x0, x1, dt)
    t2 = time.perf_counter()
    it(x0, x1, dt)
    t3 = time.perf_counter()
    print(f'{label}: construction {t1 - t0:.2f} s, first 1080p 2x2 call {t2 - t1:.2f} s, next call (host buffers) {t3 - t2:.3f} s', flush=True)
    del it
