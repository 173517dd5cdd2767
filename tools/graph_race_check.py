"""Graph replay (two lanes; option "graph" = 1, or RACE_GRAPH=2 for the direct two-lane launches) vs one-stream eager launches on inputs that CHANGE every forward: any ordering / visibility hole in the
replayed graph shows up as stale data.  Prints max|diff| per shape and forward."""
import os, sys, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'frame-interpolation_amd'), os.path.join(_R, 'tests')]
from film_hip import weights as W
from film_hip.options import PUBLISHED
from film_hip.engine import FilmEngine, hip_runtime_info
print('HIP runtime of this process: %s  version %d = %s  (FILM_NO_TORCH=%s)' % (hip_runtime_info() + (os.environ.get('FILM_NO_TORCH', '0'),)), flush=True)
w = W.make_synthetic_weights(PUBLISHED, seed=0)
fuse = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eg = FilmEngine(PUBLISHED, device=0); eg.set_weights(w); eg.set_option('fuse', fuse); eg.set_option('graph', int(os.environ.get('RACE_GRAPH', '1')))
ee = FilmEngine(PUBLISHED, device=0); ee.set_weights(w); ee.set_option('fuse', fuse); ee.set_option('graph', 0)
SHAPES = ((1, 64, 64), (1, 128, 64), (2, 64, 128), (1, 256, 256), (1, 192, 320), (1, 576, 960))[:int(os.environ.get('RACE_SHAPES', '6'))]
stale = 0
for (b, h, wd) in SHAPES:
    worst = 0.0
    for it in range(4):
        rng = np.random.default_rng(100 * h + it)
        x0 = rng.random((b, h, wd, 3), dtype=np.float32)
        x1 = rng.random((b, h, wd, 3), dtype=np.float32)
        a = eg.forward(x0, x1); c = ee.forward(x0, x1)
        d = float(np.abs(a - c).max())
        taps = {}
        for l in range(5):
            taps[l] = float(np.abs(eg.tap(f'aligned{l}') - ee.tap(f'aligned{l}')).max())
        worst = max(worst, d)
        stale += int(d != 0.0 or any(v != 0.0 for v in taps.values()))
        print(f'{b}x{h}x{wd} forward {it}: image {d:.3e} aligned {[f"{v:.1e}" for v in taps.values()]}', flush=True)
print(f'done: {stale} stale forwards')
sys.exit(1 if stale else 0)
