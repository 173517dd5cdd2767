import sys, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/frame-interpolation_amd', '/root/repo/tests']
from film_hip import weights as W
from film_hip.options import PUBLISHED
from film_hip.engine import FilmEngine
w = W.make_synthetic_weights(PUBLISHED, seed=0)
rng = np.random.default_rng(5)
x0 = rng.random((1, 256, 256, 3), dtype=np.float32); x1 = rng.random((1, 256, 256, 3), dtype=np.float32)
outs = {}
for fuse in (7, 15):
    eng = FilmEngine(PUBLISHED, device=0); eng.set_weights(w); eng.set_option('fuse', fuse); eng.set_option('graph', 0); eng.set_option('autotune', 0)
    img = eng.forward(x0, x1)
    plan = eng.plan(1, 256, 256)
    if fuse == 15:
        for o in plan['ops']:
            if '+pool' in o['tag']: print(o['tag'], 'tile', o['tile'] & 15, o['H'], o['W'], o['Cout'], o['out2'])
    outs[fuse] = {k: eng.tap(k) for k in ('feat0', 'feat1', 'feat2')}
    eng.close()
for k in ('feat0', 'feat1'):
    d = np.abs(outs[7][k] - outs[15][k])
    print(k, d.max(), 'bad frac', (d > 1e-4).mean())
    if d.max() > 1e-4:
        bad = np.argwhere(d > 1e-4)
        print('  n', sorted(set(bad[:,0])), 'y%4', np.bincount(bad[:,1] % 4), 'x%4', np.bincount(bad[:,2] % 4), 'c range', bad[:,3].min(), bad[:,3].max(), 'c%32 counts', np.bincount(bad[:,3] // 64))
