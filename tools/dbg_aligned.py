import sys, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/frame-interpolation_amd', '/root/repo/tests']
from film_hip import weights as W
from film_hip.options import PUBLISHED
from film_hip.engine import FilmEngine
from oracle import film_oracle as fo
import plan_interp as pi
w = W.make_synthetic_weights(PUBLISHED, seed=0)
rng = np.random.default_rng(5)
x0 = rng.random((1, 64, 64, 3), dtype=np.float32)
x1 = (np.roll(x0, (2, -3), axis=(1, 2)) + rng.normal(0, 0.02, x0.shape)).astype(np.float32)
want, aux = fo.film_forward(x0, x1, w, fo.Options(), return_aux=True)
fc = W.feature_channels(PUBLISHED)
for fuse, graph, lanes, nfw in ((1, 1, 1, 1), (2, 1, 1, 1), (3, 1, 0, 1), (3, 1, 1, 2), (3, 1, 1, 1)):
    if True:
        eng = FilmEngine(PUBLISHED, device=0); eng.set_weights(w)
        eng.set_option('fuse', fuse); eng.set_option('graph', graph); eng.set_option('lanes', lanes)
        for _ in range(nfw): got = eng.forward(x0, x1)
        print('fuse', fuse, 'graph', graph, 'lanes', lanes, 'forwards', nfw)
        for l in (3, 4):
            a = pi.aligned_to_reference(eng.tap(f'aligned{l}'), fc[l]); r = aux['aligned_pyramid'][l]
            C = fc[l]
            d = np.abs(a - r)
            print(f'fuse {fuse} graph {graph} aligned{l}: img0 {d[..., :3].max():.2e} feat0 {d[..., 3:3+C].max():.2e} img1 {d[..., 3+C:6+C].max():.2e} feat1 {d[..., 6+C:6+2*C].max():.2e} flows {d[..., 6+2*C:].max():.2e}')
            if l == 4 and d.max() > 1e-3:
                bad = np.argwhere(d > 1e-3); print('  bad count', len(bad), 'first', bad[:5].tolist(), 'rows', sorted(set(bad[:,1].tolist())), 'cols', sorted(set(bad[:,2].tolist())), 'chan range', bad[:,3].min(), bad[:,3].max())
        print('  image', np.abs(got - want).max())
        eng.close()
