"""Experiment: the four tiles of a 1080p 2x2 frame as ONE batch of four (what film_interpolate does) vs TWO independent
engines with two tiles each on two streams (their HBM-bound phases and workgroup tails could overlap each other's
matrix-bound phases).  Prints ms per frame for both."""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/frame-interpolation_amd']
import numpy as np, torch
from film_hip import weights as W
from film_hip.options import PUBLISHED
from film_hip.engine import FilmEngine
from film_hip.torch_io import DeviceInterpolator
w = W.make_synthetic_weights(PUBLISHED, seed=0)
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
x0 = torch.from_numpy(rng.random((1, 1080, 1920, 3), dtype=np.float32)).to(dev)
x1 = torch.from_numpy(rng.random((1, 1080, 1920, 3), dtype=np.float32)).to(dev)
e0 = FilmEngine(PUBLISHED, device=0); e0.set_weights(w)
it4 = DeviceInterpolator(e0, align=64, block_shape=[2, 2])
e1 = FilmEngine(PUBLISHED, device=0); e1.set_weights(w)
e2 = FilmEngine(PUBLISHED, device=0); e2.set_weights(w)
ita = DeviceInterpolator(e1, align=64, block_shape=[1, 2]); itb = DeviceInterpolator(e2, align=64, block_shape=[1, 2])
top0, top1 = x0[:, :540].contiguous(), x1[:, :540].contiguous()
bot0, bot1 = x0[:, 540:].contiguous(), x1[:, 540:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run4(n):
    for _ in range(n): out = it4(x0, x1)
    return out
def run22(n):
    for _ in range(n):
        with torch.cuda.stream(s1): a = ita(top0, top1)
        with torch.cuda.stream(s2): b = itb(bot0, bot1)
    return a, b
for name, fn in (('one batch of 4 tiles', run4), ('2 + 2 tiles on two streams', run22), ('one batch of 4 tiles', run4), ('2 + 2 tiles on two streams', run22)):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(20); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'{name}: {dt * 1e3:.2f} ms per frame', flush=True)
o4 = run4(1); a, b = run22(1); torch.cuda.synchronize()
print('same bits:', bool(torch.equal(o4[:, :540], a)) and bool(torch.equal(o4[:, 540:], b)))
