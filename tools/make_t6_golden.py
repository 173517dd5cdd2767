#!/usr/bin/env python
"""tests/golden/oracle_t6_tile.npz: ONE tile's whole depth-6 recursion tree of BASELINE configs[4], by the CPU oracle.

configs[4] = a 3840x2160 pair, `--block_height 4 --block_width 4 --times_to_interpolate 6`: 63 generated frames x 16 tiles =
1008 tile-forwards of 960x540 (padded to 960x576).  The reference re-tiles every mid-frame on the same grid
(eval/util.py:62-91 calling Interpolator.__call__, eval/interpolator.py:192-206), so a tile's 2^T - 1 frames depend on that
tile of the two inputs only.  This script runs the oracle (oracle/film_oracle.py, float32) over the recursion tree of tile
TILE of the seeded pair tests/inputs.frame_pair(1, 2160, 3840, seed=4) in the reference's order - each mid-frame =
crop(film_forward(pad(a), pad(b))), its raw un-clipped float output feeding the deeper levels - and stores, per generated
frame: a stride-12 pixel sample + float64 row / column sums of the full frame (every pixel enters them).  Round-off is fed
back six times; the GPU test (tests/test_gpu_t6.py) prints max|delta| per generation against this fixture.

  python tools/make_t6_golden.py          # 63 oracle tile-forwards: about 25 minutes on 8 cores
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'frame-interpolation_amd'), os.path.join(ROOT, 'tests')]
from film_hip import weights as W  # noqa: E402
from film_hip.options import PUBLISHED  # noqa: E402
from oracle import film_oracle as fo  # noqa: E402
import inputs  # noqa: E402

H, WD, SEED, BLOCK, TILE, T, STRIDE, ALIGN = 2160, 3840, 4, (4, 4), 5, 6, 12, 64


def tile_of(frame, block, tile):
    """Row-major tile `tile` of a [H,W,3] frame (image_to_patches order, eval/interpolator.py:66-99)."""
    bh, bw = block
    ph, pw = frame.shape[0] // bh, frame.shape[1] // bw
    ty, tx = divmod(tile, bw)
    return np.ascontiguousarray(frame[ty * ph:(ty + 1) * ph, tx * pw:(tx + 1) * pw])


def main():
    T_ = int(os.environ.get('T6_DEPTH', T))
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    x0, x1 = inputs.frame_pair(1, H, WD, SEED)
    a, b = tile_of(x0[0], BLOCK, TILE), tile_of(x1[0], BLOCK, TILE)
    opt = fo.Options()
    n = 2 ** T_ + 1
    frames = {0: a, n - 1: b}
    depth_of = {}
    t0 = time.time()

    it = fo.OracleInterpolator(w, align=ALIGN, block_shape=None, opt=opt)   # one tile = Interpolator.interpolate (:152-176)
    dt = np.full((1,), 0.5, np.float32)

    def mid(fa, fb):
        return it(fa[None], fb[None], dt)[0]

    # depth-first, the reference's order (eval/util.py:82-91); the order does not change any value
    def rec(lo, hi, d):
        if hi - lo < 2:
            return
        m = (lo + hi) // 2
        frames[m] = mid(frames[lo], frames[hi])
        depth_of[m] = d
        print(f'frame {m:3d} depth {d}  ({len(depth_of)}/{n - 2})  {time.time() - t0:.0f} s', flush=True)
        rec(lo, m, d + 1)
        rec(m, hi, d + 1)

    rec(0, n - 1, 1)
    idx = sorted(depth_of)
    full = np.stack([frames[k] for k in idx])
    out = os.path.join(ROOT, 'tests', 'golden', 'oracle_t6_tile.npz' if T_ == T else f'oracle_t{T_}_tile.npz')
    np.savez_compressed(
        out, frame_index=np.asarray(idx), depth=np.asarray([depth_of[k] for k in idx]), stride=np.asarray(STRIDE),
        tile=np.asarray(TILE), block=np.asarray(BLOCK), shape=np.asarray(full.shape),
        in_checksum=np.asarray([float(a.astype(np.float64).sum()), float(b.astype(np.float64).sum())]),
        sample=full[:, ::STRIDE, ::STRIDE, :].astype(np.float32),
        rowsum=full.astype(np.float64).sum(axis=2), colsum=full.astype(np.float64).sum(axis=1),
        vmin=full.min(axis=(1, 2, 3)), vmax=full.max(axis=(1, 2, 3)))
    print('wrote', out, os.path.getsize(out), 'bytes', f'{time.time() - t0:.0f} s')


if __name__ == '__main__':
    main()
