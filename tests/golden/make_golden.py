"""Generates tests/golden/tiny_golden.npz.

PROVENANCE: the reference (TF 2.6.2 + TFA 0.15.0 + Drive checkpoints) cannot run in this environment and
ships no golden vectors, so these vectors are produced by the repo's own CPU oracle
(oracle/film_oracle.py) - they pin the oracle against silent drift and give the GPU tests a committed
fixture; they are NOT outputs of the reference ("parity unpinned").

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'frame-interpolation_amd'))

from film_hip import weights as W  # noqa: E402
from film_hip.options import TINY  # noqa: E402
from oracle import film_oracle as fo  # noqa: E402


def inputs():
    rng = np.random.default_rng(2024)
    x0 = rng.random((2, 32, 40, 3), dtype=np.float32)
    x1 = np.roll(x0, (1, -2), axis=(1, 2)) + rng.normal(0, 0.02, x0.shape).astype(np.float32)
    return x0, x1.astype(np.float32)


def main():
    opt = fo.Options(pyramid_levels=TINY.pyramid_levels, fusion_pyramid_levels=TINY.fusion_pyramid_levels,
                     specialized_levels=TINY.specialized_levels, sub_levels=TINY.sub_levels,
                     flow_convs=TINY.flow_convs, flow_filters=TINY.flow_filters, filters=TINY.filters)
    w = W.make_synthetic_weights(TINY, seed=0)
    x0, x1 = inputs()
    img, aux = fo.film_forward(x0, x1, w, opt, return_aux=True)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    img64 = fo.film_forward(x0.astype(np.float64), x1.astype(np.float64), w64, opt)
    out = {'image': img, 'image_f64': img64.astype(np.float32),
           'forward_flow0': aux['forward_flow_pyramid'][0], 'backward_flow0': aux['backward_flow_pyramid'][0],
           'x0_warped': aux['x0_warped'], 'feat2_img0': aux['feature_pyramids'][0][2]}
    np.savez_compressed(os.path.join(HERE, 'tiny_golden.npz'), **out)
    print({k: v.shape for k, v in out.items()}, 'f32 vs f64 max|d|:', np.abs(img - img64).max())


if __name__ == '__main__':
    main()
