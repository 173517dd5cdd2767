#!/usr/bin/env python
"""tests/golden/oracle_big_untiled.npz: the CPU oracle (oracle/film_oracle.py, float32) on ONE untiled 3840x2240 frame
pair - a size whose level-0 activation buffers exceed 4 GiB in the HIP engine (feat0 / warped0 4.4 GB, aligned0 4.95 GB).
The oracle needs about an hour on 8 cores and ~30 GB for such a frame (its numpy warps are single-threaded), so the GPU test compares against this fixture instead of running
it: a stride-16 pixel sample of the image + float64 row / column sums of the full image (every pixel enters them).
Inputs: tests/inputs.frame_pair(1, 2240, 3840, seed=41) and film_hip.weights.make_synthetic_weights(PUBLISHED, seed=0),
both seeded, so nothing but the output is stored.

  python tools/make_big_golden.py            # 64 minutes on 8 cores, 23 GB
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

H, WD, SEED, STRIDE = 2240, 3840, 41, 16


def film_forward_lean(x0, x1, weights, opt=None):
    """oracle.film_oracle.film_forward with the same calls in the same order, but the aligned pyramid is assembled level by
    level and everything no longer needed is dropped before the fusion decoder: film_forward keeps every pyramid (and three
    generations of concatenated copies) alive, > 60 GB at 3840x2240.  Same arithmetic - concatenation is data movement and
    the warp is per channel - and tests/test_oracle_ops.py checks that the two give the same bits on a small frame."""
    import gc
    opt = opt or fo.Options()
    image_pyramids = [fo.build_image_pyramid(x0, opt), fo.build_image_pyramid(x1, opt)]
    feature_pyramids = [fo.feature_extractor(image_pyramids[0], weights, opt),
                        fo.feature_extractor(image_pyramids[1], weights, opt)]
    fwd_res = fo.pyramid_flow_estimator(feature_pyramids[0], feature_pyramids[1], weights, opt)
    bwd_res = fo.pyramid_flow_estimator(feature_pyramids[1], feature_pyramids[0], weights, opt)
    L = opt.fusion_pyramid_levels
    fwd_flow_pyr = fo.flow_pyramid_synthesis(fwd_res)[:L]
    bwd_flow_pyr = fo.flow_pyramid_synthesis(bwd_res)[:L]
    del fwd_res, bwd_res
    mid_time = np.full((x0.shape[0],), 0.5, dtype=x0.dtype)
    backward_flow = fo.multiply_pyramid(bwd_flow_pyr, mid_time)
    forward_flow = fo.multiply_pyramid(fwd_flow_pyr, 1 - mid_time)
    aligned = []
    for l in range(L):
        a = np.concatenate([image_pyramids[0][l], feature_pyramids[0][l]], axis=-1)
        fw = fo.warp(a, backward_flow[l])
        del a
        b = np.concatenate([image_pyramids[1][l], feature_pyramids[1][l]], axis=-1)
        bw = fo.warp(b, forward_flow[l])
        del b
        aligned.append(np.concatenate([fw, bw, backward_flow[l], forward_flow[l]], axis=-1))
        del fw, bw
        feature_pyramids[0][l] = feature_pyramids[1][l] = None
        gc.collect()
    del feature_pyramids, image_pyramids
    gc.collect()
    return fo.fusion(aligned, weights, opt)[..., :3]


def main():
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    x0, x1 = inputs.frame_pair(1, H, WD, SEED)
    t0 = time.time()
    img = film_forward_lean(x0, x1, w, fo.Options())
    print(f'oracle: {time.time() - t0:.0f} s, image {img.shape}, mean {img.mean():.6f}')
    out = os.path.join(ROOT, 'tests', 'golden', 'oracle_big_untiled.npz')
    np.savez_compressed(out,
                        shape=np.asarray(img.shape), stride=np.asarray(STRIDE),
                        in_checksum=np.asarray([float(x0.astype(np.float64).sum()), float(x1.astype(np.float64).sum())]),
                        sample=img[:, ::STRIDE, ::STRIDE, :].astype(np.float32),
                        rowsum=img.astype(np.float64).sum(axis=2), colsum=img.astype(np.float64).sum(axis=1))
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
