"""tfa.image.dense_image_warp stand-in (TEST INFRASTRUCTURE, see oracle/tf_shim/__init__.py).

TFA 0.15.0 semantics being stated: ``output[b, j, i, c] = bilinear(image[b], j - flow[b,j,i,0],
i - flow[b,j,i,1])`` with the query clamped so that samples outside the image replicate the edge
(interpolate_bilinear clamps floor to [0, size-2] and alpha to [0, 1]).  Implemented with
``F.grid_sample(padding_mode='border', align_corners=True)``, which clips the coordinate to
[0, size-1] before the same bilinear blend."""
import numpy as np
import torch
import torch.nn.functional as F

import tensorflow as tf

__film_shim__ = True


def dense_image_warp(image, flow, name=None):
    image = np.asarray(image)
    flow = np.asarray(flow)
    n, h, w, c = image.shape
    if h < 2 or w < 2:
        raise ValueError('Grid must be at least 2x2 (tfa interpolate_bilinear)')
    tdt = torch.float64 if image.dtype == np.float64 else torch.float32
    img = torch.from_numpy(np.ascontiguousarray(image)).to(tdt).permute(0, 3, 1, 2)
    fl = torch.from_numpy(np.ascontiguousarray(flow)).to(tdt)
    gy, gx = torch.meshgrid(torch.arange(h, dtype=tdt), torch.arange(w, dtype=tdt), indexing='ij')
    qy = gy[None] - fl[..., 0]
    qx = gx[None] - fl[..., 1]
    grid = torch.stack([2 * qx / (w - 1) - 1, 2 * qy / (h - 1) - 1], dim=-1)
    out = F.grid_sample(img, grid, mode='bilinear', padding_mode='border', align_corners=True)
    return tf._wrap(out.permute(0, 2, 3, 1).contiguous().numpy().astype(image.dtype))
