"""Deterministic structured test inputs shared by the golden generator (tools/make_ref_golden.py), the
CPU tests and the GPU parity tests.  Frames have smooth regions, hard edges and texture; the second
frame is the first one moved by a global translation of >= 8 px with a differently moving foreground
patch, plus sensor noise - so that the warps sample far from the identity (SURVEY.md 8d inputs)."""
import numpy as np


def _smooth_noise(rng, h, w, c, cell):
    gh, gw = h // cell + 3, w // cell + 3
    g = rng.random((gh, gw, c), dtype=np.float32)
    ys = (np.arange(h, dtype=np.float32) + 0.5) / cell
    xs = (np.arange(w, dtype=np.float32) + 0.5) / cell
    y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
    ty, tx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    a = g[y0][:, x0] * (1 - tx) + g[y0][:, x0 + 1] * tx
    b = g[y0 + 1][:, x0] * (1 - tx) + g[y0 + 1][:, x0 + 1] * tx
    return (a * (1 - ty) + b * ty).astype(np.float32)


def scene(h, w, seed):
    """One [h, w, 3] float32 frame in [0, 1] on a canvas larger than the frame by 32 px each side."""
    rng = np.random.default_rng(seed)
    H, W = h + 64, w + 64
    img = 0.55 * _smooth_noise(rng, H, W, 3, 48) + 0.25 * _smooth_noise(rng, H, W, 3, 7)
    img += 0.08 * rng.random((H, W, 3), dtype=np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for _ in range(6):                                   # hard-edged discs and bars
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(6, max(8, min(H, W) / 5))
        col = rng.random(3).astype(np.float32)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img[m] = 0.3 * img[m] + 0.7 * col
    x = int(rng.uniform(0.2, 0.8) * W)
    img[:, x:x + 5] *= 0.35
    return np.clip(img, 0, 1).astype(np.float32)


def frame_pair(b, h, w, seed, shift=(9, -12), fg_shift=(-5, 14), noise=0.01):
    """x0, x1 [b, h, w, 3] float32: x1 = x0 translated by `shift` (dy, dx) px, a foreground rectangle
    translated by `fg_shift` instead, + gaussian noise."""
    x0s, x1s = [], []
    for i in range(b):
        canvas = scene(h, w, seed * 1000 + i)
        rng = np.random.default_rng(seed * 1000 + i + 500)
        a = canvas[32:32 + h, 32:32 + w]
        dy, dx = shift
        bfr = canvas[32 - dy:32 - dy + h, 32 - dx:32 - dx + w].copy()
        fy, fx, fh, fw = h // 3, w // 4, max(8, h // 4), max(8, w // 5)
        gy, gx = fg_shift
        ys = np.clip(np.arange(fy, fy + fh) + gy, 0, h - 1)
        xs = np.clip(np.arange(fx, fx + fw) + gx, 0, w - 1)
        bfr[np.ix_(ys, xs)] = a[fy:fy + fh, fx:fx + fw]
        bfr = bfr + rng.normal(0, noise, bfr.shape).astype(np.float32)
        x0s.append(a.copy())
        x1s.append(np.clip(bfr, 0, 1).astype(np.float32))
    return np.stack(x0s), np.stack(x1s)


def read_png(path):
    """eval/util.py:29-41 read_image semantics: 8-bit RGB -> float32 / 255."""
    from PIL import Image, PngImagePlugin
    PngImagePlugin.MAX_TEXT_CHUNK = 1 << 30      # photos/one.png carries a large zTXt chunk
    return (np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8).astype(np.float32) /
            np.float32(255.0))
