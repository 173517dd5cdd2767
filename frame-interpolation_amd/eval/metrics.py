"""Evaluation metrics of the reference's benchmark loop (eval/eval_cli.py:160-170 -> losses/losses.py:72-113),
restated in numpy: l1, l2, ssim, psnr on float32 [B,H,W,C] images in [0,1].

tf.image.psnr(a, b, max_val=1)   = 10 * log10(1 / mean((a-b)^2))            per image
tf.image.ssim(a, b, max_val=1)   : 11x11 Gaussian window (sigma 1.5, normalised), 'VALID' filtering per channel,
                                   k1 = 0.01, k2 = 0.03, luminance * contrast-structure averaged over the valid
                                   region, then over channels (tensorflow/python/ops/image_ops_impl.py, _ssim_per_channel).
"""
from typing import Callable, Dict, List, Tuple

import numpy as np


def l1(image: np.ndarray, y: np.ndarray) -> float:
    """losses.l1_loss (losses/losses.py:72-74)."""
    return float(np.mean(np.abs(image.astype(np.float32) - y.astype(np.float32)), dtype=np.float64))


def l2(image: np.ndarray, y: np.ndarray) -> float:
    """losses.l2_loss (losses/losses.py:97-99)."""
    d = image.astype(np.float32) - y.astype(np.float32)
    return float(np.mean(d * d, dtype=np.float64))


def psnr(image: np.ndarray, y: np.ndarray, max_val: float = 1.0) -> float:
    """losses.psnr_loss (losses/losses.py:110-113): mean over the batch of the per-image PSNR."""
    image, y = _batched(image), _batched(y)
    d = image.astype(np.float64) - y.astype(np.float64)
    mse = np.mean(d * d, axis=(1, 2, 3))
    with np.errstate(divide='ignore'):
        return float(np.mean(20.0 * np.log10(max_val) - 10.0 * np.log10(mse)))


def _batched(x: np.ndarray) -> np.ndarray:
    return x if x.ndim == 4 else x[None]


def _gauss_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    g = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = -(g * g) / (2.0 * sigma * sigma)
    g = np.exp(g - g.max())
    return g / g.sum()


def _filter_valid(x: np.ndarray, g: np.ndarray) -> np.ndarray:
    """Separable 'VALID' correlation of [B,H,W,C] with the 1-D window g along H and W."""
    n = g.shape[0]
    h, w = x.shape[1], x.shape[2]
    out = np.zeros((x.shape[0], h - n + 1, w, x.shape[3]), dtype=np.float64)
    for i in range(n):
        out += g[i] * x[:, i:i + h - n + 1]
    out2 = np.zeros((x.shape[0], h - n + 1, w - n + 1, x.shape[3]), dtype=np.float64)
    for i in range(n):
        out2 += g[i] * out[:, :, i:i + w - n + 1]
    return out2


def ssim(image: np.ndarray, y: np.ndarray, max_val: float = 1.0, filter_size: int = 11, filter_sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> float:
    """losses.ssim_loss (losses/losses.py:102-107): mean over the batch of tf.image.ssim."""
    a = _batched(image).astype(np.float64)
    b = _batched(y).astype(np.float64)
    if a.shape[1] < filter_size or a.shape[2] < filter_size:
        raise ValueError(f'ssim needs images of at least {filter_size} x {filter_size}')
    g = _gauss_window(filter_size, filter_sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = _filter_valid(a, g), _filter_valid(b, g)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 * mean0 + mean1 * mean1
    luminance = (num0 + c1) / (den0 + c1)
    num1 = _filter_valid(a * b, g) * 2.0
    den1 = _filter_valid(a * a + b * b, g)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    per_channel = np.mean(luminance * cs, axis=(1, 2))   # [B, C]
    return float(np.mean(np.mean(per_channel, axis=-1)))


METRICS: Dict[str, Callable[[np.ndarray, np.ndarray], float]] = {'l1': l1, 'l2': l2, 'ssim': ssim, 'psnr': psnr}


def test_losses(names: List[str]) -> List[Tuple[str, Callable[[np.ndarray, np.ndarray], float]]]:
    """losses.test_losses for the metrics that need no VGG weights (losses/losses.py:116-160)."""
    out = []
    for n in names:
        if n not in METRICS:
            raise ValueError(f"Invalid loss name '{n}' (available here: {sorted(METRICS)}; vgg / style need the "
                             'VGG-19 MATLAB weights, which are outside the inference hot path)')
        out.append((n, METRICS[n]))
    return out
