"""CPU ORACLE for the FILM inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``frame-interpolation_amd/``) never imports anything from
``oracle/`` and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (google-research/frame-interpolation) ships no unit
tests, no golden vectors and no weights, and its arithmetic lives in two pip packages
that are not installed here and not vendored under /root/reference:
``tensorflow==2.6.2`` (requirements.txt:2) and ``tensorflow-addons==0.15.0``
(requirements.txt:4).  This oracle therefore *restates* the published semantics of the
TF/TFA ops the reference calls (Conv2D 'same', AveragePooling2D, tf.image.resize
bilinear/nearest with half_pixel_centers, tfa.image.dense_image_warp) and follows the
reference's graph code line by line; each function cites the reference file:line it
follows.  It is cross-checked against independent PyTorch built-ins
(tests/test_oracle_ops.py) but cannot be checked against TF itself in this image.

Conventions: NHWC float32 numpy arrays, weights in TF's HWIO layout
``[kh, kw, cin, cout]``, flows are (dx, dy) in channels (0, 1).
All elementwise arithmetic is done in the array dtype (float32 by default; pass
float64 arrays + float64 weights to obtain the error-floor reference).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # torch is used ONLY as a fast CPU conv engine (oneDNN); see conv2d_same_numpy for
    import torch  # the pure-numpy restatement that tests compare it with.
    import torch.nn.functional as F
except Exception:  # pragma: no cover
    torch = None
    F = None


# ----------------------------------------------------------------------------------
# Options  (models/film_net/options.py:20-80, training/config/film_net-L1.gin:17-23)
# ----------------------------------------------------------------------------------
@dataclasses.dataclass
class Options:
    """Hyper-parameters; defaults are the *published* config
    (training/config/film_net-L1.gin:17-23 - identical in the VGG/Style gins)."""
    pyramid_levels: int = 7
    fusion_pyramid_levels: int = 5
    specialized_levels: int = 3
    sub_levels: int = 4
    flow_convs: Sequence[int] = (3, 3, 3, 3)
    flow_filters: Sequence[int] = (32, 64, 128, 256)
    filters: int = 64


# ----------------------------------------------------------------------------------
# Op restatements (TF 2.6 / TFA 0.15 semantics)
# ----------------------------------------------------------------------------------
def leaky_relu(x: np.ndarray, alpha: float = 0.2) -> np.ndarray:
    """tf.nn.leaky_relu(x, alpha=0.2)  (feature_extractor.py:89-90,
    pyramid_flow_estimator.py:45-46, fusion.py:49-50): max(alpha*x, x)."""
    return np.maximum(x * x.dtype.type(alpha), x)


def same_padding(k: int) -> Tuple[int, int]:
    """TF 'SAME' padding for stride 1: total k-1, before = (k-1)//2, after = rest.
    k=3 -> (1,1); k=2 -> (0,1) (bottom/right); k=1 -> (0,0)."""
    total = k - 1
    return total // 2, total - total // 2


def conv2d_same_numpy(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    """Pure-numpy Conv2D(padding='same', strides=1): cross-correlation, zero padding.
    out[n,y,x,co] = sum_{dy,dx,ci} xpad[n,y+dy,x+dx,ci] * w[dy,dx,ci,co] + b[co]."""
    kh, kw, cin, cout = w.shape
    pt, pb = same_padding(kh)
    pl, pr = same_padding(kw)
    n, h, wd, c = x.shape
    assert c == cin
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((n, h, wd, cout), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            out += xp[:, dy:dy + h, dx:dx + wd, :] @ w[dy, dx]
    if b is not None:
        out += b
    return out


def conv2d_same(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray],
                activation: Optional[str] = None) -> np.ndarray:
    """tf.keras.layers.Conv2D(padding='same') (+ optional leaky-relu activation).
    feature_extractor.py:93-99, pyramid_flow_estimator.py:66-72, fusion.py:83-101."""
    kh, kw, cin, cout = w.shape
    if torch is None:
        y = conv2d_same_numpy(x, w, b)
    else:
        pt, pb = same_padding(kh)
        pl, pr = same_padding(kw)
        xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
        if pt or pb or pl or pr:
            xt = F.pad(xt, (pl, pr, pt, pb))
        wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1).contiguous()
        bt = None if b is None else torch.from_numpy(np.ascontiguousarray(b))
        y = F.conv2d(xt, wt, bt).permute(0, 2, 3, 1).contiguous().numpy()
    if activation == 'leaky':
        y = leaky_relu(y)
    elif activation is not None:
        raise ValueError(activation)
    return y


def avg_pool2x2(x: np.ndarray) -> np.ndarray:
    """tf.keras.layers.AveragePooling2D(pool_size=2, strides=2, padding='valid')
    (util.py:39-40, feature_extractor.py:138-139). Sum in row-major window order."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    x = x[:, :h2 * 2, :w2 * 2, :]
    s = ((x[:, 0::2, 0::2, :] + x[:, 0::2, 1::2, :]) + x[:, 1::2, 0::2, :]) + x[:, 1::2, 1::2, :]
    return s * x.dtype.type(0.25)


def _resize_axis_weights(in_size: int, out_size: int, dtype) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """compute_interpolation_weights with HalfPixelScaler (TF resize_bilinear_op):
    in = (o+0.5)*scale-0.5; lower=max(floor(in),0); upper=min(ceil(in),in_size-1);
    lerp = in - floor(in)."""
    scale = np.float32(in_size) / np.float32(out_size)
    o = np.arange(out_size, dtype=np.float32)
    src = (o + np.float32(0.5)) * scale - np.float32(0.5)
    f = np.floor(src)
    lower = np.maximum(f, 0).astype(np.int64)
    upper = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
    lerp = (src - f).astype(dtype)
    return lower, upper, lerp


def resize_bilinear(x: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """tf.image.resize(images, size) - TF2 default: bilinear, half_pixel_centers=True,
    antialias=False (pyramid_flow_estimator.py:155, util.py:113)."""
    n, h, w, c = x.shape
    oh, ow = size
    ylo, yhi, yl = _resize_axis_weights(h, oh, x.dtype)
    xlo, xhi, xl = _resize_axis_weights(w, ow, x.dtype)
    xl = xl[None, None, :, None]
    yl = yl[None, :, None, None]
    top_l = x[:, ylo][:, :, xlo]
    top_r = x[:, ylo][:, :, xhi]
    bot_l = x[:, yhi][:, :, xlo]
    bot_r = x[:, yhi][:, :, xhi]
    top = top_l + (top_r - top_l) * xl
    bot = bot_l + (bot_r - bot_l) * xl
    return top + (bot - top) * yl


def resize_nearest(x: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """tf.image.resize(..., NEAREST_NEIGHBOR) (fusion.py:133-134), TF2 half-pixel:
    src = min(floor((o+0.5)*scale), in-1)."""
    n, h, w, c = x.shape
    oh, ow = size
    ys = np.minimum(np.floor((np.arange(oh, dtype=np.float32) + np.float32(0.5)) *
                             (np.float32(h) / np.float32(oh))), h - 1).astype(np.int64)
    xs = np.minimum(np.floor((np.arange(ow, dtype=np.float32) + np.float32(0.5)) *
                             (np.float32(w) / np.float32(ow))), w - 1).astype(np.int64)
    return x[:, ys][:, :, xs]


def dense_image_warp(image: np.ndarray, flow_yx_neg: np.ndarray) -> np.ndarray:
    """tfa.image.dense_image_warp(image, flow) (TFA 0.15.0): query = grid - flow with
    flow in (dy, dx) order, then interpolate_bilinear(indexing='ij'):
    floor clamped to [0, size-2], alpha = clip(q - floor, 0, 1),
    top = ax*(tr-tl)+tl; bot = ax*(br-bl)+bl; out = ay*(bot-top)+top."""
    n, h, w, c = image.shape
    assert h >= 2 and w >= 2, 'dense_image_warp needs a grid of at least 2x2'
    dt = image.dtype
    gy, gx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    qy = gy.astype(flow_yx_neg.dtype)[None] - flow_yx_neg[..., 0]
    qx = gx.astype(flow_yx_neg.dtype)[None] - flow_yx_neg[..., 1]

    def axis(q, size):
        fl = np.minimum(np.maximum(np.floor(q), 0), size - 2)
        alpha = np.clip((q - fl).astype(dt), 0, 1)
        return fl.astype(np.int64), alpha

    fy, ay = axis(qy, h)
    fx, ax = axis(qx, w)
    b = np.arange(n)[:, None, None]
    tl = image[b, fy, fx]
    tr = image[b, fy, fx + 1]
    bl = image[b, fy + 1, fx]
    br = image[b, fy + 1, fx + 1]
    ax = ax[..., None]
    ay = ay[..., None]
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def warp(image: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """models/film_net/util.py:48-82: backward warp; FILM negates and swaps the (dx,dy)
    flow because TFA subtracts a (dy,dx) flow (util.py:70)."""
    return dense_image_warp(image, -flow[..., ::-1])


# ----------------------------------------------------------------------------------
# Graph restatement (models/film_net/*.py)
# ----------------------------------------------------------------------------------
def build_image_pyramid(image: np.ndarray, opt: Options) -> List[np.ndarray]:
    """models/film_net/util.py:23-45."""
    pyr = []
    for i in range(opt.pyramid_levels):
        pyr.append(image)
        if i < opt.pyramid_levels - 1:
            image = avg_pool2x2(image)
    return pyr


def _wb(weights: Dict[str, np.ndarray], name: str):
    return weights[name + '/kernel'], weights[name + '/bias']


def sub_tree_extractor(image: np.ndarray, n: int, weights, opt: Options) -> List[np.ndarray]:
    """feature_extractor.py:125-147 (SubTreeExtractor.call)."""
    head = image
    pyramid = []
    for i in range(n):
        w, b = _wb(weights, f'feat_net/sub_extractor/cfeat_conv_{2 * i}')
        head = conv2d_same(head, w, b, 'leaky')
        w, b = _wb(weights, f'feat_net/sub_extractor/cfeat_conv_{2 * i + 1}')
        head = conv2d_same(head, w, b, 'leaky')
        pyramid.append(head)
        if i < n - 1:
            head = avg_pool2x2(head)
    return pyramid


def feature_extractor(image_pyramid: List[np.ndarray], weights, opt: Options) -> List[np.ndarray]:
    """feature_extractor.py:163-193 (FeatureExtractor.call)."""
    sub_pyramids = []
    for i in range(len(image_pyramid)):
        capped = min(len(image_pyramid) - i, opt.sub_levels)
        sub_pyramids.append(sub_tree_extractor(image_pyramid[i], capped, weights, opt))
    feature_pyramid = []
    for i in range(len(image_pyramid)):
        features = sub_pyramids[i][0]
        for j in range(1, opt.sub_levels):
            if j <= i:
                features = np.concatenate([features, sub_pyramids[i - j][j]], axis=-1)
        feature_pyramid.append(features)
    return feature_pyramid


def predictor_name(level: int, opt: Options) -> str:
    """pyramid_flow_estimator.py:109-123: levels >= specialized_levels share weights."""
    if level < opt.specialized_levels:
        return f'predict_flow/flow_predictor_{level}'
    return 'predict_flow/flow_predictor_shared'


def flow_estimator(features_a, features_b, weights, prefix: str, num_convs: int) -> np.ndarray:
    """pyramid_flow_estimator.py:85-98 (FlowEstimator.call): concat, num_convs x
    (3x3 + leaky), 1x1 (nf/2) + leaky, 1x1 (2) linear."""
    net = np.concatenate([features_a, features_b], axis=-1)
    for i in range(num_convs + 1):
        w, b = _wb(weights, f'{prefix}/conv_{i}')
        net = conv2d_same(net, w, b, 'leaky')
    w, b = _wb(weights, f'{prefix}/conv_{num_convs + 1}')
    return conv2d_same(net, w, b, None)


def pyramid_flow_estimator(fa: List[np.ndarray], fb: List[np.ndarray], weights,
                           opt: Options) -> List[np.ndarray]:
    """pyramid_flow_estimator.py:125-163 (PyramidFlowEstimator.call)."""
    levels = len(fa)

    def nconv(level):
        idx = min(level, opt.specialized_levels)
        return opt.flow_convs[idx]

    v = flow_estimator(fa[-1], fb[-1], weights, predictor_name(levels - 1, opt), nconv(levels - 1))
    residuals = [v]
    for i in reversed(range(0, levels - 1)):
        size = fa[i].shape[1:3]
        v = resize_bilinear(v.dtype.type(2) * v, size)
        warped = warp(fb[i], v)
        v_res = flow_estimator(fa[i], warped, weights, predictor_name(i, opt), nconv(i))
        residuals.append(v_res)
        v = v_res + v
    return list(reversed(residuals))


def flow_pyramid_synthesis(residual_pyramid: List[np.ndarray]) -> List[np.ndarray]:
    """models/film_net/util.py:106-117."""
    flow = residual_pyramid[-1]
    out = [flow]
    for res in reversed(residual_pyramid[:-1]):
        flow = resize_bilinear(flow.dtype.type(2) * flow, res.shape[1:3])
        flow = res + flow
        out.append(flow)
    return list(reversed(out))


def multiply_pyramid(pyramid: List[np.ndarray], scalar: np.ndarray) -> List[np.ndarray]:
    """models/film_net/util.py:85-103: per-batch scalar * image."""
    return [im * scalar.astype(im.dtype)[:, None, None, None] for im in pyramid]


def pyramid_warp(feature_pyramid, flow_pyramid):
    """models/film_net/util.py:120-134."""
    return [warp(f, fl) for f, fl in zip(feature_pyramid, flow_pyramid)]


def concatenate_pyramids(p1, p2):
    """models/film_net/util.py:137-143."""
    return [np.concatenate([a, b], axis=-1) for a, b in zip(p1, p2)]


def fusion(pyramid: List[np.ndarray], weights, opt: Options) -> np.ndarray:
    """fusion.py:103-140 (Fusion.call)."""
    levels = opt.fusion_pyramid_levels
    if len(pyramid) != levels:
        raise ValueError('Fusion called with different number of pyramid levels '
                         f'{len(pyramid)} than it was configured for, {levels}.')
    net = pyramid[-1]
    for i in reversed(range(0, levels - 1)):
        size = pyramid[i].shape[1:3]
        net = resize_nearest(net, size)
        w, b = _wb(weights, f'fusion/convs_{i}_0')
        net = conv2d_same(net, w, b, None)
        net = np.concatenate([pyramid[i], net], axis=-1)
        w, b = _wb(weights, f'fusion/convs_{i}_1')
        net = conv2d_same(net, w, b, 'leaky')
        w, b = _wb(weights, f'fusion/convs_{i}_2')
        net = conv2d_same(net, w, b, 'leaky')
    w, b = _wb(weights, 'fusion/output_conv')
    return conv2d_same(net, w, b, None)


def film_forward(x0: np.ndarray, x1: np.ndarray, weights: Dict[str, np.ndarray],
                 opt: Optional[Options] = None, return_aux: bool = False):
    """models/film_net/interpolator.py:89-207 (create_model): the whole network at
    t=0.5 ('time' is ignored by the reference, interpolator.py:102,163).

    Returns 'image' [B,H,W,3] (un-clipped); with return_aux also the dict of
    intermediates that tests use as per-stage taps."""
    opt = opt or Options()
    if opt.pyramid_levels < opt.fusion_pyramid_levels:
        raise ValueError('config.pyramid_levels must be greater than or equal to '
                         'config.fusion_pyramid_levels.')
    div = 2 ** (opt.pyramid_levels - 1)
    if x0.shape[1] % div or x0.shape[2] % div:
        raise ValueError(f'input height/width must be divisible by {div} (options.py:36-37)')
    image_pyramids = [build_image_pyramid(x0, opt), build_image_pyramid(x1, opt)]
    feature_pyramids = [feature_extractor(image_pyramids[0], weights, opt),
                        feature_extractor(image_pyramids[1], weights, opt)]
    fwd_res = pyramid_flow_estimator(feature_pyramids[0], feature_pyramids[1], weights, opt)
    bwd_res = pyramid_flow_estimator(feature_pyramids[1], feature_pyramids[0], weights, opt)
    L = opt.fusion_pyramid_levels
    fwd_flow_pyr = flow_pyramid_synthesis(fwd_res)[:L]
    bwd_flow_pyr = flow_pyramid_synthesis(bwd_res)[:L]
    mid_time = np.full((x0.shape[0],), 0.5, dtype=x0.dtype)
    backward_flow = multiply_pyramid(bwd_flow_pyr, mid_time)
    forward_flow = multiply_pyramid(fwd_flow_pyr, 1 - mid_time)
    pyramids_to_warp = [
        concatenate_pyramids(image_pyramids[0][:L], feature_pyramids[0][:L]),
        concatenate_pyramids(image_pyramids[1][:L], feature_pyramids[1][:L]),
    ]
    forward_warped = pyramid_warp(pyramids_to_warp[0], backward_flow)
    backward_warped = pyramid_warp(pyramids_to_warp[1], forward_flow)
    aligned = concatenate_pyramids(forward_warped, backward_warped)
    aligned = concatenate_pyramids(aligned, backward_flow)
    aligned = concatenate_pyramids(aligned, forward_flow)
    prediction = fusion(aligned, weights, opt)
    image = prediction[..., :3]
    if not return_aux:
        return image
    aux = {
        'image': image,
        'image_pyramids': image_pyramids,
        'feature_pyramids': feature_pyramids,
        'forward_residual_flow_pyramid': fwd_res,
        'backward_residual_flow_pyramid': bwd_res,
        'forward_flow_pyramid': fwd_flow_pyr,
        'backward_flow_pyramid': bwd_flow_pyr,
        'x0_warped': forward_warped[0][..., 0:3],
        'x1_warped': backward_warped[0][..., 0:3],
        'aligned_pyramid': aligned,
    }
    return image, aux


# ----------------------------------------------------------------------------------
# eval/interpolator.py restatement (pad / tile wrapper around the model call)
# ----------------------------------------------------------------------------------
def pad_to_align(x: np.ndarray, align: int):
    """eval/interpolator.py:30-63 (_pad_to_align): zero pad, offset = pad//2."""
    assert np.ndim(x) == 4
    assert align > 0, 'align must be a positive number.'
    height, width = x.shape[-3:-1]
    hp = (align - height % align) if height % align != 0 else 0
    wp = (align - width % align) if width % align != 0 else 0
    oy, ox = hp // 2, wp // 2
    padded = np.zeros((x.shape[0], height + hp, width + wp, x.shape[3]), dtype=x.dtype)
    padded[:, oy:oy + height, ox:ox + width, :] = x
    bbox = {'offset_height': oy, 'offset_width': ox, 'target_height': height, 'target_width': width}
    return padded, bbox


def image_to_patches(image: np.ndarray, block_shape) -> np.ndarray:
    """eval/interpolator.py:66-99: row-major non-overlapping blocks stacked on batch."""
    bh, bw = block_shape
    height, width, channel = image.shape[-3:]
    ph, pw = height // bh, width // bw
    assert height == ph * bh, 'block_height=%d should evenly divide height=%d.' % (bh, height)
    assert width == pw * bw, 'block_width=%d should evenly divide width=%d.' % (bw, width)
    img = image.reshape(-1, height, width, channel)[0]
    return (img.reshape(bh, ph, bw, pw, channel).transpose(0, 2, 1, 3, 4)
            .reshape(bh * bw, ph, pw, channel).copy())


def patches_to_image(patches: np.ndarray, block_shape) -> np.ndarray:
    """eval/interpolator.py:102-126."""
    bh, bw = block_shape
    ph, pw, channel = patches.shape[-3:]
    return (patches.reshape(bh, bw, ph, pw, channel).transpose(0, 2, 1, 3, 4)
            .reshape(1, bh * ph, bw * pw, channel).copy())


class OracleInterpolator:
    """eval/interpolator.py:129-209 (class Interpolator) over the oracle network."""

    def __init__(self, weights: Dict[str, np.ndarray], align: Optional[int] = None,
                 block_shape=None, opt: Optional[Options] = None):
        self._weights = weights
        self._opt = opt or Options()
        self._align = align or None
        self._block_shape = block_shape or None

    def interpolate(self, x0, x1, dt):
        if self._align is not None:
            x0, bbox = pad_to_align(x0, self._align)
            x1, _ = pad_to_align(x1, self._align)
        image = film_forward(x0, x1, self._weights, self._opt)
        if self._align is not None:
            oy, ox = bbox['offset_height'], bbox['offset_width']
            image = image[:, oy:oy + bbox['target_height'], ox:ox + bbox['target_width'], :]
        return np.ascontiguousarray(image)

    def __call__(self, x0, x1, dt):
        if self._block_shape is not None and np.prod(self._block_shape) > 1:
            x0p = image_to_patches(x0, self._block_shape)
            x1p = image_to_patches(x1, self._block_shape)
            outs = [self.interpolate(a[np.newaxis], b[np.newaxis], dt) for a, b in zip(x0p, x1p)]
            return patches_to_image(np.concatenate(outs, axis=0), self._block_shape)
        return self.interpolate(x0, x1, dt)


def psnr(a: np.ndarray, b: np.ndarray, max_val: float = 1.0) -> float:
    """tf.image.psnr as used by losses/losses.py:110-113."""
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    if mse == 0:
        return float('inf')
    return 10.0 * np.log10(max_val * max_val / mse)
