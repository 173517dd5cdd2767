"""Device-resident plumbing around the engine: PyTorch-ROCm tensors for memory and streams only.

The hot path is entirely inside libfilm_hip.so (film_interpolate: pad / patch / model / crop / stitch);
torch is used here only to hold frames in HBM and to hand raw pointers + the current stream to the C-ABI.
pad_to_align / image_to_patches / patches_to_image below are torch restatements of the reference's layout
rules, kept for tests and for callers that want the patches themselves.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .engine import FilmEngine


def pinned_frame(shape) -> 'np.ndarray':
    """A float32 numpy array in page-locked host memory (the tensor that owns it rides along as `.base`): frames handed to
    Interpolator.__call__ / FilmEngine.interpolate_frames in such arrays move at the PCIe rate without the runtime pinning pageable pages per call."""
    return torch.empty(tuple(int(v) for v in shape), dtype=torch.float32, pin_memory=True).numpy()


def pad_to_align(x: torch.Tensor, align: int) -> Tuple[torch.Tensor, Tuple[int, int, int, int]]:
    """[B,H,W,C] -> zero padded to multiples of align, offset pad//2 (eval/interpolator.py:30-63)."""
    b, h, w, c = x.shape
    hp = (align - h % align) if h % align else 0
    wp = (align - w % align) if w % align else 0
    oy, ox = hp // 2, wp // 2
    if hp == 0 and wp == 0:
        return x.contiguous(), (0, 0, h, w)
    out = torch.zeros((b, h + hp, w + wp, c), dtype=x.dtype, device=x.device)
    out[:, oy:oy + h, ox:ox + w, :] = x
    return out, (oy, ox, h, w)


def image_to_patches(image: torch.Tensor, block_shape: List[int]) -> torch.Tensor:
    """[1,H,W,C] -> [bh*bw, H/bh, W/bw, C], row-major blocks (eval/interpolator.py:66-99)."""
    bh, bw = block_shape
    _, h, w, c = image.shape
    ph, pw = h // bh, w // bw
    assert h == ph * bh, 'block_height=%d should evenly divide height=%d.' % (bh, h)
    assert w == pw * bw, 'block_width=%d should evenly divide width=%d.' % (bw, w)
    return image[0].reshape(bh, ph, bw, pw, c).permute(0, 2, 1, 3, 4).reshape(bh * bw, ph, pw, c).contiguous()


def patches_to_image(patches: torch.Tensor, block_shape: List[int]) -> torch.Tensor:
    """inverse of image_to_patches (eval/interpolator.py:102-126)."""
    bh, bw = block_shape
    _, ph, pw, c = patches.shape
    return patches.reshape(bh, bw, ph, pw, c).permute(0, 2, 1, 3, 4).reshape(1, bh * ph, bw * pw, c).contiguous()


class DeviceInterpolator:
    """Interpolator.__call__ semantics (eval/interpolator.py:178-209) on frames that already live
    in HBM: float32 CUDA(=HIP) tensors in, float32 CUDA tensor out, no host round trip.

    Asynchronous on the current torch stream."""

    def __init__(self, engine: FilmEngine, align: Optional[int] = None, block_shape: Optional[List[int]] = None):
        self._engine = engine
        self._align = align or None
        self._block_shape = block_shape or None

    def _run(self, x0: torch.Tensor, x1: torch.Tensor, block_shape) -> torch.Tensor:
        assert x0.is_cuda and x0.dtype == torch.float32 and x0.shape == x1.shape and x0.shape[-1] == 3
        x0 = x0.contiguous()
        x1 = x1.contiguous()
        b, h, w, _ = x0.shape
        out = torch.empty_like(x0)
        stream = torch.cuda.current_stream(x0.device).cuda_stream
        self._engine.interpolate_frames_device(x0.data_ptr(), x1.data_ptr(), b, h, w, out.data_ptr(),
                                               align=self._align, block_shape=block_shape, stream=stream)
        return out

    def interpolate(self, x0: torch.Tensor, x1: torch.Tensor) -> torch.Tensor:
        """Interpolator.interpolate (eval/interpolator.py:152-176): pad to align, model, crop - inside
        film_interpolate (HIP kernels read these tensors and write the result tensor directly)."""
        return self._run(x0, x1, None)

    def batch(self, x0: torch.Tensor, x1: torch.Tensor) -> torch.Tensor:
        """Extension: Interpolator.__call__ applied to every pair of a batch (tiled or not) in one call."""
        bs = self._block_shape
        return self._run(x0, x1, bs if bs is not None and bs[0] * bs[1] > 1 else None)

    def __call__(self, x0: torch.Tensor, x1: torch.Tensor) -> torch.Tensor:
        if self._block_shape is not None and self._block_shape[0] * self._block_shape[1] > 1:
            if x0.shape[0] != 1:    # the reference's tiled path takes one pair (eval/interpolator.py:96-98); see .batch
                raise ValueError(f'the tiled path takes one frame pair per call, got a batch of {x0.shape[0]}; '
                                 'use DeviceInterpolator.batch for batches of tiled frames')
            return self._run(x0, x1, self._block_shape)
        return self._run(x0, x1, None)
