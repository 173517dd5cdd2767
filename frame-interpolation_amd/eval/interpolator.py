"""Drop-in for the reference's eval/interpolator.py, backed by the MI355X HIP engine.

Usage (unchanged from the reference, eval/interpolator.py:15-24):
  model_path='/tmp/saved_model/'
  it = Interpolator(model_path)
  result_batch = it.interpolate(image_batch_0, image_batch_1, batch_dt)

  image_batch_* are numpy float32 tensors in (B,H,W,C) layout, batch_dt is the sub-frame time,
  (B,) layout (ignored by film_net, which always predicts t=0.5 -
  models/film_net/interpolator.py:102,163).

What changed underneath: ``tf.saved_model.load`` + ``self._model(...)`` (reference
eval/interpolator.py:148,170-172) are replaced by ``film_create/film_set_weight/film_finalize`` +
``film_interpolate`` of libfilm_hip.so (include/film_hip.h).  Padding, cropping and patch
(un)folding follow the reference's exact layout rules and run as HIP kernels inside that call; the
numpy helpers below (_pad_to_align, image_to_patches, patches_to_image) stay importable, as in the
reference, and are what the tests hold the device path to.  The patches of a tiled frame are
independent (reference loops over them with B=1, :199-202) and go through the GPU as ONE batch.  There is no TensorFlow and no CPU fallback: without the built library and a gfx950
device, construction raises.
"""
from typing import List, Optional

import numpy as np

from film_hip import weights as weights_lib
from film_hip.engine import FilmEngine
from film_hip.options import Options, PUBLISHED


def _pad_to_align(x, align):
  """Zero-pads H and W of a [B,H,W,C] batch up to multiples of `align`, centred with offset
  pad // 2 (reference eval/interpolator.py:30-63, tf.image.pad_to_bounding_box).  Returns the
  padded batch and the crop box that undoes it."""
  assert np.ndim(x) == 4
  assert align > 0, 'align must be a positive number.'

  height, width = x.shape[-3:-1]
  height_to_pad = (align - height % align) if height % align != 0 else 0
  width_to_pad = (align - width % align) if width % align != 0 else 0
  off_h, off_w = height_to_pad // 2, width_to_pad // 2
  if height_to_pad or width_to_pad:
    padded_x = np.zeros((x.shape[0], height + height_to_pad, width + width_to_pad, x.shape[3]), dtype=x.dtype)
    padded_x[:, off_h:off_h + height, off_w:off_w + width, :] = x
  else:
    padded_x = x
  bbox_to_crop = {
      'offset_height': off_h,
      'offset_width': off_w,
      'target_height': height,
      'target_width': width
  }
  return padded_x, bbox_to_crop


def _crop_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
  return image[:, offset_height:offset_height + target_height, offset_width:offset_width + target_width, :]


def image_to_patches(image: np.ndarray, block_shape: List[int]) -> np.ndarray:
  """[.., H, W, C] image -> [bh*bw, H/bh, W/bw, C]: row-major, non-overlapping blocks (what the
  reference's tf.space_to_batch construction yields, eval/interpolator.py:66-99)."""
  block_height, block_width = block_shape
  num_blocks = block_height * block_width

  height, width, channel = image.shape[-3:]
  patch_height, patch_width = height // block_height, width // block_width

  assert height == (
      patch_height * block_height
  ), 'block_height=%d should evenly divide height=%d.' % (block_height, height)
  assert width == (
      patch_width * block_width
  ), 'block_width=%d should evenly divide width=%d.' % (block_width, width)

  img = np.reshape(image, (-1, height, width, channel))[0]
  patches = img.reshape(block_height, patch_height, block_width, patch_width, channel)
  patches = patches.transpose(0, 2, 1, 3, 4)
  return np.ascontiguousarray(patches.reshape(num_blocks, patch_height, patch_width, channel))


def patches_to_image(patches: np.ndarray, block_shape: List[int]) -> np.ndarray:
  """Unfolds patches (stacked along batch) into an image [1, H, W, C] (eval/interpolator.py:102-126)."""
  block_height, block_width = block_shape
  patch_height, patch_width, channel = patches.shape[-3:]
  image = patches.reshape(block_height, block_width, patch_height, patch_width, channel)
  image = image.transpose(0, 2, 1, 3, 4)
  return np.ascontiguousarray(
      image.reshape(1, block_height * patch_height, block_width * patch_width, channel))


class Interpolator:
  """A class for generating interpolated frames between two input frames.

  Same constructor and call signatures as the reference class (eval/interpolator.py:129-209).
  """

  def __init__(self, model_path: str,
               align: Optional[int] = None,
               block_shape: Optional[List[int]] = None,
               *, device: int = 0, options: Optional[Options] = None,
               weights=None, precision: int = 0, engine: Optional[FilmEngine] = None) -> None:
    """Loads the weights of a saved model into a HIP engine.

    Args:
      model_path: directory of the model: a TF2 SavedModel (variables bundle) or a directory
        holding film_weights.npz (see film_hip.weights.load_weights).
      align: 'If >1, pad the input size so it divides with this before inference.'
      block_shape: Number of patches along the (height, width) to sid-divide input images.
      device: (extension) HIP device ordinal.
      options: (extension) architecture hyper-parameters; default = published film_net.
      weights: (extension) an already loaded {name: array} dict; model_path is then ignored.
      precision: (extension) engine precision mode: 0 = fp32 MFMA (default), 1 = bf16x6, 2 = bf16x3
        (film_set_option "precision", include/film_hip.h).
      engine: (extension) an engine that already holds its weights (a rank that received them by broadcast,
        film_hip.sharding.sharded_interpolator); model_path and weights are then ignored.
    """
    self._options = options or PUBLISHED
    if engine is not None:
      self._engine = engine
    else:
      self._engine = FilmEngine(self._options, device=device)
      if weights is None and weights_lib.is_saved_model(model_path):
        # a TF2 SavedModel directory: the native reader behind the C-ABI (film_load_bundle) replaces
        # tf.compat.v2.saved_model.load(model_path) (reference :148) - no TensorFlow, no Python parsing
        self._engine.load_bundle(model_path)
      else:
        if weights is None:
          weights = weights_lib.load_weights(model_path, self._options)
        weights_lib.validate_weights(weights, self._options)
        self._engine.set_weights(weights)
    if precision:
      self._engine.set_option('precision', int(precision))
    self._align = align or None
    self._block_shape = block_shape or None

  @property
  def engine(self) -> FilmEngine:
    return self._engine

  @property
  def align(self) -> Optional[int]:
    return self._align

  @property
  def block_shape(self) -> Optional[List[int]]:
    return self._block_shape

  def interpolate(self, x0: np.ndarray, x1: np.ndarray,
                  dt: np.ndarray) -> np.ndarray:
    """Mid-frame for every pair of the batch (reference: Interpolator.interpolate, :152-176).

    x0, x1 are float32 [B,H,W,3]; dt is [B] and unused by film_net.  Pads to `align`, runs the
    model, crops back - one film_interpolate call.  Output is float32 [B,H,W,3], NOT clipped to [0,1].
    """
    if self._align is not None:
      assert np.ndim(x0) == 4
      assert self._align > 0, 'align must be a positive number.'
    return self._engine.interpolate_frames(x0, x1, align=self._align)

  def __call__(self, x0: np.ndarray, x1: np.ndarray,
               dt: np.ndarray) -> np.ndarray:
    """Same contract as interpolate(); with block_shape set, the (single) image is split into
    non-overlapping patches that are padded, interpolated and cropped independently and then
    stitched back (reference: Interpolator.__call__, :178-209)."""
    if self._block_shape is not None and np.prod(self._block_shape) > 1:
      block_height, block_width = self._block_shape
      height, width = x0.shape[-3:-1]
      # the reference's checks (image_to_patches, eval/interpolator.py:84-89)
      assert height == (height // block_height) * block_height, (
          'block_height=%d should evenly divide height=%d.' % (block_height, height))
      assert width == (width // block_width) * block_width, (
          'block_width=%d should evenly divide width=%d.' % (block_width, width))
      # The reference's tiled path takes ONE frame pair: image_to_patches reshapes the space_to_batch
      # result to [num_blocks, ...] (eval/interpolator.py:94-98), which fails for B > 1.  Same here -
      # no silent truncation; film_hip.torch_io.DeviceInterpolator.batch tiles whole batches.
      x0 = np.reshape(x0, (-1, height, width, x0.shape[-1]))
      x1 = np.reshape(x1, (-1, height, width, x1.shape[-1]))
      if x0.shape[0] != 1:
        raise ValueError('the tiled path (block_shape) takes one frame pair per call, got a batch of %d '
                         '(reference: image_to_patches reshape, eval/interpolator.py:96-98)' % x0.shape[0])
      # The reference runs the patches one by one with B=1; they are independent, so all of
      # them go through the engine as one batch (identical per-patch arithmetic).
      return self._engine.interpolate_frames(x0, x1, align=self._align, block_shape=self._block_shape)
    return self.interpolate(x0, x1, dt)
