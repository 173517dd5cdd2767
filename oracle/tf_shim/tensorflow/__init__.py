"""`tensorflow` stand-in (TEST INFRASTRUCTURE ONLY, see oracle/tf_shim/__init__.py).

Eager, CPU, NHWC; a tensor is a numpy array (subclass with ``.numpy()``), so the reference's own
slicing / arithmetic (``-flow[..., ::-1]``, ``2*v``, ``prediction[..., :3]``, ``mid_time[:, 0]``) is
numpy's.  Layers follow the Keras contract the reference relies on:

* ``Layer(name=...)``; ``__call__`` forwards to ``call``; child layers are found by walking the
  instance attributes (lists included), the way Keras' checkpoint tracking does - this yields both
  the *name chain* (``feat_net/sub_extractor/cfeat_conv_0``) and the *object-graph path*
  (``extract_sublevels/convs/0``) of every Conv2D.
* ``Conv2D`` gets its ``kernel`` [kh,kw,cin,cout] / ``bias`` [cout] on first call from the installed
  weight provider (``set_weight_provider``), keyed by those two strings.
* ``keras.Model(inputs, outputs)`` in this eager world is just the holder of the computed outputs;
  ``saved_model.load(path)`` returns whatever the installed loader builds (``set_saved_model_loader``).

Each op cites the reference call site it serves.
"""
import io as _io
import types as _types

import numpy as np
import torch
import torch.nn.functional as F

__film_shim__ = True
__version__ = '0.0-film-shim'

float32 = np.float32
float64 = np.float64
uint8 = np.uint8
int32 = np.int32


class Tensor(np.ndarray):
    def numpy(self):
        return np.asarray(self)


def _wrap(a) -> Tensor:
    return np.asarray(a).view(Tensor)


def _t(a):
    a = np.ascontiguousarray(np.asarray(a))
    return torch.from_numpy(a)


def convert_to_tensor(x, dtype=None):
    return _wrap(np.asarray(x, dtype=dtype))


constant = convert_to_tensor


# --------------------------------------------------------------------------- plain array ops
def shape(x):                       # util.py:82,112  pyramid_flow_estimator.py:154  fusion.py:132
    return np.asarray(np.shape(x), dtype=np.int32)


def reshape(x, shape, name=None):   # util.py:82  eval/interpolator.py:96-98,119-124
    return _wrap(np.reshape(np.asarray(x), [int(s) for s in shape]))


def concat(values, axis, name=None):  # util.py:142  feature_extractor.py:191  flow:95  fusion:136
    return _wrap(np.concatenate([np.asarray(v) for v in values], axis=axis))


def transpose(x, perm=None):        # util.py:100
    return _wrap(np.transpose(np.asarray(x), perm))


def ones_like(x):                   # interpolator.py:163
    return _wrap(np.ones_like(np.asarray(x)))


def cast(x, dtype):                 # eval/util.py:40
    return _wrap(np.asarray(x).astype(dtype))


def split(value, num_or_size_splits, axis=0):   # eval/interpolator.py:95,121
    return [_wrap(a) for a in np.split(np.asarray(value), num_or_size_splits, axis=axis)]


def stack(values, axis=0):          # eval/interpolator.py:96,122
    return _wrap(np.stack([np.asarray(v) for v in values], axis=axis))


def space_to_batch(input, block_shape, paddings, name=None):   # eval/interpolator.py:94
    """tf.space_to_batch (= space_to_batch_nd) as documented: zero-pad the spatial dims, reshape to
    [batch, H/b0, b0, W/b1, b1, C], permute to [b0, b1, batch, H/b0, W/b1, C], flatten the first
    three dims into the new batch."""
    x = np.asarray(input)
    b0, b1 = [int(v) for v in block_shape]
    (pt, pb), (pl, pr) = paddings
    x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    n, h, w, c = x.shape
    assert h % b0 == 0 and w % b1 == 0
    x = x.reshape(n, h // b0, b0, w // b1, b1, c).transpose(2, 4, 0, 1, 3, 5)
    return _wrap(x.reshape(b0 * b1 * n, h // b0, w // b1, c))


def batch_to_space(input, block_shape, crops, name=None):      # eval/interpolator.py:125
    """Inverse of space_to_batch: [b0*b1*batch, h, w, C] -> [batch, h*b0, w*b1, C], then crop."""
    x = np.asarray(input)
    b0, b1 = [int(v) for v in block_shape]
    nb, h, w, c = x.shape
    n = nb // (b0 * b1)
    x = x.reshape(b0, b1, n, h, w, c).transpose(2, 3, 0, 4, 1, 5).reshape(n, h * b0, w * b1, c)
    (ct, cb), (cl, cr) = crops
    return _wrap(x[:, ct:x.shape[1] - cb, cl:x.shape[2] - cr, :])


# --------------------------------------------------------------------------- tf.nn / tf.image / tf.io
def _leaky_relu(features, alpha=0.2, name=None):   # feature_extractor.py:90 flow:46 fusion:50
    x = np.asarray(features)
    return _wrap(F.leaky_relu(_t(x), negative_slope=float(alpha)).numpy())


nn = _types.SimpleNamespace(leaky_relu=_leaky_relu)


class _ResizeMethod:
    BILINEAR = 'bilinear'
    NEAREST_NEIGHBOR = 'nearest'


def _resize(images, size, method='bilinear', preserve_aspect_ratio=False, antialias=False, name=None):
    """tf.image.resize, TF2 semantics (half_pixel_centers=True): pyramid_flow_estimator.py:155,
    util.py:113 (bilinear), fusion.py:133-134 (nearest)."""
    x = np.asarray(images)
    oh, ow = int(size[0]), int(size[1])
    xt = _t(x).permute(0, 3, 1, 2)
    if method == 'bilinear':
        y = F.interpolate(xt, size=(oh, ow), mode='bilinear', align_corners=False, antialias=False)
    elif method == 'nearest':
        y = F.interpolate(xt, size=(oh, ow), mode='nearest')
    else:
        raise NotImplementedError(method)
    return _wrap(y.permute(0, 2, 3, 1).contiguous().numpy())


def _pad_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
    """eval/interpolator.py:54: zeros around the image."""
    x = np.asarray(image)
    h, w = x.shape[-3], x.shape[-2]
    after_h = target_height - offset_height - h
    after_w = target_width - offset_width - w
    if min(offset_height, offset_width, after_h, after_w) < 0:
        raise ValueError('target size smaller than the image')
    pads = [(0, 0)] * (x.ndim - 3) + [(offset_height, after_h), (offset_width, after_w), (0, 0)]
    return _wrap(np.pad(x, pads))


def _crop_to_bounding_box(image, offset_height, offset_width, target_height, target_width):
    """eval/interpolator.py:175."""
    x = np.asarray(image)
    return _wrap(x[..., offset_height:offset_height + target_height,
                   offset_width:offset_width + target_width, :])


image = _types.SimpleNamespace(resize=_resize, ResizeMethod=_ResizeMethod,
                               pad_to_bounding_box=_pad_to_bounding_box,
                               crop_to_bounding_box=_crop_to_bounding_box)


def _read_file(filename):           # eval/util.py:38
    with open(filename, 'rb') as f:
        return f.read()


def _write_file(filename, contents):  # eval/util.py:59
    with open(filename, 'wb') as f:
        f.write(contents)


def _decode_image(contents, channels=None):  # eval/util.py:39
    from PIL import Image, PngImagePlugin
    PngImagePlugin.MAX_TEXT_CHUNK = 1 << 30      # photos/one.png carries a large zTXt chunk
    im = Image.open(_io.BytesIO(contents))
    if channels == 3:
        im = im.convert('RGB')
    return _wrap(np.asarray(im, dtype=np.uint8))


def _encode_png(img):               # eval/util.py:58
    from PIL import Image
    buf = _io.BytesIO()
    Image.fromarray(np.asarray(img, dtype=np.uint8)).save(buf, format='PNG')
    return buf.getvalue()


io = _types.SimpleNamespace(read_file=_read_file, write_file=_write_file,
                            decode_image=_decode_image, encode_png=_encode_png)


# --------------------------------------------------------------------------- keras
_WEIGHT_PROVIDER = None
_CALL_DEPTH = 0
TOP_LEVEL_LAYERS = []     # layers called at depth 0, in call order (= the functional model's layers)
CONV_LOG = []             # (name_chain, object_path, kernel_shape) of every Conv2D that was built


def set_weight_provider(fn):
    """fn(name_chain: str, object_path: str, kernel_shape) -> (kernel HWIO, bias)."""
    global _WEIGHT_PROVIDER
    _WEIGHT_PROVIDER = fn
    del TOP_LEVEL_LAYERS[:]
    del CONV_LOG[:]


class Layer:
    _auto_ids = {}

    def __init__(self, name=None, **_kw):
        self.name = name
        self._name_chain = None
        self._object_path = None

    def _children(self):
        for attr, val in vars(self).items():
            if isinstance(val, Layer):
                yield attr, val
            elif isinstance(val, (list, tuple)):
                yield from _walk_list(attr, val)

    def _assign_paths(self, chain, path):
        self._name_chain = chain
        self._object_path = path
        for attr, child in self._children():
            if child._object_path is None:      # first owner wins (shared predictor: index 3)
                cname = child.name if child.name is not None else attr
                child._assign_paths(f'{chain}/{cname}', f'{path}/{attr}' if path else attr)

    def __call__(self, *args, **kwargs):
        global _CALL_DEPTH
        if _CALL_DEPTH == 0 and any(True for _ in self._children()):
            if self not in TOP_LEVEL_LAYERS:
                TOP_LEVEL_LAYERS.append(self)
                self._assign_paths(self.name, '')
        _CALL_DEPTH += 1
        try:
            return self.call(*args, **kwargs)
        finally:
            _CALL_DEPTH -= 1


def _walk_list(prefix, seq):
    for i, v in enumerate(seq):
        if isinstance(v, Layer):
            yield f'{prefix}/{i}', v
        elif isinstance(v, (list, tuple)):
            yield from _walk_list(f'{prefix}/{i}', v)


class Conv2D(Layer):
    """tf.keras.layers.Conv2D, strides 1: cross-correlation, zero 'same' padding
    (total k-1, floor before / rest after: k=2 pads bottom/right only) or 'valid'."""

    def __init__(self, filters, kernel_size, strides=1, padding='valid', activation=None, name=None, **_kw):
        super().__init__(name=name)
        self.filters = int(filters)             # pyramid_flow_estimator.py:77 passes num_filters/2
        ks = kernel_size if isinstance(kernel_size, (list, tuple)) else (kernel_size, kernel_size)
        self.kernel_size = (int(ks[0]), int(ks[1]))
        assert strides in (1, (1, 1), [1, 1])
        self.padding = padding.lower()
        self.activation = activation
        self.kernel = None
        self.bias = None

    def call(self, x):
        x = np.asarray(x)
        if self.kernel is None:
            kshape = self.kernel_size + (x.shape[-1], self.filters)
            if self._name_chain is None:
                raise RuntimeError('Conv2D called outside a tracked top-level layer')
            k, b = _WEIGHT_PROVIDER(self._name_chain, self._object_path, kshape)
            assert tuple(k.shape) == kshape and tuple(b.shape) == (self.filters,), (self._name_chain, k.shape, kshape)
            self.kernel, self.bias = np.asarray(k), np.asarray(b)
            CONV_LOG.append((self._name_chain, self._object_path, kshape))
        xt = _t(x).permute(0, 3, 1, 2)
        kh, kw = self.kernel_size
        if self.padding == 'same':
            pt, pl = (kh - 1) // 2, (kw - 1) // 2
            pb, pr = (kh - 1) - pt, (kw - 1) - pl
            if pt or pb or pl or pr:
                xt = F.pad(xt, (pl, pr, pt, pb))
        elif self.padding != 'valid':
            raise NotImplementedError(self.padding)
        wt = _t(self.kernel.astype(x.dtype)).permute(3, 2, 0, 1).contiguous()
        y = F.conv2d(xt, wt, _t(self.bias.astype(x.dtype)))
        y = _wrap(y.permute(0, 2, 3, 1).contiguous().numpy())
        return self.activation(y) if self.activation is not None else y


class AveragePooling2D(Layer):      # util.py:39-40  feature_extractor.py:138-139
    def __init__(self, pool_size=2, strides=None, padding='valid', name=None, **_kw):
        super().__init__(name=name)
        assert padding == 'valid'
        self.pool_size, self.strides = pool_size, strides or pool_size

    def call(self, x):
        y = F.avg_pool2d(_t(np.asarray(x)).permute(0, 3, 1, 2), self.pool_size, self.strides)
        return _wrap(y.permute(0, 2, 3, 1).contiguous().numpy())


class Lambda(Layer):                # util.py:80-81  interpolator.py:163
    def __init__(self, function, name=None, **_kw):
        super().__init__(name=name)
        self.function = function

    def call(self, x):
        return self.function(x)


class Model:
    """tf.keras.Model(inputs=..., outputs=...) of an eagerly executed graph: holds the outputs."""

    def __init__(self, inputs=None, outputs=None, **_kw):
        self.inputs, self.outputs = inputs, outputs
        self.layers = list(TOP_LEVEL_LAYERS)


def _Input(*_a, **_kw):
    raise NotImplementedError('symbolic tf.keras.Input is not provided: the shim runs create_model '
                              'eagerly on concrete arrays')


keras = _types.SimpleNamespace(
    layers=_types.SimpleNamespace(Layer=Layer, Conv2D=Conv2D, AveragePooling2D=AveragePooling2D,
                                  Lambda=Lambda),
    Model=Model, Input=_Input)
Model_ = Model

# --------------------------------------------------------------------------- saved_model
_SAVED_MODEL_LOADER = None


def set_saved_model_loader(fn):
    """fn(path) -> callable(inputs: dict, training=False) -> dict of tensors."""
    global _SAVED_MODEL_LOADER
    _SAVED_MODEL_LOADER = fn


def _load(path, *a, **kw):          # eval/interpolator.py:148
    if _SAVED_MODEL_LOADER is None:
        raise RuntimeError('tf_shim: no saved-model loader installed')
    return _SAVED_MODEL_LOADER(path)


saved_model = _types.SimpleNamespace(load=_load)
compat = _types.SimpleNamespace(v2=_types.SimpleNamespace(saved_model=saved_model))
