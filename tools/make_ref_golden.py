"""Generates tests/golden/ref_*.npz by RUNNING THE REFERENCE'S OWN PYTHON.

    python tools/make_ref_golden.py [--backend shim|tf] [--only NAME ...] [--reference /root/reference]

What runs: ``models/film_net/interpolator.py:create_model`` (and everything it calls in
feature_extractor.py / pyramid_flow_estimator.py / fusion.py / util.py / options.py), the hyper-parameters
parsed from ``training/config/film_net-L1.gin:17-23``, ``eval/interpolator.py`` (``Interpolator``,
``_pad_to_align``, ``image_to_patches``, ``patches_to_image``) and ``eval/util.py`` (``_recursive_generator``,
``interpolate_recursively_from_memory``, ``read_image``, ``write_image``) - imported unmodified from
``--reference`` as the namespace package ``reference``.

Backend ``shim`` (default; the only one possible in the build container, which has no TensorFlow):
``oracle/tf_shim`` supplies `tensorflow` / `tensorflow_addons` / `gin` with the ~25 entry points those files
call, implemented with PyTorch-CPU built-ins.  The vectors therefore pin the GRAPH (wiring, channel orders,
weight naming, padding / patch / recursion code = the reference's own) and give a second, independently
written statement of the OP semantics; they are not TensorFlow outputs.
Backend ``tf``: the same cases on a real TensorFlow 2.x + tensorflow-addons install (weights assigned to the
Keras model by object path); writes ``tests/golden/tf_*.npz``, which tests/test_ref_golden_cpu.py prefers
when present.  Untested here (no TF) - it is the recipe for the first TF-capable box.

Weights are the repo's seeded synthetic set (film_hip/weights.py; the Drive checkpoints are unreachable),
inputs come from tests/inputs.py and the reference's photos/one.png, two.png (copied to tests/golden/).
Large outputs are stored as a stride-4 pixel sample plus float64 row / column sums of the full tensor.
"""
import argparse
import importlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'frame-interpolation_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from film_hip import weights as W          # noqa: E402
from film_hip import options as O          # noqa: E402
import inputs as TI                        # noqa: E402

STRIDE = 4


# ------------------------------------------------------------------------------------------ reference
class Reference:
    """The reference modules, imported from `ref_dir` under the package name of its directory."""

    def __init__(self, ref_dir, backend):
        self.backend = backend
        if backend == 'shim':
            from oracle import tf_shim
            tf_shim.install()
        parent, pkg = os.path.split(os.path.abspath(ref_dir).rstrip('/'))
        sys.path.insert(0, parent)
        self.gin = importlib.import_module('gin')
        self.tf = importlib.import_module('tensorflow')
        self.options = importlib.import_module(f'{pkg}.models.film_net.options')
        self.model = importlib.import_module(f'{pkg}.models.film_net.interpolator')
        self.interp = importlib.import_module(f'{pkg}.eval.interpolator')
        self.util = importlib.import_module(f'{pkg}.eval.util')
        self.gin_file = os.path.join(ref_dir, 'training', 'config', 'film_net-L1.gin')
        self.ref_dir = ref_dir

    def make_options(self, opt=None):
        """Published config: from the reference's gin file; other configs: explicit keyword arguments."""
        if opt is None:
            self.gin.clear_config()
            self.gin.parse_config_file(self.gin_file) if self.backend == 'shim' else \
                self.gin.parse_config_file(self.gin_file, skip_unknown=True)
            return self.options.Options()
        if self.backend == 'shim':
            self.gin.clear_config()
        return self.options.Options(
            pyramid_levels=opt.pyramid_levels, fusion_pyramid_levels=opt.fusion_pyramid_levels,
            specialized_levels=opt.specialized_levels, sub_levels=opt.sub_levels,
            flow_convs=list(opt.flow_convs), flow_filters=list(opt.flow_filters), filters=opt.filters,
            use_aux_outputs=True)

    # -- model = callable(inputs dict, training=False) -> dict, like the loaded SavedModel ----------
    def model_fn(self, weights, opt=None):
        config = self.make_options(opt)
        config.use_aux_outputs = True
        if self.backend == 'shim':
            return _ShimModel(self, weights, config)
        return _TfModel(self, weights, config)


def canonical_name(name_chain, object_path):
    """Keras name chain / object-graph path of a Conv2D -> canonical weight name (film_hip/weights.py).
    Named layers: the chain IS the canonical name; the decoder's unnamed layers (fusion.py:83-101) go by
    their attribute path convs/<i>/<j> and output_conv."""
    root = name_chain.split('/')[0]
    if root == 'fusion':
        parts = object_path.split('/')
        if parts[0] == 'convs':
            return f'fusion/convs_{parts[1]}_{parts[2]}'
        return f'fusion/{parts[0]}'
    return name_chain


class _ShimModel:
    def __init__(self, ref, weights, config):
        self.ref, self.weights, self.config = ref, weights, config
        self.conv_log = None

    def __call__(self, inputs, training=False):
        tf = self.ref.tf
        used = []

        def provider(chain, path, kshape):
            name = canonical_name(chain, path)
            used.append((name, chain, path))
            return self.weights[name + '/kernel'], self.weights[name + '/bias']
        tf.set_weight_provider(provider)
        x0, x1, t = (tf.convert_to_tensor(inputs[k]) for k in ('x0', 'x1', 'time'))
        m = self.ref.model.create_model(x0, x1, t, self.config)
        self.conv_log = used
        self.keras_model = m
        self.layers = [l.name for l in m.layers]
        return m.outputs


def walk_conv_layers(tf, top_layers, weights, on_conv):
    """The `--backend tf` naming walk: every Conv2D reachable from the three top-level Keras layers of create_model
    (feat_net, predict_flow, fusion) through instance attributes and (nested) lists - i.e. along the edges Keras' object graph
    tracks - gets its canonical name from its name chain (named layers) / attribute path (the decoder's unnamed ones) and is
    handed to on_conv(layer, kernel, bias); first owner wins (the shared flow predictor is listed four times).  Returns
    [(canonical name, name chain, attribute path)].  Written against tf.keras; tests/test_ref_golden_cpu.py runs it over
    oracle/tf_shim's layer objects, where it must reproduce the names the shim's own weight provider logged."""
    conv_log, seen = [], set()

    def walk(obj, chain, path):
        for attr, val in list(vars(obj).items()):
            items = [(attr, val)]
            if isinstance(val, (list, tuple)):
                items = []
                stack = [(attr, val)]
                while stack:
                    p, seq = stack.pop()
                    for i, v in enumerate(seq):
                        if isinstance(v, (list, tuple)):
                            stack.append((f'{p}/{i}', v))
                        else:
                            items.append((f'{p}/{i}', v))
            for p, v in items:
                if isinstance(v, tf.keras.layers.Conv2D):
                    if id(v) in seen:
                        continue
                    seen.add(id(v))
                    named = bool(v.name) and not v.name.startswith('conv2d')     # (Keras auto-names unnamed layers conv2d_<n>; the shim leaves None)
                    ch = f'{chain}/{v.name}' if named else f'{chain}/{p.split("/")[0]}'
                    name = canonical_name(ch, f'{path}/{p}' if path else p)
                    on_conv(v, weights[name + '/kernel'], weights[name + '/bias'])
                    conv_log.append((name, ch, f'{path}/{p}' if path else p))
                elif isinstance(v, tf.keras.layers.Layer) and not attr.startswith('_keras'):
                    if id(v) in seen:
                        continue
                    seen.add(id(v))
                    walk(v, f'{chain}/{v.name or p.split("/")[0]}', f'{path}/{p}' if path else p)
    for layer in top_layers:
        if layer.name in ('feat_net', 'predict_flow', 'fusion'):
            walk(layer, layer.name, '')
    return conv_log


class _TfModel:
    def __init__(self, ref, weights, config):
        tf = ref.tf
        x0 = tf.keras.Input(shape=(None, None, 3), dtype=tf.float32, name='x0')
        x1 = tf.keras.Input(shape=(None, None, 3), dtype=tf.float32, name='x1')
        t = tf.keras.Input(shape=(1,), dtype=tf.float32, name='time')
        self.m = ref.model.create_model(x0, x1, t, config)
        self.conv_log = walk_conv_layers(tf, self.m.layers, weights, lambda layer, k, b: layer.set_weights([k, b]))
        assert len(self.conv_log) == len(weights) // 2, (len(self.conv_log), len(weights) // 2)
        self.layers = [l.name for l in self.m.layers]

    def __call__(self, inputs, training=False):
        return self.m(inputs, training=training)


def np_(x):
    return np.asarray(x.numpy() if hasattr(x, 'numpy') else x)


# ------------------------------------------------------------------------------------------ storage
def sample(x):
    """Stride-4 pixel sample + float64 row / column sums of a [B,H,W,C] tensor."""
    x = np.asarray(x)
    return {'s4': np.ascontiguousarray(x[:, ::STRIDE, ::STRIDE, :]).astype(np.float32),
            'rowsum': x.astype(np.float64).sum(axis=2), 'colsum': x.astype(np.float64).sum(axis=1),
            'shape': np.asarray(x.shape, np.int64)}


def put(out, key, x, full):
    if full:
        out[key] = np.asarray(x, np.float32)
    else:
        for k, v in sample(x).items():
            out[f'{key}.{k}'] = v


def put_aux(out, res, full):
    put(out, 'image', np_(res['image']), full)
    put(out, 'x0_warped', np_(res['x0_warped']), full)
    put(out, 'x1_warped', np_(res['x1_warped']), full)
    for d in ('forward', 'backward'):
        for l, v in enumerate(res[f'{d}_residual_flow_pyramid']):
            put(out, f'{d}_residual_flow{l}', np_(v), full or l >= 2)
        for l, v in enumerate(res[f'{d}_flow_pyramid']):
            put(out, f'{d}_flow{l}', np_(v), full or l >= 2)


def checksum(*arrays):
    return np.asarray([float(np.asarray(a, np.float64).sum()) for a in arrays])


# ------------------------------------------------------------------------------------------ cases
def case_tiny(ref, prefix):
    """Small architecture (film_hip.options.TINY), B=2, ragged 32x40, all aux taps in full, fp32 + fp64."""
    w = W.make_synthetic_weights(O.TINY, seed=0)
    x0, x1 = TI.frame_pair(2, 32, 40, seed=11, shift=(3, -4), fg_shift=(-2, 5))
    t = np.full((2, 1), 0.5, np.float32)
    model = ref.model_fn(w, O.TINY)
    res = model({'x0': x0, 'x1': x1, 'time': t})
    out = {'in_checksum': checksum(x0, x1)}
    put_aux(out, res, True)
    out['conv_names'] = np.asarray([f'{n}|{c}|{p}' for n, c, p in model.conv_log])
    out['model_layers'] = np.asarray(model.layers)
    if ref.backend == 'shim':
        w64 = {k: v.astype(np.float64) for k, v in w.items()}
        r64 = ref.model_fn(w64, O.TINY)({'x0': x0.astype(np.float64), 'x1': x1.astype(np.float64),
                                         'time': t.astype(np.float64)})
        out['image_f64'] = np_(r64['image']).astype(np.float64)
    return out


def case_256(ref, prefix):
    """BASELINE configs[1]: published net (options from film_net-L1.gin), one 256x256 pair, aux taps."""
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    x0, x1 = TI.frame_pair(1, 256, 256, seed=1)
    t = np.full((1, 1), 0.5, np.float32)
    model = ref.model_fn(w)
    res = model({'x0': x0, 'x1': x1, 'time': t})
    out = {'in_checksum': checksum(x0, x1)}
    put_aux(out, res, False)
    out['image_full'] = np_(res['image']).astype(np.float32)
    out['conv_names'] = np.asarray([f'{n}|{c}|{p}' for n, c, p in model.conv_log])
    out['model_layers'] = np.asarray(model.layers)
    cfg = model.config
    out['gin_options'] = np.asarray([cfg.pyramid_levels, cfg.fusion_pyramid_levels, cfg.specialized_levels,
                                     cfg.sub_levels, cfg.filters] + list(cfg.flow_convs) + list(cfg.flow_filters))
    if ref.backend == 'shim':
        w64 = {k: v.astype(np.float64) for k, v in w.items()}
        r64 = ref.model_fn(w64)({'x0': x0.astype(np.float64), 'x1': x1.astype(np.float64),
                                 'time': t.astype(np.float64)})
        out['image_f64'] = np_(r64['image']).astype(np.float64)
        for d in ('forward', 'backward'):
            out[f'{d}_flow0_f64.s4'] = np_(r64[f'{d}_flow_pyramid'][0])[:, ::STRIDE, ::STRIDE].astype(np.float64)
    return out


def _interpolator(ref, weights, **kw):
    """reference eval/interpolator.py:Interpolator over the model built from `weights`."""
    model = ref.model_fn(weights)
    if ref.backend == 'shim':
        ref.tf.set_saved_model_loader(lambda path: model)
        return ref.interp.Interpolator('synthetic', **kw)
    it = ref.interp.Interpolator.__new__(ref.interp.Interpolator)
    it._model, it._align, it._block_shape = model, kw.get('align') or None, kw.get('block_shape') or None
    return it


def case_photos(ref, prefix):
    """BASELINE configs[0]: photos/one.png + two.png (1024x768), t=0.5, eval.interpolator_test path:
    Interpolator(align=64, block_shape=[1,1]), read_image / write_image of eval/util.py."""
    for n in ('one.png', 'two.png'):
        dst = os.path.join(GOLDEN, f'photo_{n}')
        if not os.path.isfile(dst):    # pixel-identical re-encode without the 2 MB zTXt metadata chunk
            from PIL import Image, PngImagePlugin
            PngImagePlugin.MAX_TEXT_CHUNK = 1 << 30
            src = Image.open(os.path.join(ref.ref_dir, 'photos', n))
            Image.fromarray(np.asarray(src.convert('RGB'))).save(dst, format='PNG', optimize=True)
            assert np.array_equal(np.asarray(Image.open(dst)), np.asarray(src.convert('RGB')))
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    a = ref.util.read_image(os.path.join(GOLDEN, 'photo_one.png'))
    b = ref.util.read_image(os.path.join(GOLDEN, 'photo_two.png'))
    it = _interpolator(ref, w, align=64, block_shape=[1, 1])
    mid = it(a[None], b[None], np.full((1,), 0.5, np.float32))
    out = {'in_checksum': checksum(a, b)}
    put(out, 'image', mid, False)
    tmp = tempfile.mkdtemp()
    ref.util.write_image(os.path.join(tmp, 'mid.png'), mid[0])
    out['image_u8.s4'] = (ref.util.read_image(os.path.join(tmp, 'mid.png')) * 255 + 0.5).astype(np.uint8)[::STRIDE, ::STRIDE]
    shutil.rmtree(tmp)
    return out


def case_1080p(ref, prefix):
    """BASELINE configs[2]: 1920x1080 pair, Interpolator(align=64, block_shape=[2,2]) - four 960x540 patches,
    each padded to 960x576 inside interpolate() (eval/interpolator.py:192-206)."""
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    x0, x1 = TI.frame_pair(1, 1080, 1920, seed=2, shift=(11, -17), fg_shift=(-9, 21))
    it = _interpolator(ref, w, align=64, block_shape=[2, 2])
    mid = it(x0, x1, np.full((1,), 0.5, np.float32))
    out = {'in_checksum': checksum(x0, x1)}
    put(out, 'image', mid, False)
    return out


def case_vimeo(ref, prefix):
    """BASELINE configs[3]: batch of 8 448x256 (W x H) triplet-sized pairs, align=64 (no padding needed)."""
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    x0, x1 = TI.frame_pair(8, 256, 448, seed=3)
    it = _interpolator(ref, w, align=64)
    mid = it(x0, x1, np.full((8,), 0.5, np.float32))
    out = {'in_checksum': checksum(x0, x1)}
    put(out, 'image', mid, False)
    return out


def case_recursive(ref, prefix):
    """eval/util.py:interpolate_recursively_from_memory, T=2, on a 2x1-tiled 200x176 pair
    (patches 100x176 -> padded to 128x192), frames in the reference's depth-first order, and the PNG
    rounding of write_image."""
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    x0, x1 = TI.frame_pair(1, 200, 176, seed=4, shift=(6, -8), fg_shift=(-4, 9))
    it = _interpolator(ref, w, align=64, block_shape=[2, 1])
    frames = list(ref.util.interpolate_recursively_from_memory([x0[0], x1[0]], 2, it))
    assert len(frames) == 5
    out = {'in_checksum': checksum(x0, x1), 'frames': np.stack(frames).astype(np.float32)}
    tmp = tempfile.mkdtemp()
    u8 = []
    for i, f in enumerate(frames):
        fn = os.path.join(tmp, f'frame_{i:03d}.png')
        ref.util.write_image(fn, f)
        from PIL import Image
        u8.append(np.asarray(Image.open(fn)))
    shutil.rmtree(tmp)
    out['frames_u8'] = np.stack(u8)
    return out


def case_recursive6(ref, prefix):
    """BASELINE configs[4] in small: the reference's own eval/util.py:interpolate_recursively_from_memory at
    times_to_interpolate = 6 (63 generated frames, depth-first) over Interpolator(align=64, block_shape=[2,2]) on a
    144x176 pair (patches 72x88 -> padded to 128x128): six generations of fed-back round-off through the reference's
    recursion / tiling / padding code.  Stored per frame: stride-4 sample + float64 row / column sums."""
    w = W.make_synthetic_weights(O.PUBLISHED, seed=0)
    x0, x1 = TI.frame_pair(1, 144, 176, seed=14, shift=(7, -9), fg_shift=(-5, 11))
    it = _interpolator(ref, w, align=64, block_shape=[2, 2])
    frames = np.stack(list(ref.util.interpolate_recursively_from_memory([x0[0], x1[0]], 6, it))).astype(np.float32)
    assert frames.shape == (65, 144, 176, 3)
    out = {'in_checksum': checksum(x0, x1)}
    put(out, 'frames', frames, False)
    return out


CASES = {'recursive6': case_recursive6, 'tiny': case_tiny, '256': case_256, 'photos': case_photos, 'vimeo': case_vimeo,
         'recursive': case_recursive, '1080p': case_1080p}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='shim', choices=['shim', 'tf'])
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    ref = Reference(args.reference, args.backend)
    prefix = 'ref' if args.backend == 'shim' else 'tf'
    for name, fn in CASES.items():
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        out = fn(ref, prefix)
        out['backend'] = np.asarray(f'{args.backend}; tensorflow {getattr(ref.tf, "__version__", "?")}')
        path = os.path.join(GOLDEN, f'{prefix}_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{name}: {time.time() - t0:.1f} s -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    main()
