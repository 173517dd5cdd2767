"""Weight tensors of film_net: canonical names, shapes, synthetic generation and file IO.

Layout is the reference's: TF/Keras ``kernel`` in HWIO ``[kh, kw, cin, cout]`` float32 and
``bias`` ``[cout]`` (what ``model.save()`` writes, training/train_lib.py:280,
training/build_saved_model_cli.py:65-73).  Canonical names follow the Keras layer names
in the reference source:

  feat_net/sub_extractor/cfeat_conv_{0..2*sub_levels-1}   feature_extractor.py:117-123
  predict_flow/flow_predictor_{i}/conv_{j}                pyramid_flow_estimator.py:74-83,112-117
  predict_flow/flow_predictor_shared/conv_{j}             pyramid_flow_estimator.py:118-123
  fusion/convs_{i}_{0,1,2}                                fusion.py:76-97  (layers are unnamed upstream)
  fusion/output_conv                                      fusion.py:100-101

Because the pretrained SavedModels (README.md:69-83, Google Drive) cannot be fetched in
this environment, tests and benchmarks use *seeded synthetic* weights of exactly these
shapes; the numerics of the engine do not depend on the values.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

from .options import Options, PUBLISHED

WEIGHTS_FILE = 'film_weights.npz'


def feature_channels(opt: Options) -> List[int]:
    """Channels of the cascaded feature pyramid per level (feature_extractor.py:186-193)."""
    out = []
    for lvl in range(opt.pyramid_levels):
        c = 0
        for j in range(opt.sub_levels):
            if j <= lvl:
                c += opt.filters << j
        out.append(c)
    return out


def fusion_filters(opt: Options) -> List[int]:
    """fusion.py:75-79."""
    m, k = opt.specialized_levels, opt.filters
    return [(k << i) if i < m else (k << m) for i in range(opt.fusion_pyramid_levels - 1)]


def weight_specs(opt: Options = PUBLISHED) -> List[Tuple[str, Tuple[int, int, int, int], str]]:
    """[(layer name, HWIO kernel shape, activation 'leaky'|'linear')] in a fixed order."""
    specs = []
    k = opt.filters
    cin = 3
    for i in range(opt.sub_levels):
        specs.append((f'feat_net/sub_extractor/cfeat_conv_{2 * i}', (3, 3, cin, k << i), 'leaky'))
        specs.append((f'feat_net/sub_extractor/cfeat_conv_{2 * i + 1}', (3, 3, k << i, k << i), 'leaky'))
        cin = k << i
    fc = feature_channels(opt)
    for p in range(opt.specialized_levels + 1):
        shared = p == opt.specialized_levels
        prefix = 'predict_flow/flow_predictor_shared' if shared else f'predict_flow/flow_predictor_{p}'
        # the shared predictor first sees level `specialized_levels`, whose channel count
        # equals that of every coarser level when sub_levels <= specialized_levels+1.
        c_in = 2 * fc[min(p, opt.pyramid_levels - 1)]
        nf, nconv = opt.flow_filters[p], opt.flow_convs[p]
        for j in range(nconv):
            specs.append((f'{prefix}/conv_{j}', (3, 3, c_in, nf), 'leaky'))
            c_in = nf
        specs.append((f'{prefix}/conv_{nconv}', (1, 1, nf, nf // 2), 'leaky'))
        specs.append((f'{prefix}/conv_{nconv + 1}', (1, 1, nf // 2, 2), 'linear'))
    ff = fusion_filters(opt)
    L = opt.fusion_pyramid_levels
    for i in range(L - 1):
        aligned_c = 2 * (3 + fc[i]) + 4
        net_c = (2 * (3 + fc[L - 1]) + 4) if i == L - 2 else ff[i + 1]
        specs.append((f'fusion/convs_{i}_0', (2, 2, net_c, ff[i]), 'linear'))
        specs.append((f'fusion/convs_{i}_1', (3, 3, aligned_c + ff[i], ff[i]), 'leaky'))
        specs.append((f'fusion/convs_{i}_2', (3, 3, ff[i], ff[i]), 'leaky'))
    specs.append(('fusion/output_conv', (1, 1, ff[0], 3), 'linear'))
    return specs


def check_shared_predictor(opt: Options) -> None:
    fc = feature_channels(opt)
    s = opt.specialized_levels
    if any(c != fc[s] for c in fc[s:]):
        raise ValueError('levels sharing the flow predictor must have equal feature channels '
                         '(needs sub_levels <= specialized_levels + 1)')


def num_params(opt: Options = PUBLISHED) -> int:
    return sum(int(np.prod(s)) + s[3] for _, s, _ in weight_specs(opt))


def make_synthetic_weights(opt: Options = PUBLISHED, seed: int = 0,
                           dtype=np.float32) -> Dict[str, np.ndarray]:
    """Seeded weights with activations kept O(1) through the net and flows of a few px.

    He-style scaling for the leaky(0.2) layers (second moment preserved:
    var = 1/(0.52*fan_in)); flow heads scaled so that residual flows are ~0.5 px at the
    specialised levels and ~0.05 px per shared level (they are doubled at every finer
    level, pyramid_flow_estimator.py:155), which exercises out-of-image sampling in the
    warps; biases are small but non-zero so the bias path is tested."""
    check_shared_predictor(opt)
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape, act in weight_specs(opt):
        kh, kw, cin, cout = shape
        fan_in = kh * kw * cin
        if act == 'leaky':
            std = np.sqrt(1.0 / (0.52 * fan_in))
        else:
            std = np.sqrt(1.0 / fan_in)
        bias_std = 0.05
        bias_mean = 0.0
        if name.startswith('predict_flow') and act == 'linear':
            std *= 0.05 if 'shared' in name else 0.5
            bias_std = 0.01 if 'shared' in name else 0.1
        if name == 'fusion/output_conv':
            std *= 0.25
            bias_mean = 0.5
        w = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(bias_std) + np.float32(bias_mean)
        out[name + '/kernel'] = w.astype(dtype)
        out[name + '/bias'] = b.astype(dtype)
    return out


def validate_weights(weights: Dict[str, np.ndarray], opt: Options = PUBLISHED) -> None:
    for name, shape, _ in weight_specs(opt):
        k = weights.get(name + '/kernel')
        b = weights.get(name + '/bias')
        if k is None or b is None:
            raise KeyError(f'missing weight {name}')
        if tuple(k.shape) != tuple(shape) or tuple(b.shape) != (shape[3],):
            raise ValueError(f'{name}: expected kernel {shape}, got {tuple(k.shape)} / bias {tuple(b.shape)}')


def save_weights(model_path: str, weights: Dict[str, np.ndarray]) -> str:
    """Writes ``<model_path>/film_weights.npz`` (keys = canonical names)."""
    os.makedirs(model_path, exist_ok=True)
    fn = os.path.join(model_path, WEIGHTS_FILE)
    np.savez(fn, **{k.replace('/', '|'): np.asarray(v, dtype=np.float32) for k, v in weights.items()})
    return fn


def is_saved_model(model_path: str) -> bool:
    """True when `model_path` holds a TF2 SavedModel variables bundle and no film_weights.npz (which load_weights prefers)."""
    if not model_path or model_path.endswith('.npz') or os.path.isfile(os.path.join(model_path, WEIGHTS_FILE)):
        return False
    return os.path.isfile(os.path.join(model_path, 'variables', 'variables.index'))


def load_weights(model_path: str, opt: Options = PUBLISHED) -> Dict[str, np.ndarray]:
    """Loads the weight set of a model directory.

    ``model_path`` plays the role of the SavedModel directory passed to
    ``Interpolator(model_path)`` (eval/interpolator.py:135-148).  Accepted contents:
      * ``film_weights.npz`` written by :func:`save_weights` (canonical names), or
      * a TF2 SavedModel (``variables/variables.index`` + data shards), parsed by the
        TF-free reader in :mod:`film_hip.tf_bundle`.
    """
    fn = model_path if model_path.endswith('.npz') else os.path.join(model_path, WEIGHTS_FILE)
    if os.path.isfile(fn):
        with np.load(fn) as z:
            return {k.replace('|', '/'): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
    idx = os.path.join(model_path, 'variables', 'variables.index')
    if os.path.isfile(idx):
        from . import tf_bundle
        return tf_bundle.load_film_weights(os.path.join(model_path, 'variables', 'variables'), opt)
    raise FileNotFoundError(
        f'{model_path}: neither {WEIGHTS_FILE} nor a SavedModel variables bundle found')


def main(argv=None) -> int:
    """``python -m film_hip.weights <model dir | film_weights.npz> --export blob.bin``: the parameter set as the flat float32
    blob of ``film_export_packed`` (per layer the HWIO kernel, then the bias) - what a host hands to ``film_import_packed``
    (e.g. the ranks of a multi-GPU job).  A C / C++ host can also read the SavedModel directly: ``film_load_bundle``
    (INTEGRATION.md 2)."""
    import argparse
    ap = argparse.ArgumentParser(prog='python -m film_hip.weights')
    ap.add_argument('model_path')
    ap.add_argument('--export', required=True, help='output file: raw little-endian float32')
    args = ap.parse_args(argv)
    from .engine import FilmEngine
    eng = FilmEngine(PUBLISHED, device=-1)      # plan-only handle: no GPU needed
    eng.set_weights(load_weights(args.model_path))
    blob = eng.export_packed()
    blob.astype('<f4').tofile(args.export)
    print(f'{args.export}: {blob.size} floats ({blob.nbytes / 1e6:.1f} MB)')
    eng.close()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
