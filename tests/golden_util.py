"""Readers / comparators of the committed golden vectors under tests/golden/.

ref_*.npz : produced by tools/make_ref_golden.py --backend shim = the REFERENCE'S OWN graph / wrapper /
            recursion code executed over oracle/tf_shim (PyTorch built-ins standing in for TF ops).
tf_*.npz  : the same cases from a real TensorFlow install (--backend tf); preferred when present.
Large tensors are stored as a stride-4 pixel sample + float64 row / column sums of the full tensor."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
STRIDE = 4


def load(case):
    """-> (npz, provenance) ; provenance is 'tf' when a real-TensorFlow vector exists, else 'ref-graph'."""
    tf_path = os.path.join(GOLDEN, f'tf_{case}.npz')
    if os.path.isfile(tf_path):
        return np.load(tf_path), 'tf'
    return np.load(os.path.join(GOLDEN, f'ref_{case}.npz')), 'ref-graph'


def pinned_by_tensorflow(case):
    return os.path.isfile(os.path.join(GOLDEN, f'tf_{case}.npz'))


def check_inputs(g, *arrays):
    want = g['in_checksum']
    got = np.asarray([float(np.asarray(a, np.float64).sum()) for a in arrays])
    assert np.allclose(got, want, rtol=0, atol=1e-6), \
        f'test inputs differ from the ones the golden was made with: {got} vs {want}'


def diff(g, key, full):
    """max |full - golden| over what the golden holds for `key`: the whole tensor if stored in full, else the
    stride-4 sample, plus the mean-per-element error of the row / column sums (covers every pixel)."""
    full = np.asarray(full)
    if key in g.files:
        return float(np.abs(full - g[key]).max())
    s4 = g[f'{key}.s4']
    assert tuple(g[f'{key}.shape']) == full.shape, (key, tuple(g[f'{key}.shape']), full.shape)
    d = float(np.abs(full[:, ::STRIDE, ::STRIDE, :] - s4).max())
    rows = np.abs(full.astype(np.float64).sum(axis=2) - g[f'{key}.rowsum']).max() / full.shape[2]
    cols = np.abs(full.astype(np.float64).sum(axis=1) - g[f'{key}.colsum']).max() / full.shape[1]
    return max(d, float(rows), float(cols))
