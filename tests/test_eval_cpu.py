"""Benchmark-eval twin (eval/eval_cli.py, eval/metrics.py): metric known answers, an independent SSIM cross-check,
the results.csv layout of the reference loop - driven by a stand-in interpolator (the engine needs a GPU)."""
import os

import numpy as np
import pytest


def test_metric_known_answers():
    from eval import metrics as M
    rng = np.random.default_rng(0)
    a = rng.random((1, 32, 40, 3), dtype=np.float32)
    assert M.l1(a, a) == 0 and M.l2(a, a) == 0
    assert M.ssim(a, a) == pytest.approx(1.0, abs=1e-12)
    assert np.isinf(M.psnr(a, a))
    b = np.clip(a + 0.1, 0, 2).astype(np.float32)
    assert M.l1(b, a) == pytest.approx(0.1, rel=1e-5)
    assert M.l2(b, a) == pytest.approx(0.01, rel=1e-4)
    assert M.psnr(b, a) == pytest.approx(20.0, abs=1e-3)          # 10*log10(1/0.01)
    # batch mean of per-image PSNR, not PSNR of the batch MSE
    two = np.concatenate([a, a]); off = np.concatenate([a + 0.1, a + 0.01]).astype(np.float32)
    assert M.psnr(off, two) == pytest.approx((20.0 + 40.0) / 2, abs=1e-2)
    with pytest.raises(ValueError):
        M.ssim(a[:, :8], a[:, :8])
    with pytest.raises(ValueError):
        M.test_losses(['vgg'])


def test_ssim_against_independent_convolution():
    """Same definition through scipy.signal.correlate2d with the explicit 2-D window."""
    from scipy.signal import correlate2d
    from eval import metrics as M
    rng = np.random.default_rng(1)
    a = rng.random((20, 24, 2)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.05, a.shape), 0, 1).astype(np.float32)
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2)); g /= g.sum()
    w2 = np.outer(g, g)
    vals = []
    for c in range(2):
        x, y = a[..., c].astype(np.float64), b[..., c].astype(np.float64)
        f = lambda z: correlate2d(z, w2, mode='valid')
        mx, my = f(x), f(y)
        lum = (2 * mx * my + 1e-4) / (mx * mx + my * my + 1e-4)
        cs = (2 * f(x * y) - 2 * mx * my + 9e-4) / (f(x * x + y * y) - mx * mx - my * my + 9e-4)
        vals.append(np.mean(lum * cs))
    assert M.ssim(a, b) == pytest.approx(np.mean(vals), rel=1e-10)


def test_eval_loop_outputs(tmp_path):
    from eval import eval_cli, util
    rng = np.random.default_rng(2)
    root = tmp_path / 'data'
    for seq in ('00001/0001', '00001/0002', '00002/0001'):
        d = root / seq
        os.makedirs(d)
        for i in (1, 2, 3):
            util.write_image(str(d / f'im{i}.png'), rng.random((16, 20, 3), dtype=np.float32))
    os.makedirs(root / 'junk')
    util.write_image(str(root / 'junk' / 'only.png'), rng.random((16, 20, 3), dtype=np.float32))
    trip = eval_cli.find_triplets(str(root))
    assert [k for k, _ in trip] == ['00001_0001', '00001_0002', '00002_0001']
    assert [os.path.basename(f) for f in trip[0][1]] == ['im1.png', 'im2.png', 'im3.png']

    class Blend:   # stand-in: average of the inputs, slightly out of range to exercise the clip
        def __call__(self, x0, x1, dt):
            return 0.5 * (x0 + x1) + 0.6
    out = tmp_path / 'out'
    totals = eval_cli.run_evaluation(Blend(), trip, str(out), max_examples=2, metrics=['l1', 'psnr'], output_frames=True,
                                     model_path='m', source=str(root))
    lines = open(out / 'results.csv').read().strip().split('\n')
    assert lines[0] == 'key, l1, psnr'
    assert [l.split(',')[0] for l in lines[1:]] == ['00001_0001', '00001_0002', 'mean']
    assert set(totals) == {'l1', 'psnr'} and totals['l1'] > 0.1      # clipped at 1.0: far from the ground truth
    assert os.path.isfile(out / 'readme.txt') and os.path.isfile(out / '00001_0001_image.png')
    row = [float(v) for v in lines[1].split(',')[1:]]
    mean = [float(v) for v in lines[3].split(',')[1:]]
    row2 = [float(v) for v in lines[2].split(',')[1:]]
    assert mean[0] == pytest.approx((row[0] + row2[0]) / 2)
