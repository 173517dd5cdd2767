"""CPU tests of the oracle's op restatements (SURVEY.md 8c known-answer list).

The reference has no tests or golden vectors for these ops, so each restatement is checked
 (a) against hand-derived known answers, and
 (b) against an INDEPENDENT PyTorch built-in that implements the same published semantics
     (SURVEY.md appendix B), to guard against a misreading shared by oracle and engine.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import film_oracle as fo


def rnd(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


# ---------------------------------------------------------------- conv -----------------------
def test_conv_delta_kernel_is_identity():
    x = rnd((2, 5, 7, 3))
    w = np.zeros((3, 3, 3, 3), np.float32)
    for c in range(3):
        w[1, 1, c, c] = 1
    assert np.array_equal(fo.conv2d_same(x, w, None), x)


def test_conv2x2_same_pads_bottom_right():
    """TF 'same' with an even kernel pads 0 before / 1 after: out[y,x] = sum in[y+dy, x+dx]."""
    x = np.zeros((1, 3, 3, 1), np.float32)
    x[0, 2, 2, 0] = 1  # one-hot at the bottom-right pixel
    w = np.arange(1, 5, dtype=np.float32).reshape(2, 2, 1, 1)  # taps (0,0)=1 (0,1)=2 (1,0)=3 (1,1)=4
    y = fo.conv2d_same(x, w, None)[0, :, :, 0]
    want = np.zeros((3, 3), np.float32)
    want[2, 2] = 1  # tap (0,0) reads itself
    want[2, 1] = 2  # tap (0,1) reads x+1
    want[1, 2] = 3
    want[1, 1] = 4
    assert np.array_equal(y, want)
    assert fo.same_padding(2) == (0, 1) and fo.same_padding(3) == (1, 1) and fo.same_padding(1) == (0, 0)


@pytest.mark.parametrize('k,cin,cout', [(3, 3, 8), (3, 13, 5), (2, 6, 4), (1, 16, 2)])
def test_conv_torch_path_equals_numpy_restatement(k, cin, cout):
    x = rnd((2, 9, 11, cin), 1)
    w = rnd((k, k, cin, cout), 2)
    b = rnd((cout,), 3)
    a = fo.conv2d_same(x, w, b, 'leaky')
    r = fo.leaky_relu(fo.conv2d_same_numpy(x, w, b))
    assert np.abs(a - r).max() < 2e-5


def test_leaky_relu():
    x = np.array([-2, -0.0, 0.0, 3], np.float32)
    assert np.array_equal(fo.leaky_relu(x), np.array([-0.4, 0, 0, 3], np.float32))


# ---------------------------------------------------------------- pool -----------------------
def test_avg_pool_blocks():
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    y = fo.avg_pool2x2(x)[0, :, :, 0]
    assert np.array_equal(y, np.array([[2.5, 4.5], [10.5, 12.5]], np.float32))
    xr = rnd((2, 6, 8, 5))
    t = F.avg_pool2d(torch.from_numpy(xr).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).numpy()
    assert np.abs(fo.avg_pool2x2(xr) - t).max() < 1e-6


# ---------------------------------------------------------------- resize ---------------------
def test_resize_bilinear_x2_weights():
    """even o=2m -> 0.25*in[m-1] + 0.75*in[m] (m=0 -> in[0]); odd -> 0.75*in[m] + 0.25*in[m+1] (clamped)."""
    x = np.array([0, 4, 8, 20], np.float32).reshape(1, 1, 4, 1)
    x = np.repeat(x, 2, axis=1)  # 2 rows so that the y axis is exercised too
    y = fo.resize_bilinear(x, (4, 8))[0, 0, :, 0]
    want = np.array([0, 1, 3, 5, 7, 11, 17, 20], np.float32)
    assert np.allclose(y, want, atol=1e-6)


def test_resize_bilinear_ramp_and_constant():
    ramp = np.arange(8, dtype=np.float32).reshape(1, 1, 8, 1).repeat(4, axis=1)
    y = fo.resize_bilinear(ramp, (8, 16))[0, 3, :, 0]
    interior = y[1:-1]
    assert np.allclose(np.diff(interior), 0.5, atol=1e-6)       # a ramp stays a ramp inside
    assert y[0] == 0 and y[-1] == 7                              # and clamps at the ends
    c = np.full((1, 3, 5, 2), 1.25, np.float32)
    assert np.array_equal(fo.resize_bilinear(c, (6, 10)), np.full((1, 6, 10, 2), 1.25, np.float32))


def test_resize_bilinear_vs_torch():
    x = rnd((2, 5, 7, 2), 4)
    a = fo.resize_bilinear(np.float32(2) * x, (10, 14))
    t = F.interpolate(torch.from_numpy(2 * x).permute(0, 3, 1, 2), size=(10, 14), mode='bilinear',
                      align_corners=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(a - t).max() < 1e-5


def test_resize_nearest_is_floor_div():
    x = np.arange(12, dtype=np.float32).reshape(1, 3, 4, 1)
    y = fo.resize_nearest(x, (6, 8))
    assert np.array_equal(y, np.repeat(np.repeat(x, 2, axis=1), 2, axis=2))
    t = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), scale_factor=2, mode='nearest').permute(0, 2, 3, 1).numpy()
    assert np.array_equal(y, t)


# ---------------------------------------------------------------- warp -----------------------
def test_warp_zero_flow_is_identity_to_one_ulp():
    """With zero flow TFA's formula is NOT bitwise identity on the last row/column (floor is clamped to
    size-2, the pixel is produced as 1*(b-a)+a) - keep that form, allow 1 ulp."""
    x = rnd((1, 6, 7, 3), 5)
    y = fo.warp(x, np.zeros((1, 6, 7, 2), np.float32))
    assert np.array_equal(y[:, :-1, :-1], x[:, :-1, :-1])
    assert np.abs(y - x).max() <= 2 * np.spacing(np.abs(x).max())


def test_warp_integer_flow_is_shift_with_edge_replicate():
    x = rnd((1, 6, 8, 2), 6)
    flow = np.zeros((1, 6, 8, 2), np.float32)
    flow[..., 0] = 2    # dx: sample from x+2
    flow[..., 1] = -1   # dy: sample from y-1
    y = fo.warp(x, flow)
    ys = np.clip(np.arange(6) - 1, 0, 5)
    xs = np.clip(np.arange(8) + 2, 0, 7)
    want = x[:, ys][:, :, xs]
    assert np.abs(y - want).max() <= 1e-6


def test_warp_far_outside_gives_border_value_and_alpha_clamp():
    x = rnd((1, 4, 4, 1), 7)
    flow = np.full((1, 4, 4, 2), -100, np.float32)   # far up-left: q < 0 -> floor 0, alpha clamped to 0
    assert np.array_equal(fo.warp(x, flow), np.full_like(x, x[0, 0, 0, 0]))
    flow = np.full((1, 4, 4, 2), 100, np.float32)    # far down-right: floor = size-2, alpha clamped to 1
    y = fo.warp(x, flow)
    assert np.abs(y - x[0, 3, 3, 0]).max() <= 1e-6


def test_warp_2x2_input_and_minimum_size():
    x = np.array([[1, 2], [3, 5]], np.float32).reshape(1, 2, 2, 1)
    flow = np.full((1, 2, 2, 2), 0.5, np.float32)
    y = fo.warp(x, flow)
    assert np.isclose(y[0, 0, 0, 0], 2.75)            # centre of the four pixels
    with pytest.raises(AssertionError):
        fo.warp(np.zeros((1, 1, 4, 1), np.float32), np.zeros((1, 1, 4, 2), np.float32))


def test_warp_vs_grid_sample_border():
    x = rnd((2, 9, 12, 4), 8)
    flow = (rnd((2, 9, 12, 2), 9) * 6).astype(np.float32)      # up to ~20 px, goes out of bounds
    a = fo.warp(x, flow)
    h, w = 9, 12
    gy, gx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    qx = gx[None] + flow[..., 0]
    qy = gy[None] + flow[..., 1]
    grid = np.stack([2 * qx / (w - 1) - 1, 2 * qy / (h - 1) - 1], axis=-1).astype(np.float32)
    t = F.grid_sample(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(grid), mode='bilinear',
                      padding_mode='border', align_corners=True).permute(0, 2, 3, 1).numpy()
    assert np.abs(a - t).max() < 2e-5


# ---------------------------------------------------------------- graph pieces ---------------
def test_fusion_upsample_fold_equals_unfolded():
    """NN-upsample followed by the 2x2 'same' conv == gathering in[(y+dy)//2, (x+dx)//2] with zero beyond the
    bottom/right edge: the form the HIP conv kernel uses (ConvSeg.up)."""
    x = rnd((1, 3, 4, 5), 10)
    w = rnd((2, 2, 5, 6), 11)
    ref = fo.conv2d_same(fo.resize_nearest(x, (6, 8)), w, None)
    out = np.zeros((1, 6, 8, 6), np.float32)
    for dy in range(2):
        for dx in range(2):
            for y in range(6):
                for xx in range(8):
                    yy, xs = y + dy, xx + dx
                    if yy < 6 and xs < 8:
                        out[0, y, xx] += x[0, yy // 2, xs // 2] @ w[dy, dx]
    assert np.abs(out - ref).max() < 1e-5


def test_flow_synthesis_equals_estimator_accumulation():
    res = [rnd((1, 8 >> l, 8 >> l, 2), 20 + l) for l in range(3)]
    pyr = fo.flow_pyramid_synthesis(res)
    v = res[-1]
    for l in (1, 0):
        v = fo.resize_bilinear(np.float32(2) * v, res[l].shape[1:3])
        v = res[l] + v
        assert np.array_equal(v, pyr[l])


def test_patches_roundtrip_and_order():
    img = np.arange(1 * 4 * 6 * 2, dtype=np.float32).reshape(1, 4, 6, 2)
    p = fo.image_to_patches(img, [2, 3])
    assert p.shape == (6, 2, 2, 2)
    assert np.array_equal(p[1], img[0, 0:2, 2:4])      # row-major blocks
    assert np.array_equal(p[3], img[0, 2:4, 0:2])
    assert np.array_equal(fo.patches_to_image(p, [2, 3]), img)


@pytest.mark.parametrize('size,off', [(1080, 4), (540, 18)])
def test_pad_crop_roundtrip(size, off):
    x = rnd((1, size, 16, 3), 12)
    p, box = fo.pad_to_align(x, 64)
    assert p.shape[1] % 64 == 0 and box['offset_height'] == off
    assert np.array_equal(p[:, off:off + size, box['offset_width']:box['offset_width'] + 16], x)
    assert not p[:, :off].any() and not p[:, off + size:].any()


def test_lean_forward_of_the_big_golden_script_is_the_oracle_forward():
    """tools/make_big_golden.py assembles the aligned pyramid level by level to fit a 3840x2240 frame into host memory; on
    a small frame it must give the bits of oracle.film_oracle.film_forward."""
    import importlib.util
    import os
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    spec = importlib.util.spec_from_file_location('make_big_golden', os.path.join(root, 'tools', 'make_big_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    from oracle import film_oracle as fo
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    rng = np.random.default_rng(5)
    x0 = rng.random((1, 64, 64, 3), dtype=np.float32)
    x1 = rng.random((1, 64, 64, 3), dtype=np.float32)
    assert np.array_equal(mod.film_forward_lean(x0, x1, w, fo.Options()), fo.film_forward(x0, x1, w, fo.Options()))
