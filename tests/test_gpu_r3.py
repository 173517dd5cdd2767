"""Round-3 GPU tests: BASELINE configs[4] (4K, 4x4 tiles, times_to_interpolate 6) recursion depth, the tile-sharded
single-pair driver, the N-rank start-up helper on RCCL, a second weight set on a used engine, the SavedModel
directory path of Interpolator, and parity where F(4,3) round-off and the warp clamps bite: trained-net-like dynamic range
and flows that leave the frame at every level, each against the float32 AND the float64 oracle."""
import os
import socket

import numpy as np
import pytest

import inputs as TI
from conftest import prefers_extra_families
from test_gpu_parity import _engine

pytestmark = pytest.mark.gpu

IMAGE_TOL = 1e-3          # north_star: |delta| < 1e-3 fp32 per pixel
HALF = np.full((1,), 0.5, np.float32)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def published():
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    eng = _engine(PUBLISHED, w)
    yield PUBLISHED, w, eng
    eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: depth-6 recursion
# ---------------------------------------------------------------------------------------------------------------------
def test_depth6_recursion_of_one_4k_tile_against_the_oracle_tree(published):
    """One tile's WHOLE times_to_interpolate = 6 recursion tree of the 4K 4x4-tiled pair (63 tile-forwards of 960x540 padded
    to 960x576; eval/util.py:62-91 over eval/interpolator.py:192-206) on the default fp32 plan, against the committed
    oracle fixture (tools/make_t6_golden.py: stride-12 sample + float64 row / column sums of every generated frame).
    Round-off is fed back six times: max|delta| is printed per generation and the sixth must stay < 1e-3."""
    import torch
    from film_hip.sharding import extract_tiles, recurse_tiles
    from film_hip.torch_io import DeviceInterpolator
    path = os.path.join(GOLDEN, 'oracle_t6_tile.npz')
    if not os.path.isfile(path):
        pytest.skip('tests/golden/oracle_t6_tile.npz not generated yet (python tools/make_t6_golden.py)')
    g = np.load(path)
    opt, w, eng = published
    x0, x1 = TI.frame_pair(1, 2160, 3840, seed=4)
    block, tile, stride = [int(v) for v in g['block']], int(g['tile']), int(g['stride'])
    dev = torch.device('cuda', 0)
    a = extract_tiles(torch.from_numpy(x0[0]), block, [tile]).to(dev)
    b = extract_tiles(torch.from_numpy(x1[0]), block, [tile]).to(dev)
    assert np.allclose([float(a.double().sum()), float(b.double().sum())], g['in_checksum'], rtol=0, atol=1e-6)
    T = int(np.log2(len(g['frame_index']) + 1))
    seq = recurse_tiles(a, b, T, DeviceInterpolator(eng, align=64).batch)[:, 0].cpu().numpy()     # [2^T + 1, 540, 960, 3]
    assert seq.shape[0] == 2 ** T + 1 and np.isfinite(seq).all()
    per_depth = {}
    for i, (k, d) in enumerate(zip(g['frame_index'], g['depth'])):
        f = seq[int(k)]
        ds = float(np.abs(f[::stride, ::stride] - g['sample'][i]).max())
        dr = float(np.abs(f.astype(np.float64).sum(axis=1) - g['rowsum'][i]).max() / f.shape[1])
        dc = float(np.abs(f.astype(np.float64).sum(axis=0) - g['colsum'][i]).max() / f.shape[0])
        per_depth[int(d)] = max(per_depth.get(int(d), 0.0), ds, dr, dc)
    line = ', '.join(f'gen {d}: {per_depth[d]:.2e}' for d in sorted(per_depth))
    growth = [per_depth[d + 1] / max(per_depth[d], 1e-12) for d in sorted(per_depth)[:-1]]
    print(f'T = {T} recursion of tile {tile}: hip vs oracle max|d| per generation: {line}; growth per generation '
          f'{[round(x, 2) for x in growth]}')
    assert per_depth[T] < IMAGE_TOL and max(per_depth.values()) < IMAGE_TOL


def test_4k_4x4_chunks_replay_one_plan(published):
    """A 4K pair with 4x4 tiles = 16 tiles of 960x576 against a 15-tile invocation limit: the chunks are 8 + 8 (one cached
    plan), not 15 + 1, and the frame equals the one computed with at most 4 tiles per invocation bit for bit."""
    import torch
    from film_hip.torch_io import DeviceInterpolator
    opt, w, eng = published
    x0, x1 = TI.frame_pair(1, 2160, 3840, seed=4)
    dev = torch.device('cuda', 0)
    a, b = torch.from_numpy(x0).to(dev), torch.from_numpy(x1).to(dev)
    it = DeviceInterpolator(eng, align=64, block_shape=[4, 4])
    full = it(a, b)
    torch.cuda.synchronize()
    eng.set_option('profile', 1)
    it(a, b)
    torch.cuda.synchronize()
    prof = eng.profile()
    eng.set_option('profile', 0)
    assert prof['B'] in (8, 16), prof['B']           # never a 1-tile remainder plan
    eng.set_option('max_batch', 4)
    small = it(a, b)
    torch.cuda.synchronize()
    eng.set_option('max_batch', 0)
    assert torch.equal(full, small) and bool(torch.isfinite(full).all())


# ---------------------------------------------------------------------------------------------------------------------
# tile-sharded single pair + the N-rank start-up on RCCL (world size 1 is what one GPU allows)
# ---------------------------------------------------------------------------------------------------------------------
def test_tile_sharded_driver_is_bit_identical_to_the_tiled_path(published):
    """TileShardedRecursion on one rank (tiles run through the UNTILED entry point as a batch of small frames, stitched at
    the end) vs the tiled breadth-first driver (film_interpolate with block_shape): same frames, same bits, T = 2."""
    import torch
    from film_hip.recursive import interpolate_pair_recursively
    from film_hip.sharding import TileShardedRecursion
    from film_hip.torch_io import DeviceInterpolator
    opt, w, eng = published
    x0, x1 = TI.frame_pair(1, 200, 336, seed=12)
    dev = torch.device('cuda', 0)
    a, b = torch.from_numpy(x0[0]).to(dev), torch.from_numpy(x1[0]).to(dev)
    want = interpolate_pair_recursively(a, b, 2, DeviceInterpolator(eng, align=64, block_shape=[2, 2]))
    got = TileShardedRecursion(DeviceInterpolator(eng, align=64).batch, [2, 2], None).run(a, b, 2)
    assert got.shape == want.shape == (5, 200, 336, 3)
    assert torch.equal(got, want)


def test_sharded_interpolator_and_tile_gather_over_rccl(published, tmp_path):
    """film_hip.sharding.sharded_interpolator (rank 0 loads the model directory, status + weight broadcast over the nccl =
    RCCL backend) and the tile gather of TileShardedRecursion through the same process group, on the one GPU there is
    (world_size 1, collective forced): result bit-identical to the plain tiled Interpolator of the module's engine."""
    import torch
    import torch.distributed as dist
    from film_hip import weights as W
    from film_hip.sharding import TileShardedRecursion, sharded_interpolator
    from film_hip.torch_io import DeviceInterpolator
    opt, w, eng = published
    W.save_weights(str(tmp_path / 'model'), w)
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        it = sharded_interpolator(str(tmp_path / 'model'), 64, [2, 2], dist, 0)
        assert it.align == 64 and it.block_shape == [2, 2]
        x0, x1 = TI.frame_pair(1, 200, 336, seed=12)
        want = DeviceInterpolator(eng, align=64, block_shape=[2, 2])(torch.from_numpy(x0).to(dev), torch.from_numpy(x1).to(dev))
        assert np.array_equal(it(x0, x1, HALF), want.cpu().numpy())
        drv = TileShardedRecursion(DeviceInterpolator(it.engine, align=64).batch, [2, 2], dist, collective_at_world1=True)
        assert drv.dist is not None and drv.tiles == [0, 1, 2, 3]
        seq = drv.run(torch.from_numpy(x0[0]).to(dev), torch.from_numpy(x1[0]).to(dev), 1)
        assert seq.shape == (3, 200, 336, 3) and torch.equal(seq[1], want[0])
        with pytest.raises(Exception):        # a model directory rank 0 cannot load must raise, not hang the other ranks
            sharded_interpolator(str(tmp_path / 'missing'), 64, [2, 2], dist, 0)
        it.engine.close()
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# weights: second set on a used engine, SavedModel directory
# ---------------------------------------------------------------------------------------------------------------------
@prefers_extra_families
def test_second_weight_set_on_a_used_engine(tiny_weights):
    """forward -> set_weights(other set) -> forward of the SAME shape on the SAME handle (cached plan, layout groups the
    plan pulled in on demand) == a fresh engine that only ever saw the second set (round-2 ADVICE: the re-finalize used
    to repack group 0 only)."""
    from film_hip import weights as W
    from film_hip.options import TINY
    w2 = W.make_synthetic_weights(TINY, seed=11)
    x0, x1 = TI.frame_pair(1, 128, 96, seed=3)
    eng = _engine(TINY, tiny_weights)
    from conftest import has_extra_families
    # plans on the halo (general kernel in the default build) / F(2,3) (FILM_EXTRA_FAMILIES builds only) / default families
    for key, val in (('winograd', 0),) + ((('winograd', 2),) if has_extra_families() else ()) + (('winograd', 1),):
        eng.set_option(key, val)
        first = eng.forward(x0, x1)
        eng.set_weights(w2)
        second = eng.forward(x0, x1)
        fresh = _engine(TINY, w2)
        fresh.set_option(key, val)
        want = fresh.forward(x0, x1)
        fresh.close()
        assert np.array_equal(second, want) and not np.array_equal(first, second), (key, val)
        eng.set_weights(tiny_weights)
    eng.close()


def test_interpolator_from_a_savedmodel_directory(published, tmp_path):
    """Interpolator(model_path=<SavedModel dir>) - variables bundle written with the object-graph keys of a Keras
    model.save() (tests/bundle_writer.save_film_bundle), read back by the NATIVE reader behind the C-ABI (film_load_bundle) - gives the bits of the engine
    that took the same tensors as a dict."""
    import bundle_writer
    from eval.interpolator import Interpolator
    opt, w, eng = published
    bundle_writer.save_film_bundle(str(tmp_path / 'saved_model'), w, opt)
    assert os.path.isfile(tmp_path / 'saved_model' / 'variables' / 'variables.index')
    it = Interpolator(str(tmp_path / 'saved_model'), align=64)
    x0, x1 = TI.frame_pair(1, 120, 200, seed=6)
    got = it(x0, x1, HALF)
    want = eng.interpolate_frames(x0, x1, align=64)
    assert np.array_equal(got, want)
    it.engine.close()


# ---------------------------------------------------------------------------------------------------------------------
# parity hardening: dynamic range and large motion, default plan, vs float32 AND float64 oracle
# ---------------------------------------------------------------------------------------------------------------------
def _f64(weights):
    return {k: v.astype(np.float64) for k, v in weights.items()}


def _rescaled_weights(w, opt, s):
    """Trained-net-like dynamic range: the sub-extractor's first layer (kernel and bias) is scaled by s - leaky_relu is
    positively homogeneous, so every feature map grows by about s - and the layers that CONSUME features (flow-predictor
    conv_0, the feature rows of fusion/convs_i_1 and of the coarsest 2x2 layer) are scaled by 1 / s, so flows and the image
    stay O(1) while every convolution of the extractor and every warp works on activations of size s."""
    from film_hip import weights as W
    out = {k: v.copy() for k, v in w.items()}
    out['feat_net/sub_extractor/cfeat_conv_0/kernel'] *= np.float32(s)
    out['feat_net/sub_extractor/cfeat_conv_0/bias'] *= np.float32(s)
    for k in out:
        if k.startswith('predict_flow') and k.endswith('conv_0/kernel'):
            out[k] /= np.float32(s)
    fc = W.feature_channels(opt)
    L = opt.fusion_pyramid_levels
    for i in range(L - 1):
        k1 = out[f'fusion/convs_{i}_1/kernel']                 # input order [img0 3 | feat0 C | img1 3 | feat1 C | flows 4 | net]
        c = fc[i]
        k1[:, :, 3:3 + c] /= np.float32(s)
        k1[:, :, 6 + c:6 + 2 * c] /= np.float32(s)
    c = fc[L - 1]
    k0 = out[f'fusion/convs_{L - 2}_0/kernel']
    k0[:, :, 3:3 + c] /= np.float32(s)
    k0[:, :, 6 + c:6 + 2 * c] /= np.float32(s)
    return out


@pytest.mark.parametrize('scale', [100.0, 1000.0])
def test_parity_with_trained_net_like_dynamic_range(scale):
    """Feature activations of 10^2 / 10^3 (F(4,3)'s round-off is relative to the activation magnitude): default plan on a
    256x320 pair vs the float32 and the float64 oracle."""
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    from oracle import film_oracle as fo
    w = _rescaled_weights(W.make_synthetic_weights(PUBLISHED, seed=0), PUBLISHED, scale)
    x0, x1 = TI.frame_pair(1, 256, 320, seed=9)
    eng = _engine(PUBLISHED, w)
    got = eng.forward(x0, x1)
    feat_max = float(np.abs(eng.tap('feat0')).max())
    eng.close()
    o32, aux = fo.film_forward(x0, x1, w, fo.Options(), return_aux=True)
    o64 = fo.film_forward(x0.astype(np.float64), x1.astype(np.float64), _f64(w), fo.Options())
    e_hip32, e_hip64, e_or = (float(np.abs(got - o32).max()), float(np.abs(got - o64).max()), float(np.abs(o32 - o64).max()))
    flow0 = float(np.abs(aux['forward_flow_pyramid'][0]).max())
    print(f'scale {scale:g}: |feat0| max {feat_max:.1f}, |flow0| max {flow0:.1f} px, |image| max {np.abs(o64).max():.2f}; '
          f'hip vs f32 oracle {e_hip32:.2e}, hip vs f64 oracle {e_hip64:.2e}, f32 oracle vs f64 {e_or:.2e} '
          f'(ratio {e_hip64 / max(e_or, 1e-12):.1f})')
    assert feat_max > 0.5 * scale
    assert e_hip64 < IMAGE_TOL and e_hip32 < IMAGE_TOL
    assert e_hip64 <= max(8 * e_or, 2e-5), 'the HIP plan loses much more to round-off than the fp32 oracle itself'


def test_parity_with_flows_that_leave_the_frame_at_every_level():
    """Flow heads scaled until the level-0 flows exceed +-32 px (FILM exists for large motion): every warped level samples
    outside the frame (the clamps of util.py:48-82 / dense_image_warp) and a flow error doubles per level
    (pyramid_flow_estimator.py:151-161).  Default plan vs the float32 and the float64 oracle."""
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    from oracle import film_oracle as fo
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    for k in list(w):
        if k.startswith('predict_flow') and ('conv_4/' in k):
            w[k] = w[k] * np.float32(8.0)
    x0, x1 = TI.frame_pair(1, 256, 320, seed=10, shift=(15, -21), fg_shift=(-12, 25))
    eng = _engine(PUBLISHED, w)
    got = eng.forward(x0, x1)
    v0 = eng.tap('v0')
    eng.close()
    o32, aux = fo.film_forward(x0, x1, w, fo.Options(), return_aux=True)
    o64, aux64 = fo.film_forward(x0.astype(np.float64), x1.astype(np.float64), _f64(w), fo.Options(), return_aux=True)
    outside = []
    for l, f in enumerate(aux64['forward_flow_pyramid']):
        h, wd = f.shape[1:3]
        yy, xx = np.mgrid[0:h, 0:wd]
        qx, qy = xx + 0.5 * f[0, ..., 0], yy + 0.5 * f[0, ..., 1]
        outside.append(float(((qx < 0) | (qx > wd - 1) | (qy < 0) | (qy > h - 1)).mean()))
    fmax = float(np.abs(aux64['forward_flow_pyramid'][0]).max())
    e_flow_hip = float(np.abs(v0[:1] - aux64['forward_flow_pyramid'][0]).max())
    e_flow_or = float(np.abs(aux['forward_flow_pyramid'][0] - aux64['forward_flow_pyramid'][0]).max())
    e_hip32, e_hip64, e_or = (float(np.abs(got - o32).max()), float(np.abs(got - o64).max()), float(np.abs(o32 - o64).max()))
    print(f'large motion: |flow0| max {fmax:.1f} px, share of t=0.5 samples outside the frame per level {[round(o, 3) for o in outside]}; '
          f'flow0 error hip {e_flow_hip:.2e} px / f32 oracle {e_flow_or:.2e} px vs f64; image: hip vs f32 oracle {e_hip32:.2e}, '
          f'hip vs f64 {e_hip64:.2e}, f32 oracle vs f64 {e_or:.2e} (ratio {e_hip64 / max(e_or, 1e-12):.1f})')
    assert fmax > 32.0 and min(outside) > 0.0
    assert e_hip64 < IMAGE_TOL and e_hip32 < IMAGE_TOL
    assert e_hip64 <= max(8 * e_or, 2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# conv_wino43_kernel: every tile shape (incl. the round-3 32-pixel x 8-row "Q8" tiles) gives the same bits
# ---------------------------------------------------------------------------------------------------------------------
@prefers_extra_families      # (the default library holds seven of the seventeen tiles, the extra flavour all of them)
@pytest.mark.parametrize('b,h,w', [(1, 128, 320), (2, 192, 256), (1, 64, 960)])
def test_every_f43_tile_shape_gives_the_same_bits(published, b, h, w):
    """F(4,3) forced onto every eligible 3x3 layer ("winograd" = 3), then every Wino43Tile shape forced in turn ("w43_shape"):
    image, the feature pyramid (fused pools) and the aligned pyramid are bit-identical to the autotuned plan - the sums are
    the same k-ordered chains whatever the tile.  Covers the Q8 tiles on widths that fill / do not fill 32- and 64-pixel
    patches, the fused average pool of the sub-extractor stages and the fused RGB head."""
    from film_hip.engine import FilmEngine
    opt, wts, _ = published
    eng = FilmEngine(opt, device=0)
    eng.set_weights(wts)
    eng.set_option('winograd', 3)
    x0, x1 = TI.frame_pair(b, h, w, seed=31)
    ref = eng.forward(x0, x1)
    taps = {k: eng.tap(k) for k in ('feat0', 'feat2', 'aligned0', 'aligned1')}
    used = set()
    for shape in range(17):
        eng.set_option('w43_shape', shape)
        tiles = {((o['tile'] & 15) + (16 if o['tile'] & 4096 else 0)) for o in eng.plan(b, h, w)['ops']
                 if o['kind'] == 'conv_mfma' and (o['tile'] & 2048)}
        if shape not in tiles:
            continue          # no layer of this plan can run the shape
        used.add(shape)
        got = eng.forward(x0, x1)
        assert np.array_equal(got, ref), (shape, float(np.abs(got - ref).max()))
        for k, v in taps.items():
            assert np.array_equal(eng.tap(k), v), (shape, k)
    print('F(4,3) tile shapes exercised:', sorted(used))
    assert {12, 13, 14, 16} <= used        # the Q8 tiles ran
    eng.close()


def test_depth6_recursion_against_the_reference_recursion_code(published):
    """The frames of the reference's OWN interpolate_recursively_from_memory at times_to_interpolate = 6 over its 2x2-tiled
    Interpolator (tests/golden/ref_recursive6.npz, tools/make_ref_golden.py case_recursive6: 144x176 pair, patches padded
    to 128x128) vs the device breadth-first driver behind eval/util.py on the HIP engine: all 63 generated frames, max|delta|
    per generation printed."""
    import golden_util as G
    from eval import util
    from eval.interpolator import Interpolator
    opt, w, eng = published
    g, prov = G.load('recursive6')
    x0, x1 = TI.frame_pair(1, 144, 176, seed=14, shift=(7, -9), fg_shift=(-5, 11))
    G.check_inputs(g, x0, x1)
    it = Interpolator('', align=64, block_shape=[2, 2], engine=eng)
    frames = np.stack(list(util.interpolate_recursively_from_memory([x0[0], x1[0]], 6, it)))
    assert frames.shape == (65, 144, 176, 3)
    per = {}
    for k in range(1, 64):
        gen = 6 - ((k & -k).bit_length() - 1)             # frame k = m * 2^(6 - gen), m odd
        d = float(np.abs(frames[k, ::G.STRIDE, ::G.STRIDE] - g['frames.s4'][k]).max())
        rows = float(np.abs(frames[k].astype(np.float64).sum(axis=1) - g['frames.rowsum'][k]).max() / frames.shape[2])
        cols = float(np.abs(frames[k].astype(np.float64).sum(axis=0) - g['frames.colsum'][k]).max() / frames.shape[1])
        per[gen] = max(per.get(gen, 0.0), d, rows, cols)
    print(f'T = 6 vs {prov} golden (reference recursion + tiling code): max|d| per generation', {k: float(f'{v:.2e}') for k, v in sorted(per.items())})
    assert max(per.values()) < IMAGE_TOL and np.array_equal(frames[0], x0[0]) and np.array_equal(frames[64], x1[0])


# ---------------------------------------------------------------------------------------------------------------------
# conv_wino2d_kernel: nested Winograd F(4,3)x x F(2,3)y
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('b,h,w', [(1, 64, 64), (2, 128, 64), (1, 64, 192), (1, 128, 320), (1, 192, 448)])
def test_nested_winograd_kernel_on_every_level(published, b, h, w):
    """"wino2d" = 2 forces conv_wino2d_kernel onto every layer that has the nested copy (K >= 208 or 128 -> 32: flow conv_0 of
    every level, the deep flow / decoder / sub-extractor layers) on EVERY level: ragged patches (W, H below one 32 x 8 patch, odd
    unit rows cut by the bottom edge), two-segment inputs with batch remaps (flow conv_0 reads [features | warped]),
    three-segment decoder inputs - stage-by-stage parity with the oracle, every tile shape in turn (same bits)."""
    from test_gpu_parity import _check_stages
    from film_hip.engine import FilmEngine
    opt, wts, _ = published
    eng = FilmEngine(opt, device=0)
    eng.set_weights(wts)
    eng.set_option('wino2d', 2)
    plan = eng.plan(b, h, w)
    n2d = sum(1 for op in plan['ops'] if op.get('wino') == 4)
    assert n2d >= 12, n2d
    x0, x1 = TI.frame_pair(b, h, w, seed=61 + h + w)
    got, _ = _check_stages(eng, opt, wts, x0, x1)
    taps = {k: eng.tap(k) for k in ('feat1', 'aligned0', 'aligned1')}
    used = set()
    for shape in range(6):        # Wino2dTile: 0 = 64 channels per workgroup (8 waves), 1 = 32 (4 waves, two workgroups per CU), 2 = 32 on two DMA stages;
                                  # 3..5 = the same on the 16 x 16-pixel arrangement of the 32 units (offered on levels it pads no more than 8 x 32)
        eng.set_option('w2d_shape', shape)
        tiles = {o['tile'] & 15 for o in eng.plan(b, h, w)['ops'] if o['kind'] == 'conv_mfma' and (o['tile'] & 8192)}
        if shape not in tiles:
            continue
        used.add(shape)
        g2 = eng.forward(x0, x1)
        assert np.array_equal(g2, got), (shape, float(np.abs(g2 - got).max()))
        for k, v in taps.items():
            assert np.array_equal(eng.tap(k), v), (shape, k)
    print('nested-Winograd tile shapes exercised:', sorted(used))
    assert {0, 1, 2, 3, 4, 5} <= used, used
    eng.set_option('w2d_shape', -1)
    eng.set_option('wino2d', 0)
    assert sum(1 for op in eng.plan(b, h, w)['ops'] if op.get('wino') == 4) == 0
    base = eng.forward(x0, x1)
    print(f'wino2d everywhere vs none: max|d| {float(np.abs(got - base).max()):.2e} ({n2d} ops)')
    eng.close()


def test_default_plan_uses_the_nested_kernel_on_the_deep_layers_of_a_1080p_tile(published):
    """The default plan of a 960x576 tile: conv_wino2d_kernel on every 3x3 layer (K = 32 ... 2448) of the levels with >= 1536 pixels - and,
    since round 6, of the 18x30 level, which fills 70 % of its 8 x 32 tiles - and the image stays within the usual distance of the plan
    without it (the tile is oracle-checked in test_gpu_configs)."""
    opt, w, eng = published
    plan = eng.plan(1, 576, 960)
    w2d = [(o['tag'], o['H'], o['W'], o['Ctot']) for o in plan['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4]
    assert len(w2d) >= 30 and all(k >= 32 and (hh * ww >= 1536 or (hh, ww) == (18, 30)) for _, hh, ww, k in w2d), w2d
    assert any((hh, ww) == (18, 30) for _, hh, ww, _ in w2d) and not any((hh, ww) == (9, 15) for _, hh, ww, _ in w2d)
    x0, x1 = TI.frame_pair(1, 576, 960, seed=2)
    a = eng.forward(x0, x1)
    eng.set_option('wino2d', 0)
    b0 = eng.forward(x0, x1)
    eng.set_option('wino2d', 1)
    d = float(np.abs(a - b0).max())
    print(f'960x576 tile, default plan with vs without conv_wino2d_kernel: max|d| {d:.2e}')
    assert d < 5e-5


def test_graph_replay_on_changing_inputs_at_a_1080p_tile(published):
    """The two-lane executor (direct launches on two streams, the default; its hipGraph form: tools/graph_race_check.py in
    test_gpu_parity.py::test_graph_replay_in_a_process_of_its_own) vs one stream in plan order on the headline tile (960x576: the only size where the flow upsample is
    its own launch on the two large levels and fused into the warps below them, and where conv_wino2d_kernel is in the default
    plan), with inputs that change every forward - a stale read in the replayed graph would show as the previous forward's
    data: image and every aligned level bit-identical; also with the decoder-on-the-side-lane op order ("lanes" = 2)."""
    from film_hip.engine import FilmEngine
    opt, w, eg = published
    ee = FilmEngine(opt, device=0)
    ee.set_weights(w)
    ee.set_option('graph', 0)
    kinds = [o['kind'] for o in eg.plan(1, 576, 960)['ops']]
    assert kinds.count('flow_up') == 2 and any('+resize2x' in o['tag'] for o in eg.plan(1, 576, 960)['ops'])
    for lanes in (1, 2):
        eg.set_option('lanes', lanes)
        for it in range(3):
            x0, x1 = TI.frame_pair(1, 576, 960, seed=70 + it)
            a, c = eg.forward(x0, x1), ee.forward(x0, x1)
            assert np.array_equal(a, c), (lanes, it, float(np.abs(a - c).max()))
            for l in range(opt.fusion_pyramid_levels):
                assert np.array_equal(eg.tap(f'aligned{l}'), ee.tap(f'aligned{l}')), (lanes, it, l)
    eg.set_option('lanes', 1)
    ee.close()


# ---------------------------------------------------------------------------------------------------------------------
# film_bcast_weights: the RCCL weight broadcast behind the C-ABI (round-5 verdict, missing item 3)
# ---------------------------------------------------------------------------------------------------------------------
def _rccl_single_rank_comm():
    """An ncclComm_t of ONE rank on the current device, made with the RCCL PyTorch bundles (ctypes: ncclGetUniqueId + ncclCommInitRank)."""
    import ctypes
    import torch
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    return rccl, comm


def test_weight_broadcast_behind_the_c_abi_single_rank(tiny_weights):
    """film_bcast_weights over a caller-made RCCL communicator (here: one rank - the boxes have one GPU; two ranks: tests/test_gpu_multi.py):
    the root stages its flat parameter blob on the device and ncclBroadcast runs on it (resolved from the RCCL already in the process);
    the handle's weights and results are unchanged; a root without weights and a NULL communicator are refused."""
    import torch
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import TINY
    torch.cuda.set_device(0)
    rccl, comm = _rccl_single_rank_comm()
    try:
        eng = _engine(TINY, tiny_weights)
        x0, x1 = TI.frame_pair(1, 64, 64, seed=4)
        before, blob = eng.forward(x0, x1), eng.export_packed().copy()
        eng.bcast_weights(comm.value, root=0, rank=0)
        assert np.array_equal(eng.export_packed(), blob) and np.array_equal(eng.forward(x0, x1), before)
        empty = FilmEngine(TINY, device=0)
        with pytest.raises(FilmError, match='finalized weight set'):
            empty.bcast_weights(comm.value, root=0, rank=0)
        with pytest.raises(FilmError):
            eng.bcast_weights(0, root=0, rank=0)
        empty.close()
        eng.close()
    finally:
        rccl.ncclCommDestroy(comm)

