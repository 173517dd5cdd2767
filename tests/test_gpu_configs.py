"""GPU parity on the BASELINE.json configurations themselves, DEFAULT planner options (the kernel families the
headline number runs on): HIP engine through the C-ABI vs (a) the CPU oracle on the full tensors and (b) the
committed vectors produced by the reference's own graph code (tests/golden/ref_*.npz, tools/make_ref_golden.py;
tf_*.npz when a TensorFlow-made set exists).  Inputs are structured frames with >= 8 px motion (tests/inputs.py)
and the reference's photos.  max|delta| per stage is printed."""
import os

import numpy as np
import pytest

from conftest import needs_extra_families

import golden_util as G
import inputs as TI
from test_gpu_parity import _check_stages, _engine

from eval.interpolator import Interpolator as _REAL_INTERPOLATOR   # (one test monkeypatches the module attribute)

pytestmark = pytest.mark.gpu

IMAGE_TOL = 1e-3          # north_star: |delta| < 1e-3 fp32 per pixel
HALF = np.full((1,), 0.5, np.float32)


@pytest.fixture(scope='module')
def published():
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    eng = _engine(PUBLISHED, w)
    yield PUBLISHED, w, eng
    eng.close()


def _interp(eng, w, **kw):
    """eval.interpolator.Interpolator sharing the module's engine (construction repacks 137.7 MB of weights)."""
    it = _REAL_INTERPOLATOR.__new__(_REAL_INTERPOLATOR)
    it._options, it._engine = eng.options, eng
    it._align, it._block_shape = kw.get('align') or None, kw.get('block_shape') or None
    return it


def test_tile_960x576_stage_by_stage(published):
    """One tile of the headline workload (a 960x540 patch padded to 960x576): every pyramid level's features,
    residual flows, flows, aligned pyramid and the image, default plan (F(4,3) on the 960/480/240/120-wide levels,
    F(2,3) on the 60-wide one, split-K on 30x18 and 15x9)."""
    from oracle import film_oracle as fo
    opt, w, eng = published
    x0, x1 = TI.frame_pair(1, 1080, 1920, seed=2, shift=(11, -17), fg_shift=(-9, 21))
    p0, _ = fo.pad_to_align(x0[:, :540, :960], 64)
    p1, _ = fo.pad_to_align(x1[:, :540, :960], 64)
    assert p0.shape == (1, 576, 960, 3)
    plan = eng.plan(1, 576, 960)
    fams = sorted({(o.get('tile', 0) & 256, o.get('tile', 0) & 2048) for o in plan['ops'] if o.get('kind') == 'conv_mfma'})
    print('conv families in the default plan (wino bit, F(4,3) bit):', fams)
    _check_stages(eng, opt, w, p0, p1)


def test_1080p_2x2_tiled_frame(published):
    """BASELINE configs[2], the config the metric is quoted on: Interpolator('', align=64, block_shape=[2,2]) on a
    1920x1080 pair (eval/interpolator_test.py:73-99, eval/interpolator.py:192-206)."""
    from oracle import film_oracle as fo
    opt, w, eng = published
    g, prov = G.load('1080p')
    x0, x1 = TI.frame_pair(1, 1080, 1920, seed=2, shift=(11, -17), fg_shift=(-9, 21))
    G.check_inputs(g, x0, x1)
    got = _interp(eng, w, align=64, block_shape=[2, 2])(x0, x1, HALF)
    d_gold = G.diff(g, 'image', got)
    want = fo.OracleInterpolator(w, align=64, block_shape=[2, 2])(x0, x1, HALF)
    d_or = float(np.abs(got - want).max())
    print(f'1080p 2x2: hip vs oracle max|d| {d_or:.3e} psnr {fo.psnr(got, want):.1f} dB; hip vs {prov} golden {d_gold:.3e}')
    assert got.shape == (1, 1080, 1920, 3) and d_or < IMAGE_TOL and d_gold < IMAGE_TOL


def test_1080p_2x2_device_resident_bench_path(published):
    """The path bench.py TIMES (round-4 verdict: it was only checked at 100 x 150): frames resident in HBM, DeviceInterpolator ->
    film_interpolate(FILM_MEM_DEVICE) on torch's stream, hipGraph replay on two lanes - at the headline size, against the
    reference-graph golden, on three DIFFERENT pairs in a row (a replay that read the previous forward's data would show), and
    bit-identical to the host-buffer path of test_1080p_2x2_tiled_frame."""
    import torch
    from film_hip.torch_io import DeviceInterpolator
    opt, w, eng = published
    g, prov = G.load('1080p')
    x0, x1 = TI.frame_pair(1, 1080, 1920, seed=2, shift=(11, -17), fg_shift=(-9, 21))
    G.check_inputs(g, x0, x1)
    dev_it = DeviceInterpolator(eng, align=64, block_shape=[2, 2])
    a, b = torch.from_numpy(x0).cuda(), torch.from_numpy(x1).cuda()
    first = dev_it(a, b)
    other = dev_it(b.flip(1).contiguous(), a.flip(2).contiguous())      # different data through the same cached graph
    again = dev_it(a, b)
    torch.cuda.synchronize()
    d_gold = G.diff(g, 'image', again.cpu().numpy())
    print(f'1080p 2x2, device-resident graph path: vs {prov} golden {d_gold:.3e}')
    assert d_gold < IMAGE_TOL and torch.equal(first, again) and not torch.equal(first, other)
    host = _interp(eng, w, align=64, block_shape=[2, 2])(x0, x1, HALF)
    assert np.array_equal(again.cpu().numpy(), host)


@needs_extra_families
@pytest.mark.parametrize('precision', [1, 2])
def test_1080p_2x2_in_the_opt_in_precision_modes(published, precision):
    """The opt-in bf16x6 / bf16x3 modes (FILM_EXTRA_FAMILIES=1 builds) at the headline size, 4 x 960 x 576 (round-4 verdict: they
    were only tested at 256 x 256 while bench.py printed a 0.265 'difference' for them - a harness bug that clobbered its inputs):
    against the reference-graph golden and against the fp32 mode of the same engine."""
    from film_hip.engine import FilmEngine
    opt, w, eng = published
    g, prov = G.load('1080p')
    x0, x1 = TI.frame_pair(1, 1080, 1920, seed=2, shift=(11, -17), fg_shift=(-9, 21))
    ref = _interp(eng, w, align=64, block_shape=[2, 2])(x0, x1, HALF)
    e2 = FilmEngine(opt, device=0)
    e2.set_weights(w)
    e2.set_option('precision', precision)
    got = _interp(e2, w, align=64, block_shape=[2, 2])(x0, x1, HALF)
    e2.close()
    d_gold, d_f32 = G.diff(g, 'image', got), float(np.abs(got - ref).max())
    print(f'1080p 2x2, precision {precision}: vs {prov} golden {d_gold:.3e}, vs the fp32 mode {d_f32:.3e}')
    assert d_gold < IMAGE_TOL / 4 and d_f32 < 1e-4


def test_photos_1024x768(published):
    """BASELINE configs[0]: photos/one.png + two.png (copies under tests/golden/), t = 0.5, align 64."""
    from eval import util
    from oracle import film_oracle as fo
    opt, w, eng = published
    g, prov = G.load('photos')
    a = util.read_image(os.path.join(G.GOLDEN, 'photo_one.png'))
    b = util.read_image(os.path.join(G.GOLDEN, 'photo_two.png'))
    G.check_inputs(g, a, b)
    got = _interp(eng, w, align=64, block_shape=[1, 1])(a[None], b[None], HALF)
    d_gold = G.diff(g, 'image', got)
    want = fo.OracleInterpolator(w, align=64)(a[None], b[None], HALF)
    d_or = float(np.abs(got - want).max())
    du8 = int(np.abs(util.to_uint8(got[0])[::4, ::4].astype(np.int32) - g['image_u8.s4'].astype(np.int32)).max())
    print(f'photos: hip vs oracle max|d| {d_or:.3e} psnr {fo.psnr(got, want):.1f} dB; hip vs {prov} golden {d_gold:.3e}; uint8 {du8}')
    assert d_or < IMAGE_TOL and d_gold < IMAGE_TOL and du8 <= 1


def test_vimeo_batch_of_8(published):
    """BASELINE configs[3]: eight 448x256 pairs in one call."""
    from oracle import film_oracle as fo
    opt, w, eng = published
    g, prov = G.load('vimeo')
    x0, x1 = TI.frame_pair(8, 256, 448, seed=3)
    G.check_inputs(g, x0, x1)
    got = _interp(eng, w, align=64)(x0, x1, np.full((8,), 0.5, np.float32))
    d_gold = G.diff(g, 'image', got)
    want = fo.OracleInterpolator(w, align=64)(x0, x1, None)
    d_or = float(np.abs(got - want).max())
    print(f'vimeo b8: hip vs oracle max|d| {d_or:.3e}; hip vs {prov} golden {d_gold:.3e}')
    assert d_or < IMAGE_TOL and d_gold < IMAGE_TOL


def test_256_against_reference_graph_taps(published):
    """BASELINE configs[1] with the structured pair: image, warped images and all flow pyramids of the model's aux
    dictionary (interpolator.py:191-199) vs the reference-graph golden."""
    opt, w, eng = published
    g, prov = G.load('256')
    x0, x1 = TI.frame_pair(1, 256, 256, seed=1)
    G.check_inputs(g, x0, x1)
    aux = eng.forward_with_aux(x0, x1)
    rep = {'image': float(np.abs(aux['image'] - g['image_full']).max()),
           'image_vs_f64_truth': float(np.abs(aux['image'] - g['image_f64']).max()) if 'image_f64' in g.files else 0.0,
           'x0_warped': G.diff(g, 'x0_warped', aux['x0_warped']), 'x1_warped': G.diff(g, 'x1_warped', aux['x1_warped'])}
    for d in ('forward', 'backward'):
        for l, v in enumerate(aux[f'{d}_flow_pyramid']):
            rep[f'{d}_flow{l}'] = G.diff(g, f'{d}_flow{l}', v)
        for l, v in enumerate(aux[f'{d}_residual_flow_pyramid']):
            rep[f'{d}_res{l}'] = G.diff(g, f'{d}_residual_flow{l}', v)
    print(prov, {k: float(f'{v:.1e}') for k, v in rep.items()})
    assert max(rep.values()) < 2e-4 and rep['image'] < IMAGE_TOL


def test_recursion_T2_tiled_against_reference_order(published, monkeypatch):
    """T = 2 recursion of a 2x1-tiled 200x176 pair.  (a) the CLI's default path - device-resident
    breadth-first driver behind eval/util.interpolate_recursively_from_memory; (b) the reference-order depth-first
    host generator; both vs the frames the reference's own eval/util.py produced (golden) incl. write_image's uint8
    rounding, and vs the oracle driven by the reference-order generator."""
    from eval import util
    from oracle import film_oracle as fo
    opt, w, eng = published
    g, prov = G.load('recursive')
    x0, x1 = TI.frame_pair(1, 200, 176, seed=4, shift=(6, -8), fg_shift=(-4, 9))
    G.check_inputs(g, x0, x1)
    it = _interp(eng, w, align=64, block_shape=[2, 1])
    dev_frames = np.stack(list(util.interpolate_recursively_from_memory([x0[0], x1[0]], 2, it)))
    monkeypatch.setenv('FILM_HOST_RECURSION', '1')
    host_frames = np.stack(list(util.interpolate_recursively_from_memory([x0[0], x1[0]], 2, it)))
    monkeypatch.delenv('FILM_HOST_RECURSION')
    assert dev_frames.shape == host_frames.shape == (5, 200, 176, 3)
    assert np.array_equal(dev_frames, host_frames), 'breadth-first device recursion must be bit-identical'
    orc = np.stack(list(util.interpolate_recursively_from_memory(
        [x0[0], x1[0]], 2, fo.OracleInterpolator(w, align=64, block_shape=[2, 1]))))
    d_or = float(np.abs(dev_frames - orc).max())
    d_gold = float(np.abs(dev_frames - g['frames']).max())
    u8 = np.stack([util.to_uint8(f) for f in dev_frames])
    du8 = int(np.abs(u8.astype(np.int32) - g['frames_u8'].astype(np.int32)).max())
    print(f'recursion T=2: hip vs oracle {d_or:.3e}; hip vs {prov} golden {d_gold:.3e}; uint8 {du8}')
    assert d_or < IMAGE_TOL and d_gold < IMAGE_TOL and du8 <= 1


def test_interpolator_cli_writes_reference_frames(published, tmp_path, monkeypatch):
    """eval.interpolator_cli end to end on a directory of two PNGs (device recursion by default): frame files and
    their pixels vs the host depth-first path."""
    from eval import interpolator_cli as cli
    from eval import util
    from eval import interpolator as interpolator_lib
    opt, w, eng = published
    x0, x1 = TI.frame_pair(1, 96, 160, seed=6, shift=(4, -6), fg_shift=(-3, 5))
    d = tmp_path / 'clip'
    d.mkdir()
    util.write_image(str(d / 'a_1.png'), x0[0])
    util.write_image(str(d / 'a_2.png'), x1[0])
    monkeypatch.setattr(interpolator_lib, 'Interpolator',
                        lambda model_path, align, block_shape, precision=0: _interp(eng, w, align=align, block_shape=block_shape))
    cli.main(['--pattern', str(tmp_path / '*'), '--times_to_interpolate', '2', '--block_height', '1', '--block_width', '2'])
    files = sorted(os.listdir(d / 'interpolated_frames'))
    assert files == [f'frame_{i:03d}.png' for i in range(5)]
    a, b = util.read_image(str(d / 'a_1.png')), util.read_image(str(d / 'a_2.png'))
    monkeypatch.setenv('FILM_HOST_RECURSION', '1')
    want = list(util.interpolate_recursively_from_memory([a, b], 2, _interp(eng, w, align=64, block_shape=[1, 2])))
    for f, wnt in zip(files, want):
        got = util.read_image(str(d / 'interpolated_frames' / f))
        assert np.array_equal(util.to_uint8(wnt), (got * 255 + 0.5).astype(np.uint8)), f


def test_device_to_uint8_and_the_file_pipeline_write_the_reference_bytes(published, tmp_path, monkeypatch):
    """Round 4, SURVEY 8 f1: film_to_uint8 gives write_image's bytes (eval/util.py:51-52) on edge values (negative, > 1, exact
    k / 255 and k / 255 +- 1 ulp, halves) and on random data, odd lengths and unaligned pointers included; and the device
    pipeline of the CLI (breadth-first recursion, device quantisation, uint8 D2H on a copy stream, thread-pool PNG encode) writes
    byte-identical FILES to the round-3 path (float32 frames back, host to_uint8, serial encode) - three input frames, T = 3,
    tiled."""
    import filecmp
    import torch
    from eval import interpolator_cli as cli
    from eval import util
    opt, w, eng = published
    rng = np.random.default_rng(12)
    k = np.arange(256, dtype=np.float32) / np.float32(255.0)
    edge = np.concatenate([np.array([-1.0, -1e-9, 0.0, 1.0, 1.0 + 1e-6, 2.5, 0.5 / 255, 1.5 / 255, 254.5 / 255, 0.49999997 / 255], np.float32),
                           k, np.nextafter(k, np.float32(2)), np.nextafter(k, np.float32(-1)),
                           (np.arange(256, dtype=np.float32) + np.float32(0.5)) / np.float32(255.0), rng.random(100003, dtype=np.float32) * 1.2 - 0.1])
    for off in (0, 1, 3):
        x = edge[off:]
        src = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        dst = torch.full((x.size + 5,), 77, dtype=torch.uint8, device='cuda')
        eng.to_uint8_device(src.data_ptr(), dst.data_ptr() + off, x.size - off, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        assert np.array_equal(got[off:off + x.size - off], util.to_uint8(x[:x.size - off])), off
        assert (got[:off] == 77).all() and (got[x.size:] == 77).all()
    d = tmp_path / 'clip'
    d.mkdir()
    frames = [TI.frame_pair(1, 128, 192, seed=30 + i, shift=(3, -4), fg_shift=(-2, 3))[0][0] for i in range(3)]
    for i, fr in enumerate(frames):
        util.write_image(str(d / f'in_{i}.png'), fr)
    inputs = cli.list_input_frames(str(d))
    it = _interp(eng, w, align=64, block_shape=[2, 2])
    new_dir, old_dir = str(tmp_path / 'new'), str(tmp_path / 'old')
    cli.output_frames([], new_dir)
    n, kept = util.interpolate_pairs_to_files(inputs, 0, 2, 2, 3, it, new_dir, keep=True)
    assert n == 17 and len(kept) == 17
    cli.output_frames(list(util.interpolate_recursively_from_files(inputs, 3, it)), old_dir)
    names = sorted(os.listdir(old_dir))
    assert names == sorted(os.listdir(new_dir)) == [f'frame_{i:03d}.png' for i in range(17)]
    for nm in names:
        assert filecmp.cmp(os.path.join(new_dir, nm), os.path.join(old_dir, nm), shallow=False), nm
    for i, nm in enumerate(names):
        assert np.array_equal(kept[i], (util.read_image(os.path.join(old_dir, nm)) * 255 + 0.5).astype(np.uint8))


def test_eval_cli_on_vimeo_sized_triplets(published, tmp_path):
    """eval.eval_cli end to end on the GPU - a model directory on disk (film_weights.npz) loaded by
    Interpolator(model_path), two Vimeo-90K-sized triplet folders (448x256, im1/im2/im3.png), results.csv with
    l1 / l2 / ssim / psnr per example and the mean row; the numbers against the same metrics of the oracle's prediction."""
    from eval import eval_cli, metrics as M, util
    from film_hip import weights as W
    from oracle import film_oracle as fo
    opt, w, _ = published
    model_dir = tmp_path / 'model'
    W.save_weights(str(model_dir), w)
    root = tmp_path / 'vimeo'
    truth = {}
    for k, seq in enumerate(('00001/0001', '00001/0002')):
        x0, x1 = TI.frame_pair(1, 256, 448, seed=20 + k, shift=(4, -6), fg_shift=(-3, 5))
        mid, _ = TI.frame_pair(1, 256, 448, seed=20 + k, shift=(2, -3), fg_shift=(-2, 3))   # any plausible ground truth
        d = root / seq
        os.makedirs(d)
        for name, img in (('im1.png', x0[0]), ('im2.png', mid[0]), ('im3.png', x1[0])):
            util.write_image(str(d / name), img)
        truth[seq.replace('/', '_')] = tuple(util.read_image(str(d / n)) for n in ('im1.png', 'im2.png', 'im3.png'))
    out = tmp_path / 'out'
    assert eval_cli.main(['--model_path', str(model_dir), '--triplet_dir', str(root), '--output_dir', str(out), '--output_frames']) == 0
    rows = [l.strip().split(', ') for l in open(out / 'results.csv')]
    assert rows[0] == ['key', 'l1', 'l2', 'ssim', 'psnr'] and [r[0] for r in rows[1:]] == ['00001_0001', '00001_0002', 'mean']
    orc = fo.OracleInterpolator(w, align=64)
    for r in rows[1:3]:
        a, y, b = truth[r[0]]
        pred = np.clip(orc(a[None], b[None], None), 0.0, 1.0)
        want = [M.l1(pred, y[None]), M.l2(pred, y[None]), M.ssim(pred, y[None]), M.psnr(pred, y[None])]
        got = [float(v) for v in r[1:]]
        print(r[0], 'hip', got, 'oracle', want)
        assert np.allclose(got[:3], want[:3], rtol=0, atol=2e-6) and abs(got[3] - want[3]) < 2e-3   # dB
    assert sorted(os.listdir(out))[:4] == ['00001_0001_image.png', '00001_0001_x0.png', '00001_0001_x1.png', '00001_0001_y.png']


def test_rccl_broadcast_path_single_rank(published):
    """The device side of the N-GPU start-up on the hardware that is available (one GPU): torch.distributed with the
    nccl (= RCCL) backend and world_size 1, film_hip.sharding.broadcast_weights with a CUDA blob (film_export_packed to
    device memory, dist.broadcast on it), then film_import_packed from that device blob into a second engine, whose
    result must be bit-identical.  (N > 1 is covered on CPU with gloo: tests/test_dist_cpu.py.)"""
    import socket
    import torch
    import torch.distributed as dist
    from film_hip.engine import FilmEngine
    from film_hip.sharding import broadcast_weights
    opt, w, eng = published
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        broadcast_weights(eng, dist, src=0, device=dev)
        n = eng.packed_size()
        blob = torch.empty(n, dtype=torch.float32, device=dev)
        eng.export_packed_device(blob.data_ptr(), n)
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize(dev)
        e2 = FilmEngine(opt, device=0)
        e2.import_packed_device(blob.data_ptr(), n)
        assert float(blob.abs().sum()) > 0 and np.array_equal(e2.export_packed(), eng.export_packed())
        x0, x1 = TI.frame_pair(1, 128, 192, seed=8)
        assert np.array_equal(e2.forward(x0, x1), eng.forward(x0, x1))
        e2.close()
    finally:
        dist.destroy_process_group()


def test_untiled_4k_frame_with_buffers_above_4gib(published):
    """ONE untiled 3840x2240 pair (film_forward, no block_shape): feat0 / warped0 are 4.4 GB and aligned0 4.95 GB, beyond
    what a 32-bit buffer offset reaches - conv_wino43_kernel addresses relative to each workgroup's own halo rows.
    Against the committed oracle fixture (tools/make_big_golden.py: stride-16 sample + float64 row / column sums of the
    full image; the oracle itself needs minutes and ~30 GB of host memory for this frame)."""
    import inputs
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_big_untiled.npz')
    if not os.path.isfile(path):
        pytest.skip('tests/golden/oracle_big_untiled.npz not generated (python tools/make_big_golden.py, about an hour of CPU)')
    g = np.load(path)
    _, h, w, _ = (int(v) for v in g['shape'])
    opt, wts, eng = published
    x0, x1 = inputs.frame_pair(1, h, w, 41)
    assert np.allclose([x0.astype(np.float64).sum(), x1.astype(np.float64).sum()], g['in_checksum'], rtol=0, atol=1e-6)
    plan = eng.plan(1, h, w)
    assert max(b['floats'] for b in plan['buffers']) * 4 > 2 ** 32
    got = eng.forward(x0, x1)
    st = int(g['stride'])
    d = float(np.abs(got[:, ::st, ::st, :] - g['sample']).max())
    rows = float(np.abs(got.astype(np.float64).sum(axis=2) - g['rowsum']).max() / w)
    cols = float(np.abs(got.astype(np.float64).sum(axis=1) - g['colsum']).max() / h)
    print(f'untiled {h}x{w}: hip vs oracle fixture sample max|d| {d:.3e}, row sums {rows:.3e}, column sums {cols:.3e} per pixel')
    assert d < IMAGE_TOL and rows < IMAGE_TOL and cols < IMAGE_TOL
    # the bottom rows are the ones behind the 4 GiB mark of the level-0 buffers
    assert float(np.abs(got[:, -64::st, ::st, :] - g['sample'][:, -(64 // st):]).max()) < IMAGE_TOL


def test_two_pairs_in_one_invocation_cross_4gib_through_the_batch_index(published):
    """Two 3840x1216 pairs in ONE model invocation: feat0 / warped0 hold four images = 4.8 GB and aligned0 two = 5.4 GB,
    so the second pair lies behind the 4 GiB mark of those buffers (the batch-index term of the same 64-bit base address
    that rows of a single large frame go through).  Frame pairs are independent: the result must have the bits of two
    one-pair invocations, whose buffers are 2.4-2.7 GB."""
    import inputs
    opt, wts, eng = published
    h, w = 1216, 3840
    x0, x1 = inputs.frame_pair(2, h, w, 43)
    plan = eng.plan(2, h, w)
    assert max(b['floats'] for b in plan['buffers']) * 4 > 2 ** 32 and plan['offset32_buffer_bytes'] < 0xFFF00000
    both = eng.forward(x0, x1)
    eng.set_option('max_batch', 1)
    try:
        one = eng.forward(x0, x1)
    finally:
        eng.set_option('max_batch', 0)
    assert np.array_equal(both, one), float(np.abs(both - one).max())
    # and not trivially equal: the two pairs differ
    assert not np.array_equal(both[0], both[1])


@pytest.mark.parametrize('shape,block', [((1, 1080, 1920, 3), (2, 2)), ((1, 720, 1280, 3), (2, 2)), ((1, 256, 256, 3), None), ((2, 360, 640, 3), (2, 1)),
                                         ((1, 1080, 1920, 3), (4, 4)), ((1, 540, 960, 3), (1, 2))])
def test_host_buffer_pipeline_is_bit_identical(published, shape, block):
    """film_interpolate with host buffers (the reference's numpy -> numpy call, eval/interpolator.py:152-209) pipelines its copies with the work
    (option "host_overlap": the first layers run per input frame, the last layer per tile half - batch parts of a convolution).  Same bits as the
    unpipelined call, with one frame and several, even and odd block rows (the tail split needs one frame and an even number of block rows),
    pageable arrays, a caller-owned `out`, and pinned arrays."""
    import inputs
    from film_hip.torch_io import pinned_frame
    opt, wts, eng = published
    b, h, w, _ = shape
    x0, x1 = inputs.frame_pair(b, h, w, 77 + h)
    eng.set_option('host_overlap', 0)
    try:
        plain = eng.interpolate_frames(x0, x1, align=64, block_shape=block)
    finally:
        eng.set_option('host_overlap', 1)
    piped = eng.interpolate_frames(x0, x1, align=64, block_shape=block)
    assert np.array_equal(piped, plain), float(np.abs(piped - plain).max())
    mine = np.full_like(x0, -1.0)
    got = eng.interpolate_frames(x0, x1, align=64, block_shape=block, out=mine)
    assert got is mine and np.array_equal(mine, plain)
    p0, p1, po = pinned_frame(shape), pinned_frame(shape), pinned_frame(shape)
    p0[:] = x0; p1[:] = x1
    assert np.array_equal(eng.interpolate_frames(p0, p1, align=64, block_shape=block, out=po), plain)
    with pytest.raises(ValueError):
        eng.interpolate_frames(x0, x1, align=64, block_shape=block, out=np.empty((1, 2, 2, 3), np.float32))
