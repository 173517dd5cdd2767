"""CPU tests of the native library WITHOUT a GPU: symbol export, weight packing, the planner.

The op list the engine would launch (film_plan_json) is executed by the numpy interpreter in
tests/plan_interp.py and compared with the oracle: this validates buffer layout, concat-by-slices,
batch remaps, the folded NN-upsample and the weight permutation/zero padding on the CPU.  No compute
happens inside libfilm_hip.so in these tests (plan-only handles refuse to).
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import oracle_options, prefers_extra_families, ROOT


def _w2d_level(h, w):
    """The level-size rule of conv_wino2d_kernel (film_planner.cpp): >= 1536 pixels, or >= 256 pixels filling >= 65 % of the 8 x 32 tiles."""
    px = h * w
    return px >= 1536 or (px >= 256 and px * 100 >= 65 * (-(-h // 8) * 8) * (-(-w // 32) * 32))


def test_library_exports_every_declared_symbol():
    from film_hip import engine
    lib = engine.load_library()
    header = open(os.path.join(ROOT, 'include', 'film_hip.h')).read()
    declared = set(re.findall(r'^(?:int|void|uint32_t|const char\*)\s+(film_[a-z_0-9]+)\s*\(', header, flags=re.M))
    assert declared == set(engine.EXPORTED_SYMBOLS), declared ^ set(engine.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert engine.FilmEngine.version().startswith('gfx950')


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly without a GPU (no silent CPU/eager fallback)."""
    import torch
    from film_hip.engine import FilmEngine, FilmError, FILM_ERR_NO_DEVICE
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    x = np.zeros((1, 32, 32, 3), np.float32)
    with pytest.raises(FilmError) as e:
        eng.forward(x, x)
    assert e.value.code in (FILM_ERR_NO_DEVICE, -2)
    if not torch.cuda.is_available():
        with pytest.raises(FilmError) as e2:
            FilmEngine(TINY, device=0)
        assert e2.value.code == FILM_ERR_NO_DEVICE
        from eval.interpolator import Interpolator
        from film_hip import weights as W
        with pytest.raises(FilmError):
            Interpolator('', weights=W.make_synthetic_weights(TINY, 0), options=TINY)


def test_default_config_is_published_architecture():
    from film_hip import engine
    lib = engine.load_library()
    c = engine._Config()
    assert lib.film_default_config(ctypes.byref(c)) == 0
    assert (c.pyramid_levels, c.fusion_pyramid_levels, c.specialized_levels, c.sub_levels, c.filters) == (7, 5, 3, 4, 64)
    assert list(c.flow_convs)[:4] == [3, 3, 3, 3] and list(c.flow_filters)[:4] == [32, 64, 128, 256]


def test_weight_table_matches_survey():
    from film_hip import weights as W
    from film_hip.options import PUBLISHED
    assert W.num_params(PUBLISHED) == 34436667
    assert W.feature_channels(PUBLISHED) == [64, 192, 448, 960, 960, 960, 960]
    specs = {n: s for n, s, _ in W.weight_specs(PUBLISHED)}
    assert specs['predict_flow/flow_predictor_shared/conv_0'] == (3, 3, 1920, 256)
    assert specs['fusion/convs_3_0'] == (2, 2, 1930, 512)
    assert specs['fusion/convs_3_1'] == (3, 3, 2442, 512)
    assert specs['fusion/output_conv'] == (1, 1, 64, 3)


def test_set_weight_validation(tiny_weights):
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    with pytest.raises(FilmError):
        eng.set_weights({'nope/kernel': np.zeros((1, 1, 1, 1), np.float32)})
    with pytest.raises(FilmError):
        eng.set_weights({'fusion/output_conv/kernel': np.zeros((1, 1, 5, 3), np.float32)})
    with pytest.raises(FilmError):   # finalize with tensors missing
        eng.set_weights({'fusion/output_conv/bias': np.zeros((3,), np.float32)})
    eng.set_weights(tiny_weights)
    blob = eng.export_packed()                      # flat parameter blob: per layer the HWIO kernel, then the bias
    assert blob.size == eng.packed_size() == sum(v.size for v in tiny_weights.values())
    eng2 = FilmEngine(TINY, device=-1)
    eng2.import_packed(blob)
    assert np.array_equal(eng2.export_packed(), blob)
    # both handles build the same kernel layouts, group by group (default group first, all four on request)
    assert np.array_equal(eng2.export_layouts(), eng.export_layouts())
    n1 = eng.export_layouts().size
    for e in (eng, eng2):
        e.set_option('pack_groups', 4)
    assert eng.export_layouts().size > 1.5 * n1 and np.array_equal(eng2.export_layouts(), eng.export_layouts())
    assert np.array_equal(eng.export_layouts()[:n1], eng2.export_layouts()[:n1])


def test_packing_permutes_fusion_inputs(tiny_weights):
    """fusion/convs_i_1 rows: internal [feat0|feat1|img0 img1 bflow fflow 0x6|net] <- reference order."""
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    blob = eng.export_layouts()
    plan = eng.plan(1, 32, 32)
    L = {l['name']: l for l in plan['layers']}['fusion/convs_0_1']
    C = W.feature_channels(TINY)[0]
    ref = tiny_weights['fusion/convs_0_1/kernel']
    # MFMA-conv layers are packed K-major ([Cout][tap][ctot]); back to HWIO for the comparison
    packed = blob[L['w_off']:L['w_off'] + 9 * L['ctot'] * L['cout']].reshape(L['cout'], 3, 3, L['ctot'])
    packed = packed.transpose(1, 2, 3, 0)
    assert L['ctot'] == 2 * C + 16 + L['cout']
    assert np.array_equal(packed[:, :, :C], ref[:, :, 3:3 + C])                       # feat0
    assert np.array_equal(packed[:, :, C:2 * C], ref[:, :, 6 + C:6 + 2 * C])           # feat1
    assert np.array_equal(packed[:, :, 2 * C:2 * C + 3], ref[:, :, 0:3])               # img0
    assert np.array_equal(packed[:, :, 2 * C + 3:2 * C + 6], ref[:, :, 3 + C:6 + C])   # img1
    assert np.array_equal(packed[:, :, 2 * C + 6:2 * C + 10], ref[:, :, 6 + 2 * C:10 + 2 * C])  # flows
    assert not packed[:, :, 2 * C + 10:2 * C + 16].any()                               # zero rows
    assert np.array_equal(packed[:, :, 2 * C + 16:], ref[:, :, 10 + 2 * C:])           # net


@pytest.mark.parametrize('b,h,w', [(1, 32, 32), (2, 32, 48), (1, 64, 40)])
def test_plan_interpreter_matches_oracle_tiny(tiny_weights, b, h, w):
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    from oracle import film_oracle as fo
    import plan_interp as pi
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    eng.set_option('pack_groups', 4)     # every layout copy, so that the interpreter can check all of them
    if (b, h, w) == (2, 32, 48):
        eng.set_option('fuse', 0)        # one op per reference op (flow_up / flow_add / warp_c3 / pack_flow / pool launches)
    plan = eng.plan(b, h, w)
    assert any(op['kind'] == 'flow_up' for op in plan['ops']) == ((b, h, w) == (2, 32, 48))
    rng = np.random.default_rng(h * 7 + w)
    x0 = rng.random((b, h, w, 3), dtype=np.float32)
    x1 = rng.random((b, h, w, 3), dtype=np.float32)
    arena = pi.run_plan(plan, eng.export_layouts(), x0, x1)
    want, aux = fo.film_forward(x0, x1, tiny_weights, oracle_options(TINY), return_aux=True)
    fc = W.feature_channels(TINY)
    for l in range(TINY.pyramid_levels):
        f = pi.tap(plan, arena, f'feat{l}')
        assert np.abs(f[:b] - aux['feature_pyramids'][0][l]).max() < 1e-5
        assert np.abs(f[b:] - aux['feature_pyramids'][1][l]).max() < 1e-5
        r = pi.tap(plan, arena, f'res{l}')
        assert np.abs(r[:b] - aux['forward_residual_flow_pyramid'][l]).max() < 1e-5
        assert np.abs(r[b:] - aux['backward_residual_flow_pyramid'][l]).max() < 1e-5
    for l in range(TINY.fusion_pyramid_levels):
        a = pi.aligned_to_reference(pi.tap(plan, arena, f'aligned{l}'), fc[l])
        assert np.abs(a - aux['aligned_pyramid'][l]).max() < 1e-5
    assert np.abs(pi.tap(plan, arena, 'out') - want).max() < 1e-5


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_upsample_2x2_layer_forms_by_option(tiny_weights, mode):
    """Option "fold2x2": 1 (default) = the difference form on conv_fold4_kernel (fold = 3, tile flag 16384, the S / Sx / Sy / W11 copy),
    2 = four sub-pixel phases on the general kernel (fold = 2), 0 = one 2x2 convolution with the upsample in its gather - every form
    interpreted from its own packed weights against the oracle (fusion.py:133-135)."""
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import TINY
    from oracle import film_oracle as fo
    import plan_interp as pi
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    eng.set_option('pack_groups', 4)     # every layout copy, so that the interpreter can check all of them
    eng.set_option('fold2x2', mode)
    plan = eng.plan(1, 32, 48)
    ups = [op for op in plan['ops'] if op['kind'] == 'conv_mfma' and op['ksize'] == 2]
    assert len(ups) == TINY.fusion_pyramid_levels - 1
    for op in ups:
        assert op['fold'] == {0: 0, 1: 3, 2: 2}[mode]
        assert bool(op['tile'] & 16384) == (mode == 1)
        assert (op['segs'][0]['up'] != 0) == (mode == 0)
        if mode == 1:
            assert op['w_off'] == op['wf4_off'] >= 0
    rng = np.random.default_rng(11)
    x0 = rng.random((1, 32, 48, 3), dtype=np.float32)
    x1 = rng.random((1, 32, 48, 3), dtype=np.float32)
    arena = pi.run_plan(plan, eng.export_layouts(), x0, x1)
    want = fo.film_forward(x0, x1, tiny_weights, oracle_options(TINY))
    assert np.abs(pi.tap(plan, arena, 'out') - want).max() < 1e-5
    with pytest.raises(FilmError):
        eng.set_option('fold2x2', 3)
    eng.close()


@pytest.fixture(scope='module')
def published_packed():
    """(weights, plan-only engine of the published net with all four layout groups packed, the exported layout blob):
    1.1 GB of packing + export that the interpreter tests share."""
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED
    w = W.make_synthetic_weights(PUBLISHED, seed=0)
    eng = FilmEngine(PUBLISHED, device=-1)
    eng.set_weights(w)
    eng.set_option('pack_groups', 4)     # every layout copy, so that the interpreter can check all of them
    yield w, eng, eng.export_layouts()
    eng.close()


def test_plan_interpreter_matches_oracle_published_64(published_packed):
    from oracle import film_oracle as fo
    import plan_interp as pi
    w, eng, layouts = published_packed
    plan = eng.plan(1, 64, 64)
    rng = np.random.default_rng(3)
    x0 = rng.random((1, 64, 64, 3), dtype=np.float32)
    x1 = rng.random((1, 64, 64, 3), dtype=np.float32)
    arena = pi.run_plan(plan, layouts, x0, x1)
    want = fo.film_forward(x0, x1, w, fo.Options())
    assert np.abs(pi.tap(plan, arena, 'out') - want).max() < 2e-5
    # algorithmic conv FLOPs of the plan == SURVEY.md 8(d): 4 246 240.6875 FLOP per padded pixel
    flops = sum(op['flops'] for op in plan['ops'] if op['kind'].startswith('conv') or op['kind'] == 'flow_head')
    assert abs(flops / (64 * 64) - 4246240.6875) / 4246240.6875 < 1e-5
    warp_bytes = sum(op['bytes'] for op in plan['ops'] if op['kind'] == 'warp')
    assert abs(warp_bytes / (64 * 64) - 5201.58) / 5201.58 < 1e-3
    # "wino2d" = 2 (every layer that has the nested-Winograd copy runs conv_wino2d_kernel; the interpreter checked the packed
    # [mu][nu] copy of those layers above) AND "lanes" = 3 (the op order of option "lanes" = 2 - coarse decoder levels emitted
    # right behind the aligned levels they read, on the side lane - at this small size) in ONE more interpreter run: same ops,
    # another order, another kernel family; the interpreter's arithmetic does not depend on the family -> same bits
    # default rule: levels of >= 1536 pixels, and smaller ones (>= 256) that fill >= 65 % of their 8 x 32 tiles (here: 32x32 yes, 16x16 no)
    assert all(_w2d_level(o['H'], o['W']) for o in plan['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4)
    assert {(o['H'], o['W']) for o in plan['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4} == {(64, 64), (32, 32)}
    eng.set_option('wino2d', 2)
    eng.set_option('lanes', 3)
    plan3 = eng.plan(1, 64, 64)
    eng.set_option('lanes', 1)
    eng.set_option('wino2d', 1)
    w2d_ops = [o for o in plan3['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4]
    assert len(w2d_ops) >= 8 and all(o['w2d_off'] >= 0 and (o['tile'] & 8192) for o in w2d_ops)
    tags, tags3 = [o['tag'] for o in plan['ops']], [o['tag'] for o in plan3['ops']]
    # (the nested kernel also fuses the average pool behind a sub-extractor stage, which the direct kernel of a small level leaves to
    # a pool launch: compare the plans without those)
    strip = lambda ops: sorted(o['tag'].replace('+pool', '').replace('+output_conv', '') for o in ops if not (o['kind'] == 'pool' and o['tag'].startswith('feat_')) and o['tag'] != 'fusion_out:fusion/output_conv')
    assert strip(plan['ops']) == strip(plan3['ops']) and tags != tags3
    assert {o['lane'] for o in plan3['ops'] if o['tag'].startswith(('fusion_l3', 'fusion_l2'))} == {1}
    assert {o['lane'] for o in plan3['ops'] if o['tag'].startswith(('fusion_l1', 'fusion_l0'))} == {0}
    arena3 = pi.run_plan(plan3, layouts, x0, x1)
    assert np.array_equal(pi.tap(plan3, arena3, 'out'), pi.tap(plan, arena, 'out'))


@prefers_extra_families
def test_precision_modes_choose_kernel_families_by_shape_only():
    """The kernel family of a layer is a pure function of (layer shape, precision option) - never of batch size or
    timing.  Mode 0: no split kernels, conv_wino43_kernel (wino = 3) / conv_wino_kernel (1) by level width; mode 1: conv_halo_split_kernel (split = 1); mode 2: conv_winox3_kernel
    (wino = 2) on the Cout % 128 == 0 layers, conv_halo_split_kernel<..,3> (split = 2) on the others, conv_foldx3_kernel
    (fold with split = 2) on the decoder's large upsample + 2x2 layers; tile ids carry the matching flags."""
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import PUBLISHED
    WINO, SPLIT, X3, FOLDX3, F43, W2D = 256, 128, 512, 1024, 2048, 8192
    from conftest import has_extra_families
    eng = FilmEngine(PUBLISHED, device=-1)
    fam = {}
    modes = (0, 1, 2) if has_extra_families() else (0,)
    if not has_extra_families():     # the default library holds the fp32 families only and says so
        for key, val in (('precision', 1), ('precision', 2), ('winograd', 2), ('halo_all', 1)):
            with pytest.raises(FilmError, match='FILM_EXTRA_FAMILIES'):
                eng.set_option(key, val)
    for mode in modes:
        eng.set_option('precision', mode)
        per_batch = []
        for b in (1, 4):
            ops = [op for op in eng.plan(b, 256, 448)['ops'] if op['kind'] == 'conv_mfma']
            per_batch.append([(op['tag'], op['split'], op['wino'], op['fold']) for op in ops])
            for op in ops:
                t = op['tile']
                assert bool(t & FOLDX3) == (op['fold'] == 2 and op['split'] == 2)
                assert bool(t & WINO) == (op['wino'] in (1, 2, 3)) and bool(t & SPLIT) == (op['split'] != 0 and not op['fold'])
                assert bool(t & W2D) == (op['wino'] == 4) and (op['wino'] != 4 or (op['Ctot'] % 16 == 0 and _w2d_level(op['H'], op['W'])))
                assert bool(t & X3) == (op['wino'] == 2 or (op['split'] == 2 and not op['fold']))
                assert bool(t & F43) == (op['wino'] == 3)
        assert per_batch[0] == per_batch[1]
        fam[mode] = per_batch[0]
    assert all(s == 0 and w in (0, 1, 3, 4) for _, s, w, _ in fam[0])
    assert any(w == 4 for _, _, w, _ in fam[0])      # the nested-Winograd family: deep-K layers of the 128x224 level (mode 0 only)
    if has_extra_families():
        assert not any(w == 4 for m in (1, 2) for _, _, w, _ in fam[m])
        assert any(s == 1 for _, s, _, _ in fam[1]) and all(w == 0 for _, s, w, _ in fam[1] if s)
        assert any(w == 2 for _, _, w, _ in fam[2]) and any(s == 2 and not f for _, s, _, f in fam[2])
        assert any(s == 2 and f == 2 for _, s, _, f in fam[2])
        assert not any(s == 1 or w == 1 and s for _, s, w, _ in fam[2])
    else:
        assert all(w in (0, 3, 4) for _, _, w, _ in fam[0])     # no F(2,3) / halo / split kernel can be selected
    with pytest.raises(FilmError):
        eng.set_option('precision', 3)


def test_every_buffer_lies_inside_the_arena_and_split_k_is_batch_independent():
    """Workspace layout: buffers (including the split-K partial-sum regions added while the ops are emitted) are
    disjoint and inside arena_floats; the split-K factor of a layer depends on the level size and the layer, not on
    the batch (results must not depend on the batch size)."""
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED
    eng = FilmEngine(PUBLISHED, device=-1)
    splits = []
    for b in (1, 3):
        plan = eng.plan(b, 128, 192)
        spans = sorted((bf['off'], bf['off'] + bf['floats']) for bf in plan['buffers'])
        assert all(a[1] <= c[0] for a, c in zip(spans, spans[1:])), 'overlapping buffers'
        assert spans[-1][1] <= plan['arena_floats']
        names = {bf['name'] for bf in plan['buffers']}
        ks = {op['tag']: op['ksplit'] for op in plan['ops'] if op['kind'] == 'conv_mfma'}
        assert any(k > 1 for k in ks.values())
        for tag, k in ks.items():
            assert (k > 1) == (('splitk:' + tag) in names)
        splits.append(ks)
    assert splits[0] == splits[1]
    eng.set_option('splitk', 0)
    assert all(op['ksplit'] == 1 for op in eng.plan(1, 128, 192)['ops'] if op['kind'] == 'conv_mfma')


def test_plan_shape_errors(tiny_weights):
    from film_hip.engine import FilmEngine, FilmError, FILM_ERR_INVALID
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    with pytest.raises(FilmError) as e:
        eng.plan(1, 36, 32)          # not divisible by 2^(pyramid_levels-1) = 8 (options.py:36-37)
    assert e.value.code == FILM_ERR_INVALID and 'divisible' in e.value.msg
    with pytest.raises(FilmError):
        eng.plan(1, 8, 4)            # a warped level would be < 2x2 (tfa dense_image_warp requirement)
    with pytest.raises(FilmError):
        eng.plan(0, 32, 32)


def test_invalid_options_rejected():
    from film_hip.options import Options
    with pytest.raises(ValueError):
        Options(pyramid_levels=3, fusion_pyramid_levels=5).validate()   # interpolator.py:120-122
    with pytest.raises(ValueError):
        Options(filters=20).validate()


def test_executor_option_is_three_valued():
    """Option "graph" (include/film_hip.h): 0 = one stream, 1 = hipGraph replay, 2 = direct two-lane launches (the default since
    round 5); anything else is refused with a message.  The plan itself (ops, lanes, cross-lane edges) does not depend on it."""
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    base = eng.plan(1, 64, 64)
    for v in (0, 1, 2):
        eng.set_option('graph', v)
        assert eng.plan(1, 64, 64) == base
    for v in (-1, 3):
        with pytest.raises(FilmError, match='graph'):
            eng.set_option('graph', v)
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'film_hip.h')).read()
    assert '"graph"   0/1/2' in header and '2 (default)' in header


def test_lane_analysis_orders_every_conflict(tiny_weights):
    """Two-stream replay: for every pair of ops on different lanes that touch overlapping channels of one buffer
    (RAW / WAR / WAW) the later one must be ordered behind the earlier one through the xdeps edges + per-lane
    program order (happens-before closure computed here independently from the plan JSON)."""
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED, TINY
    for opt, shape, lanes in ((TINY, (2, 64, 96), 1), (PUBLISHED, (1, 128, 192), 1), (PUBLISHED, (4, 576, 960), 1),
                              (PUBLISHED, (4, 576, 960), 2)):   # the last: option "lanes" = 2, coarse decoder levels on the side lane
        eng = FilmEngine(opt, device=-1)
        eng.set_option('lanes', lanes)
        plan = eng.plan(*shape)
        bufs = {b['name']: b for b in plan['buffers']}
        ops = plan['ops']

        def acc(v):
            if v is None or not v.get('buf'):
                return None
            b = bufs[v['buf']]
            if b['C'] == 0 or v['stride'] != b['C']:
                return (v['buf'], 0, 1 << 30)
            c0 = (v['off'] - b['off']) % b['C']
            return (v['buf'], c0, c0 + v['C'])

        def rw(op):
            rd = [acc(sg['v']) for sg in op.get('segs', [])] if op['kind'] == 'conv_mfma' else [acc(op.get('in')), acc(op.get('in2'))]
            rd = list(rd)
            rd += [acc(op.get('in3')), acc(op.get('img_in')), acc(op.get('pack_b')), acc(op.get('pack_f'))]
            return [a for a in rd if a], [a for a in [acc(op.get('out')) if not (op.get('pw_out') or {}).get('buf') else None, acc(op.get('pw_out')), acc(op.get('out2')), acc(op.get('img_out'))] if a]

        def hit(a, b):
            return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]

        n = len(ops)
        assert {o['lane'] for o in ops} == {0, 1}
        # done[j] = set of ops guaranteed complete before op j starts
        before = [set() for _ in range(n)]
        last = {0: None, 1: None}
        for j, o in enumerate(ops):
            preds = list(o['xdeps'])
            if last[o['lane']] is not None:
                preds.append(last[o['lane']])
            for p in preds:
                assert p < j
                before[j] |= before[p] | {p}
            last[o['lane']] = j
        acc_rw = [rw(o) for o in ops]
        for j in range(n):
            rj, wj = acc_rw[j]
            for i in range(j):
                if ops[i]['lane'] == ops[j]['lane']:
                    continue
                ri, wi = acc_rw[i]
                conflict = any(hit(w, r) for w in wi for r in rj) or any(hit(w, w2) for w in wi for w2 in wj) or \
                    any(hit(r, w2) for r in ri for w2 in wj)
                if conflict:
                    assert i in before[j], (ops[i]['tag'], ops[j]['tag'])
        # no implied waits: an op is waited for at most once by the other lane, so a captured graph node has at most two
        # children (HIP 7.x replays a node whose only parent has >= 5 earlier children without waiting for that parent:
        # tools/experiments/graph_single_parent_race.hip)
        waiters = [d for o in ops for d in o['xdeps']]
        assert len(waiters) == len(set(waiters))
        for lane in (0, 1):
            seq = [d for o in ops if o['lane'] == lane for d in o['xdeps']]
            assert seq == sorted(seq)


def test_plan_interpreter_fused_ops_published_256(published_packed):
    """The default plan of a 256x256 pair carries every fusion (fuse = 31): flow upsample inside the warps, v = res + up
    inside the flow heads, image warps inside the feature warps, average pools inside the F(4,3) convolutions, the RGB head
    inside the last decoder convolution - and the
    interpreter, which gives each fused op the semantics of the separate reference ops, still reproduces the oracle.
    With fuse = 0 the plan has one op per reference op and the same result."""
    from oracle import film_oracle as fo
    import plan_interp as pi
    w, eng, layouts = published_packed
    rng = np.random.default_rng(9)
    x0 = rng.random((1, 256, 256, 3), dtype=np.float32)
    x1 = rng.random((1, 256, 256, 3), dtype=np.float32)
    want = fo.film_forward(x0, x1, w, fo.Options())
    counts = {}
    try:
        for fuse in (0, 31):
            eng.set_option('fuse', fuse)
            plan = eng.plan(1, 256, 256)
            tags = [op['tag'] for op in plan['ops']]
            counts[fuse] = len(tags)
            for mark in ('+pool', '+misc16', '+resize2x', '+v=res+up', '+output_conv'):
                assert any(mark in t for t in tags) == (fuse == 31), (fuse, mark)
            if fuse == 31:      # (an unfused plan is interpreted by test_plan_interpreter_matches_oracle_tiny[2-32-48])
                arena = pi.run_plan(plan, layouts, x0, x1)
                assert np.abs(pi.tap(plan, arena, 'out') - want).max() < 2e-5
    finally:
        eng.set_option('fuse', 31)
    assert counts[0] - counts[31] >= 26, counts


def test_untiled_4k_frame_only_f43_layers_read_the_buffers_above_4gib():
    """An untiled 3840x2240 pair has 4.4-5 GB level-0 buffers.  Only conv_wino43_kernel (addresses relative to a workgroup's
    own halo rows) and the 64-bit-pointer kernels may read them; everything behind a 32-bit whole-buffer offset
    (`offset32_buffer_bytes`, what film_forward checks and chunks batches by) stays below 4 GiB."""
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED
    eng = FilmEngine(PUBLISHED, device=-1)
    plan = eng.plan(1, 2240, 3840)
    size = {b['name']: b['floats'] * 4 for b in plan['buffers']}
    lim = 0xFFF00000
    assert max(size.values()) > 2 ** 32 and plan['offset32_buffer_bytes'] < lim
    big_readers = 0
    for op in plan['ops']:
        if op['kind'] != 'conv_mfma':
            continue
        for sg in op['segs']:
            if size[sg['v']['buf']] > lim:
                assert op['wino'] in (3, 4), (op['tag'], sg['v']['buf'])     # conv_wino43_kernel / conv_wino2d_kernel: patch-relative offsets
                big_readers += 1
            elif op['wino'] not in (3, 4):
                assert size[sg['v']['buf']] <= plan['offset32_buffer_bytes']
    assert big_readers >= 5
    # 8K untiled: a direct-convolution layer would have to read more than 4 GiB -> refused (tile it)
    assert eng.plan(1, 4352, 7680)['offset32_buffer_bytes'] > lim
    eng.close()


def test_tune_cache_text_round_trip_and_validation(tiny_weights):
    """film_export_tune / film_import_tune (autotune choices across processes): text with a version header; another
    version's text is ignored, malformed lines are refused, imported entries come back out."""
    from film_hip.engine import FilmEngine, FilmError, load_library
    from film_hip.options import TINY
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    head = eng.export_tune()
    assert head == '# film_hip tune cache v1 ' + load_library().film_version().decode() + '\n'
    text = head + '2x64x64:16:3:16:0:0:0:0:0:1:0:0|16,16,0,0\t3\n' + '4x32x32:8:3:8:0:0:0:0:0:1:0:0|8,8,0,0\t19\n'
    eng.import_tune(text)
    assert eng.export_tune() == head + ''.join(sorted(text.splitlines(True)[1:]))
    eng.import_tune('# film_hip tune cache v1 some-other-build\nabc\t5\n')      # ignored, not an error
    assert 'abc' not in eng.export_tune()
    for bad in (head + 'no tab here\n', head + 'sig\tnot-a-number\n', head + 'sig\t-4\n'):
        with pytest.raises(FilmError):
            eng.import_tune(bad)
    eng.close()


@prefers_extra_families
def test_second_weight_set_repacks_every_layout_group_a_cached_plan_reads(tiny_weights):
    """A handle whose cached plans pulled in the on-demand layout groups (Planner::need_groups: F(2,3), halo copies) gets a
    SECOND weight set (film_finalize via set_weights / film_import_packed): every group that was packed before must be packed
    again from the new tensors - the cached plan's ops keep pointing into those regions - so the layout blob has the same
    extent and the same bytes as a fresh handle that took the new set first and then built the same plan (round-2 ADVICE)."""
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    w2 = W.make_synthetic_weights(TINY, seed=11)
    eng = FilmEngine(TINY, device=-1)
    eng.set_weights(tiny_weights)
    eng.set_option('wino2d', 0)                         # (the nested kernel reads group 0: without it the levels of this frame need F(2,3) / halo copies)
    base = eng.export_layouts().size
    kinds = {(op['halo'], op['wino']) for op in eng.plan(1, 128, 96)['ops'] if op['kind'] == 'conv_mfma'}
    from conftest import has_extra_families
    if not has_extra_families():                        # default library: no plan selects those families; pull the groups in by option
        eng.set_option('pack_groups', 3)
    grown = eng.export_layouts().size
    assert grown > base, ('the test needs a plan that reads an on-demand layout group', kinds)
    eng.import_packed(eng.export_packed())              # same weights again: extent and bytes unchanged
    assert eng.export_layouts().size == grown
    before = eng.export_layouts().copy()
    eng.set_weights(w2)                                 # different weights on the used handle
    after = eng.export_layouts()
    assert after.size == grown and not np.array_equal(after, before)
    fresh = FilmEngine(TINY, device=-1)
    fresh.set_weights(w2)
    fresh.set_option('wino2d', 0)
    fresh.plan(1, 128, 96)
    if not has_extra_families():
        fresh.set_option('pack_groups', 3)
    assert np.array_equal(after, fresh.export_layouts())


def test_nested_winograd_rule_is_a_function_of_layer_and_level_size():
    """conv_wino2d_kernel runs every 3x3 layer whose channels come in sixteens / thirty-twos (round 4: K = 32 ... 2448, pooled stages
    and the RGB-head layer included - its epilogue fuses both) on levels with >= 1536 pixels and on smaller ones (>= 256) that fill 65 % of
    their 8 x 32 tiles (32x32, 16x28, 18x30: round 6) - whatever the batch size, small frames too.  Other levels and the 3-channel first layers never get it; "w2d_shape" forces one tile shape where it fits and
    is validated."""
    from film_hip.engine import FilmEngine, FilmError
    from film_hip.options import PUBLISHED
    eng = FilmEngine(PUBLISHED, device=-1)

    def nested(b, h, w):
        return sorted((o['tag'], o['H'], o['W']) for o in eng.plan(b, h, w)['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4)

    for (h, w) in ((256, 256), (256, 448), (576, 960)):
        one = nested(1, h, w)
        assert one == nested(3, h, w)                                  # never a function of the batch
        for tag, hh, ww in one:
            assert _w2d_level(hh, ww), (tag, hh, ww)
        # ... and the other way round: every 3x3 layer of such a level that is not a first layer runs it
        for o in eng.plan(1, h, w)['ops']:
            if o['kind'] == 'conv_mfma' and o['ksize'] == 3 and not o.get('c3') and _w2d_level(o['H'], o['W']):
                assert o['wino'] == 4, o['tag']
    small = [t for t, _, _ in nested(1, 256, 256)]
    assert any('flow_predictor_1/conv_0' in t for t in small) and any('flow_predictor_0/conv_0' in t for t in small), small
    big = [t for t, _, _ in nested(4, 576, 960)]
    assert len(big) == 56 and sum('+pool' in t for t in big) == 15 and any(t.endswith('convs_0_2+output_conv') for t in big), big
    for o in eng.plan(1, 256, 448)['ops']:
        if o['kind'] == 'conv_mfma' and o['wino'] == 4:
            assert o['Ctot'] % 16 == 0 and o['Cout'] % 32 == 0 and o['w2d_off'] >= 0, o['tag']
    # the tile knob: a shape a nested layer can run is taken (32-channel tile = 1; the layer with the fused RGB head needs all 64
    # channels of a pixel in one workgroup and keeps tile 0); out-of-range values are refused
    eng.set_option('w2d_shape', 1)
    tiles = {(o['tile'], '+output_conv' in o['tag']) for o in eng.plan(1, 256, 448)['ops'] if o['kind'] == 'conv_mfma' and o['wino'] == 4}
    assert tiles == {(1 | 8192 | 16, False), (0 | 8192 | 16, True)}, tiles
    eng.set_option('w2d_shape', -1)
    with pytest.raises(FilmError):
        eng.set_option('w2d_shape', 99)
