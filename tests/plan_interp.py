"""Test-only numpy interpreter of the engine's execution plan (film_plan_json).

The HIP engine plans a forward pass as a list of kernel launches over one workspace arena.  This
module executes that same op list on a numpy arena, giving every op the semantics its HIP kernel
implements (frame-interpolation_amd/csrc/*.hip).  It lets the CPU test-suite validate the planner
(buffer layout, concat-by-slices, batch remaps, upsample folding) and the weight packer
(channel permutation / zero padding) against the oracle WITHOUT a GPU.  It is test infrastructure:
nothing under frame-interpolation_amd/ imports it.
"""
from __future__ import annotations

import zlib

import numpy as np

from oracle import film_oracle as fo


_VERIFIED = set()


def _view(arena: np.ndarray, v: dict, nb: int, h: int, w: int) -> np.ndarray:
    s = v['stride']
    base = arena[v['off']:]
    need = ((nb * h * w - 1) * s + v['C'])
    assert need <= base.size, 'view exceeds arena'
    return np.lib.stride_tricks.as_strided(
        base, shape=(nb, h, w, v['C']), strides=(h * w * s * 4, w * s * 4, s * 4, 4), writeable=True)


def run_plan(plan: dict, packed: np.ndarray, x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    """Executes the plan; returns the arena (use `tap` to read named buffers)."""
    arena = np.zeros(plan['arena_floats'], dtype=np.float32)
    bufs = {b['name']: b for b in plan['buffers']}
    B = plan['B']
    # key of the "weight copies verified" cache: the CONTENT of the blob (an id() can be recycled by a re-packed blob of the same size)
    blob_key = zlib.crc32(np.ascontiguousarray(packed).view(np.uint8))
    img0 = bufs['img0']
    n = x0.size
    arena[img0['off']:img0['off'] + n] = x0.ravel()
    arena[img0['off'] + n:img0['off'] + 2 * n] = x1.ravel()
    for op in plan['ops']:
        k = op['kind']
        nb, h, w = op['NB'], op['H'], op['W']
        if k == 'conv_mfma' and op.get('c3'):
            # first-layer mode: 3-channel image input, weights packed [12 tap slots][4][Cout]
            x = np.ascontiguousarray(_view(arena, op['segs'][0]['v'], nb, h, w))
            co = op['Cout']
            w48 = packed[op['w_off']:op['w_off'] + 48 * co].reshape(12, 4, co)
            assert not w48[9:].any() and not w48[:, 3].any(), 'padding rows of the C3 pack must be zero'
            wt = np.ascontiguousarray(w48[:9, :3]).reshape(3, 3, 3, co)
            bias = packed[op['b_off']:op['b_off'] + co]
            _view(arena, op['out'], nb, h, w)[...] = fo.conv2d_same(x, wt, bias, 'leaky' if op['leaky'] else None)
        elif k == 'conv_mfma' and op.get('fold'):
            # nearest-x2 + 2x2 conv as four sub-pixel phases on the low-resolution grid (H, W); output 2H x 2W.
            # fold == 2: all phases in one op, phase q = py*2 + px, taps (a, b), a <= py, b <= px, raster order
            sg = op['segs'][0]
            assert op['fold'] in (2, 3) and len(op['segs']) == 1 and not sg['up'] and not sg['bmod'] and not op['leaky']
            x = np.ascontiguousarray(_view(arena, sg['v'], nb, h, w))
            ct, co = op['Ctot'], op['Cout']
            outv = _view(arena, op['out'], nb, 2 * h, 2 * w)
            if op['fold'] == 3:
                # conv_fold4_kernel: the difference form.  Weights [Cout/32][chunk8][plane 4][K half][32][4], planes S, Sx, Sy, W11;
                # G0 = S.I, G1 = Sx.Dx, G2 = Sy.Dy, G3 = W11.Dxy with Dx = I - I(x+1), Dy = I - I(y+1), Dxy = Dx - (I(y+1) - I(y+1,x+1))
                # (zero beyond the bottom / right edge); out = G0, G0 - G1, G0 - G2, ((G0 - G1) - G2) + G3 (+ bias)
                assert op['w_off'] == op['wf4_off'] and ct % 16 == 0 and co % 32 == 0
                w4 = packed[op['w_off']:op['w_off'] + 4 * ct * co].reshape(co // 32, ct // 8, 4, 2, 32, 4)
                w4 = w4.transpose(2, 1, 3, 5, 0, 4).reshape(4, ct, co)            # [plane][c = chunk*8 + half*4 + j][n = tile*32 + lane]
                sh = lambda a, b: np.pad(x[:, a:, b:], ((0, 0), (0, a), (0, b), (0, 0)))     # noqa: E731
                i00, i01, i10, i11 = x, sh(0, 1), sh(1, 0), sh(1, 1)
                dx = i00 - i01
                planes = [i00, dx, i00 - i10, dx - (i10 - i11)]
                g = [(pl.reshape(-1, ct) @ w4[q]).reshape(nb, h, w, co) for q, pl in enumerate(planes)]
                bias = packed[op['b_off']:op['b_off'] + co]
                outv[:, 0::2, 0::2] = g[0] + bias
                outv[:, 0::2, 1::2] = (g[0] - g[1]) + bias
                outv[:, 1::2, 0::2] = (g[0] - g[2]) + bias
                outv[:, 1::2, 1::2] = (((g[0] - g[1]) - g[2]) + g[3]) + bias
                continue
            wfx = None
            if op.get('wfx_off', -1) >= 0:
                # bf16x3 copy for conv_foldx3_kernel: [Cout][chunk16][9 (tap, phase) steps][plane][16] bf16
                raw = packed[op['wfx_off']:op['wfx_off'] + 9 * ct * co].view(np.uint16)
                wfx = (raw.astype(np.uint32) << 16).view(np.float32).reshape(co, ct // 16, 9, 2, 16).astype(np.float64).sum(axis=3)
                fold_step = {(0, 0): 0, (0, 1): 1, (0, 2): 2, (0, 3): 3, (1, 1): 4, (1, 3): 5, (2, 2): 6, (2, 3): 7, (3, 3): 8}
            for q in range(4):
                py, px = q >> 1, q & 1
                taps = [(a, b) for a in range(py + 1) for b in range(px + 1)]
                off = op['w_off'] + op['fold_woff'][q]
                wt = packed[off:off + len(taps) * ct * co].reshape(co, len(taps), ct)
                acc = np.zeros((nb, h, w, co), np.float32)
                for t, (a, b) in enumerate(taps):
                    if wfx is not None:
                        got = wfx[:, :, fold_step[(a * 2 + b, q)], :].reshape(co, ct)
                        want = wt[:, t].astype(np.float64)
                        assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -17), 'bf16x3 fold weight copy differs'
                    sh = np.zeros_like(x)
                    sh[:, :h - a, :w - b] = x[:, a:, b:]          # zero beyond the bottom / right edge
                    acc += (sh.reshape(-1, ct) @ wt[:, t].T).reshape(nb, h, w, co)
                acc += packed[op['b_off']:op['b_off'] + co]
                outv[:, py::2, px::2] = acc
        elif k == 'conv_mfma':
            parts = []
            for sg in op['segs']:
                hs, ws = (h // 2, w // 2) if sg['up'] else (h, w)
                if sg['bmod']:
                    src = _view(arena, sg['v'], sg['bmod'], hs, ws)
                    idx = (np.arange(nb) + sg['boff']) % sg['bmod']
                    src = src[idx]
                else:
                    src = _view(arena, sg['v'], nb, hs, ws)
                if sg['up']:
                    src = np.repeat(np.repeat(src, 2, axis=1), 2, axis=2)
                parts.append(src)
            x = np.ascontiguousarray(np.concatenate(parts, axis=-1))
            ks, ct, co = op['ksize'], op['Ctot'], op['Cout']
            assert x.shape[-1] == ct
            # MFMA-conv layers are packed K-major: [Cout][tap][Ctot]
            wt = packed[op['w_off']:op['w_off'] + ks * ks * ct * co].reshape(co, ks, ks, ct)
            wt = np.ascontiguousarray(wt.transpose(1, 2, 3, 0))
            # the layer's other weight copies are verified once per (layout blob, layer): the shared sub-extractor / flow
            # predictor layers appear in many ops, and several plans are run over one blob
            vkey = (blob_key, packed.size, op['w_off'], op.get('wh_off', -1), op.get('ww_off', -1), op.get('w2d_off', -1), op.get('ws_off', -1), op.get('wx_off', -1))
            check = vkey not in _VERIFIED
            _VERIFIED.add(vkey)
            if check and op.get('wh_off', -1) >= 0:
                # the layer's second copy for conv_halo_kernel, [Cout][chunk][tap][16], must hold the same weights
                wh = packed[op['wh_off']:op['wh_off'] + 9 * ct * co].reshape(co, ct // 16, 3, 3, 16)
                wh = wh.transpose(2, 3, 1, 4, 0).reshape(3, 3, ct, co)
                assert np.array_equal(wh, wt), 'halo weight copy differs'
                if op.get('halo') or op.get('split'):
                    assert ks == 3 and not any(sg['up'] for sg in op['segs'])
            if check and op.get('ww_off', -1) >= 0:
                # Winograd copy [Cout][chunk of 8][nu*3+dy][8]: u0 = g0, u1 = ((g0+g2)+g1)/2, u2 = ((g0+g2)-g1)/2, u3 = g2
                ww = packed[op['ww_off']:op['ww_off'] + 12 * ct * co].reshape(co, ct // 8, 4, 3, 8)
                ww = ww.transpose(2, 3, 1, 4, 0).reshape(4, 3, ct, co)      # [nu][dy][c][n]
                g0, g1, g2 = wt[:, 0], wt[:, 1], wt[:, 2]                    # [dy][c][n]
                half = np.float32(0.5)
                want_u = np.stack([g0, ((g0 + g2) + g1) * half, ((g0 + g2) - g1) * half, g2])
                assert np.array_equal(ww, want_u), 'Winograd weight copy differs'
                if op.get('w43_off', -1) >= 0:
                    # F(4,3) copy for conv_wino43_kernel: [Cout][chunk8][dy][nu 6][8], same float32 operation order as the packer
                    w43 = packed[op['w43_off']:op['w43_off'] + 18 * ct * co].reshape(co, ct // 8, 3, 6, 8)
                    w43 = w43.transpose(3, 2, 1, 4, 0).reshape(6, 3, ct, co)      # [nu][dy][c][n]
                    f = np.float32
                    c6, c12, c24 = f(1) / f(6), f(1) / f(12), f(1) / f(24)
                    e, o = g0 * c24 + g2 * c6, g1 * c12
                    want43 = np.stack([g0 * f(0.25), -((g0 + g2) + g1) * c6, -((g0 + g2) - g1) * c6, e + o, e - o, g2])
                    assert np.array_equal(w43, want43), 'F(4,3) weight copy differs'
                if op.get('w2d_off', -1) >= 0:
                    # nested copy for conv_wino2d_kernel: [Cout/32][chunk8][mu 4][nu 6][K half][32][4]; U = F(2,3) along dy of the
                    # F(4,3)-transformed rows (want43[nu][dy]), same float32 operation order as the packer
                    w2d = packed[op['w2d_off']:op['w2d_off'] + 24 * ct * co].reshape(co // 32, ct // 8, 4, 6, 2, 32, 4)
                    w2d = w2d.transpose(2, 3, 1, 4, 6, 0, 5).reshape(4, 6, ct, co)      # [mu][nu][c = chunk*8 + half*4 + j][n = tile*32 + lane]
                    u0, u1, u2 = want43[:, 0], want43[:, 1], want43[:, 2]               # [nu][c][n] per dy
                    want2d = np.stack([u0, ((u0 + u2) + u1) * half, ((u0 + u2) - u1) * half, u2])
                    assert np.array_equal(w2d, want2d), 'nested Winograd weight copy differs'
                if op.get('wx_off', -1) >= 0:
                    # bf16x3 copy of the transformed weights: [Cout][chunk16][dy][j][h][plane][16] bf16, nu = 2h + j,
                    # hi + mid within 2^-17 of the fp32 value (nearest split)
                    n16 = 12 * ct * co * 2
                    raw = packed[op['wx_off']:op['wx_off'] + n16 // 2].view(np.uint16)
                    pl = (raw.astype(np.uint32) << 16).view(np.float32).reshape(co, ct // 16, 3, 2, 2, 2, 16)
                    got_u = pl.astype(np.float64).sum(axis=5)                      # [n][chunk][dy][j][h][16]
                    got_u = got_u.transpose(4, 3, 2, 1, 5, 0).reshape(2, 2, 3, ct, co)   # [h][j][dy][c][n]
                    got_u = got_u.reshape(4, 3, ct, co)                            # nu = 2h + j
                    assert np.all(np.abs(got_u - want_u) <= np.abs(want_u.astype(np.float64)) * 2.0 ** -17), 'bf16x3 Winograd weight copy differs'
            if check and op.get('ws_off', -1) >= 0:
                # bf16x6 copy: three bf16 planes [Cout][chunk][tap][plane][16] that add up to the weight EXACTLY
                n16 = 9 * ct * co * 3
                raw = packed[op['ws_off']:op['ws_off'] + (n16 + 1) // 2].view(np.uint16)[:n16]
                planes = (raw.astype(np.uint32) << 16).view(np.float32).reshape(co, ct // 16, 3, 3, 3, 16)
                ws = planes.astype(np.float64).sum(axis=4).transpose(2, 3, 1, 4, 0).reshape(3, 3, ct, co)
                assert np.array_equal(ws.astype(np.float32), wt) and np.array_equal(ws, wt.astype(np.float64)), 'bf16x6 split is not exact'
                # round-to-nearest pieces: the two planes bf16x3 uses are within 2^-17 of the weight
                two = planes.astype(np.float64)[:, :, :, :, :2].sum(axis=4).transpose(2, 3, 1, 4, 0).reshape(3, 3, ct, co)
                assert np.all(np.abs(two - wt) <= np.abs(wt.astype(np.float64)) * 2.0 ** -17), 'hi + mid is not a nearest split'
            bias = packed[op['b_off']:op['b_off'] + co]
            y = fo.conv2d_same(x, wt, bias, 'leaky' if op['leaky'] else None)
            if op.get('pw_out', {}).get('buf'):     # fused 1x1 convolution (the RGB head): `out` is NOT written
                assert op.get('wino') in (3, 4) and co == 64 and op.get('ksplit', 1) <= 1 and not op.get('out2', {}).get('buf')
                c2 = op['pw_cout']
                w2 = packed[op['w2_off']:op['w2_off'] + co * c2].reshape(1, 1, co, c2)
                b2 = packed[op['b2_off']:op['b2_off'] + c2]
                _view(arena, op['pw_out'], nb, h, w)[...] = fo.conv2d_same(y, w2, b2, None)
                continue
            _view(arena, op['out'], nb, h, w)[...] = y
            if op.get('out2', {}).get('buf'):       # fused AveragePooling2D(2, 2) of the output
                assert op.get('wino') in (3, 4) and h % 2 == 0 and w % 2 == 0
                _view(arena, op['out2'], nb, h // 2, w // 2)[...] = fo.avg_pool2x2(y)
        elif k == 'flow_head':
            m = op['n']
            x = np.ascontiguousarray(_view(arena, op['in'], 1, 1, m))
            ci = op['Ctot']
            w3 = packed[op['w_off']:op['w_off'] + ci * 16].reshape(1, 1, ci, 16)
            b3 = packed[op['b_off']:op['b_off'] + 16]
            w4 = packed[op['w2_off']:op['w2_off'] + 32].reshape(1, 1, 16, 2)
            b4 = packed[op['b2_off']:op['b2_off'] + 2]
            hid = fo.conv2d_same(x, w3, b3, 'leaky')
            _view(arena, op['out'], 1, 1, m)[...] = fo.conv2d_same(hid, w4, b4, None)
            if op.get('out2', {}).get('buf'):       # fused v = residual + upsampled flow
                _view(arena, op['out2'], 1, 1, m)[...] = _view(arena, op['out'], 1, 1, m) + _view(arena, op['in2'], 1, 1, m)
        elif k == 'conv_pw':
            m = op['n']
            x = np.ascontiguousarray(_view(arena, op['in'], 1, 1, m))
            ci, co = op['Ctot'], op['Cout']
            wt = packed[op['w_off']:op['w_off'] + ci * co].reshape(1, 1, ci, co)
            bias = packed[op['b_off']:op['b_off'] + co]
            _view(arena, op['out'], 1, 1, m)[...] = fo.conv2d_same(x, wt, bias, 'leaky' if op['leaky'] else None)
            if op.get('out2', {}).get('buf'):       # flow head: fused v = residual + upsampled flow
                assert co == 2
                _view(arena, op['out2'], 1, 1, m)[...] = _view(arena, op['out'], 1, 1, m) + _view(arena, op['in2'], 1, 1, m)
        elif k == 'pool':
            x = np.ascontiguousarray(_view(arena, op['in'], nb, h, w))
            _view(arena, op['out'], nb, h // 2, w // 2)[...] = fo.avg_pool2x2(x)
        elif k == 'flow_up':
            x = np.ascontiguousarray(_view(arena, op['in'], nb, h, w))
            _view(arena, op['out'], nb, 2 * h, 2 * w)[...] = fo.resize_bilinear(np.float32(2) * x, (2 * h, 2 * w))
        elif k == 'flow_add':
            m = op['n'] // 2
            a = _view(arena, op['in'], 1, 1, m)
            b = _view(arena, op['in2'], 1, 1, m)
            _view(arena, op['out'], 1, 1, m)[...] = a + b
        elif k == 'warp':
            # one launch may carry both directions / both images of a level: the source (flow) batch of output batch n is
            # (n + src_brot) % nb ((n + flow_brot) % nb)
            src = np.roll(np.ascontiguousarray(_view(arena, op['in'], nb, h, w)), -op.get('src_brot', 0), axis=0)
            if op.get('in3', {}).get('buf'):        # fused tf.image.resize(2 * v) of the coarser level, stored to out2
                assert not op['in2']['buf'] and op['out2']['buf'] and not op.get('flow_brot', 0)
                coarse = np.ascontiguousarray(_view(arena, op['in3'], nb, h // 2, w // 2))
                flow = fo.resize_bilinear(np.float32(2) * coarse, (h, w))
                _view(arena, op['out2'], nb, h, w)[...] = flow
            else:
                flow = np.roll(np.ascontiguousarray(_view(arena, op['in2'], nb, h, w)), -op.get('flow_brot', 0), axis=0)
            _view(arena, op['out'], nb, h, w)[...] = fo.warp(src, np.float32(op['fscale']) * flow)
            if op.get('img_out', {}).get('buf'):    # fused sixteen miscellaneous channels of the aligned level
                mb = op.get('misc_nb', 0) or nb
                ims = np.ascontiguousarray(_view(arena, op['img_in'], 2 * mb, h, w))
                bf = np.ascontiguousarray(_view(arena, op['pack_b'], mb, h, w))
                ff = np.ascontiguousarray(_view(arena, op['pack_f'], mb, h, w))
                out = _view(arena, op['img_out'], mb, h, w)
                assert op['img_out']['C'] == 16 and op['fscale'] == 0.5
                out[..., 0:3] = fo.warp(ims[:mb], np.float32(0.5) * bf)    # image 0 <- backward flow
                out[..., 3:6] = fo.warp(ims[mb:], np.float32(0.5) * ff)    # image 1 <- forward flow
                out[..., 6:8] = bf * np.float32(0.5)
                out[..., 8:10] = ff * np.float32(0.5)
                out[..., 10:16] = 0
        elif k == 'pack_flow':
            m = op['n']
            bf = _view(arena, op['in'], 1, 1, m)
            ff = _view(arena, op['in2'], 1, 1, m)
            out = _view(arena, op['out'], 1, 1, m)
            out[..., 0:2] = bf * np.float32(0.5)
            out[..., 2:4] = ff * np.float32(0.5)
            out[..., 4:10] = 0
        else:
            raise ValueError(k)
    return arena


def tap(plan: dict, arena: np.ndarray, name: str) -> np.ndarray:
    b = next(x for x in plan['buffers'] if x['name'] == name)
    raw = arena[b['off']:b['off'] + b['floats']]
    if not b.get('planar'):
        return raw.reshape(b['N'], b['H'], b['W'], b['C']).copy()
    # three pixel-major planes (aligned-pyramid levels) -> [N, H, W, C], as film_get_tap returns them
    npix, parts, base = b['N'] * b['H'] * b['W'], [], 0
    for c in (b['planar'], b['planar'], b['C'] - 2 * b['planar']):
        parts.append(raw[base:base + npix * c].reshape(b['N'], b['H'], b['W'], c))
        base += npix * c
    return np.concatenate(parts, axis=-1)


# ----------------------------------------------------------------------------------------------
# Conversions between the engine's internal layouts and the reference's tensors
# ----------------------------------------------------------------------------------------------
def split_pair(x: np.ndarray, B: int):
    """[2B,...] batch (n = s*B + b) -> (first half, second half)."""
    return x[:B], x[B:]


def aligned_to_reference(a: np.ndarray, C: int) -> np.ndarray:
    """internal [feat0 C | feat1 C | img0 3 | img1 3 | bflow 2 | fflow 2 | 0x6] ->
    reference [img0 | feat0 | img1 | feat1 | bflow | fflow] (interpolator.py:167-183)."""
    f0, f1 = a[..., :C], a[..., C:2 * C]
    m = a[..., 2 * C:]
    return np.concatenate([m[..., 0:3], f0, m[..., 3:6], f1, m[..., 6:8], m[..., 8:10]], axis=-1)
