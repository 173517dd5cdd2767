"""CPU checks of the arithmetic identities the HIP kernels are built on (numpy restatements, no GPU):

* the 1-D Winograd transforms F(2,3) (conv_wino_impl.h / conv_winox3_impl.h) and F(4,3) (conv_wino43_impl.h) in the exact
  operation order of the kernels / the weight packer, against the direct 3-tap correlation in float64;
* the round-to-nearest-even bf16 split of conv_split_impl.h: hi + mid + lo == x exactly, |x - hi - mid| <= 2^-17 |x|,
  and the bf16x3 product hi*hi + hi*mid + mid*hi within 2^-15 of the exact product (4.4e-6 rms) with (near) zero mean error;
* the sub-pixel fold of nearest-x2 upsample + 2x2 'same' convolution (film_planner.cpp / film_layers.cpp / conv_foldx3_impl.h).
"""
import numpy as np

f32 = np.float32


def bf16_rne(x):
    """float32 -> bfloat16 (returned as float32), round to nearest even: film_layers.cpp bf16_rne / v_cvt_pk_bf16_f32."""
    u = np.asarray(x, f32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(f32)


def test_bf16_nearest_split_is_exact_and_tight():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(f32) * f32(3.0),
                        (rng.random(50000, dtype=f32) * f32(1e-3)),
                        np.array([0.0, 1.0, -1.0, 0.1, 255.0 / 256.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 3.0e38, 1e-30], f32)])
    hi = bf16_rne(x)
    r = x - hi                      # exact in float32
    mid = bf16_rne(r)
    q = r - mid                     # exact in float32
    lo = bf16_rne(q)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    assert np.all(np.abs(q.astype(np.float64)) <= np.abs(x.astype(np.float64)) * 2.0 ** -17)
    # truncation (the first version of the split) is biased, nearest is not: mean signed residue relative to x
    nz = x != 0
    assert abs(np.mean(q[nz].astype(np.float64) / x[nz].astype(np.float64))) < 2.0 ** -24


def test_bf16x3_product_error():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(400000).astype(f32)
    b = (rng.standard_normal(400000) * 0.05).astype(f32)

    def split(x):
        hi = bf16_rne(x)
        return hi.astype(np.float64), bf16_rne(x - hi).astype(np.float64)
    ah, am = split(a)
    bh, bm = split(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    x3 = ah * bh + ah * bm + am * bh
    rel = (x3 - exact) / exact
    # dropped: mid*mid (|mid| <= 2^-8 |x|: <= 2^-16) and the two residue terms (<= 2^-17 each)
    assert np.abs(rel).max() <= 2.0 ** -15 and np.sqrt((rel ** 2).mean()) < 6e-6
    assert abs(rel.mean()) < 2.0 ** -22              # zero mean: a K-long dot product averages it down
    # a K = 4608 dot product: bf16x3 error of the sum vs the float32 chain's own rounding error
    K = 4608
    A, B = a[:K * 64].reshape(64, K), b[:K * 64].reshape(64, K)
    ex = (A.astype(np.float64) * B.astype(np.float64)).sum(1)
    s3 = x3[:K * 64].reshape(64, K).sum(1)
    sf = np.zeros(64, f32)
    for k in range(K):
        sf = sf + A[:, k] * B[:, k]
    scale = np.abs(A.astype(np.float64) * B.astype(np.float64)).sum(1)
    assert np.abs(s3 - ex).max() / scale.max() < 1e-6 and np.abs(sf - ex).max() / scale.max() < 1e-6


def _direct3(d, g):
    return np.array([sum(np.float64(g[k]) * np.float64(d[i + k]) for k in range(3)) for i in range(len(d) - 2)])


def test_winograd_f23_and_f43_transforms():
    rng = np.random.default_rng(2)
    err23, err43 = [], []
    for _ in range(2000):
        g = (rng.standard_normal(3) * 0.1).astype(f32)
        d = rng.standard_normal(6).astype(f32)
        # F(2,3): weights as packed in film_layers.cpp, inputs as in conv_wino_impl.h store_item
        g0, g1, g2 = g
        u = [g0, ((g0 + g2) + g1) * f32(0.5), ((g0 + g2) - g1) * f32(0.5), g2]
        for t in (0, 2):                                   # two pairs out of the six inputs
            d0, d1, d2, d3 = d[t:t + 4]
            v = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
            m = [f32(u[i]) * f32(v[i]) for i in range(4)]
            y = np.array([(m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]], f32)
            err23.append(np.abs(y - _direct3(d[t:t + 4], g)).max())
        # F(4,3): weights as packed in film_layers.cpp, inputs as in conv_wino43_impl.h store_item (fused multiply-adds)
        c6, c12, c24 = f32(1) / f32(6), f32(1) / f32(12), f32(1) / f32(24)
        e, o = g0 * c24 + g2 * c6, g1 * c12
        u = [g0 * f32(0.25), -((g0 + g2) + g1) * c6, -((g0 + g2) - g1) * c6, e + o, e - o, g2]
        fma = lambda a, b, c: f32(np.float64(a) * np.float64(b) + np.float64(c))   # single rounding
        d0, d1, d2, d3, d4, d5 = d
        t1, t2 = fma(f32(-4), d2, d4), fma(f32(-4), d1, d3)
        t3, t4 = d4 - d2, f32(2) * (d3 - d1)
        v = [fma(f32(4), d0, fma(f32(-5), d2, d4)), t1 + t2, t1 - t2, t3 + t4, t3 - t4, fma(f32(4), d1, fma(f32(-5), d3, d5))]
        m = [f32(u[i]) * f32(v[i]) for i in range(6)]
        y = np.array([((m[0] + m[1]) + m[2]) + (m[3] + m[4]), (m[1] - m[2]) + f32(2) * (m[3] - m[4]),
                      (m[1] + m[2]) + f32(4) * (m[3] + m[4]), (m[1] - m[2]) + (f32(8) * (m[3] - m[4]) + m[5])], f32)
        err43.append(np.abs(y - _direct3(d, g)).max())
    # single products of O(0.1 * 1): float32 epsilon-level errors; F(4,3) amplifies them by its larger constants
    assert max(err23) < 2e-6 and max(err43) < 2e-5
    assert np.mean(err43) < 20 * np.mean(err23) + 1e-7


def test_subpixel_fold_of_upsample_and_2x2_conv():
    """nearest-x2 upsample + 2x2 'same' conv (pad bottom/right) == four phase convolutions on the low-resolution input with
    the weights of the kernel taps that read the same pixel summed: phase (py, px), tap (a, b), a <= py, b <= px."""
    rng = np.random.default_rng(3)
    h, w, ci, co = 5, 7, 3, 4
    x = rng.standard_normal((h, w, ci))
    k = rng.standard_normal((2, 2, ci, co))
    up = np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)
    pad = np.zeros((2 * h + 1, 2 * w + 1, ci))
    pad[:2 * h, :2 * w] = up
    ref = np.zeros((2 * h, 2 * w, co))
    for dy in range(2):
        for dx in range(2):
            ref += pad[dy:dy + 2 * h, dx:dx + 2 * w] @ k[dy, dx]
    xp = np.zeros((h + 1, w + 1, ci))
    xp[:h, :w] = x
    got = np.zeros_like(ref)
    steps = 0
    for py in range(2):
        for px in range(2):
            for a in range(py + 1):
                for b in range(px + 1):
                    wsum = sum(k[dy, dx] for dy in range(2) for dx in range(2) if (py & dy) == a and (px & dx) == b)
                    got[py::2, px::2] += xp[a:a + h, b:b + w] @ wsum
                    steps += 1
    assert steps == 9 and np.abs(got - ref).max() < 1e-12


def test_difference_form_of_upsample_and_2x2_conv():
    """conv_fold4_kernel's identity (conv_fold4_impl.h): with I = 0 beyond the bottom / right edge, Dx = I - I(x+1), Dy = I - I(y+1),
    Dxy = Dx - (I(y+1) - I(y+1, x+1)) and the weight sums S = W00 + W01 + W10 + W11, Sx = W01 + W11, Sy = W10 + W11:
    out(2y, 2x) = S.I, out(2y, 2x+1) = S.I - Sx.Dx, out(2y+1, 2x) = S.I - Sy.Dy, out(2y+1, 2x+1) = S.I - Sx.Dx - Sy.Dy + W11.Dxy -
    four products per low-resolution pixel for the reference op's sixteen (fusion.py:133-135) - in float64 exactly, and in float32
    within the rounding of a sum of that length."""
    rng = np.random.default_rng(5)
    h, w, ci, co = 6, 9, 24, 5
    x = rng.standard_normal((h, w, ci))
    k = rng.standard_normal((2, 2, ci, co)) * 0.2
    up = np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)
    pad = np.zeros((2 * h + 1, 2 * w + 1, ci))
    pad[:2 * h, :2 * w] = up
    ref = np.zeros((2 * h, 2 * w, co))
    for dy in range(2):
        for dx in range(2):
            ref += pad[dy:dy + 2 * h, dx:dx + 2 * w] @ k[dy, dx]
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        xp = np.zeros((h + 1, w + 1, ci), dt)
        xp[:h, :w] = x
        kk = k.astype(dt)
        i00, i01, i10, i11 = xp[:h, :w], xp[:h, 1:], xp[1:, :w], xp[1:, 1:]
        dxp = i00 - i01
        planes = (i00, dxp, i00 - i10, dxp - (i10 - i11))
        wsum = (((kk[0, 0] + kk[0, 1]) + kk[1, 0]) + kk[1, 1], kk[0, 1] + kk[1, 1], kk[1, 0] + kk[1, 1], kk[1, 1])
        g = [p @ ws for p, ws in zip(planes, wsum)]
        assert all(v.dtype == dt for v in g)
        got = np.zeros((2 * h, 2 * w, co), dt)
        got[0::2, 0::2] = g[0]
        got[0::2, 1::2] = g[0] - g[1]
        got[1::2, 0::2] = g[0] - g[2]
        got[1::2, 1::2] = ((g[0] - g[1]) - g[2]) + g[3]
        assert np.abs(got - ref).max() < tol, (dt, np.abs(got - ref).max())

