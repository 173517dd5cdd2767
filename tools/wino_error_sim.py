#!/usr/bin/env python
"""fp32 round-off of the Winograd forms of a 3x3 convolution, simulated on the CPU (numpy float32 arithmetic in the kernels' order
of operations: transform, K-ordered fp32 accumulation per (mu, nu) plane, inverse), against the float64 direct convolution:

  1-D F(4,3) along x                      (conv_wino43_kernel:   4.5  multiplies per output)
  nested F(4,3)x x F(2,3)y                (conv_wino2d_kernel:   3)
  F(4,3)x x F(4,3)y                       (not built:            2.25)

  python tools/wino_error_sim.py [K] [Cout]     default K = 2448 (decoder level 3), Cout = 8

Activations ~ N(0,1) after a leaky ReLU, weights Glorot-scaled as in film_hip/weights.py, so the outputs are O(1).  Prints max and
rms error of each form.  A planning aid for the next kernel, not a test."""
import sys
import numpy as np

f32 = np.float32
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
I3 = np.eye(3)


def acc32(v, u):
    """sum_k v[k] * u[k] in fp32, k-ordered fma chain (what one MFMA accumulator lane does, up to the 2-wide K grouping)"""
    a = np.zeros(v.shape[1:], dtype=f32) if v.ndim > 1 else f32(0)
    for k in range(v.shape[0]):
        a = (a.astype(np.float64) + v[k].astype(np.float64) * u[k].astype(np.float64)).astype(f32)   # fma: one rounding
    return a


def run(K, Cout, seed=0):
    rng = np.random.default_rng(seed)
    th, tw = 4, 4                                     # one 4x4 output tile (covers F(2,3)y twice, F(4,3)y once)
    x = rng.standard_normal((K, th + 2, tw + 2))
    x = np.where(x > 0, x, 0.2 * x)
    w = rng.standard_normal((Cout, K, 3, 3)) * np.sqrt(2.0 / (9 * K))
    ref = np.zeros((Cout, th, tw))
    for dy in range(3):
        for dx in range(3):
            ref += np.einsum('ok,kyx->oyx', w[:, :, dy, dx], x[:, dy:dy + th, dx:dx + tw])
    x32, w32 = x.astype(f32), w.astype(f32)
    out = {}
    forms = {'1-D F(4,3)x (direct in y)': (I3, I3, None, BT4, G4, AT4),
             'nested F(4,3)x x F(2,3)y': (BT2, G2, AT2, BT4, G4, AT4),
             'F(4,3)x x F(4,3)y': (BT4, G4, AT4, BT4, G4, AT4)}
    for name, (bty, gy, aty, btx, gx, atx) in forms.items():
        res = np.zeros((Cout, th, tw), dtype=np.float64)
        if aty is None:   # rows handled directly: three x-transformed rows per output row, 18 planes summed in (dy, k) order like the 1-D kernel
            for o in range(Cout):
                for y in range(th):
                    m = np.zeros(6, dtype=f32)
                    for dy in range(3):
                        v = (btx @ x32[:, y + dy, :].astype(np.float64).T).astype(f32)                # [6][K]   (transform in fp32 registers)
                        u = (gx @ w32[o, :, dy, :].astype(np.float64).T).astype(f32)                  # [6][K]
                        for nu in range(6):
                            a = m[nu]
                            for k in range(K):
                                a = f32(np.float64(a) + np.float64(v[nu, k]) * np.float64(u[nu, k]))
                            m[nu] = a
                    res[o, y] = (atx @ m.astype(np.float64)).astype(f32)
        else:
            ny = aty.shape[0]                          # output rows per y tile
            for ty in range(th // ny):
                d = x32[:, ty * ny: ty * ny + bty.shape[1], :]
                v = np.einsum('ar,krc,bc->kab', bty, d.astype(np.float64), btx).astype(f32)          # [K][mu][nu]
                for o in range(Cout):
                    u = np.einsum('ar,krc,bc->kab', gy, w32[o].astype(np.float64), gx).astype(f32)   # [K][mu][nu]
                    m = acc32(v, u)                                                                  # [mu][nu]
                    res[o, ty * ny:(ty + 1) * ny] = (aty @ m.astype(np.float64) @ atx.T).astype(f32)
        err = res - ref
        out[name] = (float(np.abs(err).max()), float(np.sqrt((err ** 2).mean())))
    return out, float(np.abs(ref).max())


if __name__ == '__main__':
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2448
    Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out, scale = run(K, Cout)
    print(f'K = {K}, {Cout} output channels, one 4x4 tile; max |output| {scale:.2f}')
    for k, (mx, rms) in out.items():
        print(f'  {k:30s} max|err| {mx:.2e}   rms {rms:.2e}')
