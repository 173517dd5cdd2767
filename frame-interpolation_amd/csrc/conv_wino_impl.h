// conv_wino_impl.h -- 3x3 Conv2D('same') + bias + leaky_relu with the 1-D Winograd transform F(2,3) along x on top
// of the halo-staged implicit GEMM: 12 matrix steps per K chunk instead of 18 for every PAIR of output pixels, i.e.
// 1.5x fewer fp32 MFMAs for the same convolution (fp32 throughout; the rounding differs from the direct sum at the
// 1e-6 level, the dtype does not).
//
// For an output row y and the pixel pair x = 2t, 2t+1 with inputs d0..d3 = in[.][2t-1 .. 2t+2]:
//     v0 = d0 - d2,  v1 = d1 + d2,  v2 = d2 - d1,  v3 = d1 - d3                      (input transform, per input row)
//     u0 = g0,  u1 = (g0 + g1 + g2)/2,  u2 = (g0 - g1 + g2)/2,  u3 = g2               (weights, per (dy, cin, cout), offline)
//     m_nu[y][t][n] = sum_dy sum_c v_nu[y+dy-1][t][c] * u_nu[dy][c][n]                (4 x 3 GEMM steps per chunk)
//     out[y][2t] = (m0 + m1) + m2,   out[y][2t+1] = (m1 - m2) - m3
//
//   * a workgroup owns TH rows x 64 pixels (32 pairs = one 32-row MFMA tile per row) x BN output channels;
//   * K chunks of EIGHT channels: the (TH+2) halo rows of a chunk are transformed ONCE on the way into LDS, image
//     [halo row][nu][pair][8 channels] = 32-byte rows whose two 16-byte K-halves are swapped when bit 3 of the row is
//     set (conflict-free ds_read_b128 / ds_write_b128).  With 16-channel chunks (tools/experiments/
//     conv_wino16_impl.h) the image is 48 KB per stage and only ONE 8-wave workgroup fits a CU; with 8 it is 24 KB and
//     two fit (4 waves per SIMD): +3...14 % (166-192 vs 146-179 TFLOP/s in direct-conv FLOPs);
//   * weights [Cout][chunk][nu*3 + dy][8]; a B stage holds the six (nu, dy) steps of two nu planes, double buffered:
//     one barrier per 6 x 4 MFMAs per wave;
//   * accumulators: 4 (nu) x TN tiles per output row; the output transform runs on them in the epilogue.
#pragma once
#include "conv_buf_impl.h"

template <int TH, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_wino_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int TM = TH / WGM;
  constexpr int WTN = BN / WGN, TN = WTN / 32;
  constexpr int HR = TH + 2;
  constexpr int A_STAGE = HR * 4 * 32 * 8;      // floats: [hy][nu][pair][8]
  constexpr int B_STAGE = 6 * BN * 8;           // floats: the 2 x 3 (nu, dy) steps of one macro step
  constexpr int ITEMS = HR * 32 * 2;            // (halo row, pair, 4-channel group)
  constexpr int AH = (ITEMS + NT - 1) / NT;     // items per thread per chunk
  constexpr int BF4 = 6 * BN * 2;               // float4 of one B stage
  constexpr int BLD = (BF4 + NT - 1) / NT;
  static_assert(TH % WGM == 0 && TM >= 1 && TN >= 1 && AH <= 2, "bad tile");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [A stage 0][A stage 1][B stage 0][B stage 1]
  float* const Bsm = smem + 2 * A_STAGE;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WGN, wn = wv % WGN;

  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr ((FLAGS & CONV_B_XCD_M) != 0) {
    const int nbx = gridDim.x, nby = gridDim.y;
    const int nwg = nbx * nby;
    const int lin = by * nbx + bx;
    const int xcd = lin & 7, idx = lin >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int nl = base + idx;
    bx = nl / nby;
    by = nl - bx * nby;
  }
  const int ntx = (p.W + 63) >> 6, nty = (p.H + TH - 1) / TH;
  const int img = bx / (ntx * nty);
  const int trem = bx - img * (ntx * nty);
  const int y0 = (trem / ntx) * TH, x0 = (trem % ntx) * 64;
  const int n0 = by * BN;

  // ---- A staging items: (halo row hy, pair tp, channel group q) ---------------------------------------
  int a_y[AH], a_x[AH];       // image row / first pixel (2*tp - 1) of the item
  unsigned a_ok[AH];          // bit j: pixel a_x + j is inside the image (and the row is); 0x10: the item exists
  int a_lds[AH];              // float index of (hy, nu = 0, tp) chunk q (swizzled) inside an A stage
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    const int f = t + NT * i;
    const bool slot = f < ITEMS;
    const int q = f & 1, tp = (f >> 1) & 31, hy = slot ? (f >> 6) : 0;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + 2 * tp;
    a_y[i] = iy; a_x[i] = ix;
    unsigned ok = slot ? 0x10u : 0u;
    if (slot && iy >= 0 && iy < p.H)
      for (int j = 0; j < 4; ++j)
        if (ix + j >= 0 && ix + j < p.W) ok |= 1u << j;
    a_ok[i] = ok;
    a_lds[i] = ((hy * 4) * 32 + tp) * 8 + ((q ^ ((tp >> 3) & 1)) << 2);
  }
  const int scol = (t & 1) * 4;
  unsigned a_off[AH];   // byte offset of pixel (a_y, a_x) channel group q in the current segment
  unsigned a_pix = 0;   // bytes per pixel of the current segment
  conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);
  int sg = 0, c0 = 0, segC = p.seg[0].C;
  auto setup_seg = [&]() {
    const ConvSeg& s = p.seg[sg];
    arsrc = conv_make_rsrc(s.ptr);
    segC = s.C;
    a_pix = (unsigned)s.stride * 4u;
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
#pragma unroll
    for (int i = 0; i < AH; ++i)  // may point outside the tensor: only dereferenced under a_ok
      a_off[i] = (unsigned)(((long long)((size_t)be * p.H + a_y[i]) * p.W + a_x[i]) * s.stride + scol) * 4u;
  };

  // ---- B staging -----------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 8;
  const int nsteps = nkc * 12;   // (chunk, nu, dy) steps; a macro step = 2 nu x 3 dy
  const int nmacro = nkc * 2;
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int f = t + NT * i;
    const bool slot = f < BF4;
    const int ch = f & 1, row = slot ? ((f >> 1) % BN) : 0, st = slot ? (f >> 1) / BN : 0;   // st = step inside the macro step
    boff[i] = (unsigned)(((size_t)(n0 + row) * nsteps * 8 + st * 8 + ch * 4) * 4);
    blds[i] = slot ? (st * BN + row) * 8 + ((ch ^ ((row >> 3) & 1)) << 2) : -1;
  }

  bf4 araw[4];   // the four pixels of the item in flight
  bf4 breg[BLD];
  bool chunk_ok = true;  // false past the last chunk: stage zeros
  auto load_item = [&](int i) {
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = chunk_ok && ((a_ok[i] >> j) & 1u);
      araw[j] = conv_buf_load(arsrc, ok ? a_off[i] + (unsigned)j * a_pix : OOB, so);
    }
  };
  auto store_item = [&](int i, int stage) {
    if (!(a_ok[i] & 0x10u)) return;
    float* As = smem + stage * A_STAGE + a_lds[i];
    const bf4 v0 = araw[0] - araw[2];
    const bf4 v1 = araw[1] + araw[2];
    const bf4 v2 = araw[2] - araw[1];
    const bf4 v3 = araw[1] - araw[3];
    *reinterpret_cast<bf4*>(As) = v0;
    *reinterpret_cast<bf4*>(As + 256) = v1;        // nu planes are 32 rows x 8 floats apart
    *reinterpret_cast<bf4*>(As + 512) = v2;
    *reinterpret_cast<bf4*>(As + 768) = v3;
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= nkc) { chunk_ok = false; return; }
    c0 += 8;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto load_b = [&](int ms) {   // macro step ms = chunk * 4 + nu
    const unsigned so = (unsigned)(ms < nmacro ? ms : nmacro - 1) * 192u;   // 6 steps x 8 floats x 4 B
#pragma unroll
    for (int i = 0; i < BLD; ++i) breg[i] = conv_buf_load(brsrc, boff[i], so);
  };
  auto store_b = [&](int stage) {
    float* Bs = Bsm + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      if (NT * (i + 1) <= BF4) *reinterpret_cast<bf4*>(Bs + blds[i]) = breg[i];
      else if (blds[i] >= 0) *reinterpret_cast<bf4*>(Bs + blds[i]) = breg[i];
    }
  };

  f32x16 acc[TM][4][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][v][j][r] = 0.f;

  // ---- fragment addresses, in float4 units: row * 2 + (K-half ^ bit 3 of the row) -------------------------------
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  constexpr int A_STAGE4 = A_STAGE / 4, B_STAGE4 = B_STAGE / 4;
  const int wy = wm * TM;
  const int swb = (l31 >> 3) & 1;
  const int a_ad = (wy * 4 * 32 + l31) * 2 + (half ^ swb);        // + ((mt + dy) * 4 + nu) * 64
  const int b_ad = 2 * A_STAGE4 + (wn * WTN + l31) * 2 + (half ^ swb);
  int a_cur = a_ad;

  auto compute = [&](auto step_c) {
    constexpr int STEP = decltype(step_c)::value;   // nu * 3 + dy
    constexpr int NU = STEP / 3, DY = STEP % 3;
    constexpr int BOFF = ((NU >> 1) & 1) * B_STAGE4 + ((NU & 1) * 3 + DY) * BN * 2;
    bf4 a[TM], b[TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) a[mt] = smem4[a_cur + ((mt + DY) * 4 + NU) * 64];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) b[nt] = smem4[b_ad + BOFF + nt * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][NU][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[nt][j], acc[mt][NU][nt], 0, 0, 0);
  };

  // ---- pipeline: macro step = two nu planes (6 steps); the weights of the next macro step are loaded before the MFMAs
  // and stored to the other B stage behind them; the A item of chunk kc+1 is loaded in macro step 0 and stored in 1.
  setup_seg();
#pragma unroll
  for (int i = 0; i < AH; ++i) { load_item(i); store_item(i, 0); }
  load_b(0);
  store_b(0);
  next_chunk(1);
  __syncthreads();
  int a_stage = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    auto macro = [&](auto m_c) {
      constexpr int MS = decltype(m_c)::value;   // 0: nu 0,1   1: nu 2,3
      load_b(kc * 2 + MS + 1);
      if constexpr (MS < AH) load_item(MS);
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, MS * 6 + 0>{});
      compute(std::integral_constant<int, MS * 6 + 1>{});
      compute(std::integral_constant<int, MS * 6 + 2>{});
      compute(std::integral_constant<int, MS * 6 + 3>{});
      compute(std::integral_constant<int, MS * 6 + 4>{});
      compute(std::integral_constant<int, MS * 6 + 5>{});
      __builtin_amdgcn_sched_barrier(0);
      store_b((MS + 1) & 1);
      if constexpr (MS < AH) store_item(MS, a_stage ^ 1);
      __syncthreads();
    };
    macro(std::integral_constant<int, 0>{});
    macro(std::integral_constant<int, 1>{});
    next_chunk(kc + 2);
    a_stage ^= 1;
    a_cur = a_ad + a_stage * A_STAGE4;
  }

  // ---- epilogue: output transform, bias + leaky_relu, 128-B row stores ---------------------------------
  // C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) = pixel pair.
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int n = n0 + wn * WTN + nt * 32 + l31;
    const float bv = p.bias[n];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int y = y0 + wy + mt;
      if (y >= p.H) continue;
      const size_t rowbase = ((size_t)img * p.H + y) * p.W;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = x0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half);
        const float m0 = acc[mt][0][nt][r], m1 = acc[mt][1][nt][r], m2 = acc[mt][2][nt][r], m3 = acc[mt][3][nt][r];
        float e = (m0 + m1) + m2 + bv;
        float o = (m1 - m2) - m3 + bv;
        if (p.leaky) { e = e > 0.f ? e : 0.2f * e; o = o > 0.f ? o : 0.2f * o; }
        if (x < p.W) p.out[(rowbase + x) * p.ostride + n] = e;
        if (x + 1 < p.W) p.out[(rowbase + x + 1) * p.ostride + n] = o;
      }
    }
  }
}

template <int TH, int BN, int WGM, int WGN, int FLAGS>
hipError_t conv_wino_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (2 * (size_t)(TH + 2) * 4 * 32 * 8 + 2 * 6 * (size_t)BN * 8) * sizeof(float);
  auto kern = conv_wino_kernel<TH, BN, WGM, WGN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  const int ntx = (p.W + 63) / 64, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
