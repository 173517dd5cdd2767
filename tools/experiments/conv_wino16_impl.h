// conv_wino16_impl.h -- EXPERIMENT (tools/conv_bench.hip only): the 16-channel-chunk version of conv_wino_kernel that
// the 8-channel version replaced (one 8-wave workgroup per CU instead of two).
// Measured on MI355X (tools/conv_bench.hip): 167-183 TFLOP/s (direct-conv FLOPs) on the large-M
// layers vs 137-143 for the direct kernels (+22-28 %), 144 vs 133 on the deep small-M layers: the 1.5x MFMA saving is
// partly paid back in LDS traffic, occupancy (128 accumulator registers -> 2 waves per SIMD, one 8-wave workgroup per
// CU) and clock (2.19 GHz at 79 % pipe occupancy).  Used for 3x3 layers with Cout % 128 == 0 on the large levels.
//
// 3x3 Conv2D('same') + bias + leaky_relu with the 1-D Winograd transform F(2,3) along x on top
// of the halo-staged implicit GEMM: 12 matrix steps per 16-channel chunk instead of 18 for every PAIR of output
// pixels, i.e. 1.5x fewer fp32 MFMAs for the same convolution (fp32 throughout; the rounding differs from the direct
// sum at the 1e-6 level, the dtype does not).
//
// For an output row y and the pixel pair x = 2t, 2t+1 with inputs d0..d3 = in[.][2t-1 .. 2t+2]:
//     v0 = d0 - d2,  v1 = d1 + d2,  v2 = d2 - d1,  v3 = d1 - d3                      (input transform, per input row)
//     u0 = g0,  u1 = (g0 + g1 + g2)/2,  u2 = (g0 - g1 + g2)/2,  u3 = g2               (weights, per (dy, cin, cout), offline)
//     m_nu[y][t][n] = sum_dy sum_c v_nu[y+dy-1][t][c] * u_nu[dy][c][n]                (4 x 3 GEMM steps per chunk)
//     out[y][2t] = (m0 + m1) + m2,   out[y][2t+1] = (m1 - m2) - m3
//
//   * a workgroup owns TH rows x 64 pixels (32 pairs = one 32-row MFMA tile per row) x BN output channels;
//   * per 16-channel chunk the (TH+2) halo rows are transformed ONCE on the way into LDS: image
//     [halo row][nu][pair][16 channels], 64-byte rows, chunk c of row r at c ^ ((r >> 2) & 3); every (nu, dy) step
//     reads its A fragments at one of two per-lane base addresses + an immediate;
//   * weights [Cout][chunk][nu*3 + dy][16]; one B stage holds the three dy steps of one nu ([3][BN][16], double
//     buffered): one barrier per 3 x 16 MFMAs per wave instead of one per 16;
//   * accumulators: 4 (nu) x TN tiles per output row; the output transform runs on them in the epilogue.
#pragma once
#include "../../frame-interpolation_amd/csrc/conv_buf_impl.h"

template <int TH, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_wino16_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int TM = TH / WGM;
  constexpr int WTN = BN / WGN, TN = WTN / 32;
  constexpr int HR = TH + 2;
  constexpr int A_STAGE = HR * 4 * 32 * 16;     // floats: [hy][nu][pair][16]
  constexpr int B_STAGE = 3 * BN * 16;          // floats: the three dy steps of one nu
  constexpr int ITEMS = HR * 32 * 4;            // (halo row, pair, 4-channel group)
  constexpr int AH = (ITEMS + NT - 1) / NT;     // items per thread per chunk
  constexpr int BF4 = 3 * BN * 4;               // float4 of one B stage
  constexpr int BLD = (BF4 + NT - 1) / NT;
  static_assert(TH % WGM == 0 && TM >= 1 && TN >= 1 && AH <= 4, "bad tile");
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
    const int q = f & 3, tp = (f >> 2) & 31, hy = slot ? (f >> 7) : 0;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + 2 * tp;
    a_y[i] = iy; a_x[i] = ix;
    unsigned ok = slot ? 0x10u : 0u;
    if (slot && iy >= 0 && iy < p.H)
      for (int j = 0; j < 4; ++j)
        if (ix + j >= 0 && ix + j < p.W) ok |= 1u << j;
    a_ok[i] = ok;
    a_lds[i] = ((hy * 4) * 32 + tp) * 16 + ((q ^ ((tp >> 2) & 3)) << 2);
  }
  const int scol = (t & 3) * 4;
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
  const int nkc = p.Ctot / 16;
  const int nsteps = nkc * 12;   // (chunk, nu, dy) steps; a macro step = the 3 dy of one (chunk, nu)
  const int nmacro = nkc * 4;
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int f = t + NT * i;
    const bool slot = f < BF4;
    const int ch = f & 3, row = slot ? ((f >> 2) % BN) : 0, dy = slot ? (f >> 2) / BN : 0;
    boff[i] = (unsigned)(((size_t)(n0 + row) * nsteps * 16 + dy * 16 + ch * 4) * 4);
    blds[i] = slot ? (dy * BN + row) * 16 + ((ch ^ ((row >> 2) & 3)) << 2) : -1;
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
    *reinterpret_cast<bf4*>(As + 512) = v1;        // nu planes are 32 rows x 16 floats apart
    *reinterpret_cast<bf4*>(As + 1024) = v2;
    *reinterpret_cast<bf4*>(As + 1536) = v3;
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= nkc) { chunk_ok = false; return; }
    c0 += 16;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto load_b = [&](int ms) {   // macro step ms = chunk * 4 + nu
    const unsigned so = (unsigned)(ms < nmacro ? ms : nmacro - 1) * 192u;
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

  // ---- fragment addresses, in float4 units (a bf4-array index is what lets hipcc prove the 16-byte alignment and emit
  // ds_read_b128 instead of pairs of ds_read2_b32) -------------------------------------------------------------
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  constexpr int A_STAGE4 = A_STAGE / 4, B_STAGE4 = B_STAGE / 4;
  const int wy = wm * TM;
  const int sw = (l31 >> 2) & 3;
  const int a_ad0 = (wy * 4 * 32 + l31) * 4 + (half ^ sw);        // + ((mt + dy) * 4 + nu) * 128
  const int a_ad1 = (wy * 4 * 32 + l31) * 4 + ((2 | half) ^ sw);
  const int b_ad0 = 2 * A_STAGE4 + (wn * WTN + l31) * 4 + (half ^ sw);
  const int b_ad1 = 2 * A_STAGE4 + (wn * WTN + l31) * 4 + ((2 | half) ^ sw);
  int a_cur0 = a_ad0, a_cur1 = a_ad1;

  auto compute = [&](auto step_c) {
    constexpr int STEP = decltype(step_c)::value;   // nu * 3 + dy
    constexpr int NU = STEP / 3, DY = STEP % 3;
    constexpr int BOFF = (NU & 1) * B_STAGE4 + DY * BN * 4;   // 4 macro steps per chunk: the stage parity is static
    bf4 a[2][TM], b[2][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      a[0][mt] = smem4[a_cur0 + ((mt + DY) * 4 + NU) * 128];
      a[1][mt] = smem4[a_cur1 + ((mt + DY) * 4 + NU) * 128];
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      b[0][nt] = smem4[b_ad0 + BOFF + nt * 128];
      b[1][nt] = smem4[b_ad1 + BOFF + nt * 128];
    }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[mt][NU][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kq][mt][j], b[kq][nt][j], acc[mt][NU][nt], 0, 0, 0);
  };

  // ---- pipeline: per macro step (one nu, three dy) the weights of the NEXT macro step are loaded before the 48 MFMAs
  // and stored to the other B stage behind them; the A items of chunk kc+1 are loaded, transformed and stored one per
  // macro step; one barrier per macro step.
  setup_seg();
#pragma unroll
  for (int i = 0; i < AH; ++i) { load_item(i); store_item(i, 0); }
  load_b(0);
  store_b(0);
  next_chunk(1);
  __syncthreads();
  int a_stage = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    auto macro = [&](auto nu_c) {
      constexpr int NU = decltype(nu_c)::value;
      load_b(kc * 4 + NU + 1);
      if constexpr (NU < AH) load_item(NU);
      __builtin_amdgcn_sched_barrier(0);
      compute(std::integral_constant<int, NU * 3 + 0>{});
      compute(std::integral_constant<int, NU * 3 + 1>{});
      compute(std::integral_constant<int, NU * 3 + 2>{});
      __builtin_amdgcn_sched_barrier(0);
      store_b((NU + 1) & 1);
      if constexpr (NU < AH) store_item(NU, a_stage ^ 1);
      __syncthreads();
    };
    macro(std::integral_constant<int, 0>{});
    macro(std::integral_constant<int, 1>{});
    macro(std::integral_constant<int, 2>{});
    macro(std::integral_constant<int, 3>{});
    next_chunk(kc + 2);
    a_stage ^= 1;
    a_cur0 = a_ad0 + a_stage * A_STAGE4;
    a_cur1 = a_ad1 + a_stage * A_STAGE4;
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
hipError_t conv_wino16_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (2 * (size_t)(TH + 2) * 4 * 32 * 16 + 2 * 3 * (size_t)BN * 16) * sizeof(float);
  auto kern = conv_wino16_kernel<TH, BN, WGM, WGN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static bool attr_set[64] = {};  // per device: the attribute belongs to the function ON the current device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  const int ntx = (p.W + 63) / 64, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
