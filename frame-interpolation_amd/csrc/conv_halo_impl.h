// conv_halo_impl.h -- 3x3 Conv2D('same') + bias + leaky_relu as an implicit GEMM whose A operand is staged ONCE per
// 16-channel chunk for all nine taps.
//
// conv_buf_kernel re-gathers the BM x 16 activation tile for every tap: nine global loads + nine LDS stores of the
// same pixels.  Here a workgroup owns a 2-D patch of TH x 32 output pixels of one image; per 16-channel chunk it
// stages the (TH+2) x 34 halo patch once and runs the nine taps out of LDS by shifting the fragment read address.
// Per chunk a 256-pixel tile moves 340 rows instead of 9 x 256: 6.8x fewer A loads / LDS stores, and the matrix
// pipe - the resource the fp32 MFMA rate makes scarce - sees that many fewer competing instructions.
//
//   * K order is (concat segment, 16-channel chunk, tap): the weights are packed [Cout][chunk][tap][16].
//   * LDS image of the halo: rows of 64 B (16 channels of one pixel), row index = hy * 40 + hx (row pitch 40
//     pixels), chunk c of a row stored at chunk position c ^ ((row >> 2) & 3).  A pitch of 40 makes the swizzle
//     term of a row `dy` below equal to the original XOR 2*(dy & 1), i.e. the two K-halves swap: every tap's
//     fragment address is one of six per-lane base addresses (3 dx shifts x 2 K-halves) + an immediate offset.
//   * the weight tile of one (chunk, tap) step is [BN][16] floats through a 3-stage LDS ring; the loads of
//     step s+2 are issued before the MFMAs of step s and stored behind them; one barrier per step.
//   * zero padding, image edges and ragged patches: buffer bounds check (offset 0xFFFFFFFF -> zeros) on the way
//     in, masked stores on the way out.
//   * segments with nearest-upsampled input (the 2x2 convs) and kernel sizes other than 3 stay on conv_buf_kernel.
#pragma once
#include "conv_buf_impl.h"

template <int TH, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_halo_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int TM = TH / WGM;                 // output rows (32-pixel MFMA tiles) per wave
  constexpr int WTN = BN / WGN, TN = WTN / 32;
  constexpr int PITCH = 40, HR = TH + 2, HC = 34;
  constexpr int A_STAGE = HR * PITCH * 16;     // floats
  constexpr int B_STAGE = BN * 16;
  constexpr int HF4 = HR * HC * 4;             // float4 of one halo chunk
  constexpr int AH = (HF4 + NT - 1) / NT;      // per thread
  constexpr int BF4 = BN * 4;
  constexpr int BLD = (BF4 + NT - 1) / NT;
  static_assert(TH % WGM == 0 && TM >= 1 && TN >= 1, "bad tile");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [A stage 0][A stage 1][B ring x3]
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
  const int ntx = (p.W + 31) >> 5, nty = (p.H + TH - 1) / TH;
  const int img = bx / (ntx * nty);
  const int trem = bx - img * (ntx * nty);
  const int y0 = (trem / ntx) * TH, x0 = (trem % ntx) * 32;
  const int n0 = by * BN;

  // ---- A staging: this thread's float4 slots of the halo chunk ------------------------------------
  int aiy[AH], aix[AH];
  bool ain[AH];
  int alds[AH];  // float index inside an A stage, -1: no slot
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    const int f = t + NT * i;
    const bool slot = f < HF4;
    const int r = slot ? (f >> 2) : 0, ch = f & 3;
    const int hy = r / HC, hx = r - hy * HC;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    aiy[i] = iy; aix[i] = ix;
    ain[i] = slot && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int lrow = hy * PITCH + hx;
    alds[i] = slot ? lrow * 16 + ((ch ^ ((lrow >> 2) & 3)) << 2) : -1;
  }
  const int scol = (t & 3) * 4;
  unsigned aoff[AH];
  conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);
  int sg = 0, c0 = 0, segC = p.seg[0].C;
  auto setup_seg = [&]() {
    const ConvSeg& s = p.seg[sg];
    arsrc = conv_make_rsrc(s.ptr);
    segC = s.C;
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
#pragma unroll
    for (int i = 0; i < AH; ++i)
      aoff[i] = ain[i] ? (unsigned)((((size_t)be * p.H + aiy[i]) * p.W + aix[i]) * s.stride + scol) * 4u : OOB;
  };

  // ---- B staging -------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 16;
  const int nsteps = nkc * 9;
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int f = t + NT * i;
    const bool slot = f < BF4;
    const int row = slot ? (f >> 2) : 0, ch = f & 3;
    boff[i] = (unsigned)(((size_t)(n0 + row) * nsteps * 16 + ch * 4) * 4);
    blds[i] = slot ? row * 16 + ((ch ^ ((row >> 2) & 3)) << 2) : -1;
  }

  bf4 areg[AH], breg[BLD];
  auto load_a = [&]() {  // the chunk (sg, c0); then advance to the next one
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int i = 0; i < AH; ++i) areg[i] = conv_buf_load(arsrc, aoff[i], so);
  };
  auto next_chunk = [&](int kc_next) {  // state for chunk kc_next (all-OOB past the end)
    if (kc_next >= nkc) {
#pragma unroll
      for (int i = 0; i < AH; ++i) aoff[i] = OOB;
      return;
    }
    c0 += 16;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto store_a = [&](int stage) {
    float* As = smem + stage * A_STAGE;
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      if (NT * (i + 1) <= HF4) *reinterpret_cast<bf4*>(As + alds[i]) = areg[i];   // every thread has this slot
      else if (alds[i] >= 0) *reinterpret_cast<bf4*>(As + alds[i]) = areg[i];
    }
  };
  auto load_b = [&](int s) {
    const unsigned so = (unsigned)(s < nsteps ? s : nsteps - 1) * 64u;
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses, in float4 units (indexing a bf4 array is what lets hipcc emit ds_read_b128: with float
  // indices it cannot prove the 16-byte alignment and falls back to pairs of ds_read2_b32 - 50-70 % bank-conflict
  // cycles measured) ------------------------------------------------------------------------------------------
  // A: LDS row (wy + mt + dy) * 40 + (l31 + dx); its swizzle term is ((l31 + dx) >> 2 & 3) ^ 2 * ((wy+mt+dy) & 1),
  // so K-half kq of that row is K-half kq ^ ((wy+mt+dy) & 1) of the base address.
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  constexpr int A_STAGE4 = A_STAGE / 4, B_STAGE4 = B_STAGE / 4;
  const int wy = wm * TM;
  int a_ad[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int px = l31 + dx;
    const int swx = (px >> 2) & 3;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      a_ad[dx][k] = (wy * PITCH + px) * 4 + ((((k ^ (wy & 1)) << 1) | half) ^ swx);
  }
  int b_ad[2];
  {
    const int sw = (l31 >> 2) & 3;
    b_ad[0] = 2 * A_STAGE4 + (wn * WTN + l31) * 4 + (half ^ sw);
    b_ad[1] = 2 * A_STAGE4 + (wn * WTN + l31) * 4 + ((2 | half) ^ sw);
  }

  int a_cur[3][2];  // a_ad + offset of the A stage being read
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) { a_cur[dx][0] = a_ad[dx][0]; a_cur[dx][1] = a_ad[dx][1]; }

  auto compute = [&](auto tap_c) {
    constexpr int TAP = decltype(tap_c)::value;
    constexpr int DY = TAP / 3, DX = TAP % 3;
    bf4 a[2][TM], b[2][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int par = (mt + DY) & 1;
      a[0][mt] = smem4[a_cur[DX][par] + (mt + DY) * PITCH * 4];
      a[1][mt] = smem4[a_cur[DX][par ^ 1] + (mt + DY) * PITCH * 4];
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      b[0][nt] = smem4[b_ad[0] + (TAP % 3) * B_STAGE4 + nt * 128];
      b[1][nt] = smem4[b_ad[1] + (TAP % 3) * B_STAGE4 + nt * 128];
    }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kq][mt][j], b[kq][nt][j], acc[mt][nt], 0, 0, 0);
  };

  // ---- pipeline -------------------------------------------------------------------------------------------
  setup_seg();
  load_a();          // chunk 0
  load_b(0);
  store_a(0);
  store_b(0);
  load_b(1);
  store_b(1);
  next_chunk(1);
  __syncthreads();
  int a_off = 0;     // float offset of the A stage being read
  for (int kc = 0; kc < nkc; ++kc) {
    const int s0 = kc * 9;
    auto step = [&](auto tap_c) {
      constexpr int TAP = decltype(tap_c)::value;
      load_b(s0 + TAP + 2);
      if constexpr (TAP == 0) load_a();  // chunk kc + 1
      __builtin_amdgcn_sched_barrier(0);
      compute(tap_c);
      __builtin_amdgcn_sched_barrier(0);
      store_b((TAP + 2) % 3);
      if constexpr (TAP == 8) store_a(a_off ? 0 : 1);
      __syncthreads();
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    next_chunk(kc + 2);
    a_off = a_off ? 0 : A_STAGE;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) { a_cur[dx][0] = a_ad[dx][0] + a_off / 4; a_cur[dx][1] = a_ad[dx][1] + a_off / 4; }
  }

  // ---- epilogue: bias + leaky_relu, 128-B row stores -------------------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); row = x inside the patch.
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
        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (x < p.W) {
          float v = acc[mt][nt][r] + bv;
          if (p.leaky) v = v > 0.f ? v : 0.2f * v;
          p.out[(rowbase + x) * p.ostride + n] = v;
        }
      }
    }
  }
}

template <int TH, int BN, int WGM, int WGN, int FLAGS>
hipError_t conv_halo_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (2 * (size_t)(TH + 2) * 40 * 16 + 3 * (size_t)BN * 16) * sizeof(float);
  auto kern = conv_halo_kernel<TH, BN, WGM, WGN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  const int ntx = (p.W + 31) / 32, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
