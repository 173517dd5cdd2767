// conv_foldx3_impl.h -- OPT-IN precision mode "bf16x3" for the decoder's nearest-x2 upsample + 2x2 'same' convolution
// (fusion.py:133-135), in its sub-pixel form: output (2y+py, 2x+px) = sum over the taps (a, b), a <= py, b <= px, of
// in(y+a, x+b) * Wsum[py][px][a][b]  -  nine (tap, phase) products per four outputs (film_engine.cpp sums the kernel
// taps that read the same input pixel at film_finalize).  The fp32 path runs the four phases as four gathers of
// conv_buf_kernel; here ONE halo-staged patch of the low-resolution input serves all nine products:
//
//   * a workgroup owns TH x 32 low-resolution pixels (-> 2TH x 64 outputs) x BN channels; the (TH+1) x 33 patch of a
//     16-channel chunk is split into bf16 hi / mid (nearest, conv_split4) ONCE on the way into LDS;
//   * nine steps per chunk, ordered by tap so that an A fragment serves up to four phases:
//         step   0    1    2  |  3    4    5  |  6    7    8
//         tap   00   00   00  | 00   01   01  | 10   10   11
//         phase  0    1    2  |  3    1    3  |  2    3    3            (phase = 2 py + px)
//     three steps form a stage (one barrier per 9*TM*TN MFMAs per wave); weights [Cout][chunk][step][plane][16] bf16
//     travel through a 3-slot LDS ring (three stages per chunk: slot = stage, compile time);
//   * accumulators: four phase planes of TM x TN tiles; the epilogue scatters them to the four output positions.
#pragma once
#include "conv_split_impl.h"

template <int TH, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64, WGM* WGN <= 4 ? 2 : 1) void conv_foldx3_kernel(ConvParams p) {   // 4-wave tiles: two workgroups per CU

  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int TM = TH / WGM;
  constexpr int WTN = BN / WGN, TN = WTN / 32;
  constexpr int HR = TH + 1, HC = 33;
  constexpr int A_PLANE = HR * HC * 32;          // bytes: one bf16 plane of the patch chunk
  constexpr int A_STAGE = 2 * A_PLANE;
  constexpr int B_PLANE = BN * 32;
  constexpr int B_STEP = 2 * B_PLANE;            // [plane][BN][32 B]
  constexpr int B_STAGE = 3 * B_STEP;
  constexpr int HF4 = HR * HC * 4;               // (patch pixel, 4-channel group)
  constexpr int AH = (HF4 + NT - 1) / NT;
  constexpr int BU = BN * 12;                    // 16-byte units of one weight stage
  constexpr int BLD = (BU + NT - 1) / NT;
  static_assert(TH % WGM == 0 && TM >= 1 && TN >= 1, "bad tile");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_b[];  // [A0][A1][B ring x3]
  unsigned char* const Bsm = smem_b + 2 * A_STAGE;

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
  const int ntx = (p.W + 31) >> 5, nty = (p.H + TH - 1) / TH;   // p.H, p.W: the low-resolution grid
  const int img = bx / (ntx * nty);
  const int trem = bx - img * (ntx * nty);
  const int y0 = (trem / ntx) * TH, x0 = (trem % ntx) * 32;
  const int n0 = by * BN;

  // ---- A staging: patch pixel (hy, hx) = input (y0 + hy, x0 + hx); zero beyond the bottom / right edge ------------
  unsigned aoff[AH];
  int alds[AH];  // byte offset of plane 0 of this thread's 4 channels, -1: no slot
  {
    const ConvSeg& s = p.seg[0];
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      const int f = t + NT * i;
      const bool slot = f < HF4;
      const int r = slot ? (f >> 2) : 0, ch = f & 3;
      const int hy = r / HC, hx = r - hy * HC;
      const int iy = y0 + hy, ix = x0 + hx;
      const bool in = slot && iy < p.H && ix < p.W;
      aoff[i] = in ? (unsigned)((((size_t)be * p.H + iy) * p.W + ix) * s.stride + ch * 4) * 4u : OOB;
      alds[i] = slot ? r * 32 + ((((ch >> 1) ^ ((r >> 3) & 1)) << 4) | ((ch & 1) << 3)) : -1;
    }
  }
  const conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);

  // ---- B staging ---------------------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 16;
  const int nstage = nkc * 3;
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int u = t + NT * i;
    const bool slot = u < BU;
    const int kb = u & 1, row = slot ? (u >> 1) % BN : 0, sp = slot ? (u >> 1) / BN : 0;   // sp = step in stage * 2 + plane
    boff[i] = (unsigned)((size_t)(n0 + row) * nstage * 192 + sp * 32 + kb * 16);
    blds[i] = slot ? sp * B_PLANE + row * 32 + ((kb ^ ((row >> 3) & 1)) << 4) : -1;
  }

  bf4 areg[AH];
  su4 breg[BLD];
  auto load_a = [&](int kc) {
    const unsigned so = (unsigned)kc * 64u;   // 16 channels x 4 B
    const bool ok = kc < nkc;
#pragma unroll
    for (int i = 0; i < AH; ++i) areg[i] = conv_buf_load(arsrc, ok ? aoff[i] : OOB, so);
  };
  auto store_a = [&](int stage) {
    unsigned char* As = smem_b + stage * A_STAGE;
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      if (NT * (i + 1) <= HF4 || alds[i] >= 0) {
        su2 hi, mid, lo;
        conv_split4<false>(areg[i], hi, mid, lo);
        *reinterpret_cast<su2*>(As + alds[i]) = hi;
        *reinterpret_cast<su2*>(As + alds[i] + A_PLANE) = mid;
      }
    }
  };
  auto load_b = [&](int s) {
    const unsigned so = (unsigned)(s < nstage ? s : nstage - 1) * 192u;
#pragma unroll
    for (int i = 0; i < BLD; ++i) breg[i] = conv_buf_load_u4(brsrc, boff[i], so);
  };
  auto store_b = [&](int ring) {
    unsigned char* Bs = Bsm + ring * B_STAGE;
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      if (NT * (i + 1) <= BU || blds[i] >= 0) *reinterpret_cast<su4*>(Bs + blds[i]) = breg[i];
  };

  f32x16 acc[4][TM][TN];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

  // fragment addresses in 16-byte units: row * 2 + (K-half ^ bit 3 of the row)
  const su4* const smem16 = reinterpret_cast<const su4*>(smem_b);
  const int wy = wm * TM;
  int a_ad[TM + 1][2];
#pragma unroll
  for (int ry = 0; ry < TM + 1; ++ry)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = (wy + ry) * HC + b + l31;
      a_ad[ry][b] = r * 2 + (half ^ ((r >> 3) & 1));
    }
  int b_ad[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int r = wn * WTN + nt * 32 + l31;
    b_ad[nt] = (2 * A_STAGE) / 16 + r * 2 + (half ^ ((r >> 3) & 1));
  }
  int a_stage_u = 0;

  auto compute = [&](auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    constexpr int TAP[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    constexpr int PH[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};
    sbf8 a[2][2][TM];   // [tap slot in this stage][plane][row]
    // taps of this stage: stage 0 -> {00}, stage 1 -> {00, 01}, stage 2 -> {10, 11}
    constexpr int T0 = TAP[3 * ST], T1 = TAP[3 * ST + 2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        a[0][pl][mt] = __builtin_bit_cast(sbf8, smem16[a_ad[mt + (T0 >> 1)][T0 & 1] + a_stage_u + pl * (A_PLANE / 16)]);
        if constexpr (T1 != T0)
          a[1][pl][mt] = __builtin_bit_cast(sbf8, smem16[a_ad[mt + (T1 >> 1)][T1 & 1] + a_stage_u + pl * (A_PLANE / 16)]);
      }
    auto step = [&](auto k_c) {
      constexpr int K = decltype(k_c)::value;
      constexpr int S = 3 * ST + K;
      constexpr int SLOT = TAP[S] == T0 ? 0 : 1;
      constexpr int Q = PH[S];
      sbf8 b[2][TN];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          b[pl][nt] = __builtin_bit_cast(sbf8, smem16[b_ad[nt] + ST * (B_STAGE / 16) + K * (B_STEP / 16) + pl * (B_PLANE / 16)]);
      // smallest partial products first: hi*mid, mid*hi, hi*hi
      constexpr int PA[3] = {0, 1, 0};
      constexpr int PB[3] = {1, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[Q][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SLOT][PA[k]][mt], b[PB[k]][nt], acc[Q][mt][nt], 0, 0, 0);
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  };

  // ---- pipeline: three stages per chunk, one barrier each; the weights of stage s+2 are requested in front of the
  // MFMAs of stage s and written to ring slot (s+2) % 3 behind them; the patch of the next chunk is requested in stage 0
  // and split + stored in stage 1 --------------------------------------------------------------------------------------
  load_a(0);
  load_b(0);
  store_a(0);
  store_b(0);
  load_b(1);
  store_b(1);
  __syncthreads();
  int a_stage = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    const int s0 = kc * 3;
    auto stage = [&](auto st_c) {
      constexpr int ST = decltype(st_c)::value;
      load_b(s0 + ST + 2);
      if constexpr (ST == 0) load_a(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(st_c);
      __builtin_amdgcn_sched_barrier(0);
      store_b((ST + 2) % 3);
      if constexpr (ST == 1) store_a(a_stage ^ 1);
      __syncthreads();
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    a_stage ^= 1;
    a_stage_u = a_stage * (A_STAGE / 16);
  }

  // ---- epilogue: bias + leaky_relu, phase (py, px) of low-resolution pixel (y, x) -> output (2y+py, 2x+px) ------------
  // C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) = pixel of the patch row.
  const int OW = 2 * p.W;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int n = n0 + wn * WTN + nt * 32 + l31;
    const float bv = p.bias[n];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      const int y = y0 + wy + mt;
      if (y >= p.H) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t rowbase = ((size_t)img * 2 * p.H + 2 * y + (q >> 1)) * OW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (x < p.W) {
            float v = acc[q][mt][nt][r] + bv;
            if (p.leaky) v = v > 0.f ? v : 0.2f * v;
            p.out[(rowbase + 2 * x + (q & 1)) * p.ostride + n] = v;
          }
        }
      }
    }
  }
}

template <int TH, int BN, int WGM, int WGN, int FLAGS>
hipError_t conv_foldx3_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = 2 * 2 * (size_t)(TH + 1) * 33 * 32 + 3 * 3 * 2 * (size_t)BN * 32;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = conv_foldx3_kernel<TH, BN, WGM, WGN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  if (p.nseg != 1 || p.seg[0].up) return hipErrorInvalidValue;
  const int ntx = (p.W + 31) / 32, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
