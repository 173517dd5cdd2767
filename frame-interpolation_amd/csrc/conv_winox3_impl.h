// conv_winox3_impl.h -- OPT-IN precision mode "bf16x3" (film_set_option("precision", 2)) for the layers the fp32 path
// runs on conv_wino_kernel: the 1-D Winograd transform F(2,3) along x (conv_wino_impl.h) with every transformed
// operand split into two bf16 pieces (round to nearest: v = hi + mid, |v - hi - mid| <= 2^-17 |v|, conv_split_impl.h)
// and each product formed as hi*hi + hi*mid + mid*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
// 12 x 3 bf16 MFMAs of 32 cycles per 16 channels and pixel pair, where the fp32 Winograd kernel issues 12 x 8 fp32
// MFMAs of 64 cycles and the direct bf16x3 kernel 18 x 3.  The bf16 matrix pipe is power limited on this part
// (MFMA-busy x clock is constant across tile shapes of conv_halo_split_kernel, profiles/), so fewer MFMAs per output
// is what buys time.
//
//   * input transform in fp32 on the way into LDS (v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3), THEN the
//     split; weights are transformed in fp32 and split once at film_finalize;
//   * a workgroup owns TH rows x 64 pixels (32 pairs = one MFMA tile per row) x BN output channels.  The four nu
//     planes are four independent GEMMs: wave half h accumulates nu = 2h, 2h+1 for its TM rows x TN channel tiles
//     (TM*TN*2 accumulator tiles instead of TM*TN*4, which is what lets a wave hold a 2 x 2 tile block and read
//     2*(TM+TN) fragments per 3*TM*TN MFMAs); the halves exchange one plane through LDS in the epilogue
//     (even pixel = (m0 + m1) + m2 on half 0, odd pixel = (m1 - m2) - m3 on half 1: the fp32 kernel's order);
//   * LDS: A image [plane][halo row][nu][pair][16 bf16 = 32 B], double buffered; weights in a 3-stage ring of
//     (dy, j) stages [h][plane][BN][32 B] holding nu = j and nu = 2 + j: one barrier per 3*TM*TN MFMAs per wave;
//     the two 16-byte K-halves of a 32-byte row are swapped when bit 3 of the row index is set (conflict-free
//     ds_read_b128, as in conv_split_impl.h);
//   * weights in HBM: [Cout][chunk of 16][dy][j][h][plane][16] bf16 (128 contiguous bytes per stage and channel).
#pragma once
#include "conv_split_impl.h"

template <int TH, int BN, int TM, int TN, int FLAGS>
__global__ __launch_bounds__((TH / TM) * (BN / (32 * TN)) * 128) void conv_winox3_kernel(ConvParams p) {
  constexpr int RG = TH / TM, NG = BN / (32 * TN), PW = RG * NG, NW = 2 * PW, NT = NW * 64;
  constexpr int HR = TH + 2;
  constexpr int A_PLANE = HR * 4 * 32 * 32;    // bytes
  constexpr int A_STAGE = 2 * A_PLANE;
  constexpr int B_PLANE = BN * 32;             // one (h, plane) block of a stage
  constexpr int B_STAGE = 4 * B_PLANE;
  constexpr int ITEMS = HR * 32 * 4;           // (halo row, pair, 4-channel group)
  constexpr int AH = (ITEMS + NT - 1) / NT;
  constexpr int BU = BN * 8;                   // 16-byte units of one weight stage
  constexpr int BLD = (BU + NT - 1) / NT;
  static_assert(TH % TM == 0 && BN % (32 * TN) == 0 && AH <= 2 && ITEMS - NT <= NT / 2 && ITEMS >= NT, "bad tile");
  static_assert(2 * PW * TM * TN * 16 * 64 * 4 <= 2 * A_STAGE + 3 * B_STAGE, "exchange buffer does not fit");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_b[];  // [A0][A1][B x3]
  unsigned char* const Bsm = smem_b + 2 * A_STAGE;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = wv / PW, pw = wv % PW;     // nu half, pair-wave
  const int rg = pw / NG, ng = pw % NG;

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

  // ---- A staging items: (halo row hy, pair tp, channel group q of four) -------------------------------------------
  int a_y[AH], a_x[AH];
  unsigned a_ok[AH];          // bit j: pixel a_x + j is inside the image (and the row is); 0x10: the item exists
  int a_lds[AH];              // byte offset of (hy, nu = 0, tp), channels 4q.. inside plane 0 of an A stage
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    // Threads past the last item repeat an item of the first half of the workgroup (same values to the same LDS
    // address): the staging code then has no divergent branch (+8 % on the 528 -> 128 layer).
    int f = t + NT * i;
    if (f >= ITEMS) f -= NT / 2;
    const int q = f & 3, tp = (f >> 2) & 31, hy = f >> 7;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + 2 * tp;
    a_y[i] = iy; a_x[i] = ix;
    unsigned ok = 0x10u;
    if (iy >= 0 && iy < p.H)
      for (int j = 0; j < 4; ++j)
        if (ix + j >= 0 && ix + j < p.W) ok |= 1u << j;
    a_ok[i] = ok;
    a_lds[i] = ((hy * 4) * 32 + tp) * 32 + ((((q >> 1) ^ ((tp >> 3) & 1)) << 4) | ((q & 1) << 3));
  }
  const int scol = (t & 3) * 4;
  unsigned a_off[AH];
  unsigned a_pix = 0;
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

  // ---- B staging ---------------------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 16;
  const int nstage = nkc * 6;    // (chunk, dy, j)
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int u = t + NT * i;
    const bool slot = u < BU;
    const int kb = u & 1, row = slot ? (u >> 1) % BN : 0, hp = slot ? (u >> 1) / BN : 0;   // hp = h * 2 + plane
    boff[i] = (unsigned)((size_t)(n0 + row) * nstage * 128 + hp * 32 + kb * 16);
    blds[i] = slot ? hp * B_PLANE + row * 32 + ((kb ^ ((row >> 3) & 1)) << 4) : -1;
  }

  bf4 araw[4];
  su4 breg[2][BLD];   // weights in flight: loaded in stage s-1, stored to the ring in stage s
  bool chunk_ok = true;
  auto load_item = [&](int i) {
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = chunk_ok && ((a_ok[i] >> j) & 1u);
      araw[j] = conv_buf_load(arsrc, ok ? a_off[i] + (unsigned)j * a_pix : OOB, so);
    }
  };
  auto store_item = [&](int i, int stage) {
    unsigned char* As = smem_b + stage * A_STAGE + a_lds[i];
    const bf4 v[4] = {araw[0] - araw[2], araw[1] + araw[2], araw[2] - araw[1], araw[1] - araw[3]};
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      su2 hi, mid, lo;
      conv_split4<false>(v[nu], hi, mid, lo);
      *reinterpret_cast<su2*>(As + nu * 1024) = hi;             // nu planes are 32 rows x 32 B apart
      *reinterpret_cast<su2*>(As + nu * 1024 + A_PLANE) = mid;
    }
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= nkc) { chunk_ok = false; return; }
    c0 += 16;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto load_b = [&](int s, int buf) {
    const unsigned so = (unsigned)(s < nstage ? s : nstage - 1) * 128u;
#pragma unroll
    for (int i = 0; i < BLD; ++i) breg[buf][i] = conv_buf_load_u4(brsrc, boff[i], so);
  };
  auto store_b = [&](int ring, int buf) {
    unsigned char* Bs = Bsm + ring * B_STAGE;
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      if (NT * (i + 1) <= BU || blds[i] >= 0) *reinterpret_cast<su4*>(Bs + blds[i]) = breg[buf][i];
  };

  f32x16 acc[TM][2][TN];   // [row][nu - 2h][channel tile]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][v][j][r] = 0.f;

  // ---- fragment addresses in 16-byte units: row * 2 + (K-half ^ bit 3 of the row) ---------------------------------
  const su4* const smem16 = reinterpret_cast<const su4*>(smem_b);
  const int wy = rg * TM;
  const int swb = (l31 >> 3) & 1;
  const int a_ad = ((wy * 4 + 2 * h) * 32 + l31) * 2 + (half ^ swb);     // + ((mt + dy) * 4 + j) * 64 + plane, stage
  const int b_ad = (2 * A_STAGE) / 16 + (2 * h) * (B_PLANE / 16) + (ng * TN * 32 + l31) * 2 + (half ^ swb);
  int a_cur = a_ad;

  // Fragment registers are double buffered across stages: the ds_reads of stage s+1 are issued in front of the
  // MFMAs of stage s (its weights sit in ring slot (s+1) % 3 since the barrier of stage s-1, its A rows in the image
  // of this chunk - or, for s+1 = stage 0 of the next chunk, in the other A buffer, complete since stage AH-1).
  sbf8 fa[2][2][TM], fb[2][2][TN];   // [buffer][plane][tile]
  auto fetch = [&](auto st_c, int a_base) {
    constexpr int ST = decltype(st_c)::value;   // dy * 2 + j; six stages per chunk -> ring slot ST % 3
    constexpr int DY = ST >> 1, J = ST & 1, BUF = ST & 1;
    const int bb = b_ad + (ST % 3) * (B_STAGE / 16);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
        fa[BUF][pl][mt] = __builtin_bit_cast(sbf8, smem16[a_base + ((mt + DY) * 4 + J) * 64 + pl * (A_PLANE / 16)]);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
        fb[BUF][pl][nt] = __builtin_bit_cast(sbf8, smem16[bb + nt * 64 + pl * (B_PLANE / 16)]);
    }
  };
  auto compute = [&](auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    constexpr int J = ST & 1, BUF = ST & 1;
    // smallest partial products first: hi*mid, mid*hi, hi*hi
    constexpr int PA[3] = {0, 1, 0};
    constexpr int PB[3] = {1, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][J][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[BUF][PA[k]][mt], fb[BUF][PB[k]][nt], acc[mt][J][nt], 0, 0, 0);
  };

  // ---- pipeline: one barrier per (dy, j) stage.  Weights of stage s+3 are requested in stage s and written to ring
  // slot s % 3 in stage s+1 (a full stage of latency cover); A item i of the next chunk is requested in stage 2i and
  // transformed + split + stored behind the MFMAs of stage 2i+1 (AH <= 2: complete by the barrier of stage 3) -------
  setup_seg();
#pragma unroll
  for (int i = 0; i < AH; ++i) { load_item(i); store_item(i, 0); }
  load_b(0, 0);
  store_b(0, 0);
  load_b(1, 0);
  store_b(1, 0);
  load_b(2, 0);
  next_chunk(1);
  __syncthreads();
  fetch(std::integral_constant<int, 0>{}, a_cur);
  if constexpr ((FLAGS & 64) != 0) fetch(std::integral_constant<int, 1>{}, a_cur);
  int a_stage = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    const int s0 = kc * 6;
    const int a_next = a_ad + (a_stage ^ 1) * (A_STAGE / 16);
    auto stage = [&](auto st_c) {
      constexpr int ST = decltype(st_c)::value;
      // FLAGS 64 / 128 / 256 / 512: timing ablations of tools/retired/conv_bench.hip (no fragment reads / no A item staging /
      // no weight staging / no barriers) - wrong results, never instantiated by the engine
      if constexpr ((FLAGS & 256) == 0) load_b(s0 + ST + 3, (ST + 1) & 1);
      if constexpr ((FLAGS & 128) == 0) if constexpr ((ST & 1) == 0 && ST / 2 < AH) load_item(ST / 2);
      if constexpr ((FLAGS & 64) == 0) fetch(std::integral_constant<int, (ST + 1) % 6>{}, ST == 5 ? a_next : a_cur);
      __builtin_amdgcn_sched_barrier(0);
      compute(st_c);
      // The item's transform + split VALU work is left to the compiler's own placement: forced into an even
      // interleave with the MFMAs (sched_group_barrier, one MFMA + 6 VALU) the kernel is 8 % slower.
      if constexpr ((FLAGS & 1024) != 0) __builtin_amdgcn_sched_barrier(0);   // A/B: VALU strictly behind the MFMAs
      if constexpr ((FLAGS & 128) == 0 && (ST & 1) == 1 && ST / 2 < AH) {
        store_item(ST / 2, a_stage ^ 1);
        if constexpr ((FLAGS & 2048) != 0) {   // A/B: forced even interleave
#pragma unroll
          for (int g = 0; g < 3 * TM * TN; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 64 / (3 * TM * TN) + 1, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((FLAGS & 256) == 0) store_b((ST + 2) % 3, ST & 1);
      if constexpr ((FLAGS & 512) == 0) __syncthreads();
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    next_chunk(kc + 2);
    a_stage ^= 1;
    a_cur = a_next;
  }

  // ---- epilogue: the halves swap one plane through LDS (the staging buffers are free after the last barrier):
  // half 0 gives m1 and finishes the even pixels (m0 + m1) + m2, half 1 gives m2 and finishes the odd pixels
  // (m1 - m2) - m3.  C/D layout of the 32x32 MFMA: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) = pair.
  float* const xbuf = reinterpret_cast<float*>(smem_b);
  constexpr int XW = TM * TN * 16 * 64;   // floats one wave gives
  auto finish = [&](auto h_c) {
    constexpr int H = decltype(h_c)::value;
    float* give = xbuf + (H * PW + pw) * XW + lane;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) give[((mt * TN + nt) * 16 + r) * 64] = acc[mt][1 - H][nt][r];
    __syncthreads();
    const float* take = xbuf + ((1 - H) * PW + pw) * XW + lane;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int n = n0 + (ng * TN + nt) * 32 + l31;
      const float bv = p.bias[n];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const int y = y0 + wy + mt;
        if (y >= p.H) continue;
        const size_t rowbase = ((size_t)img * p.H + y) * p.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + H;
          const float got = take[((mt * TN + nt) * 16 + r) * 64];
          float v = H == 0 ? (acc[mt][0][nt][r] + acc[mt][1][nt][r]) + got      // (m0 + m1) + m2
                           : (got - acc[mt][0][nt][r]) - acc[mt][1][nt][r];     // (m1 - m2) - m3
          v += bv;
          if (p.leaky) v = v > 0.f ? v : 0.2f * v;
          if (x < p.W) p.out[(rowbase + x) * p.ostride + n] = v;
        }
      }
    }
  };
  if (h == 0) finish(std::integral_constant<int, 0>{});
  else finish(std::integral_constant<int, 1>{});
}

template <int TH, int BN, int TM, int TN, int FLAGS>
hipError_t conv_winox3_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = 2 * 2 * (size_t)(TH + 2) * 4 * 32 * 32 + 3 * 4 * (size_t)BN * 32;
  constexpr int NT = (TH / TM) * (BN / (32 * TN)) * 128;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = conv_winox3_kernel<TH, BN, TM, TN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  const int ntx = (p.W + 63) / 64, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
  return hipGetLastError();
}
