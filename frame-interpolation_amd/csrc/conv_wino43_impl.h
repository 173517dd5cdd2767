// conv_wino43_impl.h -- 3x3 Conv2D('same') + bias + leaky_relu with the 1-D Winograd transform F(4,3) along x, fp32
// MFMA: per FOUR output pixels of a row 6 x 3 (nu, dy) matrix steps per K chunk instead of 36 (direct) or 24 (F(2,3),
// conv_wino_impl.h) - 2x / 1.33x fewer v_mfma_f32_32x32x2_f32.  The fp32 matrix pipe is power limited on this part
// (every large layer of the fp32 path lands at 105-128 TFLOP/s executed whatever the kernel, profiles/HISTORY.md 4.1), so fewer
// multiplies per output is what is left to buy time with.
//
// For an output row y and the pixel quad x = 4t .. 4t+3 with inputs d0..d5 = in[.][4t-1 .. 4t+4] (Lavin & Gray's
// F(4,3), points 0, +-1, +-2, inf):
//     v0 = 4 d0 - 5 d2 + d4            u0 = g0 / 4                          y0 = m0 + m1 + m2 + m3 + m4
//     v1 = (d4 - 4 d2) + (d3 - 4 d1)   u1 = -(g0 + g1 + g2) / 6             y1 = (m1 - m2) + 2 (m3 - m4)
//     v2 = (d4 - 4 d2) - (d3 - 4 d1)   u2 = -(g0 - g1 + g2) / 6             y2 = (m1 + m2) + 4 (m3 + m4)
//     v3 = (d4 - d2) + 2 (d3 - d1)     u3 = g0 / 24 + g1 / 12 + g2 / 6      y3 = (m1 - m2) + 8 (m3 - m4) + m5
//     v4 = (d4 - d2) - 2 (d3 - d1)     u4 = g0 / 24 - g1 / 12 + g2 / 6
//     v5 = 4 d1 - 5 d3 + d5            u5 = g2                              m_nu = sum_dy sum_c v_nu * u_nu
// fp32 throughout; on a K = 4608 test sum the rounding error is 3.9e-6 relative (F(2,3): 8.6e-7, direct: 6.4e-7).
//
//   * a workgroup owns TH rows x 4*QW pixels x BN output channels; QW = quads per patch row: 32 (128 pixels, one 32-row
//     MFMA tile per patch row), 16 (64 pixels, one MFMA tile per TWO patch rows: lanes 0-15 row r, 16-31 row r+1) or
//     8 (32 pixels, one MFMA tile per FOUR patch rows).  The QW = 16 tiles are 4-wave workgroups with 72 KB of LDS: TWO
//     per CU, whose barriers / fragment waits / epilogues are not in phase, and twice as many (half-size) workgroups per
//     layer for the tails of the 120-wide level.  The QW = 8 tiles (TH = 8: the same 256 pixels per workgroup, 66 KB) fit
//     the 15 * 2^k wide pyramid levels of a 960-wide tile exactly (480 = 15 x 32: no empty quads, where 64-pixel patches
//     leave 6.25 % of every patch row empty) and stage 10 halo rows per 8 output rows instead of 6 per 4;
//   * K chunks of 8 channels; the (TH+2) halo rows of a chunk are transformed ONCE on the way into LDS, image
//     [halo row][nu 6][quad QW][8 channels] (32-byte rows, K-halves swapped on bit 3 of the quad: conflict-free b128);
//   * the six nu planes are independent GEMMs.  NH = 2: wave half h accumulates nu = 3h .. 3h+2 for its TM rows x TN channel
//     tiles (TM*TN*3 accumulator tiles per wave) and the halves swap partial output sums through LDS in the epilogue.
//     NH = 1: a wave accumulates all six planes of ONE 32x32 tile (TM = TN = 1, the same 96 accumulator registers) and
//     forms y0..y3 in registers - no exchange, no epilogue barriers (12 instead of 9 fragment reads per 24 MFMAs);
//     the sums are written in the same order either way: every tile shape gives the same bits;
//   * weights [Cout][chunk][dy][nu][8] (192 contiguous bytes per dy stage and channel) through a 3-slot LDS ring
//     (slot = dy), requested three stages ahead in registers; one barrier per dy stage = 3*4*TM*TN MFMAs per wave;
//   * fragment registers triple buffered over the nu steps: the ds_reads of step s+1 are issued before the MFMAs of s.
#pragma once
#include "conv_buf_impl.h"

enum { W43_F_PF2 = 32768,      // activation loads requested TWO chunks ahead (second register set): ~4 stages of load-to-use distance
       W43_F_BG = 65536 };     // weight fragments straight from global memory (L1 / L2) into registers, requested two stages
                               // ahead: no weight ring in LDS (36 KB instead of 54 KB for the 32-channel tile = FOUR workgroups
                               // per CU), no weight stores, two barriers per chunk instead of three
// (Rounds 2-3 carried experiment flags here - wave priorities around the MFMA groups (-1..-3 %), persistent workgroups that request
// the next pair's loads in front of the epilogue (no gain), timing ablations - driven by tools/retired/conv_bench.hip; their
// measurements are in profiles/HISTORY.md 9 and profiles/r0[23]_*; the code went with round 4.)

template <int TH, int BN, int TM, int TN, int FLAGS, int QW = 32, int NH = 2>
__global__ __launch_bounds__(((TH * QW / 32) / TM) * (BN / (32 * TN)) * 64 * NH, ((FLAGS & W43_F_BG) != 0 && BN == 32 && QW <= 16) ? 4 : 2) void conv_wino43_kernel(ConvParams p) {   // 2 waves per SIMD: <= 256 VGPRs, two 4-wave workgroups per CU (W43_F_BG 32-channel tile: 4 -> <= 128 VGPRs, four per CU)
  constexpr int RPT = 32 / QW;                 // patch rows per 32-quad MFMA tile
  constexpr int MT = TH / RPT;                 // MFMA row tiles per patch; TM of them per wave
  constexpr int NU = 6 / NH;                   // nu planes per wave
  constexpr int RG = MT / TM, NG = BN / (32 * TN), PW = RG * NG, NW = NH * PW, NT = NW * 64;
  static_assert(NH == 2 || (NH == 1 && TM == 1 && TN == 1), "nu split");
  constexpr int HR = TH + 2;
  constexpr int A_PLANE = QW * 8;              // floats of one nu plane of a halo row
  constexpr int A_STAGE = HR * 6 * A_PLANE;    // floats: [hy][nu][quad][8]
  constexpr int B_PLANE = BN * 8;              // floats of one nu plane of a stage
  constexpr int B_STAGE = 6 * B_PLANE;
  constexpr int ITEMS = HR * QW * 2;           // (halo row, quad, 4-channel group)
  constexpr int BU = BN * 12;                  // float4 units of one weight stage
  constexpr int BLD = (BU + NT - 1) / NT;
  constexpr int PXW = 4 * QW;                  // patch width in pixels
  static_assert(QW == 32 || QW == 16 || QW == 8, "quads per patch row");
  static_assert(TH % RPT == 0 && MT % TM == 0 && BN % (32 * TN) == 0 && ITEMS <= NT && 2 * ITEMS > NT, "bad tile");
  constexpr bool BG = (FLAGS & W43_F_BG) != 0;
  static_assert(NH == 1 || 2 * PW * TM * TN * 16 * 64 <= 2 * A_STAGE + (BG ? 0 : 3 * B_STAGE), "exchange buffer does not fit");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [A0][A1][B ring x3]
  float* const Bsm = smem + 2 * A_STAGE;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int h = wv / PW, pw = wv % PW;     // nu half, pair-wave
  const int rg = pw / NG, ng = pw % NG;

  // ---- the (patch, channel block) pair of this workgroup, one of the NB * ntx * nty * (Cout / BN) pairs ------------------------
  const int ntx = (p.W + PXW - 1) / PXW, nty = (p.H + TH - 1) / TH;
  const int nbx = p.NB * ntx * nty, nby = p.Cout / BN;
  const int total = nbx * nby;
  const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  int img = 0, y0 = 0, x0 = 0, n0 = 0;
  auto decode = [&](int l) {
    int bx, by;
    if constexpr ((FLAGS & CONV_B_XCD_M) != 0) {   // pairs dealt to the eight XCDs as eight contiguous runs
      const int xcd = l & 7, idx = l >> 3;
      const int q = total >> 3, r = total & 7;
      const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      const int nl = base + idx;
      bx = nl / nby;
      by = nl - bx * nby;
    } else {
      by = l / nbx;
      bx = l - by * nbx;
    }
    img = bx / (ntx * nty);
    const int trem = bx - img * (ntx * nty);
    y0 = (trem / ntx) * TH;
    x0 = (trem % ntx) * PXW;
    n0 = by * BN;
  };
  decode(lin);

  // ---- the A staging item of this thread: (halo row hy, quad tq, channel group q); threads past the last item repeat
  // one (same values to the same LDS address) so that the staging code has no divergent branch -------------------------
  int f = t;
  if (f >= ITEMS) f -= ITEMS;
  const int aq = f & 1, tq = (f >> 1) % QW, ahy = (f >> 1) / QW;
  int a_x = 0;
  unsigned a_ok = 0;          // bit j: pixel a_x + j is inside the image (and the row is)
  auto setup_item = [&]() {
    const int a_y = y0 - 1 + ahy;
    a_x = x0 - 1 + 4 * tq;
    a_ok = 0;
    if (a_y >= 0 && a_y < p.H)
      for (int j = 0; j < 6; ++j)
        if (a_x + j >= 0 && a_x + j < p.W) a_ok |= 1u << j;
  };
  // K-half swap: QW >= 16 on bit 3 of the quad; QW = 8 (a 16-lane fragment read covers two halo rows of 8 quads) on the
  // parity of the halo row - either way 16 consecutive fragment lanes cover all 64 banks once
  const int a_lds = ((ahy * 6) * QW + tq) * 8 + ((aq ^ (QW >= 16 ? ((tq >> 3) & 1) : (ahy & 1))) << 2);   // float index of nu = 0
  const int scol = aq * 4;
  unsigned a_off = 0, a_pix = 0;
  conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);
  int sg = 0, c0 = 0, segC = p.seg[0].C;
  auto setup_seg = [&]() {
    const ConvSeg& s = p.seg[sg];
    segC = s.C;
    a_pix = (unsigned)s.stride * 4u;
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
    // The buffer resource starts at the first halo row of THIS patch (64-bit pointer arithmetic, uniform in the
    // workgroup); the 32-bit lane offsets only span the patch's TH + 2 rows, so a segment may be larger than 4 GiB (the
    // level-0 buffers of an untiled 4K frame).  Both may point outside the tensor: only dereferenced under a_ok.
    arsrc = conv_make_rsrc(s.ptr + ((long long)be * p.H + (y0 - 1)) * p.W * s.stride);
    a_off = (unsigned)((ahy * p.W + a_x) * s.stride + scol) * 4u;
  };

  // ---- B staging ---------------------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 8;
  const int nstage = nkc * 3;    // (chunk, dy)
  // split-K (ConvParams::ksplit, film_kernels.h): blockIdx.z = split s sums the K chunks [kbeg, kend) and writes raw partial
  // sums to part[s][pixel][Cout]; conv_splitk_reduce_kernel adds them in split order with the bias and the activation
  const int ksp = p.ksplit > 1 ? p.ksplit : 1;
  const int kbeg = (int)((long long)nkc * blockIdx.z / ksp), kend = (int)((long long)nkc * (blockIdx.z + 1) / ksp);
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int u = t + NT * i;
    const bool slot = u < BU;
    const int kb = u & 1, row = slot ? (u >> 1) % BN : 0, nu = slot ? (u >> 1) / BN : 0;
    boff[i] = (unsigned)((size_t)row * nstage * 192 + nu * 32 + kb * 16);   // + wbase (the channel block, uniform) in the scalar offset
    blds[i] = slot ? nu * B_PLANE + row * 8 + ((kb ^ ((row >> 3) & 1)) << 2) : -1;
  }

  constexpr bool PF2 = (FLAGS & W43_F_PF2) != 0;
  bf4 araw[PF2 ? 2 : 1][6];   // PF2: chunk parity -> register set
  bf4 breg[BG ? 1 : 3][BLD];   // weights in flight: requested in stage s for stage s+3, written to the ring in stage s+1
  // BG: the B fragments of this wave, [stage mod 3][nu step][channel tile]: lane (cout = l31, K half) reads the 16 bytes
  // [cout][stage][nu][4 half .. 4 half + 3] of the weight image, requested in stage s - 2
  bf4 fbg[BG ? 3 : 1][BG ? NU : 1][TN];
  unsigned bgoff[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt)
    bgoff[nt] = (unsigned)((size_t)((ng * TN + nt) * 32 + l31) * nstage * 192 + (NU * h) * 32 + half * 16);
  unsigned wbase = 0;   // byte offset of the channel block's weights: n0 * nstage * 192 (set per pair)
  auto load_bg = [&](int s, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const unsigned so = wbase + (unsigned)(s < nstage ? s : nstage - 1) * 192u;
#pragma unroll
    for (int j = 0; j < (BG ? NU : 1); ++j)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) fbg[BG ? SET : 0][j][nt] = conv_buf_load(brsrc, bgoff[nt] + (unsigned)j * 32u, so);
  };
  bool chunk_ok = true;
  auto load_item = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool ok = chunk_ok && ((a_ok >> j) & 1u);
      araw[SET][j] = conv_buf_load(arsrc, ok ? a_off + (unsigned)j * a_pix : OOB, so);
    }
  };
  auto store_item = [&](int stage, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    float* As = smem + stage * A_STAGE + a_lds;
    bf4 v[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d0 = araw[SET][0][c], d1 = araw[SET][1][c], d2 = araw[SET][2][c], d3 = araw[SET][3][c], d4 = araw[SET][4][c], d5 = araw[SET][5][c];
      const float t1 = __builtin_fmaf(-4.f, d2, d4), t2 = __builtin_fmaf(-4.f, d1, d3);
      const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
      v[0][c] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
      v[1][c] = t1 + t2;
      v[2][c] = t1 - t2;
      v[3][c] = t3 + t4;
      v[4][c] = t3 - t4;
      v[5][c] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    }
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<bf4*>(As + nu * A_PLANE) = v[nu];   // nu planes: QW rows x 8 floats apart
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= kend) { chunk_ok = false; return; }
    c0 += 8;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto load_b = [&](int s, int buf) {
    const unsigned so = wbase + (unsigned)(s < nstage ? s : nstage - 1) * 192u;
#pragma unroll
    for (int i = 0; i < BLD; ++i) breg[buf][i] = conv_buf_load(brsrc, boff[i], so);
  };
  auto store_b = [&](int ring, int buf) {
    float* Bs = Bsm + ring * B_STAGE;
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      if (NT * (i + 1) <= BU || blds[i] >= 0) *reinterpret_cast<bf4*>(Bs + blds[i]) = breg[buf][i];
  };

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  const int sb = kbeg * 3;
  // per pair (decode() done): staging state and EVERY load request of the prologue - chunk 0 (PF2: and chunk 1) of the
  // activations, the first weight stages; the LDS stores follow at the top of the pair loop (one load latency, not three)
  auto begin_pair_a = [&]() {
    setup_item();
    sg = 0; c0 = 0; chunk_ok = true;
    for (int skip = kbeg * 8; skip > 0;) {   // first chunk of this split: walk the concat segments
      const int cseg = p.seg[sg].C;
      if (skip >= cseg) { skip -= cseg; ++sg; } else { c0 = skip; skip = 0; }
    }
    setup_seg();
    load_item(C0{});
  };
  auto begin_pair_b = [&]() {
    wbase = (unsigned)n0 * (unsigned)nstage * 192u;
    if constexpr (BG) {
      load_bg(sb + 0, C0{});
      load_bg(sb + 1, C1{});
    } else {
      load_b(sb + 0, 0);
      load_b(sb + 1, 1);
      load_b(sb + 2, 2);
    }
    if constexpr (PF2) { next_chunk(kbeg + 1); load_item(C1{}); }   // chunk 1 stays in registers until chunk 0's dy = 1 stage
  };
  begin_pair_a();
  begin_pair_b();

  const int cimg = img, cy0 = y0, cx0 = x0, cn0 = n0;
  f32x16 acc[TM][NU][TN];   // [row][nu - NU*h][channel tile]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < NU; ++v)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][v][j][r] = 0.f;

  // ---- fragment addresses in float4 units: row * 2 + (K-half ^ bit 3 of the row) -------------------------------------
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  constexpr int A_STAGE4 = A_STAGE / 4, B_STAGE4 = B_STAGE / 4, B_PLANE4 = B_PLANE / 4;
  const int wm = rg * TM;                    // first MFMA row tile of this wave
  const int lrow = l31 / QW, lq = l31 % QW;  // lane -> (patch row inside the tile, quad)
  const int swb = (l31 >> 3) & 1;            // = bit 3 of the quad (QW >= 16) / parity of the tile row (QW = 8), and bit 3 of the channel
  // QW = 8: the halo row read at (mt, dy) is wm * 4 + lrow + mt * 4 + dy, parity (lrow + dy) & 1: odd dy flips the low bit
  const int a_ad = (((wm * RPT + lrow) * 6 + NU * h) * QW + lq) * 2 + (half ^ swb);     // + ((mt * RPT + dy) * 6 + j) * QW * 2, stage
  const int b_ad = 2 * A_STAGE4 + (NU * h) * B_PLANE4 + (ng * TN * 32 + l31) * 2 + (half ^ swb);
  int a_cur = a_ad;

  bf4 fa[3][TM], fb[3][TN];   // [nu step j mod 3][tile]: triple buffered
  auto fetch = [&](auto dy_c, auto j_c, int a_base) {
    constexpr int DY = decltype(dy_c)::value, J = decltype(j_c)::value;
    const int ab = (QW == 8 && (DY & 1)) ? (a_base ^ 1) : a_base;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) fa[J % 3][mt] = smem4[ab + ((mt * RPT + DY) * 6 + J) * (QW * 2)];
    if constexpr (!BG) {
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) fb[J % 3][nt] = smem4[b_ad + DY * B_STAGE4 + J * B_PLANE4 + nt * 64];
    }
  };
  auto compute = [&](auto dy_c, auto j_c) {
    constexpr int DY = decltype(dy_c)::value, J = decltype(j_c)::value;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][J][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[J % 3][mt][k], BG ? fbg[BG ? DY : 0][BG ? J : 0][nt][k] : fb[J % 3][nt][k], acc[mt][J][nt], 0, 0, 0);
  };

  // ---- pipeline ----------------------------------------------------------------------------------------------------------
  a_cur = a_ad;
  store_item(0, C0{});
  if constexpr (!BG) {
    store_b(0, 0);
    store_b(1, 1);
  }
  next_chunk(kbeg + (PF2 ? 2 : 1));
  __syncthreads();
  fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, a_cur);
  int a_stage = 0;
  // one K chunk; PAR = parity of kc (PF2: the register set that receives chunk kc + 2 and held chunk kc)
  auto chunk = [&](int kc, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const int s0 = kc * 3;
    const int a_next = a_ad + (a_stage ^ 1) * A_STAGE4;
    auto stage = [&](auto dy_c) {
      constexpr int DY = decltype(dy_c)::value;
      if constexpr (BG) load_bg(s0 + DY + 2, std::integral_constant<int, (DY + 2) % 3>{});
      else load_b(s0 + DY + 3, DY);
      if constexpr (DY == 0) load_item(std::integral_constant<int, PF2 ? PAR : 0>{});
      fetch(dy_c, std::integral_constant<int, 1>{}, a_cur);
      compute(dy_c, std::integral_constant<int, 0>{});
      fetch(dy_c, std::integral_constant<int, 2>{}, a_cur);
      compute(dy_c, std::integral_constant<int, 1>{});
      if constexpr (NU == 6) {
        fetch(dy_c, std::integral_constant<int, 3>{}, a_cur);
        compute(dy_c, std::integral_constant<int, 2>{});
        fetch(dy_c, std::integral_constant<int, 4>{}, a_cur);
        compute(dy_c, std::integral_constant<int, 3>{});
        fetch(dy_c, std::integral_constant<int, 5>{}, a_cur);
        compute(dy_c, std::integral_constant<int, 4>{});
      }
      fetch(std::integral_constant<int, (DY + 1) % 3>{}, std::integral_constant<int, 0>{}, DY == 2 ? a_next : a_cur);
      compute(dy_c, std::integral_constant<int, NU - 1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!BG) store_b((DY + 2) % 3, (DY + 2) % 3);
      if constexpr (DY == 1) store_item(a_stage ^ 1, std::integral_constant<int, PF2 ? 1 - PAR : 0>{});
      // BG: only the activation double buffer is shared.  It is written in the dy = 1 stage: the barrier of dy = 0 puts every
      // wave's reads of the previous chunk in front of that write, the barrier of dy = 1 puts the write in front of the
      // first read (the fragment prefetch at the end of dy = 2)
      if constexpr (!BG || DY != 2) __syncthreads();
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    next_chunk(kc + (PF2 ? 3 : 2));
    a_stage ^= 1;
    a_cur = a_next;
  };
  if constexpr (PF2) {
    // unconditional pairs + a peeled last chunk: with `if (kc + 1 < kend)` between the two chunks the compiler sizes every
    // s_waitcnt of the loop for the path on which the odd chunk's requests were never issued (fewer in flight = less lookahead)
    int kc = kbeg;
    for (; kc + 1 < kend; kc += 2) {
      chunk(kc, C0{});
      chunk(kc + 1, C1{});
    }
    if (kc < kend) chunk(kc, C0{});
  } else {
    for (int kc = kbeg; kc < kend; ++kc) chunk(kc, C0{});
  }

  if constexpr (BG && NH == 2) __syncthreads();   // no barrier behind the last dy = 2 stage: the exchange below reuses the activation buffers

  // ---- epilogue: y0 = (m0+m1+m2) + (m3+m4), y1 = (m1-m2) + 2(m3-m4) on half 0; y2 = (m1+m2) + 4(m3+m4),
  // y3 = (m1-m2) + (8(m3-m4) + m5) on half 1.  The halves swap the bracketed sums they lack through LDS (the staging
  // buffers are free after the last barrier), one tile per round.  C/D layout of the 32x32 MFMA: col = lane&31 (cout),
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5) = quad.
  float* const xbuf = smem;
  constexpr int XW = TM * TN * 16 * 64;   // floats one wave gives per round
  auto finish = [&](auto h_c) {
    constexpr int H = decltype(h_c)::value;
    float* const give = xbuf + (H * PW + pw) * XW + lane;
    const float* const take = xbuf + ((1 - H) * PW + pw) * XW + lane;
    float keep[TM][TN][16];   // fused pool: the activated outputs of round 0
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (round) __syncthreads();   // everybody has consumed round 0
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ma = acc[mt][1][nt][r], mb = acc[mt][2][nt][r];      // H = 0: m1, m2;  H = 1: m4, m5
            const float m0 = acc[mt][0][nt][r];                                //        m0           m3
            float g;
            if (H == 0) g = round == 0 ? ma + mb : ma - mb;                    // (m1 + m2) / (m1 - m2)
            else g = round == 0 ? m0 + ma : 2.f * (m0 - ma);                   // (m3 + m4) / 2 (m3 - m4)
            give[((mt * TN + nt) * 16 + r) * 64] = g;
          }
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int n = cn0 + (ng * TN + nt) * 32 + l31;
        const float bv = p.bias[n];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
          float val[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;      // row of the 32-row MFMA tile
            const int y = cy0 + (wm + mt) * RPT + mrow / QW;
            const int x = cx0 + 4 * (mrow % QW) + 2 * H + round;
            const float got = take[((mt * TN + nt) * 16 + r) * 64];
            const float ma = acc[mt][1][nt][r], mb = acc[mt][2][nt][r], m0 = acc[mt][0][nt][r];
            float v;
            if (H == 0) v = round == 0 ? ((m0 + ma) + mb) + got : (ma - mb) + got;                       // y0, y1
            else v = round == 0 ? got + 4.f * (m0 + ma) : got + (8.f * (m0 - ma) + mb);                  // y2, y3
            if (ksp > 1) {   // split-K: raw partial sum; bias, activation and the sum over the splits in the reduce kernel
              if (y < p.H && x < p.W) p.part[((size_t)blockIdx.z * p.M + ((size_t)cimg * p.H + y) * p.W + x) * p.Cout + n] = v;
              continue;
            }
            v += bv;
            if (p.leaky) v = v > 0.f ? v : 0.2f * v;
            val[r] = v;
            if (y < p.H && x < p.W) p.out[(((size_t)cimg * p.H + y) * p.W + x) * p.ostride + n] = v;
          }
          if constexpr (QW <= 16) {
            // fused 2x2 average pool: this lane holds patch rows 2k (register r) and 2k + 1 (register r + QW / 2) of quad
            // q = mrow % QW at x = 4q + 2H (round 0, kept) and x + 1 (round 1)
            if (p.pool_out != nullptr && ksp == 1) {
              if (round == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) keep[mt][nt][r] = val[r];
              } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  if ((r / (QW / 2)) & 1) continue;      // the odd row of a pair
                  const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
                  const int q = mrow % QW;
                  const int yp = (cy0 >> 1) + (((wm + mt) * RPT + mrow / QW) >> 1);
                  const int xp = (cx0 >> 1) + 2 * q + H;
                  const float pv = (((keep[mt][nt][r] + val[r]) + keep[mt][nt][r + QW / 2]) + val[r + QW / 2]) * 0.25f;
                  if (2 * yp < p.H && 2 * xp < p.W)
                    p.pool_out[(((size_t)cimg * (p.H >> 1) + yp) * (p.W >> 1) + xp) * p.pool_ostride + n] = pv;
                }
              }
            }
          }
        }
      }
    }
  };
  if constexpr (NH == 2) {
    if (h == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  } else {
    // all six planes in this wave: the same sums, in the same order, as the two-half exchange above
    const int n = cn0 + ng * 32 + l31;
    const float bv = p.bias[n];
    auto outputs = [&](int r, float* o) {
      const float m0 = acc[0][0][0][r], m1 = acc[0][1][0][r], m2 = acc[0][2][0][r], m3 = acc[0][3][0][r], m4 = acc[0][4][0][r],
                  m5 = acc[0][5][0][r];
      o[0] = ((m0 + m1) + m2) + (m3 + m4);
      o[1] = (m1 - m2) + 2.f * (m3 - m4);
      o[2] = (m1 + m2) + 4.f * (m3 + m4);
      o[3] = (m1 - m2) + (8.f * (m3 - m4) + m5);
      const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int y = cy0 + wm * RPT + mrow / QW;
      const int x = cx0 + 4 * (mrow % QW);
      if (ksp > 1) {   // split-K: raw partial sums
        if (y < p.H) {
          const size_t pix = ((size_t)cimg * p.H + y) * p.W + x;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (x + j < p.W) p.part[((size_t)blockIdx.z * p.M + pix + j) * p.Cout + n] = o[j];
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] += bv;
        if (p.leaky) o[j] = o[j] > 0.f ? o[j] : 0.2f * o[j];
      }
      if (y < p.H) {
        const size_t rowbase = ((size_t)cimg * p.H + y) * p.W;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x + j < p.W) p.out[(rowbase + x + j) * p.ostride + n] = o[j];
      }
    };
    if (QW <= 16 && BN == 64 && p.pw_out != nullptr) {
      // fused 1x1 convolution: the activated tile goes to LDS as [pixel = patch row * PXW + x][65] (the K loop ended with a
      // barrier: the staging buffers are free), then thread = pixel sums its 64 channels
      static_assert(NH != 1 || QW > 16 || BN != 64 || (size_t)TH * PXW * 65 <= 2 * A_STAGE + 3 * B_STAGE, "1x1 tile does not fit");
      float* const tile = smem;
      if constexpr (BG) __syncthreads();   // (no barrier behind the last stage of the K loop)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][0][0][r], m1 = acc[0][1][0][r], m2 = acc[0][2][0][r], m3 = acc[0][3][0][r], m4 = acc[0][4][0][r],
                    m5 = acc[0][5][0][r];
        float o[4];
        o[0] = ((m0 + m1) + m2) + (m3 + m4);
        o[1] = (m1 - m2) + 2.f * (m3 - m4);
        o[2] = (m1 + m2) + 4.f * (m3 + m4);
        o[3] = (m1 - m2) + (8.f * (m3 - m4) + m5);
        const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int pxl = (wm * RPT + mrow / QW) * PXW + 4 * (mrow % QW);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = o[j] + bv;
          if (p.leaky) v = v > 0.f ? v : 0.2f * v;
          tile[(pxl + j) * 65 + ng * 32 + l31] = v;
        }
      }
      __syncthreads();
      if (t < TH * PXW) {
        const int y = cy0 + t / PXW, x = cx0 + (t % PXW);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const float* row = tile + t * 65;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
          const float v = row[c];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < p.pw_cout) a[j] = __builtin_fmaf(v, p.pw_w[c * p.pw_cout + j], a[j]);
        }
        if (y < p.H && x < p.W) {
          float* d = p.pw_out + (((size_t)cimg * p.H + y) * p.W + x) * p.pw_ostride;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < p.pw_cout) d[j] = a[j] + p.pw_bias[j];
        }
      }
    } else if (QW <= 16 && p.pool_out != nullptr && ksp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {      // patch rows 2k (r) and 2k + 1 (r + QW / 2) of the same quad
        if ((r / (QW / 2)) & 1) continue;
        float t[4], bq[4];
        outputs(r, t);
        outputs(r + QW / 2, bq);
        const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int q = mrow % QW;
        const int yp = (cy0 >> 1) + ((wm * RPT + mrow / QW) >> 1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int xp = (cx0 >> 1) + 2 * q + e;
          const float pv = (((t[2 * e] + t[2 * e + 1]) + bq[2 * e]) + bq[2 * e + 1]) * 0.25f;
          if (2 * yp < p.H && 2 * xp < p.W) p.pool_out[(((size_t)cimg * (p.H >> 1) + yp) * (p.W >> 1) + xp) * p.pool_ostride + n] = pv;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o[4];
        outputs(r, o);
      }
    }
  }
}

template <int TH, int BN, int TM, int TN, int FLAGS, int QW = 32, int NH = 2>
hipError_t conv_wino43_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (2 * (size_t)(TH + 2) * 6 * QW * 8 + ((FLAGS & W43_F_BG) != 0 ? 0 : 3 * 6 * (size_t)BN * 8)) * sizeof(float);
  constexpr int NT = ((TH * QW / 32) / TM) * (BN / (32 * TN)) * 64 * NH;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = conv_wino43_kernel<TH, BN, TM, TN, FLAGS, QW, NH>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  const int ntx = (p.W + 4 * QW - 1) / (4 * QW), nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
  return hipGetLastError();
}
