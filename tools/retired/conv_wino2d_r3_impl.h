// conv_wino2d_r3_impl.h -- RETIRED (round 4: replaced by csrc/conv_wino2d_impl.h; kept for tools/w2d_bench.hip, which checks the new
// K loop against these bits).  Round 3's kernel: 3x3 Conv2D('same') + bias + leaky_relu with a NESTED Winograd transform, fp32 MFMA:
// F(4,3) along x (as conv_wino43_impl.h) x F(2,3) along y.  Per output UNIT of 2 rows x 4 pixels and per (ci, co):
// 4 (mu) x 6 (nu) = 24 multiplies for 8 outputs = 3 per output, where the 1-D F(4,3) form spends 18 / 4 = 4.5 and the
// direct convolution 9: 1.5x fewer v_mfma_f32_32x32x2_f32 than conv_wino43_kernel.
//
//   d (4 rows x 6 pixels)  --x: F(4,3) B^T per row-->  v[r][nu]  --y: F(2,3) B^T-->  V[mu][nu]:
//       V[0] = v[0] - v[2]     V[1] = v[1] + v[2]     V[2] = v[2] - v[1]     V[3] = v[1] - v[3]
//   weights  U[mu][nu]: u_nu(dy) = the F(4,3) transform of kernel row dy (as in conv_wino43_impl.h), then along dy
//       U[0] = u(0)    U[1] = ((u(0) + u(2)) + u(1)) / 2    U[2] = ((u(0) + u(2)) - u(1)) / 2    U[3] = u(2)
//   M[mu][nu] = sum_ci V[mu][nu] U[mu][nu]  (24 independent GEMMs),  x inverse per mu (conv_wino43's y0..y3), then
//       row 2k = (M'[0] + M'[1]) + M'[2]      row 2k+1 = (M'[1] - M'[2]) - M'[3]
//
// Mapping (what makes it affordable - the 2-D form was sized and rejected twice because the activation staging per MFMA
// grows; here it does not):
//   * the activation staging is conv_wino43_kernel's, unchanged: the x-transformed halo rows of a K chunk go to LDS once,
//     [halo row][nu 6][quad][8] - 10 rows per 8 output rows with the 32-pixel x 8-row patch;
//   * the y transform costs NO extra staging: a wave owns ONE mu (and 32 output channels, all six nu planes: 96 accumulator
//     registers) and forms its A fragment as (row a) +- (row b) of that LDS image on the way into the MFMA - two
//     ds_read_b128 and four v_fma per four MFMAs;
//   * a (mu, channel tile) weight slab has exactly one consumer wave, so the weights skip LDS: [Cout / 32][chunk][mu][nu]
//     [K half][32 channels][4] in memory = one fully coalesced 1 KB read per (mu, nu) step, requested a chunk ahead;
//   * one barrier per chunk (24 MFMAs per wave); the epilogue exchanges the mu planes through LDS in four rounds.
// fp32 throughout.  A different summation family from the 1-D kernels (not bit-identical to them).
#pragma once
#include "../../frame-interpolation_amd/csrc/conv_buf_impl.h"

enum { W2R_F_PFA = 128,       // touch-ahead for the activations: per chunk every staging thread reads ONE dword of the next 128-B line of
                              // one of its pixels, issued after the chunk's real loads (in-order vmcnt: it has two chunk times to
                              // land) - the item loads three to six chunks later then hit L2 instead of waiting for HBM
       W2R_F_PFB = 8192,      // the same for the weight slab of chunk kc + 3 (48 lines of 128 B: one dword load on 48 lanes)
       W2R_DBG_AHOT = 4096,   // timing ablation: every activation load from the first 64 KB of the tensor (same requests, all cache hits)
       W2R_DBG_NOAST = 32768, // timing ablation: item loads issued, no transform / LDS store
       W2R_DBG_NOALD = 65536, // timing ablation: transform + LDS store of whatever the registers hold, no item loads
       W2R_F_RAW = 131072,    // the halo patch goes to LDS RAW first: buffer_load_dwordx4 ... lds, 16 channels (one 64-B sector per pixel) per
                              // request, 16 pixels per instruction, every wave issuing its share; the staging threads then read their
                              // six pixels from LDS instead of gathering 16 B per lane from memory (3.5x fewer, fully used sectors
                              // requested; no activation registers in flight).  Needs every input segment's C % 16 == 0
       W2R_F_B2 = 262144,     // weight slabs requested TWO chunks ahead in the same two register sets: slab j of chunk kc + 2 goes into
                              // the registers of slab j of chunk kc as soon as its four MFMAs are issued
       W2R_F_GRP256 = 8, W2R_F_GRP128 = 16, W2R_F_GRP512 = 32,   // block order: groups of 256 / 128 / 512 patches, inside a group one
                              // channel block after the other (plain order = one group of ALL patches: the activations are re-read from
                              // HBM once per channel block; a group small enough for the 256-MB Infinity Cache re-reads them from there)
       W2R_F_MIDBAR = 524288, // (with ILV and B2) THREE activation stages and the chunk's one barrier behind nu step 3 instead of behind step 5:
                              // the next stage is complete and published two steps before the chunk ends, so the first fragment
                              // of chunk kc + 1 is read during step 5 of chunk kc - no LDS round trip behind the barrier, where both
                              // waves of a SIMD would sit it out together (64-channel tile: they belong to the same workgroup)
       W2R_F_LATE = 16384,    // with W2R_F_ILV: the transform sits on nu steps 2..5 instead of 0..3 - the item loads (the LAST requests of the
                              // previous chunk) get another half chunk before the wave waits for them
       W2R_F_ILV = 64,        // the transform + LDS stores of the next chunk's item are spread over the nu steps of the MFMA loop (in the
                              // gaps between MFMA groups) instead of sitting between the last MFMA and the barrier
       W2R_DBG_NOB = 256,     // timing ablations (tools/conv_bench.hip only; results are wrong on purpose): no weight loads in the K loop
       W2R_DBG_NOCOMB = 512,  // no second fragment read / no y combine
       W2R_DBG_NOA = 1024,    // no activation staging in the K loop
       W2R_DBG_NOBAR = 2048 };// no barrier in the K loop

template <int TH, int BN, int FLAGS, int QW = 8>
__global__ __launch_bounds__(4 * (BN / 32) * 64, (BN == 32) ? 2 : 1) void conv_wino2d_r3_kernel(ConvParams p) {
  constexpr int RPT = 32 / QW;                 // patch rows of UNITS per 32-unit MFMA tile (unit = 2 rows x 4 pixels)
  static_assert(QW == 8 || QW == 16, "quads per patch row");
  static_assert(TH == 2 * RPT, "one 32-unit MFMA row tile per workgroup: TH = 2 * 32 / QW");
  constexpr int NG = BN / 32, NW = 4 * NG, NT = NW * 64;
  constexpr int HR = TH + 2;
  constexpr int A_PLANE = QW * 8;              // floats of one nu plane of a halo row
  constexpr int A_STAGE = HR * 6 * A_PLANE;    // floats: [hy][nu][quad][8]
  constexpr int ITEMS = HR * QW * 2;           // (halo row, quad, 4-channel group)
  constexpr int PXW = 4 * QW;
  static_assert(ITEMS <= NT, "staging items");
  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr bool RAW = (FLAGS & W2R_F_RAW) != 0;
  constexpr bool MID = (FLAGS & W2R_F_MIDBAR) != 0;
  constexpr int NST = MID ? 3 : 2;             // activation stages
  static_assert(!MID || ((FLAGS & W2R_F_ILV) && (FLAGS & W2R_F_B2)), "W2R_F_MIDBAR needs W2R_F_ILV and W2R_F_B2");
  constexpr int PW = PXW + 2;                  // halo pixels per patch row
  constexpr int NPIX = HR * PW;
  constexpr int NI = (NPIX + 15) / 16;         // raw requests (16 pixels x 64 B = 1 KB of LDS each) per 16-channel super-chunk
  constexpr int IPW = (NI + NW - 1) / NW;      // per wave
  constexpr int R_STAGE = IPW * NW * 256;      // floats of one raw buffer (every wave issues IPW requests; those past NI write
                                               // zeros behind the patch); LDS: [A0][A1][R0][R1]
  static_assert(!RAW || QW == 8, "raw staging: 8-quad patch rows");

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [A0][A1]; the epilogue reuses it as the exchange buffer

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mu = wv & 3, ng = wv >> 2;

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
  if constexpr ((FLAGS & (W2R_F_GRP256 | W2R_F_GRP128 | W2R_F_GRP512)) != 0) {
    constexpr int G = (FLAGS & W2R_F_GRP256) ? 256 : (FLAGS & W2R_F_GRP128) ? 128 : 512;
    const int nbx = gridDim.x, nby = gridDim.y;
    const int lin = by * nbx + bx;
    const int g = lin / (G * nby), r = lin - g * (G * nby);
    const int gl = nbx - g * G < G ? nbx - g * G : G;
    by = r / gl;
    bx = g * G + (r - by * gl);
  }
  const int ntx = (p.W + PXW - 1) / PXW, nty = (p.H + TH - 1) / TH;
  const int img = bx / (ntx * nty);
  const int trem = bx - img * (ntx * nty);
  const int y0 = (trem / ntx) * TH, x0 = (trem % ntx) * PXW;
  const int n0 = by * BN;

  // ---- A staging: conv_wino43_kernel's item (halo row hy, quad tq, channel group q) on the first ITEMS threads ------------
  const bool stager = t < ITEMS;
  const int f = stager ? t : 0;
  const int aq = f & 1, tq = (f >> 1) % QW, ahy = (f >> 1) / QW;
  const int a_y = y0 - 1 + ahy, a_x = x0 - 1 + 4 * tq;
  unsigned a_ok = 0;
  if (stager && a_y >= 0 && a_y < p.H)
    for (int j = 0; j < 6; ++j)
      if (a_x + j >= 0 && a_x + j < p.W) a_ok |= 1u << j;
  // K-half swap so that a 16-lane fragment read covers all 64 banks once: QW >= 16 on bit 3 of the quad; QW = 8 (16 lanes = two
  // unit rows, i.e. halo rows TWO apart) on bit 1 of the halo row (bit 0 would give both rows the same half: measured as 35 %
  // of the LDS cycles in bank conflicts, profiles/r03_pmc_conv_bench_w2d.md)
  const int a_lds = ((ahy * 6) * QW + tq) * 8 + ((aq ^ (QW >= 16 ? ((tq >> 3) & 1) : ((ahy >> 1) & 1))) << 2);
  const int scol = aq * 4;
  unsigned a_off = 0, a_pix = 0;
  unsigned pfo[2] = {OOB, OOB};   // W2R_F_PFA: the pixel this thread touches ahead on even / odd chunks (pixels 1..4 of the quad between
                                  // the two channel-group threads and the two chunk parities: every pixel of the row once per two chunks)
  unsigned aoffj[6];   // byte offsets of the item's six pixels (out of range where the pixel is outside the image): fixed per
                       // segment, so that a chunk's loads need no address arithmetic and the registers stay theirs
  conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);
  int sg = 0, c0 = 0, segC = p.seg[0].C;
  auto setup_seg = [&]() {
    const ConvSeg& s = p.seg[sg];
    segC = s.C;
    a_pix = (unsigned)s.stride * 4u;
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
    arsrc = conv_make_rsrc(s.ptr + ((long long)be * p.H + (y0 - 1)) * p.W * s.stride);
    a_off = (unsigned)((ahy * p.W + a_x) * s.stride + scol) * 4u;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      aoffj[j] = ((a_ok >> j) & 1u) ? a_off + (unsigned)j * a_pix : OOB;
      if constexpr ((FLAGS & W2R_DBG_AHOT) != 0) aoffj[j] = ((a_ok >> j) & 1u) ? (aoffj[j] & 0xFFF0u) + (unsigned)(p.W * s.stride) * 4u : OOB;
      asm volatile("" : "+v"(aoffj[j]));   // keep it in its register (hipcc otherwise recomputes it per chunk into registers that
                                           // are still the destination of loads in flight, and has to wait for those)
    }
    if constexpr ((FLAGS & W2R_F_PFA) != 0) {
      pfo[0] = aq ? aoffj[3] : aoffj[1];
      pfo[1] = aq ? aoffj[4] : aoffj[2];
      asm volatile("" : "+v"(pfo[0]));
      asm volatile("" : "+v"(pfo[1]));
    }
  };
  const int nkc = p.Ctot / 8;
  bf4 araw[2][6];
  bool chunk_ok = true;
  auto load_item = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const unsigned so = (FLAGS & W2R_DBG_AHOT) ? ((unsigned)c0 * 4u) & 96u : (unsigned)c0 * 4u;
#pragma unroll
    for (int j = 0; j < 6; ++j) araw[SET][j] = conv_buf_load(arsrc, aoffj[j], so);
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
    for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<bf4*>(As + nu * A_PLANE) = v[nu];
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= nkc) {   // past the last chunk: the remaining (prefetch) loads read nothing
      chunk_ok = false;
#pragma unroll
      for (int j = 0; j < 6; ++j) aoffj[j] = OOB;
      pfo[0] = pfo[1] = OOB;
      return;
    }
    c0 += 8;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };

  // ---- W2R_F_RAW: raw halo patch in LDS.  Request i of a super-chunk covers linear halo pixels P = 16 i .. 16 i + 15 (P = halo
  // row * PW + pixel), lane l -> LDS slot l of the request's 1 KB (that is how `buffer_load ... lds` places the lanes).  The
  // slot <-> (pixel q = P & 15, 16-B piece c = 2 h + a) map is chosen for the READ side: the 16 lanes of a ds_read_b128
  // group are the items (a, quad 0..7) of one halo row = pixels FOUR apart, same h; slot = (q & 3) * 16 + (h ^ (i & 1)) * 8 +
  // (q >> 2) * 2 + a gives them 16 different slots mod 16 (all 64 banks once).
  unsigned rvoff[IPW];
  int rsg = 0, rc0 = 0, rsegC = 0;
  conv_rsrc_t rrsrc = conv_make_rsrc(p.seg[0].ptr);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  unsigned rbad = 0;   // bit n: request n of this lane is outside the image (or past the patch)
  auto raw_pixel = [&](int n, int& pc) {   // this lane's (pixel of the patch row block, piece) of its wave's n-th request
    const int i = wv + NW * n;
    const int q = ((lane >> 1) & 3) * 4 + (lane >> 4), h = ((lane >> 3) & 1) ^ (i & 1);
    pc = 2 * h + (lane & 1);
    return 16 * i + q;
  };
  if constexpr (RAW) {
#pragma unroll
    for (int n = 0; n < IPW; ++n) {
      int pc;
      const int P = raw_pixel(n, pc), hy = P / PW, px = P - hy * PW;
      const int y = y0 - 1 + hy, x = x0 - 1 + px;
      if (!(P < NPIX && y >= 0 && y < p.H && x >= 0 && x < p.W)) rbad |= 1u << n;
    }
  }
  auto raw_setup_seg = [&]() {
    const ConvSeg& s = p.seg[rsg];
    rsegC = s.C;
    int be = img + s.boff;
    if (s.bmod && be >= s.bmod) be -= s.bmod;
    rrsrc = conv_make_rsrc(s.ptr + ((long long)be * p.H + (y0 - 1)) * p.W * s.stride);
#pragma unroll
    for (int n = 0; n < IPW; ++n) {
      int pc;
      const int P = raw_pixel(n, pc), hy = P / PW, px = P - hy * PW;
      rvoff[n] = ((unsigned)((hy * p.W + (x0 - 1 + px)) * s.stride + pc * 4) * 4u) | (0u - ((rbad >> n) & 1u));
      asm volatile("" : "+v"(rvoff[n]));
    }
  };
  auto raw_issue = [&](int buf) {   // the next super-chunk (16 channels) of the patch -> raw buffer `buf`; advances the raw cursor
    const unsigned so = (unsigned)rc0 * 4u;
    const unsigned base = lds0 + (unsigned)(NST * A_STAGE + buf * R_STAGE) * 4u + (unsigned)wv * 1024u;
#pragma unroll
    for (int n = 0; n < IPW; ++n)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(base + (unsigned)(NW * n) * 1024u), "v"(rvoff[n]), "s"(rrsrc), "s"(so)
                   : "memory");   // (m0 is reserved in the AMDGPU backend: the compiler writes it right at each of its own uses and
                                  // never keeps a value there across other code, so it is not - and cannot be - listed as a clobber)
    rc0 += 16;
    if (rc0 >= rsegC && rsg + 1 < p.nseg) { rc0 = 0; ++rsg; raw_setup_seg(); }
  };
  // read side: float4 index (within a raw buffer) of pixel j of this thread's item, piece a = aq, for h = 0 (h = 1: ^ 8)
  unsigned xj[6];
  if constexpr (RAW) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int P = ahy * PW + 4 * tq + j, i = P >> 4, q = P & 15;
      xj[j] = (unsigned)(i * 64 + (((q & 3) * 16 + (q >> 2) * 2 + aq) | ((i & 1) << 3)));
    }
  }
  const bf4* const smem4r = reinterpret_cast<const bf4*>(smem);
  auto raw_read = [&](bf4 (&rv)[6], int buf, int h) {
    const unsigned b4 = (unsigned)(NST * A_STAGE + buf * R_STAGE) / 4u, hx = (unsigned)h << 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      asm volatile("" : "+v"(xj[j]));   // opaque: otherwise the four (buffer, h) address sets are hoisted out of the K loop (24 registers)
      rv[j] = smem4r[b4 + (xj[j] ^ hx)];
    }
  };

  // ---- weights: [Cout / 32][chunk][mu][nu][K half][32][4] floats; this wave reads slab (ct, kc, mu): 6 x 1 KB -----------------
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  const unsigned bvoff = (unsigned)((half * 32 + l31) * 16);
  const int ct = n0 / 32 + ng;
  bf4 fbg[2][6];
  float pfa[2] = {0.f, 0.f}, pfb[2] = {0.f, 0.f};   // touch-ahead destinations (never read; kept live until the load has landed)
  const unsigned pfb_off = lane < 48 ? (unsigned)lane * 128u : OOB;
  auto load_b = [&](int kc, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const unsigned so = (unsigned)(((ct * nkc + (kc < nkc ? kc : nkc - 1)) * 4 + mu) * 6) * 1024u;
#pragma unroll
    for (int j = 0; j < 6; ++j) fbg[SET][j] = conv_buf_load(brsrc, bvoff, so + (unsigned)j * 1024u);   // one address register for the six loads
  };

  f32x16 acc[6];
#pragma unroll
  for (int v = 0; v < 6; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;

  // ---- A fragments: lane (unit row ur, quad lq, K half) reads halo rows 2 ur + ra and 2 ur + rb and combines them ----------
  //   mu 0: v[0] - v[2]   mu 1: v[1] + v[2]   mu 2: v[2] - v[1]   mu 3: v[1] - v[3]
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  constexpr int A_STAGE4 = A_STAGE / 4;
  const int ur = l31 / QW, lq = l31 % QW;
  const int ra = mu == 0 ? 0 : mu == 2 ? 2 : 1, rb = mu == 0 ? 2 : mu == 1 ? 2 : mu == 2 ? 1 : 3;
  const float sgn = mu == 1 ? 1.f : -1.f;
  auto row_ad = [&](int hy) {   // float4 index of (halo row hy, nu 0, quad lq, this lane's K half)
    const int sw = QW >= 16 ? ((lq >> 3) & 1) : ((hy >> 1) & 1);
    return ((hy * 6) * QW + lq) * 2 + (half ^ sw);
  };
  const int ad_a = row_ad(2 * ur + ra), ad_b = row_ad(2 * ur + rb);

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  // Every thread issues the activation loads - threads without an item have a_ok = 0, i.e. out-of-range offsets that
  // return zero without touching memory - so that all waves have the SAME number of loads in flight: with the loads under
  // `if (stager)` the compiler has to place one s_waitcnt vmcnt(n) valid for both paths, and the staging waves then wait
  // for the weight loads they issued a moment ago (a full L2 latency per chunk)
  bf4 fa[2], fb2[2];   // A fragments (two halo rows) of the current / next nu step
  int st_cur = 0;      // activation stage of the current chunk
  const int nsc = nkc / 2;   // W2R_F_RAW: super-chunks
  if constexpr (RAW) {
    raw_setup_seg();
    raw_issue(0);
    if (nsc > 1) raw_issue(1);
    load_b(0, C0{});
    if constexpr ((FLAGS & W2R_F_B2) != 0) load_b(1, C1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stager) {
      raw_read(araw[0], 0, 0);
      store_item(0, C0{});
    }
    __syncthreads();
  } else {
    setup_seg();
    load_item(C0{});
    load_b(0, C0{});
    if constexpr ((FLAGS & W2R_F_B2) != 0) load_b(1, C1{});
    // (the chunk / segment state - c0, the buffer resource - advances on EVERY thread: kept uniform it lives in scalar
    // registers; advanced under `if (stager)` it becomes a per-lane value and every buffer load turns into a waterfall loop)
    next_chunk(1);
    load_item(C1{});
    if (stager) store_item(0, C0{});
    next_chunk(2);
    __syncthreads();
  }
  if constexpr (MID) {   // step 0 of chunk 0
    fa[0] = reinterpret_cast<const bf4*>(smem)[ad_a];
    fb2[0] = reinterpret_cast<const bf4*>(smem)[ad_b];
  }

  auto chunk = [&](int kc, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const int st_next = st_cur + 1 == NST ? 0 : st_cur + 1;
    const int sa = st_cur * A_STAGE4;
    if constexpr (RAW && PAR == 1 && (FLAGS & (W2R_DBG_NOA | W2R_DBG_NOALD)) == 0) {   // odd chunk: raw buffer (kc >> 1) & 1 was last read in chunk kc - 1; refill it with super-chunk
                                       // (kc >> 1) + 2.  Issued BEFORE the weight requests: the compiler's vmcnt counts for those stay
                                       // exact, and the wait for the last weight slab of chunk kc + 1 covers these (in-order return),
                                       // so the barrier that ends chunk kc + 1 publishes the buffer - first read in chunk kc + 2
      if ((kc >> 1) + 2 < nsc) raw_issue((kc >> 1) & 1);
    }
    constexpr bool B2 = (FLAGS & W2R_F_B2) != 0;
    if constexpr ((FLAGS & W2R_DBG_NOB) == 0 && !B2) load_b(kc + 1, std::integral_constant<int, 1 - PAR>{});
    const unsigned so2 = (unsigned)(((ct * nkc + (kc + 2 < nkc ? kc + 2 : nkc - 1)) * 4 + mu) * 6) * 1024u;   // B2: slab (kc + 2)
    if constexpr (!RAW && (FLAGS & (W2R_DBG_NOA | W2R_DBG_NOALD)) == 0) load_item(par_c);   // chunk kc + 2 into the register set chunk kc came from
    if constexpr ((FLAGS & W2R_F_PFA) != 0) {   // after the real loads: in-order completion then gives the touch two chunk times
      asm volatile("" ::"v"(pfa[PAR]));
      pfa[PAR] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, (int)pfo[PAR], (int)((unsigned)c0 * 4u + 128u), 0));
    }
    if constexpr ((FLAGS & W2R_F_PFB) != 0) {
      asm volatile("" ::"v"(pfb[PAR]));
      const unsigned so3 = (unsigned)(((ct * nkc + (kc + 3 < nkc ? kc + 3 : nkc - 1)) * 4 + mu) * 6) * 1024u;
      pfb[PAR] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, (int)pfb_off, (int)so3, 0));
    }
    __builtin_amdgcn_sched_barrier(0);   // the twelve requests of the chunk go out first, in this order (the s_waitcnt counts below rely on it)
    bf4 sv[6];   // W2R_F_ILV: the item of chunk kc + 1, transformed two channels per nu step
    constexpr bool ILV = (FLAGS & W2R_F_ILV) != 0 && (FLAGS & (W2R_DBG_NOA | W2R_DBG_NOAST)) == 0;
    auto xform = [&](int c) {
      constexpr int SET = 1 - PAR;
      const float d0 = araw[SET][0][c], d1 = araw[SET][1][c], d2 = araw[SET][2][c], d3 = araw[SET][3][c], d4 = araw[SET][4][c], d5 = araw[SET][5][c];
      const float t1 = __builtin_fmaf(-4.f, d2, d4), t2 = __builtin_fmaf(-4.f, d1, d3);
      const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
      sv[0][c] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
      sv[1][c] = t1 + t2;
      sv[2][c] = t1 - t2;
      sv[3][c] = t3 + t4;
      sv[4][c] = t3 - t4;
      sv[5][c] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    };
    float* const As_next = smem + st_next * A_STAGE + a_lds;
    constexpr bool COMB = (FLAGS & W2R_DBG_NOCOMB) == 0;
    if constexpr (!MID) {
      fa[0] = smem4[sa + ad_a];
      if constexpr (COMB) fb2[0] = smem4[sa + ad_b];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j + 1 < 6) {
        fa[(j + 1) & 1] = smem4[sa + ad_a + (j + 1) * (QW * 2)];
        if constexpr (COMB) fb2[(j + 1) & 1] = smem4[sa + ad_b + (j + 1) * (QW * 2)];
      } else if constexpr (MID) {   // step 0 of the next chunk: its stage was published by the barrier behind step 3
        fa[0] = smem4[st_next * A_STAGE4 + ad_a];
        if constexpr (COMB) fb2[0] = smem4[st_next * A_STAGE4 + ad_b];
      }
      bf4 a;
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = COMB ? __builtin_fmaf(sgn, fb2[j & 1][k], fa[j & 1][k]) : fa[j & 1][k];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], fbg[PAR][j][k], acc[j], 0, 0, 0);
      if constexpr (B2 && (FLAGS & W2R_DBG_NOB) == 0) fbg[PAR][j] = conv_buf_load(brsrc, bvoff, so2 + (unsigned)j * 1024u);
      if constexpr ((FLAGS & W2R_DBG_NOAST) != 0) {
        if (j == ((FLAGS & W2R_F_LATE) ? 2 : 0)) {
#pragma unroll
          for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(araw[1 - PAR][q][c]));
        }
      }
      if constexpr (RAW && ILV) {
        if (j == 0 && stager) raw_read(araw[1 - PAR], ((kc + 1) >> 1) & 1, 1 - PAR);   // the item of chunk kc + 1: h = (kc + 1) & 1
      }
      if constexpr (ILV) {
        if (stager) {
          constexpr int J0 = (FLAGS & W2R_F_LATE) ? 2 : 0;
          // transform steps / store steps: behind the raw read of step 0 (RAW) and in front of the barrier behind step 3 (MID)
          constexpr int JX0 = MID ? (RAW ? 1 : 0) : J0, JX1 = JX0 + 1, JS0 = MID ? 2 : J0 + 2, JS1 = JS0 + 1;
          if (j == JX0) { xform(0); xform(1); }
          if (j == JX1) { xform(2); xform(3); }
          if (j == JS0) {
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) *reinterpret_cast<bf4*>(As_next + nu * A_PLANE) = sv[nu];
          }
          if (j == JS1) {
#pragma unroll
            for (int nu = 3; nu < 6; ++nu) *reinterpret_cast<bf4*>(As_next + nu * A_PLANE) = sv[nu];
          }
        }
      }
      if constexpr (MID) {
        if (j == 3) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (RAW && PAR == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the raw requests of chunk kc - 1 (ten weight requests younger)
          __syncthreads();
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!ILV && (FLAGS & (W2R_DBG_NOA | W2R_DBG_NOAST)) == 0) { if (stager) store_item((kc + 1) & 1, std::integral_constant<int, 1 - PAR>{}); }   // chunk kc + 1
    if constexpr (RAW && B2 && PAR == 0 && !MID)   // the raw requests of chunk kc - 1 are older than the last 12 weight requests (in-order return):
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // landed before the barrier publishes them (also waits for the next chunk's slabs)
    if constexpr ((FLAGS & W2R_DBG_NOBAR) == 0 && !MID) __syncthreads();
    if constexpr (!RAW) next_chunk(kc + 3);
    st_cur = st_next;
  };
  // pairs without a condition between the two chunks (with `if (kc + 1 < nkc)` inside the loop the compiler has to size every
  // s_waitcnt for the path on which the odd chunk's requests were never issued - six fewer in flight, half the lookahead gone)
  int kc = 0;
  for (; kc + 1 < nkc; kc += 2) {
    chunk(kc, C0{});
    chunk(kc + 1, C1{});
  }
  if constexpr (!RAW) {
    if (kc < nkc) chunk(kc, C0{});
  }

  // ---- epilogue: x inverse in registers (conv_wino43's y0..y3 per mu), then the y inverse across the four mu waves of a
  // channel tile through LDS, one x position per round: waves mu = 1, 2 publish, mu = 0 forms row 2k = (m0 + m1) + m2,
  // mu = 3 forms row 2k + 1 = (m1 - m2) - m3.  C/D layout of the 32x32 MFMA: col = lane & 31 (cout), row = (r&3) + 8*(r>>2)
  // + 4*(lane>>5) = unit.
  if constexpr (MID) __syncthreads();   // no barrier behind the last chunk's steps 4 / 5 (fragment reads): the exchange buffer overlays the stages
  float* const xbuf = smem;                                  // [ng][which: mu 1 / mu 2][16 regs][64 lanes]
  const int n = n0 + ng * 32 + l31;
  const float bv = p.bias[n];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
      o[r] = j == 0 ? ((m0 + m1) + m2) + (m3 + m4) : j == 1 ? (m1 - m2) + 2.f * (m3 - m4) : j == 2 ? (m1 + m2) + 4.f * (m3 + m4)
                                                                                          : (m1 - m2) + (8.f * (m3 - m4) + m5);
    }
    if (j) __syncthreads();            // the previous round has been consumed
    if (mu == 1 || mu == 2) {
      float* give = xbuf + ((ng * 2 + (mu - 1)) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) give[r * 64] = o[r];
    }
    __syncthreads();
    if (mu == 0 || mu == 3) {
      const float* t1 = xbuf + ((ng * 2 + 0) * 16) * 64 + lane;
      const float* t2 = xbuf + ((ng * 2 + 1) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m1 = t1[r * 64], m2 = t2[r * 64];
        float v = mu == 0 ? (o[r] + m1) + m2 : (m1 - m2) - o[r];
        const int unit = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int y = y0 + 2 * (unit / QW) + (mu == 3 ? 1 : 0);
        const int x = x0 + 4 * (unit % QW) + j;
        v += bv;
        if (p.leaky) v = v > 0.f ? v : 0.2f * v;
        if (y < p.H && x < p.W) p.out[(((size_t)img * p.H + y) * p.W + x) * p.ostride + n] = v;
      }
    }
  }
}

template <int TH, int BN, int FLAGS, int QW = 8>
hipError_t conv_wino2d_r3_launch(const ConvParams& p, hipStream_t s) {
  constexpr int NWL = 4 * (BN / 32), NIL = ((TH + 2) * (4 * QW + 2) + 15) / 16;
  constexpr size_t r_bytes = (FLAGS & W2R_F_RAW) ? 2 * (size_t)(((NIL + NWL - 1) / NWL) * NWL) * 1024 : 0;
  constexpr size_t a_bytes = ((FLAGS & W2R_F_MIDBAR) ? 3 : 2) * (size_t)(TH + 2) * 6 * QW * 8 * sizeof(float) + r_bytes;
  constexpr size_t x_bytes = (size_t)(BN / 32) * 2 * 16 * 64 * sizeof(float);
  constexpr size_t lds = a_bytes > x_bytes ? a_bytes : x_bytes;
  constexpr int NT = 4 * (BN / 32) * 64;
  static_assert(lds <= (BN == 32 ? 80 : 160) * 1024, "LDS (two workgroups per CU with BN = 32)");
  if constexpr ((FLAGS & W2R_F_RAW) != 0) {
    if (p.Ctot % 16) return hipErrorInvalidValue;
    for (int i = 0; i < p.nseg; ++i)
      if (p.seg[i].C % 16 || p.seg[i].stride % 16 || p.seg[i].up) return hipErrorInvalidValue;
  }
  if (p.ksize != 3 || p.ksplit > 1 || p.Ctot % 8 || p.Cout % BN) return hipErrorInvalidValue;
  auto kern = conv_wino2d_r3_kernel<TH, BN, FLAGS, QW>;
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
  const int ntx = (p.W + 4 * QW - 1) / (4 * QW), nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, p);
  return hipGetLastError();
}
