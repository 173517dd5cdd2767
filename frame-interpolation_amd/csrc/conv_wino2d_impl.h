// conv_wino2d_impl.h -- 3x3 Conv2D('same') + bias + leaky_relu with a NESTED Winograd transform, fp32 MFMA: F(4,3) along x x F(2,3)
// along y.  Per output UNIT of 2 rows x 4 pixels and per (ci, co): 4 (mu) x 6 (nu) = 24 multiplies for 8 outputs = 3 per output
// (1-D F(4,3): 4.5, direct: 9).  Round 3's kernel (conv_wino2d_r3_kernel, tools/retired/conv_wino2d_r3_impl.h) with both activation transforms
// moved to the FRAGMENT side, and a new epilogue:
//
//   d (4 rows x 6 pixels)  --y: F(2,3) B^T-->  e[mu] = d[ra] +- d[rb]  --x: F(4,3) B^T-->  V[mu][nu]
//       mu 0: d[0] - d[2]     mu 1: d[1] + d[2]     mu 2: d[2] - d[1]     mu 3: d[1] - d[3]
//   weights  U[mu][nu] (film_layers.cpp packs them): F(4,3) of kernel row dy along x, then along dy
//       U[0] = u(0)    U[1] = ((u(0) + u(2)) + u(1)) / 2    U[2] = ((u(0) + u(2)) - u(1)) / 2    U[3] = u(2)
//   M[mu][nu] = sum_ci V[mu][nu] U[mu][nu]  (24 independent GEMMs), x inverse per mu, then
//       row 2k = (M'[0] + M'[1]) + M'[2]      row 2k+1 = (M'[1] - M'[2]) - M'[3]
//
// What the measurements of rounds 3 / 4 said, and what the kernel does about it (profiles/r04_w2d_*.log):
//   * conv_wino2d_r3_kernel staged a K chunk in two hops (raw patch -> 160 threads transform along x -> LDS image -> every wave combines
//     two rows of it along y): the waves that own the staging threads are late at the chunk's barrier.  Here nobody stages: the RAW
//     halo patch of a 16-channel super-chunk goes to LDS by DMA (`buffer_load_dwordx4 ... lds`: no registers, no VALU), three stages,
//     its requests spread over the MFMA gaps of two chunks (a request costs 60-180 cycles of issue), four chunks ahead of its first
//     reader, published by ONE barrier per super-chunk (48 MFMAs per wave).  A wave (one mu, 32 output channels, six nu planes: 96
//     accumulator registers) reads the six raw pixels of its two halo rows (12 ds_read_b128 per chunk - as many as the transformed
//     rows cost before) and transforms them in registers, every wave the same work, in the MFMA gaps of the previous chunk.
//   * The loop is ISSUE bound (two waves per SIMD: every vector instruction beside the MFMAs shows; PMC: MFMA pipe 63 % busy with
//     the x-then-y order of conv_wino2d_r3_kernel = 120 VALU instructions per 24 MFMAs, 70 % with none).  So the y combine comes FIRST
//     - on the raw rows, 24 instructions - and then ONE x transform (48): 72 per chunk.  Same linear map, another operation order:
//     this kernel is its own summation family (W2D_F_XFIRST keeps the old order - and with it conv_wino2d_r3_kernel's bits - for
//     tools/w2d_bench.hip, which checks the loop against that kernel; the planner never uses it).
//   * MFMA order inside a chunk: nu pairs, the two accumulators of a pair alternating (an MFMA never follows the one it depends on:
//     fillers between them cost issue slots, not forwarding stalls); the weight slab of a plane goes straight from L2 into registers
//     ([Cout / 32][chunk][mu][nu][K half][32][4] = one coalesced 1 KB read per (mu, nu) step) and is re-requested for chunk kc + 2
//     into the same registers right behind its last MFMA.
//   * conv_wino2d_r3_kernel's epilogue cost 26 000 cycles per workgroup (s_memtime; the K loop of a K = 208 layer: 101 000): 128 dword
//     stores per 32-channel tile from two of the four mu waves, in four exchange rounds with two barriers each.  The MFMA operands
//     are swapped here (A = weights, B = activations: C^T, the same sums) so that a lane holds four CONSECUTIVE channels of a unit;
//     the x-inverted planes go through LDS once per x position (double buffered: one barrier per round), every thread combines the
//     four mu planes of one (unit, 4-channel group) into both output rows and stores them as dwordx4: 32 stores of 1 KB per tile
//     (8 pixels x 128 contiguous bytes each), all waves storing.
//
// LDS layout of a stage (bytes): halo row r at r * 2368; in a row pixel px = 4 k + m, 16-byte piece c of its 64 bytes at
// m * 576 + c * 144 + k * 16 (k = 0..8: nine slots per (m, piece), the ninth unused for m = 2, 3).  A fragment read is
// (halo row 2 ur + ro, pixel 4 lq + j, piece 2 h + half) = lane base + an immediate, and conflict free: the eight quads of a
// unit row are 128 contiguous bytes and two rows down is 4736 = 128 (mod 256) bytes away - each ds_read_b128 lane group
// (MI355X: {0-3, 12-15, 20-27}, ...) covers all 64 banks once.  The DMA writes 64 consecutive 16-byte slots per request; which
// (pixel, piece) a lane fetches is free, so the layout costs nothing on the write side (out-of-image and padding slots carry an
// out-of-range offset and receive zeros = the 'same' padding).
//
// Needs every input segment's C % 16 == 0 (16-byte aligned pixels), Ctot % 16 == 0, a 16-byte aligned output slice.  fp32 throughout.
#pragma once
#include "conv_buf_impl.h"

enum { W2D_F_SQ = 128,         // the 32 units of a tile as 8 unit rows x 4 units = 16 x 16 pixels instead of 4 x 8 = 8 rows x 32 pixels: levels whose width is
                               // a multiple of 16 but not of 32 (144x240: 7.5 tiles per row, 6.25 % of the MFMA columns masked) tile exactly.  18 halo rows of 18
                               // pixels (row pitch 82 slots) fill the same 24 requests per stage.  Same sums, same bits.
       W2D_F_EPI1 = 16,        // tools only (round 6, rejected): epilogue on FOUR exchange buffers - every wave writes its four x positions, ONE barrier, then the
                               // four rounds of reads / stores (instead of a barrier per round on two buffers); 64 KB (32 channels) / 128 KB (64) of LDS, not with
                               // the fused 1x1.  Stand-alone -2..-12 % for the 8 x 32 tiles, +5..7 % for the 32-channel 16 x 16 ones; in the engine nothing
                               // (same-box A/B: conv_wino2d_kernel 29.18 vs 29.16-29.6 ms per forward, the K > 528 layers slower): profiles/r06_w2d_epilogue_one_barrier.log
       W2D_F_XEPI = 8,         // the instantiation that can do split-K (raw partial sums), the fused AveragePooling2D and the fused 1x1 convolution (round 6): the launcher
                               // refuses those ConvParams without it, and the plain instantiation carries none of their code - per exchange round of the epilogue
                               // ~35 instructions of tests, branches and register copies that 45 of the 56 layers of a plan never need, each of which waits for an
                               // issue slot behind the co-resident workgroup's K loop (profiles/r06_w2d_plain_epilogue.log)
       W2D_F_CHAIN = 64,       // a workgroup walks ConvParams::chain consecutive pixel tiles (same output channels): the DMA cursor and the weight
                               // requests run on into the next tile while this one finishes, the epilogue's exchange buffers lie BEHIND the stages
                               // (see "chained tiles" below).  Same sums, same bits.
       W2D_F_XFIRST = 32,      // tools only: x transform per row, then the y combine (120 VALU per chunk; the bits of conv_wino2d_r3_kernel)
       W2D_DBG_NOXF = 256,     // timing ablations (tools only; results are wrong on purpose): no transforms (raw pixels as fragments)
       W2D_DBG_NODMA = 512,    // no DMA requests in the K loop
       W2D_DBG_NOB = 1024,     // no weight requests in the K loop
       W2D_DBG_NOBAR = 2048,   // no barrier in the K loop
       W2D_DBG_NORD = 4096,    // no fragment reads in the K loop
       W2D_DBG_TIME = 8192 };  // wave 0 of every workgroup writes s_memtime at kernel entry / first MFMA / last MFMA / exit (and three stamps inside the epilogue) to
                               // p.part[workgroup * 16 ..], behind the first DMA request / its own stage-0 share / the first barrier, and the
                               // cycles it spent in the K loop's s_waitcnt + barrier pairs (slot 7)
                               // (tools/w2d_bench.hip prints the averages)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the MFMA gaps of a chunk, every index a compile-time constant
template <class F, int... G>
__device__ __forceinline__ void w2d_for_each(F&& f, std::integer_sequence<int, G...>) { (f(std::integral_constant<int, G>{}), ...); }

// BN = 32 NG output channels per workgroup, one wave per (mu, 32 channels): 4 NG waves, two waves per SIMD either way (BN = 64: one
// workgroup of 8 waves per CU; BN = 32: two of 4 - their prologues / epilogues overlap the other's K loop: the usual winner)
// the multiply-shift quotients of the DMA slot decomposition (slot / 148, rest / 36, rest / 9) are exact on the ranges they are used on
constexpr unsigned w2d_magic20(unsigned d) { return ((1u << 20) + d - 1u) / d; }
constexpr bool w2d_quotient_exact(unsigned d, unsigned n) {   // (x * magic) >> 20 == x / d for x < n; 24-bit operands, a product below 2^32
  for (unsigned x = 0; x < n; ++x) if (((x * w2d_magic20(d)) >> 20) != x / d) return false;
  return n < (1u << 12) && w2d_magic20(d) < (1u << 20);
}

// DMA slot -> (halo row r, halo pixel px, 16-byte piece c) of the stage layout above, as a compile-time table: r << 24 | px << 8 | c << 4, ~0 for
// the padding slots of a row / behind the last halo row.  The kernel reads a lane's NREQ / NW entries with one vector load each at kernel
// entry (6 KB per arrangement, cache resident) instead of decomposing the slot number with three multiply-shift quotients per request -
// 115 of the ~150 vector instructions in front of the first DMA request (each one waits for an issue slot the co-resident workgroup's K
// loop leaves free: round 6, "table prologue" below).
template <bool SQ> struct W2dSlotTable { unsigned v[24 * 64]; };
template <bool SQ> constexpr W2dSlotTable<SQ> w2d_make_slot_table() {
  constexpr int QW = SQ ? 4 : 8, TH = SQ ? 16 : 8, HR = TH + 2, PW = 4 * QW + 2;
  constexpr int CO4 = (PW + 3) / 4, MO4 = 4 * CO4, RP4 = 4 * MO4 + (SQ ? 2 : 4);
  W2dSlotTable<SQ> t{};
  for (int sl = 0; sl < 24 * 64; ++sl) {
    const int r = sl / RP4, rem = sl % RP4, m = rem / MO4, rr = rem % MO4, c = rr / CO4, k = rr % CO4, px = 4 * k + m;
    const bool ok = r < HR && rem < 4 * MO4 && px < PW;
    t.v[sl] = ok ? ((unsigned)r << 24 | (unsigned)px << 8 | (unsigned)c << 4) : 0xFFFFFFFFu;
  }
  return t;
}
template <bool SQ> __device__ const W2dSlotTable<SQ> w2d_slot_table = w2d_make_slot_table<SQ>();

template <int BN, int FLAGS, int NS_ = 3>
__global__ __launch_bounds__(4 * (BN / 32) * 64, (BN == 32) ? 2 : 1) void conv_wino2d_kernel(ConvParams p) {
  constexpr bool SQ = (FLAGS & W2D_F_SQ) != 0;
  constexpr int QW = SQ ? 4 : 8, TH = SQ ? 16 : 8, HR = TH + 2, PXW = 4 * QW, PW = PXW + 2;   // units per tile row; tile rows; halo rows; tile / halo pixels per row
  static_assert(QW * (TH / 2) == 32, "32 units per tile");
  constexpr int NG = BN / 32, NW = 4 * NG;
  constexpr int CO4 = (PW + 3) / 4, MO4 = 4 * CO4, RP4 = 4 * MO4 + (SQ ? 2 : 4);   // piece stride, m stride, row pitch in 16-byte slots (8 x 32: 144, 576, 2368 bytes;
                                                                                    // 16 x 16: 80, 320, 1312 bytes - two halo rows down is 2624 = 64 (mod 256) bytes away)
  constexpr int NREQ = 24;                      // DMA requests (1 KB each) per stage: 10 rows x 148 slots = 1480, 18 rows x 82 slots = 1476 <= 1536
  constexpr int STAGE4 = NREQ * 64;             // slots per stage
  constexpr int NS = NS_;                       // stages (3; the chained 32-channel tile: 2, so that two workgroups with their exchange buffers fit a CU)
  constexpr bool CHAIN = (FLAGS & W2D_F_CHAIN) != 0;
  static_assert(NS == 2 || NS == 3, "stages");
  constexpr int IPW = NREQ / NW;                // requests per wave and super-chunk
  static_assert(NREQ % NW == 0, "requests per wave");
  static_assert(w2d_quotient_exact(RP4, NREQ * 64) && w2d_quotient_exact(MO4, RP4) && w2d_quotient_exact(CO4, MO4), "slot quotients");
  static_assert(HR * RP4 <= STAGE4, "stage size");
  constexpr bool XF = (FLAGS & W2D_F_XFIRST) != 0;
  constexpr bool XE = (FLAGS & W2D_F_XEPI) != 0;
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [stage 0][stage 1][stage 2]; the epilogue reuses it as the exchange buffer

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mu = wv & 3, ng = wv >> 2;

  unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tmA = 0, tmB = 0, tmC = 0, tmW = 0, tmE1 = 0, tmE2 = 0, tmE3 = 0;
  unsigned long long rt0 = 0;
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) { rt0 = __builtin_amdgcn_s_memrealtime(); tm0 = __builtin_readcyclecounter(); }

  // Prologue diet (round 6).  With two workgroups per CU the ~500 instructions between kernel entry and the first DMA request issue at
  // ~19 cycles each - the co-resident workgroup's K loop owns the issue slots - i.e. 9 000-11 000 cycles of a 45 000-cycle workgroup
  // (profiles/r06_w2d_valu_diet.log).  So: quotients by host-made reciprocals (ConvParams::mg_*; exact for x d < 2^32, which the
  // launcher checks), unsigned slot arithmetic without branches, and no accumulator clearing (the first chunk's MFMAs take C = 0).
  // Table prologue (round 6, second pass).  (a) Every kernel argument the code in front of the first DMA request reads is loaded HERE, in one
  // batch behind one s_waitcnt: hipcc placed each s_load at its first use, behind the branches of the fallback divisions - seven dependent
  // round trips to the scalar cache (~250 cycles each with nothing else to issue) before the first request.  The reciprocals are now
  // mandatory (0 only for a divisor of 1; the launcher refuses what it cannot make exact), so no division code and no branches are left.
  // (b) A lane's slot entries come from w2d_slot_table (above), requested first of all.
  int aH = p.H, aW = p.W, aNB = p.NB, aCtot = p.Ctot, aKsplit = p.ksplit, aNseg = p.nseg, aChain = p.chain;
  unsigned a_tpi = p.mg_tpi, a_ntx = p.mg_ntx, a_nby = p.mg_nby;
  int ntx_ = p.tl_ntx, tpi_ = p.tl_tpi;
  int aGx = (int)gridDim.x, aGy = (int)gridDim.y;
  const float* aS0ptr = p.seg[0].ptr;
  int aS0stride = p.seg[0].stride, aS0C = p.seg[0].C, aS0boff = p.seg[0].boff, aS0bmod = p.seg[0].bmod;
  const float* aWptr = p.w;
  unsigned rtab[(24 / (4 * (BN / 32)))];
  if constexpr (!CHAIN) {
#pragma unroll
    for (int n = 0; n < 24 / NW; ++n) rtab[n] = w2d_slot_table<SQ>.v[t + 64 * NW * n];
  }
  {
    unsigned long long q0 = (unsigned long long)(uintptr_t)aS0ptr, q1 = (unsigned long long)(uintptr_t)aWptr;
    asm volatile("" : "+s"(aH), "+s"(aW), "+s"(aNB), "+s"(aCtot), "+s"(aKsplit), "+s"(aNseg), "+s"(aChain), "+s"(a_tpi), "+s"(a_ntx), "+s"(a_nby), "+s"(ntx_), "+s"(tpi_),
                      "+s"(aGx), "+s"(aGy), "+s"(q0), "+s"(q1), "+s"(aS0stride), "+s"(aS0C), "+s"(aS0boff), "+s"(aS0bmod));
    aS0ptr = reinterpret_cast<const float*>((uintptr_t)q0);
    aWptr = reinterpret_cast<const float*>((uintptr_t)q1);
  }
  unsigned long long tmK = 0, tmT = 0;   // W2D_DBG_TIME: the kernel arguments are here / the slot-table entries are here
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmK = __builtin_readcyclecounter();
  // x / d by the launcher's reciprocal; magic = 0 says d = 1 (a scalar select spelled out: hipcc made a branch around the s_mul_hi_u32)
  auto udiv = [](unsigned x, unsigned magic) -> unsigned {
    unsigned q;
    const unsigned h = __umulhi(x, magic);
    asm("s_cmp_eq_u32 %2, 0\n\ts_cselect_b32 %0, %1, %3" : "=s"(q) : "s"(x), "s"(magic), "s"(h) : "scc");
    return q;
  };
  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr ((FLAGS & CONV_B_XCD_M) != 0) {
    const int nbx = aGx, nby = aGy;
    const int nwg = nbx * nby;
    const int lin = by * nbx + bx;
    const int xcd = lin & 7, idx = lin >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int nl = base + idx;
    bx = (int)udiv((unsigned)nl, a_nby);
    by = nl - bx * nby;
  }
  // Chained tiles (W2D_F_CHAIN): this workgroup owns the pixel tiles [g0, g0 + nt) of the launch, nt <= p.chain, one after the other.
  // What a tile costs beside its K loop is mostly WAITING - 7 000 (one workgroup per CU) to 17 000 (two) cycles between kernel entry and
  // the first MFMA, 4 ms of a 31 ms forward (profiles/r06_w2d_idle_budget.md) - so the next tile's first super-chunks and weight slabs
  // are requested by the ordinary in-loop DMA / weight stream of the current tile (the cursor simply runs on: no burst in front of
  // the epilogue, which is what sank round 4's attempt), land during its last chunks and its epilogue, and the next K loop starts
  // right behind the epilogue.  The exchange buffers then cannot overlay the stages: they lie behind them.
  const int ntx = ntx_, tpi = tpi_;                             // tiles per row, per image (the launcher's: (W + PXW - 1) / PXW, x (H + TH - 1) / TH)
  const int chain = CHAIN ? (aChain > 1 ? aChain : 1) : 1;
  const int g0 = bx * chain;
  const int nt = CHAIN ? min(chain, aNB * tpi - g0) : 1;       // tiles of this workgroup
  int img = (int)udiv((unsigned)g0, a_tpi);
  const int trow0 = (int)udiv((unsigned)(g0 - img * tpi), a_ntx);
  int y0 = trow0 * TH, x0 = (g0 - img * tpi - trow0 * ntx) * PXW;   // the tile the MFMAs / the epilogue are at
  int c_t = 0, c_img = img, c_y0 = y0, c_x0 = x0;               // the tile the DMA cursor is at
  const int n0 = by * BN;

  auto uniform_ptr = [](const float* q) -> const float* {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v)));
  };
  // ---- DMA side: request n of this wave fills slots 64 * (wv + NW n) ... + 63 of a stage; lane -> (halo row, pixel, piece) -----
  unsigned rvoff[IPW];
  int rsg = 0, rc0 = 0, rsegC = 0;
  conv_rsrc_t rrsrc = conv_make_rsrc(uniform_ptr(aS0ptr));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  // Which (halo row, pixel, 16-byte piece) a lane's slot is, and whether that pixel lies in the image, does not depend on the input
  // segment: (pixel offset from the halo patch's first pixel) << 2 | piece, or ~0 for padding / outside ('same' padding = zeros), once per
  // workgroup.  The slot's (row, pixel, piece) is a table entry (w2d_slot_table); a tile whose halo lies inside the image (a scalar test)
  // needs no bounds tests at all; a segment set-up then forms the byte offsets with two 24-bit multiply-adds per request.
  // Chained tiles: the tile-INDEPENDENT part of it (halo row | pixel << 8 | piece << 16 | slot in use << 31) is kept instead, and a tile /
  // segment set-up forms the offsets from it and the cursor's tile - behind an opaque copy, or hipcc hoists the unpacked fields out
  // of the tile loop (18 registers, spilled, reloaded with s_waitcnt vmcnt(0) in the middle of the K loop).
  unsigned rpk[IPW];
  auto quot20 = [](unsigned x, unsigned magic) -> unsigned { unsigned q; asm("v_mul_u32_u24 %0, %1, %2\n\tv_lshrrev_b32 %0, 20, %0" : "=v"(q) : "v"(x), "s"(magic)); return q; };
  auto mad24 = [](unsigned a, int b, unsigned c) -> unsigned { unsigned q; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(q) : "v"(a), "s"(b), "v"(c)); return q; };
  auto madu24 = [](unsigned a, unsigned b, unsigned c) -> unsigned { unsigned q; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(q) : "v"(a), "s"(b), "v"(c)); return q; };
  auto set_rpk = [&]() {   // for the cursor's tile (chained: tile independent, called once)
    const int H = aH, W = aW;
    if constexpr (CHAIN) {
#pragma unroll
      for (int n = 0; n < IPW; ++n) {
        // slot -> (halo row r, pixel 4 k + m, piece c): multiply-shift quotients, exact on [0, NREQ * 64) (static_asserts above)
        const unsigned sl = 64u * (unsigned)(wv + NW * n) + (unsigned)lane;
        const unsigned r = quot20(sl, w2d_magic20(RP4)), rem = mad24(r, -RP4, sl);
        const unsigned m = quot20(rem, w2d_magic20(MO4)), rr = mad24(m, -MO4, rem);
        const unsigned c = quot20(rr, w2d_magic20(CO4)), k = mad24(c, -CO4, rr);
        const unsigned px = 4u * k + m;
        const unsigned slot_ok = (unsigned)(r < (unsigned)HR) & (unsigned)(rem < 4u * MO4) & (unsigned)(px < (unsigned)PW);
        rpk[n] = r | px << 8 | c << 16 | (slot_ok ? 0x80000000u : 0u);
      }
    } else {
      // rpk = the table entry, or ~0 where the slot's pixel lies outside the image: nothing to do for a tile whose halo lies inside (a scalar
      // test; the buffer resource of a segment starts at the halo patch's first pixel (c_y0 - 1, c_x0 - 1): offsets count from there)
      const bool inside = c_y0 >= 1 && c_y0 + TH + 1 <= H && c_x0 >= 1 && c_x0 + PXW + 1 <= W;
      if (inside) {   // (one branch for all requests: hipcc tested `inside` once per request)
#pragma unroll
        for (int n = 0; n < IPW; ++n) rpk[n] = rtab[n];
      } else {
#pragma unroll
        for (int n = 0; n < IPW; ++n) {
          const unsigned tv = rtab[n];
          const unsigned y = (unsigned)(c_y0 - 1) + (tv >> 24), x = (unsigned)(c_x0 - 1) + ((tv >> 8) & 255u);   // (wraps below 0: fails the unsigned bound;
          rpk[n] = ((y < (unsigned)H) & (x < (unsigned)W)) ? tv : OOB;                                           //  a padding slot: row 255)
        }
      }
    }
  };
  set_rpk();
  // Every slot entry is waited for HERE, in front of the first DMA request.  hipcc's s_waitcnt pass does not count the DMA statements (inline asm): it
  // placed "vmcnt(3) ... vmcnt(0)" for table entries 2..5 BETWEEN them, and with the requests outstanding that the pass does not know of, vmcnt(1) /
  // vmcnt(0) waited for the first DMA requests to COMPLETE before the fourth and fifth went out - an HBM round trip inside the issue sequence (found
  // with two more W2D_DBG_TIME stamps: entry -> all six requests issued 6 400 -> 3 000 cycles, profiles/r06_w2d_table_wait.log).
#pragma unroll
  for (int n = 0; n < IPW; ++n) asm volatile("" : "+v"(rpk[n]));
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmT = __builtin_readcyclecounter();
  auto raw_setup_seg_of = [&](const float* sptr, int sstride, int sC, int sboff, int sbmod) {
    rsegC = __builtin_amdgcn_readfirstlane(sC);
    int be = c_img + sboff;
    if (sbmod && be >= sbmod) be -= sbmod;
    // (the finished pointer through readfirstlane: should hipcc ever reload `p` with vector loads - it does behind an atomic - a buffer
    // resource in VGPRs cannot feed the DMA statement - nor a buffer load without a waterfall loop)
    const int spix = (be * aH + (c_y0 - 1)) * aW + (CHAIN ? 0 : c_x0 - 1);   // (32 bits, may be negative at the first tile; x the pixel pitch in 64)
    rrsrc = conv_make_rsrc(uniform_ptr(sptr + (long long)spix * sstride));
    const unsigned st4 = (unsigned)sstride * 4u;
    if constexpr (CHAIN) {
#pragma unroll
      for (int n = 0; n < IPW; ++n) {
        unsigned ri = rpk[n];
        asm volatile("" : "+v"(ri));
        const int r = (int)(ri & 255u), px = (int)((ri >> 8) & 255u);
        const int y = c_y0 - 1 + r, x = c_x0 - 1 + px;
        const bool ok = (int)ri < 0 && (unsigned)y < (unsigned)aH && (unsigned)x < (unsigned)aW;
        rvoff[n] = ok ? (unsigned)(r * aW + x) * st4 + ((ri >> 16) & 3u) * 16u : OOB;
      }
    } else {
      const unsigned wst4 = (unsigned)aW * st4;   // (a halo row's bytes < 2^24: the launcher)
#pragma unroll
      for (int n = 0; n < IPW; ++n) {
        const unsigned tv = rpk[n];
        unsigned off, tmp;   // (r W + px) pitch + 16 c; one statement: one temporary (this runs inside the K loop too, at the register limit)
        asm("v_and_b32 %0, 48, %2\n\tv_bfe_u32 %1, %2, 8, 8\n\tv_mad_u32_u24 %0, %1, %3, %0\n\tv_lshrrev_b32 %1, 24, %2\n\tv_mad_u32_u24 %0, %1, %4, %0"
            : "=&v"(off), "=&v"(tmp) : "v"(tv), "s"(st4), "s"(wst4));
        rvoff[n] = tv == OOB ? OOB : off;
      }
    }
  };
  auto raw_setup_seg = [&]() {
    const ConvSeg& s = p.seg[rsg];
    raw_setup_seg_of(s.ptr, s.stride, s.C, s.boff, s.bmod);
  };
  auto dma_piece = [&](int n, int stage) {   // request n of this wave's share of the cursor's super-chunk -> stage
    const unsigned so = (unsigned)rc0 * 4u;
    const unsigned base = lds0 + (unsigned)stage * (STAGE4 * 16u) + (unsigned)(wv + NW * n) * 1024u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(base), "v"(rvoff[n]), "s"(rrsrc), "s"(so)
                 : "memory");   // (m0 is written in the statement that reads it: the compiler never keeps a value there across other code)
  };
  auto dma_advance = [&]() {   // the cursor moves to the next super-chunk (16 channels), into the next input segment behind the last one
    rc0 += 16;
    if (rc0 >= rsegC) {
      if (rsg + 1 < aNseg) { rc0 = 0; ++rsg; raw_setup_seg(); }
      else if constexpr (CHAIN) {   // ... and into the next tile of the chain behind the last segment (a uniform branch, once per tile)
        if (c_t + 1 < nt) {
          ++c_t;
          const int g = g0 + c_t;
          c_img = g / tpi;
          const int tr = g - c_img * tpi;
          c_y0 = (tr / ntx) * TH; c_x0 = (tr % ntx) * PXW;
          rc0 = 0; rsg = 0;
          raw_setup_seg();
        }
      }
    }
  };
  // split-K (ConvParams::ksplit, film_kernels.h; round 5): blockIdx.z = split z sums the super-chunks [sc0, sc1) of the K loop and
  // writes RAW partial sums to part[z][pixel][Cout]; conv_splitk_reduce_kernel adds them in split order with the bias and the
  // activation.  The planner uses it for the K >= 768 layers of levels with <= 4096 pixels, whose workgroups do not fill the chip (36x60:
  // 640 workgroups on 512 slots); on the 72x120 level (2304 workgroups = 4.5 rounds) two K ranges gained 2-4 % stand-alone and nothing in
  // the forward (profiles/r05_w2d_splitk.log): not used there.
  const int ksp = XE && aKsplit > 1 ? aKsplit : 1;
  const int nsc_all = aCtot >> 4;
  int sc0 = 0, sc1 = nsc_all;
  if (ksp > 1) {   // (a uniform branch, 32-bit quotients: 460 instead of 740 instructions between kernel entry and the first DMA request of
                   // an unsplit launch - round 4's kernel had 380.  Same-box A/B on the K <= 64 ... 528 layer shapes: no measurable
                   // difference, the scalar work hides behind the other waves' - profiles/r05_w2d_prologue_fix.log)
    sc0 = (int)((unsigned)nsc_all * blockIdx.z / (unsigned)ksp);            // nsc_all * ksplit < 2^16 * 2^4
    sc1 = (int)((unsigned)nsc_all * (blockIdx.z + 1u) / (unsigned)ksp);
    int c = sc0 * 16;   // the DMA cursor starts at this split's first super-chunk: walk the concat segments
    while (rsg + 1 < aNseg && c >= p.seg[rsg].C) { c -= p.seg[rsg].C; ++rsg; }
    rc0 = c;
  }
  auto raw_issue = [&](int stage) {
#pragma unroll
    for (int n = 0; n < IPW; ++n) dma_piece(n, stage);
    dma_advance();
  };

  // ---- weights: [Cout / 32][chunk][mu][nu][K half][32][4] floats; this wave reads slab (ct, kc, mu): 6 x 1 KB -----------------
  const int nkc = aCtot / 8, nsc = sc1 - sc0, kc0 = 2 * sc0, kc1 = 2 * sc1;
  const conv_rsrc_t brsrc = conv_make_rsrc(uniform_ptr(aWptr));
  const unsigned bvoff = (unsigned)((half * 32 + l31) * 16);
  const int ct = n0 / 32 + ng;
  // (past the end of the K range: a chained workgroup wraps to the first chunks - the next tile's; otherwise the last chunk again, unused)
  auto slab = [&](int kc) { return (unsigned)(((ct * nkc + (CHAIN ? (kc < nkc ? kc : kc - nkc) : (kc < nkc ? kc : nkc - 1))) * 4 + mu) * 6) * 1024u; };
  bf4 fbg[2][6];

  f32x16 acc[6];   // never cleared: the first MFMA of every plane in a tile's first chunk takes C = 0 (96 v_mov less in front of the K loop)

  // ---- fragments: lane (unit row ur, quad lq, K half) reads pixels 4 lq .. 4 lq + 5 of halo rows 2 ur + ra and 2 ur + rb --------
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  const int ur = l31 / QW, lq = l31 % QW;
  const int ra = mu == 0 ? 0 : mu == 2 ? 2 : 1, rb = mu == 0 ? 2 : mu == 1 ? 2 : mu == 2 ? 1 : 3;
  const float sgn = mu == 1 ? 1.f : -1.f;
  const int ix_a = (2 * ur + ra) * RP4 + half * CO4 + lq, ix_b = (2 * ur + rb) * RP4 + half * CO4 + lq;
  // PACKED fp32 math for both transforms (tools/experiments/mfma_valu_overlap.hip: a VALU instruction between two fp32 MFMAs costs
  // ~4 cycles of matrix-pipe time - the f32 MFMA and the VALU share the SIMD - and v_pk_fma_f32 costs what v_fma_f32 costs for
  // twice the work): a lane's four channels are two float2, 36 instructions per chunk instead of 72.  Same operations per element.
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 A[2][6][2];      // A[kc & 1]: the fragments of chunk kc (all six nu planes, this lane's four channels as two pairs)
  bf4 d[6];           // raw pixels of row a (then: of the combined row)
  bf4 d2[6];          // raw pixels of row b

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  auto read_row = [&](bf4 (&dst)[6], int ix, int stage, auto h_c) {
    constexpr int IMM = decltype(h_c)::value * 2 * CO4;
    const int ixs = ix + stage * STAGE4;
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j] = smem4[ixs + IMM + (j & 3) * MO4 + (j >> 2)];
  };
  auto pair_of = [](const bf4& v, int q) -> f2 { return q ? f2{v[2], v[3]} : f2{v[0], v[1]}; };
  auto set_pair = [](bf4& v, int q, f2 x) { if (q) { v[2] = x[0]; v[3] = x[1]; } else { v[0] = x[0]; v[1] = x[1]; } };
  // (hipcc scalarises a <2 x float> fma / add written in C here - 72 v_fma_f32 again - so the packed instructions are spelled out;
  // plain `asm`, not volatile: they are pure functions of their operands and the compiler schedules them like any other)
  auto fma2 = [](float k, f2 x, f2 y) -> f2 {
    f2 r;
    const f2 kk = {k, k};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(kk), "v"(x), "v"(y));
    return r;
  };
  auto add2 = [](f2 x, f2 y) -> f2 { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto mul2 = [](float k, f2 x) -> f2 {
    f2 r;
    const f2 kk = {k, k};
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(kk), "v"(x));
    return r;
  };
  auto max1 = [](float x, float y) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto sub2 = [](f2 x, f2 y) -> f2 { f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
  // F(4,3) B^T of channel pair q of the six pixels in d: 12 operations (v3 / v4 = t3 +- 2 (d3 - d1): the doubling is exact, so
  // fma(+-2, d3 - d1, t3) rounds the very sum conv_wino2d_r3_kernel formed with a multiplication and an addition)
  f2 xt[4];
  auto xf_t = [&](int q) {
    const f2 d1 = pair_of(d[1], q), d2v = pair_of(d[2], q), d3 = pair_of(d[3], q), d4 = pair_of(d[4], q);
    xt[0] = fma2(-4.f, d2v, d4);
    xt[1] = fma2(-4.f, d1, d3);
    xt[2] = sub2(d4, d2v);
    xt[3] = sub2(d3, d1);
  };
  auto xf_v = [&](int q, int nu) -> f2 {
    if constexpr ((FLAGS & W2D_DBG_NOXF) != 0) return pair_of(d[nu], q);
    switch (nu) {
      case 0: return fma2(4.f, pair_of(d[0], q), fma2(-5.f, pair_of(d[2], q), pair_of(d[4], q)));
      case 1: return add2(xt[0], xt[1]);
      case 2: return sub2(xt[0], xt[1]);
      case 3: return fma2(2.f, xt[3], xt[2]);
      case 4: return fma2(-2.f, xt[3], xt[2]);
      default: return fma2(4.f, pair_of(d[1], q), fma2(-5.f, pair_of(d[3], q), pair_of(d[5], q)));
    }
  };
  // hipcc sinks a value whose only readers sit behind the next branch (the next chunk) out of the MFMA gap it was written in -
  // down to one block of VALU instructions in front of the chunk that needs them: an empty statement "using" it pins it
  auto pin = [](f2& v) { asm volatile("" : "+v"(v)); };
  // the whole preparation of a chunk's fragments in one go (prologue only)
  auto prepare = [&](f2 (&An)[6][2], int stage, auto h_c) {
    read_row(d, ix_a, stage, h_c);
    read_row(d2, ix_b, stage, h_c);
    if constexpr (!XF) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) set_pair(d[j], q, fma2(sgn, pair_of(d2[j], q), pair_of(d[j], q)));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xf_t(q);
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) An[nu][q] = xf_v(q, nu);
    }
    if constexpr (XF) {
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = d2[j];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        xf_t(q);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) An[nu][q] = fma2(sgn, xf_v(q, nu), An[nu][q]);
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------------------
  // Order of the prologue's requests.  Stage 0 and the first weight slab go out FIRST and alone; everything else - the second slab,
  // super-chunks 1 and 2: 18 of a wave's 27 requests, ~100 cycles of the CU's address unit each - only behind the barrier that says
  // stage 0 has landed.  Issued up front (rounds 3-4a) they queued in front of the slower waves' stage-0 requests: the barrier came
  // 5 800 cycles after wave 0's own share had landed, now 2 900 (profiles/r04_w2d_prologue_order.log: -2 .. -10 % per layer, most on
  // the short-K layers; super-chunk 1 now lands during chunk 0's MFMAs instead of during the wait).
  if (ksp > 1) raw_setup_seg();   // (a split's cursor may start in any segment)
  else raw_setup_seg_of(aS0ptr, aS0stride, aS0C, aS0boff, aS0bmod);
  raw_issue(0);
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmA = __builtin_readcyclecounter();
#pragma unroll
  for (int j = 0; j < 6; ++j) fbg[0][j] = conv_buf_load(brsrc, bvoff, slab(kc0) + (unsigned)j * 1024u);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // this wave's share of stage 0 (older than the six weight requests)
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmB = __builtin_readcyclecounter();
  __syncthreads();
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmC = __builtin_readcyclecounter();
#pragma unroll
  for (int j = 0; j < 6; ++j) fbg[1][j] = conv_buf_load(brsrc, bvoff, slab(kc0 + 1) + (unsigned)j * 1024u);
  __builtin_amdgcn_sched_barrier(0);
  const int gtot = CHAIN ? nt * nsc : nsc;   // super-chunks of this workgroup (all its tiles)
  if (gtot > 1) raw_issue(1);
  if constexpr (NS > 2) { if (gtot > 2) raw_issue(2); }
  prepare(A[0], 0, C0{});

  // ---- K loop: chunk kc = (super-chunk s, half h).  Super-chunk s lives in stage s % 3; chunk kc prepares the fragments of chunk
  // kc + 1 while its own MFMAs run.  The barrier at the top of chunk (s, 1) publishes super-chunk s + 1 (first read right behind it)
  // and says that nobody reads stage s % 3 any more: super-chunk s + 3 goes there, its requests in gaps of chunks (s, 1), (s + 1, 0).
  // Two stages (NS = 2): super-chunk s + 2 goes to stage s % 2, ALL its requests in gaps of chunk (s, 1) - it then has chunk (s + 1, 0)
  // to land in, and the six weight requests of that chunk are younger than it (the vmcnt below).
  int st_s = 0, st_n = 1;    // stages of super-chunks s and s + 1
  int st_dma = 0;            // stage of the super-chunk whose requests are being issued
  bool dma_on = false;
  int gs = 0;                // super-chunks this workgroup has finished (over all its tiles)
  constexpr int P1 = NS == 2 ? IPW : (IPW + 1) / 2, P0 = IPW - P1;   // requests issued in the gaps of chunk (s, 1) / of chunk (s + 1, 0)
  // request index a gap carries (-1: none) when a chunk issues CNT of them: 1 -> gap 13; 2 -> 5, 17; 3 -> 5, 13, 21; 6 -> 2, 5, 9, 13, 17, 21
  // (never gaps 0 / 1: the fragment reads)
  auto dma_slot = [](int g, int cnt) constexpr -> int {
    if (cnt <= 0) return -1;
    if (cnt == 6) return g == 2 ? 0 : (g >= 5 && (g - 5) % 4 == 0) ? 1 + (g - 5) / 4 : -1;
    const int step = 24 / cnt, at = cnt == 1 ? 13 : 5;
    return (g % step == at && g / step < cnt) ? g / step : -1;
  };
  auto chunk = [&](int kc, auto h_c, auto first_c, auto last_c) {
    constexpr int H = decltype(h_c)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;   // the first chunk of a tile: its k = 0 MFMAs start the accumulators
    // The two chunks of the LAST super-chunk (round 6): no weight requests (chunk kc + 2 does not exist - they used to fetch the last slab again,
    // and the epilogue began by waiting for those twelve loads to return: its registers overlay fbg), no DMA, and in its second chunk no barrier
    // (nothing to publish), no fragment reads and no transforms (they prepared chunk kc + 1 for nobody).  Same MFMAs in the same order.
    // (Chained tiles run on into the next tile: no last chunk there.)
    constexpr bool LAST = decltype(last_c)::value != 0 && !CHAIN;
    constexpr bool PREP = !(LAST && H == 1);                // this chunk prepares the fragments of chunk kc + 1
    using RH = std::integral_constant<int, 1 - H>;
    const int rs = H == 0 ? st_s : st_n;   // stage of chunk kc + 1
    f2(&Ac)[6][2] = A[H];
    f2(&An)[6][2] = A[1 - H];
    if constexpr (H == 1 && !LAST) {
      // This wave's requests for super-chunk s + 1 are older than the weight requests of the last chunks (in-order return): the last
      // one went out in a gap of chunk kc - 3, at least 14 weight requests ago (s + 1 < 3: in the prologue, behind it DMA requests
      // and the 6 weight requests of chunk 0).  Everybody else's are published by the barrier.
      unsigned long long tw0 = 0;
      if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tw0 = __builtin_readcyclecounter();
      if constexpr ((FLAGS & W2D_DBG_NOB) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else {
        if (NS == 2 || gs == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      }
      if constexpr ((FLAGS & W2D_DBG_NOBAR) == 0) __syncthreads();
      if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmW += __builtin_readcyclecounter() - tw0;   // s_waitcnt + barrier of this super-chunk
      dma_on = CHAIN ? gs + NS < gtot : (kc >> 1) + NS < sc1;
      st_dma = st_s;
    }
    const unsigned so2 = LAST ? 0u : slab(kc + 2);
    __builtin_amdgcn_sched_barrier(0);
    auto gap = [&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int jp = g >> 3, i8 = g & 7, k = i8 >> 1, j = 2 * jp + (i8 & 1);
      // A = weights (row = output channel), B = activations (column = unit): C^T of conv_wino2d_r3_kernel's tile, the same k-ordered sums
      if constexpr (FIRST && k == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fbg[H][j][k], Ac[j][k >> 1][k & 1], zero, 0, 0, 0);
      } else {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fbg[H][j][k], Ac[j][k >> 1][k & 1], acc[j], 0, 0, 0);
      }
      if constexpr ((FLAGS & W2D_DBG_NOB) == 0 && !LAST) {
        if constexpr (k == 3) fbg[H][j] = conv_buf_load(brsrc, bvoff, so2 + (unsigned)j * 1024u);   // consumed: chunk kc + 2's slab into the same registers
      }
      if constexpr ((FLAGS & W2D_DBG_NORD) == 0 && PREP) {
        if constexpr (g == 0) read_row(d, ix_a, rs, RH{});
        if constexpr (!XF && g == 1) read_row(d2, ix_b, rs, RH{});
        if constexpr (XF && g == 10) read_row(d, ix_b, rs, RH{});
      }
      if constexpr (!PREP) {
      } else if constexpr (!XF) {
        if constexpr (g >= 4 && g < 10) {   // y combine of pixel g - 4: two packed operations
          constexpr int jj = g - 4;
#pragma unroll
          for (int q = 0; q < 2; ++q) { f2 v = fma2(sgn, pair_of(d2[jj], q), pair_of(d[jj], q)); pin(v); set_pair(d[jj], q, v); }
        }
        if constexpr (g >= 10 && g < 16) {   // x transform: channel pair (g - 10) / 3 in three parts of four packed operations
          constexpr int q = (g - 10) / 3, part = (g - 10) % 3;
          if constexpr (part == 0) {
            xf_t(q);
            pin(xt[0]); pin(xt[1]); pin(xt[2]); pin(xt[3]);
          } else if constexpr (part == 1) {
            An[0][q] = xf_v(q, 0);
            An[5][q] = xf_v(q, 5);
            pin(An[0][q]); pin(An[5][q]);
          } else {
#pragma unroll
            for (int nu = 1; nu < 5; ++nu) { An[nu][q] = xf_v(q, nu); pin(An[nu][q]); }
          }
        }
      } else {
        if constexpr (g >= 2 && g < 8) {           // row a: channel pair (g - 2) / 3 in three parts
          constexpr int q = (g - 2) / 3, part = (g - 2) % 3;
          if constexpr (part == 0) {
            xf_t(q);
            pin(xt[0]); pin(xt[1]); pin(xt[2]); pin(xt[3]);
          } else if constexpr (part == 1) {
            An[0][q] = xf_v(q, 0);
            An[5][q] = xf_v(q, 5);
            pin(An[0][q]); pin(An[5][q]);
          } else {
#pragma unroll
            for (int nu = 1; nu < 5; ++nu) { An[nu][q] = xf_v(q, nu); pin(An[nu][q]); }
          }
        }
        if constexpr (g >= 12 && g < 18) {         // row b + y combine
          constexpr int q = (g - 12) / 3, part = (g - 12) % 3;
          if constexpr (part == 0) {
            xf_t(q);
            pin(xt[0]); pin(xt[1]); pin(xt[2]); pin(xt[3]);
          } else if constexpr (part == 1) {
            An[0][q] = fma2(sgn, xf_v(q, 0), An[0][q]);
            An[5][q] = fma2(sgn, xf_v(q, 5), An[5][q]);
            pin(An[0][q]); pin(An[5][q]);
          } else {
#pragma unroll
            for (int nu = 1; nu < 5; ++nu) { An[nu][q] = fma2(sgn, xf_v(q, nu), An[nu][q]); pin(An[nu][q]); }
          }
        }
      }
      if constexpr ((FLAGS & W2D_DBG_NODMA) == 0 && !LAST) {   // a DMA request in a gap without fragment reads
        constexpr int CNT = H == 1 ? P1 : P0, N0 = H == 1 ? 0 : P1;
        constexpr int idx = dma_slot(g, CNT);
        if constexpr (idx >= 0) {
          if (dma_on) {
            dma_piece(N0 + idx, st_dma);
            if constexpr (N0 + idx == IPW - 1) dma_advance();   // behind the super-chunk's last request
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    w2d_for_each(gap, std::make_integer_sequence<int, 24>{});
    if constexpr (H == 1) {
      st_s = st_n;
      st_n = st_n + 1 == NS ? 0 : st_n + 1;
      ++gs;
    }
  };

  // ---- epilogue.  C/D layout of the 32x32 MFMA with A = weights: column = lane & 31 = unit, row = (r&3) + 8*(r>>2) + 4*(lane>>5) =
  // output channel of the wave's 32.  x inverse in registers (per mu plane; conv_wino2d_r3_kernel's sums: the products by 2, 4, 8 are
  // exact, so the fused forms round the same values), then per x position jx one round: every wave writes its plane
  // [unit][32 channels] (16-byte pieces swizzled by the unit: conflict-free both ways), barrier, thread (unit, 4-channel group)
  // reads the four mu planes and forms row 2k = (m0 + m1) + m2 and row 2k + 1 = (m1 - m2) - m3 (conv_wino2d_r3_kernel's), bias,
  // leaky_relu, two dwordx4 stores.  Two exchange buffers: the writes of round jx + 1 do not wait for the readers of round jx.
  auto epilogue = [&]() {
    // (chained: everything below that depends only on the thread index is loop invariant over the tiles, and hipcc hoists it out of the tile
    // loop - index registers, bias values and pointer bases then live through the K loop, which spilled.  An opaque copy of the thread
    // index keeps the epilogue's values inside the epilogue.)
    int te = (int)threadIdx.x;
    if constexpr (CHAIN) asm volatile("" : "+v"(te));
    const int t = te, lane = t & 63, l31 = lane & 31, half = lane >> 5;
    if constexpr (!CHAIN) __syncthreads();   // the exchange buffers overlay the stages (every wave's fragment reads are done)
    if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tmE1 = __builtin_readcyclecounter();
                                             // (chained: they lie behind the stages, which already hold the next tile's first super-chunks; a buffer's
                                             // next writers - round jx of the NEXT tile - are a whole K loop of barriers behind its last readers)
    float o[4][16];
  #pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
      const float s34 = m3 + m4, d34 = m3 - m4, d12 = m1 - m2, s12 = m1 + m2;
      o[0][r] = ((m0 + m1) + m2) + s34;
      o[1][r] = __builtin_fmaf(2.f, d34, d12);
      o[2][r] = __builtin_fmaf(4.f, s34, s12);
      o[3][r] = d12 + __builtin_fmaf(8.f, d34, m5);
    }
    bf4* const xb = reinterpret_cast<bf4*>(smem) + (CHAIN ? NS * STAGE4 : 0);   // [buffer 2][ng][mu][unit 32][piece 8] float4
    constexpr int XB4 = NG * 4 * 256;
    constexpr bool E1 = (FLAGS & W2D_F_EPI1) != 0 && !CHAIN;
    static_assert(CHAIN || E1 || 2 * XB4 <= NS * STAGE4, "exchange buffers");
    const int widx = (ng * 4 + mu) * 256 + l31 * 8;   // + ((2 g + half) ^ (unit & 7))
    const int rng = t >> 8, run = (t >> 3) & 31, rcg = t & 7;   // reader: channel tile, unit, 4-channel group
    const int ridx = rng * 1024 + run * 8 + (rcg ^ (run & 7));  // + mu * 256
    const int nrd = n0 + rng * 32 + rcg * 4;
    const bool rawsum = XE && ksp > 1;   // split-K: no bias, no activation, [split][pixel][Cout]
    const bool has_pw = XE && p.pw_out != nullptr;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (!rawsum) { const float* const bp = p.bias + nrd; b0 = bp[0]; b1 = bp[1]; b2 = bp[2]; b3 = bp[3]; }   // (one branch: four ternaries were four)
    const int oy = y0 + 2 * (run / QW), ox = x0 + 4 * (run % QW);
    const int ostr = rawsum ? p.Cout : p.ostride;
    const float slope = (p.leaky && !rawsum) ? 0.2f : 1.f;
    // (the tile's corner in 64-bit SCALAR arithmetic, the thread's pixel inside the tile with 24-bit multiplies: the one 64-bit expression
    // was six v_mul_lo_u32 + three v_mad_u64_u32, quarter rate, per pointer; 8 rows x W x the pixel pitch < 2^31: the launcher)
    const int dyo = 2 * (run / QW), dxo = 4 * (run % QW), nth = rng * 32 + rcg * 4;
    // (the corner's PIXEL index in 32 bits - a launch has fewer than 2^31 pixels, the launcher checks - then one 32 x 32 -> 64-bit product with the pixel
    // pitch: the all-64-bit expression was ~30 scalar instructions per pointer, and with two workgroups per CU every instruction outside the K loop waits
    // ~18 cycles for its issue slot)
    const int cpix = (img * p.H + y0) * p.W + x0;
    float* obase = p.out;
    if (rawsum) { asm volatile(""); obase = p.part + (size_t)blockIdx.z * p.M * p.Cout; }   // (the empty statement keeps it a branch: hipcc computed the 64-bit product for every launch)
    float* const ocorner = obase + ((long long)cpix * ostr + n0);
    float* const orow = ocorner + (__umul24(__umul24(dyo, p.W) + dxo, ostr) + nth);
    // Fused AveragePooling2D(2, 2) of the activated output (ConvParams::pool_out; H, W even): the thread holds rows 2k, 2k + 1 of its
    // unit; x = 4 q + jx pairs up over two rounds: (((o(y,x) + o(y,x+1)) + o(y+1,x)) + o(y+1,x+1)) * 0.25, pool_vec_kernel's order.
    // Fused 1x1 convolution (ConvParams::pw_out; BN = Cout = 64): the activated tile goes to LDS as [pixel 256][68] behind the exchange
    // buffers instead of to `out` (68: 16-byte rows whose float4 pieces fall into 16 different bank groups for 16 consecutive pixels -
    // conflict-free ds_write_b128 from the (unit, channel group) threads and ds_read_b128 from the pixel threads); the 1x1 weights go
    // to LDS once as [c][4]; thread = pixel then sums its 64 channels, one fma chain per output in channel order (conv_pw_kernel's).
    float* pool_base = nullptr;
    if (XE && p.pool_out) {
      asm volatile("");
      const int ppix = (img * (p.H >> 1) + (y0 >> 1)) * (p.W >> 1) + (x0 >> 1);
      pool_base = p.pool_out + ((long long)ppix * p.pool_ostride + n0) + (__umul24(__umul24(dyo >> 1, p.W >> 1) + (dxo >> 1), p.pool_ostride) + nth);
    }
    constexpr int PWS = 68;
    float* const pwt = smem + 2 * XB4 * 4;   // (never with W2D_F_CHAIN: the launcher refuses the fused 1x1 there)
    float* const pww = pwt + TH * PXW * PWS;   // [64][4]
    if (has_pw && t < 256) pww[t] = (t & 3) < p.pw_cout ? p.pw_w[(t >> 2) * p.pw_cout + (t & 3)] : 0.f;   // (published by the rounds' barriers)
    bf4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = {0.f, 0.f, 0.f, 0.f};
    auto write_round = [&](int jx, bf4* xw) {
  #pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf4 v;
        v[0] = o[jx][4 * g]; v[1] = o[jx][4 * g + 1]; v[2] = o[jx][4 * g + 2]; v[3] = o[jx][4 * g + 3];
        xw[widx + ((2 * g + half) ^ (l31 & 7))] = v;
      }
    };
    if constexpr (E1) {
  #pragma unroll
      for (int jx = 0; jx < 4; ++jx) write_round(jx, xb + jx * XB4);
      __syncthreads();
    }
  #pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      bf4* const xw = xb + (E1 ? jx : (jx & 1)) * XB4;
      if constexpr (!E1) {
        write_round(jx, xw);
        __syncthreads();
      }
      if constexpr ((FLAGS & W2D_DBG_TIME) != 0) { if (jx == 0) tmE2 = __builtin_readcyclecounter(); if (jx == 3) tmE3 = __builtin_readcyclecounter(); }
      const bf4 m0 = xw[ridx], m1 = xw[ridx + 256], m2 = xw[ridx + 512], m3 = xw[ridx + 768];
      // Packed, and without compare / select: max(v, slope v) is leaky_relu(0.2) for slope = 0.2 and v itself for slope = 1 (same bits as the
      // v > 0 ? v : 0.2 v form, NaN and -0 included); 24 instead of 46 vector instructions per round (they are taken from the matrix pipe of
      // the co-resident workgroup's K loop: profiles/r06_w2d_valu_diet.log)
      bf4 r0, r1;
  #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f2 bq = q ? f2{b2, b3} : f2{b0, b1};
        const f2 v0 = add2(add2(add2(pair_of(m0, q), pair_of(m1, q)), pair_of(m2, q)), bq);
        const f2 v1 = add2(sub2(sub2(pair_of(m1, q), pair_of(m2, q)), pair_of(m3, q)), bq);
        const f2 s0 = mul2(slope, v0), s1 = mul2(slope, v1);
        set_pair(r0, q, f2{max1(v0[0], s0[0]), max1(v0[1], s0[1])});   // (asm: fmaxf on the results of asm statements costs two canonicalising
        set_pair(r1, q, f2{max1(v1[0], s1[0]), max1(v1[1], s1[1])});   // v_max_f32 x, x, x more per element)
      }
      if (!has_pw) {
        if (ox + jx < p.W) {
          float* const o0 = orow + (size_t)jx * ostr;
          if (oy < p.H) *reinterpret_cast<bf4*>(o0) = r0;
          if (oy + 1 < p.H) *reinterpret_cast<bf4*>(o0 + (size_t)p.W * ostr) = r1;
        }
        if (XE && pool_base != nullptr) {
          if (jx & 1) {
            bf4 pv;
  #pragma unroll
            for (int c = 0; c < 4; ++c) pv[c] = (((k0[c] + r0[c]) + k1[c]) + r1[c]) * 0.25f;
            if (oy < p.H && ox + jx < p.W) *reinterpret_cast<bf4*>(pool_base + (size_t)(jx >> 1) * p.pool_ostride) = pv;
          } else {
            k0 = r0; k1 = r1;
          }
        }
      } else {
        float* const tr = pwt + ((2 * (run / QW)) * PXW + 4 * (run % QW) + jx) * PWS + rng * 32 + rcg * 4;
        *reinterpret_cast<bf4*>(tr) = r0;
        *reinterpret_cast<bf4*>(tr + PXW * PWS) = r1;
      }
    }
    if (has_pw) {
      __syncthreads();
      if (t < TH * PXW) {
        const int y = y0 + t / PXW, x = x0 + (t % PXW);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const bf4* row = reinterpret_cast<const bf4*>(pwt + t * PWS);
        const bf4* wq = reinterpret_cast<const bf4*>(pww);
  #pragma unroll 4
        for (int c4 = 0; c4 < 16; ++c4) {
          const bf4 v = row[c4];
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bf4 w = wq[c4 * 4 + k];
  #pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = __builtin_fmaf(v[k], w[j], a[j]);   // (columns past pw_cout: zero weights, never stored)
          }
        }
        if (y < p.H && x < p.W) {
          float* dd = p.pw_out + (((size_t)img * p.H + y) * p.W + x) * p.pw_ostride;
  #pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < p.pw_cout) dd[j] = a[j] + p.pw_bias[j];
        }
      }
    }
  };
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) tm1 = __builtin_readcyclecounter();
  for (int ti = 0; ti < nt; ++ti) {
    if (ti > 0) {   // the next tile of the chain: its first super-chunks are in the stages, its first two weight slabs in fbg, the
                    // fragments of its chunk 0 in A[0] (prepared by the last chunk of the tile before, like any chunk kc + 1)
      const int g = g0 + ti;
      img = g / tpi;
      const int tr = g - img * tpi;
      y0 = (tr / ntx) * TH; x0 = (tr % ntx) * PXW;
    }
    chunk(kc0, C0{}, C1{}, C0{});
    chunk(kc0 + 1, C1{}, C0{}, C0{});
    const int kc_end = CHAIN ? kc1 : kc1 - 2;
    for (int kc = kc0 + 2; kc < kc_end; kc += 2) {
      chunk(kc, C0{}, C0{}, C0{});
      chunk(kc + 1, C1{}, C0{}, C0{});
    }
    if constexpr (!CHAIN) {
      if (nsc > 1) {   // (a K range of one super-chunk - K = 16, or a split's remainder - ends with the ordinary pair above)
        chunk(kc1 - 2, C0{}, C0{}, C1{});
        chunk(kc1 - 1, C1{}, C0{}, C1{});
      }
    }
    if constexpr ((FLAGS & W2D_DBG_TIME) != 0) { if (ti + 1 == nt) tm2 = __builtin_readcyclecounter(); }
    epilogue();
    if constexpr (CHAIN) {
      if (ti + 1 < nt) {
        // Only the next tile's FIRST weight slab (fbg[0], requested by this tile's second-to-last chunk) lives through the epilogue: the second
        // one and the fragments of its chunk 0 are formed again here - 48 registers the epilogue needs (it spilled with them alive).
#pragma unroll
        for (int j = 0; j < 6; ++j) fbg[1][j] = conv_buf_load(brsrc, bvoff, slab(kc0 + 1) + (unsigned)j * 1024u);
        __builtin_amdgcn_sched_barrier(0);
        prepare(A[0], st_s, C0{});
      }
    }
  }
  if constexpr ((FLAGS & W2D_DBG_TIME) != 0) {
    const unsigned long long tm3 = __builtin_readcyclecounter();
    if (t == 0) {
      // 16 slots per workgroup; 8 / 9: the 100 MHz wall clock at entry / exit (s_memtime counts shader cycles, per CU group: only stamps of
      // one CU compare); 10: HW_ID (bits 8..15: CU, SH, SE) | XCC_ID << 32 - which CU ran this workgroup
      unsigned long long* o8 = reinterpret_cast<unsigned long long*>(p.part) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
      o8[0] = tm0; o8[1] = tm1; o8[2] = tm2; o8[3] = tm3; o8[4] = tmA; o8[5] = tmB; o8[6] = tmC; o8[7] = tmW;
      o8[8] = rt0; o8[9] = __builtin_amdgcn_s_memrealtime();
      o8[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      o8[11] = tmE1; o8[12] = tmE2; o8[13] = tmE3; o8[14] = tmK; o8[15] = tmT;   // epilogue: behind its entry barrier / behind the barriers of exchange rounds 0 and 3
    }
  }
}

inline int& conv_wino2d_debug_extra_lds() { static int v = 0; return v; }   // tools only: bytes of dynamic LDS added to a launch (occupancy experiments)

template <int BN, int FLAGS, int NS = 3>
hipError_t conv_wino2d_launch(const ConvParams& p, hipStream_t s) {
  constexpr bool CHAIN = (FLAGS & W2D_F_CHAIN) != 0;
  // NS stages of 24 KB; the exchange buffers (2 x BN / 32 x 16 KB) fit inside them - or, chained, lie behind them (80 KB for the 32-channel
  // tile on two stages: two workgroups per CU; 136 KB for the 64-channel one); the fused 1x1 adds its [256][68] tile and its weights
  constexpr bool E1 = (FLAGS & W2D_F_EPI1) != 0 && !CHAIN;
  if (!(FLAGS & W2D_F_XEPI) && (p.ksplit > 1 || p.pool_out || p.pw_out)) return hipErrorInvalidValue;   // (conv_wino2d_launch_any picks the instantiation)
  if (E1 && p.pw_out) return hipErrorInvalidValue;   // (the fused 1x1's pixel tile lies behind TWO exchange buffers)
  const size_t lds = CHAIN ? (size_t)NS * 24 * 1024 + (size_t)2 * (BN / 32) * 16 * 1024
                           : p.pw_out ? (size_t)2 * (BN / 32) * 16 * 1024 + 256 * 68 * 4 + 1024
                           : E1 ? std::max((size_t)NS * 24 * 1024, (size_t)4 * (BN / 32) * 16 * 1024) : (size_t)NS * 24 * 1024;
  constexpr int NT = 4 * (BN / 32) * 64;
  if (p.ksize != 3 || p.Ctot % 16 || p.Cout % BN) return hipErrorInvalidValue;
  if (CHAIN && (p.ksplit > 1 || p.pw_out || p.chain < 1)) return hipErrorInvalidValue;
  if (p.ksplit > 1 && (!p.part || (reinterpret_cast<uintptr_t>(p.part) & 15) || p.Cout % 4 || p.pool_out || p.pw_out || p.ksplit > p.Ctot / 16)) return hipErrorInvalidValue;
  if (p.pw_out) {   // a workgroup must hold every channel of its pixels
    if (BN != 64 || p.Cout != 64 || p.pool_out || p.pw_cout < 1 || p.pw_cout > 4) return hipErrorInvalidValue;
  } else if (p.ostride % 4 || (reinterpret_cast<uintptr_t>(p.out) & 15)) return hipErrorInvalidValue;   // dwordx4 stores
  if (p.pool_out && ((p.H | p.W) & 1 || p.pool_ostride % 4 || (reinterpret_cast<uintptr_t>(p.pool_out) & 15))) return hipErrorInvalidValue;
  {   // pixel indices in 32 bits (tile corners, halo corners: one more row)
    long long nimg = p.NB;
    for (int i = 0; i < p.nseg; ++i) nimg = std::max<long long>(nimg, (long long)p.seg[i].bmod);
    if ((nimg + 1) * p.H * p.W >= (1ll << 31)) return hipErrorInvalidValue;
  }
  if (p.W <= 0 || p.H <= 0 || p.W >= (1 << 20) || p.ostride >= (1 << 22) || p.pool_ostride >= (1 << 22) || p.Cout >= (1 << 22) ||
      (long long)8 * p.W * (p.ostride > p.Cout ? p.ostride : p.Cout) >= (1ll << 31)) return hipErrorInvalidValue;   // 24-bit multiplies of the DMA offsets: 10 halo rows x W pixels < 2^24
  for (int i = 0; i < p.nseg; ++i)
    if ((long long)p.W * p.seg[i].stride * 4 >= (1ll << 24) || p.seg[i].C % 16 || p.seg[i].stride % 4 || p.seg[i].stride <= 0 || p.seg[i].stride >= (1 << 22) || p.seg[i].up || (reinterpret_cast<uintptr_t>(p.seg[i].ptr) & 15)) return hipErrorInvalidValue;
  auto kern = conv_wino2d_kernel<BN, FLAGS, NS>;
  static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
  if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, 144 * 1024); e != hipSuccess) return e;
  constexpr bool SQ = (FLAGS & W2D_F_SQ) != 0;
  const int ntx = SQ ? (p.W + 15) / 16 : (p.W + 31) / 32, nty = SQ ? (p.H + 15) / 16 : (p.H + 7) / 8;
  const int ntiles = p.NB * ntx * nty, chain = CHAIN ? p.chain : 1;
  dim3 grid((unsigned)((ntiles + chain - 1) / chain), p.Cout / BN, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
  ConvParams q = p;   // + the reciprocals of the workgroup decomposition: ceil(2^32 / d), exact for x d < 2^32; 0 = the divisor is 1 (the kernel has no division code)
  bool exact = true;
  auto magic = [&exact](unsigned long long d, unsigned long long xmax) -> unsigned {
    if (d <= 1) return 0u;
    if (xmax * d >= (1ull << 32)) { exact = false; return 0u; }
    return (unsigned)(((1ull << 32) + d - 1) / d);
  };
  q.mg_nby = magic(grid.y, (unsigned long long)grid.x * grid.y);
  q.mg_tpi = magic((unsigned long long)ntx * nty, (unsigned long long)ntiles + chain);
  q.mg_ntx = magic(ntx, (unsigned long long)ntx * nty);
  q.tl_ntx = ntx; q.tl_tpi = ntx * nty;
  if (!exact) return hipErrorInvalidValue;   // (> 2^32 / tiles-per-image workgroups: no plan comes near)
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds + (size_t)conv_wino2d_debug_extra_lds(), s, q);
  return hipGetLastError();
}

// the instantiation a ConvParams block needs: the extended epilogue only for split-K, fused pooling, fused 1x1
template <int BN, int FLAGS, int NS = 3>
hipError_t conv_wino2d_launch_any(const ConvParams& p, hipStream_t s) {
  if (p.ksplit > 1 || p.pool_out || p.pw_out) return conv_wino2d_launch<BN, FLAGS | W2D_F_XEPI, NS>(p, s);
  return conv_wino2d_launch<BN, FLAGS, NS>(p, s);
}
