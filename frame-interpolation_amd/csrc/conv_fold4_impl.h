// conv_fold4_impl.h -- nearest-neighbour x2 upsample + Conv2D 2x2 ('same': pad 0 before / 1 after) of the fusion decoder
// (fusion.py:83-84,133-135) in its DIFFERENCE form, fp32 MFMA: FOUR multiplies per low-resolution pixel and (ci, co) for the four
// output pixels it produces - the sub-pixel fold of conv_buf_kernel (film_kernels.h, fold = 2) needs nine, the reference op sixteen.
//
//   out(2y + py, 2x + px) = sum_{dy, dx} W[dy][dx] . I(y + (py & dy), x + (px & dx))            (the fold; I = 0 beyond the bottom / right edge)
//   with  Dx = I(y, x) - I(y, x+1),  Dy = I(y, x) - I(y+1, x),  Dxy = Dx - (I(y+1, x) - I(y+1, x+1)):
//     I(y, x+1) = I - Dx,   I(y+1, x) = I - Dy,   I(y+1, x+1) = I - Dx - Dy + Dxy,   so
//   G0 = S . I      S  = ((W00 + W01) + W10) + W11           out(2y,   2x)   = G0
//   G1 = Sx . Dx    Sx = W01 + W11                           out(2y,   2x+1) = G0 - G1
//   G2 = Sy . Dy    Sy = W10 + W11                           out(2y+1, 2x)   = G0 - G2
//   G3 = W11 . Dxy                                           out(2y+1, 2x+1) = ((G0 - G1) - G2) + G3
// Four GEMMs over K = Cin that share one pixel tile; the differences are 8 packed VALU operations per 8-channel chunk and lane against
// 32 MFMAs (conv_wino2d_kernel: 36 per 24), the output combination happens in the accumulator registers of ONE wave (no exchange).
// An exact regrouping of the reference sum; the roundings differ (1e-7 relative, like the Winograd kernels): its own summation family.
//
// Structure = conv_wino2d_kernel's (conv_wino2d_impl.h; the measurements behind it are cited there): the RAW halo patch of a 16-channel
// super-chunk ((TH + 1) x 33 low-resolution pixels) goes to LDS by DMA (`buffer_load_dwordx4 ... lds`), three stages, one barrier per
// super-chunk; a wave owns 4 rows x 8 pixels x 32 NCT output channels x the four planes (64 NCT accumulator registers), reads its four
// raw pixels per chunk (4 ds_read_b128) and forms the differences in the MFMA gaps of the previous chunk; the weight slab of a chunk
// ([Cout / 32][chunk][plane 4][K half][32][4]: 1 KB per (32-channel tile, plane)) goes straight from L2 into registers and is re-requested for
// the next chunk behind the last MFMA of its plane.  MFMA order: plane, k, ct - the two accumulators of a plane alternate.
//
// LDS layout of a stage in 16-byte slots: halo row r at r * 148; pixel px, 16-byte piece c of its 64 bytes at (px >> 2) * 16 + c * 4 +
// (px & 3).  A DMA request (64 consecutive slots) then covers the whole 64-byte sectors of 16 pixels, and a fragment read (lane = row
// l31 >> 3, pixel 8 wv + (l31 & 7) + dc, piece 2 h + half) is conflict free: four consecutive pixels are four consecutive slots mod 16
// and the rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) are 148 = 4 (mod 16) slots apart.
//
// Needs one input segment with C % 16 == 0 (16-byte aligned pixels), Cout % (32 NCT) == 0, a 16-byte aligned output slice.
#pragma once
#include "conv_buf_impl.h"

enum { F4_DBG_TIME = 8192 };   // tools only: wave 0 of every workgroup stamps s_memtime like W2D_DBG_TIME (p.part[workgroup * 16 ..])

template <class F, int... G>
__device__ __forceinline__ void f4_for_each(F&& f, std::integer_sequence<int, G...>) { (f(std::integral_constant<int, G>{}), ...); }

template <int NCT, int FLAGS>
__global__ __launch_bounds__(256, 2) void conv_fold4_kernel(ConvParams p) {
  constexpr int TH = 4, PXW = 32, HR = TH + 1, PW = PXW + 1, NW = 4;
  constexpr int BN = 32 * NCT;
  constexpr int RP4 = 148;                      // row pitch in 16-byte slots (9 pixel quads x 16 slots = 144, + 4: = 4 mod 16)
  constexpr int NREQ = 12;                      // DMA requests (1 KB each) per stage: 5 rows x 148 slots = 740 <= 768
  constexpr int STAGE4 = NREQ * 64;
  constexpr int NS = 3;
  constexpr int IPW = NREQ / NW;                // requests per wave and super-chunk
  constexpr int NB8 = 4 * NCT;                  // weight requests (1 KB each) per wave and chunk
  constexpr int NG = 16 * NCT;                  // MFMAs (= gaps) per chunk
  static_assert(HR * RP4 <= STAGE4 && NREQ % NW == 0, "stage size");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // [stage 0][stage 1][stage 2]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tmW = 0, rt0 = 0;
  if constexpr ((FLAGS & F4_DBG_TIME) != 0) { rt0 = __builtin_amdgcn_s_memrealtime(); tm0 = __builtin_readcyclecounter(); }

  // (round 6, conv_wino2d_kernel's recipe: every kernel argument the prologue reads in ONE batch of scalar loads, the workgroup decomposition by the launcher's
  // reciprocals - there were three 40-instruction integer divisions here - and tile counts from the launcher; outside the K loop every instruction waits ~18
  // cycles for an issue slot behind the co-resident workgroup's K loop)
  int aH = p.H, aW = p.W, aCtot = p.Ctot, aKsplit = p.ksplit;
  unsigned a_tpi = p.mg_tpi, a_ntx = p.mg_ntx, a_nby = p.mg_nby;
  int ntx_ = p.tl_ntx, tpi_ = p.tl_tpi, aGx = (int)gridDim.x, aGy = (int)gridDim.y;
  const float* aS0ptr = p.seg[0].ptr;
  int aS0stride = p.seg[0].stride, aS0boff = p.seg[0].boff, aS0bmod = p.seg[0].bmod;
  const float* aWptr = p.w;
  {
    unsigned long long q0 = (unsigned long long)(uintptr_t)aS0ptr, q1 = (unsigned long long)(uintptr_t)aWptr;
    asm volatile("" : "+s"(aH), "+s"(aW), "+s"(aCtot), "+s"(aKsplit), "+s"(a_tpi), "+s"(a_ntx), "+s"(a_nby), "+s"(ntx_), "+s"(tpi_), "+s"(aGx), "+s"(aGy),
                      "+s"(q0), "+s"(q1), "+s"(aS0stride), "+s"(aS0boff), "+s"(aS0bmod));
    aS0ptr = reinterpret_cast<const float*>((uintptr_t)q0);
    aWptr = reinterpret_cast<const float*>((uintptr_t)q1);
  }
  auto udiv = [](unsigned x, unsigned magic) -> unsigned {   // x / d by the launcher's reciprocal; magic = 0: d = 1
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
  const int ntx = ntx_;
  const int img = (int)udiv((unsigned)bx, a_tpi);
  const int trem = bx - img * tpi_;
  const int trow = (int)udiv((unsigned)trem, a_ntx);
  const int y0 = trow * TH, x0 = (trem - trow * ntx) * PXW;
  const int n0 = by * BN;

  auto uniform_ptr = [](const float* q) -> const float* {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v)));
  };
  // ---- DMA side: request n of this wave fills slots 64 * (wv + NW n) ... + 63 of a stage; lane -> (halo row, pixel, piece) -----
  int be = img + aS0boff;
  if (aS0bmod && be >= aS0bmod) be -= aS0bmod;
  const int spix = (be * aH + y0) * aW;   // (a launch has fewer than 2^31 pixels: the launcher checks)
  const conv_rsrc_t rrsrc = conv_make_rsrc(uniform_ptr(aS0ptr + (long long)spix * aS0stride));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  unsigned rvoff[IPW];
  {
    const unsigned st4 = (unsigned)aS0stride * 4u;
#pragma unroll
    for (int n = 0; n < IPW; ++n) {
      const int sl = 64 * (wv + NW * n) + lane;
      const int r = sl / RP4, rem = sl - r * RP4;
      const int px = ((rem >> 4) << 2) | (rem & 3), c = (rem >> 2) & 3;
      const int y = y0 + r, x = x0 + px;
      const bool ok = r < HR && rem < 144 && px < PW && y < aH && x < aW;   // else padding / beyond the bottom or right edge: zeros
      rvoff[n] = ok ? (unsigned)(r * aW + x) * st4 + (unsigned)c * 16u : OOB;
    }
  }
  // split-K (ConvParams::ksplit): blockIdx.z = split z sums the super-chunks [sc0, sc1) and writes RAW sums of the four output pixels to
  // part[z][output pixel][Cout] (the outputs are linear in the four planes); conv_splitk_reduce_kernel adds them in split order with the bias.
  // For the decoder's coarsest layer (36x60 low-resolution pixels, K = 1936: 1152 workgroups on 768 slots = 1.5 rounds of 0.3 ms).
  const int ksp = aKsplit > 1 ? aKsplit : 1;
  const int nsc_all = aCtot >> 4;
  const int sc0 = ksp > 1 ? (int)((unsigned)nsc_all * blockIdx.z / (unsigned)ksp) : 0;
  const int sc1 = ksp > 1 ? (int)((unsigned)nsc_all * (blockIdx.z + 1u) / (unsigned)ksp) : nsc_all;
  int rc0 = sc0 * 16;   // first channel of the DMA cursor's super-chunk
  auto dma_piece = [&](int n, int stage) {
    const unsigned so = (unsigned)rc0 * 4u;
    const unsigned base = lds0 + (unsigned)stage * (STAGE4 * 16u) + (unsigned)(wv + NW * n) * 1024u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(base), "v"(rvoff[n]), "s"(rrsrc), "s"(so)
                 : "memory");
  };
  auto raw_issue = [&](int stage) {
#pragma unroll
    for (int n = 0; n < IPW; ++n) dma_piece(n, stage);
    rc0 += 16;
  };

  // ---- weights: [Cout / 32][chunk][plane 4][K half][32][4] floats; every wave of the workgroup reads the slabs of the NCT tiles -----
  const int nkc = aCtot / 8, nsc = sc1 - sc0, kc0 = 2 * sc0, kc1 = 2 * sc1;
  const conv_rsrc_t brsrc = conv_make_rsrc(uniform_ptr(aWptr));
  const unsigned bvoff = (unsigned)((half * 32 + l31) * 16);
  const unsigned tstep = (unsigned)nkc * 4096u;   // bytes between two 32-channel tiles
  auto slab = [&](int kc) { return (unsigned)(by * NCT * nkc + (kc < nkc ? kc : nkc - 1)) * 4096u; };
  bf4 fb[4][NCT];

  f32x16 acc[4][NCT];   // never cleared: the k = 0 MFMAs of the first chunk take C = 0 (64 NCT v_mov less in front of the K loop)

  // ---- fragments: lane (row lr, pixel 8 wv + lc, K half) reads the raw pixels (lr + dr, px + dc), dr, dc in {0, 1} ---------------
  const bf4* const smem4 = reinterpret_cast<const bf4*>(smem);
  const int lr = l31 >> 3, lc = l31 & 7;
  const int pxa = 8 * wv + lc, pxb = pxa + 1;
  const int ix_a = lr * RP4 + ((pxa >> 2) << 4) + (pxa & 3) + half * 4;   // dc = 0 (+ RP4: dr = 1; + 8: the second chunk of the super-chunk)
  const int ix_b = lr * RP4 + ((pxb >> 2) << 4) + (pxb & 3) + half * 4;   // dc = 1
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 A[2][4][2];      // A[kc & 1][plane]: the lane's four channels of chunk kc as two pairs
  bf4 d[4];           // raw pixels (0,0) (0,1) (1,0) (1,1)
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  auto read_raw = [&](int stage, auto h_c) {
    constexpr int IMM = decltype(h_c)::value * 8;
    const int o = stage * STAGE4 + IMM;
    d[0] = smem4[ix_a + o]; d[1] = smem4[ix_b + o]; d[2] = smem4[ix_a + o + RP4]; d[3] = smem4[ix_b + o + RP4];
  };
  auto pair_of = [](const bf4& v, int q) -> f2 { return q ? f2{v[2], v[3]} : f2{v[0], v[1]}; };
  auto sub2 = [](f2 x, f2 y) -> f2 { f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto pin = [](f2& v) { asm volatile("" : "+v"(v)); };
  // the planes of channel pair q from the raw pixels in d: I, Dx, Dy, Dxy (four packed subtractions)
  auto planes = [&](f2 (&An)[4][2], int q) {
    const f2 i00 = pair_of(d[0], q), i01 = pair_of(d[1], q), i10 = pair_of(d[2], q), i11 = pair_of(d[3], q);
    An[0][q] = i00;
    An[1][q] = sub2(i00, i01);
    An[2][q] = sub2(i00, i10);
    An[3][q] = sub2(An[1][q], sub2(i10, i11));
  };

  // ---- prologue (conv_wino2d_kernel's order: stage 0 and the first slab alone, the rest behind the stage-0 barrier) ---------------
  raw_issue(0);
#pragma unroll
  for (int j = 0; j < NB8; ++j) fb[j / NCT][j % NCT] = conv_buf_load(brsrc, bvoff, slab(kc0) + (unsigned)(j % NCT) * tstep + (unsigned)(j / NCT) * 1024u);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB8) : "memory");   // this wave's share of stage 0 (older than the weight requests)
  __syncthreads();
  if (nsc > 1) raw_issue(1);
  if (nsc > 2) raw_issue(2);
  read_raw(0, C0{});
  planes(A[0], 0);
  planes(A[0], 1);

  // ---- K loop: chunk kc = (super-chunk s, half h); super-chunk s lives in stage s % 3; chunk kc prepares the fragments of chunk kc + 1
  // during its own MFMAs.  The barrier at the top of chunk (s, 1) publishes super-chunk s + 1 and frees stage s % 3 for super-chunk
  // s + 3, whose requests go out in gaps of chunks (s, 1) and (s + 1, 0).
  int st_s = 0, st_n = 1, st_dma = 0;
  bool dma_on = false;
  constexpr int P1 = (IPW + 1) / 2, P0 = IPW - P1;
  static_assert(P0 > 0, "the cursor advances behind the last request of chunk (s + 1, 0)");
  auto chunk = [&](int kc, auto h_c, auto first_c, auto last_c) {
    constexpr int H = decltype(h_c)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;   // the first chunk: its k = 0 MFMAs start the accumulators
    // the two chunks of the LAST super-chunk: no DMA code, and in the second one no barrier, no weight requests (chunk kc + 1 does not exist: the epilogue
    // used to wait for those loads), no raw reads, no differences - conv_wino2d_kernel's last-chunk form
    constexpr bool LAST = decltype(last_c)::value != 0;
    constexpr bool PREP = !(LAST && H == 1);
    using RH = std::integral_constant<int, 1 - H>;
    const int rs = H == 0 ? st_s : st_n;   // stage of chunk kc + 1
    f2(&Ac)[4][2] = A[H];
    f2(&An)[4][2] = A[1 - H];
    if constexpr (H == 1 && !LAST) {
      // This wave's requests for super-chunk s + 1 went out in chunks kc - 3 / kc - 2 at the latest (s + 1 < 3: in the prologue), i.e. in
      // front of the 2 NB8 weight requests of chunks kc - 2 and kc - 1 (in-order return); everybody else's are published by the barrier.
      unsigned long long tw0 = 0;
      if constexpr ((FLAGS & F4_DBG_TIME) != 0) tw0 = __builtin_readcyclecounter();
      if (kc == kc0 + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB8) : "memory");   // (stage 1: only chunk 0's weight requests are younger for sure)
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB8) : "memory");
      __syncthreads();
      if constexpr ((FLAGS & F4_DBG_TIME) != 0) tmW += __builtin_readcyclecounter() - tw0;
      dma_on = (kc >> 1) + NS < sc1;
      st_dma = st_s;
    }
    const unsigned so1 = PREP ? slab(kc + 1) : 0u;
    __builtin_amdgcn_sched_barrier(0);
    auto gap = [&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int q = g / (4 * NCT), k = (g / NCT) & 3, c = g % NCT;
      if constexpr (FIRST && k == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[q][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[q][c][k], Ac[q][k >> 1][k & 1], zero, 0, 0, 0);
      } else {
        acc[q][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[q][c][k], Ac[q][k >> 1][k & 1], acc[q][c], 0, 0, 0);   // A = weights, B = pixels: C^T
      }
      if constexpr (PREP) {
        if constexpr (k == 3) fb[q][c] = conv_buf_load(brsrc, bvoff, so1 + (unsigned)c * tstep + (unsigned)q * 1024u);   // consumed: the next chunk's slab
        if constexpr (g == 0) read_raw(rs, RH{});
        if constexpr (g == 4) { planes(An, 0); pin(An[0][0]); pin(An[1][0]); pin(An[2][0]); pin(An[3][0]); }
        if constexpr (g == 6) { planes(An, 1); pin(An[0][1]); pin(An[1][1]); pin(An[2][1]); pin(An[3][1]); }
      }
      if constexpr (!LAST) {   // a DMA request in a gap without fragment reads / differences
        constexpr int CNT = H == 1 ? P1 : P0, N0 = H == 1 ? 0 : P1;
        constexpr int step = (NG - 9) / CNT;   // gaps 9, 9 + step, ...
        if constexpr (g >= 9 && (g - 9) % step == 0 && (g - 9) / step < CNT) {
          if (dma_on) {
            dma_piece(N0 + (g - 9) / step, st_dma);
            if constexpr (H == 0 && (g - 9) / step == CNT - 1) rc0 += 16;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    f4_for_each(gap, std::make_integer_sequence<int, NG>{});
    if constexpr (H == 1) {
      st_s = st_n;
      st_n = st_n + 1 == NS ? 0 : st_n + 1;
    }
  };
  if constexpr ((FLAGS & F4_DBG_TIME) != 0) tm1 = __builtin_readcyclecounter();
  chunk(kc0, C0{}, C1{}, C0{});
  chunk(kc0 + 1, C1{}, C0{}, C0{});
  for (int kc = kc0 + 2; kc < kc1 - 2; kc += 2) {
    chunk(kc, C0{}, C0{}, C0{});
    chunk(kc + 1, C1{}, C0{}, C0{});
  }
  if (nsc > 1) {   // (a K range of one super-chunk ends with the ordinary pair above)
    chunk(kc1 - 2, C0{}, C0{}, C1{});
    chunk(kc1 - 1, C1{}, C0{}, C1{});
  }
  if constexpr ((FLAGS & F4_DBG_TIME) != 0) tm2 = __builtin_readcyclecounter();

  // ---- epilogue.  C/D layout with A = weights: column = lane & 31 = pixel of the wave's 4 x 8 tile, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // = output channel of the 32: a lane holds four CONSECUTIVE channels of its pixel per register group.  The four output pixels of a
  // low-resolution pixel are combinations of the lane's own four accumulators: no exchange between waves.  Stored straight from these
  // registers a dwordx4 store scatters 32-byte pieces over 32 pixels (measured: 26 000 - 49 000 cycles of epilogue per workgroup, as long
  // as the K = 128 layer's whole K loop), so every phase goes through a wave-private LDS tile [pixel 32][BN channels] (16-byte pieces
  // swizzled by the pixel: conflict free both ways) and leaves as 64 / PP pixels x 128 NCT contiguous bytes per store.
  __syncthreads();   // the tiles overlay the stages (every wave's fragment reads are done)
  // Round 6, the instruction count of this epilogue (21 000 - 41 000 cycles per workgroup, 15 - 23 % of a workgroup's life on the K = 128 ... 512 layers:
  // profiles/r06_fold4_bench.log): the combinations, the bias and leaky_relu in PACKED fp32 (same operations per element; max(v, slope v) for the
  // activation, like conv_wino2d_kernel), and every store address = a scalar tile corner + a scalar (pass, phase) offset + ONE per-thread offset instead of
  // a 64-bit expression of quarter-rate multiplies per store.
  constexpr int PP = 8 * NCT;                       // 16-byte pieces per pixel
  constexpr int PPP = 64 / PP;                      // pixels a wave stores per pass (8 / 4): a pass is a fixed row and column block of its 4 x 8 tile
  bf4* const xt = reinterpret_cast<bf4*>(smem) + wv * (32 * PP);
  auto pair_at = [](const f32x16& a, int i) -> f2 { return f2{a[i], a[i + 1]}; };
  auto add2 = [](f2 x, f2 y) -> f2 { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto mul2 = [](float k, f2 x) -> f2 { f2 r; const f2 kk = {k, k}; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(kk), "v"(x)); return r; };
  auto max1 = [](float x, float y) -> float { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  const bool rawsum = ksp > 1;   // split-K: raw sums, no bias, no activation, [split][output pixel][Cout]
  f2 bias2[NCT][4][2];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) { bias2[c][g][0] = f2{0.f, 0.f}; bias2[c][g][1] = f2{0.f, 0.f}; }
  if (!rawsum) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* const bp = p.bias + n0 + c * 32 + 8 * g + 4 * half;
        bias2[c][g][0] = f2{bp[0], bp[1]}; bias2[c][g][1] = f2{bp[2], bp[3]};
      }
  }
  const float slope = (p.leaky != 0 && !rawsum) ? 0.2f : 1.f;
  const int wbase = l31 * PP;
  const int rpx = lane / PP, rpi = lane % PP;       // reader: pixel within a pass, piece
  const int ostr = rawsum ? p.Cout : p.ostride;
  float* obase = p.out;
  if (rawsum) { asm volatile(""); obase = p.part + (size_t)blockIdx.z * ((size_t)p.M * 4) * p.Cout; }   // (p.M = low-resolution pixels)
  // output pixel (2 y + py, 2 x + px): corner of the tile's 8 x 64 output pixels (scalar), the thread's column inside a pass (one 24-bit product)
  const int cpix = (img * 2 * aH + 2 * y0) * (2 * aW) + 2 * x0;
  float* const corner = obase + ((long long)cpix * ostr + n0);
  const int toff = (int)__umul24((unsigned)(2 * (8 * wv + rpx)), (unsigned)ostr) + 4 * rpi;
  const int xcol = x0 + 8 * wv + rpx;               // the thread's low-resolution column in a pass whose column block starts at 0
  const int w2o = 2 * aW * ostr;                    // one output row (8 output rows x 2 W x the pixel pitch < 2^31: the launcher)
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (the two channel pairs of a piece step by step, alternating: hipcc puts an s_nop between two DEPENDENT inline-asm vector instructions that follow
        // each other - 135 of them in this epilogue when each pair's chain was written out on its own)
        bf4 v;
        f2 u[2], su[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const f2 g0 = pair_at(acc[0][c], 4 * g + 2 * e2);
          u[e2] = ph == 0 ? g0 : sub2(g0, pair_at(acc[ph == 2 ? 2 : 1][c], 4 * g + 2 * e2));   // G0 | G0 - G1 | G0 - G2 | G0 - G1
        }
        if (ph == 3) {
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) u[e2] = sub2(u[e2], pair_at(acc[2][c], 4 * g + 2 * e2));
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) u[e2] = add2(u[e2], pair_at(acc[3][c], 4 * g + 2 * e2));   // ((G0 - G1) - G2) + G3
        }
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) u[e2] = add2(u[e2], bias2[c][g][e2]);
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) su[e2] = mul2(slope, u[e2]);
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) { v[2 * e2] = max1(u[e2][0], su[e2][0]); v[2 * e2 + 1] = max1(u[e2][1], su[e2][1]); }
        xt[wbase + ((c * 8 + g * 2 + half) ^ (l31 & (PP - 1)))] = v;
      }
#pragma unroll
    for (int it = 0; it < PP / 2; ++it) {
      const int prow = (it * PPP) >> 3, pcol = (it * PPP) & 7;   // the pass's row / first column of the wave's 4 x 8 tile (compile-time)
      const int px = it * PPP + rpx;                             // pixel of the tile
      const bf4 v = xt[px * PP + (rpi ^ (px & (PP - 1)))];
      if (y0 + prow < aH) {                                      // (uniform)
        float* const rowp = corner + ((2 * prow + (ph >> 1)) * w2o + (2 * pcol + (ph & 1)) * ostr);
        if (xcol + pcol < aW) *reinterpret_cast<bf4*>(rowp + toff) = v;
      }
    }
  }
  if constexpr ((FLAGS & F4_DBG_TIME) != 0) {
    const unsigned long long tm3 = __builtin_readcyclecounter();
    if (t == 0) {
      unsigned long long* o8 = reinterpret_cast<unsigned long long*>(p.part) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
      o8[0] = tm0; o8[1] = tm1; o8[2] = tm2; o8[3] = tm3; o8[7] = tmW;
      o8[8] = rt0; o8[9] = __builtin_amdgcn_s_memrealtime();
      o8[10] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
  }
}

template <int NCT, int FLAGS>
hipError_t conv_fold4_launch(const ConvParams& p, hipStream_t s) {
  constexpr int BN = 32 * NCT;
  const size_t lds = (size_t)3 * 12 * 1024;
  if (p.ksize != 2 || p.fold != 3 || p.nseg != 1 || p.Ctot % 16 || p.Cout % BN || p.pool_out || p.pw_out) return hipErrorInvalidValue;
  if (p.ksplit > 1 && (!p.part || (reinterpret_cast<uintptr_t>(p.part) & 15) || p.ksplit > p.Ctot / 16)) return hipErrorInvalidValue;
  if (p.seg[0].C != p.Ctot || p.seg[0].stride % 4 || p.seg[0].up || (reinterpret_cast<uintptr_t>(p.seg[0].ptr) & 15)) return hipErrorInvalidValue;
  if (p.ostride % 4 || (reinterpret_cast<uintptr_t>(p.out) & 15)) return hipErrorInvalidValue;   // dwordx4 stores
  if (p.W <= 0 || p.H <= 0 || p.ostride >= (1 << 22) || p.Cout >= (1 << 22) ||
      (long long)16 * p.W * (p.ostride > p.Cout ? p.ostride : p.Cout) >= (1ll << 31) ||                 // eight output rows in 32-bit offsets, 24-bit column product
      ((long long)std::max(p.NB, p.seg[0].bmod) + 1) * 4 * p.H * p.W >= (1ll << 31)) return hipErrorInvalidValue;   // pixel indices (input and output) in 32 bits
  const int ntx = (p.W + 31) / 32, nty = (p.H + 3) / 4;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
  ConvParams q = p;   // + the reciprocals of the workgroup decomposition (conv_wino2d_launch's): ceil(2^32 / d), exact for x d < 2^32; 0 = the divisor is 1
  bool exact = true;
  auto magic = [&exact](unsigned long long d, unsigned long long xmax) -> unsigned {
    if (d <= 1) return 0u;
    if (xmax * d >= (1ull << 32)) { exact = false; return 0u; }
    return (unsigned)(((1ull << 32) + d - 1) / d);
  };
  q.mg_nby = magic(grid.y, (unsigned long long)grid.x * grid.y);
  q.mg_tpi = magic((unsigned long long)ntx * nty, (unsigned long long)grid.x + 1);
  q.mg_ntx = magic(ntx, (unsigned long long)ntx * nty);
  q.tl_ntx = ntx; q.tl_tpi = ntx * nty;
  if (!exact) return hipErrorInvalidValue;
  hipLaunchKernelGGL((conv_fold4_kernel<NCT, FLAGS>), grid, dim3(256), lds, s, q);
  return hipGetLastError();
}
