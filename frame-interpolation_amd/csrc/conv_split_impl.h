// conv_split_impl.h -- OPT-IN precision modes "bf16x6" / "bf16x3" (film_set_option("precision", 1 / 2); the default
// engine path is the fp32 MFMA of conv_wino_impl.h / conv_halo_impl.h / conv_buf_impl.h): the halo-staged 3x3
// convolution with every fp32 operand split into bf16 pieces x = hi + mid + lo (round-to-nearest-even pieces,
// 8 + 8 + 8 significant bits, the sum is EXACT - see conv_split4) and the product formed on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate; bf16 x bf16 products are exact in fp32) from
//   NPROD = 6:  hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)     dropped: mid*lo, lo*mid, lo*lo <= 2^-25 |ab|
//   NPROD = 3:  hi*hi + (hi*mid + mid*hi)                                  dropped: mid*mid <= 2^-16 |ab| plus two residues <= 2^-17 |ab|
//                                                                         (worst case 2^-15; measured rms 4.4e-6, mean 1e-9: zero mean)
// Six (three) bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16 K: 2.67x (5.3x) the matrix
// rate.  bf16x6 is fp32-level accurate (max |delta| 2.8e-5 on outputs of O(10) against the fp32 kernels,
// tools/retired/conv_bench.hip: the same as between two fp32 kernels that sum K in a different order); bf16x3 stages only
// the hi and mid planes (2/3 of the LDS and of the split VALU work).  Measured speed of bf16x6: 1.45-1.65x - at
// 68 % MFMA-pipe occupancy the bf16 matrix pipe is power limited (clock 1.9 GHz), like every dense bf16 GEMM on
// this part.
//
//   * activations stay fp32 in HBM; the split happens once per staged element on the way into LDS (the halo
//     staging amortises it over the nine taps); weights are split once on the host side:
//     [Cout][chunk][tap][plane][16] bf16 (bf16x3 reads planes 0 and 1 of the same copy).
//   * LDS image, plane major: [plane][row][16 bf16 = 32 B]; the two 16-byte K-halves of row r are swapped when bit 3 of
//     r is set.  Reads: the 16 rows of a ds_read_b128 lane group then hit 16 distinct bank quads (rows r and r+8 /
//     r+24 would collide otherwise); writes: 4 rows x 32 contiguous bytes per 16-lane ds_write_b64 group.  (The first
//     version used 112-byte [row][plane] rows: conflict-free reads, but 20-29 % conflict cycles from the stores.)
#pragma once
#include "conv_buf_impl.h"

typedef __bf16 sbf8 __attribute__((ext_vector_type(8)));
typedef unsigned su4 __attribute__((ext_vector_type(4)));
typedef unsigned su2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ su4 conv_buf_load_u4(conv_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(su4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

typedef __bf16 sbf2 __attribute__((ext_vector_type(2)));
typedef float sf2 __attribute__((ext_vector_type(2)));

// Round-to-nearest-even split of four floats into bf16 planes (each plane: 4 bf16 = 2 dwords; channel i in the low
// half of dword i/2): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  Both differences are exact in
// fp32 and the last one fits 8 significant bits, so hi + mid + lo == x EXACTLY (|x - hi| <= 2^-9 |x| is a multiple of
// ulp(x) -> 16 bits; |x - hi - mid| <= 2^-17 |x| -> 8 bits).  Rounding (v_cvt_pk_bf16_f32) instead of truncation keeps
// the pieces' signs uncorrelated with x, which is what makes the 2-plane "bf16x3" mode unbiased.
// x - bf16 piece, for both halves of a packed pair: v_dot2c_f32_bf16 with the constant pairs (-1, 0) / (0, -1) reads the
// bf16 halves in place (one instruction per element instead of shift / mask + subtract; x - hi is exact in fp32,
// so the fused dot product returns exactly the same value)
__device__ __forceinline__ sf2 conv_sub_bf16_pair(sf2 v, sbf2 p) {
  // The constants go through opaque s_mov's: as literals hipcc folds the pair (-1, 0) = 0x0000BF80 into the INLINE
  // constant "-1.0", which the instruction reads as 0xBF800000 = (0, -1) - both halves then subtract the high piece.
  unsigned k0, k1;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(k0));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(k1));
  const sbf2 c0 = __builtin_bit_cast(sbf2, k0), c1 = __builtin_bit_cast(sbf2, k1);
  sf2 r;
  r.x = __builtin_amdgcn_fdot2_f32_bf16(p, c0, v.x, false);
  r.y = __builtin_amdgcn_fdot2_f32_bf16(p, c1, v.y, false);
  return r;
}

template <bool WITH_LO>
__device__ __forceinline__ void conv_split4(bf4 x, su2& hi, su2& mid, su2& lo) {
  unsigned hp[2], mp[2], lp[2] = {0u, 0u};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const sf2 v = {x[2 * j], x[2 * j + 1]};
    const sbf2 h = __builtin_convertvector(v, sbf2);
    const sf2 r = conv_sub_bf16_pair(v, h);
    const sbf2 m = __builtin_convertvector(r, sbf2);
    hp[j] = __builtin_bit_cast(unsigned, h);
    mp[j] = __builtin_bit_cast(unsigned, m);
    if constexpr (WITH_LO) lp[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(conv_sub_bf16_pair(r, m), sbf2));
  }
  hi.x = hp[0]; hi.y = hp[1];
  mid.x = mp[0]; mid.y = mp[1];
  lo.x = lp[0]; lo.y = lp[1];
}

template <int TH, int BN, int WGM, int WGN, int NPROD, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_halo_split_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int TM = TH / WGM;
  constexpr int WTN = BN / WGN, TN = WTN / 32;
  constexpr int HR = TH + 2, HC = 34;
  constexpr int A_PLANE = HR * HC * 32;          // bytes: one bf16 plane of the halo chunk
  constexpr int B_PLANE = BN * 32;
  constexpr int NPL = NPROD > 3 ? 3 : 2;         // planes staged: bf16x3 (hi*hi + hi*mid + mid*hi) never reads lo
  constexpr int A_STAGE = NPL * A_PLANE;         // bytes
  constexpr int B_STAGE = NPL * B_PLANE;
  constexpr int HF4 = HR * HC * 4;
  constexpr int AH = (HF4 + NT - 1) / NT;
  constexpr int BU = BN * 2 * NPL;               // 16-byte units of one weight step
  constexpr int BLD = (BU + NT - 1) / NT;
  static_assert(TH % WGM == 0 && TM >= 1 && TN >= 1, "bad tile");
  static_assert(NPROD == 6 || NPROD == 3, "bf16x6 or bf16x3");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_b[];  // [A0][A1][B x3]
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
  const int ntx = (p.W + 31) >> 5, nty = (p.H + TH - 1) / TH;
  const int img = bx / (ntx * nty);
  const int trem = bx - img * (ntx * nty);
  const int y0 = (trem / ntx) * TH, x0 = (trem % ntx) * 32;
  const int n0 = by * BN;

  // ---- A staging ---------------------------------------------------------------------------------------
  int aiy[AH], aix[AH];
  bool ain[AH];
  int alds[AH];  // byte offset of plane 0 of this thread's 4 channels, -1: no slot
#pragma unroll
  for (int i = 0; i < AH; ++i) {
    const int f = t + NT * i;
    const bool slot = f < HF4;
    const int r = slot ? (f >> 2) : 0, ch = f & 3;
    const int hy = r / HC, hx = r - hy * HC;
    aiy[i] = y0 - 1 + hy; aix[i] = x0 - 1 + hx;
    ain[i] = slot && aiy[i] >= 0 && aiy[i] < p.H && aix[i] >= 0 && aix[i] < p.W;
    alds[i] = slot ? r * 32 + ((((ch >> 1) ^ ((r >> 3) & 1)) << 4) | ((ch & 1) << 3)) : -1;
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

  // ---- B staging ---------------------------------------------------------------------------------------
  const int nkc = p.Ctot / 16;
  const int nsteps = nkc * 9;
  const conv_rsrc_t brsrc = conv_make_rsrc(p.w);
  unsigned boff[BLD];
  int blds[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int u = t + NT * i;   // (plane, row, K-half): 8 consecutive lanes store 4 rows x 32 contiguous bytes
    const bool slot = u < BU;
    const int kb = u & 1, row = slot ? (u >> 1) % BN : 0, pl = slot ? (u >> 1) / BN : 0;
    boff[i] = (unsigned)((size_t)(n0 + row) * nsteps * 96 + pl * 32 + kb * 16);
    blds[i] = slot ? pl * B_PLANE + row * 32 + ((kb ^ ((row >> 3) & 1)) << 4) : -1;
  }

  bf4 areg[AH];
  su4 breg[BLD];
  auto load_a = [&]() {
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int i = 0; i < AH; ++i) areg[i] = conv_buf_load(arsrc, aoff[i], so);
  };
  auto next_chunk = [&](int kc_next) {
    if (kc_next >= nkc) {
#pragma unroll
      for (int i = 0; i < AH; ++i) aoff[i] = OOB;
      return;
    }
    c0 += 16;
    if (c0 >= segC) { c0 = 0; ++sg; setup_seg(); }
  };
  auto store_a = [&](int stage) {
    unsigned char* As = smem_b + stage * A_STAGE;
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      if (NT * (i + 1) <= HF4 || alds[i] >= 0) {
        su2 hi, mid, lo;
        conv_split4<NPL == 3>(areg[i], hi, mid, lo);
        *reinterpret_cast<su2*>(As + alds[i]) = hi;
        *reinterpret_cast<su2*>(As + alds[i] + A_PLANE) = mid;
        if constexpr (NPL == 3) *reinterpret_cast<su2*>(As + alds[i] + 2 * A_PLANE) = lo;
      }
    }
  };
  auto load_b = [&](int s) {
    const unsigned so = (unsigned)(s < nsteps ? s : nsteps - 1) * 96u;
#pragma unroll
    for (int i = 0; i < BLD; ++i) breg[i] = conv_buf_load_u4(brsrc, boff[i], so);
  };
  auto store_b = [&](int stage) {
    unsigned char* Bs = Bsm + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      if (NT * (i + 1) <= BU || blds[i] >= 0) *reinterpret_cast<su4*>(Bs + blds[i]) = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses in 16-byte units (su4 array index -> ds_read_b128).  A: row (wy+mt+dy)*34 + dx + l31, one
  // address per (row of the wave, tap) because the K-half swap depends on bit 3 of the row.
  const su4* const smem16 = reinterpret_cast<const su4*>(smem_b);
  const int wy = wm * TM;
  int a_ad[TM + 2][3];
#pragma unroll
  for (int ry = 0; ry < TM + 2; ++ry)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int r = (wy + ry) * HC + dx + l31;
      a_ad[ry][dx] = r * 2 + (half ^ ((r >> 3) & 1));
    }
  int b_ad[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int r = wn * WTN + nt * 32 + l31;
    b_ad[nt] = (2 * A_STAGE) / 16 + r * 2 + (half ^ ((r >> 3) & 1));
  }
  int a_stage_u = 0;  // 16-byte offset of the A stage being read

  auto compute = [&](auto tap_c) {
    constexpr int TAP = decltype(tap_c)::value;
    constexpr int DY = TAP / 3, DX = TAP % 3;
    sbf8 a[NPL][TM], b[NPL][TN];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
        a[pl][mt] = __builtin_bit_cast(sbf8, smem16[a_ad[mt + DY][DX] + a_stage_u + pl * (A_PLANE / 16)]);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
        b[pl][nt] = __builtin_bit_cast(sbf8, smem16[b_ad[nt] + (TAP % 3) * (B_STAGE / 16) + pl * (B_PLANE / 16)]);
    }
    // smallest partial products first
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int k = 6 - NPROD; k < 6; ++k)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[k]][mt], b[PB[k]][nt], acc[mt][nt], 0, 0, 0);
  };

  setup_seg();
  load_a();
  load_b(0);
  store_a(0);
  store_b(0);
  load_b(1);
  store_b(1);
  next_chunk(1);
  __syncthreads();
  int a_stage = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    const int s0 = kc * 9;
    auto step = [&](auto tap_c) {
      constexpr int TAP = decltype(tap_c)::value;
      load_b(s0 + TAP + 2);
      if constexpr (TAP == 0) load_a();
      __builtin_amdgcn_sched_barrier(0);
      compute(tap_c);
      __builtin_amdgcn_sched_barrier(0);
      store_b((TAP + 2) % 3);
      if constexpr (TAP == 8) store_a(a_stage ^ 1);
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
    a_stage ^= 1;
    a_stage_u = a_stage * (A_STAGE / 16);
  }

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

template <int TH, int BN, int WGM, int WGN, int NPROD, int FLAGS>
hipError_t conv_halo_split_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t npl = NPROD > 3 ? 3 : 2;
  constexpr size_t lds = 2 * npl * (size_t)(TH + 2) * 34 * 32 + 3 * npl * (size_t)BN * 32;
  auto kern = conv_halo_split_kernel<TH, BN, WGM, WGN, NPROD, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  const int ntx = (p.W + 31) / 32, nty = (p.H + TH - 1) / TH;
  dim3 grid((unsigned)(p.NB * ntx * nty), p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
