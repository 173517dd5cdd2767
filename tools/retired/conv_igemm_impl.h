// conv_igemm_impl.h -- the implicit-GEMM convolution kernel template (see conv_igemm.hip for the overview).
// Round 1's general kernel.  What still uses it: the first-layer (3-channel, [48][Cout] weights) mode for Cout other than 32 / 64
// (conv_c3_kernel covers those two), and tools/retired/conv_bench.hip as the A/B baseline; every other layer runs conv_buf_kernel or one of
// the Winograd kernels.
//
// Template knobs (all compile time):
//   BM x BN        workgroup tile (pixels x output channels)
//   WGM x WGN      wave grid (4 or 8 waves); each wave owns (BM/WGM) x (BN/WGN) as 32x32 MFMA tiles
//   BKC            input channels per K-step (16 or 32); A rows are stored with stride BKC+4 floats
//   FLAGS          CONV_F_* bits below
#pragma once
#include "film_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int {
  CONV_F_SETPRIO = 1,    // raise wave priority around the MFMA cluster
  CONV_F_TAP_INNER = 2,  // K order (segment, chunk, tap) instead of (tap, segment, chunk): the 9 taps of one
                         // 16/32-channel chunk are consecutive steps, so the re-gathered pixels are still in L1/L2
  CONV_F_XCD_M = 4,      // XCD-aware block mapping: each XCD walks a contiguous range of M tiles, N fastest
  CONV_F_C3 = 8,         // first layer (feature_extractor.py:119-120): the single segment is the 3-channel image
                         // (stride 3); K = 12 tap slots x 4 (3 channels + one zero), i.e. 3 steps of 4 taps; the
                         // weights are packed [48][Cout] with row = tap*4 + channel (zero rows for the padding)
  // ablation switches for tools/retired/conv_bench.hip only (results are wrong on purpose):
  CONV_F_DBG_NOGLOBAL = 64,   // no global loads / LDS stores inside the K loop
  CONV_F_DBG_NOLDSREAD = 128, // no LDS fragment reads inside the K loop (operands from registers)
  CONV_F_DBG_NOBARRIER = 256, // no barrier inside the K loop
  CONV_F_PF1 = 512,
  CONV_F_DBG_SAMEPIX = 1024,
  CONV_F_DBG_NOLDSWRITE = 2048,  // global loads issued and waited for, but no LDS store (timing only)
  CONV_F_DBG_NOLOAD = 4096,      // LDS stores of stale registers, no global loads (timing only)  // every A row reads pixel 0 of its image (all gathers hit L1/L2)           // single-step prefetch (the first version of the pipeline; kept for A/B runs)
};

template <int BM, int BN, int WGM, int WGN, int BKC, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_igemm_kernel(ConvParams p) {
  constexpr int NT = WGM * WGN * 64;               // threads
  constexpr int AST = BKC + 4;                     // A row stride in LDS (floats): conflict-free ds_read_b128
  constexpr int WTM = BM / WGM, WTN = BN / WGN;    // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;      // 32x32 MFMA tiles per wave
  constexpr int A_TPR = BKC / 4;                   // threads (float4) per A row
  constexpr int A_RPP = NT / A_TPR;                // A rows staged per pass
  constexpr int AROWS = BM / A_RPP;                // passes = rows per thread
  constexpr int BF4 = BKC * BN / 4;                // float4 in a B tile
  constexpr int BLD = (BF4 + NT - 1) / NT;         // B float4 per thread
  constexpr int A_SZ = BM * AST, B_SZ = BKC * BN;
  constexpr bool TAP_INNER = (FLAGS & CONV_F_TAP_INNER) != 0;
  constexpr bool C3 = (FLAGS & CONV_F_C3) != 0;
  static_assert(!C3 || (BKC == 16 && !TAP_INNER), "C3 mode: 16-wide steps of 4 taps");
  static_assert(BM % A_RPP == 0 && TM >= 1 && TN >= 1 && AROWS >= 1, "bad tile");
  static_assert(BLD >= 1 && BLD <= 4, "B staging holds up to four float4 per thread");
  static_assert(BKC == 16 || BKC == 32, "BKC");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x (A tile + B tile)

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WGN, wn = wv % WGN;

  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr ((FLAGS & CONV_F_XCD_M) != 0) {
    // hardware round-robins consecutive workgroups over the 8 XCDs (private L2 each): give every XCD a
    // contiguous chunk of the (m-major, n-fastest) tile list so that the blocks sharing an A tile / a halo
    // share an L2.  Bijective for any grid size (cdna_hip_programming.md T1).
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
  const int m0 = bx * BM;
  const int n0 = by * BN;

  // ---- per-thread A staging rows -------------------------------------------------------------
  const int arow = t / A_TPR;
  const int acol = (t % A_TPR) * 4;
  int ab[AROWS], ay[AROWS], ax[AROWS];
  bool avalid[AROWS];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + arow + A_RPP * i;
    avalid[i] = m < p.M;
    const int mm = avalid[i] ? m : 0;
    const int b = mm / HW;
    const int r = mm - b * HW;
    const int y = r / p.W;
    ab[i] = b; ay[i] = y; ax[i] = r - y * p.W;
  }
  const int pad = (p.ksize - 1) >> 1;
  const int ntaps = p.ksize * p.ksize;

  // ---- per-thread B staging: unconditional loads (a predicated load makes hipcc branch around it and wait
  // vmcnt(0) right behind it); if the tile has fewer float4 than threads the upper threads re-read a valid
  // element and skip the LDS store.
  constexpr bool B_ALL = (BF4 % NT) == 0;
  const float* bbase[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int f = (t + NT * i) % BF4;
    const int krow = f / (BN / 4), n4 = f % (BN / 4);
    bbase[i] = p.w + (size_t)krow * p.Cout + n0 + n4 * 4;
  }
  const bool bstore = B_ALL || t < BF4;

  // ---- K iteration state ---------------------------------------------------------------------
  int tap = 0, sg = 0, c0 = 0, segoff = 0;
  const float* aptr[AROWS];
  bool ainb[AROWS];       // in-image mask of the staged rows for the CURRENT (tap, segment)

  auto setup_a = [&]() {  // (tap, segment) -> per-row source pointer (out-of-image rows point at a valid pixel)
    if constexpr (C3) {
      // `tap` counts K-steps here; this thread's float4 slot is tap slot 4*step + (t % 4)
      const int slot = tap * 4 + (t & 3);
      const int dy = slot / 3 - 1, dx = slot % 3 - 1;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int yy = ay[i] + dy, xx = ax[i] + dx;
        const bool inb = slot < 9 && avalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        const size_t pix = ((size_t)ab[i] * p.H + (inb ? yy : 0)) * p.W + (inb ? xx : 0);
        aptr[i] = p.seg[0].ptr + pix * 3;
        ainb[i] = inb;
      }
      return;
    }
    const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
    const ConvSeg& s = p.seg[sg];
    const int Hs = s.up ? (p.H >> 1) : p.H, Ws = s.up ? (p.W >> 1) : p.W;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int yy = ay[i] + dy, xx = ax[i] + dx;
      const bool inb = avalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      if (s.up) { yy >>= 1; xx >>= 1; }
      int be = ab[i] + s.boff;
      if (s.bmod && be >= s.bmod) be -= s.bmod;
      size_t pix = ((size_t)be * Hs + (inb ? yy : 0)) * Ws + (inb ? xx : 0);
      if constexpr ((FLAGS & CONV_F_DBG_SAMEPIX) != 0) pix = 0;
      aptr[i] = s.ptr + pix * s.stride + acol;
      ainb[i] = inb;
    }
  };

  // Two staging register sets (named scalars / per-set arrays with static indices: small arrays indexed at
  // run time are left in scratch by hipcc).  Set X holds the loads of one K-step, set Y those of the next.
  struct Stage {
    float4 a[AROWS];
    float4 b0, b1, b2, b3;
    bool inb[AROWS];
  };
  Stage sx, sy;
  auto load_global = [&](Stage& st, bool real) {  // real = false: padding step, its A rows are stored as zeros
    if constexpr ((FLAGS & CONV_F_DBG_NOLOAD) != 0) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) { asm volatile("" : "+v"(st.a[i].x), "+v"(st.a[i].y), "+v"(st.a[i].z), "+v"(st.a[i].w)); st.inb[i] = ainb[i] && real; }
      asm volatile("" : "+v"(st.b0.x), "+v"(st.b0.y), "+v"(st.b0.z), "+v"(st.b0.w));
      return;
    }
    if constexpr (C3) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) st.a[i] = make_float4(aptr[i][0], aptr[i][1], aptr[i][2], 0.f);
    } else {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) st.a[i] = *reinterpret_cast<const float4*>(aptr[i] + c0);
    }
    const size_t koff = C3 ? (size_t)tap * 16 * p.Cout
                           : (size_t)(tap * p.Ctot + segoff + c0) * p.Cout;  // weight row of this K-step
    st.b0 = *reinterpret_cast<const float4*>(bbase[0] + koff);
    if constexpr (BLD > 1) st.b1 = *reinterpret_cast<const float4*>(bbase[BLD > 1 ? 1 : 0] + koff);
    if constexpr (BLD > 2) st.b2 = *reinterpret_cast<const float4*>(bbase[BLD > 2 ? 2 : 0] + koff);
    if constexpr (BLD > 3) st.b3 = *reinterpret_cast<const float4*>(bbase[BLD > 3 ? 3 : 0] + koff);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) st.inb[i] = ainb[i] && real;
  };
  auto store_lds = [&](const Stage& st, int buf) {
    if constexpr ((FLAGS & CONV_F_DBG_NOLDSWRITE) != 0) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) asm volatile("" ::"v"(st.a[i].x), "v"(st.a[i].y), "v"(st.a[i].z), "v"(st.a[i].w));
      asm volatile("" ::"v"(st.b0.x), "v"(st.b0.w));
      if constexpr (BLD > 1) asm volatile("" ::"v"(st.b1.x), "v"(st.b1.w));
      if constexpr (BLD > 2) asm volatile("" ::"v"(st.b2.x), "v"(st.b2.w));
      if constexpr (BLD > 3) asm volatile("" ::"v"(st.b3.x), "v"(st.b3.w));
      return;
    }
    float* As = smem + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      float4 v = st.a[i];
      v.x = st.inb[i] ? v.x : 0.f; v.y = st.inb[i] ? v.y : 0.f;
      v.z = st.inb[i] ? v.z : 0.f; v.w = st.inb[i] ? v.w : 0.f;
      *reinterpret_cast<float4*>(As + (arow + A_RPP * i) * AST + acol) = v;
    }
    if (bstore) {
      *reinterpret_cast<float4*>(Bs + (t % BF4) * 4) = st.b0;
      if constexpr (BLD > 1) *reinterpret_cast<float4*>(Bs + (t + NT) * 4) = st.b1;
      if constexpr (BLD > 2) *reinterpret_cast<float4*>(Bs + (t + 2 * NT) * 4) = st.b2;
      if constexpr (BLD > 3) *reinterpret_cast<float4*>(Bs + (t + 3 * NT) * 4) = st.b3;
    }
  };
  auto advance = [&]() {
    if constexpr (C3) {
      ++tap;
      setup_a();
    } else if constexpr (TAP_INNER) {
      if (++tap == ntaps) {
        tap = 0;
        c0 += BKC;
        if (c0 >= p.seg[sg].C) { c0 = 0; segoff += p.seg[sg].C; ++sg; }
      }
      setup_a();
    } else {
      c0 += BKC;
      if (c0 >= p.seg[sg].C) {
        c0 = 0;
        segoff += p.seg[sg].C;
        if (++sg == p.nseg) { sg = 0; segoff = 0; ++tap; }
        setup_a();
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = C3 ? 3 : ntaps * (p.Ctot / BKC);

  // MFMAs of one K-step from LDS buffer `buf`
  auto compute = [&](int buf) {
    const float* As = smem + buf * (A_SZ + B_SZ);
    const float* Bs = As + A_SZ;
#pragma unroll
    for (int kh = 0; kh < BKC / 16; ++kh) {
      // fragment reads of 16 K-values first (2 x ds_read_b128 per M tile, 8 x ds_read_b32 per N tile), then the
      // 8*TM*TN MFMAs.  A lane's float4 holds 4 K-values: lanes 0-31 take channels {0..3}, lanes 32-63 {4..7} of
      // each 8-channel group, and B is read with the same permutation (a permutation of K only reorders the sum).
      float4 a[2][TM];
      float b[2][4][TN];
      if constexpr ((FLAGS & CONV_F_DBG_NOLDSREAD) != 0) {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) { a[kq][mt] = sx.a[0]; asm volatile("" : "+v"(a[kq][mt].x), "+v"(a[kq][mt].y), "+v"(a[kq][mt].z), "+v"(a[kq][mt].w)); }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) { b[kq][j][nt] = sx.b0.x; asm volatile("" : "+v"(b[kq][j][nt])); }
        }
      } else {
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
          for (int mt = 0; mt < TM; ++mt)
            a[kq][mt] = *reinterpret_cast<const float4*>(As + (wm * WTM + mt * 32 + l31) * AST + kh * 16 + kq * 8 + half * 4);
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
              b[kq][j][nt] = Bs[(kh * 16 + kq * 8 + half * 4 + j) * BN + wn * WTN + nt * 32 + l31];
      }
      if constexpr ((FLAGS & CONV_F_SETPRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mt = 0; mt < TM; ++mt) {
            const float av = j == 0 ? a[kq][mt].x : j == 1 ? a[kq][mt].y : j == 2 ? a[kq][mt].z : a[kq][mt].w;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kq][j][nt], acc[mt][nt], 0, 0, 0);
          }
        }
      }
      if constexpr ((FLAGS & CONV_F_SETPRIO) != 0) __builtin_amdgcn_s_setprio(0);
    }
  };
  constexpr bool NOGLOBAL = (FLAGS & CONV_F_DBG_NOGLOBAL) != 0;
  auto sync = [&]() { if constexpr ((FLAGS & CONV_F_DBG_NOBARRIER) == 0) __syncthreads(); };

  if constexpr ((FLAGS & CONV_F_PF1) != 0) {
    // ---- one-step prefetch: loads of step s+1 are issued before the MFMAs of step s -----------------
    setup_a();
    load_global(sx, true);
    store_lds(sx, 0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool more = s + 1 < nsteps;
      if constexpr (!NOGLOBAL) { if (more) { advance(); load_global(sx, true); } }
      compute(cur);
      if constexpr (!NOGLOBAL) { if (more) store_lds(sx, cur ^ 1); }
      sync();
      if constexpr (!NOGLOBAL) cur ^= 1;
    }
  } else {
    // ---- two-step prefetch ----------------------------------------------------------------------------
    // While step s computes from LDS buffer (s & 1), the registers of one set hold step s+1 (stored to the
    // other LDS buffer after the MFMAs) and the loads of step s+2 are in flight into the other set: the
    // staging loads get two K-steps (~2 x 2048 MFMA cycles per wave) to land instead of one.  Loads are
    // issued unconditionally every half-iteration (past the end they re-read the last step's addresses) so
    // that the compiler's counted vmcnt waits stay exact: a conditional load would merge to vmcnt(0).
    // The loop body is two K-steps with no exit in the middle (a mid-loop break makes hipcc keep two copies of
    // the accumulators: 128 AGPRs instead of 64); an odd step count is padded with one step whose A rows are
    // zeros, which adds exact zeros to every accumulator.
    setup_a();
    load_global(sx, true);                         // step 0
    if (nsteps > 1) advance();
    load_global(sy, nsteps > 1);                   // step 1 (or the padding step)
    store_lds(sx, 0);
    __syncthreads();
    // sched_barrier(0) pins the three phases of a half-iteration in program order: without it hipcc sinks the
    // loads below the MFMAs (to reuse the fragment registers) and hoists the zero-fill selects + their vmcnt
    // wait to the top of the next half, which puts the whole memory latency back on the critical path.
    for (int s = 0; s < nsteps; s += 2) {
      if constexpr (!NOGLOBAL) { if (s + 2 < nsteps) advance(); load_global(sx, s + 2 < nsteps); }   // step s+2
      __builtin_amdgcn_sched_barrier(0);
      compute(0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!NOGLOBAL) store_lds(sy, 1);                                                    // step s+1
      sync();
      if constexpr (!NOGLOBAL) { if (s + 3 < nsteps) advance(); load_global(sy, s + 3 < nsteps); }   // step s+3
      __builtin_amdgcn_sched_barrier(0);
      compute(NOGLOBAL ? 0 : 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!NOGLOBAL) store_lds(sx, 0);                                                    // step s+2
      sync();
    }
  }

  // ---- epilogue: bias + leaky_relu, 128-B row stores ---------------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int n = n0 + wn * WTN + nt * 32 + l31;
    const float bv = p.bias[n];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + wm * WTM + mt * 32 + row;
        if (m < p.M) {
          float v = acc[mt][nt][r] + bv;
          if (p.leaky) v = v > 0.f ? v : 0.2f * v;
          p.out[(size_t)m * p.ostride + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int BKC, int FLAGS>
hipError_t conv_igemm_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(BM * (BKC + 4) + BKC * BN) * sizeof(float);
  auto kern = conv_igemm_kernel<BM, BN, WGM, WGN, BKC, FLAGS>;
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
  dim3 grid((p.M + BM - 1) / BM, p.Cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
