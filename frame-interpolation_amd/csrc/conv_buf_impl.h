// conv_buf_impl.h -- implicit-GEMM convolution, register-staged, with the leanest K loop we could build.
//
// Same mapping as tools/retired/conv_igemm_impl.h (M = output pixels, N = Cout, K walked in 16-channel steps over
// (tap, concat segment, chunk); fp32 v_mfma_f32_32x32x2_f32; bias + leaky_relu epilogue).  Measured on MI355X
// (tools/retired/conv_bench.hip + rocprofv3 PMC): every non-MFMA vector instruction issued inside the K loop costs
// matrix-pipe time that the co-resident waves do not win back, while the memory system is nowhere near a limit
// (gathers from one pixel run exactly as fast).  So this kernel spends no VALU instruction per K-step:
//
//   * buffer_load_dwordx4 with a per-lane 32-bit offset (fixed per tap) + a scalar offset (the channel chunk):
//     no 64-bit address arithmetic per step;
//   * 'same' zero padding and the ragged last M tile by the buffer bounds check: an out-of-image lane carries
//     offset 0xFFFFFFFF, the hardware returns zeros - no select instructions;
//   * per tap the offsets are  centre + (dy*W + dx)*stride  with a 9-bit in-image mask per staged row, computed
//     once per block (nearest-upsampled segments, fusion.py:133-134, recompute their offsets per tap instead);
//   * weights packed K-contiguous per output channel ([Cout][taps*Ctot]) so that the B tile is staged and read
//     exactly like the A tile: one ds_write_b128 per staged float4, one ds_read_b128 per fragment;
//   * LDS rows are 64 B, unpadded, chunk c of row r stored at chunk position c ^ ((r >> 2) & 3): conflict-free
//     for the ds_write_b128 (8-lane groups cover 128 contiguous bytes) and for the ds_read_b128 fragment reads;
//   * two register stages + two LDS stages: the loads of step s+2 are issued before the MFMAs of step s, the
//     registers of step s+1 go to LDS after them, one barrier per step.
#pragma once
#include <atomic>
#include <type_traits>
#include <utility>

#include "film_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float bf4 __attribute__((ext_vector_type(4)));
typedef int bi4 __attribute__((ext_vector_type(4)));

enum : int {
  CONV_B_XCD_M = 4,  // XCD-contiguous block mapping (each XCD walks a contiguous range of M tiles)
};

// raw buffer resource over [p, p + 4 GiB): stride 0, num_records = 0xFFFFFFFF bytes, gfx9 dword3 for raw
// 32-bit access.  A lane whose offset is 0xFFFFFFFF fails the bounds check and loads zeros.
typedef __amdgpu_buffer_rsrc_t conv_rsrc_t;
__device__ __forceinline__ conv_rsrc_t conv_make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function ON the current device and is idempotent: set on the first launch
// per (kernel instantiation, device).  The "already set" flags are atomics (relaxed is enough: a second thread that misses the flag
// only repeats the call) - several handles on several host threads may launch the same instantiation (include/film_hip.h).
struct ConvLdsAttrFlags { std::atomic<bool> set[64]; };
inline hipError_t conv_allow_dynamic_lds(const void* kern, ConvLdsAttrFlags& flags, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && flags.set[dev].load(std::memory_order_relaxed)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && tracked) flags.set[dev].store(true, std::memory_order_relaxed);
  return e;
}

__device__ __forceinline__ bf4 conv_buf_load(conv_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(bf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

template <int BM, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_buf_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPP = NT / 4;              // tile rows staged per pass (4 threads x float4 per 64-B row)
  constexpr int AR = BM / RPP;             // A rows per thread
  constexpr int BR = (BN + RPP - 1) / RPP; // B rows per thread (threads beyond the tile repeat a row)
  constexpr int STAGE = (BM + BN) * 16;    // floats per LDS stage
  static_assert(BM % RPP == 0 && TM >= 1 && TN >= 1 && RPP % 16 == 0, "bad tile");
  constexpr unsigned OOB = 0xFFFFFFFFu;

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // 2 stages

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
  const int m0 = bx * BM;
  const int n0 = by * BN;

  // ---- staging bookkeeping ---------------------------------------------------------------------
  const int srow = t >> 2;                 // row within a pass
  const int scol = (t & 3) * 4;            // logical channel offset of this thread's float4
  const int pcol = ((t & 3) ^ ((t >> 4) & 3)) * 4;  // swizzled position inside the LDS row (RPP % 16 == 0)
  const int ksz = p.ksize;
  const int pad = (ksz - 1) >> 1;
  // folded upsample + 2x2 (film_kernels.h): phase from the params (fold = 1) or from blockIdx.z (fold = 2)
  const int fpy = p.fold == 2 ? (int)(blockIdx.z >> 1) : p.py, fpx = p.fold == 2 ? (int)(blockIdx.z & 1) : p.px;
  const int ntaps = p.fold == 2 ? (fpy + 1) * (fpx + 1) : p.fold ? p.ftaps : ksz * ksz;
  auto tap_dy = [&](int tp) { return p.fold == 2 ? (fpx ? tp >> 1 : tp) : p.fold ? (int)p.tdy[tp] : tp / ksz - pad; };
  auto tap_dx = [&](int tp) { return p.fold == 2 ? (fpx ? tp & 1 : 0) : p.fold ? (int)p.tdx[tp] : tp % ksz - pad; };
  const float* const wbase = p.fold == 2 ? p.w + p.fold_woff[blockIdx.z] : p.w;
  const int ksplit = p.fold ? 1 : (p.ksplit > 1 ? p.ksplit : 1);
  const int split = ksplit > 1 ? (int)blockIdx.z : 0;
  const int HW = p.H * p.W;
  int ab[AR], ay[AR], ax[AR];
  unsigned amask[AR];                      // bit tap: the tap's source pixel is inside the image (and m < M)
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + srow + RPP * i;
    const bool valid = m < p.M;
    const int mm = valid ? m : 0;
    const int b = mm / HW;
    const int r = mm - b * HW;
    const int y = r / p.W;
    const int x = r - y * p.W;
    ab[i] = b; ay[i] = y; ax[i] = x;
    unsigned mask = 0;
    for (int tp = 0; tp < ntaps; ++tp) {
      const int yy = y + tap_dy(tp), xx = x + tap_dx(tp);
      if (valid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1u << tp;
    }
    amask[i] = mask;
  }
  const int Ktot = ntaps * p.Ctot;
  const conv_rsrc_t brsrc = conv_make_rsrc(wbase);
  unsigned boff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) boff[i] = (unsigned)(((n0 + (srow + RPP * i) % BN) * Ktot + scol) * 4);
  const bool bstore = (BN % RPP == 0) || srow < BN;  // BR == 1 and fewer rows than threads: upper threads idle

  // per (tap, segment): per-row byte offsets of the source pixel (OOB outside the image)
  int tap = 0, sg = 0, c0 = 0, segC = p.seg[0].C;
  conv_rsrc_t arsrc = conv_make_rsrc(p.seg[0].ptr);
  unsigned aoff[AR];
  unsigned acen[AR];  // centre-pixel offsets of the current segment (segments without upsampling)
  auto setup_seg = [&]() {
    const ConvSeg& s = p.seg[sg];
    arsrc = conv_make_rsrc(s.ptr);
    segC = s.C;
    if (!s.up) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int be = ab[i] + s.boff;
        if (s.bmod && be >= s.bmod) be -= s.bmod;
        acen[i] = (unsigned)((((size_t)be * p.H + ay[i]) * p.W + ax[i]) * s.stride + scol) * 4u;
      }
    }
  };
  auto setup_tap = [&]() {
    const ConvSeg& s = p.seg[sg];
    const int dy = tap_dy(tap), dx = tap_dx(tap);
    if (!s.up) {
      const unsigned delta = (unsigned)((dy * p.W + dx) * s.stride * 4);
#pragma unroll
      for (int i = 0; i < AR; ++i) aoff[i] = ((amask[i] >> tap) & 1u) ? acen[i] + delta : OOB;
    } else {
      const int Hs = p.H >> 1, Ws = p.W >> 1;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int yy = (ay[i] + dy) >> 1, xx = (ax[i] + dx) >> 1;
        int be = ab[i] + s.boff;
        if (s.bmod && be >= s.bmod) be -= s.bmod;
        const unsigned off = (unsigned)((((size_t)be * Hs + yy) * Ws + xx) * s.stride + scol) * 4u;
        aoff[i] = ((amask[i] >> tap) & 1u) ? off : OOB;
      }
    }
  };
  auto advance = [&]() {
    c0 += 16;
    if (c0 >= segC) {
      c0 = 0;
      if (++sg == p.nseg) { sg = 0; ++tap; }
      if (p.nseg > 1) setup_seg();
      setup_tap();
    }
  };

  struct Stage {
    bf4 a[AR];
    bf4 b[BR];
  };
  Stage sx, sy;
  const int nsteps_all = ntaps * (p.Ctot / 16);
  const int k_begin = (int)((long long)split * nsteps_all / ksplit);
  const int nsteps = (int)((long long)(split + 1) * nsteps_all / ksplit);   // end of this split's K range
  int kstep = k_begin;
  if (k_begin > 0) {   // seek (tap, segment, chunk) to step k_begin
    const int spt = p.Ctot / 16;
    tap = k_begin / spt;
    int rem = (k_begin - tap * spt) * 16;
    sg = 0;
    while (rem >= p.seg[sg].C) { rem -= p.seg[sg].C; ++sg; }
    c0 = rem;
  }
  // loads of K-step `kstep`; past the end: A from nowhere (zeros), B repeats the last step (finite x 0)
  auto load_global = [&](Stage& st) {
    const unsigned asoff = (unsigned)c0 * 4u;
#pragma unroll
    for (int i = 0; i < AR; ++i) st.a[i] = conv_buf_load(arsrc, aoff[i], asoff);
    const unsigned bsoff = (unsigned)(kstep < nsteps ? kstep : nsteps - 1) * 64u;
#pragma unroll
    for (int i = 0; i < BR; ++i) st.b[i] = conv_buf_load(brsrc, boff[i], bsoff);
    ++kstep;
    if (kstep < nsteps) advance();
    else {
#pragma unroll
      for (int i = 0; i < AR; ++i) aoff[i] = OOB;
    }
  };
  auto store_lds = [&](const Stage& st, int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * 16;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<bf4*>(As + (srow + RPP * i) * 16 + pcol) = st.a[i];
    if (bstore) {
#pragma unroll
      for (int i = 0; i < BR; ++i) *reinterpret_cast<bf4*>(Bs + ((srow + RPP * i) % BN) * 16 + pcol) = st.b[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads: row * 16 floats + swizzled chunk; chunk of (kq, half) = kq*2 + half
  const int sw = (l31 >> 2) & 3;
  const int ch0 = (half ^ sw) * 4, ch1 = ((2 + half) ^ sw) * 4;
  const int a_row = (wm * WTM + l31) * 16;
  const int b_row = (BM + wn * WTN + l31) * 16;
  auto compute = [&](int buf) {
    const float* S = smem + buf * STAGE;
    bf4 a[2][TM], b[2][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
      a[0][mt] = *reinterpret_cast<const bf4*>(S + a_row + mt * 512 + ch0);
      a[1][mt] = *reinterpret_cast<const bf4*>(S + a_row + mt * 512 + ch1);
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      b[0][nt] = *reinterpret_cast<const bf4*>(S + b_row + nt * 512 + ch0);
      b[1][nt] = *reinterpret_cast<const bf4*>(S + b_row + nt * 512 + ch1);
    }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[kq][nt][j], a[kq][mt][j], acc[mt][nt], 0, 0, 0);   // A = weights: C^T, the same sums
  };

  // ---- two-step prefetch pipeline (see tools/retired/conv_igemm_impl.h for the reasoning behind the shape of this loop) --
  setup_seg();
  setup_tap();
  load_global(sx);  // step 0
  load_global(sy);  // step 1
  store_lds(sx, 0);
  __syncthreads();
  for (int s = k_begin; s < nsteps; s += 2) {
    load_global(sx);  // step s+2
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    store_lds(sy, 1);  // step s+1
    __syncthreads();
    load_global(sy);  // step s+3
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    store_lds(sx, 0);  // step s+2
    __syncthreads();
  }

  // ---- epilogue: bias + leaky_relu.  The MFMA ran with A = weights, B = pixels: C/D column = lane & 31 = pixel row of the tile,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5) = output channel - a lane holds four CONSECUTIVE channels of its pixel per register group:
  // four dwordx4 stores per 32x32 tile instead of sixteen dword stores (round 4: the store count, not the bytes, was the cost).
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    const int m = m0 + wm * WTM + mt * 32 + l31;
    if (m >= p.M) continue;
    size_t opix = (size_t)m;
    if (p.fold) {  // low-resolution pixel (b, y, x) -> output pixel (b, 2y+py, 2x+px) of the 2H x 2W image
      const int bq = m / HW, rr = m - bq * HW, y = rr / p.W, x = rr - y * p.W;
      opix = ((size_t)bq * 2 * p.H + 2 * y + fpy) * (2 * p.W) + 2 * x + fpx;
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * WTN + nt * 32 + 8 * g + 4 * half;
        bf4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = acc[mt][nt][4 * g + c];
        if (ksplit > 1) {  // raw partial sums; bias + activation in conv_splitk_reduce_kernel
          *reinterpret_cast<bf4*>(p.part + ((size_t)split * p.M + m) * p.Cout + n) = v;
          continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float u = v[c] + p.bias[n + c];
          if (p.leaky) u = u > 0.f ? u : 0.2f * u;
          v[c] = u;
        }
        *reinterpret_cast<bf4*>(p.out + opix * p.ostride + n) = v;
      }
  }
}

template <int BM, int BN, int WGM, int WGN, int FLAGS>
hipError_t conv_buf_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 16 * sizeof(float);
  auto kern = conv_buf_kernel<BM, BN, WGM, WGN, FLAGS>;
  if constexpr (lds > 64 * 1024) {
    static ConvLdsAttrFlags attr_flags;   // one per kernel instantiation (this launcher is a template)
    if (const hipError_t e = conv_allow_dynamic_lds(reinterpret_cast<const void*>(kern), attr_flags, (int)lds); e != hipSuccess) return e;
  }
  if (p.ksplit > 1 && p.fold) return hipErrorInvalidValue;
  if (p.ksplit > 1 ? (p.Cout % 4 || (reinterpret_cast<uintptr_t>(p.part) & 15)) : (p.ostride % 4 || (reinterpret_cast<uintptr_t>(p.out) & 15))) return hipErrorInvalidValue;   // dwordx4 stores
  dim3 grid((p.M + BM - 1) / BM, p.Cout / BN, p.fold == 2 ? 4 : (p.ksplit > 1 ? p.ksplit : 1));
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, p);
  return hipGetLastError();
}
