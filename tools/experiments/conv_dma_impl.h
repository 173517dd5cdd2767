// conv_dma_impl.h -- EXPERIMENT (tools/conv_bench.hip only, not part of libfilm_hip.so): implicit-GEMM convolution
// with LDS-DMA operand staging.  Measured slower than register staging on MI355X (MFMA pipe busy 72 % vs 84 % /
// 93 %): every global_load_lds instruction costs ~200 cycles of matrix-pipe time on its SIMD, independent of the
// addresses it touches.  Kept as the record of that measurement.
//
// Same mapping as conv_igemm_impl.h (M = output pixels, N = Cout, K walked in 16-channel steps over
// (tap, concat segment, chunk), fp32 v_mfma_f32_32x32x2_f32, bias + leaky_relu epilogue), different pipeline:
//
//   * both operands go global -> LDS with `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR
//     staging, no ds_write): the A tile [BM pixels][16 channels] gathered from NHWC, the B tile
//     [BN output channels][16 k] from weights packed K-contiguous per output channel ([Cout][taps*Ctot]).
//   * zero padding of the 'same' convolution: a lane whose pixel is outside the image reads 16 bytes of a
//     zero page instead (the DMA source address is per lane).
//   * the LDS image of a DMA is lane-linear (wave base + 16*lane), so rows are 64 B with no padding; bank
//     conflicts of the ds_read_b128 fragment reads are avoided by swizzling on the SOURCE side: the 16-byte
//     chunk c of tile row r is stored at chunk position c ^ ((r >> 2) & 3).
//   * three LDS stages: while step s is computed, step s+1 has landed or is landing and the DMAs of step s+2
//     are issued; one raw s_barrier per step behind a counted `s_waitcnt vmcnt(N)` that leaves the newest
//     stage in flight across the barrier.
//   * the K loop is unrolled by the 3 stages (static LDS offsets); the step count is padded to a multiple of 3
//     with steps whose A rows all come from the zero page (adds exact zeros).
#pragma once
#include <type_traits>
#include <utility>

#include "../../frame-interpolation_amd/csrc/film_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int {
  CONV_D_XCD_M = 4,  // XCD-contiguous block mapping (same meaning as CONV_F_XCD_M)
  CONV_D_DBG_NODMA = 64,      // ablations (wrong results on purpose): no DMA issue inside the K loop
  CONV_D_DBG_NOVMWAIT = 128,  //   no vmcnt wait inside the K loop
  CONV_D_DBG_NOBARRIER = 256, //   no barrier inside the K loop
  CONV_D_DBG_SAMEB = 2048,    // tools/conv_bench.hip only: B rows of one DMA instruction are contiguous (1 KiB block)
  CONV_D_DBG_SAMEPIX = 1024,  // tools/conv_bench.hip only: every A row reads pixel 0 (all gathers hit L1/L2)
};

template <int N>
__device__ __forceinline__ void conv_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void conv_wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void conv_barrier() { asm volatile("s_barrier" ::: "memory"); }

typedef float v4f __attribute__((ext_vector_type(4)));

// ds_read_b128 at LDS byte address addr + OFF (OFF is folded into the 16-bit offset field where it fits)
template <int OFF>
__device__ __forceinline__ v4f conv_lds_read128(unsigned addr) {
  v4f r;
  constexpr unsigned hi = (unsigned)OFF & ~0xFFFFu, lo = (unsigned)OFF & 0xFFFFu;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr + hi), "n"(lo));
  return r;
}
// N fragments at addr + BASE + i * 2048 (32 tile rows apart)
template <int BASE, int N, int... I>
__device__ __forceinline__ void conv_lds_read_frags(v4f (&f)[N], unsigned addr, std::integer_sequence<int, I...>) {
  ((f[I] = conv_lds_read128<BASE + I * 2048>(addr)), ...);
}

__device__ __forceinline__ void conv_glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// zero page: DMA source of the out-of-image pixels
__device__ __attribute__((aligned(64))) float conv_dma_zero_page[16];

template <int BM, int BN, int WGM, int WGN, int FLAGS>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_dma_kernel(ConvParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM / 16 / NW;          // A DMA instructions per wave per stage (16 rows each)
  constexpr int BQ = BN / 16;               // B DMA instructions per stage
  constexpr int BI = (BQ + NW - 1) / NW;    // per wave (surplus waves repeat an instruction: same bytes, same place)
  constexpr int NI = AI + BI;               // DMA instructions per wave per stage
  constexpr int STAGE = (BM + BN) * 16;     // floats per stage
  static_assert(BM % (16 * NW) == 0 && BN % 16 == 0 && TM >= 1 && TN >= 1, "bad tile");

  extern __shared__ __attribute__((aligned(1024))) float smem[];  // 3 stages

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WGN, wn = wv % WGN;

  int bx = blockIdx.x, by = blockIdx.y;
  if constexpr ((FLAGS & CONV_D_XCD_M) != 0) {
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

  // ---- DMA source bookkeeping ------------------------------------------------------------------
  // lane -> (row within the 16-row instruction, physical chunk); the logical (channel) chunk it fetches is
  // swizzled with bits 2..3 of the row so that the fragment reads below are conflict free.
  const int drow = lane >> 2;
  const int dcol = ((lane & 3) ^ ((lane >> 4) & 3)) * 4;  // first channel (float index) of the fetched chunk
  int ab[AI], ay[AI], ax[AI];
  bool avalid[AI];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + (wv + NW * i) * 16 + drow;
    avalid[i] = m < p.M;
    const int mm = avalid[i] ? m : 0;
    const int b = mm / HW;
    const int r = mm - b * HW;
    const int y = r / p.W;
    ab[i] = b; ay[i] = y; ax[i] = r - y * p.W;
  }
  const int pad = (p.ksize - 1) >> 1;
  const int ntaps = p.ksize * p.ksize;
  const int Ktot = ntaps * p.Ctot;
  const float* bptr[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int qb = (wv + NW * i) % BQ;
    bptr[i] = p.w + (size_t)(n0 + qb * 16 + drow) * Ktot + dcol;
    if constexpr ((FLAGS & CONV_D_DBG_SAMEB) != 0) bptr[i] = p.w + (size_t)(n0 + qb * 16) * Ktot + drow * 16 + dcol;
  }
  const float* const zsrc = conv_dma_zero_page + dcol;

  int tap = 0, sg = 0, c0 = 0, segC = p.seg[0].C;
  const float* aptr[AI];
  bool ainb[AI];
  auto setup_a = [&]() {
    const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
    const ConvSeg& s = p.seg[sg];
    const int Hs = s.up ? (p.H >> 1) : p.H, Ws = s.up ? (p.W >> 1) : p.W;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int yy = ay[i] + dy, xx = ax[i] + dx;
      const bool inb = avalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      if (s.up) { yy >>= 1; xx >>= 1; }
      int be = ab[i] + s.boff;
      if (s.bmod && be >= s.bmod) be -= s.bmod;
      size_t pix = ((size_t)be * Hs + (inb ? yy : 0)) * Ws + (inb ? xx : 0);
      if constexpr ((FLAGS & CONV_D_DBG_SAMEPIX) != 0) pix = 0;
      aptr[i] = s.ptr + pix * s.stride + dcol;
      ainb[i] = inb;
    }
  };
  auto advance = [&]() {
    c0 += 16;
    if (c0 >= segC) {
      c0 = 0;
      if (++sg == p.nseg) { sg = 0; ++tap; }
      segC = p.seg[sg].C;
      setup_a();
    }
  };
  int kstep = 0;  // K-step the next issue() fetches
  const int nsteps = ntaps * (p.Ctot / 16);
  // one stage: AI + BI DMA instructions per wave.  Past the last K-step the A rows come from the zero page and
  // the B rows repeat the last step (finite values x 0).
  auto issue = [&](float* stage) {
    const bool real = kstep < nsteps;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const float* src = (real && ainb[i]) ? aptr[i] + c0 : zsrc;
      conv_glds16(src, stage + (wv + NW * i) * 256);
    }
    const int kb = (real ? kstep : nsteps - 1) * 16;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int qb = (wv + NW * i) % BQ;
      conv_glds16(bptr[i] + kb, stage + (BM / 16 + qb) * 256);
    }
    ++kstep;
    if (kstep < nsteps) advance();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment reads ---------------------------------------------------------------------------
  // Inline-asm ds_read_b128 with hand-counted lgkmcnt waits: a compiler-visible LDS load would be preceded by
  // `s_waitcnt vmcnt(0)` (hipcc assumes it may alias the LDS-DMA writes in flight), which drains the pipeline.
  // Byte address = row * 64 + swizzled chunk * 16; chunk of (kq, half) = kq*2 + half.
  const int sw = (l31 >> 2) & 3;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned a_ad0 = lds0 + (wm * WTM + l31) * 64 + (half ^ sw) * 16;              // kq = 0
  const unsigned a_ad1 = lds0 + (wm * WTM + l31) * 64 + ((2 + half) ^ sw) * 16;        // kq = 1
  const unsigned b_ad0 = lds0 + (BM + wn * WTN + l31) * 64 + (half ^ sw) * 16;
  const unsigned b_ad1 = lds0 + (BM + wn * WTN + l31) * 64 + ((2 + half) ^ sw) * 16;

  auto compute = [&](auto stage_c) {
    constexpr int SB = decltype(stage_c)::value * STAGE * 4;  // byte offset of the stage
    v4f a[2][TM], b[2][TN];
    conv_lds_read_frags<SB>(a[0], a_ad0, std::make_integer_sequence<int, TM>{});
    conv_lds_read_frags<SB>(b[0], b_ad0, std::make_integer_sequence<int, TN>{});
    conv_lds_read_frags<SB>(a[1], a_ad1, std::make_integer_sequence<int, TM>{});
    conv_lds_read_frags<SB>(b[1], b_ad1, std::make_integer_sequence<int, TN>{});
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      if (kq == 0) conv_wait_lgkmcnt<TM + TN>(); else conv_wait_lgkmcnt<0>();
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) asm volatile("" : "+v"(a[kq][mt]));
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) asm volatile("" : "+v"(b[kq][nt]));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kq][mt][j], b[kq][nt][j], acc[mt][nt], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;

  float* const st0 = smem;
  float* const st1 = smem + STAGE;
  float* const st2 = smem + 2 * STAGE;

  setup_a();
  issue(st0);
  issue(st1);
  conv_wait_vmcnt<NI>();  // stage 0 landed (this wave's part)
  conv_barrier();
  const int nloop = (nsteps + 2) / 3;
  auto issue_l = [&](float* st) { if constexpr ((FLAGS & CONV_D_DBG_NODMA) == 0) issue(st); };
  auto sync_l = [&]() {
    if constexpr ((FLAGS & (CONV_D_DBG_NOVMWAIT | CONV_D_DBG_NODMA)) == 0) conv_wait_vmcnt<NI>();
    if constexpr ((FLAGS & CONV_D_DBG_NOBARRIER) == 0) conv_barrier();
  };
  for (int it = 0; it < nloop; ++it) {
    issue_l(st2);
    compute(S0{});
    sync_l();
    issue_l(st0);
    compute(S1{});
    sync_l();
    issue_l(st1);
    compute(S2{});
    sync_l();
  }
  conv_wait_vmcnt<0>();  // the surplus stages issued past the end

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

template <int BM, int BN, int WGM, int WGN, int FLAGS>
hipError_t conv_dma_launch(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = 3 * (size_t)(BM + BN) * 16 * sizeof(float);
  auto kern = conv_dma_kernel<BM, BN, WGM, WGN, FLAGS>;
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
