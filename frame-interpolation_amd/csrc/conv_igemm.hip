// conv_igemm.hip -- Conv2D('same', stride 1) + bias + leaky_relu(0.2) as an fp32 implicit GEMM on
// the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain).
//
// Replaces every tf.keras.layers.Conv2D of the reference hot path whose Cout is a multiple of 32:
//   feature_extractor.py:93-99,142-143   pyramid_flow_estimator.py:66-76,85-98   fusion.py:83-97,135-138
//
// Mapping:  M = NB*H*W output pixels (raster order), N = Cout, K = taps x (concatenated input channels).
//   * one workgroup = 256 threads = 4 waves computes a BM x BN tile; each wave owns a grid of 32x32
//     MFMA tiles with their accumulators in registers.
//   * K is walked in steps of 16 channels of one tap of one concat segment.  The A tile
//     (BM pixels x 16 channels, gathered from NHWC with zero fill outside the image) and the B tile
//     (16 x BN weights) are staged global -> registers -> LDS, double buffered, one barrier per step;
//     the global loads of step s+1 are issued before the MFMAs of step s.
//   * A rows are stored with a stride of 20 floats so that the ds_read_b128 fragment reads are
//     bank-conflict free (MI355X_MICROARCH.md, LDS table: rows distinct mod 16 per 16-lane group).
//     A lane's float4 gives it 4 K-values; lanes 0-31 take channels {0..3}, lanes 32-63 {4..7} of
//     each 8-channel group and the B fragment is read with the same permutation, so MFMA j of a
//     group multiplies channels (j, 4+j) -- a permutation of K, which only reorders the sum.
//   * the epilogue adds the bias, applies leaky_relu and stores 128-byte rows into the channel
//     slice of the destination buffer (the consumer's concat input).
#include "film_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = FILM_BK;
constexpr int AST = 20;  // LDS row stride of the A tile in floats (16 + 4 pad, keeps 16-B alignment)

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // 32x32 MFMA tiles per wave
  constexpr int AROWS = BM / 64;                 // A rows staged per thread
  constexpr int BF4 = BK * BN / 4;               // float4 in a B tile
  constexpr int BLD = (BF4 + 255) / 256;         // B float4 staged per thread
  constexpr int A_SZ = BM * AST, B_SZ = BK * BN;
  static_assert(TM >= 1 && TN >= 1 && AROWS >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wv / WGN, wn = wv % WGN;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread A staging rows -------------------------------------------------------------
  const int arow = t >> 2;          // 0..63
  const int acol = (t & 3) * 4;     // channel offset inside the 16-channel chunk
  int ab[AROWS], ay[AROWS], ax[AROWS];
  bool avalid[AROWS];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int m = m0 + arow + 64 * i;
    avalid[i] = m < p.M;
    int mm = avalid[i] ? m : 0;
    int b = mm / HW;
    int r = mm - b * HW;
    int y = r / p.W;
    ab[i] = b; ay[i] = y; ax[i] = r - y * p.W;
  }
  const int pad = (p.ksize - 1) >> 1;
  const int ntaps = p.ksize * p.ksize;

  // ---- per-thread B staging ------------------------------------------------------------------
  // float4 index f = t + 256*i -> (krow = f / (BN/4), n4 = f % (BN/4))
  // Every thread issues its loads unconditionally (a predicated load makes hipcc branch around it and
  // wait vmcnt(0) right behind it - cdna_hip_programming.md "three .s-level traps" (c)); when the B tile
  // has fewer than 256 float4 (BN = 32) the upper threads re-read a valid element and skip the LDS store.
  constexpr bool B_ALL = (BF4 % 256) == 0;
  const float* bbase[BLD];
#pragma unroll
  for (int i = 0; i < BLD; ++i) {
    const int f = (t + 256 * i) % BF4;
    const int krow = f / (BN / 4), n4 = f % (BN / 4);
    bbase[i] = p.w + (size_t)krow * p.Cout + n0 + n4 * 4;
  }
  const bool bstore = B_ALL || t < BF4;

  // ---- K iteration state ---------------------------------------------------------------------
  int tap = 0, sg = 0, c0 = 0;
  const float* aptr[AROWS];
  bool ainb[AROWS];       // in-image mask of the rows for the CURRENT (tap, segment)
  bool ainb_prev[AROWS];  // mask that belongs to the data sitting in ra[] (set by load_global's caller)

  auto setup_a = [&]() {  // called when (tap, segment) changes
    const int dy = tap / p.ksize - pad, dx = tap % p.ksize - pad;
    const ConvSeg& s = p.seg[sg];
    const int Hs = s.up ? (p.H >> 1) : p.H, Ws = s.up ? (p.W >> 1) : p.W;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int yy = ay[i] + dy, xx = ax[i] + dx;
      bool inb = avalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      if (s.up) { yy >>= 1; xx >>= 1; }
      int be = ab[i] + s.boff;
      if (s.bmod && be >= s.bmod) be -= s.bmod;
      size_t pix = ((size_t)be * Hs + (inb ? yy : 0)) * Ws + (inb ? xx : 0);
      aptr[i] = s.ptr + pix * s.stride + acol;
      ainb[i] = inb;
    }
  };

  static_assert(BLD == 1 || BLD == 2, "B staging holds one or two float4 per thread");
  float4 ra[AROWS];
  float4 rb0, rb1;  // named scalars: a 2-element array here is left in scratch memory by hipcc
  int kstep = 0;  // global K-step index == weight row / 16
  auto load_global = [&]() {
    // out-of-image rows point at a valid pixel (setup_a) and are zeroed when written to LDS
#pragma unroll
    for (int i = 0; i < AROWS; ++i) ra[i] = *reinterpret_cast<const float4*>(aptr[i] + c0);
    const size_t koff = (size_t)kstep * BK * p.Cout;
    rb0 = *reinterpret_cast<const float4*>(bbase[0] + koff);
    if constexpr (BLD == 2) rb1 = *reinterpret_cast<const float4*>(bbase[BLD - 1] + koff);
  };
  auto store_lds = [&](int buf) {
    float* As = smem + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      float4 v = ra[i];
      v.x = ainb_prev[i] ? v.x : 0.f; v.y = ainb_prev[i] ? v.y : 0.f;
      v.z = ainb_prev[i] ? v.z : 0.f; v.w = ainb_prev[i] ? v.w : 0.f;
      *reinterpret_cast<float4*>(As + (arow + 64 * i) * AST + acol) = v;
    }
    if (bstore) {
      *reinterpret_cast<float4*>(Bs + (t % BF4) * 4) = rb0;
      if constexpr (BLD == 2) *reinterpret_cast<float4*>(Bs + (t + 256) * 4) = rb1;
    }
  };
  auto advance = [&]() {
    ++kstep;
    c0 += BK;
    if (c0 >= p.seg[sg].C) {
      c0 = 0;
      if (++sg == p.nseg) { sg = 0; ++tap; }
      setup_a();
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = ntaps * (p.Ctot / BK);

  setup_a();
  load_global();
#pragma unroll
  for (int i = 0; i < AROWS; ++i) ainb_prev[i] = ainb[i];
  store_lds(0);
  __syncthreads();

  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) {
      advance();
      load_global();
#pragma unroll
      for (int i = 0; i < AROWS; ++i) ainb_prev[i] = ainb[i];
    }

    const float* As = smem + cur * (A_SZ + B_SZ);
    const float* Bs = As + A_SZ;
    // all fragment reads of the step first (16 K-values: 2 x ds_read_b128 per M tile, 8 x ds_read_b32
    // per N tile), then the 8*TM*TN MFMAs: the compiler retires the reads with counted lgkmcnt waits
    // while the matrix pipe is already busy.
    float4 a[2][TM];
    float b[2][4][TN];
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
        a[kq][mt] = *reinterpret_cast<const float4*>(As + (wm * WTM + mt * 32 + l31) * AST + kq * 8 + half * 4);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) b[kq][j][nt] = Bs[(kq * 8 + half * 4 + j) * BN + wn * WTN + nt * 32 + l31];
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

    if (more) store_lds(cur ^ 1);
    __syncthreads();
    cur ^= 1;
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

template <int BM, int BN, int WGM, int WGN>
hipError_t launch(const ConvParams& p, hipStream_t s) {
  dim3 grid((p.M + BM - 1) / BM, p.Cout / BN);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace

hipError_t film_launch_conv(const ConvParams& p, int tile, hipStream_t s) {
  switch (tile) {
    case TILE_128x128: return launch<128, 128, 2, 2>(p, s);
    case TILE_256x64: return launch<256, 64, 4, 1>(p, s);
    case TILE_256x32: return launch<256, 32, 4, 1>(p, s);
    case TILE_64x64: return launch<64, 64, 2, 2>(p, s);
    case TILE_128x32: return launch<128, 32, 4, 1>(p, s);
    case TILE_128x64: return launch<128, 64, 2, 2>(p, s);
    case TILE_256x128: return launch<256, 128, 4, 1>(p, s);
    default: return hipErrorInvalidValue;
  }
}
