// RETIRED (round 4): the A/B tool of rounds 1-3 (conv_igemm / halo / F(2,3) / F(4,3) variants).  It no longer builds against csrc/: the
// experiment flags it drove (W43_DBG_*, W43_F_SETPRIO / OLDLOOP / PERSIST / PRE*) and the round-3 nested-Winograd template were removed.
// tools/w2d_bench.hip is the current tool; the logs this one produced are under profiles/r0[1-3]_*.
// conv_bench.hip -- micro-benchmark of conv_igemm_kernel variants on the real layer shapes of a 1080p
// 2x2-tiled forward (4 tiles of 960x576; both images / both directions batched where the engine does).
// Development tool, not part of the product library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/conv_bench.hip -o /tmp/conv_bench
//   /tmp/conv_bench [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../frame-interpolation_amd/csrc/conv_igemm_impl.h"
#include "../frame-interpolation_amd/csrc/conv_split_impl.h"
#include "../frame-interpolation_amd/csrc/conv_winox3_impl.h"
#include "../frame-interpolation_amd/csrc/conv_wino43_impl.h"
#include "../frame-interpolation_amd/csrc/conv_wino2d_impl.h"
#include "../frame-interpolation_amd/csrc/conv_wino_impl.h"
#include "../frame-interpolation_amd/csrc/conv_buf_impl.h"
#include "../frame-interpolation_amd/csrc/conv_halo_impl.h"

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

__global__ void fill_kernel(float* dst, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    dst[i] = (float)(int)x * (1.0f / 2147483648.0f) * scale;
  }
}

__global__ void checksum_kernel(const float* a, size_t n, double* out) {
  __shared__ double sh[256];
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += fabs((double)a[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, sh[0]);
}

typedef hipError_t (*LaunchFn)(const ConvParams&, hipStream_t);
struct Variant { const char* name; int bm, bn, bkc; int wkind; LaunchFn fn; };  // wkind 0: [K][N]  1: [N][K]  2: [N][chunk][tap][16]
#define V(BM, BN, WM, WN, BKC, FL) {#BM "x" #BN " w" #WM "x" #WN " bk" #BKC " f" #FL, BM, BN, BKC, 0, conv_igemm_launch<BM, BN, WM, WN, BKC, FL>}

#define B(BM, BN, WM, WN, FL) {"buf " #BM "x" #BN " w" #WM "x" #WN " f" #FL, BM, BN, 16, 1, conv_buf_launch<BM, BN, WM, WN, FL>}
#define H(TH, BN, WM, WN, FL) {"halo " #TH "x32x" #BN " w" #WM "x" #WN " f" #FL, TH * 32, BN, 16, 2, conv_halo_launch<TH, BN, WM, WN, FL>}
#define W8(TH, BN, WM, WN) {"wino " #TH "x64x" #BN " w" #WM "x" #WN " f4", TH * 64, BN, 8, 5, conv_wino_launch<TH, BN, WM, WN, 4>}
#define X(TH, BN, TM, TN) {"winox3 " #TH "x64x" #BN " t" #TM "x" #TN " f4", TH * 64, BN, 16, 6, conv_winox3_launch<TH, BN, TM, TN, 4>}
#define XA(TH, BN, TM, TN, FL) {"winox3 " #TH "x64x" #BN " t" #TM "x" #TN " f" #FL, TH * 64, BN, 16, 6, conv_winox3_launch<TH, BN, TM, TN, FL>}
#define F43(TH, BN, TM, TN) {"wino43 " #TH "x128x" #BN " t" #TM "x" #TN " f4", TH * 128, BN, 8, 7, conv_wino43_launch<TH, BN, TM, TN, 4>}
#define F43Q(TH, BN, TM, TN, FL) {"wino43 q16 " #TH "x64x" #BN " t" #TM "x" #TN " f" #FL, TH * 64, BN, 8, 7, conv_wino43_launch<TH, BN, TM, TN, FL, 16>}
#define F43N(TH, BN, FL) {"wino43 q16 nh1 " #TH "x64x" #BN " t1x1 f" #FL, TH * 64, BN, 8, 7, conv_wino43_launch<TH, BN, 1, 1, FL, 16, 1>}
#define F43Q8(TH, BN, TM, TN, FL) {"wino43 q8 " #TH "x32x" #BN " t" #TM "x" #TN " f" #FL, TH * 32, BN, 8, 7, conv_wino43_launch<TH, BN, TM, TN, FL, 8>}
#define F43N8(TH, BN, FL) {"wino43 q8 nh1 " #TH "x32x" #BN " t1x1 f" #FL, TH * 32, BN, 8, 7, conv_wino43_launch<TH, BN, 1, 1, FL, 8, 1>}
template <LaunchFn F, int G>
hipError_t persist_launch(const ConvParams& p, hipStream_t s) { ConvParams q = p; q.persist = G; return F(q, s); }
#define PQ8(TH, BN, TM, TN, FL, G) {"wino43 q8 pers" #G " " #TH "x32x" #BN " t" #TM "x" #TN " f" #FL, TH * 32, BN, 8, 7, persist_launch<conv_wino43_launch<TH, BN, TM, TN, FL, 8>, G>}
#define PN8(TH, BN, FL, G) {"wino43 q8 nh1 pers" #G " " #TH "x32x" #BN " t1x1 f" #FL, TH * 32, BN, 8, 7, persist_launch<conv_wino43_launch<TH, BN, 1, 1, FL, 8, 1>, G>}
#define PQ(TH, BN, TM, TN, FL, G) {"wino43 q16 pers" #G " " #TH "x64x" #BN " t" #TM "x" #TN " f" #FL, TH * 64, BN, 8, 7, persist_launch<conv_wino43_launch<TH, BN, TM, TN, FL, 16>, G>}
#define PN(TH, BN, FL, G) {"wino43 q16 nh1 pers" #G " " #TH "x64x" #BN " t1x1 f" #FL, TH * 64, BN, 8, 7, persist_launch<conv_wino43_launch<TH, BN, 1, 1, FL, 16, 1>, G>}
#define W2D(TH, BN, FL, QW) {"wino2d q" #QW " " #TH "x" #BN " f" #FL, TH * 4 * QW, BN, 8, 8, conv_wino2d_launch<TH, BN, FL, QW>}
#define F43F(TH, BN, TM, TN, FL) {"wino43 " #TH "x128x" #BN " t" #TM "x" #TN " f" #FL, TH * 128, BN, 8, 7, conv_wino43_launch<TH, BN, TM, TN, FL>}
#define S(TH, BN, WM, WN, NP) {"split" #NP " " #TH "x32x" #BN " w" #WM "x" #WN " f4", TH * 32, BN, 16, 3, conv_halo_split_launch<TH, BN, WM, WN, NP, 4>}
static Variant variants[] = {
    V(128, 128, 2, 2, 16, 4), B(128, 128, 2, 2, 4),
    W8(4, 64, 4, 2), W8(4, 128, 4, 2), W8(4, 128, 4, 4), W8(4, 32, 4, 1), W8(8, 64, 8, 2), W8(8, 32, 8, 1), W8(2, 64, 2, 2),
    F43(4, 64, 1, 2), F43(4, 64, 2, 1), F43(4, 32, 1, 1), F43F(4, 64, 2, 1, 68), F43F(4, 64, 2, 1, 0),
    F43Q(4, 64, 2, 1, 4), F43Q(4, 64, 1, 2, 4), F43Q(4, 32, 1, 1, 4), F43Q(4, 64, 2, 1, 0),
    F43N(4, 64, 4), F43N(4, 64, 0),
    F43Q(4, 64, 2, 1, 260), F43Q(4, 64, 2, 1, 3844), F43N(4, 64, 32772), F43N(4, 64, 32768), F43Q(4, 64, 1, 2, 32772), F43Q(4, 64, 2, 1, 32772), F43Q(4, 32, 1, 1, 32772),
    F43Q(4, 32, 1, 1, 65540), F43N(4, 64, 65540), F43Q(4, 64, 2, 1, 65540),
    F43Q8(8, 64, 2, 1, 32772), F43Q8(8, 64, 1, 2, 32772), F43N8(8, 64, 32772), F43Q8(8, 32, 1, 1, 65540), F43Q8(8, 32, 1, 1, 32772),
    F43Q8(8, 64, 2, 1, 49156), F43Q8(8, 64, 1, 2, 49156), F43N8(8, 64, 49156), F43Q8(8, 32, 1, 1, 49156),
    W2D(8, 64, 4, 8), W2D(8, 32, 4, 8), W2D(4, 64, 4, 16), W2D(4, 32, 4, 16), W2D(8, 64, 0, 8),
    W2D(8, 64, 68, 8), W2D(8, 32, 68, 8), W2D(4, 64, 68, 16), W2D(4, 32, 68, 16),
    W2D(8, 64, 16452, 8), W2D(8, 32, 16452, 8), W2D(8, 64, 16580, 8), W2D(8, 32, 16580, 8), W2D(8, 64, 24772, 8), W2D(8, 32, 24772, 8),
    W2D(8, 64, 20548, 8), W2D(8, 32, 20548, 8), W2D(8, 64, 49220, 8), W2D(8, 32, 49220, 8), W2D(8, 64, 81988, 8), W2D(8, 32, 81988, 8), W2D(8, 64, 17476, 8), W2D(8, 32, 17476, 8), W2D(8, 64, 18500, 8), W2D(8, 32, 18500, 8), W2D(8, 64, 16708, 8), W2D(8, 32, 16708, 8),
    W2D(8, 64, 147524, 8), W2D(8, 32, 147524, 8),
    W2D(8, 64, 147780, 8), W2D(8, 32, 147780, 8), W2D(8, 64, 148036, 8), W2D(8, 32, 148036, 8), W2D(8, 64, 148548, 8), W2D(8, 32, 148548, 8), W2D(8, 64, 149572, 8), W2D(8, 32, 149572, 8), W2D(8, 64, 180292, 8), W2D(8, 32, 180292, 8), W2D(8, 64, 213060, 8), W2D(8, 32, 213060, 8),
    W2D(8, 64, 278596, 8), W2D(8, 32, 278596, 8), W2D(8, 64, 409668, 8), W2D(8, 32, 409668, 8),
    W2D(8, 64, 409664, 8), W2D(8, 32, 409664, 8), W2D(8, 64, 409672, 8), W2D(8, 32, 409672, 8), W2D(8, 64, 409680, 8), W2D(8, 32, 409680, 8), W2D(8, 64, 409696, 8), W2D(8, 32, 409696, 8), W2D(8, 64, 278592, 8), W2D(8, 32, 278592, 8), W2D(8, 64, 278600, 8), W2D(8, 32, 278600, 8), W2D(8, 64, 278608, 8), W2D(8, 32, 278608, 8), W2D(8, 64, 278624, 8), W2D(8, 32, 278624, 8),
    W2D(8, 64, 786496, 8), W2D(8, 32, 786496, 8), W2D(8, 64, 786500, 8), W2D(8, 32, 786500, 8), W2D(8, 64, 917568, 8), W2D(8, 64, 917572, 8), W2D(4, 64, 786500, 16), W2D(4, 32, 786500, 16), W2D(4, 64, 278596, 16), W2D(4, 32, 278596, 16),
    W2D(8, 64, 196, 8), W2D(8, 32, 196, 8), W2D(8, 64, 8388, 8), W2D(8, 32, 8388, 8), W2D(8, 64, 4164, 8), W2D(8, 32, 4164, 8),
    W2D(8, 64, 260, 8), W2D(8, 64, 516, 8), W2D(8, 64, 1028, 8), W2D(8, 64, 2052, 8), W2D(8, 64, 3844, 8), W2D(8, 32, 260, 8), W2D(8, 32, 516, 8), W2D(8, 32, 1028, 8), W2D(8, 32, 3844, 8),
    // persistent launches (flag 524288); + 131072: the next pair's first activation chunk requested before the epilogue, + 262144: its whole prologue
    PQ(4, 64, 2, 1, 557060, 2), PN(4, 64, 557060, 2), PQ(4, 32, 1, 1, 589828, 4),
    PQ(4, 64, 2, 1, 688132, 2), PN(4, 64, 688132, 2), PN(4, 64, 950276, 2),
    PQ8(8, 64, 2, 1, 557060, 2), PN8(8, 64, 557060, 2), PQ8(8, 32, 1, 1, 589828, 4),
    PQ8(8, 64, 2, 1, 688132, 2), PN8(8, 64, 688132, 2), PN8(8, 64, 950276, 2),
    X(4, 128, 2, 2), XA(4, 128, 2, 2, 1028), XA(4, 128, 2, 2, 2052), XA(4, 64, 2, 1, 1028), XA(4, 64, 2, 1, 2052), X(4, 64, 1, 2), X(4, 64, 2, 1), X(4, 32, 1, 1),
    S(4, 64, 4, 1, 6), S(4, 64, 4, 1, 3), S(8, 64, 4, 1, 6), S(4, 128, 2, 2, 6), S(8, 128, 4, 2, 6), S(8, 128, 4, 2, 3), S(8, 128, 2, 2, 3), S(16, 128, 4, 2, 3), S(16, 64, 4, 1, 3), S(8, 64, 2, 1, 3), S(16, 128, 4, 1, 3), S(8, 64, 4, 1, 3), S(4, 128, 2, 2, 3), S(8, 32, 4, 1, 3), S(8, 32, 4, 1, 6), S(8, 64, 2, 2, 6),
    H(8, 128, 4, 2, 4), H(8, 64, 4, 1, 4), H(8, 32, 4, 1, 4), H(4, 64, 4, 1, 4),
    H(4, 128, 2, 2, 4), H(8, 64, 2, 2, 4), B(256, 64, 4, 1, 4), B(128, 64, 2, 2, 4), B(64, 64, 2, 2, 4),
    B(256, 128, 4, 2, 4), B(256, 32, 4, 1, 4), B(128, 32, 4, 1, 4),
    V(128, 128, 2, 2, 16, 2052), V(128, 128, 2, 2, 16, 4100), V(128, 128, 2, 2, 16, 1028),
    V(128, 128, 2, 2, 16, 68),
    V(128, 128, 2, 2, 16, 452),
    V(256, 64, 4, 1, 16, 4),
    V(256, 32, 4, 1, 16, 4),
};

// [tap*C + c][N] -> [N][chunk][tap][16]
__global__ void pack_halo_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)9 * C * N) return;
  const int k = (int)(i / N), n = (int)(i % N);
  const int tap = k / C, c = k % C;
  dst[(((size_t)n * (C / 16) + c / 16) * 9 + tap) * 16 + c % 16] = src[i];
}

// [tap*C + c][N] fp32 -> [N][chunk][tap][plane][16] bf16, exact 3-way round-to-nearest split
__global__ void pack_split_kernel(const float* src, unsigned short* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)9 * C * N) return;
  const int k = (int)(i / N), n = (int)(i % N);
  const int tap = k / C, c = k % C;
  const float x = src[i];
  auto rne = [](float v) { unsigned u = __float_as_uint(v); u += 0x7FFFu + ((u >> 16) & 1u); return u & 0xFFFF0000u; };
  const unsigned hb = rne(x);
  const float r = x - __uint_as_float(hb);
  const unsigned mb = rne(r);
  const float q = r - __uint_as_float(mb);
  unsigned short* d = dst + ((((size_t)n * (C / 16) + c / 16) * 9 + tap) * 3) * 16 + c % 16;
  d[0] = (unsigned short)(hb >> 16); d[16] = (unsigned short)(mb >> 16); d[32] = (unsigned short)(rne(q) >> 16);
}

// [tap*C + c][N] -> [N][chunk][nu*3 + dy][16]: F(2,3) weight transform along x
__global__ void pack_wino_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)3 * C * N) return;
  const int n = (int)(i % N);
  const int c = (int)((i / N) % C), dy = (int)(i / ((size_t)N * C));
  const float g0 = src[((size_t)(dy * 3 + 0) * C + c) * N + n], g1 = src[((size_t)(dy * 3 + 1) * C + c) * N + n],
              g2 = src[((size_t)(dy * 3 + 2) * C + c) * N + n];
  const float u[4] = {g0, ((g0 + g2) + g1) * 0.5f, ((g0 + g2) - g1) * 0.5f, g2};
  for (int nu = 0; nu < 4; ++nu)
    dst[(((size_t)n * (C / 16) + c / 16) * 12 + nu * 3 + dy) * 16 + c % 16] = u[nu];
}

// [tap*C + c][N] -> [N][chunk8][nu*3 + dy][8]
__global__ void pack_wino8_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)3 * C * N) return;
  const int n = (int)(i % N);
  const int c = (int)((i / N) % C), dy = (int)(i / ((size_t)N * C));
  const float g0 = src[((size_t)(dy * 3 + 0) * C + c) * N + n], g1 = src[((size_t)(dy * 3 + 1) * C + c) * N + n],
              g2 = src[((size_t)(dy * 3 + 2) * C + c) * N + n];
  const float u[4] = {g0, ((g0 + g2) + g1) * 0.5f, ((g0 + g2) - g1) * 0.5f, g2};
  for (int nu = 0; nu < 4; ++nu)
    dst[(((size_t)n * (C / 8) + c / 8) * 12 + nu * 3 + dy) * 8 + c % 8] = u[nu];
}

// [tap*C + c][N] -> [N][chunk16][dy][j][h][plane][16] bf16: F(2,3) weight transform in fp32, then nearest hi / mid split
__global__ void pack_winox3_kernel(const float* src, unsigned short* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)3 * C * N) return;
  const int n = (int)(i % N);
  const int c = (int)((i / N) % C), dy = (int)(i / ((size_t)N * C));
  const float g0 = src[((size_t)(dy * 3 + 0) * C + c) * N + n], g1 = src[((size_t)(dy * 3 + 1) * C + c) * N + n],
              g2 = src[((size_t)(dy * 3 + 2) * C + c) * N + n];
  const float u[4] = {g0, ((g0 + g2) + g1) * 0.5f, ((g0 + g2) - g1) * 0.5f, g2};
  auto rne = [](float v) { unsigned w = __float_as_uint(v); w += 0x7FFFu + ((w >> 16) & 1u); return w & 0xFFFF0000u; };
  for (int nu = 0; nu < 4; ++nu) {
    unsigned short* d = dst + ((((size_t)n * (C / 16) + c / 16) * 3 + dy) * 2 + (nu & 1)) * 64 + (nu >> 1) * 32 + c % 16;
    const unsigned hb = rne(u[nu]);
    d[0] = (unsigned short)(hb >> 16);
    d[16] = (unsigned short)(rne(u[nu] - __uint_as_float(hb)) >> 16);
  }
}

// [tap*C + c][N] -> [N][chunk8][dy][nu 6][8]: F(4,3) weight transform along x (conv_wino43_impl.h)
__global__ void pack_wino43_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)3 * C * N) return;
  const int n = (int)(i % N);
  const int c = (int)((i / N) % C), dy = (int)(i / ((size_t)N * C));
  const float g0 = src[((size_t)(dy * 3 + 0) * C + c) * N + n], g1 = src[((size_t)(dy * 3 + 1) * C + c) * N + n],
              g2 = src[((size_t)(dy * 3 + 2) * C + c) * N + n];
  const float u[6] = {g0 * 0.25f, -((g0 + g2) + g1) * (1.f / 6.f), -((g0 + g2) - g1) * (1.f / 6.f),
                      (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) + g1 * (1.f / 12.f), (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) - g1 * (1.f / 12.f), g2};
  for (int nu = 0; nu < 6; ++nu)
    dst[((((size_t)n * (C / 8) + c / 8) * 3 + dy) * 6 + nu) * 8 + c % 8] = u[nu];
}

// [tap*C + c][N] -> [N/32][chunk8][mu 4][nu 6][K half][32][4]: F(4,3) along x, then F(2,3) along y (conv_wino2d_impl.h)
__global__ void pack_wino2d_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)C * N) return;
  const int n = (int)(i % N), c = (int)(i / N);
  float u[3][6];
  for (int dy = 0; dy < 3; ++dy) {
    const float g0 = src[((size_t)(dy * 3 + 0) * C + c) * N + n], g1 = src[((size_t)(dy * 3 + 1) * C + c) * N + n],
                g2 = src[((size_t)(dy * 3 + 2) * C + c) * N + n];
    u[dy][0] = g0 * 0.25f;
    u[dy][1] = -((g0 + g2) + g1) * (1.f / 6.f);
    u[dy][2] = -((g0 + g2) - g1) * (1.f / 6.f);
    u[dy][3] = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) + g1 * (1.f / 12.f);
    u[dy][4] = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) - g1 * (1.f / 12.f);
    u[dy][5] = g2;
  }
  for (int nu = 0; nu < 6; ++nu) {
    const float U[4] = {u[0][nu], ((u[0][nu] + u[2][nu]) + u[1][nu]) * 0.5f, ((u[0][nu] + u[2][nu]) - u[1][nu]) * 0.5f, u[2][nu]};
    for (int mu = 0; mu < 4; ++mu)
      dst[(((((size_t)(n / 32) * (C / 8) + c / 8) * 4 + mu) * 6 + nu) * 2 + (c % 8) / 4) * 128 + (n % 32) * 4 + c % 4] = U[mu];
  }
}

__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));  // non-negative floats order like unsigned ints
}

// [K][N] -> [N][K]
__global__ void transpose_kernel(const float* src, float* dst, int K, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)K * N) return;
  const int k = (int)(i / N), n = (int)(i % N);
  dst[(size_t)n * K + k] = src[i];
}

struct Shape { const char* name; int NB, H, W, C, Cout, ks; };
static Shape shapes[] = {
    {"fusion_1_1  M=552960 C=528->128 3x3", 4, 288, 480, 528, 128, 3},
    {"feat_conv3  M=1.1M   C=128->128 3x3", 8, 288, 480, 128, 128, 3},
    {"feat_conv1  M=4.4M   C=64->64 3x3", 8, 576, 960, 64, 64, 3},
    {"flow_l0_c0  M=4.4M   C=128->32 3x3", 8, 576, 960, 128, 32, 3},
    {"ragged      36x60    C=64->64 3x3", 3, 36, 60, 64, 64, 3},
    {"fusion_0_1  M=2.2M   C=208->64 3x3", 4, 576, 960, 208, 64, 3},
    {"flow_l1_c0  M=1.1M   C=384->64 3x3", 8, 288, 480, 384, 64, 3},
    {"fusion_3_1  M=34560  C=2448->512 3x3", 4, 72, 120, 2448, 512, 3},
    {"flow_l3_c0  M=69120  C=1920->256 3x3", 8, 72, 120, 1920, 256, 3},
    {"flow_l0_c1  M=4.4M   C=32->32 3x3", 8, 576, 960, 32, 32, 3},
    {"flow_l1_c1  M=1.1M   C=64->64 3x3", 8, 288, 480, 64, 64, 3},
    {"fusion_2_1  M=138240 C=1168->256 3x3", 4, 144, 240, 1168, 256, 3},
    {"feat_conv5  M=276480 C=256->256 3x3", 8, 144, 240, 256, 256, 3},
    {"fusion_0_2  M=2.2M   C=64->64 3x3", 4, 576, 960, 64, 64, 3},
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;      // -1: all shapes
  const char* only_variant = argc > 3 ? argv[3] : nullptr;   // substring filter on the variant name
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double* d_sum;
  CK(hipMalloc(&d_sum, sizeof(double)));
  int shape_idx = -1;
  for (const Shape& sh : shapes) {
    if (++shape_idx != only_shape && only_shape >= 0) continue;
    const size_t M = (size_t)sh.NB * sh.H * sh.W;
    const size_t n_in = M * sh.C, n_w = (size_t)sh.ks * sh.ks * sh.C * sh.Cout, n_out = M * sh.Cout;
    float *d_in, *d_w, *d_wt, *d_wh, *d_ww, *d_w8, *d_b, *d_out, *d_zero, *d_ref, *d_md;
    unsigned short *d_ws, *d_wx;
    float* d_w43;
    float* d_w2d;
    CK(hipMalloc(&d_in, n_in * 4));
    CK(hipMalloc(&d_w, n_w * 4));
    CK(hipMalloc(&d_wt, n_w * 4));
    CK(hipMalloc(&d_wh, n_w * 4));
    CK(hipMalloc(&d_ws, n_w * 6));
    CK(hipMalloc(&d_ww, n_w * 4 * 12 / 9 + 64));
    CK(hipMalloc(&d_w8, n_w * 4 * 12 / 9 + 64));
    CK(hipMalloc(&d_wx, n_w * 4 * 12 / 9 + 64));
    CK(hipMalloc(&d_w43, n_w * 4 * 18 / 9 + 64));
    CK(hipMalloc(&d_w2d, n_w * 4 * 24 / 9 + 64));
    CK(hipMalloc(&d_ref, n_out * 4));
    CK(hipMalloc(&d_md, 4));
    CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    CK(hipMalloc(&d_b, sh.Cout * 4));
    CK(hipMalloc(&d_out, n_out * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_in, n_in, 1u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_w, n_w, 2u, 0.05f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, d_b, (size_t)sh.Cout, 3u, 0.1f);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, st, d_w, d_wt, sh.ks * sh.ks * sh.C, sh.Cout);
    if (sh.ks == 3) hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((n_w / 3 + 255) / 256)), dim3(256), 0, st, d_w, d_ww, sh.C, sh.Cout);
    if (sh.ks == 3 && sh.C % 16 == 0) hipLaunchKernelGGL(pack_winox3_kernel, dim3((unsigned)((n_w / 3 + 255) / 256)), dim3(256), 0, st, d_w, d_wx, sh.C, sh.Cout);
    if (sh.ks == 3) hipLaunchKernelGGL(pack_wino43_kernel, dim3((unsigned)((n_w / 3 + 255) / 256)), dim3(256), 0, st, d_w, d_w43, sh.C, sh.Cout);
    if (sh.ks == 3 && sh.Cout % 32 == 0) hipLaunchKernelGGL(pack_wino2d_kernel, dim3((unsigned)((n_w / 9 + 255) / 256)), dim3(256), 0, st, d_w, d_w2d, sh.C, sh.Cout);
    if (sh.ks == 3) hipLaunchKernelGGL(pack_wino8_kernel, dim3((unsigned)((n_w / 3 + 255) / 256)), dim3(256), 0, st, d_w, d_w8, sh.C, sh.Cout);
    if (sh.ks == 3) hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, st, d_w, d_ws, sh.C, sh.Cout);
    if (sh.ks == 3) hipLaunchKernelGGL(pack_halo_kernel, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, st, d_w, d_wh, sh.C, sh.Cout);
    CK(hipStreamSynchronize(st));
    ConvParams p{};
    p.nseg = 1;
    p.seg[0].ptr = d_in; p.seg[0].stride = sh.C; p.seg[0].C = sh.C;
    p.ksize = sh.ks; p.w = d_w; p.bias = d_b; p.out = d_out; p.ostride = sh.Cout;
    p.NB = sh.NB; p.H = sh.H; p.W = sh.W; p.Cout = sh.Cout; p.Ctot = sh.C; p.leaky = 1; p.M = (int)M;
    const double flops = 2.0 * M * sh.Cout * sh.ks * sh.ks * sh.C;
    printf("== %s  (%.1f GFLOP)\n", sh.name, flops * 1e-9);
    double ref_sum = -1;
    for (const Variant& v : variants) {
      if (sh.Cout % v.bn || sh.C % v.bkc) continue;
      if (only_variant) {   // comma-separated substrings: any match
        bool hit = false;
        std::string flt(only_variant);
        for (size_t b = 0; b <= flt.size();) {
          const size_t e = flt.find(',', b) == std::string::npos ? flt.size() : flt.find(',', b);
          if (e > b && strstr(v.name, flt.substr(b, e - b).c_str())) hit = true;
          b = e + 1;
        }
        if (!hit) continue;
      }
      CK(hipMemsetAsync(d_out, 0, n_out * 4, st));
      if (v.wkind >= 2 && sh.ks != 3) continue;
      p.w = v.wkind == 8 ? d_w2d : v.wkind == 7 ? d_w43 : v.wkind == 6 ? reinterpret_cast<const float*>(d_wx) : v.wkind == 5 ? d_w8 : v.wkind == 3 ? reinterpret_cast<const float*>(d_ws) : v.wkind == 2 ? d_wh : v.wkind == 1 ? d_wt : d_w;
      CK(v.fn(p, st));  // warm + correctness
      CK(hipMemsetAsync(d_sum, 0, sizeof(double), st));
      hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, st, d_out, n_out, d_sum);
      double sum = 0;
      CK(hipMemcpyAsync(&sum, d_sum, sizeof(double), hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      float maxdiff = 0.f;
      if (ref_sum < 0) {
        ref_sum = sum;
        CK(hipMemcpyAsync(d_ref, d_out, n_out * 4, hipMemcpyDeviceToDevice, st));
      } else {
        CK(hipMemsetAsync(d_md, 0, 4, st));
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, st, d_out, d_ref, n_out, d_md);
        CK(hipMemcpyAsync(&maxdiff, d_md, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      }
      float best = 1e30f, tot = 0;
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK(v.fn(p, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms); tot += ms;
      }
      const double rel = fabs(sum - ref_sum) / ref_sum;
      printf("   %-24s  min %8.3f ms  avg %8.3f ms  %7.1f TF/s  chk %s (%.1e) max|d| %.2e\n", v.name, best, tot / reps,
             flops / best * 1e-9, rel < 1e-5 ? "ok" : "(ablation)", rel, maxdiff);
      fflush(stdout);
    }
    CK(hipFree(d_in)); CK(hipFree(d_w)); CK(hipFree(d_wt)); CK(hipFree(d_zero)); CK(hipFree(d_wh)); CK(hipFree(d_ww)); CK(hipFree(d_w8)); CK(hipFree(d_w43)); CK(hipFree(d_w2d)); CK(hipFree(d_wx)); CK(hipFree(d_ws)); CK(hipFree(d_ref)); CK(hipFree(d_md)); CK(hipFree(d_b)); CK(hipFree(d_out));
  }
  return 0;
}
