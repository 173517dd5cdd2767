// w2d_bench.hip -- micro-benchmark + bit-identity check of the nested-Winograd kernel conv_wino2d_kernel (its tiles against each other;
// the W2D_F_XFIRST variants keep the summation order - and the bits - of round 3's kernel, which left the tree in round 5) and of its
// split-K form against the unsplit result, with the 1-D F(4,3) kernel as the timing reference, on the layer shapes they run in a 1080p
// 2x2-tiled forward, plus a two-segment input with a batch remap and ragged levels.  Development tool (lean sibling of conv_bench.hip: compiles in a minute), not part of the product library.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/w2d_bench.hip -o tools/bin/w2d_bench
//   tools/bin/w2d_bench [reps] [shape index | -1] [variant substring[,substring...]]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../frame-interpolation_amd/csrc/conv_wino43_impl.h"
#include "../frame-interpolation_amd/csrc/conv_wino2d_impl.h"

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

__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = fabsf(a[i] - b[i]);
    m = fmaxf(m, d == d ? d : 1e30f);   // NaN counts as a difference
  }
  atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));  // non-negative floats order like unsigned ints
}

typedef hipError_t (*LaunchFn)(const ConvParams&, hipStream_t);
// fam 3: split-K (its own sums: the distance from family 1 is printed, never counted as a mismatch);
// fam 0: the bits of round 3's kernel (the W2D_F_XFIRST order; reference = the first variant run); fam 1: conv_wino2d_kernel's own family (reference = the
// first fam-1 variant run; its distance from family 0 is printed); fam -1: timing ablation (wrong on purpose); fam 2: the 1-D F(4,3)
// kernel conv_wino43_kernel (its own weights and sums: timing reference, distance from family 0 printed)
struct Variant { const char* name; int bn; int fam; LaunchFn fn; };
#define W2N(NAME, BN, FL) {"w2d " NAME, BN, ((FL) & 0x3F00) ? -1 : ((FL) & W2D_F_XFIRST) ? 0 : 1, conv_wino2d_launch<BN, FL>}
// split-K: conv_wino2d_kernel with S K ranges (blockIdx.z) + the ordered reduction (the engine's conv_splitk_reduce_kernel, restated)
static float* g_part = nullptr;
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                            int M, int Cout, int ostride, int S, int leaky) {
  const int G = Cout >> 2;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= (unsigned)M * (unsigned)G) return;
  const unsigned m = idx / (unsigned)G, g = idx - m * (unsigned)G;
  const size_t plane = (size_t)M * Cout;
  const float4* src = reinterpret_cast<const float4*>(part + (size_t)m * Cout + g * 4);
  float4 a = *reinterpret_cast<const float4*>(bias + g * 4);
  for (int sidx = 0; sidx < S; ++sidx) {
    const float4 v = src[(size_t)sidx * (plane >> 2)];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (leaky) {
    a.x = a.x > 0.f ? a.x : 0.2f * a.x; a.y = a.y > 0.f ? a.y : 0.2f * a.y;
    a.z = a.z > 0.f ? a.z : 0.2f * a.z; a.w = a.w > 0.f ? a.w : 0.2f * a.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)m * ostride + g * 4) = a;
}
template <int BN, int S>
static hipError_t w2d_split(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  p.ksplit = S; p.part = g_part;
  const hipError_t e = conv_wino2d_launch<BN, 4 | W2D_F_XEPI>(p, st);
  if (e != hipSuccess) return e;
  const unsigned units = (unsigned)p.M * (unsigned)(p.Cout >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((units + 255) / 256), dim3(256), 0, st, p.part, p.bias, p.out, p.M, p.Cout, p.ostride, S, p.leaky);
  return hipGetLastError();
}
#define W2S(NAME, BN, S) {"w2d " NAME, BN, 3, w2d_split<BN, S>}
// chained tiles (W2D_F_CHAIN): CH consecutive pixel tiles per workgroup, NS stages; the same sums as the unchained kernel
template <int BN, int NS, int CH, int FL>
static hipError_t w2d_chain(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  p.chain = CH;
  return conv_wino2d_launch<BN, 4 | W2D_F_CHAIN | FL, NS>(p, st);
}
#define W2C(NAME, BN, NS, CH) {"w2d " NAME, BN, 1, w2d_chain<BN, NS, CH, 0>}
// occupancy experiment: the unchained two-stage tile with EXTRA KB of dynamic LDS it does not use
template <int EXTRA_KB>
static hipError_t w2d_ns2_lds(const ConvParams& p, hipStream_t st) {
  conv_wino2d_debug_extra_lds() = EXTRA_KB * 1024;
  const hipError_t e = conv_wino2d_launch<32, 4, 2>(p, st);
  conv_wino2d_debug_extra_lds() = 0;
  return e;
}
#define W43(NAME, BN, ...) {"w43 " NAME, BN, 2, conv_wino43_launch<__VA_ARGS__>}
static Variant variants[] = {
    W43("q16 4x64 t21 p2", 64, 4, 64, 2, 1, 4 | W43_F_PF2, 16), W43("q16 4x64 n1 p2", 64, 4, 64, 1, 1, 4 | W43_F_PF2, 16, 1), W43("q8 8x64 t21 p2", 64, 8, 64, 2, 1, 4 | W43_F_PF2, 8),
    W43("q8 8x64 n1 p2", 64, 8, 64, 1, 1, 4 | W43_F_PF2, 8, 1), W43("q16 4x32 bg", 32, 4, 32, 1, 1, 4 | W43_F_BG, 16), W43("q8 8x32 bg", 32, 8, 32, 1, 1, 4 | W43_F_BG, 8), W43("q8 8x32 p2", 32, 8, 32, 1, 1, 4 | W43_F_PF2, 8),
    W2N("64 xf", 64, 4 | W2D_F_XFIRST), W2N("32 xf", 32, 4 | W2D_F_XFIRST),
    W2N("64", 64, 4), W2N("32", 32, 4), W2N("64 plain", 64, 0), W2N("32 plain", 32, 0),
    W2S("64 split2", 64, 2), W2S("32 split2", 32, 2), W2S("64 split3", 64, 3), W2S("32 split3", 32, 3), W2S("64 split4", 64, 4), W2S("32 split4", 32, 4),
    W2C("32 ns2 ch1", 32, 2, 1), W2C("32 ns2 ch2", 32, 2, 2), W2C("32 ns2 ch4", 32, 2, 4), W2C("32 ns2 ch8", 32, 2, 8),
    W2C("64 ns3 ch1", 64, 3, 1), W2C("64 ns3 ch2", 64, 3, 2), W2C("64 ns3 ch4", 64, 3, 4), W2C("64 ns3 ch8", 64, 3, 8),
    {"w2d 32 ns2 plain", 32, 1, conv_wino2d_launch<32, 4, 2>},
    {"w2d 32 ns2 xe", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_XEPI, 2>}, {"w2d 32 ns2 sq xe", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_SQ | W2D_F_XEPI, 2>},
    {"w2d 64 ns3 xe", 64, 1, conv_wino2d_launch<64, 4 | W2D_F_XEPI, 3>},
    {"w2d 32 ns2 e1", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_EPI1, 2>}, {"w2d 32 ns2 sq e1", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_SQ | W2D_F_EPI1, 2>},
    {"w2d 64 ns3 e1", 64, 1, conv_wino2d_launch<64, 4 | W2D_F_EPI1, 3>}, {"w2d 64 ns3 sq e1", 64, 1, conv_wino2d_launch<64, 4 | W2D_F_SQ | W2D_F_EPI1, 3>},
    {"w2d 32 ns2 sq", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_SQ, 2>}, {"w2d 32 ns3 sq", 32, 1, conv_wino2d_launch<32, 4 | W2D_F_SQ, 3>}, {"w2d 64 ns3 sq", 64, 1, conv_wino2d_launch<64, 4 | W2D_F_SQ, 3>},
    {"w2d 32 ns2 lds+24", 32, 1, w2d_ns2_lds<24>}, {"w2d 32 ns2 lds+30", 32, 1, w2d_ns2_lds<30>}, {"w2d 32 ns2 lds+32", 32, 1, w2d_ns2_lds<32>},
    W2N("64 time", 64, 4 | W2D_DBG_TIME), W2N("32 time", 32, 4 | W2D_DBG_TIME),
    {"w2d 32 ns2 time", 32, -1, conv_wino2d_launch<32, 4 | W2D_DBG_TIME, 2>},
    {"w2d 32 ns2 ch4 time", 32, -1, w2d_chain<32, 2, 4, W2D_DBG_TIME>}, {"w2d 64 ns3 ch4 time", 64, -1, w2d_chain<64, 3, 4, W2D_DBG_TIME>},
    W2N("64 abl-noxf", 64, 4 | W2D_DBG_NOXF), W2N("32 abl-noxf", 32, 4 | W2D_DBG_NOXF),
    W2N("64 abl-nodma", 64, 4 | W2D_DBG_NODMA), W2N("32 abl-nodma", 32, 4 | W2D_DBG_NODMA),
    W2N("64 abl-nob", 64, 4 | W2D_DBG_NOB), W2N("32 abl-nob", 32, 4 | W2D_DBG_NOB),
    W2N("64 abl-nobar", 64, 4 | W2D_DBG_NOBAR), W2N("32 abl-nobar", 32, 4 | W2D_DBG_NOBAR),
    W2N("64 abl-nord", 64, 4 | W2D_DBG_NORD | W2D_DBG_NOXF), W2N("32 abl-nord", 32, 4 | W2D_DBG_NORD | W2D_DBG_NOXF),
    W2N("64 abl-mfma", 64, 4 | W2D_DBG_NORD | W2D_DBG_NOXF | W2D_DBG_NODMA | W2D_DBG_NOB | W2D_DBG_NOBAR),
    W2N("32 abl-mfma", 32, 4 | W2D_DBG_NORD | W2D_DBG_NOXF | W2D_DBG_NODMA | W2D_DBG_NOB | W2D_DBG_NOBAR),
};

// C2 > 0: the input is two segments, [C - C2 channels of buffer A | C2 channels of buffer B read with the batch halves swapped]
struct Shape { const char* name; int NB, H, W, C, Cout, C2; };
static Shape shapes[] = {
    {"fusion_1_1  4x288x480  528->128", 4, 288, 480, 528, 128, 0},
    {"fusion_3_1  4x72x120  2448->512", 4, 72, 120, 2448, 512, 0},
    {"flow_l3_c0  8x72x120  1920->256", 8, 72, 120, 1920, 256, 0},
    {"fusion_2_1  4x144x240 1168->256", 4, 144, 240, 1168, 256, 0},
    {"flow_l2_c0  8x144x240  896->128", 8, 144, 240, 896, 128, 0},
    {"flow_l1_c0  8x288x480  384->64", 8, 288, 480, 384, 64, 0},
    {"fusion_0_1  4x576x960  208->64", 4, 576, 960, 208, 64, 0},
    {"flow_l0_c0  8x576x960  128->32", 8, 576, 960, 128, 32, 0},
    {"fusion_2_2  4x144x240  256->256", 4, 144, 240, 256, 256, 0},
    {"feat_conv7  8x72x120   512->512", 8, 72, 120, 512, 512, 0},
    {"feat_conv1  8x576x960   64->64", 8, 576, 960, 64, 64, 0},
    {"feat_conv2  8x288x480   64->128", 8, 288, 480, 64, 128, 0},
    {"feat_conv3  8x288x480  128->128", 8, 288, 480, 128, 128, 0},
    {"feat_conv4  8x144x240  128->256", 8, 144, 240, 128, 256, 0},
    {"feat_conv5  8x144x240  256->256", 8, 144, 240, 256, 256, 0},
    {"flow_l0_c1  8x576x960   32->32", 8, 576, 960, 32, 32, 0},
    {"flow_l1_c1  8x288x480   64->64", 8, 288, 480, 64, 64, 0},
    {"flow_l2_c1  8x144x240  128->128", 8, 144, 240, 128, 128, 0},
    {"fusion_0_2  4x576x960   64->64", 4, 576, 960, 64, 64, 0},
    {"fusion_1_2  4x288x480  128->128", 4, 288, 480, 128, 128, 0},
    {"two-seg     4x144x240  (96|48)->64, batch halves swapped in seg 1", 4, 144, 240, 144, 64, 48},
    {"flow_l4_c0  8x36x60   1920->256", 8, 36, 60, 1920, 256, 0},
    {"256:fus_2_1 1x64x64   1168->256", 1, 64, 64, 1168, 256, 0},
    {"two-seg deep 2x72x120 (1040|880)->256, batch halves swapped in seg 1", 2, 72, 120, 1920, 256, 880},
    {"ragged      3x36x60    64->64", 3, 36, 60, 64, 64, 0},
    {"ragged2     2x50x70    (32|16)->32", 2, 50, 70, 48, 32, 16},
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;      // -1: all shapes
  const char* only_variant = argc > 3 ? argv[3] : nullptr;   // substring filter on the variant name
  // tools/bin/w2d_bench reps -2 variants NB H W C Cout: ONE shape from the command line (every 3x3 layer of a plan: tools/w2d_idle_budget.py)
  std::vector<Shape> shape_list(std::begin(shapes), std::end(shapes));
  if (only_shape == -2 && argc > 8) { shape_list.assign(1, Shape{"custom", atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), 0}); }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double* d_sum;
  CK(hipMalloc(&d_sum, sizeof(double)));
  int shape_idx = -1, bad = 0;
  for (const Shape& sh : shape_list) {
    if (++shape_idx != only_shape && only_shape >= 0) continue;
    const size_t M = (size_t)sh.NB * sh.H * sh.W;
    const int C1 = sh.C - sh.C2;
    const int strideA = C1 + 16, strideB = sh.C2 ? sh.C2 + 32 : 0;   // the segments are channel slices of wider buffers
    const size_t n_a = M * strideA, n_b = M * strideB, n_w = (size_t)9 * sh.C * sh.Cout, n_out = M * sh.Cout;
    float *d_a, *d_bb = nullptr, *d_w, *d_w2d, *d_w43, *d_b, *d_out, *d_ref, *d_ref1, *d_md;
    CK(hipMalloc(&d_a, n_a * 4));
    if (n_b) CK(hipMalloc(&d_bb, n_b * 4));
    CK(hipMalloc(&d_w, n_w * 4));
    CK(hipMalloc(&d_w2d, n_w * 4 * 24 / 9 + 64));
    CK(hipMalloc(&d_w43, n_w * 4 * 18 / 9 + 64));
    CK(hipMalloc(&d_ref, n_out * 4));
    CK(hipMalloc(&d_ref1, n_out * 4));
    CK(hipMalloc(&d_md, 4));
    CK(hipMalloc(&g_part, n_out * 4 * 4));
    CK(hipMalloc(&d_b, sh.Cout * 4));
    CK(hipMalloc(&d_out, n_out * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_a, n_a, 1u, 1.0f);
    if (n_b) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_bb, n_b, 7u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_w, n_w, 2u, 0.05f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, d_b, (size_t)sh.Cout, 3u, 0.1f);
    hipLaunchKernelGGL(pack_wino43_kernel, dim3((unsigned)((n_w / 3 + 255) / 256)), dim3(256), 0, st, d_w, d_w43, sh.C, sh.Cout);
    hipLaunchKernelGGL(pack_wino2d_kernel, dim3((unsigned)((n_w / 9 + 255) / 256)), dim3(256), 0, st, d_w, d_w2d, sh.C, sh.Cout);
    CK(hipStreamSynchronize(st));
    unsigned long long* d_tm;
    const size_t n_tm = (size_t)sh.NB * ((sh.H + 7) / 8) * ((sh.W + 31) / 32) * (sh.Cout / 32) * 8;
    CK(hipMalloc(&d_tm, n_tm * 16 + 16384));
    CK(hipMemset(d_tm, 0, 16384));
    ConvParams p{};
    p.part = reinterpret_cast<float*>(d_tm);
    p.nseg = sh.C2 ? 2 : 1;
    p.seg[0].ptr = d_a + 16; p.seg[0].stride = strideA; p.seg[0].C = C1;
    if (sh.C2) { p.seg[1].ptr = d_bb + 16; p.seg[1].stride = strideB; p.seg[1].C = sh.C2; p.seg[1].boff = sh.NB / 2; p.seg[1].bmod = sh.NB; }
    p.ksize = 3; p.w = d_w2d; p.bias = d_b; p.out = d_out; p.ostride = sh.Cout;
    p.NB = sh.NB; p.H = sh.H; p.W = sh.W; p.Cout = sh.Cout; p.Ctot = sh.C; p.leaky = 1; p.M = (int)M;
    const double flops = 2.0 * M * sh.Cout * 9 * sh.C;
    printf("== %d %s  (%.1f GFLOP direct)\n", shape_idx, sh.name, flops * 1e-9);
    bool have_ref[2] = {false, false};
    for (const Variant& v : variants) {
      if (sh.Cout % v.bn) continue;
      if (only_variant) {   // comma-separated substrings: any match
        bool hit = false;
        std::string flt(only_variant);
        for (size_t b = 0; b <= flt.size();) {
          const size_t e = flt.find(',', b) == std::string::npos ? flt.size() : flt.find(',', b);
          if (e > b && (flt[b] == '=' ? flt.substr(b + 1, e - b - 1) == v.name : strstr(v.name, flt.substr(b, e - b).c_str()) != nullptr)) hit = true;   // "=name": exact
          b = e + 1;
        }
        if (!hit) continue;
      }
      CK(hipMemsetAsync(d_out, 0xFF, n_out * 4, st));   // NaN: an unwritten output shows
      p.w = v.fam == 2 ? d_w43 : d_w2d;
      hipError_t le = v.fn(p, st);
      if (le != hipSuccess) { printf("   %-24s  refused (%s)\n", v.name, hipGetErrorString(le)); (void)hipGetLastError(); continue; }
      CK(hipMemsetAsync(d_sum, 0, sizeof(double), st));
      hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, st, d_out, n_out, d_sum);
      double sum = 0;
      CK(hipMemcpyAsync(&sum, d_sum, sizeof(double), hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      if (sum != sum) { printf("   %-24s  NaN in the output (unwritten element?)\n", v.name); ++bad; }
      float maxdiff = 0.f, fam_dist = -1.f;
      auto diff_to = [&](const float* ref) {
        float md = 0.f;
        CK(hipMemsetAsync(d_md, 0, 4, st));
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, st, d_out, ref, n_out, d_md);
        CK(hipMemcpyAsync(&md, d_md, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        return md;
      };
      const int fam = v.fam < 0 ? 1 : v.fam == 3 ? 1 : v.fam;   // (ablations are variants of the y-first loop)
      float* const refs[2] = {d_ref, d_ref1};
      if (v.fam == 3) {
        if (have_ref[1]) fam_dist = diff_to(d_ref1);
      } else if (v.fam == 2) {
        if (have_ref[0]) fam_dist = diff_to(d_ref);
      } else if (v.fam >= 0 && !have_ref[fam]) {
        have_ref[fam] = true;
        CK(hipMemcpyAsync(refs[fam], d_out, n_out * 4, hipMemcpyDeviceToDevice, st));
        CK(hipStreamSynchronize(st));
        if (fam == 1 && have_ref[0]) fam_dist = diff_to(d_ref);
      } else if (have_ref[fam]) maxdiff = diff_to(refs[fam]);
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
      if (strstr(v.name, "time")) {   // per-workgroup phase times (shader clock ticks = 100 MHz s_memtime? printed raw) and the per-CU timeline
        const size_t nwg = (size_t)sh.NB * ((sh.H + 7) / 8) * ((sh.W + 31) / 32) * (sh.Cout / v.bn);
        std::vector<unsigned long long> tm(nwg * 16);
        CK(hipMemcpy(tm.data(), d_tm, nwg * 128, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi = 0, pa = 0, pb = 0, pc = 0, wait = 0, life = 0, cyc = 0, rt = 0, ea = 0, eb = 0, ec = 0, ed = 0, ka = 0, kb = 0;
        // s_memtime counts shader cycles on a counter of the workgroup's own CU group: stamps compare within one CU only.  Per CU (HW_ID bits
        // 8..15 + XCC_ID): its span (first entry .. last exit), the summed lifetimes of its workgroups, how many it ran.
        struct CuAcc { unsigned long long t0 = ~0ull, t1 = 0; double busy = 0; int n = 0; };
        std::map<unsigned long long, CuAcc> cus;
        for (size_t i = 0; i < nwg; ++i) {
          const unsigned long long* e = &tm[i * 16];
          pa += (double)(e[4] - e[0]); pb += (double)(e[5] - e[4]); pc += (double)(e[6] - e[5]);
          pro += (double)(e[1] - e[0]); loop += (double)(e[2] - e[1]); epi += (double)(e[3] - e[2]);
          wait += (double)e[7]; life += (double)(e[3] - e[0]);
          if (e[14]) { ka += (double)(e[14] - e[0]); kb += (double)(e[15] - e[0]); }
          if (e[11]) { ea += (double)(e[11] - e[2]); eb += (double)(e[12] - e[11]); ec += (double)(e[13] - e[12]); ed += (double)(e[3] - e[13]); }
          cyc += (double)(e[3] - e[0]); rt += (double)(e[9] - e[8]);
          CuAcc& c = cus[(e[10] & 0xFF00ull) | (e[10] >> 32 << 16)];
          if (e[0] < c.t0) c.t0 = e[0];
          if (e[3] > c.t1) c.t1 = e[3];
          c.busy += (double)(e[3] - e[0]); ++c.n;
        }
        double span_max = 0, span_avg = 0, busy_avg = 0; int n_min = 1 << 30, n_max = 0;
        for (auto& kv : cus) {
          const double sp = (double)(kv.second.t1 - kv.second.t0);
          span_max = sp > span_max ? sp : span_max; span_avg += sp / cus.size(); busy_avg += kv.second.busy / cus.size();
          n_min = kv.second.n < n_min ? kv.second.n : n_min; n_max = kv.second.n > n_max ? kv.second.n : n_max;
        }
        const double ghz = rt > 0 ? cyc / rt * 0.1 : 0;   // shader cycles per 10 ns tick of s_memrealtime
        printf("   [time] %zu workgroups on %zu CUs (%d..%d each): prologue %.0f (entry -> first DMA issued %.0f, -> stage 0 landed %.0f, -> barrier passed %.0f, -> fragments of chunk 0 ready)  K loop %.0f (of it s_waitcnt + barrier %.0f)  epilogue %.0f cycles (averages); span per CU max %.0f avg %.0f cycles, clock %.3f GHz\n", nwg, cus.size(), n_min, n_max,
               pro / nwg, pa / nwg, pb / nwg, pc / nwg, loop / nwg, wait / nwg, epi / nwg, span_max, span_avg, ghz);
        if (ka > 0) printf("   [time] prologue: kernel arguments in registers %.0f cycles after entry, slot-table entries (and the bounds tests) %.0f\n", ka / nwg, kb / nwg);
        if (ea > 0) printf("   [time] epilogue: last MFMA -> entry barrier passed %.0f, -> round 0 published (x inverse, first writes, barrier) %.0f, -> round 3 published %.0f, -> exit %.0f cycles\n",
                           ea / nwg, eb / nwg, ec / nwg, ed / nwg);
        // one machine-readable line per timed launch (tools/w2d_idle_budget.py): per-workgroup averages in shader cycles, per-CU span / summed
        // workgroup lifetimes, the clock, and the event time of this variant's fastest repetition
        printf("   [time-json] {\"variant\": \"%s\", \"shape\": \"%s\", \"NB\": %d, \"H\": %d, \"W\": %d, \"C\": %d, \"Cout\": %d, \"bn\": %d, \"nwg\": %zu, \"prologue\": %.1f, \"loop\": %.1f, \"wait\": %.1f, "
               "\"epilogue\": %.1f, \"life\": %.1f, \"ncu\": %zu, \"span_max\": %.0f, \"span_avg\": %.0f, \"cu_busy_avg\": %.0f, \"wg_per_cu_min\": %d, \"wg_per_cu_max\": %d, \"ghz\": %.4f, \"ms\": %.5f}\n",
               v.name, sh.name, sh.NB, sh.H, sh.W, sh.C, sh.Cout, v.bn, nwg, pro / nwg, loop / nwg, wait / nwg, epi / nwg, life / nwg, cus.size(), span_max, span_avg, busy_avg, n_min, n_max, ghz, best);
        // busy share of one CU slot: follow the workgroups that ran on the CU of workgroup 0 (same hw_id CU/SE bits and XCC)
      }
      const bool abl = v.fam < 0;
      if (!abl && maxdiff != 0.f) ++bad;
      if (v.fam == 3 && have_ref[1] && !(fam_dist < 1e-2f)) ++bad;   // a split-K result far from the unsplit one (wrong range / cursor)
      printf("   %-24s  min %8.3f ms  avg %8.3f ms  %7.1f TF/s direct-eq  %s max|d| %.2e", v.name, best, tot / reps, flops / best * 1e-9,
             abl ? "(ablation)" : v.fam == 3 ? "split-K family" : v.fam == 2 ? "1-D family" : maxdiff == 0.f ? (fam ? "bit-identical (family 1)" : "bit-identical (family 0)") : "MISMATCH", maxdiff);
      if (fam_dist >= 0.f) printf("   [vs family %d: max|d| %.2e, checksum %.6e]", v.fam == 3 ? 1 : 0, fam_dist, sum);
      printf("\n");
      fflush(stdout);
    }
    CK(hipFree(d_tm)); CK(hipFree(g_part)); g_part = nullptr;
    CK(hipFree(d_a)); if (d_bb) CK(hipFree(d_bb)); CK(hipFree(d_w)); CK(hipFree(d_w2d)); CK(hipFree(d_w43)); CK(hipFree(d_ref)); CK(hipFree(d_ref1)); CK(hipFree(d_md)); CK(hipFree(d_b)); CK(hipFree(d_out));
  }
  printf("mismatches: %d\n", bad);
  return bad ? 1 : 0;
}
