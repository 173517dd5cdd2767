// fold4_bench.hip -- micro-benchmark + check of conv_fold4_kernel (upsample + 2x2 conv in its difference form, conv_fold4_impl.h)
// against the sub-pixel fold on conv_buf_kernel (fold = 2: the kernel these layers ran on until round 6) on the four decoder layers of
// a 1080p 2x2-tiled forward, plus ragged levels.  The two are different summation families: the distance is printed and bounded.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/fold4_bench.hip -o tools/bin/fold4_bench
//   tools/bin/fold4_bench [reps] [shape index | -1]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../frame-interpolation_amd/csrc/conv_fold4_impl.h"

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

// HWIO [2][2][C][N] -> the four phase-summed K-major copies of fold = 2 (film_layers.cpp): phase (py, px) [N][ntaps * C], taps (a, b) raster
__global__ void pack_fold2_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)C * N) return;
  const int n = (int)(i % N), c = (int)(i / N);
  float* df = dst;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int nt = (py + 1) * (px + 1);
      int t = 0;
      for (int a = 0; a <= py; ++a)
        for (int b = 0; b <= px; ++b, ++t) {
          float acc = 0.f;
          for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx)
              if ((py & dy) == a && (px & dx) == b) acc += src[((size_t)(dy * 2 + dx) * C + c) * N + n];
          df[(size_t)n * nt * C + (size_t)t * C + c] = acc;
        }
      df += (size_t)nt * C * N;
    }
}

// HWIO [2][2][C][N] -> [N / 32][chunk8][plane 4][K half][32][4] (conv_fold4_impl.h): S, Sx, Sy, W11
__global__ void pack_fold4_kernel(const float* src, float* dst, int C, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)C * N) return;
  const int n = (int)(i % N), c = (int)(i / N);
  const float w00 = src[((size_t)0 * C + c) * N + n], w01 = src[((size_t)1 * C + c) * N + n], w10 = src[((size_t)2 * C + c) * N + n],
              w11 = src[((size_t)3 * C + c) * N + n];
  const float pl[4] = {((w00 + w01) + w10) + w11, w01 + w11, w10 + w11, w11};
  for (int q = 0; q < 4; ++q)
    dst[((((size_t)(n / 32) * (C / 8) + c / 8) * 4 + q) * 2 + (c % 8) / 4) * 128 + (n % 32) * 4 + c % 4] = pl[q];
}

// compares the channel slice [c0, c0 + C) of every pixel (the rest of the buffer is the 0xFF fill: must stay untouched - counted too)
__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out, float* amax, int ostride, int c0, int C) {
  float m = 0.f, am = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % ostride);
    if (ch < c0 || ch >= c0 + C) {   // outside the slice: both must still hold the fill pattern
      if (__float_as_uint(a[i]) != 0xFFFFFFFFu) m = 1e30f;
      continue;
    }
    const float d = fabsf(a[i] - b[i]);
    m = fmaxf(m, d == d ? d : 1e30f);
    am = fmaxf(am, fabsf(b[i]));
  }
  atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
  atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(am));
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                            int M, int Cout, int ostride, int S, int leaky) {   // the engine's conv_splitk_reduce_kernel, restated
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
  if (leaky) { a.x = a.x > 0.f ? a.x : 0.2f * a.x; a.y = a.y > 0.f ? a.y : 0.2f * a.y; a.z = a.z > 0.f ? a.z : 0.2f * a.z; a.w = a.w > 0.f ? a.w : 0.2f * a.w; }
  *reinterpret_cast<float4*>(out + (size_t)m * ostride + g * 4) = a;
}
static float* g_part = nullptr;
template <int NCT, int S>
static hipError_t fold4_split(const ConvParams& p0, hipStream_t st) {
  ConvParams p = p0;
  p.ksplit = S; p.part = g_part;
  const hipError_t e = conv_fold4_launch<NCT, 4>(p, st);
  if (e != hipSuccess) return e;
  const unsigned units = (unsigned)p.M * 4u * (unsigned)(p.Cout >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((units + 255) / 256), dim3(256), 0, st, p.part, p.bias, p.out, p.M * 4, p.Cout, p.ostride, S, p.leaky);
  return hipGetLastError();
}
typedef hipError_t (*LaunchFn)(const ConvParams&, hipStream_t);
struct Variant { const char* name; int bn; int fold; LaunchFn fn; };
static Variant variants[] = {
    {"buf 128x128 fold2", 128, 2, conv_buf_launch<128, 128, 2, 2, 4>},
    {"buf 256x64 fold2", 64, 2, conv_buf_launch<256, 64, 4, 1, 4>},
    {"buf 128x64 fold2", 64, 2, conv_buf_launch<128, 64, 2, 2, 4>},
    {"fold4 x64", 64, 3, conv_fold4_launch<2, 4>},
    {"fold4 x64 plain", 64, 3, conv_fold4_launch<2, 0>},
    {"fold4 x32", 32, 3, conv_fold4_launch<1, 4>},
    {"fold4 x64 time", 64, 3, conv_fold4_launch<2, 4 | F4_DBG_TIME>},
    {"fold4 x32 split2", 32, 3, fold4_split<1, 2>}, {"fold4 x64 split2", 64, 3, fold4_split<2, 2>},
    {"fold4 x32 split3", 32, 3, fold4_split<1, 3>}, {"fold4 x64 split4", 64, 3, fold4_split<2, 4>}, {"fold4 x32 split4", 32, 3, fold4_split<1, 4>},
};

struct Shape { const char* name; int NB, H, W, C, Cout; };   // H, W: the LOW-resolution input grid
static Shape shapes[] = {
    {"fusion_3_0  4x36x60   1936->512", 4, 36, 60, 1936, 512},
    {"fusion_2_0  4x72x120   512->256", 4, 72, 120, 512, 256},
    {"fusion_1_0  4x144x240  256->128", 4, 144, 240, 256, 128},
    {"fusion_0_0  4x288x480  128->64", 4, 288, 480, 128, 64},
    {"256:fus_3_0 1x16x16   1936->512", 1, 16, 16, 1936, 512},
    {"vimeo:fus_3_0 8x16x28 1936->512", 8, 16, 28, 1936, 512},
    {"ragged      3x7x45      48->64", 3, 7, 45, 48, 64},
    {"tiny K      2x4x32      16->64", 2, 4, 32, 16, 64},
    {"K = 32      2x9x33      32->128", 2, 9, 33, 32, 128},
};

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int shape_idx = -1, bad = 0;
  for (const Shape& sh : shapes) {
    if (++shape_idx != only_shape && only_shape >= 0) continue;
    const size_t Min = (size_t)sh.NB * sh.H * sh.W, Mout = Min * 4;
    const int strideA = sh.C + 16, ostride = sh.Cout + 32;   // channel slices of wider buffers
    const size_t n_a = Min * strideA, n_w = (size_t)4 * sh.C * sh.Cout, n_out = Mout * ostride;
    float *d_a, *d_w, *d_w2, *d_w4a, *d_w4b, *d_b, *d_out, *d_ref, *d_md;
    CK(hipMalloc(&d_a, n_a * 4)); CK(hipMalloc(&d_w, n_w * 4)); CK(hipMalloc(&d_w2, (size_t)9 * sh.C * sh.Cout * 4));
    CK(hipMalloc(&d_w4a, n_w * 4)); CK(hipMalloc(&d_w4b, n_w * 4)); CK(hipMalloc(&d_b, sh.Cout * 4));
    CK(hipMalloc(&d_out, n_out * 4)); CK(hipMalloc(&d_ref, n_out * 4)); CK(hipMalloc(&d_md, 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_a, n_a, 1u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, d_w, n_w, 2u, 0.05f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, d_b, (size_t)sh.Cout, 3u, 0.1f);
    const unsigned pb = (unsigned)(((size_t)sh.C * sh.Cout + 255) / 256);
    hipLaunchKernelGGL(pack_fold2_kernel, dim3(pb), dim3(256), 0, st, d_w, d_w2, sh.C, sh.Cout);
    hipLaunchKernelGGL(pack_fold4_kernel, dim3(pb), dim3(256), 0, st, d_w, d_w4a, sh.C, sh.Cout);
    CK(hipStreamSynchronize(st));
    unsigned long long* d_tm;
    const size_t n_tm = (size_t)sh.NB * ((sh.H + 3) / 4) * ((sh.W + 31) / 32) * (sh.Cout / 32) * 16;
    CK(hipMalloc(&d_tm, n_tm * 8 + 16384));
    CK(hipMalloc(&g_part, (size_t)4 * Mout * sh.Cout * 4));
    ConvParams p{};
    p.part = reinterpret_cast<float*>(d_tm);
    p.nseg = 1;
    p.seg[0].ptr = d_a + 16; p.seg[0].stride = strideA; p.seg[0].C = sh.C;
    p.ksize = 2; p.bias = d_b; p.out = d_out + 32; p.ostride = ostride;
    p.NB = sh.NB; p.H = sh.H; p.W = sh.W; p.Cout = sh.Cout; p.Ctot = sh.C; p.leaky = 0; p.M = (int)Min;
    long long rel = 0;
    for (int q = 0; q < 4; ++q) { p.fold_woff[q] = rel; rel += (long long)((q >> 1) + 1) * ((q & 1) + 1) * sh.C * sh.Cout; }
    const double flops = 2.0 * Mout * sh.Cout * 4 * sh.C;   // the reference op's
    printf("== %d %s  (%.1f GFLOP of the reference op)\n", shape_idx, sh.name, flops * 1e-9);
    bool have_ref = false;
    for (const Variant& v : variants) {
      if (sh.Cout % v.bn) continue;
      p.fold = v.fold;
      p.w = v.fold == 2 ? d_w2 : d_w4a;
      CK(hipMemsetAsync(d_out, 0xFF, n_out * 4, st));
      hipError_t le = v.fn(p, st);
      if (le != hipSuccess) { printf("   %-22s  refused (%s)\n", v.name, hipGetErrorString(le)); (void)hipGetLastError(); continue; }
      CK(hipStreamSynchronize(st));
      float md[2] = {0.f, 0.f};
      if (!have_ref) {
        have_ref = true;
        CK(hipMemcpy(d_ref, d_out, n_out * 4, hipMemcpyDeviceToDevice));
      } else {
        CK(hipMemsetAsync(d_md, 0, 8, st));
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, st, d_out, d_ref, n_out, d_md, d_md + 1, ostride, 32, sh.Cout);   // (the NaN gaps between the slices compare as NaN != NaN -> 1e30: masked below)
        CK(hipMemcpyAsync(md, d_md, 8, hipMemcpyDeviceToHost, st));
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
      if (strstr(v.name, "time")) {
        const size_t nwg = (size_t)sh.NB * ((sh.H + 3) / 4) * ((sh.W + 31) / 32) * (sh.Cout / v.bn);
        std::vector<unsigned long long> tm(nwg * 16);
        CK(hipMemcpy(tm.data(), d_tm, nwg * 128, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi = 0, wait = 0, cyc = 0, rt = 0;
        for (size_t i = 0; i < nwg; ++i) {
          const unsigned long long* e = &tm[i * 16];
          pro += (double)(e[1] - e[0]); loop += (double)(e[2] - e[1]); epi += (double)(e[3] - e[2]); wait += (double)e[7];
          cyc += (double)(e[3] - e[0]); rt += (double)(e[9] - e[8]);
        }
        const double mf = 128.0 * (sh.C / 8) * 16 * (v.bn / 32);
        printf("   [time] %zu workgroups: prologue %.0f  K loop %.0f (s_waitcnt + barrier %.0f; fair share of the MFMAs alone %.0f)  epilogue %.0f cycles; clock %.3f GHz\n", nwg,
               pro / nwg, loop / nwg, wait / nwg, mf, epi / nwg, rt > 0 ? cyc / rt * 0.1 : 0.0);
      }
      const bool same_family = have_ref && md[0] == 0.f;
      const float tol = 2e-5f * fmaxf(md[1], 1.f);
      const bool ok = md[0] <= tol;
      if (!ok) ++bad;
      printf("   %-22s  min %8.3f ms  avg %8.3f ms  %7.1f TF/s of the reference op   max|d| vs first %.2e (max|ref| %.2e)%s\n", v.name, best, tot / reps, flops / best * 1e-9,
             md[0], md[1], ok ? (same_family ? "" : "  ok") : "  MISMATCH");
      fflush(stdout);
    }
    CK(hipFree(g_part)); g_part = nullptr;
    CK(hipFree(d_tm)); CK(hipFree(d_a)); CK(hipFree(d_w)); CK(hipFree(d_w2)); CK(hipFree(d_w4a)); CK(hipFree(d_w4b)); CK(hipFree(d_b)); CK(hipFree(d_out)); CK(hipFree(d_ref)); CK(hipFree(d_md));
  }
  printf("mismatches: %d\n", bad);
  return bad ? 1 : 0;
}
