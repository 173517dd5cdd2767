// warp_bench.hip -- warp_vec_kernel on the shapes of a 1080p 2x2-tiled forward, with and without the sixteen miscellaneous channels of
// an aligned level (development tool).  Smooth synthetic flows (like the benchmark pair's: ~95 % of the neighbours share corners).
// (Commit 709bc1e carried a `variant` knob in WarpParams and compared the modes that were tried - corners from the row above, from the
// neighbouring lane, the miscellaneous channels as a 16-lane slice: profiles/r04_warp_bench_modes.log.)
//   hipcc --offload-arch=gfx950 -O2 -Iframe-interpolation_amd/csrc tools/warp_bench.hip frame-interpolation_amd/csrc/build/misc_kernels.o -o tools/bin/warp_bench
#include "film_kernels.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int NB, H, W, C; };

int main() {
  const Shape shapes[] = {{"level 0", 4, 576, 960, 64}, {"level 1", 4, 288, 480, 192}, {"level 2", 4, 144, 240, 448}, {"level 3", 4, 72, 120, 960}};
  for (const Shape& sh : shapes) {
    const int64_t npix = (int64_t)sh.NB * sh.H * sh.W;
    const int dstride = 2 * sh.C + 16;
    std::vector<float> hsrc(npix * sh.C), hflow(npix * 2), hflow2(npix * 2), himg(2 * npix * 3);
    unsigned r = 12345u;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (float)(r >> 8) * (1.f / 16777216.f); };
    for (float& v : hsrc) v = rnd();
    for (float& v : himg) v = rnd();
    for (int64_t i = 0; i < npix; ++i) {
      const int x = (int)(i % sh.W), y = (int)((i / sh.W) % sh.H);
      hflow[2 * i] = 2.3f + sinf(x * 0.011f) * cosf(y * 0.013f) + 0.02f * rnd();
      hflow[2 * i + 1] = -1.7f + cosf(x * 0.009f) * sinf(y * 0.012f) + 0.02f * rnd();
      hflow2[2 * i] = -hflow[2 * i] + 0.1f * rnd();
      hflow2[2 * i + 1] = -hflow[2 * i + 1] + 0.1f * rnd();
    }
    float *src, *flow, *flow2, *img, *dst, *ref;
    CK(hipMalloc(&src, hsrc.size() * 4)); CK(hipMalloc(&flow, hflow.size() * 4)); CK(hipMalloc(&flow2, hflow.size() * 4));
    CK(hipMalloc(&img, himg.size() * 4)); CK(hipMalloc(&dst, npix * dstride * 4)); CK(hipMalloc(&ref, npix * dstride * 4));
    CK(hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(flow, hflow.data(), hflow.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(flow2, hflow2.data(), hflow2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int misc = 0; misc < 2; ++misc) {
      std::vector<float> want;
      for (int mode : {0}) {
        WarpParams p{};
        p.src = src; p.sstride = sh.C; p.C = sh.C; p.flow = flow; p.fscale = 0.5f; p.dst = dst + sh.C; p.dstride = dstride;
        p.NB = sh.NB; p.H = sh.H; p.W = sh.W;
        if (misc) { p.src3 = img; p.src3b = img + npix * 3; p.s3stride = 3; p.dst3 = dst + 2 * sh.C; p.d3stride = dstride; p.pack_b = flow2; p.pack_f = flow; }
        CK(hipMemset(dst, 0, npix * dstride * 4));
        for (int i = 0; i < 5; ++i) CK(film_launch_warp(p, nullptr));
        CK(hipEventRecord(e0, nullptr));
        const int iters = 20;
        for (int i = 0; i < iters; ++i) CK(film_launch_warp(p, nullptr));
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        std::vector<float> got(npix * dstride);
        CK(hipMemcpy(got.data(), dst, got.size() * 4, hipMemcpyDeviceToHost));
        if (want.empty()) want = got;
        const bool same = memcmp(want.data(), got.data(), got.size() * 4) == 0;
        const double bytes = 4.0 * npix * (2.0 * sh.C + 2 + (misc ? 12 : 0));
        (void)mode; (void)same;
        printf("%s  %dx%dx%dx%d  %s : %.4f ms  %.2f TB/s\n", sh.name, sh.NB, sh.H, sh.W, sh.C, misc ? "features + misc16" : "features only    ", ms, bytes / ms * 1e-9);
      }
    }
    CK(hipFree(src)); CK(hipFree(flow)); CK(hipFree(flow2)); CK(hipFree(img)); CK(hipFree(dst)); CK(hipFree(ref));
  }
  return 0;
}
