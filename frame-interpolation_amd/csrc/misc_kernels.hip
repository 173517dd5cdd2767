// misc_kernels.hip -- the HBM-bound / small kernels of the FILM hot path (gfx950, wave64).
//
//   flow_head    fused 1x1 (Cin->16, leaky) + 1x1 (16->2) flow head  pyramid_flow_estimator.py:77-83
//   conv_pw      1x1 Conv2D with Cout <= 16 (flow / RGB heads)   pyramid_flow_estimator.py:77-83, fusion.py:100-101
//   pool2x2      AveragePooling2D(2,2,'valid')                   util.py:39-44, feature_extractor.py:138-146
//   flow_up      tf.image.resize(2*v) bilinear x2                pyramid_flow_estimator.py:155, util.py:113
//   flow_add     v = residual + v                                pyramid_flow_estimator.py:161, util.py:114
//   warp         util.warp -> tfa.image.dense_image_warp         util.py:48-82
//   pack_flow    0.5*flow into the aligned pyramid               interpolator.py:163-165,182-183
//
// This file is compiled with -ffp-contract=off: the reference evaluates every lerp as separate
// multiply / add tensor ops (one rounding each), so a*b+c must not become an fma here.
#include "film_kernels.h"

namespace {

__device__ __forceinline__ float leaky02(float v) { return v > 0.f ? v : 0.2f * v; }

// ------------------------------------------------------------------------------------------------
// conv_pw: thread = pixel, all COUT (<= 16) outputs in registers; weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_pw_kernel(ConvPwParams p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [Cin][COUT]
  const int nw = p.Cin * COUT;
  for (int i = threadIdx.x; i < nw; i += 256) wsm[i] = p.w[i];
  __syncthreads();
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const float* src = p.in + m * p.istride;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  for (int c = 0; c < p.Cin; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + c);
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int o = 0; o < COUT; ++o) acc[o] = __builtin_fmaf(vv[k], wsm[(c + k) * COUT + o], acc[o]);
    }
  }
  float* dst = p.out + m * p.ostride;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    float v = acc[o] + p.bias[o];
    if (p.leaky) v = leaky02(v);
    dst[o] = v;
    if (COUT == 2 && p.add != nullptr) p.sum[m * 2 + o] = v + p.add[m * 2 + o];   // v = residual + upsampled flow
  }
}

// ------------------------------------------------------------------------------------------------
// flow_head: thread = pixel; hidden = leaky(W3^T x + b3) (16 values, registers), out = W4^T hidden + b4.
// Same fma order as the two separate 1x1 convolutions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_head_kernel(FlowHeadParams p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [Cin][16] then [16][2]
  const int n3 = p.Cin * 16;
  for (int i = threadIdx.x; i < n3; i += 256) wsm[i] = p.w3[i];
  if (threadIdx.x < 32) wsm[n3 + threadIdx.x] = p.w4[threadIdx.x];
  __syncthreads();
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const float* src = p.in + m * p.istride;
  float h[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) h[o] = 0.f;
  // 32 input channels at a time: the eight 16-byte loads of a pixel are requested together (one memory latency per
  // 32 channels instead of one per 4; this kernel is latency bound: 544 fmas per 136 bytes)
  for (int c0 = 0; c0 < p.Cin; c0 += 32) {
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = c0 + 4 * q < p.Cin ? *reinterpret_cast<const float4*>(src + c0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (c0 + 4 * q >= p.Cin) break;
      const float vv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 0; o < 16; ++o) h[o] = __builtin_fmaf(vv[k], wsm[(c0 + 4 * q + k) * 16 + o], h[o]);
      }
    }
  }
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    const float hv = leaky02(h[o] + p.b3[o]);
    o0 = __builtin_fmaf(hv, wsm[n3 + o * 2], o0);
    o1 = __builtin_fmaf(hv, wsm[n3 + o * 2 + 1], o1);
  }
  const float2 r = make_float2(o0 + p.b4[0], o1 + p.b4[1]);
  reinterpret_cast<float2*>(p.out)[m] = r;
  if (p.add != nullptr) {   // v = residual + upsampled flow (pyramid_flow_estimator.py:161)
    const float2 u = reinterpret_cast<const float2*>(p.add)[m];
    reinterpret_cast<float2*>(p.sum)[m] = make_float2(r.x + u.x, r.y + u.y);
  }
}

// ------------------------------------------------------------------------------------------------
// Row-wise kernels (pool, warp): grid = (units of a row / 256, rows, images), so that the only per-thread division is
// "unit of the row -> (pixel x, channel group g)": a float multiply by 1/G and one correction step (exact for
// units < 2^24; the launchers check) instead of three 64-bit divisions by run-time values, which made these
// kernels instruction bound (warp_vec_kernel: 640 instructions for four 16-byte loads and one store).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_unit(unsigned i, int G, int& x, int& g) {
  const float rg = 1.0f / (float)G;
  x = (int)((float)i * rg);   // off by at most one
  int r = (int)i - x * G;
  if (r < 0) { --x; r += G; }
  else if (r >= G) { ++x; r -= G; }
  g = r;
}

// ------------------------------------------------------------------------------------------------
// pool2x2: thread = (output pixel, float4 channel group) or (output pixel) for C == 3.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_vec_kernel(PoolParams p) {
  const int G = p.C >> 2;
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)(Wo * G)) return;
  int x, g;
  split_unit(i, G, x, g);
  const int y = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t pix = (b * Ho + y) * Wo + x;
  const float* s00 = p.in + ((b * p.H + 2 * y) * p.W + 2 * x) * p.istride + g * 4;
  const float4 a = *reinterpret_cast<const float4*>(s00);
  const float4 bq = *reinterpret_cast<const float4*>(s00 + p.istride);
  const float4 c = *reinterpret_cast<const float4*>(s00 + (int64_t)p.W * p.istride);
  const float4 d = *reinterpret_cast<const float4*>(s00 + (int64_t)p.W * p.istride + p.istride);
  float4 r;
  r.x = (((a.x + bq.x) + c.x) + d.x) * 0.25f;
  r.y = (((a.y + bq.y) + c.y) + d.y) * 0.25f;
  r.z = (((a.z + bq.z) + c.z) + d.z) * 0.25f;
  r.w = (((a.w + bq.w) + c.w) + d.w) * 0.25f;
  *reinterpret_cast<float4*>(p.out + pix * p.ostride + g * 4) = r;
}

__global__ __launch_bounds__(256) void pool_c3_kernel(PoolParams p) {
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  if (x >= Wo) return;
  const int y = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t pix = (b * Ho + y) * Wo + x;
  const float* s00 = p.in + ((b * p.H + 2 * y) * p.W + 2 * x) * p.istride;
  const float* s10 = s00 + (int64_t)p.W * p.istride;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    p.out[pix * p.ostride + c] = (((s00[c] + s00[p.istride + c]) + s10[c]) + s10[p.istride + c]) * 0.25f;
}

// ------------------------------------------------------------------------------------------------
// flow_up: out = bilinear_x2(2 * in), TF2 half-pixel-centre rule.
//   src = (o + 0.5) * 0.5 - 0.5; lo = max(floor(src), 0); hi = min(ceil(src), n - 1); t = src - floor(src)
//   top = tl + (tr - tl) * tx; bot = bl + (br - bl) * tx; out = top + (bot - top) * ty
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void axis_x2(int o, int n, int& lo, int& hi, float& t) {
  const float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  const float f = floorf(src);
  lo = max((int)f, 0);
  hi = min((int)ceilf(src), n - 1);
  t = src - f;
}

__global__ __launch_bounds__(256) void flow_up_kernel(FlowUpParams p) {
  const int Ho = p.h * 2, Wo = p.w * 2;
  const int64_t total = (int64_t)p.NB * Ho * Wo;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int x = (int)(pix % Wo);
  const int64_t t2 = pix / Wo;
  const int y = (int)(t2 % Ho);
  const int64_t b = t2 / Ho;
  int ylo, yhi, xlo, xhi;
  float ty, tx;
  axis_x2(y, p.h, ylo, yhi, ty);
  axis_x2(x, p.w, xlo, xhi, tx);
  const float2* img = reinterpret_cast<const float2*>(p.in) + b * p.h * p.w;
  float2 tl = img[(int64_t)ylo * p.w + xlo], tr = img[(int64_t)ylo * p.w + xhi];
  float2 bl = img[(int64_t)yhi * p.w + xlo], br = img[(int64_t)yhi * p.w + xhi];
  float2 o;
  {
    const float a = 2.f * tl.x, bq = 2.f * tr.x, c = 2.f * bl.x, d = 2.f * br.x;
    const float top = a + (bq - a) * tx, bot = c + (d - c) * tx;
    o.x = top + (bot - top) * ty;
  }
  {
    const float a = 2.f * tl.y, bq = 2.f * tr.y, c = 2.f * bl.y, d = 2.f * br.y;
    const float top = a + (bq - a) * tx, bot = c + (d - c) * tx;
    o.y = top + (bot - top) * ty;
  }
  reinterpret_cast<float2*>(p.out)[pix] = o;
}

__global__ __launch_bounds__(256) void flow_add_kernel(FlowAddParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  p.out[i] = p.a[i] + p.b[i];
}

// ------------------------------------------------------------------------------------------------
// warp: TFA dense_image_warp / interpolate_bilinear semantics, per axis
//   q = coord + s*flow; f = min(max(floor(q), 0), size-2); alpha = clip(q - f, 0, 1)
//   top = ax*(tr - tl) + tl; bot = ax*(br - bl) + bl; out = ay*(bot - top) + top
// thread = (pixel, float4 channel group): lanes of a wave cover consecutive channel groups of the
// same pixel, so the four corner reads and the store are contiguous 16-B-per-lane accesses.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void warp_axis(float q, int size, int& f, float& alpha) {
  float fl = floorf(q);
  fl = fminf(fmaxf(fl, 0.f), (float)(size - 2));
  f = (int)fl;
  alpha = fminf(fmaxf(q - fl, 0.f), 1.f);
}

__device__ __forceinline__ float lerp3(float tl, float tr, float bl, float br, float ax, float ay) {
  const float top = ax * (tr - tl) + tl;
  const float bot = ax * (br - bl) + bl;
  return ay * (bot - top) + top;
}

// Flow of output pixel (y, x) of image b.  Either read from p.flow, or - fused resize - computed from the coarser
// level's flow p.coarse [NB][H/2][W/2][2] exactly as flow_up_kernel does (bilinear x2 of 2*v, half-pixel centres);
// the caller stores it to p.flow_out once per pixel.
__device__ __forceinline__ float2 warp_flow_at(const WarpParams& p, int64_t b, int y, int x) {
  if (p.coarse == nullptr) return reinterpret_cast<const float2*>(p.flow)[(b * p.H + y) * p.W + x];
  const int h = p.H >> 1, w = p.W >> 1;
  int ylo, yhi, xlo, xhi;
  float ty, tx;
  axis_x2(y, h, ylo, yhi, ty);
  axis_x2(x, w, xlo, xhi, tx);
  const float2* img = reinterpret_cast<const float2*>(p.coarse) + b * h * w;
  const float2 tl = img[(int64_t)ylo * w + xlo], tr = img[(int64_t)ylo * w + xhi];
  const float2 bl = img[(int64_t)yhi * w + xlo], br = img[(int64_t)yhi * w + xhi];
  float2 o;
  {
    const float a = 2.f * tl.x, bq = 2.f * tr.x, c = 2.f * bl.x, d = 2.f * br.x;
    const float top = a + (bq - a) * tx, bot = c + (d - c) * tx;
    o.x = top + (bot - top) * ty;
  }
  {
    const float a = 2.f * tl.y, bq = 2.f * tr.y, c = 2.f * bl.y, d = 2.f * br.y;
    const float top = a + (bq - a) * tx, bot = c + (d - c) * tx;
    o.y = top + (bot - top) * ty;
  }
  return o;
}

// grid = (feature workgroups of a row band + misc workgroups, row bands of WARP_ROWS, images).
// Feature workgroup = 16 consecutive pixels x ONE 64-channel slice (16 float4 groups) x the 8 rows of the band; a thread owns
// (x, channel group g) for the 8 rows, four rows in flight at a time: the flows of all eight rows first (the second half's flow
// round trip overlaps the first half's corner loads), then per half the corner loads, the lerps and four stores.
//  * Rows of a band share source rows: for smooth flows (real frames; the benchmark pair: 92-97 % of the neighbours) the top
//    corners of output row y + 1 are the bottom corners of row y.  A thread keeps the bottom corners it loaded and skips the top
//    loads of the next row when the source pixel index says they are the same pixel (any flow field gives the same values, so
//    the same bits; tests/test_gpu_parity.py::test_warp_corner_sharing_is_bit_exact): 10 instead of 16 loads per half, +3-4 %
//    (tools/warp_bench.hip, profiles/r04_warp_bench_modes.log).  Taking the right corners from the lane of pixel x + 1
//    (ds_bpermute) on top of that did not pay (mode 3 there): the L2 request count is not what bounds this kernel.
//  * 16 pixels x 64 channels per workgroup whatever the channel count: the corner pixels neighbouring outputs share are read by
//    one workgroup (round 2 mapped 256 consecutive (pixel, group) units of a row to a workgroup, i.e. one pixel of a 960-channel
//    level: read over-fetch 1.57x at the fabric against 1.32x).
// Misc workgroups (behind the feature workgroups of the band; only the second t = 0.5 warp of a level has them): the sixteen
// miscellaneous channels of an aligned-pyramid level, [warp(img0) 3 | warp(img1) 3 | 0.5 bflow 2 | 0.5 fflow 2 | 0 x 6].  One thread =
// one pixel of one row of the band: 2 flows, 8 three-float corner loads (warp_c3_kernel's arithmetic per channel,
// pack_flow_kernel's for the flows), then the 64 bytes of a pixel go through LDS so that four consecutive lanes store the four
// 16-byte quarters of one pixel - full 64-byte lines.  Rounds 2-3 wrote these channels with 3 + 3 + 10 scalar stores per pixel at the
// pixel pitch of the level (576 B ... 7.7 KB) from two launches, i.e. 16 four-byte L2 write requests per pixel: the t = 0.5 warps
// ran at 2.8-3.0 TB/s against 5.2 TB/s for the flow-estimator warps of the same size (profiles/r03_per_op_profile.json).  A 16-lane
// "channel slice" (lane c = channel c) was tried too: as many memory instructions as a full 64-channel slice for 1/8 of the bytes,
// +0.31 ms on a 0.28 ms launch.
constexpr int WARP_ROWS = 8;
constexpr int WARP_TX = 16;    // pixels of a feature workgroup
constexpr int WARP_BATCH = 4;  // rows whose corner loads are in flight together (8: 160 registers, three waves per SIMD - 2 % slower).
                               // 106 registers = four waves per SIMD; forcing five (96, as before the corner reuse) spills 16
__global__ __launch_bounds__(256) void warp_vec_kernel(WarpParams p) {
  const int G = p.C >> 2;
  const unsigned nsl = (unsigned)(G + 15) >> 4;
  const unsigned nfeat_blocks = ((unsigned)(p.W + WARP_TX - 1) / WARP_TX) * nsl;
  const int yb = blockIdx.y * WARP_ROWS;
  const int64_t b = blockIdx.z;
  if (blockIdx.x >= nfeat_blocks) {
    __shared__ float4 stage[256 * 4 + 64];   // [pixel][quarter], one float4 of padding per 16 pixels
    if (p.misc_nb > 0 && b >= p.misc_nb) return;   // (a pair launch: the second half of the batches has no miscellaneous channels)
    const unsigned u0 = (blockIdx.x - nfeat_blocks) * 256u;
    const unsigned u = u0 + threadIdx.x;
    const unsigned band = (unsigned)(p.W * WARP_ROWS);
    int k = 0, x = 0;
    split_unit(min(u, band - 1u), p.W, k, x);
    const int y = min(yb + k, p.H - 1);
    const int64_t pix = (b * p.H + y) * p.W + x;
    const float2 bf = reinterpret_cast<const float2*>(p.pack_b)[pix];
    const float2 ff = reinterpret_cast<const float2*>(p.pack_f)[pix];
    struct __attribute__((packed, aligned(4))) F3 { float r, g, b; };
    float o[6];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
      const float2 fl = si ? ff : bf;   // image 0 is sampled with the backward flow, image 1 with the forward flow
      const float qy = (float)y + p.fscale * fl.y;
      const float qx = (float)x + p.fscale * fl.x;
      int fy, fx;
      float ay, ax;
      warp_axis(qy, p.H, fy, ay);
      warp_axis(qx, p.W, fx, ax);
      const float* s00 = (si ? p.src3b : p.src3) + ((b * p.H + fy) * p.W + fx) * p.s3stride;
      const float* s10 = s00 + (int64_t)p.W * p.s3stride;
      const F3 tl = *reinterpret_cast<const F3*>(s00), tr = *reinterpret_cast<const F3*>(s00 + p.s3stride);
      const F3 bl = *reinterpret_cast<const F3*>(s10), br = *reinterpret_cast<const F3*>(s10 + p.s3stride);
      o[3 * si + 0] = lerp3(tl.r, tr.r, bl.r, br.r, ax, ay);
      o[3 * si + 1] = lerp3(tl.g, tr.g, bl.g, br.g, ax, ay);
      o[3 * si + 2] = lerp3(tl.b, tr.b, bl.b, br.b, ax, ay);
    }
    float4* mine = stage + threadIdx.x * 4 + (threadIdx.x >> 4);
    mine[0] = float4{o[0], o[1], o[2], o[3]};
    mine[1] = float4{o[4], o[5], bf.x * 0.5f, bf.y * 0.5f};
    mine[2] = float4{ff.x * 0.5f, ff.y * 0.5f, 0.f, 0.f};
    mine[3] = float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // thread t stores quarter t & 3 of the pixels (t >> 2) + 64 i: a wave's store = 16 pixels x 64 contiguous bytes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lp = (threadIdx.x >> 2) + 64u * i, uu = u0 + lp;
      if (uu >= band) break;
      int kk, xx;
      split_unit(uu, p.W, kk, xx);
      if (yb + kk >= p.H) break;
      const float4 v = stage[lp * 4 + (lp >> 4) + (threadIdx.x & 3)];
      *reinterpret_cast<float4*>(p.dst3 + ((b * p.H + yb + kk) * p.W + xx) * p.d3stride + (threadIdx.x & 3) * 4) = v;
    }
    return;
  }
  const unsigned xt = blockIdx.x / nsl, sl = blockIdx.x - xt * nsl;
  const int x = (int)(xt * WARP_TX + (threadIdx.x >> 4)), g = (int)(sl * 16 + (threadIdx.x & 15));
  if (x >= p.W || g >= G) return;
  const int64_t rowpitch = (int64_t)p.W * p.sstride;
  int bsrc = (int)b + p.src_brot, bflw = (int)b + p.flow_brot;
  if (bsrc >= p.NB) bsrc -= p.NB;
  if (bflw >= p.NB) bflw -= p.NB;
  const float* const img = p.src + (int64_t)bsrc * p.H * rowpitch + g * 4;
  float2 flw[WARP_ROWS];
#pragma unroll
  for (int k = 0; k < WARP_ROWS; ++k) flw[k] = warp_flow_at(p, p.coarse ? b : (int64_t)bflw, min(yb + k, p.H - 1), x);
  float4 cbl = float4{0.f, 0.f, 0.f, 0.f}, cbr = cbl;   // bottom corners of the previous row of the band
  int cpix = -(1 << 30);
#pragma unroll
  for (int k0 = 0; k0 < WARP_ROWS; k0 += WARP_BATCH) {
    if (yb + k0 >= p.H) break;
    float ay[WARP_BATCH], ax[WARP_BATCH];
    int pix[WARP_BATCH];
    bool top_old[WARP_BATCH];
#pragma unroll
    for (int j = 0; j < WARP_BATCH; ++j) {
      const int y = min(yb + k0 + j, p.H - 1);     // rows past the end repeat the last one (loads only, no store)
      const float2 fl = flw[k0 + j];
      if (p.flow_out != nullptr && g == 0 && yb + k0 + j < p.H)
        reinterpret_cast<float2*>(p.flow_out)[(b * p.H + y) * p.W + x] = fl;
      const float qy = (float)y + p.fscale * fl.y;
      const float qx = (float)x + p.fscale * fl.x;
      int fy, fx;
      warp_axis(qy, p.H, fy, ay[j]);
      warp_axis(qx, p.W, fx, ax[j]);
      pix[j] = fy * p.W + fx;
      top_old[j] = pix[j] == (j ? pix[j - 1] : cpix) + p.W;
    }
    float4 tl[WARP_BATCH], tr[WARP_BATCH], bl[WARP_BATCH], br[WARP_BATCH];
#pragma unroll
    for (int j = 0; j < WARP_BATCH; ++j) {
      const float* s00 = img + (int64_t)pix[j] * p.sstride;
      bl[j] = *reinterpret_cast<const float4*>(s00 + rowpitch);
      br[j] = *reinterpret_cast<const float4*>(s00 + rowpitch + p.sstride);
      tl[j] = tr[j] = float4{0.f, 0.f, 0.f, 0.f};
      if (!top_old[j]) {
        tl[j] = *reinterpret_cast<const float4*>(s00);
        tr[j] = *reinterpret_cast<const float4*>(s00 + p.sstride);
      }
    }
#pragma unroll
    for (int j = 0; j < WARP_BATCH; ++j)
      if (top_old[j]) { tl[j] = j ? bl[j - 1] : cbl; tr[j] = j ? br[j - 1] : cbr; }
    cbl = bl[WARP_BATCH - 1]; cbr = br[WARP_BATCH - 1]; cpix = pix[WARP_BATCH - 1];
#pragma unroll
    for (int j = 0; j < WARP_BATCH; ++j) {
      const int y = yb + k0 + j;
      if (y >= p.H) break;
      float4 o;
      o.x = lerp3(tl[j].x, tr[j].x, bl[j].x, br[j].x, ax[j], ay[j]);
      o.y = lerp3(tl[j].y, tr[j].y, bl[j].y, br[j].y, ax[j], ay[j]);
      o.z = lerp3(tl[j].z, tr[j].z, bl[j].z, br[j].z, ax[j], ay[j]);
      o.w = lerp3(tl[j].w, tr[j].w, bl[j].w, br[j].w, ax[j], ay[j]);
      *reinterpret_cast<float4*>(p.dst + ((b * p.H + y) * p.W + x) * p.dstride + g * 4) = o;
    }
  }
}

__global__ __launch_bounds__(256) void warp_c3_kernel(WarpParams p) {
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  if (x >= p.W) return;
  const int y = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t pix = (b * p.H + y) * p.W + x;
  const float2 fl = warp_flow_at(p, b, y, x);
  const float qy = (float)y + p.fscale * fl.y;
  const float qx = (float)x + p.fscale * fl.x;
  int fy, fx;
  float ay, ax;
  warp_axis(qy, p.H, fy, ay);
  warp_axis(qx, p.W, fx, ax);
  const float* s00 = p.src + ((b * p.H + fy) * p.W + fx) * p.sstride;
  const float* s10 = s00 + (int64_t)p.W * p.sstride;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    p.dst[pix * p.dstride + c] = lerp3(s00[c], s00[p.sstride + c], s10[c], s10[p.sstride + c], ax, ay);
}

__global__ __launch_bounds__(256) void pack_flow_kernel(PackFlowParams p) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.npix) return;
  const float2 bf = reinterpret_cast<const float2*>(p.bflow)[pix];
  const float2 ff = reinterpret_cast<const float2*>(p.fflow)[pix];
  float* d = p.dst + pix * p.dstride;
  d[0] = bf.x * 0.5f; d[1] = bf.y * 0.5f;
  d[2] = ff.x * 0.5f; d[3] = ff.y * 0.5f;
#pragma unroll
  for (int i = 4; i < 10; ++i) d[i] = 0.f;
}

__global__ __launch_bounds__(256) void fill_random_kernel(float* dst, int64_t n, uint32_t seed) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    dst[i] = (float)(int32_t)x * (1.0f / 2147483648.0f);
  }
}

// eval/util.py:51-52 (write_image): clip(x * 255, 0, 255) + 0.5, truncated to uint8 - the same float32 operations in the same order
// (no fused multiply-add: the file is built with -ffp-contract=off), four values per thread
__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  auto cv = [](float x) -> uint32_t {
    float v = x * 255.f;
    v = v < 0.f ? 0.f : v;      // np.clip = minimum(maximum(x, 0), 255)
    v = v > 255.f ? 255.f : v;
    return (uint32_t)(v + 0.5f);
  };
  if (i + 3 < n && ((reinterpret_cast<uintptr_t>(src + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst + i) & 3) == 0)) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    *reinterpret_cast<uint32_t*>(dst + i) = cv(v.x) | (cv(v.y) << 8) | (cv(v.z) << 16) | (cv(v.w) << 24);
  } else {
    for (int64_t k = i; k < n && k < i + 4; ++k) dst[k] = (uint8_t)cv(src[k]);
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t film_launch_conv_pw(const ConvPwParams& p, hipStream_t s) {
  const dim3 grid(blocks_for(p.M)), block(256);
  const size_t sh = (size_t)p.Cin * p.Cout * sizeof(float);
  switch (p.Cout) {
    case 2: hipLaunchKernelGGL(conv_pw_kernel<2>, grid, block, sh, s, p); break;
    case 3: hipLaunchKernelGGL(conv_pw_kernel<3>, grid, block, sh, s, p); break;
    case 16: hipLaunchKernelGGL(conv_pw_kernel<16>, grid, block, sh, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t film_launch_flow_head(const FlowHeadParams& p, hipStream_t s) {
  const size_t sh = ((size_t)p.Cin * 16 + 32) * sizeof(float);
  hipLaunchKernelGGL(flow_head_kernel, dim3(blocks_for(p.M)), dim3(256), sh, s, p);
  return hipGetLastError();
}

hipError_t film_launch_pool(const PoolParams& p, hipStream_t s) {
  const int Ho = p.H / 2, Wo = p.W / 2;
  const int64_t units = p.C == 3 ? Wo : (int64_t)Wo * (p.C / 4);
  if (units >= (1 << 24) || Ho > 65535 || p.NB > 65535) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((units + 255) / 256), (unsigned)Ho, (unsigned)p.NB);
  if (p.C == 3) {
    hipLaunchKernelGGL(pool_c3_kernel, grid, dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL(pool_vec_kernel, grid, dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

hipError_t film_launch_flow_up(const FlowUpParams& p, hipStream_t s) {
  const int64_t opix = (int64_t)p.NB * p.h * 2 * p.w * 2;
  hipLaunchKernelGGL(flow_up_kernel, dim3(blocks_for(opix)), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t film_launch_flow_add(const FlowAddParams& p, hipStream_t s) {
  hipLaunchKernelGGL(flow_add_kernel, dim3(blocks_for(p.n)), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t film_launch_warp(const WarpParams& p, hipStream_t s) {
  const int64_t units = p.C == 3 ? p.W : (int64_t)p.W * (p.C / 4);
  if (units >= (1 << 24) || p.H > 65535 || p.NB > 65535) return hipErrorInvalidValue;
  if (p.coarse != nullptr && ((p.H | p.W) & 1)) return hipErrorInvalidValue;
  if (p.C == 3) {
    if (p.flow_out != nullptr || p.dst3 != nullptr || p.src_brot || p.flow_brot) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_c3_kernel, dim3((unsigned)((units + 255) / 256), (unsigned)p.H, (unsigned)p.NB), dim3(256), 0, s, p);
  } else {
    if (p.dst3 != nullptr && (!p.src3 || !p.src3b || !p.pack_b || !p.pack_f)) return hipErrorInvalidValue;
    if (p.src_brot < 0 || p.src_brot >= p.NB || p.flow_brot < 0 || p.flow_brot >= p.NB || (p.coarse != nullptr && p.flow_brot)) return hipErrorInvalidValue;
    if (p.dst3 != nullptr && ((p.d3stride & 3) || (reinterpret_cast<uintptr_t>(p.dst3) & 15))) return hipErrorInvalidValue;
    const int64_t blocks = (int64_t)((p.W + WARP_TX - 1) / WARP_TX) * ((p.C / 4 + 15) / 16) + (p.dst3 != nullptr ? ((int64_t)p.W * WARP_ROWS + 255) / 256 : 0);
    hipLaunchKernelGGL(warp_vec_kernel, dim3((unsigned)blocks, (unsigned)((p.H + WARP_ROWS - 1) / WARP_ROWS), (unsigned)p.NB), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

hipError_t film_launch_pack_flow(const PackFlowParams& p, hipStream_t s) {
  hipLaunchKernelGGL(pack_flow_kernel, dim3(blocks_for(p.npix)), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---- Interpolator.__call__ data movement ----------------------------------------------------------------
// thread = one float of the tile buffer (frame_to_tiles) / of the frame (tiles_to_frame); both sides are
// contiguous in x*3+c, so consecutive threads read and write consecutive floats of a row.
__global__ __launch_bounds__(256) void frame_to_tiles_kernel(TileMapParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int row = p.TW * 3;
  const int64_t total = (int64_t)p.ntiles * p.TH * row;
  if (i >= total) return;
  const int xc = (int)(i % row);
  const int64_t r = i / row;
  const int y = (int)(r % p.TH);
  const int n = (int)(r / p.TH) + p.tile0;
  const int b = n / (p.bh * p.bw), t = n % (p.bh * p.bw);
  const int ty = t / p.bw, tx = t % p.bw;
  const int sy = y - p.oy, sxc = xc - p.ox * 3;
  float v = 0.f;  // tf.image.pad_to_bounding_box pads with zeros
  if (sy >= 0 && sy < p.ph && sxc >= 0 && sxc < p.pw * 3)
    v = p.src[(((int64_t)b * p.H + ty * p.ph + sy) * p.W + tx * p.pw) * 3 + sxc];
  p.dst[i] = v;
}

__global__ __launch_bounds__(256) void tiles_to_frame_kernel(TileMapParams p) {
  // thread = one float of the patches [tile0, tile0 + ntiles) as they lie in the frame
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int row = p.pw * 3;
  const int64_t total = (int64_t)p.ntiles * p.ph * row;
  if (i >= total) return;
  const int xc = (int)(i % row);
  const int64_t r = i / row;
  const int y = (int)(r % p.ph);
  const int ln = (int)(r / p.ph);
  const int n = ln + p.tile0;
  const int b = n / (p.bh * p.bw), t = n % (p.bh * p.bw);
  const int ty = t / p.bw, tx = t % p.bw;
  const float v = p.src[(((int64_t)ln * p.TH + p.oy + y) * p.TW + p.ox) * 3 + xc];
  p.dst[(((int64_t)b * p.H + ty * p.ph + y) * p.W + tx * p.pw) * 3 + xc] = v;
}

hipError_t film_launch_frame_to_tiles(const TileMapParams& p, hipStream_t s) {
  hipLaunchKernelGGL(frame_to_tiles_kernel, dim3(blocks_for((int64_t)p.ntiles * p.TH * p.TW * 3)), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t film_launch_tiles_to_frame(const TileMapParams& p, hipStream_t s) {
  hipLaunchKernelGGL(tiles_to_frame_kernel, dim3(blocks_for((int64_t)p.ntiles * p.ph * p.pw * 3)), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t film_launch_to_uint8(const float* src, uint8_t* dst, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(to_uint8_kernel, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, s, src, dst, n);
  return hipGetLastError();
}

hipError_t film_launch_fill_random(float* dst, int64_t n, uint32_t seed, hipStream_t s) {
  hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, s, dst, n, seed);
  return hipGetLastError();
}
