// conv_c3_impl.h -- the first layer of the sub-extractor (feature_extractor.py:119-120, `cfeat_conv_0`): 3x3 Conv2D('same')
// of the 3-channel image, K = 27, + bias + leaky_relu.  0.2 % of the FLOPs and 2 GB of output per 1080p step: an HBM-write
// problem (output 1.13 GB at level 0 = 0.2 ms at 5.6 TB/s) that the first-generation implicit-GEMM kernel
// (conv_igemm_kernel's 3-channel mode: LDS staging of a [256][48] A tile per step) ran at 0.5 ms.
//
// Here nothing goes through LDS: K = 27 (+1 zero) is 14 steps of v_mfma_f32_32x32x2_f32.
//   * B operand = the weights, 14 registers per 32-channel tile per lane, loaded ONCE per wave from the first-layer
//     layout [12 tap slots][4][Cout] and kept for every tile the wave computes;
//   * A operand = lane (pixel m = lane & 31, k half = lane >> 5) needs in[y + dy - 1][x0 + m + dx - 1][c] for
//     k = 2 step + half = tap * 3 + c: 14 dword buffer loads per 32-pixel tile, neighbouring lanes read neighbouring
//     pixels (12-byte stride), out-of-image taps fail the buffer bounds check and read zero = the 'same' padding;
//   * a wave walks ROWS rows of a 32-pixel column strip, a workgroup = 4 waves = 128 pixels x ROWS rows; the next tile's
//     loads are requested before the current tile's MFMAs;
//   * epilogue: C/D layout col = lane & 31 = output channel, 16 registers = rows m -> one 128-byte row of 32 channels per
//     register and wave half, straight to the channel slice of the destination.
// k ascends (tap, channel) in one fma chain per output, like the kernel it replaces.  COUT = the 64 or 32 output channels of a
// workgroup; blockIdx.z walks the channel blocks of a wider first layer (filters = 96, 128, ...: any multiple of 32).
#pragma once
#include "conv_buf_impl.h"

template <int COUT, int ROWS>
__global__ __launch_bounds__(256) void conv_c3_kernel(ConvParams p) {
  constexpr int TN = COUT / 32;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int ntx = (p.W + 127) >> 7;
  const int tx = (int)blockIdx.x % ntx, ty = (int)blockIdx.x / ntx;
  const int img = blockIdx.y;
  const int nb0 = (int)blockIdx.z * COUT;   // first output channel of this workgroup
  const int x0 = tx * 128 + wv * 32, yb = ty * ROWS;
  if (x0 >= p.W) return;

  // weights: k = 2 s + half -> (tap, c); row tap * 4 + c of the [48][Cout] first-layer pack; k = 27 is the zero pad
  float wb[TN][14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + half;
    const int tap = k / 3, c = k - tap * 3;
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) wb[nt][s] = k < 27 ? p.w[(tap * 4 + c) * p.Cout + nb0 + nt * 32 + l31] : 0.f;
  }
  float bias[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) bias[nt] = p.bias[nb0 + nt * 32 + l31];

  const ConvSeg& sg = p.seg[0];
  int be = img + sg.boff;
  if (sg.bmod && be >= sg.bmod) be -= sg.bmod;
  const conv_rsrc_t rsrc = conv_make_rsrc(sg.ptr);
  const int x = x0 + l31;
  // per-lane tap table: byte offset of (dy, dx, c) relative to pixel (y, x), and whether the column is inside the image
  int toff[14];
  unsigned colok = 0, dyv = 0;   // bit s: column valid; 2 bits per s: dy
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + half;
    const int tap = k < 27 ? k / 3 : 0, c = k < 27 ? k - tap * 3 : 0;
    const int dy = tap / 3, dx = tap - dy * 3;
    toff[s] = (((dy - 1) * p.W + (dx - 1)) * sg.stride + c) * 4;
    const int xx = x + dx - 1;
    if (k < 27 && xx >= 0 && xx < p.W) colok |= 1u << s;
    dyv |= (unsigned)dy << (2 * s);
  }
  const unsigned OOB = 0xFFFFFFFFu;
  auto load_tile = [&](int y, float* a) {
    const unsigned base = (unsigned)((((size_t)be * p.H + y) * p.W + x) * sg.stride) * 4u;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int yy = y + (int)((dyv >> (2 * s)) & 3u) - 1;
      const bool ok = ((colok >> s) & 1u) && yy >= 0 && yy < p.H && y < p.H;
      a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? (int)(base + (unsigned)toff[s]) : (int)OOB, 0, 0));
    }
  };

  float a_cur[14], a_nxt[14];
  load_tile(yb, a_cur);
#pragma unroll 1
  for (int r = 0; r < ROWS; ++r) {
    const int y = yb + r;
    if (y >= p.H) break;
    if (r + 1 < ROWS) load_tile(y + 1, a_nxt);
    f32x16 acc[TN];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[nt][q] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], wb[nt][s], acc[nt], 0, 0, 0);
    // Stores: buffer stores from a SCALAR row corner (image, row y, first pixel x0, first channel) + one per-lane offset + a scalar
    // offset per accumulator register; pixels past the row end fall outside the resource's size and are dropped by the bounds check.
    // (Round 6: sixteen 64-bit pointers in registers and a branch per store made this 176 VGPRs = two waves per SIMD.)
    const int npx = p.W - x0 < 32 ? p.W - x0 : 32;
    float* const corner = p.out + (((size_t)img * p.H + y) * p.W + x0) * p.ostride + nb0;
    const conv_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(corner, 0, (int)((unsigned)npx * (unsigned)p.ostride * 4u), 0x00020000);
    const unsigned pitch = (unsigned)p.ostride * 4u;
    const unsigned ovoff = (unsigned)(4 * half) * pitch + (unsigned)l31 * 4u;
    const float slope = p.leaky ? 0.2f : 1.f;   // max(v, slope v): leaky_relu(0.2) or v, the bits of the compare-and-select form
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2);   // + 4 half: in ovoff
        const float v = acc[nt][q] + bias[nt];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, __builtin_fmaxf(v, slope * v)), orsrc, (int)(ovoff + (unsigned)(nt * 128)), (int)((unsigned)m * pitch), 0);
      }
#pragma unroll
    for (int s = 0; s < 14; ++s) a_cur[s] = a_nxt[s];
  }
}

template <int COUT>
hipError_t conv_c3_launch(const ConvParams& p, hipStream_t s) {
  constexpr int ROWS = 8;
  if (p.nseg != 1 || p.ksize != 3 || p.Cout % COUT || p.seg[0].C != 3 || p.ksplit > 1) return hipErrorInvalidValue;
  const int ntx = (p.W + 127) / 128, nty = (p.H + ROWS - 1) / ROWS;
  hipLaunchKernelGGL((conv_c3_kernel<COUT, ROWS>), dim3((unsigned)(ntx * nty), (unsigned)p.NB, (unsigned)(p.Cout / COUT)), dim3(256), 0, s, p);
  return hipGetLastError();
}
