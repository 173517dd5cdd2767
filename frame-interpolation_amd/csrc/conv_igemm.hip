// conv_igemm.hip -- dispatcher of the convolution kernels: Conv2D('same', stride 1) + bias + leaky_relu(0.2) as
// implicit GEMMs on the gfx950 matrix cores.
//
// Replaces every tf.keras.layers.Conv2D of the reference hot path whose Cout is a multiple of 32:
//   feature_extractor.py:93-99,142-143   pyramid_flow_estimator.py:66-76,85-98   fusion.py:83-97,135-138
//
// Common mapping: M = output pixels, N = Cout, K = taps x (concatenated input channels); 4, 8 or 16 waves per
// workgroup, each wave owning a grid of 32x32 MFMA tiles with their accumulators in registers; inputs gathered from
// channel slices of NHWC buffers with 32-bit buffer offsets (zero padding = the bounds check), the epilogue adds the
// bias, applies leaky_relu and stores 128-byte rows into the channel slice of the destination buffer (the consumer's
// concat input).  fp32 throughout (v_mfma_f32_32x32x2_f32) except the opt-in bf16x6 mode.
//
// Kernel templates (the family of a layer is a pure function of its shape and the engine options, never of timing):
//   conv_wino_impl.h   3x3 convs on the large levels: 1-D Winograd F(2,3) along x on the halo staging, 1.5x fewer
//                      fp32 MFMAs.
//   conv_halo_impl.h   3x3 convs with deep K: the activation halo patch is staged once per 16-channel chunk and the
//                      nine taps run out of LDS (6.8x fewer A loads / LDS stores, less L2 traffic, higher clocks).
//   conv_split_impl.h  precision mode bf16x6: the halo kernel with every fp32 operand split exactly into three bf16
//                      pieces, six partial products on the bf16 matrix pipe.
//   conv_buf_impl.h    the general kernel (1x1, small levels, and the sub-pixel-folded upsample + 2x2 convs): buffer
//                      loads with hardware zero fill, K-major weights, no vector instruction per K-step besides loads,
//                      LDS traffic and MFMAs.
//   conv_fold4_impl.h  the decoder's nearest x2 upsample + 2x2 conv in its difference form: 4 multiplies per low-resolution pixel
//                      instead of the 9 of the sub-pixel fold on conv_buf_kernel.
//   conv_c3_impl.h     the 3-channel first layer (K = 27): no LDS, weights in registers.
// (conv_igemm_impl.h, the first-generation kernel that ran the first layer of configurations with filters other than 32 / 64 until
// round 4, is under tools/retired/.)
// The default library instantiates only the families a default plan can select: conv_wino2d / conv_wino43 / conv_fold4 / conv_buf / conv_c3.
// FILM_EXTRA_FAMILIES=1 at build time (film_hip/build.py, Makefile EXTRA=1) adds the ones behind opt-in options: the bf16 split
// precision modes (conv_split / conv_winox3 / conv_foldx3) and the F(2,3) / halo fp32 kernels (options winograd = 2, halo_all).
#include "conv_buf_impl.h"
#include "conv_wino43_impl.h"
#include "conv_wino2d_impl.h"
#include "conv_fold4_impl.h"
#include "conv_c3_impl.h"
#ifdef FILM_EXTRA_FAMILIES
#include "conv_halo_impl.h"
#include "conv_split_impl.h"
#include "conv_wino_impl.h"
#include "conv_winox3_impl.h"
#include "conv_foldx3_impl.h"
#endif

template <int F>
static hipError_t launch_shape(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case TILE_128x128: return conv_buf_launch<128, 128, 2, 2, F>(p, s);
    case TILE_256x64: return conv_buf_launch<256, 64, 4, 1, F>(p, s);
    case TILE_256x32: return conv_buf_launch<256, 32, 4, 1, F>(p, s);
    case TILE_64x64: return conv_buf_launch<64, 64, 2, 2, F>(p, s);
    case TILE_128x32: return conv_buf_launch<128, 32, 4, 1, F>(p, s);
    case TILE_128x64: return conv_buf_launch<128, 64, 2, 2, F>(p, s);
    case TILE_256x128: return conv_buf_launch<256, 128, 4, 2, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

// first layer (3-channel image): conv_c3_kernel, 64 output channels per workgroup (32 when Cout is not a multiple of 64)
static hipError_t launch_c3(const ConvParams& p, int shape, hipStream_t s) {
  if (shape != TILE_C3_DIRECT) return hipErrorInvalidValue;
  return p.Cout % 64 == 0 ? conv_c3_launch<64>(p, s) : conv_c3_launch<32>(p, s);
}

#ifdef FILM_EXTRA_FAMILIES
template <int F>
static hipError_t launch_halo(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case HALO_8x128: return conv_halo_launch<8, 128, 4, 2, F>(p, s);
    case HALO_8x64: return conv_halo_launch<8, 64, 4, 1, F>(p, s);
    case HALO_8x32: return conv_halo_launch<8, 32, 4, 1, F>(p, s);
    case HALO_4x64: return conv_halo_launch<4, 64, 4, 1, F>(p, s);
    case HALO_4x128: return conv_halo_launch<4, 128, 2, 2, F>(p, s);
    case HALO_4x32: return conv_halo_launch<4, 32, 4, 1, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int F, int NP>
static hipError_t launch_split(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case HALO_8x128: return conv_halo_split_launch<8, 128, 4, 2, NP, F>(p, s);
    case HALO_8x64: return conv_halo_split_launch<8, 64, 4, 1, NP, F>(p, s);
    case HALO_8x32: return conv_halo_split_launch<8, 32, 4, 1, NP, F>(p, s);
    case HALO_4x64: return conv_halo_split_launch<4, 64, 4, 1, NP, F>(p, s);
    case HALO_4x128: return conv_halo_split_launch<4, 128, 2, 2, NP, F>(p, s);
    case HALO_4x32: return conv_halo_split_launch<4, 32, 4, 1, NP, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int F>
static hipError_t launch_wino(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case WINO_4x128: return conv_wino_launch<4, 128, 4, 2, F>(p, s);
    case WINO_4x64: return conv_wino_launch<4, 64, 4, 1, F>(p, s);
    case WINO_4x128_W16: return conv_wino_launch<4, 128, 4, 4, F>(p, s);
    case WINO_4x32: return conv_wino_launch<4, 32, 4, 1, F>(p, s);
    case WINO_8x64_W16: return conv_wino_launch<8, 64, 8, 2, F>(p, s);
    case WINO_8x32_W8: return conv_wino_launch<8, 32, 8, 1, F>(p, s);
    case WINO_2x64: return conv_wino_launch<2, 64, 2, 2, F>(p, s);
    case WINO_4x64_W8: return conv_wino_launch<4, 64, 4, 2, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

#endif   // FILM_EXTRA_FAMILIES

// The default library holds the SEVEN conv_wino43_kernel tiles film_w43_shape_built() names (film_kernels.h; a default plan runs this
// kernel on two layers of a 256x256 frame's 32x32 level only - the others were 27 instantiations nobody could select without an
// option); the FILM_EXTRA_FAMILIES flavour holds all seventeen, for the tile-shape tests and the "winograd" = 3 / "wino2d" = 0 A/B runs.
template <int F>
static hipError_t launch_wino43(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case W43_Q16_4x64_T21_P2: return conv_wino43_launch<4, 64, 2, 1, F | W43_F_PF2, 16>(p, s);
    case W43_Q16_4x32_T11_P2: return conv_wino43_launch<4, 32, 1, 1, F | W43_F_PF2, 16>(p, s);
    case W43_Q16_4x64_N1_P2: return conv_wino43_launch<4, 64, 1, 1, F | W43_F_PF2, 16, 1>(p, s);
    case W43_Q8_8x64_T21_P2: return conv_wino43_launch<8, 64, 2, 1, F | W43_F_PF2, 8>(p, s);
    case W43_Q8_8x64_N1_P2: return conv_wino43_launch<8, 64, 1, 1, F | W43_F_PF2, 8, 1>(p, s);
    case W43_Q8_8x32_T11_BG: return conv_wino43_launch<8, 32, 1, 1, F | W43_F_BG, 8>(p, s);
    case W43_Q8_8x32_T11_P2: return conv_wino43_launch<8, 32, 1, 1, F | W43_F_PF2, 8>(p, s);
#ifdef FILM_EXTRA_FAMILIES
    case W43_4x64_T21: return conv_wino43_launch<4, 64, 2, 1, F>(p, s);
    case W43_4x64_T12: return conv_wino43_launch<4, 64, 1, 2, F>(p, s);
    case W43_4x32_T11: return conv_wino43_launch<4, 32, 1, 1, F>(p, s);
    case W43_Q16_4x64_T21: return conv_wino43_launch<4, 64, 2, 1, F, 16>(p, s);
    case W43_Q16_4x64_T12: return conv_wino43_launch<4, 64, 1, 2, F, 16>(p, s);
    case W43_Q16_4x32_T11: return conv_wino43_launch<4, 32, 1, 1, F, 16>(p, s);
    case W43_Q16_4x64_N1: return conv_wino43_launch<4, 64, 1, 1, F, 16, 1>(p, s);
    case W43_Q16_4x64_T12_P2: return conv_wino43_launch<4, 64, 1, 2, F | W43_F_PF2, 16>(p, s);
    case W43_Q16_4x32_T11_BG: return conv_wino43_launch<4, 32, 1, 1, F | W43_F_BG, 16>(p, s);
    case W43_Q8_8x64_T12_P2: return conv_wino43_launch<8, 64, 1, 2, F | W43_F_PF2, 8>(p, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

template <int F>
static hipError_t launch_wino2d(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case W2D_8x64: return conv_wino2d_launch_any<64, F>(p, s);
    case W2D_8x32: return conv_wino2d_launch_any<32, F>(p, s);
    case W2D_8x32_S2: return conv_wino2d_launch_any<32, F, 2>(p, s);
    case W2D_16x64: return conv_wino2d_launch_any<64, F | W2D_F_SQ>(p, s);
    case W2D_16x32: return conv_wino2d_launch_any<32, F | W2D_F_SQ>(p, s);
    case W2D_16x32_S2: return conv_wino2d_launch_any<32, F | W2D_F_SQ, 2>(p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int F>
static hipError_t launch_fold4(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case F4_4x64: return conv_fold4_launch<2, F>(p, s);
    case F4_4x32: return conv_fold4_launch<1, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

#ifdef FILM_EXTRA_FAMILIES
template <int F>
static hipError_t launch_winox3(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case WX3_4x128_T22: return conv_winox3_launch<4, 128, 2, 2, F>(p, s);
    case WX3_4x64_T12: return conv_winox3_launch<4, 64, 1, 2, F>(p, s);
    case WX3_4x64_T21: return conv_winox3_launch<4, 64, 2, 1, F>(p, s);
    case WX3_4x32_T11: return conv_winox3_launch<4, 32, 1, 1, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int F>
static hipError_t launch_foldx3(const ConvParams& p, int shape, hipStream_t s) {
  switch (shape) {
    case FX3_4x64: return conv_foldx3_launch<4, 64, 4, 1, F>(p, s);
    case FX3_8x64: return conv_foldx3_launch<8, 64, 8, 1, F>(p, s);
    case FX3_4x128: return conv_foldx3_launch<4, 128, 4, 2, F>(p, s);
    default: return hipErrorInvalidValue;
  }
}
#endif   // FILM_EXTRA_FAMILIES


// Second half of a split-K convolution (film_kernels.h, ConvParams::ksplit): out = act(bias + part[0] + part[1] + ...),
// partials added in split order.  Thread = one float4 of an output pixel.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                 float* __restrict__ out, int M, int Cout, int ostride,
                                                                 int S, int leaky) {
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

static hipError_t film_launch_conv_main(const ConvParams& p, int tile, hipStream_t s) {
  const int shape = (tile & (CONV_TILE_XCD - 1)) + (((tile & CONV_TILE_EXT) && (tile & CONV_TILE_F43)) ? 16 : 0);
  if ((p.pool_out != nullptr || p.pw_out != nullptr) && !(tile & CONV_TILE_W2D) && !((tile & CONV_TILE_WINO) && (tile & CONV_TILE_F43) && !(tile & CONV_TILE_X3))) return hipErrorInvalidValue;
  if (tile & CONV_TILE_FOLD4) return (tile & CONV_TILE_XCD) ? launch_fold4<CONV_B_XCD_M>(p, shape, s) : launch_fold4<0>(p, shape, s);   // (the launcher checks the op)
  if (tile & CONV_TILE_W2D) {
    if (p.ksize != 3) return hipErrorInvalidValue;   // (fused pool / 1x1: checked by the launcher)
    return (tile & CONV_TILE_XCD) ? launch_wino2d<CONV_B_XCD_M>(p, shape, s) : launch_wino2d<0>(p, shape, s);
  }
#ifndef FILM_EXTRA_FAMILIES
  // families that are not in this build (the planner never selects them without FILM_EXTRA_FAMILIES)
  if ((tile & (CONV_TILE_FOLDX3 | CONV_TILE_SPLIT | CONV_TILE_HALO | CONV_TILE_X3)) || ((tile & CONV_TILE_WINO) && !(tile & CONV_TILE_F43)))
    return hipErrorNotSupported;
  if (tile & CONV_TILE_WINO) {
    if (p.ksize != 3) return hipErrorInvalidValue;
    if (p.pool_out != nullptr && shape < W43_Q16_4x64_T21) return hipErrorInvalidValue;   // fused pool: 64-pixel tiles only
    if (p.pw_out != nullptr && ((shape != W43_Q16_4x64_N1 && shape != W43_Q16_4x64_N1_P2 && shape != W43_Q8_8x64_N1_P2) || p.Cout != 64 || p.ksplit > 1 ||
                                p.pool_out != nullptr || p.pw_cout < 1 || p.pw_cout > 4))
      return hipErrorInvalidValue;   // fused 1x1: a workgroup must hold all 64 channels of its pixels in one wave set
    return (tile & CONV_TILE_XCD) ? launch_wino43<CONV_B_XCD_M>(p, shape, s) : launch_wino43<0>(p, shape, s);
  }
#else
  if (tile & CONV_TILE_FOLDX3) {
    if (p.ksize != 2 || p.fold != 2) return hipErrorInvalidValue;
    return (tile & CONV_TILE_XCD) ? launch_foldx3<CONV_B_XCD_M>(p, shape, s) : launch_foldx3<0>(p, shape, s);
  }
  if (tile & CONV_TILE_WINO) {
    if (p.ksize != 3) return hipErrorInvalidValue;
    if (tile & CONV_TILE_X3) return (tile & CONV_TILE_XCD) ? launch_winox3<CONV_B_XCD_M>(p, shape, s) : launch_winox3<0>(p, shape, s);
    if ((tile & CONV_TILE_F43) && p.pool_out != nullptr && shape < W43_Q16_4x64_T21) return hipErrorInvalidValue;   // fused pool: 64-pixel tiles only
    if (p.pw_out != nullptr && (!(tile & CONV_TILE_F43) || (shape != W43_Q16_4x64_N1 && shape != W43_Q16_4x64_N1_P2 && shape != W43_Q8_8x64_N1_P2) || p.Cout != 64 || p.ksplit > 1 ||
                                p.pool_out != nullptr || p.pw_cout < 1 || p.pw_cout > 4))
      return hipErrorInvalidValue;   // fused 1x1: a workgroup must hold all 64 channels of its pixels in one wave set
    if (tile & CONV_TILE_F43) return (tile & CONV_TILE_XCD) ? launch_wino43<CONV_B_XCD_M>(p, shape, s) : launch_wino43<0>(p, shape, s);
    return (tile & CONV_TILE_XCD) ? launch_wino<CONV_B_XCD_M>(p, shape, s) : launch_wino<0>(p, shape, s);
  }
  if (tile & CONV_TILE_SPLIT) {
    if (p.ksize != 3) return hipErrorInvalidValue;
    if (tile & CONV_TILE_X3) return (tile & CONV_TILE_XCD) ? launch_split<CONV_B_XCD_M, 3>(p, shape, s) : launch_split<0, 3>(p, shape, s);
    return (tile & CONV_TILE_XCD) ? launch_split<CONV_B_XCD_M, 6>(p, shape, s) : launch_split<0, 6>(p, shape, s);
  }
  if (tile & CONV_TILE_HALO) {
    if (p.ksize != 3) return hipErrorInvalidValue;
    return (tile & CONV_TILE_XCD) ? launch_halo<CONV_B_XCD_M>(p, shape, s) : launch_halo<0>(p, shape, s);
  }
#endif   // FILM_EXTRA_FAMILIES
  if (tile & CONV_TILE_C3) return launch_c3(p, shape, s);
  return (tile & CONV_TILE_XCD) ? launch_shape<CONV_B_XCD_M>(p, shape, s) : launch_shape<0>(p, shape, s);
}

hipError_t film_launch_conv(const ConvParams& p, int tile, hipStream_t s) {
  // split-K is implemented by conv_buf_kernel, conv_wino43_kernel, conv_wino2d_kernel and conv_fold4_kernel
  const bool can_split = !(tile & (CONV_TILE_FOLDX3 | CONV_TILE_SPLIT | CONV_TILE_HALO | CONV_TILE_C3 | CONV_TILE_X3)) &&
                         (!(tile & CONV_TILE_WINO) || (tile & CONV_TILE_F43));
  if (p.ksplit > 1 && !can_split) return hipErrorInvalidValue;
  const hipError_t e = film_launch_conv_main(p, tile, s);
  if (e != hipSuccess || p.ksplit <= 1) return e;
  const long long Mout = (long long)p.M * (p.fold == 3 ? 4 : 1);   // conv_fold4_kernel: M counts the low-resolution pixels, four outputs each
  if (!p.part || (p.Cout & 3) || Mout * (p.Cout >> 2) >= (1ll << 32)) return hipErrorInvalidValue;
  const unsigned units = (unsigned)Mout * (unsigned)(p.Cout >> 2);
  hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((units + 255) / 256), dim3(256), 0, s, p.part, p.bias, p.out, (int)Mout, p.Cout,
                     p.ostride, p.ksplit, p.leaky);
  return hipGetLastError();
}
