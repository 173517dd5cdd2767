// film_kernels.h -- parameter blocks and launchers of the gfx950 kernels behind libfilm_hip.so.
//
// Every activation is a pixel-major ("NHWC") float32 buffer [N][H][W][stride]; a kernel reads
// or writes a *channel slice* of such a buffer (pointer pre-offset by the first channel,
// `stride` = floats per pixel of the underlying buffer).  That is how every tf.concat of the
// reference (feature_extractor.py:191, pyramid_flow_estimator.py:95, interpolator.py:167-183,
// fusion.py:136) disappears: producers write straight into their slice of the consumer's input.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FILM_MAX_SEG 4
#define FILM_BK 16  // channels per K-step of the implicit GEMM; every concat segment is a multiple

// One K-segment of a (virtually concatenated) convolution input.
struct ConvSeg {
  const float* ptr;  // [NBs][Hs][Ws][stride], first channel of the segment
  int stride;        // floats per pixel
  int C;             // channels in the segment, multiple of FILM_BK
  int boff, bmod;    // batch remap: b' = b + boff; if (b' >= bmod) b' -= bmod   (bmod = 0: none)
  int up;            // 1: source is (H/2, W/2); nearest-neighbour x2 on read (fusion.py:133-134)
};

// Conv2D(padding='same', stride 1) [+ bias] [+ leaky_relu(0.2)] as implicit GEMM:
//   M = NB*H*W output pixels, N = Cout, K = ksize^2 * Ctot, K ordered (tap, segment, channel).
// Every segment is addressed with 32-bit byte offsets from seg.ptr: the buffer behind it must be < 4 GiB - except in
// conv_wino43_kernel, whose buffer resource starts at the workgroup's own first halo row (any size).
struct ConvParams {
  ConvSeg seg[FILM_MAX_SEG];
  int nseg;
  int ksize;          // 1, 2, 3; TF 'same': pad_before = (ksize-1)/2, rest after
  const float* w;     // conv_buf_kernel: packed [Cout][ksize*ksize*Ctot] (K contiguous per output channel);
                      // conv_wino_kernel: [Cout][Ctot/8][12][8] (F(2,3)-transformed along x);
                      // conv_halo_kernel: [Cout][Ctot/16][9][16]; conv_halo_split_kernel: [Cout][Ctot/16][9][3][16] bf16;
                      // conv_igemm_kernel first-layer mode: [48][Cout]
  const float* bias;  // [Cout]
  float* out;         // [NB][H][W][ostride], first output channel
  int ostride;
  int NB, H, W;
  int Cout, Ctot;
  int leaky;
  int M;
  // Sub-pixel fold of nearest-x2-upsample + 2x2 conv (fusion.py:133-135), conv_buf_kernel only.  fold = 1: the
  // launch computes output phase (py, px), i.e. out[2y+py][2x+px] for every pixel (y, x) of the LOW-resolution input
  // (H, W, M describe that grid; the output is 2H x 2W), from `ftaps` taps (tdy[t], tdx[t]) in {0,1}^2 of it, with
  // weights [Cout][ftaps * Ctot] pre-summed over the kernel taps that read the same input pixel.
  // fold = 2: all four phases in ONE launch, phase = blockIdx.z = py*2 + px, taps (a, b) with a <= py, b <= px in
  // raster order, weights of phase q at w + fold_woff[q] (py, px, ftaps, tdy, tdx are then ignored).
  // fold = 3 (conv_fold4_kernel, conv_fold4_impl.h): the same op in its difference form - four GEMMs over K = Ctot on the planes
  // I, Dx, Dy, Dxy of the low-resolution input, combined in the epilogue; weights [Cout/32][chunk of 8][plane 4][K half][32][4]
  // (S, Sx, Sy, W11); py, px, ftaps, tdy, tdx, fold_woff are ignored.
  int fold, py, px, ftaps;
  signed char tdy[4], tdx[4];
  long long fold_woff[4];
  // Split-K (conv_buf_kernel, fold == 0): ksplit > 1 -> blockIdx.z = split s sums the K steps
  // [s * nsteps / ksplit, (s + 1) * nsteps / ksplit) and writes the raw partial sums to part[s][M][Cout]; film_launch_conv
  // then adds the partials in split order, the bias and the activation (conv_splitk_reduce_kernel).  No atomics: the
  // result is a fixed function of (shape, ksplit).  For the deep, small-M layers of the coarse pyramid levels, whose
  // K loop (up to 1080 steps) otherwise runs on a handful of workgroups.
  int ksplit;
  float* part;
  // conv_wino2d_kernel with W2D_F_CHAIN: consecutive pixel tiles one workgroup walks (>= 1; conv_wino2d_impl.h, "chained tiles")
  int chain;
  // conv_wino2d_kernel, filled by ITS LAUNCHER (callers leave them 0): ceil(2^32 / d) for d = tiles per image, tiles per row, channel blocks of
  // the grid - the workgroup -> (image, tile row, tile column, channel block) decomposition then costs three s_mul_hi_u32 instead of three
  // 40-instruction integer divisions in front of the first DMA request (0: divide; conv_wino2d_impl.h, "prologue diet")
  unsigned mg_tpi, mg_ntx, mg_nby;
  int tl_ntx, tl_tpi;   // ... and the tiles per row / per image themselves
  // Fused AveragePooling2D(2, 2) of the output (feature_extractor.py:138-146: every sub-extractor stage but the last
  // is followed by a pool), conv_wino43_kernel's 64-pixel tiles only: the epilogue also writes
  // pool_out[img][y/2][x/2][n] = (((o(y,x) + o(y,x+1)) + o(y+1,x)) + o(y+1,x+1)) * 0.25 (pool_vec_kernel's order) from
  // the activated outputs it holds in registers - the former pool launch and its re-read of the feature map.  H, W even.
  float* pool_out;    // nullptr: none
  int pool_ostride;
  // Fused 1x1 convolution behind this layer (conv_wino43_kernel, the NH = 1 tiles with BN = Cout = 64: a workgroup holds every
  // channel of its 256 pixels): pw_out[pixel][j] = sum_c act(out[pixel][c]) * pw_w[c][j] + pw_bias[j], one fma chain per output
  // in channel order like conv_pw_kernel; `out` itself is NOT written.  The RGB head of the fusion decoder (fusion.py:138-140).
  const float* pw_w;      // [Cout][pw_cout]; nullptr: none
  const float* pw_bias;
  float* pw_out;
  int pw_ostride, pw_cout;   // pw_cout <= 4
};

// Flow head of a predictor with 32 filters (pyramid_flow_estimator.py:77-83): 1x1 conv Cin -> 16 + leaky_relu,
// then 1x1 conv 16 -> 2, fused: the 16-channel intermediate stays in registers.
struct FlowHeadParams {
  const float* in;
  int istride;
  int Cin;            // multiple of 4
  const float* w3;    // [Cin][16]
  const float* b3;
  const float* w4;    // [16][2]
  const float* b4;
  float* out;         // [M][2]
  int M;
  const float* add;   // optional [M][2]: the upsampled flow of the coarser level; then sum = out + add is stored too
  float* sum;         //          [M][2]  (v = residual + v, pyramid_flow_estimator.py:161 - the former flow_add launch)
};

// 1x1 convolution with a tiny output width (Cout <= 16): flow heads and the RGB head.
struct ConvPwParams {
  const float* in;
  int istride;
  int Cin;            // multiple of 4
  const float* w;     // [Cin][Cout]
  const float* bias;
  float* out;
  int ostride;
  int Cout;
  int leaky;
  int M;              // pixels
  const float* add;   // Cout == 2 only, optional [M][2]: sum = out + add is stored as well (see FlowHeadParams)
  float* sum;
};

// AveragePooling2D(2, 2, 'valid') on a channel slice (util.py:39-40, feature_extractor.py:138-139).
struct PoolParams {
  const float* in;
  int istride;
  float* out;
  int ostride;
  int C;
  int NB, H, W;       // INPUT spatial dims (even)
};

// tf.image.resize(2*v, 2x) bilinear, half-pixel centres (pyramid_flow_estimator.py:155).
struct FlowUpParams {
  const float* in;    // [NB][h][w][2]
  float* out;         // [NB][2h][2w][2]
  int NB, h, w;
};

// v = r + up  (pyramid_flow_estimator.py:161, util.py:114)
struct FlowAddParams {
  const float* a;
  const float* b;
  float* out;
  int64_t n;          // floats
};

// util.warp (util.py:48-82 -> tfa.image.dense_image_warp): backward bilinear gather with
// edge clamp.  flow is (dx,dy); the sampled position is (y + s*flow_y, x + s*flow_x).
struct WarpParams {
  const float* src;
  int sstride;
  int C;              // multiple of 4, or exactly 3
  const float* flow;  // [NB][H][W][2]
  float fscale;       // 1 in the flow estimator, 0.5 for the mid-frame warps (interpolator.py:163-165)
  float* dst;
  int dstride;
  int NB, H, W;
  // Fused tf.image.resize(2*v) of the flow estimator (pyramid_flow_estimator.py:155): when `coarse` is set the flow of
  // this level is bilinear-x2(2 * coarse) ([NB][H/2][W/2][2], same arithmetic as flow_up_kernel) instead of `flow`,
  // and it is written to flow_out [NB][H][W][2] (C % 4 == 0 launches only) - the former flow_up launch.
  const float* coarse;
  float* flow_out;
  // Fused sixteen miscellaneous channels of an aligned-pyramid level (interpolator.py:167-183 warps [image | features] as one
  // tensor and appends the two half flows): dst3[pix] = {warp(src3, 0.5 pack_b) 3, warp(src3b, 0.5 pack_f) 3, 0.5 pack_b 2,
  // 0.5 pack_f 2, 0 x 6}, pixel stride d3stride (multiple of 4, dst3 16-byte aligned), by one thread per pixel in workgroups behind the
  // feature workgroups of a row band - the former
  // warp_c3 x 2 + pack_flow launches.  src3 / src3b: the two images [NB][H][W][3] (pixel stride s3stride), pack_b / pack_f:
  // backward / forward flow [NB][H][W][2].  dst3 == nullptr: none.
  const float* src3;
  const float* src3b;
  int s3stride;
  float* dst3;
  int d3stride;
  const float* pack_b;
  const float* pack_f;
  // One launch for both flow directions / both images of a level (batch n = 0 .. NB - 1 of dst and flow_out): the source image of
  // batch n is (n + src_brot) % NB, its flow (n + flow_brot) % NB (`flow`, not `coarse`); the miscellaneous channels exist for the
  // first misc_nb batches only (0: for all NB).
  int src_brot = 0, flow_brot = 0, misc_nb = 0;
};

// Writes channels [6..15] of the 16-wide "misc" group of an aligned-pyramid level:
// backward_flow*0.5 (2), forward_flow*0.5 (2), zeros (6)  (interpolator.py:163-165,182-183).
struct PackFlowParams {
  const float* bflow;  // [B][H][W][2]
  const float* fflow;
  float* dst;          // first misc channel + 6
  int dstride;
  int64_t npix;
};

// Interpolator.__call__ data movement (eval/interpolator.py:30-63 _pad_to_align, :66-126 image_to_patches /
// patches_to_image, :192-206): frame [B][H][W][3] <-> tiles [B*bh*bw][TH][TW][3], tile n = (b, ty, tx) row-major,
// each patch (H/bh x W/bw) zero-padded to (TH, TW) with the patch at offset (oy, ox) = (pad_h//2, pad_w//2).
struct TileMapParams {
  const float* src;
  float* dst;
  int B, H, W;       // frame
  int bh, bw;        // blocks per frame
  int ph, pw;        // patch = H/bh x W/bw
  int TH, TW;        // padded tile
  int oy, ox;        // patch offset inside the tile
  int tile0, ntiles; // tiles [tile0, tile0 + ntiles) of the frame batch are in the tile buffer
};

// Tile id = shape index + CONV_TILE_XCD when the XCD-contiguous block mapping is used.
enum ConvTile { TILE_128x128 = 0, TILE_256x64 = 1, TILE_256x32 = 2, TILE_64x64 = 3, TILE_128x32 = 4,
                TILE_128x64 = 5, TILE_256x128 = 6 /* 8 waves, 4x2 */, TILE_SHAPES = 7,
                TILE_C3_DIRECT = 7 /* with CONV_TILE_C3: conv_c3_kernel (no LDS, weights in registers, conv_c3_impl.h) */, CONV_TILE_XCD = 16,
                CONV_TILE_C3 = 32 /* first-layer mode: 3-channel image input, [48][Cout] weights */,
                CONV_TILE_HALO = 64 /* conv_halo_kernel: shape index = HaloTile, weights [Cout][chunk][tap][16] */,
                CONV_TILE_WINO = 256 /* conv_wino_kernel (F(2,3) along x): shape index = WinoTile, weights
                                        [Cout][chunk of 8][nu*3+dy][8] */,
                CONV_TILE_SPLIT = 128 /* conv_halo_split_kernel (precision mode bf16x6): shape index = HaloTile,
                                         weights [Cout][chunk][tap][3 planes][16] bf16 */,
                CONV_TILE_X3 = 512 /* precision mode bf16x3.  With CONV_TILE_SPLIT: conv_halo_split_kernel on planes hi, mid with
                                      three products; with CONV_TILE_WINO: conv_winox3_kernel, shape index = WinoX3Tile,
                                      weights [Cout][chunk][dy][j][h][plane][16] bf16 */,
                CONV_TILE_F43 = 2048 /* with CONV_TILE_WINO: conv_wino43_kernel (F(4,3) along x, fp32): shape index = Wino43Tile,
                                        weights [Cout][chunk of 8][dy][nu 6][8] */,
                CONV_TILE_W2D = 8192 /* conv_wino2d_kernel (nested F(4,3)x x F(2,3)y, fp32): shape index = Wino2dTile, weights
                                        [Cout/32][chunk of 8][mu 4][nu 6][K half][32][4] */,
                CONV_TILE_EXT = 4096 /* conv_wino43_kernel: shape index = (tile & 15) + 16 */,
                CONV_TILE_FOLD4 = 16384 /* conv_fold4_kernel (nearest x2 upsample + 2x2 conv in its difference form, fp32): shape index =
                                           Fold4Tile, weights [Cout/32][chunk of 8][plane 4][K half][32][4] */,
                CONV_TILE_FOLDX3 = 1024 /* conv_foldx3_kernel (precision mode bf16x3, folded upsample + 2x2): shape index =
                                           FoldX3Tile, weights [Cout][chunk][9 (tap, phase) steps][plane][16] bf16 */ };
// conv_wino43_kernel tiles (CONV_TILE_WINO | CONV_TILE_F43): patch rows x 128 pixels x output channels, wave block TM x TN
// 0-2: 128-pixel patches, 8 waves, one workgroup per CU; 3-5 ("Q16"): 64-pixel patches (an MFMA row tile = two patch rows x 16
// quads), 4 waves, 72 / 54 KB of LDS -> two workgroups per CU.  Same k-ordered sums: the autotuner picks freely among them.
enum Wino43Tile { W43_4x64_T21 = 0, W43_4x64_T12 = 1, W43_4x32_T11 = 2, W43_Q16_4x64_T21 = 3, W43_Q16_4x64_T12 = 4, W43_Q16_4x32_T11 = 5,
                  W43_Q16_4x64_N1 = 6 /* a wave owns all six nu planes of one 32x32 tile: no epilogue exchange */,
                  /* the Q16 tiles with the activation loads requested two K chunks ahead (second register set) */
                  W43_Q16_4x64_T21_P2 = 7, W43_Q16_4x64_T12_P2 = 8, W43_Q16_4x32_T11_P2 = 9, W43_Q16_4x64_N1_P2 = 10,
                  /* the 32-channel Q16 tile with the weight fragments read straight from global memory (no weight ring: 36 KB of
                     LDS, <= 128 VGPRs -> FOUR workgroups per CU; the short K loops of the 32-channel layers are latency bound) */
                  W43_Q16_4x32_T11_BG = 11,
                  /* "Q8": 32-pixel patches, 8 rows (an MFMA row tile = FOUR patch rows x 8 quads): the same 256 pixels per
                     workgroup, no empty quads on the 15 * 2^k wide levels, 10 halo rows per 8 output rows; all with the
                     activation loads two chunks ahead */
                  W43_Q8_8x64_T21_P2 = 12, W43_Q8_8x64_N1_P2 = 13, W43_Q8_8x32_T11_BG = 14, W43_Q8_8x64_T12_P2 = 15,
                  /* ids >= 16 carry CONV_TILE_EXT in the tile id (shape = (tile & 15) + 16).  The 32-channel Q8 tile WITH the
                     weight ring: 48 KB of LDS, 152 VGPRs -> three workgroups per CU (flow level 0 conv_0: -7 % against the BG tile) */
                  W43_Q8_8x32_T11_P2 = 16 };
// the Wino43Tile shapes this build of the library instantiates (conv_igemm.hip): all of them with FILM_EXTRA_FAMILIES, seven without
static inline bool film_w43_shape_built(int sh) {
#ifdef FILM_EXTRA_FAMILIES
  return sh >= 0 && sh <= W43_Q8_8x32_T11_P2;
#else
  return sh == W43_Q16_4x64_T21_P2 || sh == W43_Q16_4x32_T11_P2 || sh == W43_Q16_4x64_N1_P2 || sh == W43_Q8_8x64_T21_P2 || sh == W43_Q8_8x64_N1_P2 ||
         sh == W43_Q8_8x32_T11_BG || sh == W43_Q8_8x32_T11_P2;
#endif
}
// conv_wino2d_kernel tiles (CONV_TILE_W2D): one 32-unit MFMA tile (unit = 2 rows x 4 pixels; 8 rows x 32 pixels) x output channels;
// 64 channels = 8 waves (one workgroup per CU), 32 channels = 4 waves (two per CU).  Same sums: the autotuner picks freely.
// W2D_8x32_S2 (round 6): the 32-channel tile on TWO DMA stages instead of three (48 KB of LDS; all requests of a super-chunk in the gaps of ONE
// chunk, one super-chunk less requested in the prologue): 4-15 % faster on every layer with K <= 208, equal above (profiles/r06_w2d_chain_ns2.log).
// W2D_16x* (round 6): the 32 units as 8 unit rows x 4 units = 16 x 16 pixels (W2D_F_SQ, conv_wino2d_impl.h).  Offered where it pads a level no more than the 8 x 32
// arrangement: a 144x240 level tiles exactly (8 x 32: 7.5 tiles per row) - 5-7.5 % faster there, and the autotuner also picks it on some exact-fit levels
// (profiles/r06_w2d_square_tile.log).
enum Wino2dTile { W2D_8x64 = 0, W2D_8x32 = 1, W2D_8x32_S2 = 2, W2D_16x64 = 3, W2D_16x32 = 4, W2D_16x32_S2 = 5, W2D_SHAPES = 6 };
inline bool film_w2d_square(int shape) { return shape >= W2D_16x64 && shape <= W2D_16x32_S2; }
inline bool film_w2d_64(int shape) { return shape == W2D_8x64 || shape == W2D_16x64; }
// conv_fold4_kernel tiles (CONV_TILE_FOLD4): 4 rows x 32 low-resolution pixels x output channels; four waves side by side, each 4 x 8 pixels
// x all channels of the tile.  64 channels: 235 VGPRs, two workgroups per CU; 32 channels: 139 VGPRs, three.  Same sums.
enum Fold4Tile { F4_4x64 = 0, F4_4x32 = 1 };
// conv_foldx3_kernel tiles (CONV_TILE_FOLDX3): low-resolution patch rows x 32 pixels x output channels (waves M x N)
enum FoldX3Tile { FX3_4x64 = 0 /* 4x1 */, FX3_8x64 = 1 /* 8x1 */, FX3_4x128 = 2 /* 4x2 */ };
// conv_winox3_kernel tiles (CONV_TILE_WINO | CONV_TILE_X3): patch rows x 64 pixels x output channels, wave block TM x TN
enum WinoX3Tile { WX3_4x128_T22 = 0, WX3_4x64_T12 = 1, WX3_4x64_T21 = 3, WX3_4x32_T11 = 4 };   // all 8 waves
// conv_wino_kernel tiles: patch rows x 64 pixels x output channels (waves M x N)
enum WinoTile { WINO_4x128 = 0 /* 4x2 */, WINO_4x64 = 1 /* 4x1 */, WINO_4x128_W16 = 2 /* 4x4: 16 waves */,
                WINO_4x64_W8 = 3 /* 4x2: 32 channels per wave */, WINO_4x32 = 4 /* 4x1 */, WINO_8x64_W16 = 5 /* 8x2 */,
                WINO_8x32_W8 = 6 /* 8x1 */, WINO_2x64 = 7 /* 2x2 */, WINO_SHAPES = 8 };
// conv_halo_kernel tiles: patch rows x 32 pixels x output channels (waves M x N)
enum HaloTile { HALO_8x128 = 0 /* 4x2 */, HALO_8x64 = 1 /* 4x1 */, HALO_8x32 = 2 /* 4x1 */, HALO_4x64 = 3 /* 4x1 */,
                HALO_4x128 = 4 /* 2x2 */, HALO_4x32 = 5 /* 4x1 */, HALO_SHAPES = 6 };

struct TileShape { int bm, bn; };
static inline TileShape film_tile_shape(int tile) {
  switch (tile & (CONV_TILE_XCD - 1)) {
    case TILE_128x128: return {128, 128};
    case TILE_256x64: return {256, 64};
    case TILE_256x32: return {256, 32};
    case TILE_64x64: return {64, 64};
    case TILE_128x64: return {128, 64};
    case TILE_256x128: return {256, 128};
    default: return {128, 32};
  }
}

hipError_t film_launch_conv(const ConvParams& p, int tile, hipStream_t s);
hipError_t film_launch_conv_pw(const ConvPwParams& p, hipStream_t s);
hipError_t film_launch_flow_head(const FlowHeadParams& p, hipStream_t s);
hipError_t film_launch_pool(const PoolParams& p, hipStream_t s);
hipError_t film_launch_flow_up(const FlowUpParams& p, hipStream_t s);
hipError_t film_launch_flow_add(const FlowAddParams& p, hipStream_t s);
hipError_t film_launch_warp(const WarpParams& p, hipStream_t s);
hipError_t film_launch_pack_flow(const PackFlowParams& p, hipStream_t s);
hipError_t film_launch_frame_to_tiles(const TileMapParams& p, hipStream_t s);   // pad + image_to_patches
hipError_t film_launch_tiles_to_frame(const TileMapParams& p, hipStream_t s);   // crop + patches_to_image
// write_image's rounding on the device: dst[i] = uint8(clip(src[i] * 255, 0, 255) + 0.5)  (eval/util.py:51-52)
hipError_t film_launch_to_uint8(const float* src, uint8_t* dst, int64_t n, hipStream_t s);
// fills n floats with a deterministic pseudo-random pattern in [-1, 1) (autotune inputs only)
hipError_t film_launch_fill_random(float* dst, int64_t n, uint32_t seed, hipStream_t s);
