// film_planner.cpp -- the planner: restates the *graph* of models/film_net/interpolator.py:89-207 as a static list of kernel
// launches over one workspace arena for a given (B,H,W):
//   image pyramids     util.py:23-45                -> pool ops on [2B,...,3] (both images in one batch)
//   feature extractor  feature_extractor.py:163-193 -> conv ops writing straight into the cascaded slots
//   flow estimator     pyramid_flow_estimator.py:125-163 -> both directions batched as 2B
//   flow synthesis     util.py:106-117              -> reuses the estimator's v (identical arithmetic)
//   warps + concat     interpolator.py:163-183      -> warp ops writing into the aligned pyramid
//   fusion             fusion.py:103-140            -> NN-upsample folded into the 2x2 conv's gather
// decides the kernel family of every convolution from its shape and the options (never from timing or the batch size),
// derives the cross-lane dependencies of the two-stream replay, and renders a plan as JSON (film_plan_json).
#include "film_internal.h"

namespace film_internal {

const char* const kKindName[OP_KINDS] = {"conv_mfma", "flow_head", "conv_pw", "pool", "flow_up", "flow_add", "warp", "pack_flow"};

namespace {
// ---------------------------------------------------------------------------------------------
// Planner
// ---------------------------------------------------------------------------------------------
struct Planner {
  film_t* h;
  Plan* P;
  int64_t cursor = 0;

  bool bad = false;
  std::string bad_msg;

  int add_buffer(const std::string& name, int N, int H, int W, int C) {
    Buffer b{name, cursor, N, H, W, C, (int64_t)N * H * W * C};
    cursor += (b.floats + 63) & ~int64_t(63);  // 256-byte alignment
    P->bufs.push_back(b);
    return (int)P->bufs.size() - 1;
  }
  int add_scratch(const std::string& name, int64_t floats) {
    Buffer b{name, cursor, 0, 0, 0, 0, std::max<int64_t>(floats, 64)};
    cursor += (b.floats + 63) & ~int64_t(63);
    P->bufs.push_back(b);
    return (int)P->bufs.size() - 1;
  }
  // view of channels [coff, coff+C) of batches [batch0, ...) of a buffer
  View view(int buf, int batch0, int coff, int C) const {
    const Buffer& b = P->bufs[buf];
    View v;
    v.buf = buf;
    v.off = b.off + (int64_t)batch0 * b.H * b.W * b.C + coff;
    v.stride = b.C;
    v.C = C;
    return v;
  }
  // Part 0 / 1 / 2 of an aligned-pyramid level = warp(features of image 0) / warp(features of image 1) / the sixteen
  // miscellaneous channels.  Interleaved buffer: channel slices of [N][H][W][2C + 16]; planar buffer: three pixel-major planes.
  View aligned_part(int buf, int part, int C) const {
    const Buffer& b = P->bufs[buf];
    if (!b.planar) return view(buf, 0, part * C, part < 2 ? C : 16);
    View v;
    v.buf = buf;
    v.off = b.off + (int64_t)part * b.N * b.H * b.W * C;
    v.stride = v.C = part < 2 ? C : 16;
    return v;
  }
  static View sub(View v, int coff, int C) { v.off += coff; v.C = C; return v; }
  // scratch view: reinterpret the start of a scratch buffer as [*][*][*][C]
  View scratch(int buf, int C) const {
    View v;
    v.buf = buf; v.off = P->bufs[buf].off; v.stride = C; v.C = C;
    return v;
  }

  static int choose_tile(int64_t M, int Cout) {
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (Cout / bn); };
    int shape;
    if (Cout % 128 == 0) shape = blocks(128, 128) >= 768 ? TILE_128x128 : TILE_64x64;
    else if (Cout % 64 == 0) shape = blocks(256, 64) >= 768 ? TILE_256x64 : TILE_64x64;
    else shape = blocks(256, 32) >= 768 ? TILE_256x32 : TILE_128x32;
    return shape | CONV_TILE_XCD;
  }

  static int choose_halo_tile(int Cout) {
    return (Cout % 64 == 0 ? HALO_4x64 : HALO_8x32) | CONV_TILE_HALO | CONV_TILE_XCD;
  }

  // the weight layout group a kernel family reads must be packed (and on the device) before the plan can run
  void need_groups(int n) {
    if (!h->finalized || n <= h->groups_packed) return;
    if (film_ensure_groups_(h, n) != FILM_OK) { bad = true; bad_msg = h->err; }
  }

  void conv(const std::string& tag, const std::string& layer, std::vector<SegDesc> segs, View out, int NB, int H,
            int W, bool leaky) {
    const LayerPack& L = h->layers[h->layer_idx.at(layer)];
    OpDesc op;
    op.kind = OP_CONV;
    op.tag = tag + ":" + layer;
    op.nseg = (int)segs.size();
    int ctot = 0;
    for (int i = 0; i < op.nseg; ++i) { op.seg[i] = segs[i]; ctot += segs[i].v.C; }
    op.ksize = L.kh; op.leaky = leaky; op.Cout = L.cout; op.Ctot = ctot;
    if (ctot != L.ctot() || out.C != L.cout || op.nseg > FILM_MAX_SEG || !L.kmajor()) {
      bad = true;
      bad_msg = "planner: channel mismatch at " + op.tag;
    }
    op.w_off = L.w_off; op.b_off = L.b_off; op.wh_off = L.wh_off; op.ws_off = L.ws_off; op.ww_off = L.ww_off; op.wx_off = L.wx_off; op.wfx_off = L.wfx_off; op.wf4_off = L.wf4_off; op.w43_off = L.w43_off; op.w2d_off = L.w2d_off;
    if (h->opt_fold && L.wf_off >= 0 && op.nseg == 1 && segs[0].up && !(H & 1) && !(W & 1)) {
      // nearest x2 + 2x2 'same' conv == four phase convolutions on the low-resolution input: kernel tap (dy, dx) of
      // output (2y+py, 2x+px) reads input ((2y+py+dy)>>1, (2x+px+dx)>>1) = (y + (py&dy), x + (px&dx)), so phase
      // (0,0) has ONE distinct input pixel, (0,1) and (1,0) two, (1,1) four: 9 taps per 4 outputs instead of 16.
      // The weights of taps that read the same pixel are summed at film_finalize (exact regrouping of the sum).
      // One launch runs the four phases (blockIdx.z): 4x the blocks of a phase launch, better tails.
      OpDesc f = op;
      f.tag = op.tag + ":phases";
      f.fold = 2;
      f.seg[0].up = 0;
      f.out = out; f.NB = NB;
      f.H = H / 2; f.W = W / 2;
      f.w_off = L.wf_off;
      int64_t rel = 0;
      for (int q = 0; q < 4; ++q) { f.fold_woff[q] = rel; rel += (int64_t)((q >> 1) + 1) * ((q & 1) + 1) * ctot * L.cout; }
      f.halo = f.split = f.wino = 0;
      f.tile = choose_tile((int64_t)NB * f.H * f.W * 2, L.cout);
      if (h->opt_precision == 2 && L.wfx_off >= 0 && L.cout % 64 == 0 && ((int64_t)f.H * f.W >= 2048 || h->opt_halo_all)) {
        f.split = 2;   // precision mode bf16x3: one halo-staged patch, nine (tap, phase) steps (conv_foldx3_kernel)
        f.tile = FX3_4x64 | CONV_TILE_FOLDX3 | CONV_TILE_XCD;
        need_groups(4);
      }
      if (f.split == 0 && h->opt_fold4 && h->opt_precision == 0 && L.wf4_off >= 0 && segs[0].v.stride % 4 == 0 && segs[0].v.off % 4 == 0) {
        // the difference form (conv_fold4_impl.h): four GEMMs over K on I, Dx, Dy, Dxy - 4 multiplies per low-resolution pixel, not 9.
        // Its own summation family, chosen by the layer alone (every folded layer whose channels come in sixteens)
        f.fold = 3;
        f.w_off = L.wf4_off;
        for (int q = 0; q < 4; ++q) f.fold_woff[q] = 0;
        f.tile = (L.cout % 64 == 0 ? F4_4x64 : F4_4x32) | CONV_TILE_FOLD4 | CONV_TILE_XCD;
        // split-K like the nested kernel's (below), for deep K on a TINY level: the decoder's coarsest layer of a 256x256 pair (K = 1936 on
        // 16x16 low-resolution pixels = 4 pixel tiles x 16 channel blocks: 64 workgroups of 242 chunks) 0.139 -> 0.048 ms with four K
        // ranges; on the 36x60 level of a 1080p tile (1152 workgroups) every split LOSES (0.589 -> 0.61-0.64 ms), eight 448x256 pairs
        // (16x28) lose 10 % (tools/fold4_bench, profiles/r06_fold4_bench.log).  Factor from the level size and the layer only - never the batch.
        if (h->opt_splitk && (int64_t)f.H * f.W <= 1024 && ctot >= 768) {
          f.ksplit = 4;
          const int sb = add_scratch("splitk:" + f.tag, (int64_t)f.ksplit * NB * H * W * L.cout);
          f.part_off = P->bufs[sb].off;
        }
      }
      f.flops = 2.0 * NB * H * W * L.cout * L.kh * L.kw * L.cin;   // algorithmic FLOPs of the reference op
      f.bytes = 4.0 * NB * H * W * (L.cin / 4.0 + L.cout);
      P->ops.push_back(f);
      return;
    }
    op.out = out; op.NB = NB; op.H = H; op.W = W;
    // every MFMA convolution stores dwordx4 (four channels of a pixel per lane): the destination slice must start on a multiple of
    // four floats of a pixel whose pitch is one.  film_create only accepts configurations (filters % 32 == 0, flow filters 32 or
    // % 64) for which every slice does; a layout that breaks this must fail HERE with a message, not as hipErrorInvalidValue at
    // launch time (round-4 ADVICE)
    if (out.off % 4 != 0 || out.stride % 4 != 0 || L.cout % 32 != 0) {
      bad = true;
      bad_msg = "planner: " + op.tag + " writes " + std::to_string(L.cout) + " channels at float offset " + std::to_string(out.off) + " of a " +
                std::to_string(out.stride) + "-float pixel: the convolution kernels need Cout % 32 == 0 and 16-byte aligned output slices";
    }
    const int64_t M = (int64_t)NB * H * W;
    // Kernel family by layer shape only (never by timing, and not by the batch size): the two kernels sum K in a
    // different order, so the choice must be a pure function of the layer for results to be reproducible across
    // batch sizes and runs.  Halo staging pays where K is deep (traffic bound) or N is too narrow to amortise the
    // per-tap A gather; measured in tools/retired/conv_bench.hip.
    bool any_up = false;
    for (int i = 0; i < op.nseg; ++i) any_up |= segs[i].up != 0;
    const int64_t px = (int64_t)H * W;
    if (h->opt_halo_all)  // tuning / test knob: every eligible 3x3 conv, whatever its size
      op.halo = L.has_halo() && !any_up;
    else
      op.halo = L.has_halo() && !any_up && px >= 8192 &&
                (ctot >= 768 || (ctot >= 512 && px >= 100000) || L.cout == 32);
    // precision mode bf16x6: every 3x3 conv that is large enough to be matrix-pipe bound
    // (op.split: 1 = bf16x6, 2 = bf16x3 - same kernel, two planes and three products)
    op.split = (h->opt_precision != 0 && L.has_halo() && !any_up && (px >= 2048 || h->opt_halo_all)) ? h->opt_precision : 0;
    // Winograd F(2,3) along x: where the 1.5x MFMA saving survives its LDS / occupancy cost - wide N, large M.
    // In precision mode bf16x3 the same layers run the Winograd form of the split kernel (conv_winox3_kernel).
    // F(4,3) needs the level width to fill its 64-pixel patches (at most 15 % of the last patch of a row empty)
    // (levels below 2048 pixels: its Q8 tiles take 32-pixel patches - the 32x32 level of a 256x256 frame)
    const bool w43_width = L.w43_off >= 0 && (h->opt_wino == 3 || 64 * ((W + 63) / 64) * 100 <= 115 * W ||
                                              (px < 2048 && 32 * ((W + 31) / 32) * 100 <= 115 * W));
    // deep K on a small level (the 36x60 level of a 1080p tile, K = 1920; round 4: down to 1024 pixels - the K = 1920 / 2448 layers of
    // the 32x32 level of a 256x256 frame ran 180-230 us each on the direct kernel, 0.4 of that frame's 2.8 ms): only with F(4,3) AND
    // its split-K, which cuts the few long workgroups of such a layer into enough pieces to fill the chip
    const bool deep_small = L.cout % 128 == 0 && px >= 1024 && px < 8192 && ctot > 1024 && w43_width && h->opt_splitk &&
                            h->opt_wino == 1 && h->opt_precision == 0;
    op.wino = op.split != 1 && L.ww_off >= 0 && !any_up && h->opt_wino != 0 &&
              ((L.cout % 128 == 0 && (px >= 8192 || (px >= 2048 && ctot <= 1024))) || (L.cout % 64 == 0 && px >= 30000) ||
               px >= 100000 || h->opt_wino >= 2 || deep_small);
    if (op.wino && h->opt_precision == 2 && (op.split == 2 || h->opt_wino >= 2)) {
      // wino: 1 = fp32 conv_wino_kernel, 2 = conv_winox3_kernel.  The Winograd form wins with the 2 x 2 wave block of
      // its 128-channel tile (0.88-0.94x the time of conv_halo_split_kernel<..,3> per layer, 427 vs 367 TFLOP/s at
      // K = 22 032) and loses with the 64-channel tiles (1.08-1.30x: twice the A staging per MFMA) - per-op profiles of
      // the two plans and tools/retired/conv_bench.hip agree.
      if (L.cout % 128 == 0 || h->opt_wino >= 2) op.split = 0, op.wino = 2;
      else if (op.split == 2) op.wino = 0;
    }
    // fp32: F(4,3) along x (conv_wino43_kernel, 2x fewer MFMAs than direct where F(2,3) has 1.5x) on the levels whose width
    // fills its 64-pixel patches (the Q16 tiles; at most 15 % of the last patch of a row empty: 960 ... 60, 448, 256 ...); wino = 3.  "winograd" = 2 / 3 force
    // F(2,3) / F(4,3) onto every eligible layer (tests).
    if (op.wino == 1 && h->opt_wino != 2 && w43_width) op.wino = 3;
    // Nested F(4,3)x x F(2,3)y (conv_wino2d_kernel, 1.5x fewer MFMAs than the 1-D form): round 4 - EVERY 3x3 layer whose channels come
    // in sixteens (K = 32 ... 2448) on levels of at least opt_w2d_min_px pixels per image (default 1536: eight or more of its 8x32
    // patches per image; below that the 1-D kernel's split-K fills the chip better at batch 1).  A family of its own, a function of the
    // layer and the level size only.  A/B over the BASELINE configs: profiles/r04_w2d_min_px_ab.log.
    // conv_wino2d_kernel moves the halo patch by DMA in 16-channel pieces and stores four channels per lane: every input segment
    // a multiple of 16 channels at 16-byte aligned pixels, a 16-byte aligned output slice (true of every such layer of the
    // published net; a property of the layer's buffers, so still a function of the layer only)
    bool w2d_layout = ctot % 16 == 0 && out.off % 4 == 0 && out.stride % 4 == 0;
    // (round 6: ... and a halo row of a segment, W pixels x the pixel pitch, below 2^24 bytes - the DMA offsets are two 24-bit multiply-adds)
    for (int i = 0; i < op.nseg; ++i)
      w2d_layout = w2d_layout && segs[i].v.C % 16 == 0 && segs[i].v.off % 4 == 0 && segs[i].v.stride % 4 == 0 && (int64_t)W * segs[i].v.stride * 4 < (1ll << 24);
    // Round 6: ... and on SMALLER levels (>= opt_w2d_small_px = 256 pixels) whose shape fills at least 65 % of its 8-row x 32-pixel tiles:
    // 18x30 (a 1080p tile's level 5: 70 %), 32x32 (a 256x256 pair: 100 %), 16x28 (a 448x256 pair: 87 %) yes, 16x16 (50 %) and 9x15 (26 %)
    // no - its split-K (round 5) fills the chip where the tile count does not.  A/B of the threshold over three configs:
    // profiles/r06_w2d_small_levels_ab.log (256x256 2.43 -> 2.35 ms, eight 448x256 pairs 15.7 -> 15.1, 1080p -0.1 ms).
    const int64_t w2d_tile_px = (int64_t)((H + 7) / 8) * 8 * ((W + 31) / 32) * 32;
    const bool w2d_level = px >= h->opt_w2d_min_px || (h->opt_w2d_small_px > 0 && px >= h->opt_w2d_small_px && px * 100 >= 65 * w2d_tile_px);
    if (L.w2d_off >= 0 && w2d_layout && !any_up && h->opt_precision == 0 && op.split == 0 &&
        (h->opt_wino2d == 2 || (h->opt_wino2d == 1 && h->opt_wino == 1 && w2d_level)))
      op.wino = 4;
    if (op.split || op.wino) op.halo = 0;
#ifndef FILM_EXTRA_FAMILIES
    // default build: the halo / F(2,3) / bf16 split kernels are not in the library (film_set_option refuses the options that ask for
    // them); a layer the nested / F(4,3) kernels cannot take runs on the general kernel
    op.halo = 0; op.split = 0;
    if (op.wino == 1 || op.wino == 2) op.wino = 0;
#endif
    need_groups(op.split || op.wino == 2 ? 4 : op.halo ? 3 : op.wino == 1 ? 2 : 1);
    op.tile = op.wino == 4 ? ((L.cout % 64 == 0 ? W2D_8x64 : W2D_8x32) | CONV_TILE_W2D | CONV_TILE_XCD)
              : op.wino == 3 ? ((L.cout % 64 == 0 ? W43_Q16_4x64_T21_P2 : W43_Q16_4x32_T11_P2) | CONV_TILE_WINO | CONV_TILE_F43 | CONV_TILE_XCD)
              : op.wino == 2 ? ((L.cout % 128 == 0 ? WX3_4x128_T22 : L.cout % 64 == 0 ? WX3_4x64_T12 : WX3_4x32_T11) | CONV_TILE_WINO | CONV_TILE_X3 | CONV_TILE_XCD)
              : op.wino ? ((L.cout % 64 == 0 ? WINO_4x64_W8 : WINO_4x32) | CONV_TILE_WINO | CONV_TILE_XCD)
              : op.split ? ((L.cout % 128 == 0 ? HALO_8x128 : L.cout % 64 == 0 ? HALO_4x64 : HALO_8x32) | CONV_TILE_SPLIT | (op.split == 2 ? CONV_TILE_X3 : 0) | CONV_TILE_XCD)
              : op.halo ? choose_halo_tile(L.cout) : choose_tile(M, L.cout);
    // Split-K for the deep layers of the coarse levels: the whole K loop (up to 1377 steps) of such a layer otherwise runs
    // on a handful of workgroups and IS the latency of the level (0.39 ms per flow-predictor conv_0 at 16 pixels).  The
    // factor depends on the per-image pixel count and the layer only - never on the batch - so results stay independent
    // of the batch size; partial sums are added in split order (no atomics).
    if (h->opt_splitk && !op.halo && !op.split && !op.wino && !op.c3 && L.kmajor() && px <= 4096) {
      const int nsteps = L.kh * L.kw * ctot / 16;
      // shallow layers: the extra launch costs more than it saves
      int S = nsteps < 128 ? 1 : px <= 64 ? 16 : px <= 256 ? 8 : px <= 1024 ? 4 : nsteps >= 256 ? 2 : 1;
      while (S > 1 && nsteps / S < 32) S >>= 1;
      if (S > 1 && L.cout % 4 == 0) {
        op.ksplit = S;
        const int sb = add_scratch("splitk:" + op.tag, (int64_t)S * M * L.cout);
        op.part_off = P->bufs[sb].off;
      }
    }
    // Split-K for the F(4,3) kernel on levels whose workgroup count does not fill the 512 workgroup slots of the chip
    // evenly: a 72x120 level with 512 output channels is 1152 workgroups = 2.25 rounds, the last one on a quarter of
    // the CUs for the full duration of a deep K loop.  Two K ranges double the workgroup count at half the length; the partial sums are added in split order by conv_splitk_reduce_kernel.  Factor from
    // the level size and the layer only.
    if (h->opt_splitk && op.wino == 3 && L.cout % 4 == 0) {
      // 72x120 level, measured (profiles/r02_per_op_profile.json vs the run before): -10 % on the K = 2448 / 1920 layers,
      // +7..10 % on its K <= 512 layers (reduce kernel + twice the prologues / epilogues) -> deep K only
      // small levels (<= 4096 pixels per image: the 36x60 level of a 1080p tile, the 64x64 level of a 256x256 frame) have
      // 9-16 patches per image and channel block: K ranges of >= 128 channels, up to 8 of them (A/B on the GPU: 36x60 level
      // 2.90 -> 2.31 ms per 1080p step incl. its K = 1920 layer moving here from conv_buf_kernel; 64x64 level of a 256x256
      // pair 1.05 -> 0.86 ms; K < 512 left alone - the split would also undo the fused pooling of those layers)
      int S = 1;
      if (px <= 4096 && ctot >= 512) S = std::min(8, ctot / 128);
      else if (px <= 16384 && ctot >= 1024) S = 2;
      if (S > 1) {
        op.ksplit = S;
        const int sb = add_scratch("splitk:" + op.tag, (int64_t)S * M * L.cout);
        op.part_off = P->bufs[sb].off;
      }
    }
    // Split-K for the nested kernel (round 5) on the SMALL levels only (<= 4096 pixels per image: the 36x60 level of a 1080p tile - 640
    // workgroups of its K = 1920 layer on 512 slots - and the 64x64 level of a 256x256 pair, 64-128 workgroups): K >= 768, up to four
    // K ranges.  Measured (profiles/r05_w2d_splitk.log): K = 1920 @ 8x36x60 0.653 -> 0.531 ms (0.61 -> 0.54 in the forward), K = 1168 @
    // 1x64x64 0.152 -> 0.084 ms, 256x256 pair 2.91 -> 2.82 ms.  NOT on the 72x120 level: its K = 1920 / 2448 layers are 2304 workgroups
    // = 4.5 rounds of the 512 slots and two K ranges gain 2-4 % stand-alone, but nothing in the forward (1.84 -> 1.89, 2.48 -> 2.49-2.53
    // ms per layer with the weights streaming from HBM and the partial sums written and read back: 141 MB for the K = 2448 layer).
    // No layer with a fused pool / 1x1 head is that deep.  Factor from the level size and the layer only - never the batch.
    if (h->opt_splitk && h->opt_w2d_splitk && op.wino == 4 && L.cout % 4 == 0) {
      int S = 1;
      if (px <= 4096 && ctot >= 768) S = std::min(4, ctot / 384);
      if (h->opt_w2d_splitk > 1 && px <= 1024 && ctot >= 384) S = std::min(h->opt_w2d_splitk, ctot / 192);   // A/B knob: a higher cap on tiny levels
      if (S > 1) {
        op.ksplit = S;
        const int sb = add_scratch("splitk:" + op.tag, (int64_t)S * M * L.cout);
        op.part_off = P->bufs[sb].off;
      }
    }
    op.flops = 2.0 * M * L.cout * L.kh * L.kw * L.cin;
    op.bytes = 4.0 * M * (L.cin + L.cout);
    P->ops.push_back(op);
  }
  void conv_pw(const std::string& tag, const std::string& layer, View in, View out, int64_t M, bool leaky) {
    const LayerPack& L = h->layers[h->layer_idx.at(layer)];
    OpDesc op;
    op.kind = OP_CONV_PW; op.tag = tag + ":" + layer;
    op.in = in; op.out = out; op.n = M; op.leaky = leaky; op.Cout = L.cout; op.Ctot = L.cin;
    op.w_off = L.w_off; op.b_off = L.b_off;
    op.flops = 2.0 * M * L.cout * L.cin; op.bytes = 4.0 * M * (L.cin + L.cout);
    P->ops.push_back(op);
  }
  void pool(const std::string& tag, View in, View out, int NB, int H, int W) {
    OpDesc op;
    op.kind = OP_POOL; op.tag = tag; op.in = in; op.out = out; op.NB = NB; op.H = H; op.W = W;
    op.bytes = 4.0 * NB * H * W * in.C * 1.25;
    P->ops.push_back(op);
  }
  void warp(const std::string& tag, View src, View flow, View dst, int NB, int H, int W, float fscale,
            bool count_flow = true) {
    OpDesc op;
    op.kind = OP_WARP; op.tag = tag; op.in = src; op.in2 = flow; op.out = dst;
    op.NB = NB; op.H = H; op.W = W; op.fscale = fscale;
    // read source once + flow, write once.  The image part of a [image|features] warp is
    // a second launch here; its re-read of the flow is not algorithmic traffic.
    op.bytes = 4.0 * NB * H * W * (2.0 * src.C + (count_flow ? 2 : 0));
    P->ops.push_back(op);
  }

  int build(int B, int H, int W) {
    const film_config& c = h->cfg;
    const int L = c.pyramid_levels, FL = c.fusion_pyramid_levels;
    const int N2 = 2 * B;
    auto fc = feature_channels(c);
    auto ff = fusion_filters(c);
    auto HL = [&](int l) { return H >> l; };
    auto WL = [&](int l) { return W >> l; };
    P->B = B; P->H = H; P->W = W;

    // ---- buffers ---------------------------------------------------------------------------
    std::vector<int> img(L), feat(L), res(L), v(L), vup(L), warped(L), aligned(FL), fu_u(FL), fu_a(FL), fu_b(FL);
    for (int l = 0; l < L; ++l) img[l] = add_buffer("img" + std::to_string(l), N2, HL(l), WL(l), 3);
    for (int l = 0; l < L; ++l) feat[l] = add_buffer("feat" + std::to_string(l), N2, HL(l), WL(l), fc[l]);
    // feature-extractor scratch: stage-j conv_2j output and pooled input, sized for the largest use
    int64_t fx_sz = 0, fxp_sz = 0, fp_sz = 0;
    for (int i = 0; i < L; ++i)
      for (int j = 0; j < std::min(L - i, c.sub_levels); ++j) {
        fx_sz = std::max<int64_t>(fx_sz, (int64_t)N2 * HL(i + j) * WL(i + j) * (c.filters << j));
        if (j + 1 < std::min(L - i, c.sub_levels))
          fxp_sz = std::max<int64_t>(fxp_sz, (int64_t)N2 * HL(i + j + 1) * WL(i + j + 1) * (c.filters << j));
      }
    // one scratch pair per pyramid level's subtree: the subtrees are independent chains (two of them run beside the
    // rest on the side stream of the replay graph) and 288 GB of HBM makes sharing pointless
    std::vector<int> fx_a_v(L), fx_p_v(L);
    for (int i = 0; i < L; ++i) {
      int64_t a_sz = 0, p_sz = 0;
      for (int j = 0; j < std::min(L - i, c.sub_levels); ++j) {
        a_sz = std::max<int64_t>(a_sz, (int64_t)N2 * HL(i + j) * WL(i + j) * (c.filters << j));
        if (j + 1 < std::min(L - i, c.sub_levels))
          p_sz = std::max<int64_t>(p_sz, (int64_t)N2 * HL(i + j + 1) * WL(i + j + 1) * (c.filters << j));
      }
      fx_a_v[i] = add_scratch("scratch_fx_a" + std::to_string(i), a_sz);
      fx_p_v[i] = add_scratch("scratch_fx_p" + std::to_string(i), p_sz);
    }
    (void)fx_sz; (void)fxp_sz;
    for (int l = 0; l < L; ++l) {
      const int nf = c.flow_filters[predictor_index(c, l)];
      fp_sz = std::max<int64_t>(fp_sz, (int64_t)N2 * HL(l) * WL(l) * nf);
    }
    int fp[3];
    for (int k = 0; k < 3; ++k) fp[k] = add_scratch("scratch_fp_" + std::to_string(k), fp_sz);
    for (int l = 0; l < L; ++l) res[l] = add_buffer("res" + std::to_string(l), N2, HL(l), WL(l), 2);
    for (int l = 0; l < L - 1; ++l) {
      vup[l] = add_buffer("vup" + std::to_string(l), N2, HL(l), WL(l), 2);
      v[l] = add_buffer("v" + std::to_string(l), N2, HL(l), WL(l), 2);
      warped[l] = add_buffer("warped" + std::to_string(l), N2, HL(l), WL(l), fc[l]);
    }
    v[L - 1] = res[L - 1];  // coarsest: the DC term is the flow itself (pyramid_flow_estimator.py:149-150)
    // Planar aligned levels: the two feature warps and the miscellaneous channels of a level each write ONE contiguous plane
    // (256 ... 3840 contiguous bytes per pixel and launch) instead of 256-byte pieces at the 576-byte ... 7.7-KB pixel pitch of an
    // interleaved [feat0 | feat1 | misc16] pixel - the t = 0.5 warps then run like the flow-estimator warps of the same size
    // (0.28 -> 0.20 ms on the 576x960x64 level) - and the decoder reads the planes as three input segments in the same channel
    // order (same weights, same sums).  The coarsest fusion level stays interleaved: its only reader is the folded 2x2 layer,
    // which takes one segment.
    for (int l = 0; l < FL; ++l) {
      aligned[l] = add_buffer("aligned" + std::to_string(l), B, HL(l), WL(l), 2 * fc[l] + 16);
      if (h->opt_planar && l < FL - 1 && fc[l] % 16 == 0) P->bufs[aligned[l]].planar = fc[l];
    }
    for (int i = 0; i < FL - 1; ++i) {
      fu_u[i] = add_buffer("fusion_up" + std::to_string(i), B, HL(i), WL(i), ff[i]);
      fu_a[i] = add_buffer("fusion_a" + std::to_string(i), B, HL(i), WL(i), ff[i]);
      fu_b[i] = add_buffer("fusion_b" + std::to_string(i), B, HL(i), WL(i), ff[i]);
    }
    const int out = add_buffer("out", B, H, W, 3);
    P->arena_floats = cursor;

    // ---- image pyramids (util.py:23-45), both images as one batch of 2B ----------------------
    for (int l = 0; l + 1 < L; ++l) {
      pool("image_pyramid_l" + std::to_string(l + 1), view(img[l], 0, 0, 3), view(img[l + 1], 0, 0, 3), N2, HL(l), WL(l));
      // on the side stream: the level-0 subtree (main stream) reads img[0] only and starts at once; six 6-us launches less in front of it
      // (256x256: 2.31 -> 2.28 ms per step, 1080p: -0.1 ms; profiles/r06_pool_lane_ab.log)
      P->ops.back().lane = 1;
    }

    // ---- cascaded feature extractor (feature_extractor.py:163-193) --------------------------------
    for (int i = 0; i < L; ++i) {
      const int n = std::min(L - i, c.sub_levels);
      const int fx_a = fx_a_v[i], fx_p = fx_p_v[i];
      const size_t first_op = P->ops.size();
      for (int j = 0; j < n; ++j) {
        const int lv = i + j, k = c.filters << j;
        const std::string tg = "feat_s" + std::to_string(i) + "_" + std::to_string(j);
        const std::string w0 = "feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * j);
        const std::string w1 = "feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * j + 1);
        View tmp = scratch(fx_a, k);
        if (j == 0) {
          const LayerPack& Lp = h->layers[h->layer_idx.at(w0)];
          OpDesc op;
          op.kind = OP_CONV; op.c3 = 1; op.tag = tg + ":" + w0;
          op.nseg = 1; op.seg[0].v = view(img[i], 0, 0, 3);
          op.ksize = 3; op.leaky = 1; op.Cout = k; op.Ctot = 3;
          op.w_off = Lp.w_off; op.b_off = Lp.b_off;
          op.out = tmp; op.NB = N2; op.H = HL(lv); op.W = WL(lv);
          op.tile = TILE_C3_DIRECT | CONV_TILE_XCD | CONV_TILE_C3;
          op.flops = 2.0 * N2 * HL(lv) * WL(lv) * k * 27; op.bytes = 4.0 * N2 * HL(lv) * WL(lv) * (3 + k);
          P->ops.push_back(op);
        } else {
          SegDesc s; s.v = scratch(fx_p, k >> 1);
          conv(tg, w0, {s}, tmp, N2, HL(lv), WL(lv), true);
        }
        SegDesc s1; s1.v = tmp;
        View dst = view(feat[lv], 0, slot_offset(c, j), k);
        conv(tg, w1, {s1}, dst, N2, HL(lv), WL(lv), true);
        if (j < n - 1) {
          OpDesc& cv = P->ops.back();
          if ((h->opt_fuse & 8) && cv.kind == OP_CONV && (cv.wino == 3 || cv.wino == 4) && cv.ksplit <= 1 && !(HL(lv) & 1) && !(WL(lv) & 1)) {
            // AveragePooling2D in the epilogue of the Winograd kernels (a lane / thread holds both rows of a 2x2 block)
            cv.tag += "+pool";
            cv.out2 = scratch(fx_p, k);
          } else
            pool(tg + ":pool", dst, scratch(fx_p, k), N2, HL(lv), WL(lv));
        }
      }
      // every subtree but the level-0 one (75 % of the extractor's FLOPs) goes to the side stream: they and the coarse
      // flow levels that need only them are small, latency-bound launches that hide under the level-0 subtree
      // Small frames are latency bound: the flow chain l6 -> l0 can only start when the coarse subtrees (3..6) are done,
      // so those go first on the side stream while the main stream works through subtrees 0, 1, 2 (256x256: level-3 flow
      // starts after ~1.0 ms instead of ~1.5 ms).  Large frames keep subtrees 1.. on the side stream (tail filling).
      const int side_from = (int64_t)H * W <= 512 * 512 ? 3 : 1;
      if (i >= side_from) for (size_t q = first_op; q < P->ops.size(); ++q) P->ops[q].lane = 1;
    }

    // ---- bidirectional coarse-to-fine flow (pyramid_flow_estimator.py:125-163) ---------------------
    // batch n = d*B + b: d = 0 forward (a = image 0, b = image 1), d = 1 backward.
    for (int l = L - 1; l >= 0; --l) {
      const size_t first_flow_op = P->ops.size();
      const std::string tg = "flow_l" + std::to_string(l);
      const int pi = predictor_index(c, l);
      const int nf = c.flow_filters[pi], nconv = c.flow_convs[pi];
      const std::string prefix = predictor_prefix(c, l);
      const int Hl = HL(l), Wl = WL(l);
      SegDesc sa; sa.v = view(feat[l], 0, 0, fc[l]);
      SegDesc sb;
      if (l == L - 1) {
        sb.v = view(feat[l], 0, 0, fc[l]); sb.boff = B; sb.bmod = N2;  // the other image's features
      } else {
        // tf.image.resize(2 * v) inside the warps that consume it (fuse bit 1) pays on the small, launch-bound levels; on the
        // large ones every (pixel, channel group) thread of the warp would recompute the flow behind four gathers, a second
        // memory round trip in front of the corner loads: there the separate (2-channel) resize launch is faster - 1080p step
        // 45.8 -> 45.5 ms, warp class 3.18 -> 2.90 + 0.06 ms (profiles/r03_fuse_ab.log).  Same arithmetic either way.
        const bool fuse_up = (h->opt_fuse & 1) && (int64_t)Hl * Wl < 100000;
        if (!fuse_up) {
          OpDesc up;
          up.kind = OP_FLOW_UP; up.tag = tg + ":resize2x";
          up.in = view(v[l + 1], 0, 0, 2); up.out = view(vup[l], 0, 0, 2);
          up.NB = N2; up.H = HL(l + 1); up.W = WL(l + 1);
          up.bytes = 4.0 * N2 * Hl * Wl * 2 * 1.25;
          P->ops.push_back(up);
        }
        // Both directions in ONE launch (batch n = d * B + b, like every other op of the estimator): direction d warps the OTHER
        // image's features, i.e. the source batch is rotated by B (pyramid_flow_estimator.py:150-158 runs the two directions as
        // two calls; the arithmetic per pixel is the same).
        warp(tg + ":warp_d01", view(feat[l], 0, 0, fc[l]), view(vup[l], 0, 0, 2), view(warped[l], 0, 0, fc[l]), N2, Hl, Wl, 1.f);
        P->ops.back().src_brot = B;
        if (fuse_up) {
          // tf.image.resize(2 * v) (pyramid_flow_estimator.py:155) inside the warp: the flow of this level is computed
          // from the coarser level's v by every thread of a pixel and stored once (to vup, which v = res + up reads)
          OpDesc& w = P->ops.back();
          w.tag += "+resize2x";
          w.in2 = View();
          w.in3 = view(v[l + 1], 0, 0, 2);
          w.out2 = view(vup[l], 0, 0, 2);
        }
        sb.v = view(warped[l], 0, 0, fc[l]);
      }
      View cur = scratch(fp[0], nf);
      conv(tg, prefix + "/conv_0", {sa, sb}, cur, N2, Hl, Wl, true);
      int which = 0;
      for (int j = 1; j < nconv; ++j) {
        View nxt = scratch(fp[which ^ 1], nf);
        SegDesc s; s.v = cur;
        conv(tg, prefix + "/conv_" + std::to_string(j), {s}, nxt, N2, Hl, Wl, true);
        cur = nxt; which ^= 1;
      }
      View hid = scratch(fp[2], nf / 2);
      const std::string l3 = prefix + "/conv_" + std::to_string(nconv), l4 = prefix + "/conv_" + std::to_string(nconv + 1);
      if ((nf / 2) % 32 == 0) {
        SegDesc s; s.v = cur;
        conv(tg, l3, {s}, hid, N2, Hl, Wl, true);
        conv_pw(tg, l4, hid, view(res[l], 0, 0, 2), (int64_t)N2 * Hl * Wl, false);
        if ((h->opt_fuse & 2) && l < L - 1) {   // v = res + up in the head's epilogue
          OpDesc& pw = P->ops.back();
          pw.tag += "+v=res+up";
          pw.in2 = view(vup[l], 0, 0, 2); pw.out2 = view(v[l], 0, 0, 2);
        }
      } else {  // nf / 2 == 16: both 1x1 convs in one kernel, the 16-channel hidden layer stays in registers
        const LayerPack& L3 = h->layers[h->layer_idx.at(l3)];
        const LayerPack& L4 = h->layers[h->layer_idx.at(l4)];
        OpDesc op;
        op.kind = OP_FLOW_HEAD; op.tag = tg + ":" + l3 + "+conv_" + std::to_string(nconv + 1);
        op.in = cur; op.out = view(res[l], 0, 0, 2); op.n = (int64_t)N2 * Hl * Wl; op.Ctot = nf;
        op.w_off = L3.w_off; op.b_off = L3.b_off; op.w2_off = L4.w_off; op.b2_off = L4.b_off;
        op.flops = 2.0 * op.n * (nf * 16 + 16 * 2); op.bytes = 4.0 * op.n * (nf + 2);
        if ((h->opt_fuse & 2) && l < L - 1) {
          op.tag += "+v=res+up";
          op.in2 = view(vup[l], 0, 0, 2); op.out2 = view(v[l], 0, 0, 2);
        }
        P->ops.push_back(op);
      }
      if (l < L - 1 && !(h->opt_fuse & 2)) {
        OpDesc ad;
        ad.kind = OP_FLOW_ADD; ad.tag = tg + ":v=res+up";
        ad.in = view(res[l], 0, 0, 2); ad.in2 = view(vup[l], 0, 0, 2); ad.out = view(v[l], 0, 0, 2);
        ad.n = (int64_t)N2 * Hl * Wl * 2; ad.bytes = 4.0 * ad.n * 3;
        P->ops.push_back(ad);
      }
      // levels >= 4 read features of subtrees >= 1 only (level 3 needs the last stage of the level-0 subtree)
      if (l >= 4) for (size_t q = first_flow_op; q < P->ops.size(); ++q) P->ops[q].lane = 1;
    }

    // ---- warp to t = 0.5 and build the aligned pyramid (interpolator.py:153-183) ------------------
    // util.flow_pyramid_synthesis recomputes exactly the v sequence above, so v is reused.
    // image s is sampled with the flow of the opposite direction: image 0 <- backward flow (d=1).
    // Emitted coarse to fine and on the side stream: level l only needs v[l], which the flow estimator finishes
    // early for the coarse levels; the level-0 warps (60 % of the warp bytes, HBM bound) then overlap the
    // fusion convolutions (matrix-pipe bound) of the main stream.
    auto emit_align = [&](int l) {
      const size_t first_align_op = P->ops.size();
      const std::string tg = "align_l" + std::to_string(l);
      // Planar level: the planes of image 0 and image 1 follow each other, i.e. they are ONE [2B][H][W][C] array - both feature warps
      // in one launch (batch n = s * B + b; image s is sampled with the flow of the opposite direction: the flow batch is rotated by B).
      const bool pair = P->bufs[aligned[l]].planar > 0;
      if (pair) {
        warp(tg + ":warp_feat01", view(feat[l], 0, 0, fc[l]), view(v[l], 0, 0, 2), aligned_part(aligned[l], 0, fc[l]), N2, HL(l), WL(l), 0.5f);
        P->ops.back().flow_brot = B;
      }
      for (int s = 0; s < 2; ++s) {
        View fl = view(v[l], (1 - s) * B, 0, 2);
        if (!pair)
          warp(tg + ":warp_feat" + std::to_string(s), view(feat[l], s * B, 0, fc[l]), fl,
               aligned_part(aligned[l], s, fc[l]), B, HL(l), WL(l), 0.5f);
        if (!(h->opt_fuse & 4))
          warp(tg + ":warp_img" + std::to_string(s), view(img[l], s * B, 0, 3), fl,
               sub(aligned_part(aligned[l], 2, fc[l]), 3 * s, 3), B, HL(l), WL(l), 0.5f, false);
      }
      if (h->opt_fuse & 4) {
        // the sixteen miscellaneous channels [warp(img0) 3 | warp(img1) 3 | 0.5 bflow 2 | 0.5 fflow 2 | 0 x 6] ride in the second
        // feature warp of the level as one more channel slice (one full 64-byte line per pixel and store)
        OpDesc& w = P->ops.back();
        w.tag += "+misc16";
        if (pair) w.misc_nb = B;
        w.img_in = view(img[l], 0, 0, 3);             // [2B]: image 0, image 1
        w.img_out = aligned_part(aligned[l], 2, fc[l]);
        w.pack_b = view(v[l], B, 0, 2);   // backward flow (d = 1): samples image 0
        w.pack_f = view(v[l], 0, 0, 2);   // forward flow  (d = 0): samples image 1
        w.bytes += 4.0 * B * HL(l) * WL(l) * 12.0;   // both 3-channel images, read + written (SURVEY 8d counts the image with the features)
      } else {
        OpDesc pk;
        pk.kind = OP_PACK_FLOW; pk.tag = tg + ":flows";
        pk.in = view(v[l], B, 0, 2);   // backward flow (d = 1)
        pk.in2 = view(v[l], 0, 0, 2);  // forward flow  (d = 0)
        pk.out = sub(aligned_part(aligned[l], 2, fc[l]), 6, 10);
        pk.n = (int64_t)B * HL(l) * WL(l); pk.bytes = 4.0 * pk.n * 14;
        P->ops.push_back(pk);
      }
      for (size_t q = first_align_op; q < P->ops.size(); ++q) P->ops[q].lane = 1;
    };

    // ---- fusion decoder (fusion.py:103-140) ---------------------------------------------------------
    View net = view(aligned[FL - 1], 0, 0, 2 * fc[FL - 1] + 16);
    auto emit_fusion = [&](int i, int lane) {
      const size_t first_op = P->ops.size();
      const std::string tg = "fusion_l" + std::to_string(i);
      const std::string base = "fusion/convs_" + std::to_string(i);
      SegDesc su; su.v = net; su.up = 1;
      conv(tg, base + "_0", {su}, view(fu_u[i], 0, 0, ff[i]), B, HL(i), WL(i), false);
      SegDesc s1; s1.v = view(fu_u[i], 0, 0, ff[i]);
      if (P->bufs[aligned[i]].planar) {
        SegDesc a0, a1, a2;
        a0.v = aligned_part(aligned[i], 0, fc[i]); a1.v = aligned_part(aligned[i], 1, fc[i]); a2.v = aligned_part(aligned[i], 2, fc[i]);
        conv(tg, base + "_1", {a0, a1, a2, s1}, view(fu_a[i], 0, 0, ff[i]), B, HL(i), WL(i), true);
      } else {
        SegDesc s0; s0.v = view(aligned[i], 0, 0, 2 * fc[i] + 16);
        conv(tg, base + "_1", {s0, s1}, view(fu_a[i], 0, 0, ff[i]), B, HL(i), WL(i), true);
      }
      SegDesc s2; s2.v = view(fu_a[i], 0, 0, ff[i]);
      conv(tg, base + "_2", {s2}, view(fu_b[i], 0, 0, ff[i]), B, HL(i), WL(i), true);
      net = view(fu_b[i], 0, 0, ff[i]);
      for (size_t q = first_op; q < P->ops.size(); ++q) P->ops[q].lane = lane;
    };
    // Option "lanes" >= 2 (NOT the default - measured 2 % slower, see opt_lanes), large frames: the COARSE decoder levels (>= 2) join the side stream right behind the
    // aligned levels they read - fusion level i only needs aligned[i], aligned[i + 1] / the level above, i.e. the flow
    // of level i, which the estimator finishes while it still has levels i - 1 .. 0 to go.  Their matrix-bound
    // convolutions then run beside the estimator's chain of HBM-bound warps, 1x1 heads and short launches on the main
    // stream instead of behind it; the fine levels (1, 0) stay on the main stream, where the level-0 warps of the side
    // stream overlap them as before.  Small frames keep the decoder behind the estimator (latency bound).
    const bool early_fusion = h->opt_lanes >= 2 && ((int64_t)H * W > 512 * 512 || h->opt_lanes >= 3) && FL >= 4;   // 3: test knob, any frame size
    emit_align(FL - 1);
    if (early_fusion) {
      for (int i = FL - 2; i >= 0; --i) {
        emit_align(i);
        if (i >= 2) emit_fusion(i, 1);
      }
      for (int i = std::min(FL - 2, 1); i >= 0; --i) emit_fusion(i, 0);
    } else {
      for (int i = FL - 2; i >= 0; --i) emit_align(i);
      for (int i = FL - 2; i >= 0; --i) emit_fusion(i, 0);
    }
    {
      // RGB head (fusion.py:138-140): a 1x1 convolution of the last decoder layer.  Fused (option fuse bit 16) into that
      // layer's epilogue when it runs on a Winograd kernel with 64 output channels and no split: the 64-channel
      // activation (566 MB per 1080p step) is then neither written nor read back.
      OpDesc& last = P->ops.back();
      const LayerPack& LO = h->layers[h->layer_idx.at("fusion/output_conv")];
      if ((h->opt_fuse & 16) && last.kind == OP_CONV && (last.wino == 3 || last.wino == 4) && last.Cout == 64 && last.ksplit <= 1 && last.out2.buf < 0 &&
          LO.cout <= 4 && LO.cin == 64) {
        last.tag += "+output_conv";
        last.pw_out = view(out, 0, 0, 3); last.pw_cout = LO.cout;
        last.w2_off = LO.w_off; last.b2_off = LO.b_off;
        last.tile = last.wino == 4 ? (W2D_8x64 | CONV_TILE_W2D | CONV_TILE_XCD) : (W43_Q16_4x64_N1_P2 | CONV_TILE_WINO | CONV_TILE_F43 | CONV_TILE_XCD);
        last.flops += 2.0 * (double)B * H * W * LO.cout * LO.cin;
        last.bytes = 4.0 * (double)B * H * W * (64 + LO.cout);
      } else {
        conv_pw("fusion_out", "fusion/output_conv", net, view(out, 0, 0, 3), (int64_t)B * H * W, false);
      }
    }
    if (bad) return fail(h, FILM_ERR_INVALID, "%s", bad_msg.c_str());
    P->arena_floats = cursor;   // the split-K partial-sum regions are added while the ops are emitted
    analyze_lanes();
    return FILM_OK;
  }

  // ---- cross-lane dependencies from buffer overlap -------------------------------------------------------------
  struct Access { int buf; int c0, c1; };  // channels [c0, c1) of every pixel of a buffer (scratch: everything)
  Access access(const View& v) const {
    const Buffer& b = P->bufs[v.buf];
    if (b.C == 0 || v.stride != b.C) return {v.buf, 0, 1 << 30};  // scratch, or a reinterpreted view: whole buffer
    const int c0 = (int)((v.off - b.off) % b.C);
    return {v.buf, c0, c0 + v.C};
  }
  static bool overlap(const Access& a, const Access& b) { return a.buf == b.buf && a.c0 < b.c1 && b.c0 < a.c1; }
  void accesses(const OpDesc& op, std::vector<Access>& rd, std::vector<Access>& wr) const {
    rd.clear(); wr.clear();
    if (op.kind == OP_CONV) { for (int i = 0; i < op.nseg; ++i) rd.push_back(access(op.seg[i].v)); }
    else { if (op.in.buf >= 0) rd.push_back(access(op.in)); if (op.in2.buf >= 0) rd.push_back(access(op.in2)); }
    if (op.in3.buf >= 0) rd.push_back(access(op.in3));
    if (op.img_in.buf >= 0) rd.push_back(access(op.img_in));
    if (op.pack_b.buf >= 0) { rd.push_back(access(op.pack_b)); rd.push_back(access(op.pack_f)); }
    if (op.out.buf >= 0 && op.pw_out.buf < 0) wr.push_back(access(op.out));
    if (op.pw_out.buf >= 0) wr.push_back(access(op.pw_out));
    if (op.out2.buf >= 0) wr.push_back(access(op.out2));
    if (op.img_out.buf >= 0) wr.push_back(access(op.img_out));
  }
  // For every op: the LAST op of the other lane it conflicts with (RAW, WAR or WAW on overlapping channels of a
  // buffer).  Waiting for the last one is enough: a lane executes in program order.  For the same reason a wait is
  // dropped when the lane has already waited for that op or a later one of the other lane: every op of a lane is then
  // waited for at most ONCE by the other lane, i.e. a graph node has at most two children (its lane successor and one
  // cross-lane waiter).  That is not only tidier: the graph replay of HIP 7.0/7.2 starts a node whose ONLY parent
  // already has >= 5 earlier-captured children without waiting for that parent
  // (tools/experiments/graph_single_parent_race.hip).
  void analyze_lanes() {
    const size_t n = P->ops.size();
    std::vector<std::vector<Access>> rd(n), wr(n);
    for (size_t i = 0; i < n; ++i) accesses(P->ops[i], rd[i], wr[i]);
    long waited[2] = {-1, -1};   // per lane: the latest op of the other lane it has waited for
    for (size_t j = 0; j < n; ++j) {
      OpDesc& oj = P->ops[j];
      const int lj = oj.lane ? 1 : 0;
      for (size_t ii = j; ii-- > 0;) {
        const OpDesc& oi = P->ops[ii];
        if ((oi.lane ? 1 : 0) == lj) continue;
        if ((long)ii <= waited[lj]) break;   // ordered already (and so is everything before it)
        bool hit = false;
        for (const Access& w : wr[ii]) {
          for (const Access& r : rd[j]) hit |= overlap(w, r);
          for (const Access& w2 : wr[j]) hit |= overlap(w, w2);
        }
        for (const Access& r : rd[ii])
          for (const Access& w2 : wr[j]) hit |= overlap(r, w2);
        if (hit) { oj.xdeps.push_back((int)ii); P->ops[ii].signal = true; waited[lj] = (long)ii; break; }
      }
    }
  }
};


}  // namespace

int plan_build(film_t* h, Plan* P, int B, int H, int W) {
  Planner pl{h, P};
  return pl.build(B, H, W);
}

// The conv kernels other than conv_wino43_kernel address their inputs with 32-bit byte offsets from the start of the
// buffer (buffer loads): every activation buffer THEY read must stay below 4 GiB.  conv_wino43_kernel addresses relative
// to the workgroup's own halo rows and every other kernel with 64-bit pointers, so the large levels of a large frame
// (F(4,3) layers only: an untiled 4K frame has 4.4-5 GB level-0 buffers) are not limited.  Largest limited buffer of a
// B = 1 plan, in bytes (buffers scale linearly with the batch; the kernel family of a layer does not depend on it).
int64_t limited_buffer_bytes(const Plan* P) {
  int64_t mx = 1;
  for (const OpDesc& op : P->ops) {
    if (op.kind != OP_CONV || op.wino == 3 || op.wino == 4) continue;
    for (int i = 0; i < op.nseg; ++i) mx = std::max(mx, P->bufs[op.seg[i].v.buf].floats * (int64_t)sizeof(float));
  }
  return mx;
}

namespace {
void json_view(std::ostringstream& o, const char* key, const View& v, const Plan& P) {
  o << "\"" << key << "\":{\"buf\":\"" << (v.buf >= 0 ? P.bufs[v.buf].name : std::string("")) << "\",\"off\":" << v.off
    << ",\"stride\":" << v.stride << ",\"C\":" << v.C << "}";
}

}  // namespace

std::string plan_json(film_t* h, const Plan& P) {
  std::ostringstream o;
  o << "{\"B\":" << P.B << ",\"H\":" << P.H << ",\"W\":" << P.W << ",\"arena_floats\":" << P.arena_floats
    << ",\"offset32_buffer_bytes\":" << limited_buffer_bytes(&P)
    << ",\"packed_floats\":" << h->packed_floats << ",\"buffers\":[";
  for (size_t i = 0; i < P.bufs.size(); ++i) {
    const Buffer& b = P.bufs[i];
    o << (i ? "," : "") << "{\"name\":\"" << b.name << "\",\"off\":" << b.off << ",\"N\":" << b.N << ",\"H\":" << b.H
      << ",\"W\":" << b.W << ",\"C\":" << b.C << ",\"floats\":" << b.floats << ",\"planar\":" << b.planar << "}";
  }
  o << "],\"layers\":[";
  for (size_t i = 0; i < h->layers.size(); ++i) {
    const LayerPack& L = h->layers[i];
    o << (i ? "," : "") << "{\"name\":\"" << L.name << "\",\"kh\":" << L.kh << ",\"kw\":" << L.kw << ",\"cin\":" << L.cin
      << ",\"cout\":" << L.cout << ",\"ctot\":" << L.ctot() << ",\"w_off\":" << L.w_off << ",\"b_off\":" << L.b_off << ",\"wh_off\":" << L.wh_off << ",\"ws_off\":" << L.ws_off << ",\"ww_off\":" << L.ww_off << ",\"wx_off\":" << L.wx_off << "}";
  }
  o << "],\"ops\":[";
  for (size_t i = 0; i < P.ops.size(); ++i) {
    const OpDesc& op = P.ops[i];
    o << (i ? "," : "") << "{\"kind\":\"" << kKindName[op.kind] << "\",\"tag\":\"" << op.tag << "\",\"NB\":" << op.NB
      << ",\"H\":" << op.H << ",\"W\":" << op.W << ",\"ksize\":" << op.ksize << ",\"leaky\":" << op.leaky
      << ",\"Cout\":" << op.Cout << ",\"Ctot\":" << op.Ctot << ",\"tile\":" << op.tile << ",\"w_off\":" << op.w_off
      << ",\"b_off\":" << op.b_off << ",\"wf4_off\":" << op.wf4_off << ",\"wh_off\":" << op.wh_off << ",\"halo\":" << op.halo << ",\"ws_off\":" << op.ws_off << ",\"split\":" << op.split << ",\"ww_off\":" << op.ww_off << ",\"wx_off\":" << op.wx_off << ",\"wfx_off\":" << op.wfx_off << ",\"w43_off\":" << op.w43_off << ",\"w2d_off\":" << op.w2d_off << ",\"wino\":" << op.wino << ",\"fold\":" << op.fold << ",\"ksplit\":" << op.ksplit << ",\"py\":" << op.py
      << ",\"px\":" << op.px << ",\"ftaps\":" << op.ftaps << ",\"tdy\":[" << op.tdy[0] << "," << op.tdy[1] << "," << op.tdy[2] << "," << op.tdy[3]
      << "],\"tdx\":[" << op.tdx[0] << "," << op.tdx[1] << "," << op.tdx[2] << "," << op.tdx[3] << "]"
      << ",\"fold_woff\":[" << op.fold_woff[0] << "," << op.fold_woff[1] << "," << op.fold_woff[2] << "," << op.fold_woff[3] << "]" << ",\"lane\":" << op.lane << ",\"xdeps\":["
      << [&] { std::string d; for (size_t q = 0; q < op.xdeps.size(); ++q) d += (q ? "," : "") + std::to_string(op.xdeps[q]); return d; }() << "]" << ",\"w2_off\":" << op.w2_off << ",\"b2_off\":" << op.b2_off << ",\"c3\":" << op.c3
      << ",\"fscale\":" << op.fscale << ",\"src_brot\":" << op.src_brot << ",\"flow_brot\":" << op.flow_brot << ",\"misc_nb\":" << op.misc_nb << ",\"n\":" << op.n << ",\"flops\":" << op.flops
      << ",\"bytes\":" << op.bytes << ",";
    json_view(o, "in", op.in, P); o << ",";
    json_view(o, "in2", op.in2, P); o << ",";
    json_view(o, "in3", op.in3, P); o << ",";
    json_view(o, "out2", op.out2, P); o << ",";
    json_view(o, "pw_out", op.pw_out, P); o << ",\"pw_cout\":" << op.pw_cout << ",";
    json_view(o, "img_in", op.img_in, P); o << ",";
    json_view(o, "img_out", op.img_out, P); o << ",";
    json_view(o, "pack_b", op.pack_b, P); o << ",";
    json_view(o, "pack_f", op.pack_f, P); o << ",";
    json_view(o, "out", op.out, P);
    o << ",\"segs\":[";
    for (int k = 0; k < op.nseg; ++k) {
      o << (k ? "," : "") << "{";
      json_view(o, "v", op.seg[k].v, P);
      o << ",\"boff\":" << op.seg[k].boff << ",\"bmod\":" << op.seg[k].bmod << ",\"up\":" << op.seg[k].up << "}";
    }
    o << "]}";
  }
  o << "]}";
  return o.str();
}


}  // namespace film_internal
