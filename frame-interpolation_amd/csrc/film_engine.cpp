// film_engine.cpp -- C-ABI (include/film_hip.h) and executor of the MI355X FILM inference engine: handle lifetime, options,
// per-(B,H,W) plan cache with workspace arenas, per-shape autotune of the tile shapes, hipGraph capture / replay on two
// streams, per-op profiling, batch chunking, film_forward / film_interpolate / film_get_tap.  The planner lives in
// film_planner.cpp, the layer table and the weight packer in film_layers.cpp, the shared structures in film_internal.h.
// There is no CPU execution path here: plan-only handles (device = -1) can pack weights and describe
// plans, every compute entry point needs a HIP device.
#include "film_internal.h"

thread_local std::string film_internal::g_create_error;

namespace film_internal {

int fail(film_t* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}


}  // namespace film_internal

using namespace film_internal;

namespace {

// ---------------------------------------------------------------------------------------------
// Executor
// ---------------------------------------------------------------------------------------------
inline const float* cptr(const float* base, const View& v) { return base + v.off; }
inline float* mptr(float* base, const View& v) { return base + v.off; }

hipError_t launch_op(const OpDesc& op, float* arena, const float* wts, hipStream_t s) {
  switch (op.kind) {
    case OP_CONV: {
      ConvParams p{};
      p.nseg = op.nseg;
      for (int i = 0; i < op.nseg; ++i) {
        p.seg[i].ptr = cptr(arena, op.seg[i].v);
        p.seg[i].stride = op.seg[i].v.stride;
        p.seg[i].C = op.seg[i].v.C;
        p.seg[i].boff = op.seg[i].boff; p.seg[i].bmod = op.seg[i].bmod; p.seg[i].up = op.seg[i].up;
      }
      p.ksize = op.ksize;
      p.w = wts + ((op.tile & CONV_TILE_W2D) ? op.w2d_off : (op.tile & CONV_TILE_FOLDX3) ? op.wfx_off
                   : (op.tile & CONV_TILE_WINO) ? ((op.tile & CONV_TILE_X3) ? op.wx_off : (op.tile & CONV_TILE_F43) ? op.w43_off : op.ww_off) : (op.tile & CONV_TILE_SPLIT) ? op.ws_off
                   : (op.tile & CONV_TILE_HALO) ? op.wh_off : op.w_off);
      p.bias = wts + op.b_off;
      p.out = mptr(arena, op.out); p.ostride = op.out.stride;
      p.NB = op.NB; p.H = op.H; p.W = op.W; p.Cout = op.Cout; p.Ctot = op.Ctot; p.leaky = op.leaky;
      p.M = op.NB * op.H * op.W;
      p.fold = op.fold; p.py = op.py; p.px = op.px; p.ftaps = op.ftaps;
      p.ksplit = op.ksplit; p.part = op.ksplit > 1 ? arena + op.part_off : nullptr;
      if (op.out2.buf >= 0) { p.pool_out = mptr(arena, op.out2); p.pool_ostride = op.out2.stride; }
      if (op.pw_out.buf >= 0) {
        p.pw_w = wts + op.w2_off; p.pw_bias = wts + op.b2_off; p.pw_out = mptr(arena, op.pw_out); p.pw_ostride = op.pw_out.stride; p.pw_cout = op.pw_cout;
      }
      for (int q = 0; q < 4; ++q) { p.tdy[q] = (signed char)op.tdy[q]; p.tdx[q] = (signed char)op.tdx[q]; p.fold_woff[q] = op.fold_woff[q]; }
      return film_launch_conv(p, op.tile, s);
    }
    case OP_FLOW_HEAD: {
      FlowHeadParams p{};
      p.in = cptr(arena, op.in); p.istride = op.in.stride; p.Cin = op.Ctot;
      p.w3 = wts + op.w_off; p.b3 = wts + op.b_off; p.w4 = wts + op.w2_off; p.b4 = wts + op.b2_off;
      p.out = mptr(arena, op.out); p.M = (int)op.n;
      if (op.out2.buf >= 0) { p.add = cptr(arena, op.in2); p.sum = mptr(arena, op.out2); }
      return film_launch_flow_head(p, s);
    }
    case OP_CONV_PW: {
      ConvPwParams p{};
      p.in = cptr(arena, op.in); p.istride = op.in.stride; p.Cin = op.Ctot;
      p.w = wts + op.w_off; p.bias = wts + op.b_off;
      p.out = mptr(arena, op.out); p.ostride = op.out.stride; p.Cout = op.Cout; p.leaky = op.leaky;
      p.M = (int)op.n;
      if (op.out2.buf >= 0) { p.add = cptr(arena, op.in2); p.sum = mptr(arena, op.out2); }
      return film_launch_conv_pw(p, s);
    }
    case OP_POOL: {
      PoolParams p{};
      p.in = cptr(arena, op.in); p.istride = op.in.stride; p.out = mptr(arena, op.out); p.ostride = op.out.stride;
      p.C = op.in.C; p.NB = op.NB; p.H = op.H; p.W = op.W;
      return film_launch_pool(p, s);
    }
    case OP_FLOW_UP: {
      FlowUpParams p{};
      p.in = cptr(arena, op.in); p.out = mptr(arena, op.out); p.NB = op.NB; p.h = op.H; p.w = op.W;
      return film_launch_flow_up(p, s);
    }
    case OP_FLOW_ADD: {
      FlowAddParams p{};
      p.a = cptr(arena, op.in); p.b = cptr(arena, op.in2); p.out = mptr(arena, op.out); p.n = op.n;
      return film_launch_flow_add(p, s);
    }
    case OP_WARP: {
      WarpParams p{};
      p.src = cptr(arena, op.in); p.sstride = op.in.stride; p.C = op.in.C;
      p.flow = op.in2.buf >= 0 ? cptr(arena, op.in2) : nullptr; p.fscale = op.fscale;
      p.dst = mptr(arena, op.out); p.dstride = op.out.stride;
      p.NB = op.NB; p.H = op.H; p.W = op.W;
      p.src_brot = op.src_brot; p.flow_brot = op.flow_brot; p.misc_nb = op.misc_nb;
      if (op.in3.buf >= 0) { p.coarse = cptr(arena, op.in3); p.flow_out = mptr(arena, op.out2); }
      if (op.img_out.buf >= 0) {   // misc16: both images (img_in = [2 NB] images), both flows
        p.src3 = cptr(arena, op.img_in); p.s3stride = op.img_in.stride;
        p.src3b = p.src3 + (int64_t)(op.misc_nb > 0 ? op.misc_nb : op.NB) * op.H * op.W * op.img_in.stride;
        p.dst3 = mptr(arena, op.img_out); p.d3stride = op.img_out.stride;
        p.pack_b = cptr(arena, op.pack_b); p.pack_f = cptr(arena, op.pack_f);
      }
      return film_launch_warp(p, s);
    }
    case OP_PACK_FLOW: {
      PackFlowParams p{};
      p.bflow = cptr(arena, op.in); p.fflow = cptr(arena, op.in2);
      p.dst = mptr(arena, op.out); p.dstride = op.out.stride; p.npix = op.n;
      return film_launch_pack_flow(p, s);
    }
  }
  return hipErrorInvalidValue;
}

std::vector<int> halo_candidates(int Cout) {
  std::vector<int> shapes;
  if (Cout % 128 == 0) shapes = {HALO_4x64, HALO_4x128, HALO_8x64, HALO_8x128};
  else if (Cout % 64 == 0) shapes = {HALO_4x64, HALO_8x64, HALO_4x32, HALO_8x32};
  else shapes = {HALO_8x32, HALO_4x32};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_HALO); out.push_back(sh | CONV_TILE_HALO | CONV_TILE_XCD); }
  return out;
}

std::vector<int> wino_candidates(int Cout) {
  std::vector<int> shapes = Cout % 128 == 0 ? std::vector<int>{WINO_4x64_W8, WINO_8x64_W16, WINO_2x64, WINO_4x128_W16, WINO_4x128}
                            : Cout % 64 == 0 ? std::vector<int>{WINO_4x64_W8, WINO_8x64_W16, WINO_2x64, WINO_4x32}
                                             : std::vector<int>{WINO_4x32, WINO_8x32_W8};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_WINO); out.push_back(sh | CONV_TILE_WINO | CONV_TILE_XCD); }
  return out;
}

// tile id of a Wino43Tile shape: ids >= 16 go into the low four bits with CONV_TILE_EXT set (bit 4 is CONV_TILE_XCD)
inline int w43_tile(int sh) { return (sh & 15) | (sh >= 16 ? CONV_TILE_EXT : 0) | CONV_TILE_WINO | CONV_TILE_F43; }

std::vector<int> wino43_candidates(int Cout, bool pool = false, bool pw = false) {
  if (pw) {   // the fused 1x1 needs every channel of a pixel in one workgroup: the NH = 1 tiles at Cout = 64
    std::vector<int> out;
    for (int sh : {W43_Q16_4x64_N1, W43_Q16_4x64_N1_P2, W43_Q8_8x64_N1_P2}) {
      if (!film_w43_shape_built(sh)) continue;
      out.push_back(w43_tile(sh)); out.push_back(w43_tile(sh) | CONV_TILE_XCD);
    }
    return out;
  }
  // the 64-pixel ("Q16", two workgroups per CU) tiles won every layer of the 1080p plan against the 128-pixel ones
  // (profiles/r02_conv_bench_w43.log); one 128-pixel tile stays in the list for shapes nobody measured.  The 32-pixel x
  // 8-row ("Q8") tiles win on the 480-wide level (15 patches per row exactly: -3..5 %) and, with 32 channels and the weight
  // ring (three workgroups per CU), on the 128 -> 32 layer of flow level 0 (-7 %): profiles/r03_conv_bench_w43.log
  std::vector<int> shapes = Cout % 64 == 0 ? std::vector<int>{W43_4x64_T21, W43_Q16_4x64_T21, W43_Q16_4x64_T12, W43_Q16_4x32_T11, W43_Q16_4x64_N1,
                                                              W43_Q16_4x64_T21_P2, W43_Q16_4x64_T12_P2, W43_Q16_4x32_T11_P2, W43_Q16_4x64_N1_P2, W43_Q16_4x32_T11_BG,
                                                              W43_Q8_8x64_T21_P2, W43_Q8_8x64_T12_P2, W43_Q8_8x64_N1_P2, W43_Q8_8x32_T11_BG, W43_Q8_8x32_T11_P2}
                                            : std::vector<int>{W43_4x32_T11, W43_Q16_4x32_T11, W43_Q16_4x32_T11_P2, W43_Q16_4x32_T11_BG, W43_Q8_8x32_T11_BG, W43_Q8_8x32_T11_P2};
  std::vector<int> out;
  for (int sh : shapes) {
    if (!film_w43_shape_built(sh)) continue;   // (the default library holds seven of the seventeen tiles)
    if (pool && (sh == W43_4x64_T21 || sh == W43_4x64_T12 || sh == W43_4x32_T11)) continue;   // the fused pool needs a <= 64-pixel tile
    out.push_back(w43_tile(sh)); out.push_back(w43_tile(sh) | CONV_TILE_XCD);
  }
  return out;
}

// (H, W: the level.  The square arrangement is a candidate where its tiles pad the level no more than the 8 x 32 ones.)
std::vector<int> wino2d_candidates(int Cout, bool pw, int H, int W) {
  const int64_t pad_r = (int64_t)((W + 31) / 32) * ((H + 7) / 8), pad_s = (int64_t)((W + 15) / 16) * ((H + 15) / 16);
  const bool sq = pad_s <= pad_r;
  std::vector<int> shapes;
  if (pw || Cout % 64 == 0) { shapes.push_back(W2D_8x64); if (sq) shapes.push_back(W2D_16x64); }
  if (!pw) {   // (the fused 1x1 needs every channel of a pixel in one workgroup)
    shapes.push_back(W2D_8x32); shapes.push_back(W2D_8x32_S2);
    if (sq) { shapes.push_back(W2D_16x32); shapes.push_back(W2D_16x32_S2); }
  }
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_W2D); out.push_back(sh | CONV_TILE_W2D | CONV_TILE_XCD); }
  return out;
}

std::vector<int> fold4_candidates(int Cout) {
  std::vector<int> out;
  for (int sh : (Cout % 64 == 0 ? std::vector<int>{F4_4x64, F4_4x32} : std::vector<int>{F4_4x32})) { out.push_back(sh | CONV_TILE_FOLD4); out.push_back(sh | CONV_TILE_FOLD4 | CONV_TILE_XCD); }
  return out;
}

std::vector<int> foldx3_candidates(int Cout) {
  std::vector<int> shapes = Cout % 128 == 0 ? std::vector<int>{FX3_4x64, FX3_8x64, FX3_4x128} : std::vector<int>{FX3_4x64, FX3_8x64};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_FOLDX3); out.push_back(sh | CONV_TILE_FOLDX3 | CONV_TILE_XCD); }
  return out;
}

std::vector<int> winox3_candidates(int Cout) {
  std::vector<int> shapes = Cout % 128 == 0 ? std::vector<int>{WX3_4x128_T22, WX3_4x64_T12, WX3_4x64_T21}
                            : Cout % 64 == 0 ? std::vector<int>{WX3_4x64_T12, WX3_4x64_T21, WX3_4x32_T11}
                                             : std::vector<int>{WX3_4x32_T11};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_WINO | CONV_TILE_X3); out.push_back(sh | CONV_TILE_WINO | CONV_TILE_X3 | CONV_TILE_XCD); }
  return out;
}

std::vector<int> split_candidates(int Cout, bool x3) {
  std::vector<int> out;
  for (int t : halo_candidates(Cout)) out.push_back((t & ~CONV_TILE_HALO) | CONV_TILE_SPLIT | (x3 ? CONV_TILE_X3 : 0));
  return out;
}

std::vector<int> tile_candidates(int Cout) {
  std::vector<int> shapes;
  if (Cout % 128 == 0) shapes = {TILE_128x128, TILE_256x128, TILE_256x64, TILE_128x64, TILE_64x64};
  else if (Cout % 64 == 0) shapes = {TILE_256x64, TILE_128x64, TILE_64x64, TILE_256x32, TILE_128x32};
  else shapes = {TILE_256x32, TILE_128x32};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh); out.push_back(sh | CONV_TILE_XCD); }
  return out;
}

std::string conv_signature(const OpDesc& op) {
  std::ostringstream o;
  o << op.NB << 'x' << op.H << 'x' << op.W << ':' << op.Cout << ':' << op.ksize << ':' << op.out.stride << ':' << op.c3 << ':' << op.halo << ':' << op.split << ':' << op.wino << ':' << op.fold << ':' << op.ksplit << ':' << (op.out2.buf >= 0) << ':' << op.pw_cout;
  for (int i = 0; i < op.nseg; ++i)
    o << '|' << op.seg[i].v.C << ',' << op.seg[i].v.stride << ',' << op.seg[i].up << ',' << op.seg[i].bmod;
  return o.str();
}

hipError_t launch_op(const OpDesc& op, float* arena, const float* wts, hipStream_t s);

// Measure, don't guess: every distinct conv shape of a plan is timed once with each tile shape that fits
// its Cout (random activations, the real weights) and keeps the fastest.  The choice cannot change the
// results: every output element is the same k-ordered fma chain whatever the tile.
std::vector<int> conv_candidates(const OpDesc& op) {
  std::vector<int> cands = op.fold == 3 ? fold4_candidates(op.Cout) : (op.fold == 2 && op.split == 2) ? foldx3_candidates(op.Cout) : op.wino == 4 ? wino2d_candidates(op.Cout, op.pw_out.buf >= 0, op.H, op.W) : op.wino == 3 ? wino43_candidates(op.Cout, op.out2.buf >= 0, op.pw_out.buf >= 0) : op.wino == 2 ? winox3_candidates(op.Cout) : op.wino ? wino_candidates(op.Cout) : op.split ? split_candidates(op.Cout, op.split == 2) : op.halo ? halo_candidates(op.Cout) : tile_candidates(op.Cout);
  if (op.c3) {   // the 3-channel first layer has one kernel (conv_c3_kernel)
    cands.clear();
    cands.push_back(TILE_C3_DIRECT | CONV_TILE_C3);
  }
  return cands;
}

int autotune_plan(film_t* h, Plan* P) {
  bool need = false;
  for (const OpDesc& op : P->ops) {
    if (op.kind != OP_CONV) continue;
    const std::string sig = conv_signature(op);
    if (h->tune_cache.count(sig)) continue;
    auto it = h->tune_import.find(sig);
    if (it != h->tune_import.end()) {   // an earlier process measured this shape: keep its choice if it is still a candidate
      const std::vector<int> cands = conv_candidates(op);
      if (std::find(cands.begin(), cands.end(), it->second) != cands.end()) { h->tune_cache[sig] = it->second; continue; }
    }
    need = true;
  }
  if (need) {
    HIPCHK(h, film_launch_fill_random(P->arena, P->arena_floats, 0x9e3779b9u, h->stream));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    for (OpDesc& op : P->ops) {
      if (op.kind != OP_CONV) continue;
      const std::string sig = conv_signature(op);
      if (h->tune_cache.count(sig)) continue;
      int best = op.tile;
      float best_ms = 1e30f;
      const std::vector<int> cands = conv_candidates(op);
      auto time_once = [&](int tile, float* ms) -> int {
        OpDesc trial = op;
        trial.tile = tile;
        HIPCHK(h, hipEventRecord(e0, h->stream));
        HIPCHK(h, launch_op(trial, P->arena, h->packed_dev, h->stream));
        HIPCHK(h, hipEventRecord(e1, h->stream));
        HIPCHK(h, hipEventSynchronize(e1));
        HIPCHK(h, hipEventElapsedTime(ms, e0, e1));
        return FILM_OK;
      };
      std::vector<std::pair<float, int>> timed;
      for (int tile : cands) {
        OpDesc trial = op;
        trial.tile = tile;
        HIPCHK(h, launch_op(trial, P->arena, h->packed_dev, h->stream));  // warm
        float ms_min = 1e30f, ms_sum = 0.f;
        // at least two timed launches; with the "tune_ms" option keep going until that much kernel time has been
        // spent on the candidate (long enough for the power-limited clock to settle)
        for (int rep = 0; rep < 2 || (ms_sum < (float)h->opt_tune_ms && rep < 64); ++rep) {
          float ms = 0;
          int trc = time_once(tile, &ms);
          if (trc) return trc;
          ms_min = std::min(ms_min, ms);
          ms_sum += ms;
        }
        timed.push_back({ms_min, tile});
      }
      // Run-off: the candidates within 6 % of the fastest (at most four) are timed four more times each, round robin, so
      // that a single lucky launch (clock state, neighbours in L2) does not decide a layer that runs every forward.
      std::sort(timed.begin(), timed.end());
      size_t nfin = 0;
      while (nfin < timed.size() && nfin < 4 && timed[nfin].first <= timed[0].first * 1.06f) ++nfin;
      if (nfin > 1)
        for (int round = 0; round < 4; ++round)
          for (size_t c = 0; c < nfin; ++c) {
            float ms = 0;
            int trc = time_once(timed[c].second, &ms);
            if (trc) return trc;
            timed[c].first = std::min(timed[c].first, ms);
          }
      for (size_t c = 0; c < std::max<size_t>(nfin, 1) && c < timed.size(); ++c)
        if (timed[c].first < best_ms) { best_ms = timed[c].first; best = timed[c].second; }
      // conv_wino2d_kernel: a 64-channel tile within 2 % of the fastest wins - it reads its input patch half as often (45.2 -> 40.2
      // GB of fabric reads per 1080p forward with the tile forced, same step time: profiles/r04_w2d_tile64_ab.log)
      if (op.wino == 4 && !film_w2d_64(best & 15)) {
        float ms64 = 1e30f;
        int t64 = -1;
        for (size_t c = 0; c < std::max<size_t>(nfin, 1) && c < timed.size(); ++c)
          if (film_w2d_64(timed[c].second & 15) && timed[c].first < ms64) { ms64 = timed[c].first; t64 = timed[c].second; }
        if (t64 >= 0 && ms64 <= best_ms * 1.02f) best = t64;
      }
      h->tune_cache[sig] = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(h, hipMemsetAsync(P->arena, 0, (size_t)P->arena_floats * sizeof(float), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  for (OpDesc& op : P->ops)
    if (op.kind == OP_CONV) op.tile = h->tune_cache.at(conv_signature(op));
  return FILM_OK;
}

void free_plan(Plan* p) {
  if (!p) return;
  if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  for (auto e : p->ev) (void)hipEventDestroy(e);
  for (auto e : p->lane_ev) if (e) (void)hipEventDestroy(e);
  if (p->arena) (void)hipFree(p->arena);
}

int get_plan(film_t* h, int B, int H, int W, bool need_device, Plan** out) {
  const int div = 1 << (h->cfg.pyramid_levels - 1);
  if (B < 1 || H < 1 || W < 1) return fail(h, FILM_ERR_INVALID, "B, H, W must be positive");
  if (H % div || W % div)
    return fail(h, FILM_ERR_INVALID, "input height and width (%d x %d) must be divisible by %d = 2^(pyramid_levels-1); "
                "pad first (Interpolator align)", H, W, div);
  // tfa dense_image_warp needs a >= 2x2 grid at every warped level
  const int wl = std::max(h->cfg.pyramid_levels - 2, h->cfg.fusion_pyramid_levels - 1);
  if ((H >> wl) < 2 || (W >> wl) < 2)
    return fail(h, FILM_ERR_INVALID, "input %d x %d too small: warped pyramid level %d would be smaller than 2x2", H, W, wl);
  if ((int64_t)2 * B * H * W >= (int64_t)1 << 31) return fail(h, FILM_ERR_INVALID, "batch too large (2*B*H*W must fit int32)");
  for (auto& p : h->plans)
    if (p->B == B && p->H == H && p->W == W && (!need_device || p->arena)) { *out = p.get(); p->last_use = ++h->tick; return FILM_OK; }
  if (need_device)   // a description-only plan of this shape (film_plan_json, unit_buffer_bytes) is superseded, not kept beside the new one
    for (size_t i = 0; i < h->plans.size(); ++i)
      if (h->plans[i]->B == B && h->plans[i]->H == H && h->plans[i]->W == W) {
        if (h->last_plan == h->plans[i].get()) h->last_plan = nullptr;
        free_plan(h->plans[i].get());
        h->plans.erase(h->plans.begin() + i);
        break;
      }
  std::unique_ptr<Plan> P(new Plan);
  int rc = plan_build(h, P.get(), B, H, W);
  if (rc) return rc;
  if (need_device) {
    // keep at most 3 device plans alive (workspaces are GBs at 1080p tiles)
    size_t alive = 0;
    for (auto& p : h->plans) alive += p->arena != nullptr;
    while (alive >= 3) {
      size_t victim = h->plans.size();
      for (size_t i = 0; i < h->plans.size(); ++i)
        if (h->plans[i]->arena && (victim == h->plans.size() || h->plans[i]->last_use < h->plans[victim]->last_use)) victim = i;
      if (victim == h->plans.size()) break;
      if (h->last_plan == h->plans[victim].get()) h->last_plan = nullptr;
      // a graph launch of the victim on the caller's stream may still be running (device-resident callers are
      // asynchronous): its graph, events and workspace must outlive it
      (void)hipSetDevice(h->device);
      (void)hipDeviceSynchronize();
      free_plan(h->plans[victim].get());
      h->plans.erase(h->plans.begin() + victim);
      --alive;
    }
    hipError_t e = hipMalloc(&P->arena, (size_t)P->arena_floats * sizeof(float));
    if (e != hipSuccess && alive > 0) {   // out of memory: give back the other plans' workspaces and try once more
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      for (size_t i = h->plans.size(); i-- > 0;)
        if (h->plans[i]->arena) { free_plan(h->plans[i].get()); h->plans.erase(h->plans.begin() + i); }
      h->last_plan = nullptr;
      e = hipMalloc(&P->arena, (size_t)P->arena_floats * sizeof(float));
    }
    if (e != hipSuccess) {
      P->arena = nullptr;
      return fail(h, FILM_ERR_NOMEM, "workspace hipMalloc of %.1f MB failed: %s", P->arena_floats * 4e-6, hipGetErrorString(e));
    }
    HIPCHK(h, hipMemsetAsync(P->arena, 0, (size_t)P->arena_floats * sizeof(float), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->opt_autotune && h->finalized) {
      int trc = autotune_plan(h, P.get());
      if (trc) { free_plan(P.get()); return trc; }
    }
  }
  if (h->opt_w43_shape >= 0)   // test knob: one tile shape for every F(4,3) op it fits (same bits as any other, by construction)
    for (OpDesc& op : P->ops) {
      if (op.kind != OP_CONV || op.wino != 3) continue;
      const std::vector<int> cands = conv_candidates(op);
      const int want = w43_tile(h->opt_w43_shape) | CONV_TILE_XCD;
      if (std::find(cands.begin(), cands.end(), want) != cands.end()) op.tile = want;
    }
  if (h->opt_w2d_shape >= 0)   // the same for the nested-Winograd tiles
    for (OpDesc& op : P->ops) {
      if (op.kind != OP_CONV || op.wino != 4) continue;
      const std::vector<int> cands = conv_candidates(op);
      const int want = h->opt_w2d_shape | CONV_TILE_W2D | CONV_TILE_XCD;
      if (std::find(cands.begin(), cands.end(), want) != cands.end()) op.tile = want;
    }
  if (h->opt_fold4_shape >= 0)   // ... and for conv_fold4_kernel's
    for (OpDesc& op : P->ops) {
      if (op.kind != OP_CONV || op.fold != 3) continue;
      const std::vector<int> cands = conv_candidates(op);
      const int want = h->opt_fold4_shape | CONV_TILE_FOLD4 | CONV_TILE_XCD;
      if (std::find(cands.begin(), cands.end(), want) != cands.end()) op.tile = want;
    }
  P->last_use = ++h->tick;
  *out = P.get();
  h->plans.push_back(std::move(P));
  return FILM_OK;
}

int copy_out_string(film_t* h, const std::string& s, char* buf, int64_t cap, int64_t* needed) {
  if (needed) *needed = (int64_t)s.size() + 1;
  if (!buf || cap < (int64_t)s.size() + 1) {
    if (!buf && needed) return FILM_OK;  // size query
    return fail(h, FILM_ERR_INVALID, "buffer too small: need %lld bytes", (long long)s.size() + 1);
  }
  memcpy(buf, s.c_str(), s.size() + 1);
  return FILM_OK;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int film_to_uint8(const float* src, unsigned char* dst, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return FILM_ERR_INVALID;
  return film_launch_to_uint8(src, dst, n, (hipStream_t)stream) == hipSuccess ? FILM_OK : FILM_ERR_HIP;
}

#ifndef FILM_SRC_ID
#define FILM_SRC_ID "unknown"
#endif
#ifdef FILM_EXTRA_FAMILIES
#define FILM_FLAVOUR "+extra"
#else
#define FILM_FLAVOUR ""
#endif
// "gfx950;film_hip r6;src=<sha1[:12] of csrc/ + include/film_hip.h>[+extra]": ties tune caches, bench lines and PMC summaries to the
// kernel sources they were produced with (film_hip/build.py source_id(), `make print-src-id`)
const char* film_version(void) { return "gfx950;film_hip r6;src=" FILM_SRC_ID FILM_FLAVOUR; }

int film_default_config(film_config* cfg) {
  if (!cfg) return FILM_ERR_INVALID;
  memset(cfg, 0, sizeof *cfg);
  cfg->pyramid_levels = 7; cfg->fusion_pyramid_levels = 5; cfg->specialized_levels = 3; cfg->sub_levels = 4;
  cfg->filters = 64;
  const int fcv[4] = {3, 3, 3, 3}, ffl[4] = {32, 64, 128, 256};
  for (int i = 0; i < 4; ++i) { cfg->flow_convs[i] = fcv[i]; cfg->flow_filters[i] = ffl[i]; }
  return FILM_OK;
}

int film_create(film_t** out, int device, const film_config* cfg) {
  if (!out) return fail(nullptr, FILM_ERR_INVALID, "out is NULL");
  *out = nullptr;
  std::unique_ptr<film_handle> h(new film_handle);
  if (cfg) h->cfg = *cfg; else film_default_config(&h->cfg);
  int rc = validate_config(h.get(), h->cfg);
  if (rc) { g_create_error = h->err; return rc; }
  h->device = device;
  h->plan_only = device < 0;
  if (!h->plan_only) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      return fail(nullptr, FILM_ERR_NO_DEVICE, "no HIP device available (%s); libfilm_hip has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device >= ndev) return fail(nullptr, FILM_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, FILM_ERR_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(nullptr, FILM_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess)
      return fail(nullptr, FILM_ERR_HIP, "hipStreamCreate failed");
  }
  build_layers(h.get());
  *out = h.release();
  return FILM_OK;
}

void film_destroy(film_t* h) {
  if (!h) return;
  if (!h->plan_only) {
    (void)hipSetDevice(h->device);
    // forwards may still be running on the caller's stream and on the side lane: workspaces, events and graphs must outlive them
    (void)hipDeviceSynchronize();
  }
  for (auto& p : h->plans) free_plan(p.get());
  if (h->packed_dev) (void)hipFree(h->packed_dev);
  if (h->stage) (void)hipFree(h->stage);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (hipEvent_t& e : h->pipe_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  delete h;
}

const char* film_last_error(const film_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int film_set_option(film_t* h, const char* key, int64_t value) {
  if (!h || !key) return FILM_ERR_INVALID;
  if (!strcmp(key, "graph")) {
    if (value < 0 || value > 2) return fail(h, FILM_ERR_INVALID, "graph: 0 (one stream, eager), 1 (hipGraph replay) or 2 (two lanes, eager: the default)");
    h->opt_graph = (int)value;
  }
  else if (!strcmp(key, "profile")) h->opt_profile = value != 0;
  else if (!strcmp(key, "autotune")) h->opt_autotune = value != 0;
  else if (!strcmp(key, "max_batch")) h->opt_max_batch = value > 0 ? (int)value : 0;
  else if (!strcmp(key, "host_overlap")) h->opt_host_overlap = value != 0;
  else if (!strcmp(key, "tune_ms")) h->opt_tune_ms = value > 0 ? (int)value : 0;
  else if (!strcmp(key, "splitk")) {
    if ((value != 0) != (h->opt_splitk != 0)) {  // plans carry the op list: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_splitk = value != 0;
    }
  }
  else if (!strcmp(key, "pack_groups")) {
    if (value < 1 || value > 4) return fail(h, FILM_ERR_INVALID, "pack_groups: 1 .. 4");
    if (h->finalized) { int rc = film_ensure_groups_(h, (int)value); if (rc) return rc; }
  }
  else if (!strcmp(key, "fuse")) {
    if ((int)(value & 31) != h->opt_fuse) {  // plans carry the op list: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_fuse = (int)(value & 31);
    }
  }
  else if (!strcmp(key, "fold2x2")) {
    if (value < 0 || value > 2) return fail(h, FILM_ERR_INVALID, "fold2x2: 0, 1 or 2");
    const int fold = value != 0, fold4 = value == 1;
    if (fold != h->opt_fold || fold4 != h->opt_fold4) {  // plans carry the op list: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_fold = fold;
      h->opt_fold4 = fold4;
    }
  }
  else if (!strcmp(key, "planar")) {
    if ((value != 0) != (h->opt_planar != 0)) {  // plans carry the buffer layout: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_planar = value != 0;
    }
  }
  else if (!strcmp(key, "winograd")) {
    if (value < 0 || value > 3) return fail(h, FILM_ERR_INVALID, "winograd: 0, 1, 2 or 3");
#ifndef FILM_EXTRA_FAMILIES
    if (value == 2) return fail(h, FILM_ERR_INVALID, "winograd = 2 (F(2,3) kernel on every level) needs a library built with FILM_EXTRA_FAMILIES=1");
#endif
    if ((int)value != h->opt_wino) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_wino = (int)value;
    }
  }
  else if (!strcmp(key, "halo_all")) {
#ifndef FILM_EXTRA_FAMILIES
    if (value) return fail(h, FILM_ERR_INVALID, "halo_all needs a library built with FILM_EXTRA_FAMILIES=1");
#endif
    if ((value != 0) != (h->opt_halo_all != 0)) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_halo_all = value != 0;
    }
  }
  else if (!strcmp(key, "lanes")) {
    if (value < 0 || value > 3) return fail(h, FILM_ERR_INVALID, "lanes: 0, 1, 2 or 3");
    if ((int)value != h->opt_lanes) {  // plans carry the lane of every op and the op order: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_lanes = (int)value;
    }
  }
  else if (!strcmp(key, "wino2d")) {
    if (value < 0 || value > 2) return fail(h, FILM_ERR_INVALID, "wino2d: 0, 1 or 2");
    if ((int)value != h->opt_wino2d) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_wino2d = (int)value;
    }
  }
  else if (!strcmp(key, "w2d_small_px")) {
    if (value < 0 || value > (1 << 30)) return fail(h, FILM_ERR_INVALID, "w2d_small_px: pixels per image, 0 = never");
    if ((int)value != h->opt_w2d_small_px) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_w2d_small_px = (int)value;
    }
  }
  else if (!strcmp(key, "w2d_min_px")) {
    if (value < 1 || value > (1 << 30)) return fail(h, FILM_ERR_INVALID, "w2d_min_px: pixels per image");
    if ((int)value != h->opt_w2d_min_px) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_w2d_min_px = (int)value;
    }
  }
  else if (!strcmp(key, "w2d_splitk")) {
    if (value < 0 || value > 16) return fail(h, FILM_ERR_INVALID, "w2d_splitk: 0 (off), 1 (default rule) or 2..16 (A/B: the cap on levels of <= 1024 pixels)");
    if ((int)value != h->opt_w2d_splitk) {  // plans carry the split factors: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_w2d_splitk = (int)value;
    }
  }
  else if (!strcmp(key, "w2d_shape")) {
    if (value < -1 || value >= W2D_SHAPES) return fail(h, FILM_ERR_INVALID, "w2d_shape: -1 (autotuned) or a Wino2dTile shape index");
    if ((int)value != h->opt_w2d_shape) {  // plans carry the tile choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_w2d_shape = (int)value;
    }
  }
  else if (!strcmp(key, "fold4_shape")) {
    if (value < -1 || value > F4_4x32) return fail(h, FILM_ERR_INVALID, "fold4_shape: -1 (autotuned) or a Fold4Tile shape index");
    if ((int)value != h->opt_fold4_shape) {  // plans carry the tile choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_fold4_shape = (int)value;
    }
  }
  else if (!strcmp(key, "w43_shape")) {
    if (value < -1 || value > 31) return fail(h, FILM_ERR_INVALID, "w43_shape: -1 (autotuned) or a Wino43Tile shape index");
    if ((int)value != h->opt_w43_shape) {  // plans carry the tile choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_w43_shape = (int)value;
    }
  }
  else if (!strcmp(key, "precision")) {
    if (value != 0 && value != 1 && value != 2) return fail(h, FILM_ERR_INVALID, "precision: 0 (f32), 1 (bf16x6) or 2 (bf16x3)");
#ifndef FILM_EXTRA_FAMILIES
    if (value) return fail(h, FILM_ERR_INVALID, "precision %d (bf16 split modes) needs a library built with FILM_EXTRA_FAMILIES=1; this build runs fp32 MFMA only", (int)value);
#endif
    if ((int)value != h->opt_precision) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_precision = (int)value;
    }
  }
  else return fail(h, FILM_ERR_NOTFOUND, "unknown option '%s'", key);
  return FILM_OK;
}

int film_plan_json(film_t* h, int B, int H, int W, char* buf, int64_t cap, int64_t* needed) {
  if (!h) return FILM_ERR_INVALID;
  Plan* P = nullptr;
  int rc = get_plan(h, B, H, W, false, &P);
  if (rc) return rc;
  return copy_out_string(h, plan_json(h, *P), buf, cap, needed);
}

int film_profile_json(film_t* h, char* buf, int64_t cap, int64_t* needed) {
  if (!h) return FILM_ERR_INVALID;
  if (h->profile_json.empty()) return fail(h, FILM_ERR_STATE, "no profiled forward yet (film_set_option(\"profile\", 1))");
  return copy_out_string(h, h->profile_json, buf, cap, needed);
}

// Autotune choices as text: a header line with the library version, then one "<conv shape signature>\t<tile id>" line per
// measured shape (this handle's own measurements + imported ones it has not needed yet).
int film_export_tune(film_t* h, char* buf, int64_t cap, int64_t* needed) {
  if (!h) return FILM_ERR_INVALID;
  std::ostringstream o;
  o << "# film_hip tune cache v1 " << film_version() << "\n";
  std::map<std::string, int> all = h->tune_import;
  for (const auto& kv : h->tune_cache) all[kv.first] = kv.second;
  for (const auto& kv : all) o << kv.first << '\t' << kv.second << '\n';
  return copy_out_string(h, o.str(), buf, cap, needed);
}

// Takes the text of film_export_tune.  A cache written by another library version is ignored (returns FILM_OK, imports
// nothing: tile ids are only meaningful within one build); entries are validated when a plan first needs them - a tile
// that is not a candidate of the op's kernel family is measured again.  Results never depend on the cache: every tile of
// a family produces the same bits.
int film_import_tune(film_t* h, const char* text) {
  if (!h || !text) return fail(h, FILM_ERR_INVALID, "NULL argument");
  std::istringstream in(text);
  std::string line;
  if (!std::getline(in, line)) return FILM_OK;
  const std::string want = std::string("# film_hip tune cache v1 ") + film_version();
  if (line != want) return FILM_OK;
  std::map<std::string, int> got;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    const size_t tab = line.rfind('\t');
    if (tab == std::string::npos || tab == 0 || tab + 1 >= line.size()) return fail(h, FILM_ERR_INVALID, "tune cache: malformed line '%s'", line.c_str());
    char* end = nullptr;
    const long tile = strtol(line.c_str() + tab + 1, &end, 10);
    if (*end != 0 || tile < 0 || tile > (1 << 20)) return fail(h, FILM_ERR_INVALID, "tune cache: malformed line '%s'", line.c_str());
    got[line.substr(0, tab)] = (int)tile;
  }
  for (const auto& kv : got) h->tune_import[kv.first] = kv.second;
  return FILM_OK;
}

namespace {
constexpr int64_t kMaxBufferBytes = 0xFFF00000ll;
// One model invocation also keeps its workspace below this (a fifth of the HBM): 15 tiles of 960x576, one untiled 4K frame
constexpr int64_t kMaxArenaBytes = 64ll << 30;
// ... and below 60 % of the HBM this handle could get right now (free memory + what its own cached plans hold): other
// ranks' handles, torch's allocator or a smaller part may share the device.
int64_t arena_budget_bytes(film_t* h) {
  int64_t cap = kMaxArenaBytes;
  if (!h->plan_only) {
    size_t fr = 0, tot = 0;
    if (hipSetDevice(h->device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) {
      int64_t held = 0;
      for (auto& p : h->plans) if (p->arena) held += p->arena_floats * (int64_t)sizeof(float);
      cap = std::min<int64_t>(cap, ((int64_t)fr + held) / 10 * 6);
    } else {
      (void)hipGetLastError();
    }
  }
  return std::max<int64_t>(cap, 1);
}
int64_t unit_buffer_bytes(film_t* h, int H, int W, int* rc, int* max_units) {
  Plan* P1 = nullptr;
  *rc = get_plan(h, 1, H, W, false, &P1);
  if (*rc) return 0;
  const int64_t lim = limited_buffer_bytes(P1);
  const int64_t arena = std::max<int64_t>(1, P1->arena_floats * (int64_t)sizeof(float));
  *max_units = (int)std::max<int64_t>(1, std::min<int64_t>(kMaxBufferBytes / lim, arena_budget_bytes(h) / arena));
  return lim;
}
// Chunk size for n independent units with at most maxc per invocation: the largest divisor of n in (maxc / 2, maxc] if
// there is one, so that every invocation runs the SAME cached plan (16 * 2^k tiles of a 4K recursion with maxc = 15 ->
// chunks of 8, never a 15 + 1 split that would build, tune and capture a second plan for the remainder); else maxc.
int balanced_chunk(int n, int maxc) {
  if (n <= maxc) return n;
  for (int c = maxc; 2 * c > maxc; --c)
    if (n % c == 0) return c;
  return maxc;
}

int forward_chunk(film_t* h, const float* x0, const float* x1, int B, int H, int W, float* out, int mem_kind, void* stream);
int run_plan(film_t* h, Plan* P, hipStream_t s);
hipStream_t pick_stream(film_t* h, int mem_kind, void* stream);
}  // namespace

int film_forward(film_t* h, const float* x0, const float* x1, int B, int H, int W, float* out, int mem_kind, void* stream) {
  if (!h || !x0 || !x1 || !out) return fail(h, FILM_ERR_INVALID, "NULL argument");
  if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle: film_forward needs a HIP device (no CPU fallback)");
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  if (mem_kind != FILM_MEM_HOST && mem_kind != FILM_MEM_DEVICE) return fail(h, FILM_ERR_INVALID, "bad mem_kind");
  if (B < 1) return fail(h, FILM_ERR_INVALID, "B, H, W must be positive");
  int rc = 0;
  int bmax = 1;
  const int64_t unit = unit_buffer_bytes(h, H, W, &rc, &bmax);
  if (rc) return rc;
  if (unit > kMaxBufferBytes)
    return fail(h, FILM_ERR_INVALID, "a %d x %d frame needs a %.1f GB activation buffer in front of a kernel that addresses 4 GiB per "
                "buffer - tile the frame (Interpolator block_shape)", H, W, unit * 1e-9);
  if (h->opt_max_batch) bmax = std::min(bmax, h->opt_max_batch);
  const size_t frame = (size_t)H * W * 3;
  int chunk = balanced_chunk(B, bmax);
  for (int b0 = 0; b0 < B;) {  // independent frame pairs: the batch splits with no change in results
    const int nb = std::min(chunk, B - b0);
    rc = forward_chunk(h, x0 + b0 * frame, x1 + b0 * frame, nb, H, W, out + b0 * frame, mem_kind, stream);
    if (rc == FILM_ERR_NOMEM && nb > 1) { chunk = (nb + 1) / 2; continue; }   // workspace did not fit: smaller chunks (nothing was launched)
    if (rc) return rc;
    b0 += nb;
  }
  return FILM_OK;
}

namespace {
int interpolate_host_pipeline(film_t* h, Plan* P, TileMapParams tp, const float* x0, const float* x1, float* out, float* st, size_t frame_bytes, hipStream_t s);
}

int film_interpolate(film_t* h, const float* x0, const float* x1, int B, int H, int W, int align, int block_h,
                     int block_w, float* out, int mem_kind, void* stream) {
  if (!h || !x0 || !x1 || !out) return fail(h, FILM_ERR_INVALID, "NULL argument");
  if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle: film_interpolate needs a HIP device (no CPU fallback)");
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  if (mem_kind != FILM_MEM_HOST && mem_kind != FILM_MEM_DEVICE) return fail(h, FILM_ERR_INVALID, "bad mem_kind");
  if (B < 1 || H < 1 || W < 1) return fail(h, FILM_ERR_INVALID, "B, H, W must be positive");
  const int bh = block_h > 0 ? block_h : 1, bw = block_w > 0 ? block_w : 1;
  // the reference's asserts (eval/interpolator.py:84-89), same messages
  if (H % bh) return fail(h, FILM_ERR_INVALID, "block_height=%d should evenly divide height=%d.", bh, H);
  if (W % bw) return fail(h, FILM_ERR_INVALID, "block_width=%d should evenly divide width=%d.", bw, W);
  TileMapParams tp{};
  tp.B = B; tp.H = H; tp.W = W; tp.bh = bh; tp.bw = bw; tp.ph = H / bh; tp.pw = W / bw;
  const int hp = (align > 0 && tp.ph % align) ? align - tp.ph % align : 0;   // _pad_to_align, eval/interpolator.py:45-52
  const int wp = (align > 0 && tp.pw % align) ? align - tp.pw % align : 0;
  tp.TH = tp.ph + hp; tp.TW = tp.pw + wp; tp.oy = hp / 2; tp.ox = wp / 2;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = 0;
  int tmax = 1;
  const int64_t unit = unit_buffer_bytes(h, tp.TH, tp.TW, &rc, &tmax);
  if (rc) return rc;
  if (unit > kMaxBufferBytes)
    return fail(h, FILM_ERR_INVALID, "a %d x %d tile needs a %.1f GB activation buffer in front of a kernel that addresses 4 GiB per "
                "buffer - use a finer block_shape", tp.TH, tp.TW, unit * 1e-9);
  const int ntiles = B * bh * bw;
  if (h->opt_max_batch) tmax = std::min(tmax, h->opt_max_batch);
  hipStream_t s = pick_stream(h, mem_kind, stream);
  const size_t frame_bytes = (size_t)B * H * W * 3 * sizeof(float);
  const float *d0 = x0, *d1 = x1;
  float* dout = out;
  if (mem_kind == FILM_MEM_HOST) {  // stage whole frames in HBM: [x0 | x1 | out]
    if (h->stage_bytes < 3 * frame_bytes) {
      if (h->stage) { HIPCHK(h, hipStreamSynchronize(s)); HIPCHK(h, hipFree(h->stage)); h->stage = nullptr; h->stage_bytes = 0; }
      hipError_t e = hipMalloc(&h->stage, 3 * frame_bytes);
      if (e != hipSuccess) return fail(h, FILM_ERR_NOMEM, "frame staging hipMalloc of %.1f MB failed", 3 * frame_bytes * 1e-6);
      h->stage_bytes = 3 * frame_bytes;
    }
    float* st = (float*)h->stage;
    const size_t nf = frame_bytes / sizeof(float);
    d0 = st; d1 = st + nf; dout = st + 2 * nf;
    // Host pipeline (round 6): one chunk on the direct two-lane executor - see interpolate_host_pipeline below
    if (h->opt_host_overlap && !h->opt_profile && h->opt_graph == 2 && h->opt_lanes != 0 && ntiles <= tmax && h->stream2) {
      Plan* P = nullptr;
      rc = get_plan(h, ntiles, tp.TH, tp.TW, true, &P);
      if (rc == FILM_OK) return interpolate_host_pipeline(h, P, tp, x0, x1, out, st, frame_bytes, s);
      if (rc != FILM_ERR_NOMEM) return rc;   // (workspace did not fit: the chunked path below)
    }
    HIPCHK(h, hipMemcpyAsync(st, x0, frame_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(st + nf, x1, frame_bytes, hipMemcpyHostToDevice, s));
  }
  int chunk = balanced_chunk(ntiles, tmax);
  for (int t0 = 0; t0 < ntiles;) {
    const int nt = std::min(chunk, ntiles - t0);
    Plan* P = nullptr;
    rc = get_plan(h, nt, tp.TH, tp.TW, true, &P);
    if (rc == FILM_ERR_NOMEM && nt > 1) { chunk = (nt + 1) / 2; continue; }   // workspace did not fit: smaller chunks
    if (rc) return rc;
    const Buffer& img0 = P->bufs[P->find("img0")];
    const Buffer& ob = P->bufs[P->find("out")];
    tp.tile0 = t0; tp.ntiles = nt;
    tp.src = d0; tp.dst = P->arena + img0.off;
    HIPCHK(h, film_launch_frame_to_tiles(tp, s));
    tp.src = d1; tp.dst = P->arena + img0.off + (int64_t)nt * tp.TH * tp.TW * 3;
    HIPCHK(h, film_launch_frame_to_tiles(tp, s));
    rc = run_plan(h, P, s);
    if (rc) return rc;
    tp.src = P->arena + ob.off; tp.dst = dout;
    HIPCHK(h, film_launch_tiles_to_frame(tp, s));
    t0 += nt;
  }
  if (mem_kind == FILM_MEM_HOST) {
    HIPCHK(h, hipMemcpyAsync(out, dout, frame_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
  }
  return FILM_OK;
}

namespace {
// ---- batch parts of an op (the overlapped host path of film_interpolate) --------------------------------------------------------
// A convolution whose input is ONE segment without a batch remap computes every image of its batch independently: images
// [part * NB / nparts, (part + 1) * NB / nparts) as a launch of their own give the same bits (the kernel family and the tile are the op's).
bool batch_splittable(const OpDesc& op, int nparts) {
  return op.kind == OP_CONV && op.nseg == 1 && op.seg[0].bmod == 0 && op.seg[0].boff == 0 && op.seg[0].up == 0 && op.ksplit <= 1 && op.fold == 0 &&
         op.NB >= nparts && op.NB % nparts == 0 && (op.out2.buf < 0 || (!(op.H & 1) && !(op.W & 1)));
}
OpDesc batch_part(const OpDesc& op, int part, int nparts) {
  OpDesc q = op;
  q.NB = op.NB / nparts;
  const int64_t px = (int64_t)part * q.NB * op.H * op.W;
  q.seg[0].v.off += px * op.seg[0].v.stride;
  q.out.off += px * op.out.stride;
  if (op.out2.buf >= 0) q.out2.off += (int64_t)part * q.NB * (op.H / 2) * (op.W / 2) * op.out2.stride;
  if (op.pw_out.buf >= 0) q.pw_out.off += px * op.pw_out.stride;
  return q;
}
// Hooks of film_interpolate's host-buffer pipeline into the two-lane issue (see there): `head` = the leading main-lane convolutions that run per
// input frame (part 0 = the tiles of x0 - launched by the caller BEFORE the second frame's upload; issue_lanes launches part 1), `tail` = the
// last op (the decoder's last layer + RGB head) runs as two tile halves with `mid_tail` between them (stitch + download of the first half).
struct LanePipe {
  std::vector<size_t> head;
  bool tail = false;
  std::function<hipError_t()> mid_tail;
};

// The plan's ops on two lanes: lane 0 on `main`, lane 1 (small subtrees, coarse flow levels, the t = 0.5 warps) on the handle's side
// stream, forked from and joined to `main`; cross-lane ordering = the events found by Planner::analyze_lanes (each op is waited for
// at most once by the other lane - see there for why that matters to a graph replay).  Called inside a stream capture (graph = 1:
// the events become graph edges) or directly (graph = 2: real events; one set per plan, re-recorded every forward - a wait refers
// to the record that precedes it in program order, and everything of forward n + 1 is ordered behind forward n's join on `main`).
hipError_t issue_lanes(film_t* h, Plan* P, hipStream_t main, bool capturing, const LanePipe* lp = nullptr) {
  const size_t nops = P->ops.size();
  if (P->lane_ev.size() < nops + 2) P->lane_ev.resize(nops + 2, nullptr);
  hipError_t ev_err = hipSuccess;
  auto event_of = [&](size_t i) -> hipEvent_t {
    if (!P->lane_ev[i]) {
      hipError_t e = hipEventCreateWithFlags(&P->lane_ev[i], hipEventDisableTiming);
      if (e != hipSuccess) { ev_err = e; P->lane_ev[i] = nullptr; }
    }
    return P->lane_ev[i];
  };
  const bool two_lanes = h->opt_lanes != 0;
  hipError_t le = hipSuccess;
  if (two_lanes) {
    le = hipEventRecord(event_of(nops), main);
    if (le == hipSuccess) le = hipStreamWaitEvent(h->stream2, event_of(nops), 0);
  }
  for (size_t i = 0; i < nops && le == hipSuccess; ++i) {
    const OpDesc& op = P->ops[i];
    const int lane = (two_lanes && op.lane == 1) ? 1 : 0;
    hipStream_t ls = lane ? h->stream2 : main;
    if (two_lanes)
      for (int d : op.xdeps) {
        le = hipStreamWaitEvent(ls, event_of((size_t)d), 0);
        if (le != hipSuccess) break;
      }
    if (le != hipSuccess) break;
    if (lp && std::find(lp->head.begin(), lp->head.end(), i) != lp->head.end()) le = launch_op(batch_part(op, 1, 2), P->arena, h->packed_dev, ls);
    else if (lp && lp->tail && i + 1 == nops) {
      le = launch_op(batch_part(op, 0, 2), P->arena, h->packed_dev, ls);
      if (le == hipSuccess) le = lp->mid_tail();
      if (le == hipSuccess) le = launch_op(batch_part(op, 1, 2), P->arena, h->packed_dev, ls);
    } else le = launch_op(op, P->arena, h->packed_dev, ls);
    if (le == hipSuccess && two_lanes && op.signal) le = hipEventRecord(event_of(i), ls);
  }
  if (two_lanes && le == hipSuccess) {
    le = hipEventRecord(event_of(nops + 1), h->stream2);
    if (le == hipSuccess) le = hipStreamWaitEvent(main, event_of(nops + 1), 0);
  } else if (two_lanes && !capturing) {
    // a launch or an event call failed behind the fork: the side stream may still hold lane-1 work that nothing on `main` is ordered
    // behind.  Drain it before the error goes back to the caller, who may reuse or free the buffers of this plan (round-5 ADVICE).
    // (inside a capture the streams carry no work: the capture itself is invalidated and ended by the caller)
    (void)hipStreamSynchronize(h->stream2);
  }
  return le == hipSuccess ? ev_err : le;
}

// Executes the plan on stream s (inputs already in the plan's img0 buffer, result left in its out buffer).
int run_plan(film_t* h, Plan* P, hipStream_t s) {
  const int B = P->B, H = P->H, W = P->W;
  if (h->opt_profile) {
    const size_t n = P->ops.size();
    while (P->ev.size() < n + 1) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); P->ev.push_back(e); }
    HIPCHK(h, hipEventRecord(P->ev[0], s));
    for (size_t i = 0; i < n; ++i) {
      HIPCHK(h, launch_op(P->ops[i], P->arena, h->packed_dev, s));
      HIPCHK(h, hipEventRecord(P->ev[i + 1], s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    struct Acc { int launches = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Acc> cls;
    std::ostringstream ops;
    for (size_t i = 0; i < n; ++i) {
      float ms = 0;
      HIPCHK(h, hipEventElapsedTime(&ms, P->ev[i], P->ev[i + 1]));
      Acc& a = cls[kKindName[P->ops[i].kind]];
      a.launches++; a.ms += ms; a.flops += P->ops[i].flops; a.bytes += P->ops[i].bytes;
      ops << (i ? "," : "") << "{\"tag\":\"" << P->ops[i].tag << "\",\"kind\":\"" << kKindName[P->ops[i].kind] << "\",\"ms\":" << ms
          << ",\"flops\":" << P->ops[i].flops << ",\"bytes\":" << P->ops[i].bytes << ",\"tile\":" << P->ops[i].tile << "}";
    }
    std::ostringstream o;
    o << "{\"B\":" << B << ",\"H\":" << H << ",\"W\":" << W << ",\"classes\":{";
    bool first = true;
    for (auto& kv : cls) {
      o << (first ? "" : ",") << "\"" << kv.first << "\":{\"launches\":" << kv.second.launches << ",\"ms\":" << kv.second.ms
        << ",\"flops\":" << kv.second.flops << ",\"bytes\":" << kv.second.bytes << "}";
      first = false;
    }
    o << "},\"ops\":[" << ops.str() << "]}";
    h->profile_json = o.str();
  } else if (h->opt_graph == 1) {
    if (!P->graph_exec) {
      // capture on the handle's own stream, replay on whichever stream the caller wants
      HIPCHK(h, hipStreamSynchronize(s));
      HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      const hipError_t le = issue_lanes(h, P, h->stream, true);
      hipError_t ce = hipStreamEndCapture(h->stream, &P->graph);
      if (le != hipSuccess) return fail(h, FILM_ERR_HIP, "kernel launch failed during capture: %s", hipGetErrorString(le));
      HIPCHK(h, ce);
      HIPCHK(h, hipGraphInstantiate(&P->graph_exec, P->graph, nullptr, nullptr, 0));
    }
    HIPCHK(h, hipGraphLaunch(P->graph_exec, s));
  } else if (h->opt_graph == 2 && h->opt_lanes != 0) {
    // the DEFAULT: the same two lanes and the same event edges, launched directly - lane 0 on the caller's stream, lane 1 on the
    // handle's side stream (issue_lanes; why not a hipGraph by default: film_internal.h, opt_graph)
    const hipError_t le = issue_lanes(h, P, s, false);
    if (le != hipSuccess) return fail(h, FILM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
  } else {
    for (const OpDesc& op : P->ops) HIPCHK(h, launch_op(op, P->arena, h->packed_dev, s));
  }
  h->last_plan = P;
  return FILM_OK;
}

// film_interpolate with HOST buffers, one chunk, direct two-lane executor.  The reference's call is numpy -> numpy (eval/interpolator.py:152-209): the
// copies are part of it.  Instead of upload, upload, work, download:
//   main stream:  H2D x0 | tiles of x0 | first layers on x0's tiles ("head" parts 0) | wait E0 | the same layers on x1's tiles | ... the plan ... |
//                 last layer, first tile half | stitch first half, record E1 | last layer, second half | stitch | D2H second half
//   side stream:  (behind the tiles of x0) H2D x1 | tiles of x1 | record E0 | lane 1 of the plan ... | wait E1 | D2H first half
// The API calls are made in this order, so that it also holds for pageable memory, whose copies block the calling thread: the GPU works on x0
// while the host copies x1, and on the second half of the last layer while the host receives the first half of the frame.
// Splitting a convolution's batch into two launches cannot change a bit (batch_part).
int interpolate_host_pipeline(film_t* h, Plan* P, TileMapParams tp, const float* x0, const float* x1, float* out, float* st, size_t frame_bytes, hipStream_t s) {
  const int nt = P->B;
  const size_t nf = frame_bytes / sizeof(float);
  for (hipEvent_t& e : h->pipe_ev)
    if (!e) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const Buffer& img0 = P->bufs[P->find("img0")];
  const Buffer& ob = P->bufs[P->find("out")];
  LanePipe lp;
  for (size_t i = 0; i < P->ops.size() && lp.head.size() < 2; ++i) {   // the leading main-lane convolutions that read nothing from the side lane
    const OpDesc& op = P->ops[i];
    if (op.lane == 1) continue;
    if (!batch_splittable(op, 2) || !op.xdeps.empty() || op.NB != 2 * nt) break;
    lp.head.push_back(i);
  }
  const OpDesc& last = P->ops.back();
  // the tiles of a frame are row-major blocks: the first half of the tiles = the upper half of the frame when there is one frame and an even
  // number of block rows
  lp.tail = tp.B == 1 && tp.bh % 2 == 0 && last.lane == 0 && last.pw_out.buf >= 0 && last.NB == nt && batch_splittable(last, 2) && P->ops.size() > lp.head.size() + 1 &&
            std::find(lp.head.begin(), lp.head.end(), P->ops.size() - 1) == lp.head.end();
  tp.tile0 = 0;
  lp.mid_tail = [&]() -> hipError_t {
    TileMapParams t2 = tp;
    t2.ntiles = nt / 2; t2.src = P->arena + ob.off; t2.dst = st + 2 * nf;
    hipError_t e = film_launch_tiles_to_frame(t2, s);
    if (e == hipSuccess) e = hipEventRecord(h->pipe_ev[1], s);
    return e;
  };
  HIPCHK(h, hipMemcpyAsync(st, x0, frame_bytes, hipMemcpyHostToDevice, s));
  tp.ntiles = nt; tp.src = st; tp.dst = P->arena + img0.off;
  HIPCHK(h, film_launch_frame_to_tiles(tp, s));
  HIPCHK(h, hipEventRecord(h->pipe_ev[0], s));
  HIPCHK(h, hipStreamWaitEvent(h->stream2, h->pipe_ev[0], 0));   // (the side stream: behind whatever `s` held before this call, too)
  for (size_t i : lp.head) HIPCHK(h, launch_op(batch_part(P->ops[i], 0, 2), P->arena, h->packed_dev, s));
  HIPCHK(h, hipMemcpyAsync(st + nf, x1, frame_bytes, hipMemcpyHostToDevice, h->stream2));
  tp.src = st + nf; tp.dst = P->arena + img0.off + (int64_t)nt * tp.TH * tp.TW * 3;
  HIPCHK(h, film_launch_frame_to_tiles(tp, h->stream2));
  HIPCHK(h, hipEventRecord(h->pipe_ev[0], h->stream2));
  HIPCHK(h, hipStreamWaitEvent(s, h->pipe_ev[0], 0));
  const hipError_t le = issue_lanes(h, P, s, false, &lp);
  if (le != hipSuccess) { (void)hipStreamSynchronize(h->stream2); return fail(h, FILM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le)); }
  h->last_plan = P;
  const int64_t tile_floats = (int64_t)tp.TH * tp.TW * 3;
  if (lp.tail) {
    TileMapParams t2 = tp;
    t2.tile0 = nt / 2; t2.ntiles = nt - nt / 2; t2.src = P->arena + ob.off + (int64_t)(nt / 2) * tile_floats; t2.dst = st + 2 * nf;
    HIPCHK(h, film_launch_tiles_to_frame(t2, s));
    const size_t half = frame_bytes / 2;   // (one frame, an even number of block rows: the upper half of the rows)
    HIPCHK(h, hipStreamWaitEvent(h->stream2, h->pipe_ev[1], 0));
    HIPCHK(h, hipMemcpyAsync(out, st + 2 * nf, half, hipMemcpyDeviceToHost, h->stream2));
    HIPCHK(h, hipMemcpyAsync((char*)out + half, (const char*)(st + 2 * nf) + half, frame_bytes - half, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(h->stream2));
  } else {
    tp.tile0 = 0; tp.ntiles = nt; tp.src = P->arena + ob.off; tp.dst = st + 2 * nf;
    HIPCHK(h, film_launch_tiles_to_frame(tp, s));
    HIPCHK(h, hipMemcpyAsync(out, st + 2 * nf, frame_bytes, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(h, hipStreamSynchronize(s));
  return FILM_OK;
}

hipStream_t pick_stream(film_t* h, int mem_kind, void* stream) {
  // stream == NULL: host buffers -> the handle's own (non-blocking) stream, synchronised before returning;
  // device buffers -> the NULL (legacy default) stream, i.e. ordered with the caller's default-stream work
  // (torch's default stream IS the NULL stream, and its handle is 0).
  return stream ? (hipStream_t)stream : (mem_kind == FILM_MEM_DEVICE ? (hipStream_t) nullptr : h->stream);
}

int forward_chunk(film_t* h, const float* x0, const float* x1, int B, int H, int W, float* out, int mem_kind, void* stream) {
  HIPCHK(h, hipSetDevice(h->device));
  Plan* P = nullptr;
  int rc = get_plan(h, B, H, W, true, &P);
  if (rc) return rc;
  hipStream_t s = pick_stream(h, mem_kind, stream);
  const size_t in_bytes = (size_t)B * H * W * 3 * sizeof(float);
  const hipMemcpyKind kin = mem_kind == FILM_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const hipMemcpyKind kout = mem_kind == FILM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  const Buffer& img0 = P->bufs[P->find("img0")];
  const Buffer& ob = P->bufs[P->find("out")];
  HIPCHK(h, hipMemcpyAsync(P->arena + img0.off, x0, in_bytes, kin, s));
  HIPCHK(h, hipMemcpyAsync(P->arena + img0.off + (int64_t)B * H * W * 3, x1, in_bytes, kin, s));
  rc = run_plan(h, P, s);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(out, P->arena + ob.off, in_bytes, kout, s));
  if (mem_kind == FILM_MEM_HOST) HIPCHK(h, hipStreamSynchronize(s));
  return FILM_OK;
}
}  // namespace

int film_get_tap(film_t* h, const char* name, float* dst, int64_t cap, int64_t dims[4]) {
  if (!h || !name) return FILM_ERR_INVALID;
  if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle has no device");
  Plan* P = h->last_plan;
  if (!P || !P->arena) return fail(h, FILM_ERR_STATE, "no forward has run yet");
  const int bi = P->find(name);
  if (bi < 0) return fail(h, FILM_ERR_NOTFOUND, "unknown tap '%s'", name);
  const Buffer& b = P->bufs[bi];
  for (const OpDesc& op : P->ops)   // a fused 1x1 head keeps this activation on chip: nothing ever writes the buffer
    if (op.kind == OP_CONV && op.pw_out.buf >= 0 && op.out.buf == bi)
      return fail(h, FILM_ERR_STATE, "tap '%s' is not materialised: the RGB head is fused into the layer that produces it "
                  "(film_set_option \"fuse\" without bit 16 keeps it)", name);
  if (dims) { dims[0] = b.N; dims[1] = b.H; dims[2] = b.W; dims[3] = b.C; }
  if (!dst) return FILM_OK;  // shape query
  if (cap < b.size()) return fail(h, FILM_ERR_INVALID, "capacity %lld < %lld floats", (long long)cap, (long long)b.size());
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  if (!b.planar) {
    HIPCHK(h, hipMemcpy(dst, P->arena + b.off, (size_t)b.size() * sizeof(float), hipMemcpyDeviceToHost));
    return FILM_OK;
  }
  // three pixel-major planes -> [N][H][W][C]
  std::vector<float> raw((size_t)b.size());
  HIPCHK(h, hipMemcpy(raw.data(), P->arena + b.off, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
  const int64_t npix = (int64_t)b.N * b.H * b.W;
  const int pc[3] = {b.planar, b.planar, b.C - 2 * b.planar};
  int64_t base = 0;
  int coff = 0;
  for (int part = 0; part < 3; ++part) {
    for (int64_t i = 0; i < npix; ++i) std::memcpy(dst + i * b.C + coff, raw.data() + base + i * pc[part], (size_t)pc[part] * sizeof(float));
    base += npix * pc[part];
    coff += pc[part];
  }
  return FILM_OK;
}

}  // extern "C"

