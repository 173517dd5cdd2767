// film_engine.cpp -- C-ABI (include/film_hip.h), weight store/packer, planner and executor of the
// MI355X FILM inference engine.
//
// The planner restates the *graph* of models/film_net/interpolator.py:89-207 as a static list of
// kernel launches over one workspace arena for a given (B,H,W):
//   image pyramids     util.py:23-45                -> pool ops on [2B,...,3] (both images in one batch)
//   feature extractor  feature_extractor.py:163-193 -> conv ops writing straight into the cascaded slots
//   flow estimator     pyramid_flow_estimator.py:125-163 -> both directions batched as 2B
//   flow synthesis     util.py:106-117              -> reuses the estimator's v (identical arithmetic)
//   warps + concat     interpolator.py:163-183      -> warp ops writing into the aligned pyramid
//   fusion             fusion.py:103-140            -> NN-upsample folded into the 2x2 conv's gather
// There is no CPU execution path here: plan-only handles (device = -1) can pack weights and describe
// plans, every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/film_hip.h"
#include "film_kernels.h"

namespace {

thread_local std::string g_create_error;

enum OpKind { OP_CONV = 0, OP_FLOW_HEAD, OP_CONV_PW, OP_POOL, OP_FLOW_UP, OP_FLOW_ADD, OP_WARP, OP_PACK_FLOW, OP_KINDS };
const char* kKindName[OP_KINDS] = {"conv_mfma", "flow_head", "conv_pw", "pool", "flow_up", "flow_add", "warp", "pack_flow"};

struct Buffer {
  std::string name;
  int64_t off;  // floats from arena base
  int N, H, W, C;      // all 0 for scratch regions (reinterpreted per use)
  int64_t floats;      // extent
  int64_t size() const { return floats; }
};

// A channel slice of (a batch range of) a workspace buffer, or of a scratch region.
struct View {
  int buf = -1;
  int64_t off = 0;  // floats from arena base to the first element of the view
  int stride = 0;   // floats per pixel
  int C = 0;
};

struct SegDesc {
  View v;
  int boff = 0, bmod = 0, up = 0;
};

struct OpDesc {
  int kind = 0;
  std::string tag;
  // conv
  SegDesc seg[FILM_MAX_SEG];
  int nseg = 0;
  int ksize = 1, leaky = 0, Cout = 0, Ctot = 0, tile = 0;
  int c3 = 0;                         // first-layer mode of the conv kernel (3-channel image input)
  int64_t w_off = 0, b_off = 0;
  int64_t w2_off = 0, b2_off = 0;     // flow_head: second 1x1 conv
  int64_t wh_off = -1;                // conv: the layer's conv_halo_kernel weight copy (-1: none)
  int64_t ws_off = -1;                // conv: the layer's bf16x6 weight copy
  int ksplit = 1;                     // conv (conv_buf_kernel): split-K factor, partial sums at part_off (film_kernels.h)
  int64_t part_off = 0;
  int fold = 0, py = 0, px = 0;       // conv: sub-pixel phase of a folded upsample + 2x2 conv (H, W = low-res grid)
  int ftaps = 0; int tdy[4] = {0, 0, 0, 0}, tdx[4] = {0, 0, 0, 0};
  int64_t fold_woff[4] = {0, 0, 0, 0};  // fold == 2: weight offset of phase q relative to w_off
  int64_t ww_off = -1;                // conv: the layer's Winograd F(2,3) weight copy
  int64_t wx_off = -1;                // conv: ... and its 2-plane bf16 split (precision mode bf16x3, conv_winox3_kernel)
  int64_t w43_off = -1;               // conv: the layer's Winograd F(4,3) weight copy (conv_wino43_kernel)
  int64_t w2d_off = -1;               // conv: the layer's nested F(4,3) x F(2,3) weight copy (conv_wino2d_kernel; deep-K layers only)
  int64_t wfx_off = -1;               // conv: phase-summed weights of a folded 2x2 layer as bf16 hi / mid (conv_foldx3_kernel)
  int wino = 0;                       // conv: 1 = runs on conv_wino_kernel, 2 = on conv_winox3_kernel (precision bf16x3), 3 = conv_wino43_kernel,
                                      //       4 = conv_wino2d_kernel (nested F(4,3) x F(2,3))
  int split = 0;                      // conv: runs on conv_halo_split_kernel (precision mode bf16x6)
  int lane = 0;                       // graph replay: 0 = main stream, 1 = side stream (small / HBM-bound work)
  std::vector<int> xdeps;             // ops on the OTHER lane this op must wait for (from the buffer overlap analysis)
  bool signal = false;                // some op on the other lane waits for this one
  int halo = 0;                       // conv: runs on conv_halo_kernel (decided by shape, see Planner::conv)
  // generic views
  View in, in2, out;
  View pack_b, pack_f, pack_out;   // warp: fused pack_flow (0.5 * flows into the aligned pyramid)
  View img_in, img_out;   // warp: fused 3-channel image warp with the same flow (t = 0.5 stage)
  View pw_out; int pw_cout = 0;   // conv: fused 1x1 convolution behind it (weights w2_off / b2_off) writes pw_out; `out` is not written then
  View in3, out2;   // warp: coarser flow to upsample / the upsampled flow it stores; flow heads: in2 = upsampled flow, out2 = v = out + in2
  int NB = 0, H = 0, W = 0;  // conv/warp: output dims; pool: input dims; flow_up: input dims
  float fscale = 1.f;
  int64_t n = 0;
  double flops = 0;  // algorithmic FLOPs (reference channel counts)
  double bytes = 0;  // algorithmic bytes (read once + write once)
};

struct LayerPack {
  std::string name;
  int kh, kw, cin, cout;     // reference shape
  std::vector<int> perm;     // internal input channel -> reference input channel, -1 = zero row
  bool c3 = false;           // first layer: packed as [12 tap slots][4][Cout] (row = tap*4 + channel, rest zero)
  // layers run by the MFMA conv kernel (Cout % 32 == 0) are packed K-contiguous per output channel:
  // [Cout][kh*kw*ctot] with k = tap*ctot + channel; the 1x1 heads keep [ctot][Cout]
  bool kmajor() const { return !c3 && cout % 32 == 0; }
  int64_t w_off = 0, b_off = 0;
  int64_t wh_off = -1;       // 3x3 K-major layers: second copy packed for conv_halo_kernel, [Cout][ctot/16][9][16]
  int64_t wf_off = -1;       // 2x2 layers behind a nearest upsample: the four sub-pixel phases, pre-summed weights,
                             //     phase (py,px) at wf_off + fold_phase_off(py,px): [Cout][ntaps_p * ctot], 9*ctot*cout in all
  int64_t ww_off = -1;       // ... the F(2,3)-along-x transformed copy for conv_wino_kernel, [Cout][ctot/8][12][8]
  int64_t wfx_off = -1;      // 2x2 layers after an upsample: the phase-summed weights as bf16 hi / mid for conv_foldx3_kernel,
                             //     [Cout][ctot/16][9 (tap, phase) steps][plane][16] bf16
  int64_t w43_off = -1;      // ... the F(4,3)-along-x transformed copy for conv_wino43_kernel, [Cout][ctot/8][3 dy][6 nu][8]
  int64_t w2d_off = -1;      // deep-K layers (has_w2d): the nested F(4,3)x x F(2,3)y copy for conv_wino2d_kernel,
                             //     [Cout/32][ctot/8][mu 4][nu 6][K half][32][4] (24 values per (ci, co): 2.67x the kernel)
  int64_t wx_off = -1;       // ... and the transformed copy split into bf16 hi / mid for conv_winox3_kernel,
                             //     [Cout][ctot/16][dy][j][h][plane][16] bf16 (nu = 2h + j)
  int64_t ws_off = -1;       // ... and the bf16x6 copy for conv_halo_split_kernel, [Cout][ctot/16][9][3][16] bf16
                             //     (offset in floats; 1.5 floats per weight)
  bool has_halo() const { return kmajor() && kh == 3 && kw == 3; }
  bool has_fold() const { return kmajor() && kh == 2 && kw == 2; }
  // conv_wino2d_kernel against the best 1-D F(4,3) tile of the same run (tools/conv_bench.hip, profiles/r03_conv_bench_w2d.log):
  // 1.12-1.24x at K = 384 ... 2448, 1.06x at 256 -> 256, 1.02x at 208 -> 64, 1.17x at 128 -> 32 (where the 1-D kernel's
  // 32-channel tile is weak), 0.97x at 128 -> 128, 0.85x at K = 64 (its activation staging per MFMA is 1.5x the 1-D kernel's)
  bool has_w2d() const { return has_halo() && (ctot() >= 208 || (ctot() >= 128 && cout == 32)); }
  int ctot() const { return (int)perm.size(); }
  int64_t packed_rows() const { return c3 ? 48 : (int64_t)kh * kw * ctot(); }
};

struct HostTensor {
  std::vector<int64_t> dims;
  std::vector<float> data;
};

struct Plan {
  int B = 0, H = 0, W = 0;
  std::vector<Buffer> bufs;
  std::vector<OpDesc> ops;
  int64_t arena_floats = 0;
  float* arena = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  hipGraph_t graph = nullptr;
  std::vector<hipEvent_t> ev;
  std::vector<hipEvent_t> lane_ev;    // one per signalling op (index = op index), lazily created; + fork/join at the end
  uint64_t last_use = 0;
  int find(const std::string& n) const {
    for (size_t i = 0; i < bufs.size(); ++i)
      if (bufs[i].name == n) return (int)i;
    return -1;
  }
};

}  // namespace

struct film_handle {
  void* stage = nullptr;       // device staging of whole frames for film_interpolate(FILM_MEM_HOST)
  size_t stage_bytes = 0;
  int device = -1;
  bool plan_only = true;
  film_config cfg{};
  hipStream_t stream = nullptr;
  std::string err;
  std::map<std::string, HostTensor> host_w;
  std::vector<LayerPack> layers;
  std::map<std::string, int> layer_idx;
  int64_t packed_floats = 0;          // floats of the PACKED PREFIX (groups [0, groups_packed)); group_end[3] = all layouts
  int64_t group_end[4] = {0, 0, 0, 0};  // end offset of layout group g (see film_create): 0 base, 1 F(2,3), 2 halo, 3 bf16 splits
  int groups_packed = 0;
  std::vector<float> packed_host;
  float* packed_dev = nullptr;
  bool finalized = false;
  std::vector<std::unique_ptr<Plan>> plans;
  Plan* last_plan = nullptr;
  uint64_t tick = 0;
  int opt_graph = 1, opt_profile = 0, opt_autotune = 1;
  int opt_max_batch = 0;  // 0: only the 4 GiB-per-buffer limit
  int opt_splitk = 1;     // 1: split-K (ksplit partial sums + ordered reduction) for the deep layers of levels with <= 1024 pixels
  int opt_fuse = 31;       // 1: flow_up fused into the flow-estimator warps, v = res + up into the flow heads (same arithmetic, 12 launches fewer)
  int opt_fold = 1;       // 1: nearest-upsample + 2x2 conv as four sub-pixel phase convolutions (9 taps per 4 outputs)
  int opt_wino = 1;       // 0: never, 1: Winograd kernels (F(4,3) / F(2,3)) where measured faster (default), 2 / 3: F(2,3) / F(4,3) on every eligible 3x3 conv
  int opt_halo_all = 0;   // 1: halo / split kernels for every eligible 3x3 conv regardless of size (tests, tuning)
  int opt_tune_ms = 0;    // autotune: minimum kernel time spent per candidate (0: two launches)
  int opt_lanes = 1;      // >= 1: replay graphs use a second (side) stream for independent small / HBM-bound work; 2: and (large frames) for
                          // the coarse decoder levels, emitted right behind the aligned levels they read (measured SLOWER: 48.3-48.4 ms
                          // against 47.4-47.6 ms per 1080p step, profiles/r03_lanes_ab.log - two matrix-bound streams share the CUs
                          // worse than one; kept as a tested option, not the default)
  hipStream_t stream2 = nullptr;
  int opt_precision = 0;  // 0: fp32 MFMA everywhere (default); 1: bf16x6 exact-split MFMA for the large 3x3 convs; 2: bf16x3
  int opt_wino2d = 1;     // nested Winograd kernel: 0 never, 1 (default) the deep-K layers of the large levels, 2 every layer that has the copy (tests)
  int opt_w43_shape = -1; // tests: >= 0 = every conv_wino43_kernel op that can run this Wino43Tile shape does (instead of the autotuned one)
  std::string profile_json;
  std::map<std::string, int> tune_cache;  // conv shape signature -> fastest tile
  std::map<std::string, int> tune_import; // choices of an earlier process (film_import_tune): taken, if still a candidate of
                                          // the op's kernel family, instead of timing the candidates again
};

extern "C" int film_ensure_groups_(film_t* h, int n);   // packs + uploads weight layout groups [groups_packed, n) on demand (internal)

namespace {

int fail(film_t* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(h, FILM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Architecture helpers (mirror frame-interpolation_amd/film_hip/weights.py)
// ---------------------------------------------------------------------------------------------
std::vector<int> feature_channels(const film_config& c) {  // feature_extractor.py:186-193
  std::vector<int> out;
  for (int l = 0; l < c.pyramid_levels; ++l) {
    int ch = 0;
    for (int j = 0; j < c.sub_levels; ++j)
      if (j <= l) ch += c.filters << j;
    out.push_back(ch);
  }
  return out;
}
int slot_offset(const film_config& c, int j) {  // channel offset of sub-pyramid stage j in a feature level
  int o = 0;
  for (int k = 0; k < j; ++k) o += c.filters << k;
  return o;
}
std::vector<int> fusion_filters(const film_config& c) {  // fusion.py:75-79
  std::vector<int> out;
  for (int i = 0; i < c.fusion_pyramid_levels - 1; ++i)
    out.push_back(i < c.specialized_levels ? (c.filters << i) : (c.filters << c.specialized_levels));
  return out;
}
std::string predictor_prefix(const film_config& c, int level) {  // pyramid_flow_estimator.py:109-123
  if (level < c.specialized_levels) return "predict_flow/flow_predictor_" + std::to_string(level);
  return "predict_flow/flow_predictor_shared";
}
int predictor_index(const film_config& c, int level) { return std::min(level, c.specialized_levels); }

int validate_config(film_t* h, const film_config& c) {
  if (c.pyramid_levels < 1 || c.pyramid_levels > 12) return fail(h, FILM_ERR_INVALID, "pyramid_levels out of range");
  if (c.pyramid_levels < c.fusion_pyramid_levels || c.fusion_pyramid_levels < 2)
    return fail(h, FILM_ERR_INVALID, "config.pyramid_levels must be greater than or equal to config.fusion_pyramid_levels.");
  if (c.specialized_levels < 1 || c.specialized_levels > c.pyramid_levels || c.specialized_levels > FILM_MAX_SPECIALIZED)
    return fail(h, FILM_ERR_INVALID, "specialized_levels out of range");
  if (c.sub_levels < 1 || c.sub_levels > c.specialized_levels + 1)
    return fail(h, FILM_ERR_INVALID, "sub_levels must be within [1, specialized_levels+1]");
  if (c.filters <= 0 || c.filters % 32) return fail(h, FILM_ERR_INVALID, "filters must be a positive multiple of 32");
  for (int i = 0; i <= c.specialized_levels; ++i) {
    int nf = c.flow_filters[i];
    if (nf <= 0 || nf % 32 || !(nf / 2 == 16 || (nf / 2) % 32 == 0))
      return fail(h, FILM_ERR_INVALID, "flow_filters[%d]=%d unsupported (need 32 or a multiple of 64)", i, nf);
    if (c.flow_convs[i] < 1) return fail(h, FILM_ERR_INVALID, "flow_convs[%d] must be >= 1", i);
  }
  return FILM_OK;
}

// ---------------------------------------------------------------------------------------------
// Layer table + packing
// ---------------------------------------------------------------------------------------------
std::vector<int> identity_perm(int n) {
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  return p;
}
// internal channel order of an aligned-pyramid level: [feat0 C | feat1 C | img0 3 | img1 3 | bflow 2 | fflow 2 | 0 x6]
// reference order (interpolator.py:167-183):          [img0 3 | feat0 C | img1 3 | feat1 C | bflow 2 | fflow 2]
std::vector<int> aligned_perm(int C) {
  std::vector<int> p;
  for (int c = 0; c < C; ++c) p.push_back(3 + c);
  for (int c = 0; c < C; ++c) p.push_back(3 + C + 3 + c);
  for (int j = 0; j < 3; ++j) p.push_back(j);
  for (int j = 0; j < 3; ++j) p.push_back(3 + C + j);
  for (int j = 0; j < 4; ++j) p.push_back(2 * (3 + C) + j);
  for (int j = 0; j < 6; ++j) p.push_back(-1);
  return p;
}

void build_layers(film_t* h) {
  const film_config& c = h->cfg;
  h->layers.clear();
  h->layer_idx.clear();
  auto add = [&](const std::string& name, int kh, int kw, int cin, int cout, std::vector<int> perm) {
    LayerPack L;
    L.name = name; L.kh = kh; L.kw = kw; L.cin = cin; L.cout = cout; L.perm = std::move(perm);
    h->layer_idx[name] = (int)h->layers.size();
    h->layers.push_back(std::move(L));
  };
  int cin = 3;
  for (int i = 0; i < c.sub_levels; ++i) {
    int k = c.filters << i;
    add("feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * i), 3, 3, cin, k, identity_perm(cin));
    if (i == 0) h->layers.back().c3 = true;
    add("feat_net/sub_extractor/cfeat_conv_" + std::to_string(2 * i + 1), 3, 3, k, k, identity_perm(k));
    cin = k;
  }
  auto fc = feature_channels(c);
  for (int p = 0; p <= c.specialized_levels; ++p) {
    std::string prefix = predictor_prefix(c, p);
    int ci = 2 * fc[std::min(p, c.pyramid_levels - 1)];
    int nf = c.flow_filters[p], nconv = c.flow_convs[p];
    for (int j = 0; j < nconv; ++j) {
      add(prefix + "/conv_" + std::to_string(j), 3, 3, ci, nf, identity_perm(ci));
      ci = nf;
    }
    add(prefix + "/conv_" + std::to_string(nconv), 1, 1, nf, nf / 2, identity_perm(nf));
    add(prefix + "/conv_" + std::to_string(nconv + 1), 1, 1, nf / 2, 2, identity_perm(nf / 2));
  }
  auto ff = fusion_filters(c);
  const int FL = c.fusion_pyramid_levels;
  for (int i = 0; i < FL - 1; ++i) {
    const int aligned_ref = 2 * (3 + fc[i]) + 4;
    std::vector<int> p0;
    int net_c;
    if (i == FL - 2) { net_c = 2 * (3 + fc[FL - 1]) + 4; p0 = aligned_perm(fc[FL - 1]); }
    else { net_c = ff[i + 1]; p0 = identity_perm(net_c); }
    add("fusion/convs_" + std::to_string(i) + "_0", 2, 2, net_c, ff[i], p0);
    std::vector<int> p1 = aligned_perm(fc[i]);
    for (int j = 0; j < ff[i]; ++j) p1.push_back(aligned_ref + j);
    add("fusion/convs_" + std::to_string(i) + "_1", 3, 3, aligned_ref + ff[i], ff[i], p1);
    add("fusion/convs_" + std::to_string(i) + "_2", 3, 3, ff[i], ff[i], identity_perm(ff[i]));
  }
  add("fusion/output_conv", 1, 1, ff[0], 3, identity_perm(ff[0]));
  // Weight layouts in four contiguous GROUPS, packed on demand (film_finalize packs group 0; the planner asks for the
  // others when a plan first needs them) so that the default fp32 path neither builds nor broadcasts the copies it never
  // reads:  0 = what the default plan runs on: K-major / first-layer / 1x1 layouts + biases, the phase-summed 2x2
  //             layers, the F(4,3) copy                                                     (3.1x the parameters)
  //         1 = F(2,3) copy (conv_wino_kernel: levels narrower than the F(4,3) patches)
  //         2 = halo copy (conv_halo_kernel: option winograd = 0 / halo_all)
  //         3 = bf16 split copies (precision modes bf16x6 / bf16x3)
  int64_t off = 0;
  auto al = [&]() { off = (off + 3) & ~int64_t(3); };
  for (auto& L : h->layers) {
    L.w_off = off; off += L.packed_rows() * L.cout; al();
    L.b_off = off; off += L.cout; al();
    if (L.has_fold()) { L.wf_off = off; off += (int64_t)9 * L.ctot() * L.cout; al(); }
    if (L.has_halo()) { L.w43_off = off; off += L.packed_rows() * L.cout / 9 * 18; al(); }
    if (L.has_w2d()) { L.w2d_off = off; off += L.packed_rows() * L.cout / 9 * 24; al(); }
  }
  h->group_end[0] = off;
  for (auto& L : h->layers)
    if (L.has_halo()) { L.ww_off = off; off += L.packed_rows() * L.cout / 9 * 12; al(); }
  h->group_end[1] = off;
  for (auto& L : h->layers)
    if (L.has_halo()) { L.wh_off = off; off += L.packed_rows() * L.cout; al(); }
  h->group_end[2] = off;
  for (auto& L : h->layers) {
    if (L.has_fold()) { L.wfx_off = off; off += (int64_t)9 * L.ctot() * L.cout; al(); }
    if (L.has_halo()) {
      L.ws_off = off; off += (L.packed_rows() * L.cout * 3 + 1) / 2; al();
      L.wx_off = off; off += L.packed_rows() * L.cout / 9 * 12; al();
    }
  }
  h->group_end[3] = off;
  h->packed_floats = 0;
  h->groups_packed = 0;
}

// ---------------------------------------------------------------------------------------------
// Planner
// ---------------------------------------------------------------------------------------------
struct Planner {
  film_t* h;
  Plan* P;
  int64_t cursor = 0;

  bool bad = false;
  std::string bad_msg;
  bool w2d_ok = true;   // cleared by the caller for a layer whose epilogue fusion (average pool) only conv_wino43_kernel has

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
    op.w_off = L.w_off; op.b_off = L.b_off; op.wh_off = L.wh_off; op.ws_off = L.ws_off; op.ww_off = L.ww_off; op.wx_off = L.wx_off; op.wfx_off = L.wfx_off; op.w43_off = L.w43_off; op.w2d_off = L.w2d_off;
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
      f.flops = 2.0 * NB * H * W * L.cout * L.kh * L.kw * L.cin;   // algorithmic FLOPs of the reference op
      f.bytes = 4.0 * NB * H * W * (L.cin / 4.0 + L.cout);
      P->ops.push_back(f);
      return;
    }
    op.out = out; op.NB = NB; op.H = H; op.W = W;
    const int64_t M = (int64_t)NB * H * W;
    // Kernel family by layer shape only (never by timing, and not by the batch size): the two kernels sum K in a
    // different order, so the choice must be a pure function of the layer for results to be reproducible across
    // batch sizes and runs.  Halo staging pays where K is deep (traffic bound) or N is too narrow to amortise the
    // per-tap A gather; measured in tools/conv_bench.hip.
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
    const bool w43_width = L.w43_off >= 0 && (h->opt_wino == 3 || 64 * ((W + 63) / 64) * 100 <= 115 * W);
    // deep K on a small level (the 36x60 level of a 1080p tile, K = 1920): only with F(4,3) AND its split-K, which cuts the
    // few long workgroups of such a layer into enough pieces to fill the chip
    const bool deep_small = L.cout % 128 == 0 && px >= 2048 && px < 8192 && ctot > 1024 && w43_width && h->opt_splitk &&
                            h->opt_wino == 1 && h->opt_precision == 0;
    op.wino = op.split != 1 && L.ww_off >= 0 && !any_up && h->opt_wino != 0 &&
              ((L.cout % 128 == 0 && (px >= 8192 || (px >= 2048 && ctot <= 1024))) || (L.cout % 64 == 0 && px >= 30000) ||
               px >= 100000 || h->opt_wino >= 2 || deep_small);
    if (op.wino && h->opt_precision == 2 && (op.split == 2 || h->opt_wino >= 2)) {
      // wino: 1 = fp32 conv_wino_kernel, 2 = conv_winox3_kernel.  The Winograd form wins with the 2 x 2 wave block of
      // its 128-channel tile (0.88-0.94x the time of conv_halo_split_kernel<..,3> per layer, 427 vs 367 TFLOP/s at
      // K = 22 032) and loses with the 64-channel tiles (1.08-1.30x: twice the A staging per MFMA) - per-op profiles of
      // the two plans and tools/conv_bench.hip agree.
      if (L.cout % 128 == 0 || h->opt_wino >= 2) op.split = 0, op.wino = 2;
      else if (op.split == 2) op.wino = 0;
    }
    // fp32: F(4,3) along x (conv_wino43_kernel, 2x fewer MFMAs than direct where F(2,3) has 1.5x) on the levels whose width
    // fills its 64-pixel patches (the Q16 tiles; at most 15 % of the last patch of a row empty: 960 ... 60, 448, 256 ...); wino = 3.  "winograd" = 2 / 3 force
    // F(2,3) / F(4,3) onto every eligible layer (tests).
    if (op.wino == 1 && h->opt_wino != 2 && w43_width) op.wino = 3;
    // Nested F(4,3)x x F(2,3)y (conv_wino2d_kernel, 1.5x fewer MFMAs again): the deep-K layers (K >= 384: the first layer of
    // every flow predictor but level 0's, the wide layer of every decoder level but level 0's, cfeat_conv_7) on levels large
    // enough to fill the chip without split-K.  A family of its own (a function of the layer and the level size only).
    if (L.w2d_off >= 0 && w2d_ok && !any_up && h->opt_precision == 0 &&
        (h->opt_wino2d == 2 || (h->opt_wino2d == 1 && op.wino == 3 && h->opt_wino == 1 && px >= 8192)))
      op.wino = 4;
    if (op.split || op.wino) op.halo = 0;
    need_groups(op.split || op.wino == 2 ? 4 : op.halo ? 3 : op.wino == 1 ? 2 : 1);
    op.tile = op.wino == 4 ? ((L.cout % 64 == 0 ? W2D_Q8_8x64 : W2D_Q8_8x32) | CONV_TILE_W2D | CONV_TILE_XCD)
              : op.wino == 3 ? ((L.cout % 64 == 0 ? W43_Q16_4x64_T21_P2 : W43_Q16_4x32_T11_BG) | CONV_TILE_WINO | CONV_TILE_F43 | CONV_TILE_XCD)
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
    for (int l = 0; l < FL; ++l) aligned[l] = add_buffer("aligned" + std::to_string(l), B, HL(l), WL(l), 2 * fc[l] + 16);
    for (int i = 0; i < FL - 1; ++i) {
      fu_u[i] = add_buffer("fusion_up" + std::to_string(i), B, HL(i), WL(i), ff[i]);
      fu_a[i] = add_buffer("fusion_a" + std::to_string(i), B, HL(i), WL(i), ff[i]);
      fu_b[i] = add_buffer("fusion_b" + std::to_string(i), B, HL(i), WL(i), ff[i]);
    }
    const int out = add_buffer("out", B, H, W, 3);
    P->arena_floats = cursor;

    // ---- image pyramids (util.py:23-45), both images as one batch of 2B ----------------------
    for (int l = 0; l + 1 < L; ++l)
      pool("image_pyramid_l" + std::to_string(l + 1), view(img[l], 0, 0, 3), view(img[l + 1], 0, 0, 3), N2, HL(l), WL(l));

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
          op.tile = ((k == 64 || k == 32) ? TILE_C3_DIRECT : k % 64 == 0 ? TILE_256x64 : TILE_256x32) | CONV_TILE_XCD | CONV_TILE_C3;
          op.flops = 2.0 * N2 * HL(lv) * WL(lv) * k * 27; op.bytes = 4.0 * N2 * HL(lv) * WL(lv) * (3 + k);
          P->ops.push_back(op);
        } else {
          SegDesc s; s.v = scratch(fx_p, k >> 1);
          conv(tg, w0, {s}, tmp, N2, HL(lv), WL(lv), true);
        }
        SegDesc s1; s1.v = tmp;
        View dst = view(feat[lv], 0, slot_offset(c, j), k);
        w2d_ok = !(j < n - 1 && (h->opt_fuse & 8));   // a pooled stage keeps the kernel that fuses the pool into its epilogue
        conv(tg, w1, {s1}, dst, N2, HL(lv), WL(lv), true);
        w2d_ok = true;
        if (j < n - 1) {
          OpDesc& cv = P->ops.back();
          if ((h->opt_fuse & 8) && cv.kind == OP_CONV && cv.wino == 3 && cv.ksplit <= 1 && !(HL(lv) & 1) && !(WL(lv) & 1)) {
            // AveragePooling2D in the epilogue of the F(4,3) kernel (its 64-pixel tiles hold both rows of a 2x2 block)
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
        if (!(h->opt_fuse & 1)) {
          OpDesc up;
          up.kind = OP_FLOW_UP; up.tag = tg + ":resize2x";
          up.in = view(v[l + 1], 0, 0, 2); up.out = view(vup[l], 0, 0, 2);
          up.NB = N2; up.H = HL(l + 1); up.W = WL(l + 1);
          up.bytes = 4.0 * N2 * Hl * Wl * 2 * 1.25;
          P->ops.push_back(up);
        }
        for (int d = 0; d < 2; ++d) {  // warp the OTHER image's features with this direction's flow
          warp(tg + ":warp_d" + std::to_string(d), view(feat[l], (1 - d) * B, 0, fc[l]), view(vup[l], d * B, 0, 2),
               view(warped[l], d * B, 0, fc[l]), B, Hl, Wl, 1.f);
          if (h->opt_fuse & 1) {
            // tf.image.resize(2 * v) (pyramid_flow_estimator.py:155) inside the warp: the flow of this level is computed
            // from the coarser level's v by every thread of a pixel and stored once (to vup, which v = res + up reads)
            OpDesc& w = P->ops.back();
            w.tag += "+resize2x";
            w.in2 = View();
            w.in3 = view(v[l + 1], d * B, 0, 2);
            w.out2 = view(vup[l], d * B, 0, 2);
          }
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
      for (int s = 0; s < 2; ++s) {
        View fl = view(v[l], (1 - s) * B, 0, 2);
        warp(tg + ":warp_feat" + std::to_string(s), view(feat[l], s * B, 0, fc[l]), fl,
             view(aligned[l], 0, s * fc[l], fc[l]), B, HL(l), WL(l), 0.5f);
        if (h->opt_fuse & 4) {   // the 3-channel image rides in the same launch (extra row units)
          OpDesc& w = P->ops.back();
          w.tag += "+img";
          w.img_in = view(img[l], s * B, 0, 3);
          w.img_out = view(aligned[l], 0, 2 * fc[l] + 3 * s, 3);
          w.bytes += 4.0 * B * HL(l) * WL(l) * 6.0;
        } else
        warp(tg + ":warp_img" + std::to_string(s), view(img[l], s * B, 0, 3), fl,
             view(aligned[l], 0, 2 * fc[l] + 3 * s, 3), B, HL(l), WL(l), 0.5f, false);
      }
      if (h->opt_fuse & 4) {   // 0.5 * flows ride in the second feature warp of the level
        OpDesc& w = P->ops.back();
        w.tag += "+flows";
        w.pack_b = view(v[l], B, 0, 2);   // backward flow (d = 1)
        w.pack_f = view(v[l], 0, 0, 2);   // forward flow  (d = 0)
        w.pack_out = view(aligned[l], 0, 2 * fc[l] + 6, 10);
      } else {
      OpDesc pk;
      pk.kind = OP_PACK_FLOW; pk.tag = tg + ":flows";
      pk.in = view(v[l], B, 0, 2);   // backward flow (d = 1)
      pk.in2 = view(v[l], 0, 0, 2);  // forward flow  (d = 0)
      pk.out = view(aligned[l], 0, 2 * fc[l] + 6, 10);
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
      SegDesc s0; s0.v = view(aligned[i], 0, 0, 2 * fc[i] + 16);
      SegDesc s1; s1.v = view(fu_u[i], 0, 0, ff[i]);
      conv(tg, base + "_1", {s0, s1}, view(fu_a[i], 0, 0, ff[i]), B, HL(i), WL(i), true);
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
      // layer's epilogue when it runs on conv_wino43_kernel with 64 output channels and no split: the 64-channel
      // activation (566 MB per 1080p step) is then neither written nor read back.
      OpDesc& last = P->ops.back();
      const LayerPack& LO = h->layers[h->layer_idx.at("fusion/output_conv")];
      if ((h->opt_fuse & 16) && last.kind == OP_CONV && last.wino == 3 && last.Cout == 64 && last.ksplit <= 1 && last.out2.buf < 0 &&
          LO.cout <= 4 && LO.cin == 64) {
        last.tag += "+output_conv";
        last.pw_out = view(out, 0, 0, 3); last.pw_cout = LO.cout;
        last.w2_off = LO.w_off; last.b2_off = LO.b_off;
        last.tile = W43_Q16_4x64_N1_P2 | CONV_TILE_WINO | CONV_TILE_F43 | CONV_TILE_XCD;
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
    if (op.pack_out.buf >= 0) wr.push_back(access(op.pack_out));
  }
  // For every op: the LAST op of the other lane it conflicts with (RAW, WAR or WAW on overlapping channels of a
  // buffer).  Waiting for the last one is enough: a lane executes in program order.
  void analyze_lanes() {
    const size_t n = P->ops.size();
    std::vector<std::vector<Access>> rd(n), wr(n);
    for (size_t i = 0; i < n; ++i) accesses(P->ops[i], rd[i], wr[i]);
    for (size_t j = 0; j < n; ++j) {
      OpDesc& oj = P->ops[j];
      for (size_t ii = j; ii-- > 0;) {
        const OpDesc& oi = P->ops[ii];
        if (oi.lane == oj.lane) continue;
        bool hit = false;
        for (const Access& w : wr[ii]) {
          for (const Access& r : rd[j]) hit |= overlap(w, r);
          for (const Access& w2 : wr[j]) hit |= overlap(w, w2);
        }
        for (const Access& r : rd[ii])
          for (const Access& w2 : wr[j]) hit |= overlap(r, w2);
        if (hit) { oj.xdeps.push_back((int)ii); P->ops[ii].signal = true; break; }
      }
    }
  }
};

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
      if (op.in3.buf >= 0) { p.coarse = cptr(arena, op.in3); p.flow_out = mptr(arena, op.out2); }
      if (op.img_in.buf >= 0) {
        p.src3 = cptr(arena, op.img_in); p.s3stride = op.img_in.stride;
        p.dst3 = mptr(arena, op.img_out); p.d3stride = op.img_out.stride;
      }
      if (op.pack_out.buf >= 0) {
        p.pack_b = cptr(arena, op.pack_b); p.pack_f = cptr(arena, op.pack_f);
        p.pack_dst = mptr(arena, op.pack_out); p.pack_stride = op.pack_out.stride;
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

// float -> bfloat16, round to nearest even (what v_cvt_pk_bf16_f32 does; weights are finite)
static inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
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
    for (int sh : {W43_Q16_4x64_N1, W43_Q16_4x64_N1_P2, W43_Q8_8x64_N1_P2}) { out.push_back(w43_tile(sh)); out.push_back(w43_tile(sh) | CONV_TILE_XCD); }
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
    if (pool && (sh == W43_4x64_T21 || sh == W43_4x64_T12 || sh == W43_4x32_T11)) continue;   // the fused pool needs a <= 64-pixel tile
    out.push_back(w43_tile(sh)); out.push_back(w43_tile(sh) | CONV_TILE_XCD);
  }
  return out;
}

std::vector<int> wino2d_candidates(int Cout) {
  std::vector<int> shapes = Cout % 64 == 0 ? std::vector<int>{W2D_Q8_8x64, W2D_Q8_8x32, W2D_Q16_4x64, W2D_Q16_4x32} : std::vector<int>{W2D_Q8_8x32, W2D_Q16_4x32};
  std::vector<int> out;
  for (int sh : shapes) { out.push_back(sh | CONV_TILE_W2D); out.push_back(sh | CONV_TILE_W2D | CONV_TILE_XCD); }
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
  std::vector<int> cands = (op.fold == 2 && op.split == 2) ? foldx3_candidates(op.Cout) : op.wino == 4 ? wino2d_candidates(op.Cout) : op.wino == 3 ? wino43_candidates(op.Cout, op.out2.buf >= 0, op.pw_out.buf >= 0) : op.wino == 2 ? winox3_candidates(op.Cout) : op.wino ? wino_candidates(op.Cout) : op.split ? split_candidates(op.Cout, op.split == 2) : op.halo ? halo_candidates(op.Cout) : tile_candidates(op.Cout);
  if (op.c3) {
    // conv_c3_kernel and the 3-channel mode of conv_igemm_kernel pair the K = 27 products differently (different
    // rounding): one family per layer shape, never a timing decision - the direct kernel wherever it exists
    cands.clear();
    if (op.Cout == 64 || op.Cout == 32) cands.push_back(TILE_C3_DIRECT | CONV_TILE_C3);
    else
      for (int sh : (op.Cout % 64 == 0 ? std::vector<int>{TILE_256x64, TILE_128x64} : std::vector<int>{TILE_256x32, TILE_128x32})) {
        cands.push_back(sh | CONV_TILE_C3);
        cands.push_back(sh | CONV_TILE_C3 | CONV_TILE_XCD);
      }
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
  std::unique_ptr<Plan> P(new Plan);
  Planner pl{h, P.get()};
  int rc = pl.build(B, H, W);
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
  P->last_use = ++h->tick;
  *out = P.get();
  h->plans.push_back(std::move(P));
  return FILM_OK;
}

// ---------------------------------------------------------------------------------------------
// JSON helpers
// ---------------------------------------------------------------------------------------------
int64_t limited_buffer_bytes(const Plan* P);   // (below, next to the batch-chunking rule)

void json_view(std::ostringstream& o, const char* key, const View& v, const Plan& P) {
  o << "\"" << key << "\":{\"buf\":\"" << (v.buf >= 0 ? P.bufs[v.buf].name : std::string("")) << "\",\"off\":" << v.off
    << ",\"stride\":" << v.stride << ",\"C\":" << v.C << "}";
}

std::string plan_json(film_t* h, const Plan& P) {
  std::ostringstream o;
  o << "{\"B\":" << P.B << ",\"H\":" << P.H << ",\"W\":" << P.W << ",\"arena_floats\":" << P.arena_floats
    << ",\"offset32_buffer_bytes\":" << limited_buffer_bytes(&P)
    << ",\"packed_floats\":" << h->packed_floats << ",\"buffers\":[";
  for (size_t i = 0; i < P.bufs.size(); ++i) {
    const Buffer& b = P.bufs[i];
    o << (i ? "," : "") << "{\"name\":\"" << b.name << "\",\"off\":" << b.off << ",\"N\":" << b.N << ",\"H\":" << b.H
      << ",\"W\":" << b.W << ",\"C\":" << b.C << ",\"floats\":" << b.floats << "}";
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
      << ",\"b_off\":" << op.b_off << ",\"wh_off\":" << op.wh_off << ",\"halo\":" << op.halo << ",\"ws_off\":" << op.ws_off << ",\"split\":" << op.split << ",\"ww_off\":" << op.ww_off << ",\"wx_off\":" << op.wx_off << ",\"wfx_off\":" << op.wfx_off << ",\"w43_off\":" << op.w43_off << ",\"w2d_off\":" << op.w2d_off << ",\"wino\":" << op.wino << ",\"fold\":" << op.fold << ",\"ksplit\":" << op.ksplit << ",\"py\":" << op.py
      << ",\"px\":" << op.px << ",\"ftaps\":" << op.ftaps << ",\"tdy\":[" << op.tdy[0] << "," << op.tdy[1] << "," << op.tdy[2] << "," << op.tdy[3]
      << "],\"tdx\":[" << op.tdx[0] << "," << op.tdx[1] << "," << op.tdx[2] << "," << op.tdx[3] << "]"
      << ",\"fold_woff\":[" << op.fold_woff[0] << "," << op.fold_woff[1] << "," << op.fold_woff[2] << "," << op.fold_woff[3] << "]" << ",\"lane\":" << op.lane << ",\"xdeps\":["
      << [&] { std::string d; for (size_t q = 0; q < op.xdeps.size(); ++q) d += (q ? "," : "") + std::to_string(op.xdeps[q]); return d; }() << "]" << ",\"w2_off\":" << op.w2_off << ",\"b2_off\":" << op.b2_off << ",\"c3\":" << op.c3
      << ",\"fscale\":" << op.fscale << ",\"n\":" << op.n << ",\"flops\":" << op.flops
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
    json_view(o, "pack_out", op.pack_out, P); o << ",";
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

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8.  Host-side helper of the SavedModel
// variables reader (film_hip/tf_bundle.py): TensorFlow stores a masked crc32c per tensor and per index block.
uint32_t film_crc32c(uint32_t crc, const void* data, int64_t n) {
  static uint32_t tab[8][256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
    init = true;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = tab[7][v & 0xFF] ^ tab[6][(v >> 8) & 0xFF] ^ tab[5][(v >> 16) & 0xFF] ^ tab[4][(v >> 24) & 0xFF] ^
        tab[3][(v >> 32) & 0xFF] ^ tab[2][(v >> 40) & 0xFF] ^ tab[1][(v >> 48) & 0xFF] ^ tab[0][(v >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

const char* film_version(void) { return "gfx950;film_hip r3"; }

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
    (void)hipStreamSynchronize(h->stream);
  }
  for (auto& p : h->plans) free_plan(p.get());
  if (h->packed_dev) (void)hipFree(h->packed_dev);
  if (h->stage) (void)hipFree(h->stage);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  delete h;
}

const char* film_last_error(const film_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int film_set_weight(film_t* h, const char* name, const float* data, const int64_t* dims, int ndim) {
  if (!h || !name || !data || !dims || ndim < 1 || ndim > 4) return fail(h, FILM_ERR_INVALID, "bad argument");
  std::string nm(name);
  const size_t slash = nm.rfind('/');
  if (slash == std::string::npos) return fail(h, FILM_ERR_NOTFOUND, "unknown weight '%s'", name);
  const std::string layer = nm.substr(0, slash), kind = nm.substr(slash + 1);
  auto it = h->layer_idx.find(layer);
  if (it == h->layer_idx.end() || (kind != "kernel" && kind != "bias")) return fail(h, FILM_ERR_NOTFOUND, "unknown weight '%s'", name);
  const LayerPack& L = h->layers[it->second];
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  if (kind == "kernel") {
    if (ndim != 4 || dims[0] != L.kh || dims[1] != L.kw || dims[2] != L.cin || dims[3] != L.cout)
      return fail(h, FILM_ERR_INVALID, "%s: expected HWIO [%d,%d,%d,%d]", name, L.kh, L.kw, L.cin, L.cout);
  } else if (ndim != 1 || dims[0] != L.cout) {
    return fail(h, FILM_ERR_INVALID, "%s: expected [%d]", name, L.cout);
  }
  HostTensor t;
  t.dims.assign(dims, dims + ndim);
  t.data.assign(data, data + n);
  h->host_w[nm] = std::move(t);
  h->finalized = false;
  return FILM_OK;
}

// Uploads the floats [from, to) of the packed blob.  The device buffer is sized for every layout group once (1.1 GB of
// 288): groups packed later land at their fixed offsets and no plan has to be rebuilt.
static int upload_packed(film_t* h, int64_t from, int64_t to) {
  if (h->plan_only || to <= from) return FILM_OK;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->packed_dev) HIPCHK(h, hipMalloc(&h->packed_dev, (size_t)h->group_end[3] * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->packed_dev + from, h->packed_host.data() + from, (size_t)(to - from) * sizeof(float), hipMemcpyHostToDevice));
  return FILM_OK;
}

// Packs the layouts of `group` of layer L for the output channels [co0, co1) (the unit of work of the packing threads;
// every output channel owns disjoint ranges of every layout).  Layers without a per-channel layout (first layer, 1x1
// heads) and the biases are handled by the caller of co0 == 0.
static void pack_layer_group(film_t* h, const LayerPack& L, int group, int co0, int co1) {
  const float* src = h->host_w.at(L.name + "/kernel").data.data();
  float* const base = h->packed_host.data();
  const int ct = L.ctot();
  if (group == 0 && co0 == 0) {
    memcpy(base + L.b_off, h->host_w.at(L.name + "/bias").data.data(), sizeof(float) * L.cout);
    float* dst = base + L.w_off;
    if (L.c3) {
      for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < 3; ++c)
          memcpy(dst + ((size_t)tap * 4 + c) * L.cout, src + ((size_t)tap * 3 + c) * L.cout, sizeof(float) * L.cout);
    } else if (!L.kmajor()) {
      for (int tap = 0; tap < L.kh * L.kw; ++tap)
        for (int ci = 0; ci < ct; ++ci) {
          const int ref = L.perm[ci];
          if (ref < 0) continue;  // zero row (padding channel)
          memcpy(dst + ((size_t)tap * ct + ci) * L.cout, src + ((size_t)tap * L.cin + ref) * L.cout, sizeof(float) * L.cout);
        }
    }
  }
  if (!L.kmajor()) return;
  const int ntap = L.kh * L.kw;
  const size_t ktot = (size_t)ntap * ct;
  const size_t nkc = (size_t)ct / 16;
  // ---- K-major copy (group 0), halo copy (group 2), bf16x6 planes (group 3): one pass per (tap, 16-channel chunk); the
  // 16 source rows (each `cout` contiguous floats) stay in L1 while every output channel receives its 16 k values
  float* dk = group == 0 ? base + L.w_off : nullptr;
  float* dh = group == 2 && L.wh_off >= 0 ? base + L.wh_off : nullptr;
  uint16_t* ds = group == 3 && L.ws_off >= 0 ? reinterpret_cast<uint16_t*>(base + L.ws_off) : nullptr;
  if (dk || dh || ds)
    for (int tap = 0; tap < ntap; ++tap)
      for (size_t kc = 0; kc < nkc; ++kc) {
        const float* rows[16];
        for (int j = 0; j < 16; ++j) {
          const int ref = L.perm[kc * 16 + j];
          rows[j] = ref < 0 ? nullptr : src + ((size_t)tap * L.cin + ref) * L.cout;  // nullptr: zero (padding) channel
        }
        for (int co = co0; co < co1; ++co) {
          float v[16];
          for (int j = 0; j < 16; ++j) v[j] = rows[j] ? rows[j][co] : 0.f;
          if (dk) memcpy(dk + (size_t)co * ktot + (size_t)tap * ct + kc * 16, v, sizeof(v));
          if (dh) memcpy(dh + (((size_t)co * nkc + kc) * 9 + tap) * 16, v, sizeof(v));
          if (ds) {  // exact 3-way bf16 split, round-to-nearest-even pieces (same as conv_split4 on the device)
            uint16_t* d = ds + (((size_t)co * nkc + kc) * 9 + tap) * 48;
            for (int j = 0; j < 16; ++j) {
              const uint16_t hb = bf16_rne(v[j]);
              const float r = v[j] - bf16_to_float(hb);
              const uint16_t mb = bf16_rne(r);
              const float q = r - bf16_to_float(mb);
              d[j] = hb; d[16 + j] = mb; d[32 + j] = bf16_rne(q);
            }
          }
        }
      }
  // ---- sub-pixel phases of upsample + 2x2: weights of the taps that read the same input pixel, summed (fp32: group 0;
  // bf16 hi / mid for conv_foldx3_kernel: group 3)
  if (L.wf_off >= 0 && (group == 0 || group == 3)) {
    float* df = base + L.wf_off;
    uint16_t* dfx = reinterpret_cast<uint16_t*>(base + L.wfx_off);
    const size_t nk16f = (size_t)ct / 16;
    // step of (tap a*2+b, phase py*2+px) in conv_foldx3_kernel's order (taps 00 00 00 | 00 01 01 | 10 10 11)
    static const int kFoldStep[4][4] = {{0, 1, 2, 3}, {-1, 4, -1, 5}, {-1, -1, 6, 7}, {-1, -1, -1, 8}};
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int nt = (py + 1) * (px + 1);
        const size_t kph = (size_t)nt * ct;
        int t = 0;
        for (int a = 0; a <= py; ++a)
          for (int b = 0; b <= px; ++b, ++t)
            for (int ci = 0; ci < ct; ++ci) {
              const int ref = L.perm[ci];
              if (ref < 0) continue;
              for (int co = co0; co < co1; ++co) {
                float acc = 0.f;  // kernel taps (dy, dx) with (py & dy) == a and (px & dx) == b, in raster order
                for (int dy = 0; dy < 2; ++dy)
                  for (int dx = 0; dx < 2; ++dx)
                    if ((py & dy) == a && (px & dx) == b) acc += src[((size_t)(dy * 2 + dx) * L.cin + ref) * L.cout + co];
                if (group == 0) df[(size_t)co * kph + (size_t)t * ct + ci] = acc;
                else {
                  const int step = kFoldStep[a * 2 + b][py * 2 + px];
                  uint16_t* d = dfx + (((size_t)co * nk16f + ci / 16) * 9 + step) * 32 + ci % 16;
                  const uint16_t hb = bf16_rne(acc);
                  d[0] = hb;
                  d[16] = bf16_rne(acc - bf16_to_float(hb));
                }
              }
            }
        df += kph * L.cout;
      }
  }
  // ---- nested Winograd copy (group 0, deep-K layers): U[mu][nu] = the F(2,3) transform along dy of the F(4,3)-transformed
  // kernel rows u_nu(dy) (the same u as the w43 copy), [Cout/32][chunk8][mu 4][nu 6][K half][32][4]
  if (group == 0 && L.w2d_off >= 0) {
    float* d2 = base + L.w2d_off;
    const size_t nk8 = (size_t)ct / 8;
    for (size_t kc = 0; kc < nk8; ++kc) {
      const float* rows[3][3][8];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          for (int j = 0; j < 8; ++j) {
            const int ref = L.perm[kc * 8 + j];
            rows[dy][dx][j] = ref < 0 ? nullptr : src + ((size_t)(dy * 3 + dx) * L.cin + ref) * L.cout;
          }
      for (int co = co0; co < co1; ++co)
        for (int j = 0; j < 8; ++j) {
          float u[3][6];
          for (int dy = 0; dy < 3; ++dy) {
            const float g0 = rows[dy][0][j] ? rows[dy][0][j][co] : 0.f, g1 = rows[dy][1][j] ? rows[dy][1][j][co] : 0.f,
                        g2 = rows[dy][2][j] ? rows[dy][2][j][co] : 0.f;
            const float e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), o = g1 * (1.f / 12.f);
            u[dy][0] = g0 * 0.25f;
            u[dy][1] = -((g0 + g2) + g1) * (1.f / 6.f);
            u[dy][2] = -((g0 + g2) - g1) * (1.f / 6.f);
            u[dy][3] = e + o;
            u[dy][4] = e - o;
            u[dy][5] = g2;
          }
          for (int nu = 0; nu < 6; ++nu) {
            const float U[4] = {u[0][nu], ((u[0][nu] + u[2][nu]) + u[1][nu]) * 0.5f, ((u[0][nu] + u[2][nu]) - u[1][nu]) * 0.5f, u[2][nu]};
            for (int mu = 0; mu < 4; ++mu)
              d2[(((((size_t)(co / 32) * nk8 + kc) * 4 + mu) * 6 + nu) * 2 + j / 4) * 128 + (co % 32) * 4 + j % 4] = U[mu];
          }
        }
    }
  }
  // ---- Winograd copies along x: F(4,3) [Cout][chunk8][dy][nu 6][8] (group 0), F(2,3) [Cout][chunk8][nu*3+dy][8]
  // (u0 = g0, u1 = ((g0+g2)+g1)/2, u2 = ((g0+g2)-g1)/2, u3 = g2; group 1) and its bf16 hi / mid planes (group 3)
  if (L.ww_off >= 0 && (group == 0 || group == 1 || group == 3)) {
    float* dw = base + L.ww_off;
    uint16_t* dx3 = reinterpret_cast<uint16_t*>(base + L.wx_off);
    float* d43 = base + L.w43_off;
    const size_t nk8 = (size_t)ct / 8, nk16 = (size_t)ct / 16;
    for (int dy = 0; dy < 3; ++dy)
      for (size_t kc = 0; kc < nk8; ++kc) {
        const float* rows[3][8];
        for (int dx = 0; dx < 3; ++dx)
          for (int j = 0; j < 8; ++j) {
            const int ref = L.perm[kc * 8 + j];
            rows[dx][j] = ref < 0 ? nullptr : src + ((size_t)(dy * 3 + dx) * L.cin + ref) * L.cout;
          }
        for (int co = co0; co < co1; ++co) {
          float g[3][8];
          for (int dx = 0; dx < 3; ++dx)
            for (int j = 0; j < 8; ++j) g[dx][j] = rows[dx][j] ? rows[dx][j][co] : 0.f;
          if (group == 0) {
            float* d = d43 + ((((size_t)co * nk8 + kc) * 3 + dy) * 6) * 8;
            for (int j = 0; j < 8; ++j) {
              const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
              const float e = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), o = g1 * (1.f / 12.f);
              d[0 * 8 + j] = g0 * 0.25f;
              d[1 * 8 + j] = -((g0 + g2) + g1) * (1.f / 6.f);
              d[2 * 8 + j] = -((g0 + g2) - g1) * (1.f / 6.f);
              d[3 * 8 + j] = e + o;
              d[4 * 8 + j] = e - o;
              d[5 * 8 + j] = g2;
            }
            continue;
          }
          float u[4][8];
          for (int j = 0; j < 8; ++j) {
            const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
            u[0][j] = g0; u[1][j] = ((g0 + g2) + g1) * 0.5f; u[2][j] = ((g0 + g2) - g1) * 0.5f; u[3][j] = g2;
          }
          for (int nu = 0; nu < 4; ++nu) {
            if (group == 1) { memcpy(dw + (((size_t)co * nk8 + kc) * 12 + nu * 3 + dy) * 8, u[nu], sizeof(u[nu])); continue; }
            // the same transformed weights as nearest bf16 hi / mid planes (nu = 2h + j)
            uint16_t* d = dx3 + ((((size_t)co * nk16 + kc / 2) * 3 + dy) * 2 + (nu & 1)) * 64 + (nu >> 1) * 32 + (kc & 1) * 8;
            for (int j = 0; j < 8; ++j) {
              const uint16_t hb = bf16_rne(u[nu][j]);
              d[j] = hb;
              d[16 + j] = bf16_rne(u[nu][j] - bf16_to_float(hb));
            }
          }
        }
      }
  }
}

// Packs layout groups [h->groups_packed, n) from the HWIO tensors (kept on the host) and uploads them.  Work items =
// (layer, 32 output channels), pulled from an atomic counter by up to 32 threads: 137.7 MB of parameters into the
// default group 0 in well under a second on the hosts this runs on (it took 7 s single-threaded for every layout).
int film_ensure_groups_(film_t* h, int n) {
  if (n <= h->groups_packed) return FILM_OK;
  if (n > 4) n = 4;
  for (const LayerPack& L : h->layers)
    if (!h->host_w.count(L.name + "/kernel") || !h->host_w.count(L.name + "/bias")) return fail(h, FILM_ERR_STATE, "missing weight '%s'", L.name.c_str());
  const int64_t from = h->groups_packed ? h->group_end[h->groups_packed - 1] : 0, to = h->group_end[n - 1];
  h->packed_host.resize((size_t)to, 0.f);
  struct Item { const LayerPack* L; int co0, co1; };
  std::vector<Item> items;
  for (const LayerPack& L : h->layers)
    for (int co = 0; co < L.cout; co += 32) items.push_back({&L, co, std::min(L.cout, co + 32)});
  for (int g = h->groups_packed; g < n; ++g) {
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      for (size_t i; (i = next.fetch_add(1)) < items.size();) pack_layer_group(h, *items[i].L, g, items[i].co0, items[i].co1);
    };
    const unsigned nth = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nth; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
  }
  h->groups_packed = n;
  h->packed_floats = to;
  return upload_packed(h, from, to);
}

// layout groups the current options need at once (the planner asks for more when a plan needs them)
static int groups_for_options(const film_t* h) {
  int n = 1;
  if (h->opt_wino == 2) n = std::max(n, 2);
  if (h->opt_wino == 0 || h->opt_halo_all) n = std::max(n, 3);
  if (h->opt_precision) n = 4;
  return n;
}

int film_finalize(film_t* h) {
  if (!h) return FILM_ERR_INVALID;
  // A handle that has already run may hold cached plans whose ops point into layout groups 1..3 (F(2,3), halo, bf16
  // copies pulled in by Planner::need_groups).  A second weight set must reach those regions too, or such a plan
  // would mix the new group-0 layouts with the previous set's copies: re-pack everything that was packed before.
  const int prev = h->groups_packed;
  h->groups_packed = 0;
  h->packed_floats = 0;
  h->packed_host.clear();
  h->finalized = false;
  if (!h->plan_only && prev > 0) {   // replays of the previous weight set may still be in flight on the caller's stream
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
  }
  int rc = film_ensure_groups_(h, std::max(groups_for_options(h), prev));
  if (rc) return rc;
  h->finalized = true;
  return FILM_OK;
}

// ---- the parameter set as ONE flat blob (what ranks exchange): per layer, in layer order, the HWIO kernel then the bias ----
static int64_t flat_floats(const film_t* h) {
  int64_t n = 0;
  for (const LayerPack& L : h->layers) n += (int64_t)L.kh * L.kw * L.cin * L.cout + L.cout;
  return n;
}

int film_packed_size(film_t* h, int64_t* n) {
  if (!h || !n) return FILM_ERR_INVALID;
  *n = flat_floats(h);
  return FILM_OK;
}

int film_export_packed(film_t* h, float* dst, int64_t cap, int mem_kind) {
  if (!h || !dst) return FILM_ERR_INVALID;
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  const int64_t n = flat_floats(h);
  if (cap < n) return fail(h, FILM_ERR_INVALID, "capacity %lld < %lld floats", (long long)cap, (long long)n);
  std::vector<float> tmp;
  float* out = dst;
  if (mem_kind != FILM_MEM_HOST) {
    if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle has no device");
    tmp.resize((size_t)n);
    out = tmp.data();
  }
  int64_t off = 0;
  for (const LayerPack& L : h->layers) {
    const HostTensor& k = h->host_w.at(L.name + "/kernel");
    const HostTensor& bq = h->host_w.at(L.name + "/bias");
    memcpy(out + off, k.data.data(), k.data.size() * sizeof(float)); off += (int64_t)k.data.size();
    memcpy(out + off, bq.data.data(), bq.data.size() * sizeof(float)); off += (int64_t)bq.data.size();
  }
  if (mem_kind != FILM_MEM_HOST) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(dst, tmp.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  }
  return FILM_OK;
}

int film_import_packed(film_t* h, const float* src, int64_t n, int mem_kind) {
  if (!h || !src) return FILM_ERR_INVALID;
  if (n != flat_floats(h)) return fail(h, FILM_ERR_INVALID, "blob has %lld floats, expected %lld", (long long)n, (long long)flat_floats(h));
  std::vector<float> tmp;
  const float* in = src;
  if (mem_kind != FILM_MEM_HOST) {
    if (h->plan_only) return fail(h, FILM_ERR_NO_DEVICE, "plan-only handle has no device");
    tmp.resize((size_t)n);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(tmp.data(), src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    in = tmp.data();
  }
  int64_t off = 0;
  for (const LayerPack& L : h->layers) {
    HostTensor k, bq;
    k.dims = {L.kh, L.kw, L.cin, L.cout};
    k.data.assign(in + off, in + off + (int64_t)L.kh * L.kw * L.cin * L.cout); off += (int64_t)k.data.size();
    bq.dims = {L.cout};
    bq.data.assign(in + off, in + off + L.cout); off += L.cout;
    h->host_w[L.name + "/kernel"] = std::move(k);
    h->host_w[L.name + "/bias"] = std::move(bq);
  }
  return film_finalize(h);
}

// the kernel-layout blob (debug / tests): the packed prefix [0, *n)
int film_export_layouts(film_t* h, float* dst, int64_t cap, int64_t* n) {
  if (!h) return FILM_ERR_INVALID;
  if (!h->finalized) return fail(h, FILM_ERR_STATE, "film_finalize has not been called");
  if (n) *n = h->packed_floats;
  if (!dst) return FILM_OK;
  if (cap < h->packed_floats) return fail(h, FILM_ERR_INVALID, "capacity %lld < %lld floats", (long long)cap, (long long)h->packed_floats);
  memcpy(dst, h->packed_host.data(), (size_t)h->packed_floats * sizeof(float));
  return FILM_OK;
}

int film_set_option(film_t* h, const char* key, int64_t value) {
  if (!h || !key) return FILM_ERR_INVALID;
  if (!strcmp(key, "graph")) h->opt_graph = value != 0;
  else if (!strcmp(key, "profile")) h->opt_profile = value != 0;
  else if (!strcmp(key, "autotune")) h->opt_autotune = value != 0;
  else if (!strcmp(key, "max_batch")) h->opt_max_batch = value > 0 ? (int)value : 0;
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
    if ((value != 0) != (h->opt_fold != 0)) {  // plans carry the op list: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_fold = value != 0;
    }
  }
  else if (!strcmp(key, "winograd")) {
    if (value < 0 || value > 3) return fail(h, FILM_ERR_INVALID, "winograd: 0, 1, 2 or 3");
    if ((int)value != h->opt_wino) {  // plans carry the kernel choice: drop them
      if (!h->plan_only) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
      for (auto& p : h->plans) free_plan(p.get());
      h->plans.clear();
      h->last_plan = nullptr;
      h->opt_wino = (int)value;
    }
  }
  else if (!strcmp(key, "halo_all")) {
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
    HIPCHK(h, hipMemcpyAsync(st, x0, frame_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(st + nf, x1, frame_bytes, hipMemcpyHostToDevice, s));
    d0 = st; d1 = st + nf; dout = st + 2 * nf;
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
  } else if (h->opt_graph) {
    if (!P->graph_exec) {
      // capture on the handle's own stream, replay on whichever stream the caller wants
      HIPCHK(h, hipStreamSynchronize(s));
      // two capture streams: lane 1 (small subtrees, coarse flow levels, the t = 0.5 warps) forks from and joins
      // the main stream; cross-lane ordering = the events found by Planner::analyze_lanes
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
      HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      hipError_t le = hipSuccess;
      if (two_lanes) {
        le = hipEventRecord(event_of(nops), h->stream);
        if (le == hipSuccess) le = hipStreamWaitEvent(h->stream2, event_of(nops), 0);
      }
      long prev_in_lane[2] = {-1, -1};
      // Relay edges.  When lane X waits for the LAST op p of lane Y, the next op q of lane Y additionally waits for an
      // event recorded on X behind that waiter.  By stream order q already follows p, the captured graph has the edge
      // p -> q (checked with hipGraphGetEdges, tools/experiments/capture_edges.hip), and still: with the flow upsample
      // fused into the warps AND v = res + up fused into the heads, the replayed graph ran the level-4 t = 0.5 warps
      // (q) on the PREVIOUS forward's v4 - deterministically, on every shape (ROCm 7.2; tools/dbg_graph_race.py; the
      // eager path and every other fuse setting were clean).  With the extra parent the replay is bit-identical to the
      // eager launches on changing inputs (tests/test_gpu_parity.py::test_graph_replay_on_changing_inputs).
      hipEvent_t relay[2] = {nullptr, nullptr};
      std::vector<hipEvent_t> relay_pool;
      for (size_t i = 0; i < nops && le == hipSuccess; ++i) {
        const OpDesc& op = P->ops[i];
        const int lane = (two_lanes && op.lane == 1) ? 1 : 0;
        hipStream_t ls = lane ? h->stream2 : h->stream;
        bool waited_on_last = false;
        if (two_lanes) {
          for (int d : op.xdeps) {
            le = hipStreamWaitEvent(ls, event_of((size_t)d), 0);
            if (le != hipSuccess) break;
            if ((long)d == prev_in_lane[1 - lane]) waited_on_last = true;
          }
          if (le == hipSuccess && relay[lane]) { le = hipStreamWaitEvent(ls, relay[lane], 0); relay[lane] = nullptr; }
        }
        if (le == hipSuccess) le = launch_op(op, P->arena, h->packed_dev, ls);
        if (le == hipSuccess && two_lanes && op.signal) le = hipEventRecord(event_of(i), ls);
        if (le == hipSuccess && two_lanes && waited_on_last && !relay[1 - lane]) {
          hipEvent_t e = nullptr;
          le = hipEventCreateWithFlags(&e, hipEventDisableTiming);
          if (le != hipSuccess) break;
          relay_pool.push_back(e);
          le = hipEventRecord(e, ls);
          relay[1 - lane] = e;
        }
        prev_in_lane[lane] = (long)i;
      }
      for (hipEvent_t e : relay_pool) P->lane_ev.push_back(e);   // destroyed with the plan
      if (two_lanes && le == hipSuccess) {
        le = hipEventRecord(event_of(nops + 1), h->stream2);
        if (le == hipSuccess) le = hipStreamWaitEvent(h->stream, event_of(nops + 1), 0);
      }
      hipError_t ce = hipStreamEndCapture(h->stream, &P->graph);
      if (le == hipSuccess) le = ev_err;
      if (le != hipSuccess) return fail(h, FILM_ERR_HIP, "kernel launch failed during capture: %s", hipGetErrorString(le));
      HIPCHK(h, ce);
      HIPCHK(h, hipGraphInstantiate(&P->graph_exec, P->graph, nullptr, nullptr, 0));
    }
    HIPCHK(h, hipGraphLaunch(P->graph_exec, s));
  } else {
    for (const OpDesc& op : P->ops) HIPCHK(h, launch_op(op, P->arena, h->packed_dev, s));
  }
  h->last_plan = P;
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
  HIPCHK(h, hipMemcpy(dst, P->arena + b.off, (size_t)b.size() * sizeof(float), hipMemcpyDeviceToHost));
  return FILM_OK;
}

}  // extern "C"
