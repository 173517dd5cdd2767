// film_internal.h -- what the translation units of libfilm_hip.so share: the plan / op / layer / handle structures and the
// internal functions that cross file boundaries.  Nothing here is part of the C-ABI (include/film_hip.h).
//
//   film_engine.cpp   C-ABI entry points, options, executor (launch, autotune, hipGraph capture, profiling), chunking
//   film_planner.cpp  Planner: the graph of models/film_net/interpolator.py:89-207 as an op list over one workspace arena,
//                     kernel-family decisions, the two-lane dependency analysis, film_plan_json's text
//   film_layers.cpp   the layer table (weight names, shapes, channel permutations) and the kernel-layout packer
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/film_hip.h"
#include "film_kernels.h"

namespace film_internal {


extern thread_local std::string g_create_error;   // message of a failed film_create (film_engine.cpp)

enum OpKind { OP_CONV = 0, OP_FLOW_HEAD, OP_CONV_PW, OP_POOL, OP_FLOW_UP, OP_FLOW_ADD, OP_WARP, OP_PACK_FLOW, OP_KINDS };
extern const char* const kKindName[OP_KINDS];

struct Buffer {
  std::string name;
  int64_t off;  // floats from arena base
  int N, H, W, C;      // all 0 for scratch regions (reinterpreted per use)
  int64_t floats;      // extent
  // > 0: the buffer is stored as three pixel-major planes, [N][H][W][planar] x 2 then [N][H][W][C - 2 planar], instead of
  // [N][H][W][C] (the aligned pyramid levels: each plane is written contiguously by one warp launch; film_get_tap interleaves)
  int planar = 0;
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
  int64_t wf4_off = -1;               // conv: the difference-form planes S, Sx, Sy, W11 of a folded 2x2 layer (conv_fold4_kernel; = w_off when fold == 3)
  int wino = 0;                       // conv: 1 = runs on conv_wino_kernel, 2 = on conv_winox3_kernel (precision bf16x3), 3 = conv_wino43_kernel,
                                      //       4 = conv_wino2d_kernel (nested F(4,3) x F(2,3))
  int split = 0;                      // conv: runs on conv_halo_split_kernel (precision mode bf16x6)
  int lane = 0;                       // graph replay: 0 = main stream, 1 = side stream (small / HBM-bound work)
  std::vector<int> xdeps;             // ops on the OTHER lane this op must wait for (from the buffer overlap analysis)
  bool signal = false;                // some op on the other lane waits for this one
  int halo = 0;                       // conv: runs on conv_halo_kernel (decided by shape, see Planner::conv)
  // generic views
  View in, in2, out;
  // warp: the fused sixteen miscellaneous channels of an aligned level (t = 0.5 stage): img_in = both images [2 NB][H][W][3],
  // pack_b / pack_f = backward / forward flow, img_out = [warp(img0) 3 | warp(img1) 3 | 0.5 bflow 2 | 0.5 fflow 2 | 0 x 6]
  View pack_b, pack_f;
  View img_in, img_out;
  View pw_out; int pw_cout = 0;   // conv: fused 1x1 convolution behind it (weights w2_off / b2_off) writes pw_out; `out` is not written then
  View in3, out2;   // warp: coarser flow to upsample / the upsampled flow it stores; flow heads: in2 = upsampled flow, out2 = v = out + in2
  int NB = 0, H = 0, W = 0;  // conv/warp: output dims; pool: input dims; flow_up: input dims
  float fscale = 1.f;
  int src_brot = 0, flow_brot = 0, misc_nb = 0;   // warp: one launch for both directions / images of a level (WarpParams)
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
  int64_t wf4_off = -1;      // 2x2 layers behind a nearest upsample: the four planes of the difference form (conv_fold4_impl.h) S = ((W00 + W01) + W10) +
                             //     W11, Sx = W01 + W11, Sy = W10 + W11, W11 as [Cout/32][ctot/8][plane 4][K half][32][4]: 4*ctot*cout in all
  int64_t w43_off = -1;      // ... the F(4,3)-along-x transformed copy for conv_wino43_kernel, [Cout][ctot/8][3 dy][6 nu][8]
  int64_t w2d_off = -1;      // has_w2d layers: the nested F(4,3)x x F(2,3)y copy for conv_wino2d_kernel,
                             //     [Cout/32][ctot/8][mu 4][nu 6][K half][32][4] (24 values per (ci, co): 2.67x the kernel)
  int64_t wx_off = -1;       // ... and the transformed copy split into bf16 hi / mid for conv_winox3_kernel,
                             //     [Cout][ctot/16][dy][j][h][plane][16] bf16 (nu = 2h + j)
  int64_t ws_off = -1;       // ... and the bf16x6 copy for conv_halo_split_kernel, [Cout][ctot/16][9][3][16] bf16
                             //     (offset in floats; 1.5 floats per weight)
  bool has_halo() const { return kmajor() && kh == 3 && kw == 3; }
  bool has_fold() const { return kmajor() && kh == 2 && kw == 2; }
  // conv_wino2d_kernel against the best 1-D F(4,3) tile of the same run (tools/w2d_bench.hip, profiles/r04_w2d_vs_w43.log): 16-30 %
  // faster on EVERY 3x3 layer of the 1080p plan, K = 32 ... 2448 (round 3's kernel lost below K = 208: its four-round epilogue of dword
  // stores cost 26 000 cycles per workgroup) - every 3x3 layer whose channels come in sixteens and thirty-twos carries the copy
  bool has_w2d() const { return has_halo() && ctot() % 16 == 0 && cout % 32 == 0; }
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


}  // namespace film_internal

struct film_handle {
  void* stage = nullptr;       // device staging of whole frames for film_interpolate(FILM_MEM_HOST)
  size_t stage_bytes = 0;
  int device = -1;
  bool plan_only = true;
  film_config cfg{};
  hipStream_t stream = nullptr;
  std::string err;
  std::map<std::string, film_internal::HostTensor> host_w;
  std::vector<film_internal::LayerPack> layers;
  std::map<std::string, int> layer_idx;
  int64_t packed_floats = 0;          // floats of the PACKED PREFIX (groups [0, groups_packed)); group_end[3] = all layouts
  int64_t group_end[4] = {0, 0, 0, 0};  // end offset of layout group g (see film_create): 0 base, 1 F(2,3), 2 halo, 3 bf16 splits
  int groups_packed = 0;
  std::vector<float> packed_host;
  float* packed_dev = nullptr;
  bool finalized = false;
  std::vector<std::unique_ptr<film_internal::Plan>> plans;
  film_internal::Plan* last_plan = nullptr;
  uint64_t tick = 0;
  // How a plan is executed (option "graph"): 2 (default) = its ops are launched directly on two lanes - lane 0 on the caller's stream,
  // lane 1 on the handle's side stream, ordered by the events of Planner::analyze_lanes; 1 = the same two lanes captured once into a
  // hipGraph and replayed; 0 = one stream, in plan order (the reference the tests compare the other two with: bit-identical).
  // Why the graph is not the default (round 5): the HIP runtime PyTorch 2.10+rocm7.0 bundles (7.0.51831) crashes in the FIRST
  // hipGraphLaunch of a freshly instantiated multi-branch graph when the process has created and destroyed enough streams before -
  // a null-ish dereference in the function that assigns the exec's parallel streams (reads past its own stream vector; backtrace and
  // disassembly: profiles/r05_hipgraph_first_launch_crash.md).  Reproduced by the GPU test-suite in file order (8 engines, 23 graphs
  // into the process); not reproducible in a short process.  A linear graph (lanes = 0) takes the runtime's single-stream path and
  // is safe but serial (+2 ms per 1080p step); direct launches cost ~170 runtime calls per forward on the host, asynchronously.
  int opt_graph = 2;
  int opt_profile = 0, opt_autotune = 1;
  int opt_max_batch = 0;  // 0: only the 4 GiB-per-buffer limit
  int opt_host_overlap = 1;   // film_interpolate(FILM_MEM_HOST): 1 = the second frame's upload behind the first frame's first layers, the first half of the
                              // result downloaded behind the second half's last layer (film_engine.cpp, "host pipeline"); 0 = copies, then work, then copy
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};   // its two events (second frame in place / first half stitched), lazily created
  int opt_splitk = 1;     // 1: split-K (ksplit partial sums + ordered reduction) for the deep layers of levels with <= 1024 pixels
  int opt_fuse = 31;       // 1: flow_up fused into the flow-estimator warps, v = res + up into the flow heads (same arithmetic, 12 launches fewer)
  int opt_planar = 1;     // 1: aligned-pyramid levels as three planes (feat0 | feat1 | misc16), each written contiguously by its warp
  int opt_fold = 1;       // 1: nearest-upsample + 2x2 conv as four sub-pixel phase convolutions (9 taps per 4 outputs)
  int opt_fold4 = 1;      // (with opt_fold) 1: ... in the difference form on conv_fold4_kernel (4 multiplies per low-resolution pixel instead of 9)
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
  int opt_w2d_min_px = 1536; // conv_wino2d_kernel runs the 3x3 layers of levels with at least this many pixels per image (planner rule;
                             // profiles/r04_w2d_min_px_ab.log: 8 or more 8x32 patches per image - 32x56 yes, 32x32 no)
  int opt_w2d_small_px = 256; // ... and of smaller levels down to this many pixels when the level fills >= 65 % of its 8x32 tiles (0: never)
  int opt_w2d_splitk = 1; // 1: split-K for the nested kernel's K >= 768 layers on levels of <= 4096 pixels, S = min(4, K / 384) (planner rule; option "w2d_splitk" for A/B runs)
  int opt_w2d_shape = -1; // tests: >= 0 = every conv_wino2d_kernel op that can run this Wino2dTile shape does
  int opt_fold4_shape = -1; // tests: >= 0 = every conv_fold4_kernel op that can run this Fold4Tile shape does
  int opt_w43_shape = -1; // tests: >= 0 = every conv_wino43_kernel op that can run this Wino43Tile shape does (instead of the autotuned one)
  std::string profile_json;
  std::map<std::string, int> tune_cache;  // conv shape signature -> fastest tile
  std::map<std::string, int> tune_import; // choices of an earlier process (film_import_tune): taken, if still a candidate of
                                          // the op's kernel family, instead of timing the candidates again
};

extern "C" int film_ensure_groups_(film_t* h, int n);   // packs + uploads weight layout groups [groups_packed, n) on demand (internal)

namespace film_internal {

int fail(film_t* h, int code, const char* fmt, ...);    // stores the message on the handle (or for film_create), returns `code`

#define HIPCHK(h, expr)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(h, FILM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)


// ---- film_layers.cpp: architecture helpers (mirror frame-interpolation_amd/film_hip/weights.py) and the layer table
std::vector<int> feature_channels(const film_config& c);   // feature_extractor.py:186-193
int slot_offset(const film_config& c, int j);              // channel offset of sub-pyramid stage j in a feature level
std::vector<int> fusion_filters(const film_config& c);     // fusion.py:75-79
std::string predictor_prefix(const film_config& c, int level);   // pyramid_flow_estimator.py:109-123
int predictor_index(const film_config& c, int level);
int validate_config(film_t* h, const film_config& c);
void build_layers(film_t* h);

// ---- film_planner.cpp
int plan_build(film_t* h, Plan* P, int B, int H, int W);   // fills P->bufs / ops / arena_floats for (B, H, W)
int64_t limited_buffer_bytes(const Plan* P);                // largest buffer a kernel with whole-buffer 32-bit offsets reads
std::string plan_json(film_t* h, const Plan& P);

}  // namespace film_internal
