/*
 * film_hip.h  --  C-ABI of libfilm_hip.so, the MI355X (gfx950) FILM inference engine.
 *
 * This is the drop-in boundary for the reference's one operator call on the hot path:
 *
 *     self._model = tf.compat.v2.saved_model.load(model_path)        eval/interpolator.py:148
 *     result = self._model({'x0','x1','time'}, training=False)       eval/interpolator.py:170-171
 *     image  = result['image']                                       eval/interpolator.py:172
 *
 * i.e. "load the film_net weights" and "run models/film_net (interpolator.py:89-207) on a
 * batch of frame pairs and return the t=0.5 frame".  Everything here is plain C: pointers
 * and sizes, no torch / numpy types.  The Python host
 * (frame-interpolation_amd/eval/interpolator.py) binds it with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Tensors are NHWC float32, weights are TF HWIO [kh,kw,cin,cout] float32 (the layout a
 * Keras SavedModel stores, training/train_lib.py:280).
 *
 * Every function returns 0 on success or a negative FILM_ERR_* code; the message is
 * available from film_last_error().  A handle owns one device, one stream, the packed
 * weights and a per-(B,H,W) cached execution plan + workspace.  Handles are not
 * thread-safe - ONE in-flight issuer per handle: a forward launches on the caller's stream and on
 * the handle's side stream with one set of events per plan, so two threads (or two streams) must
 * not issue forwards of the same handle concurrently.  Several handles may coexist, also on
 * several host threads (tests/test_host_threads_cpu.py runs that under ThreadSanitizer).
 */
#ifndef FILM_HIP_H_
#define FILM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FILM_OK 0
#define FILM_ERR_INVALID (-1)    /* bad argument / shape (reference: assert / ValueError)      */
#define FILM_ERR_STATE (-2)      /* call order (e.g. forward before finalize)                   */
#define FILM_ERR_NO_DEVICE (-3)  /* no HIP device: the engine has NO CPU fallback               */
#define FILM_ERR_HIP (-4)        /* a HIP runtime call failed                                   */
#define FILM_ERR_NOMEM (-5)
#define FILM_ERR_NOTFOUND (-6)   /* unknown weight / tap name                                   */

#define FILM_MEM_HOST 0   /* x0/x1/out are host pointers: H2D + D2H done by the call          */
#define FILM_MEM_DEVICE 1 /* x0/x1/out are device pointers on the handle's device              */

#define FILM_MAX_SPECIALIZED 7

typedef struct film_handle film_t;

/* models/film_net/options.py:20-80 ; defaults = training/config/film_net-L1.gin:17-23 */
typedef struct film_config {
  int32_t pyramid_levels;
  int32_t fusion_pyramid_levels;
  int32_t specialized_levels;
  int32_t sub_levels;
  int32_t filters;
  int32_t flow_convs[FILM_MAX_SPECIALIZED + 1];   /* specialized_levels+1 entries used */
  int32_t flow_filters[FILM_MAX_SPECIALIZED + 1]; /* specialized_levels+1 entries used */
} film_config;

/* Fills *cfg with the published architecture (film_net-L1.gin:17-23). */
int film_default_config(film_config* cfg);

/* Creates an engine on HIP device `device` (>= 0).
 * device == -1 creates a PLAN-ONLY handle: it can take weights, pack them and describe
 * plans (film_plan_json / film_export_packed) without a GPU, and every compute entry point
 * returns FILM_ERR_NO_DEVICE.  There is no CPU execution path in this library.
 * Replaces: tf.compat.v2.saved_model.load (eval/interpolator.py:148), object creation half. */
int film_create(film_t** out, int device, const film_config* cfg);
void film_destroy(film_t* h);

/* Message of the last failing call on this handle (or of film_create when h == NULL). */
const char* film_last_error(const film_t* h);

/* Supplies one tensor by canonical name, e.g.
 *   "feat_net/sub_extractor/cfeat_conv_0/kernel"  dims = {3,3,3,64}
 *   "predict_flow/flow_predictor_shared/conv_4/bias"  dims = {2}
 * (names: frame-interpolation_amd/film_hip/weights.py).  Data is copied.
 * Replaces: the variable restore inside saved_model.load (eval/interpolator.py:148). */
int film_set_weight(film_t* h, const char* name, const float* data, const int64_t* dims, int ndim);

/* Checks that every tensor of the architecture is present with the right shape, repacks
 * HWIO into the engine's K-major layout (zero-padding / permuting concat segments) and
 * uploads the blob to the device. */
int film_finalize(film_t* h);

/* The parameter set as ONE flat blob of floats - per layer, in layer order, the HWIO kernel then the bias (137.7 MB for the
 * published net): size, export to / import from a caller buffer.  This is what ranks exchange: rank 0 reads the model,
 * the others receive the blob by RCCL broadcast and film_import_packed() it (= film_set_weight of every tensor +
 * film_finalize).  Kernel layouts are built per handle, on demand: film_finalize packs what the default fp32 plan reads
 * (K-major + F(4,3) + phase-summed 2x2 copies, 3.0x the parameters, multi-threaded, well under a second); the F(2,3), halo
 * and bf16-split copies are packed when an option or a plan first needs them. */
int film_packed_size(film_t* h, int64_t* n_floats);
int film_export_packed(film_t* h, float* dst, int64_t capacity_floats, int mem_kind);
int film_import_packed(film_t* h, const float* src, int64_t n_floats, int mem_kind);

/* The weight broadcast itself, for hosts without torch.distributed (north_star: "RCCL broadcast of weights over xGMI"; the
 * reference has nothing to replace here - every beam worker calls tf.saved_model.load on its own, eval/interpolator_cli.py:127-150).
 * Collective over `nccl_comm` (an ncclComm_t of the CALLER - the library creates no communicator and links no RCCL: ncclBroadcast is
 * resolved at run time from the RCCL already in the process, else from librccl.so): rank `root` must hold a finalized weight set and sends
 * the flat parameter blob (film_packed_size floats, staged in a device buffer of handle `h`'s GPU), every other rank receives it and
 * film_import_packed()s it (= film_finalize with the received set).  `rank` = the caller's rank in the communicator; `stream` = the HIP
 * stream the broadcast is enqueued on (NULL: the handle's own), synchronised before the function returns.  Every rank must call it. */
int film_bcast_weights(film_t* h, void* nccl_comm, int root, int rank, void* stream);

/* Debug / tests: the kernel-layout blob packed so far (offsets as in film_plan_json; "pack_groups" option packs more).
 * dst == NULL: size query. */
int film_export_layouts(film_t* h, float* dst, int64_t capacity_floats, int64_t* n_floats);

/* Runs film_net on B frame pairs: x0, x1 [B,H,W,3] -> out [B,H,W,3] (un-clipped), t = 0.5.
 * H and W must be divisible by 2^(pyramid_levels-1) (options.py:36-37) - pad first, as
 * Interpolator.interpolate does (eval/interpolator.py:166-168).
 * `stream`: a hipStream_t to run on.  NULL means: with FILM_MEM_HOST the handle's own stream (the call
 * synchronises before returning); with FILM_MEM_DEVICE the NULL (legacy default) stream, so the work
 * is ordered with the caller's default-stream work (PyTorch's default stream is the NULL stream) and
 * the call is asynchronous.
 * Replaces: self._model(inputs, training=False)['image'] (eval/interpolator.py:170-172). */
int film_forward(film_t* h, const float* x0, const float* x1, int B, int H, int W, float* out,
                 int mem_kind, void* stream);

/* Interpolator.__call__ on the device (eval/interpolator.py:178-209), frames x0, x1 [B,H,W,3] -> out [B,H,W,3]:
 *   block_h * block_w <= 1 : zero-pad H, W up to multiples of `align` (offset pad//2, _pad_to_align :30-63), run the
 *                            model, crop (Interpolator.interpolate :152-176);
 *   block_h * block_w  > 1 : split every frame into block_h x block_w row-major patches (image_to_patches
 *                            :66-99, same divisibility asserts), pad EACH patch to `align`, run all patches as one
 *                            batch (the reference loops over them with B = 1, :199-206), crop each, stitch
 *                            (patches_to_image :102-126).  The reference's tiled path takes one frame pair; B > 1
 *                            here tiles every pair of the batch the same way (used by the breadth-first
 *                            recursion driver, film_hip/recursive.py).
 * align <= 0 means no padding (the reference's align=None); H, W (or the patch) must then be divisible by
 * 2^(pyramid_levels-1).  Pad / patch / crop / stitch are two HIP kernels that read the caller's frames and write
 * the plan's input buffer directly (and back), so with FILM_MEM_DEVICE nothing but the frames themselves is copied.
 * Batches are processed in chunks of tiles that keep one invocation's workspace below 64 GiB (and below 60 % of the HBM the
 * handle can get at that moment) and every buffer read through a 32-bit whole-buffer offset below 4 GiB; the chunk is the
 * largest divisor of the tile count inside that bound where one exists, so every invocation replays one cached plan
 * (16 * 2^k tiles of a 4K recursion -> chunks of 8), and it is halved when a workspace allocation fails (results unchanged).
 * `stream` and mem_kind as for film_forward. */
int film_interpolate(film_t* h, const float* x0, const float* x1, int B, int H, int W, int align, int block_h,
                     int block_w, float* out, int mem_kind, void* stream);

/* Execution options.  Keys:
 *   "autotune" 0/1 time every distinct conv shape of a new plan with each fitting tile shape and keep
 *                  the fastest (default 1; cannot change results - same k-ordered fma chain per output)
 *   "graph"   0/1/2  how a plan's ops reach the device.  2 (default): direct launches on two lanes - lane 0 on the caller's
 *                  stream, lane 1 (small subtrees, coarse flow levels, the t = 0.5 warps) on a side stream of the handle, ordered by
 *                  events; 1: the same two lanes captured once per (B, H, W) into a hipGraph and replayed; 0: one stream, plan order.
 *                  Identical results.  The graph is opt-in since round 5: the HIP 7.0 runtime PyTorch 2.10 bundles can crash in the
 *                  first launch of a fresh multi-branch graph late in a long-lived process (profiles/r05_hipgraph_first_launch_crash.md)
 *   "profile" 0/1  record a hipEvent pair around every kernel of the next forwards
 *                  (forces graph off); read the result with film_profile_json
 *   (options marked [extra] exist only in a library built with FILM_EXTRA_FAMILIES=1 - libfilm_hip_extra.so, see film_version();
 *   the default library, which holds only the kernel families a default plan can select, refuses them with a message)
 *   "precision" p  [extra for 1, 2]  0 (default): every convolution on the exact fp32 MFMA.  1: "bf16x6" - the large 3x3
 *                  convolutions split each fp32 operand exactly into three bf16 pieces and accumulate the six
 *                  partial products >= 2^-16 in fp32 on the bf16 matrix pipe (dropped terms < 2^-23 relative);
 *                  about 1.5x faster, results differ from mode 0 at the level of a changed summation order.
 *                  2: "bf16x3" - the same kernels with a nearest 2-way split (hi + mid, |x - hi - mid| <= 2^-17 |x|)
 *                  and the three products hi*hi + hi*mid + mid*hi: per-product error <= 2^-15 relative in the worst
 *                  case, 4.4e-6 rms, zero mean (not an fp32 result; 7e-6 from the oracle end to end against the 1e-3
 *                  image bound, see tests).
 *                  Changing it drops the cached plans.
 *   "lanes"   0/1/2  >= 1 (default 1): replay graphs run the independent small / HBM-bound kernels (feature subtrees of the coarse
 *                  pyramid levels, coarse flow levels, the t = 0.5 warps) on a second stream beside the main chain;
 *                  2 (measured 2 % slower at 1080p, not the default): on frames larger than 512 x 512 the coarse decoder levels (>= 2) join that stream right behind
 *                  the aligned levels they read, so that their matrix-bound convolutions run beside the flow estimator's
 *                  chain of HBM-bound warps and short launches instead of behind it.  Ordering between the two streams
 *                  comes from a buffer-overlap analysis of the plan.  3 = the same on frames of any size (tests).  Drops the
 *                  cached plans.
 *   "splitk"  0/1  1 (default): the deep convolutions of pyramid levels with <= 4096 pixels per image run split-K
 *                  (2-16 partial sums over K ranges, added in split order by a second kernel: deterministic, and the
 *                  factor depends on the level size and the layer only, never on the batch); it is what bounds the
 *                  latency of small frames.  Changing it drops the cached plans.
 *   "pack_groups" n  pack (and upload) the weight layout groups 1..n now: 1 default fp32 layouts, 2 + F(2,3) copy, 3 + halo
 *                  copy, 4 + bf16 split copies (normally packed on demand)
 *   "fuse"    bits 31 (default): small-launch fusion, identical arithmetic and bit-identical results.  1: tf.image.resize(2 * v)
 *                  of the flow estimator inside the warp kernels that consume it; 2: v = residual + upsampled flow inside
 *                  the flow-head kernels; 4: the warped images and the half flows of the t = 0.5 stage (the sixteen miscellaneous
 *                  channels of an aligned level) inside the second feature warp of the level; 8: AveragePooling2D of the sub-extractor
 *                  stages in the epilogue of the F(4,3) convolution in front of it (about 35 launches fewer per forward in all);
 *                  16: the RGB head (1x1 convolution, fusion.py:138-140) in the epilogue of the last decoder layer, whose
 *                  64-channel output is then never written.  0: one launch per reference op.  Drops the cached plans.
 *   "fold2x2" 0-2  the decoder's nearest-x2 upsample + 2x2 convolution (fusion.py:133-135) on the LOW-resolution input.
 *                  1 (default): in its difference form - four products per low-resolution pixel (with I, Dx = I - I(x+1),
 *                  Dy = I - I(y+1), Dxy and the weight sums S, Sx, Sy, W11: out(2y+py, 2x+px) = S.I - px Sx.Dx - py Sy.Dy +
 *                  py px W11.Dxy), conv_fold4_kernel; 2: as four sub-pixel phase convolutions with pre-summed weights (9 taps
 *                  per 4 outputs instead of 16; the general kernel).  Both are exact regroupings of the sum, the rounding
 *                  differs at the 1e-7 level.  0: one 2x2 convolution with the upsample folded into its gather.
 *                  Drops the cached plans.
 *   "planar" 0/1   1 (default): an aligned-pyramid level (interpolator.py:167-183) is stored as three pixel-major planes -
 *                  warp(features of image 0), warp(features of image 1), the sixteen image / flow channels - each written
 *                  contiguously by its warp, and read by the decoder as three input segments in the reference's channel
 *                  order (same weights, same sums, same bits); film_get_tap("aligned<l>") interleaves them.  0: one
 *                  [N][H][W][2C + 16] buffer per level.  Drops the cached plans.
 *   "winograd" w   1 (default): the large 3x3 convolutions use a 1-D Winograd transform along x - F(4,3) (2x fewer
 *                  fp32 multiplies, 128-pixel patches) on the levels whose width fills its patches, F(2,3) (1.5x fewer)
 *                  elsewhere; fp32 throughout, the rounding differs from the direct sum at the 1e-6 (F(2,3)) /
 *                  5e-6 (F(4,3)) level.  0: direct kernels only.  2 [extra] / 3: F(2,3) / F(4,3) on every eligible 3x3
 *                  convolution (tests).  Changing it drops the cached plans.
 *   "halo_all" 0/1 [extra] run every eligible 3x3 convolution on the halo-staged kernels whatever its size (default 0:
 *                  only where measured faster); "tune_ms" n: autotune spends at least n ms per candidate.
 *                  Test / tuning knobs; "halo_all" drops the cached plans.
 *   "wino2d"  0/1/2  1 (default): every 3x3 convolution whose channels come in sixteens, on levels of >= "w2d_min_px" (1536) pixels per image
 *                  - and on smaller ones down to "w2d_small_px" (256) pixels that fill >= 65 % of their 8-row x 32-pixel tiles - runs the nested
 *                  Winograd form F(4,3) along x times F(2,3) along y (conv_wino2d_kernel: 3 multiplies per output where the 1-D
 *                  F(4,3) kernel spends 4.5 and the direct convolution 9; fp32 throughout, rounding at the level of the 1-D form).
 *                  0: never.  2: every layer that has the weight copy, on every level (tests).  Drops the cached plans.
 *   "w2d_min_px" n / "w2d_small_px" n  the two level-size thresholds of that rule (A/B runs; a function of the level only - never the batch).
 *   "w2d_splitk" 0/1  1 (default): the nested kernel's K >= 768 layers on levels of <= 4096 pixels per image (the 36x60 level of a 1080p
 *                  tile, the 64x64 level of a 256x256 pair) run as up to four K ranges + the ordered reduction, like "splitk"
 *                  (which also switches it off); factor from the level size and the layer only.  Drops the cached plans.
 *   "w43_shape" n  test knob: every convolution on conv_wino43_kernel that can run tile shape n (Wino43Tile, film_kernels.h)
 *                  does, instead of the autotuned shape; -1 (default) = autotuned.  Results cannot change.  Drops the cached plans.
 *   "w2d_shape" n  the same for conv_wino2d_kernel (Wino2dTile: 0..2 = 8 rows x 32 pixels, 3..5 = 16 x 16 pixels - the latter only on levels it pads no more).
 *   "fold4_shape" n  the same for conv_fold4_kernel (Fold4Tile).
 *   "max_batch" n  process at most n frame pairs / tiles per model invocation (0 = only the built-in limits: 64 GiB of
 *                  workspace, 4 GiB per buffer read through a whole-buffer 32-bit offset); frame pairs are independent,
 *                  results do not change
 *   "host_overlap" 0/1  1 (default): film_interpolate with FILM_MEM_HOST pipelines its copies with the work - the second frame is uploaded
 *                  while the first layers run on the first frame's tiles, the upper half of the result is downloaded while the last layer
 *                  computes the lower half (one frame, an even number of block rows); 0: upload, work, download.  Same bits. */
int film_set_option(film_t* h, const char* key, int64_t value);

/* Autotune choices across processes.  film_export_tune writes text (buf / capacity / needed as film_plan_json): a header
 * line with the library version, then "<conv shape signature>\t<tile id>" per shape this handle measured or imported.
 * film_import_tune takes such text before the first forward: shapes found in it are not measured again (first call of a
 * 1080p plan 2.5 s -> the plan build + graph capture alone).  Text from another library version is ignored, entries are
 * validated against the kernel family of the op that wants them, and results never depend on the cache (every tile of a
 * family produces the same bits).  film_hip.engine reads / writes the file named by $FILM_TUNE_CACHE with these. */
int film_export_tune(film_t* h, char* buf, int64_t capacity, int64_t* needed);
int film_import_tune(film_t* h, const char* text);

/* Per-kernel-class timing of the last profiled forward as JSON
 * {"classes": {"conv_mfma": {"launches": n, "ms": t, "flops": f, "bytes": b}, ...}}. */
int film_profile_json(film_t* h, char* buf, int64_t capacity, int64_t* needed);

/* Description of the plan for (B,H,W) as JSON: named workspace buffers (offset, dims),
 * the op list with every kernel parameter and the packed-weight offsets.  Works on
 * plan-only handles; tests interpret it against a numpy arena to validate planner and
 * weight packing without a GPU.  "offset32_buffer_bytes" = the largest buffer a kernel with whole-buffer 32-bit
 * offsets reads (must stay below 4 GiB; conv_wino43_kernel and the pointer-addressed kernels are not limited). */
int film_plan_json(film_t* h, int B, int H, int W, char* buf, int64_t capacity, int64_t* needed);

/* Copies a named workspace buffer of the last forward (see "buffers" in film_plan_json) to a
 * host array; dims receives {N,H,W,C}.  Debug / parity taps, mirrors the aux outputs of
 * models/film_net/interpolator.py:191-199. */
int film_get_tap(film_t* h, const char* name, float* dst, int64_t capacity_floats, int64_t dims[4]);

/* write_image's quantisation on the device (replaces the host loop of the reference's eval/util.py:44-59 write_image,
 * lines 51-52): dst[i] = uint8(clip(src[i] * 255, 0, 255) + 0.5), the same float32 operations in the same order = the same bytes.
 * src (float32) and dst (uint8) are DEVICE pointers to n values; asynchronous on `stream` (NULL: the default stream) of the
 * current device.  The frames of a recursion then cross PCIe as 1 byte per value instead of 4.  Returns FILM_OK or a negative error. */
int film_to_uint8(const float* src, unsigned char* dst, int64_t n, void* stream);

/* The variable-restore half of `tf.compat.v2.saved_model.load(model_path)` (reference eval/interpolator.py:148): reads the
 * checkpoint of a Keras SavedModel - `<path>/variables/variables.index` + `.data-0000N-of-0000M`, written by model.save()
 * (training/train_lib.py:280, training/build_saved_model_cli.py:65-73) - without TensorFlow and without Python, places every
 * film_net tensor (film_set_weight) and finalizes the handle (film_finalize).  `path`: the SavedModel directory, or a bundle
 * prefix (".../variables").  Tensors are found by their Keras object-graph attribute paths (extract_sublevels/convs/i,
 * _predictors/p/_convs/j, convs/i/j, output_conv - the names in feature_extractor.py:118-123,160, pyramid_flow_estimator.py:
 * 74-83,111-123, fusion.py:64-101), independent of the layer_with_weights-N numbering; a tensor not found that way is taken
 * by shape ONLY if its shape is unique among the unplaced tensors and the unused variables - anything ambiguous is an error,
 * never a guess.  verify_crc != 0 checks the masked crc32c of every index block and tensor.  report (may be NULL) receives
 * one line per tensor, "<name>\t<path|shape>\t<checkpoint key>\n" (size query: capacity 0, *needed = bytes incl. NUL).
 * Errors: FILM_ERR_NOTFOUND (no bundle at `path`, a tensor missing), FILM_ERR_INVALID (corrupt / unsupported file, crc). */
int film_load_bundle(film_t* h, const char* path, int verify_crc, char* report, int64_t report_capacity, int64_t* report_needed);

/* CRC-32C (Castagnoli) continued from `crc` (0 to start) over n bytes.  Host helper of the TF-free SavedModel
 * variables reader (frame-interpolation_amd/film_hip/tf_bundle.py), which checks the masked crc32c TensorFlow
 * stores per tensor and per index block; part of replacing tf.saved_model.load (eval/interpolator.py:148). */
uint32_t film_crc32c(uint32_t crc, const void* data, int64_t n);

/* Library build info: "gfx950;film_hip r<round>;src=<12 hex digits of the sha1 over csrc/ and this header>[+extra]" ("+extra": built
 * with FILM_EXTRA_FAMILIES=1, i.e. with the opt-in kernel families).  Tune caches, bench lines and PMC summaries carry it. */
const char* film_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FILM_HIP_H_ */
