// Host-only builds of libfilm_hip (tools/sanitize/build_host.sh): the kernel launchers live in the .hip translation units, which g++ cannot
// compile.  A plan-only handle (device = -1) never reaches them; these definitions make the library link and abort if one is called.
#include <cstdio>
#include <cstdlib>

#include "../../frame-interpolation_amd/csrc/film_kernels.h"

#define STUB(sig) sig { fprintf(stderr, "kernel launcher reached in a host-only build: %s\n", __func__); abort(); }
STUB(hipError_t film_launch_conv(const ConvParams&, int, hipStream_t))
STUB(hipError_t film_launch_conv_pw(const ConvPwParams&, hipStream_t))
STUB(hipError_t film_launch_flow_head(const FlowHeadParams&, hipStream_t))
STUB(hipError_t film_launch_pool(const PoolParams&, hipStream_t))
STUB(hipError_t film_launch_flow_up(const FlowUpParams&, hipStream_t))
STUB(hipError_t film_launch_flow_add(const FlowAddParams&, hipStream_t))
STUB(hipError_t film_launch_warp(const WarpParams&, hipStream_t))
STUB(hipError_t film_launch_pack_flow(const PackFlowParams&, hipStream_t))
STUB(hipError_t film_launch_frame_to_tiles(const TileMapParams&, hipStream_t))
STUB(hipError_t film_launch_tiles_to_frame(const TileMapParams&, hipStream_t))
STUB(hipError_t film_launch_to_uint8(const float*, uint8_t*, int64_t, hipStream_t))
STUB(hipError_t film_launch_fill_random(float*, int64_t, uint32_t, hipStream_t))
