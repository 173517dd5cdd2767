// capture_edges.hip -- which edges does hipStreamBeginCapture record when one captured stream's events are waited for by
// another?  (Development tool; found while debugging stale t = 0.5 warps under the two-lane graph, see film_engine.cpp.)
//   hipcc --offload-arch=gfx950 tools/experiments/capture_edges.hip -o tools/bin/capture_edges && tools/bin/capture_edges
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k(int* p, int v) { p[0] = v; }
static int run(int variant) {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int* d;
  CK(hipMalloc(&d, 64 * sizeof(int)));
  hipEvent_t fork, join, e1, e2;
  for (hipEvent_t* e : {&fork, &join, &e1, &e2}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, s1));
  CK(hipStreamWaitEvent(s2, fork, 0));
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s2, d + 1, 1);   // K1 on s2
  CK(hipEventRecord(e1, s2));
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s2, d + 2, 2);   // K2 on s2
  CK(hipEventRecord(e2, s2));
  if (variant == 1 || variant == 2) { CK(hipStreamWaitEvent(s1, e2, 0)); hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s1, d + 10, 10); }   // KA waits K2
  if (variant == 2 || variant == 3) { CK(hipStreamWaitEvent(s1, e1, 0)); hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s1, d + 11, 11); }   // KB waits K1 (older)
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s2, d + 3, 3);   // K3 on s2: must depend on K2
  CK(hipEventRecord(join, s2));
  CK(hipStreamWaitEvent(s1, join, 0));
  hipGraph_t g;
  CK(hipStreamEndCapture(s1, &g));
  size_t nn = 0, ne = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(g, nodes.data(), &nn));
  std::map<hipGraphNode_t, int> id;
  for (size_t i = 0; i < nn; ++i) {
    hipKernelNodeParams kp{};
    int v = -1;
    if (hipGraphKernelNodeGetParams(nodes[i], &kp) == hipSuccess && kp.kernelParams) v = *reinterpret_cast<int*>(kp.kernelParams[1]);
    id[nodes[i]] = v;
  }
  CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  std::vector<hipGraphNode_t> from(ne), to(ne);
  CK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
  printf("variant %d: %zu nodes, edges:", variant, nn);
  bool k2k3 = false;
  for (size_t i = 0; i < ne; ++i) { printf(" K%d->K%d", id[from[i]], id[to[i]]); if (id[from[i]] == 2 && id[to[i]] == 3) k2k3 = true; }
  printf("   [K2->K3 %s]\n", k2k3 ? "present" : "MISSING");
  return 0;
}
int main() { for (int v = 0; v < 4; ++v) if (run(v)) return 1; return 0; }
