// graph_many_streams.hip -- film-free attempt to reproduce the hipGraphLaunch SIGSEGV of profiles/r05_hipgraph_first_launch_crash.md:
// a process that has created and destroyed many stream pairs and instantiated / destroyed many TWO-BRANCH graphs launches a fresh
// one.  The faulting runtime function fills the exec's parallel-stream table on the FIRST launch of an exec with more than one branch.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/graph_many_streams.hip -o tools/bin/graph_many_streams
//   tools/bin/graph_many_streams [engines 12] [graphs per engine 6] [extra idle streams 8] [branches 2]
//   (against PyTorch's bundled runtime: LD_PRELOAD=<site-packages>/torch/lib/libamdhip64.so tools/bin/graph_many_streams ...)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void bump(float* p, int n, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] += v;
}

struct Engine { hipStream_t s = nullptr, s2 = nullptr; std::vector<hipGraph_t> g; std::vector<hipGraphExec_t> x; float* buf = nullptr; };

// the engine's capture pattern: fork from the capture stream into `branches - 1` side lanes, a few kernels per lane, join
static void capture(Engine& e, int nk, int branches, std::vector<hipStream_t>& lanes) {
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipStreamBeginCapture(e.s, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, e.s));
  for (int b = 1; b < branches; ++b) CK(hipStreamWaitEvent(b == 1 ? e.s2 : lanes[b - 2], fork, 0));
  for (int k = 0; k < nk; ++k)
    for (int b = 0; b < branches; ++b)
      hipLaunchKernelGGL(bump, dim3(64), dim3(256), 0, b == 0 ? e.s : b == 1 ? e.s2 : lanes[b - 2], e.buf + b * 16384, 16384, 1.f);
  for (int b = 1; b < branches; ++b) {
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventRecord(join, b == 1 ? e.s2 : lanes[b - 2]));
    CK(hipStreamWaitEvent(e.s, join, 0));
  }
  hipGraph_t g;
  CK(hipStreamEndCapture(e.s, &g));
  hipGraphExec_t x;
  CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  e.g.push_back(g); e.x.push_back(x);
}

int main(int argc, char** argv) {
  const int NE = argc > 1 ? atoi(argv[1]) : 12, NG = argc > 2 ? atoi(argv[2]) : 6, NIDLE = argc > 3 ? atoi(argv[3]) : 8, NB = argc > 4 ? atoi(argv[4]) : 2;
  std::vector<hipStream_t> idle(NIDLE), lanes(NB > 2 ? NB - 2 : 0);
  for (auto& s : idle) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));   // PyTorch's / RCCL's streams in the real process
  for (auto& s : lanes) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipStream_t user;
  CK(hipStreamCreate(&user));
  std::vector<Engine> eng(NE);
  long launches = 0;
  for (int i = 0; i < NE; ++i) {
    Engine& e = eng[i];
    CK(hipStreamCreateWithFlags(&e.s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&e.s2, hipStreamNonBlocking));
    CK(hipMalloc(&e.buf, 16384 * 4 * (NB + 1)));
    for (int g = 0; g < NG; ++g) {
      capture(e, 3 + g, NB, lanes);
      // first launch of a fresh exec, alternately on the engine's stream, the NULL stream and a user stream (the tests do all three)
      hipStream_t ls = g % 3 == 0 ? e.s : g % 3 == 1 ? (hipStream_t) nullptr : user;
      CK(hipGraphLaunch(e.x.back(), ls)); ++launches;
      CK(hipGraphLaunch(e.x.back(), ls)); ++launches;
      if (g % 2) { CK(hipStreamSynchronize(ls)); CK(hipGraphExecDestroy(e.x.back())); CK(hipGraphDestroy(e.g.back())); e.x.pop_back(); e.g.pop_back(); }   // dropped plans
    }
    if (i % 3 == 2) {   // an engine closes: its graphs and BOTH streams go away while others live on
      Engine& d = eng[i - 1];
      CK(hipDeviceSynchronize());
      for (auto x : d.x) CK(hipGraphExecDestroy(x));
      for (auto g : d.g) CK(hipGraphDestroy(g));
      d.x.clear(); d.g.clear();
      CK(hipStreamDestroy(d.s)); CK(hipStreamDestroy(d.s2)); d.s = d.s2 = nullptr;
      CK(hipFree(d.buf)); d.buf = nullptr;
    }
    for (Engine& o : eng) for (auto x : o.x) if (o.s) { CK(hipGraphLaunch(x, o.s)); ++launches; }   // older execs keep replaying
  }
  CK(hipDeviceSynchronize());
  int ver = 0;
  CK(hipRuntimeGetVersion(&ver));
  printf("no crash: %d engines x %d graphs of %d branches, %d idle streams, %ld launches, runtime %d\n", NE, NG, NB, NIDLE, launches, ver);
  return 0;
}
