// graph_single_parent_race.hip -- stand-alone reduction of the stale read seen under FILM's two-lane hipGraph replay.
// Two capturing streams.  Lane 1: ... P, Q   (Q reads what P writes; same stream, so the captured graph has the edge P -> Q and Q has
// no other parent).  Lane 0: ... B, where B waits for an event recorded behind P (P gets a second child that was captured BEFORE Q).
// Every replay bumps a device counter first; P spins, then writes the counter; Q compares.  A replay that runs Q before P has finished
// sees the previous replay's value.  Variants: waiters = how many lane-0 kernels wait for P's event (FILM had 5: every op of lane 0
// that touched a buffer P had touched waited again), chain/fan = the waiters follow each other on lane 0 (so all waits but the first
// are implied) or sit on streams of their own, relay = Q additionally waits for an event recorded behind the first waiter (the
// round-2/3 workaround; the planner now waits at most once per op, Planner::analyze_lanes).
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/graph_single_parent_race.hip -o tools/bin/graph_single_parent_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return -1; } } while (0)
__global__ void bump(int* it) { it[0] += 1; }
__global__ void work(int* sink, int spin) { for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64); if (sink) sink[0] = spin; }
__global__ void produce(const int* it, int* v, int spin) { for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64); v[0] = it[0]; }
__global__ void consume(const int* it, const int* v, int* stale) { if (v[0] != it[0]) atomicAdd(stale, 1); }

static int run(int chain, int waiters, bool relay, int spin, int replays, bool fan = false) {
  hipStream_t s0, s1, user;
  for (hipStream_t* s : {&s0, &s1, &user}) CK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  int* d;
  CK(hipMalloc(&d, 64 * sizeof(int)));
  CK(hipMemset(d, 0, 64 * sizeof(int)));
  int *it = d, *v = d + 1, *stale = d + 2, *sink = d + 8;
  std::vector<hipStream_t> ws(fan ? waiters : 0);   // fan: every waiter on a stream of its own (no transitively implied edges)
  for (hipStream_t& w : ws) CK(hipStreamCreateWithFlags(&w, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(8 + waiters);
  for (hipEvent_t& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s0, it);
  CK(hipEventRecord(ev[0], s0));
  CK(hipStreamWaitEvent(s1, ev[0], 0));                                       // fork
  for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s0, sink + i, 1);       // lane 0 before B
  for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s1, sink + 8 + i, 1);   // lane 1 before P
  hipLaunchKernelGGL(produce, dim3(1), dim3(1), 0, s1, it, v, spin);          // P
  CK(hipEventRecord(ev[1], s1));
  for (int i = 0; i < waiters; ++i) {                                          // B...: lane 0 kernels that wait for P
    hipStream_t w = fan ? ws[i] : s0;
    if (fan) CK(hipStreamWaitEvent(w, ev[0], 0));
    CK(hipStreamWaitEvent(w, ev[1], 0));
    hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, w, sink + 16 + i, 1);
    if (i == 0 && relay) { CK(hipEventRecord(ev[2], w)); CK(hipStreamWaitEvent(s1, ev[2], 0)); }
    if (fan) { CK(hipEventRecord(ev[8 + i], w)); CK(hipStreamWaitEvent(s0, ev[8 + i], 0)); }
  }
  hipLaunchKernelGGL(consume, dim3(1), dim3(1), 0, s1, it, v, stale);          // Q: next kernel of P's own stream
  CK(hipEventRecord(ev[3], s1));
  CK(hipStreamWaitEvent(s0, ev[3], 0));                                       // join
  hipGraph_t g;
  CK(hipStreamEndCapture(s0, &g));
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, user));
  CK(hipStreamSynchronize(user));
  int h[3];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(d));
  return h[2];
}

int main() {
  int rt = 0;
  (void)hipRuntimeGetVersion(&rt);
  printf("HIP runtime %d\n", rt);
  const int replays = 200;
  for (int spin : {0, 200})
    for (int fan = 0; fan < 2; ++fan)
      for (int waiters : {0, 1, 2, 3, 4, 5, 8})
        for (int relay = 0; relay <= (waiters ? 1 : 0); ++relay)
          printf("spin %4d  %s  waiters %d  relay %d : %3d of %d replays read the previous replay's value\n", spin, fan ? "fan  " : "chain", waiters, relay,
                 run(3, waiters, relay != 0, spin, replays, fan != 0), replays);
  return 0;
}
