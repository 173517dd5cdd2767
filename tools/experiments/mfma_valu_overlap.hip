// mfma_valu_overlap.hip -- does a VALU instruction issued between two v_mfma_f32_32x32x2_f32 cost matrix-pipe time on gfx950?
// One workgroup per CU, W waves per SIMD (W = 1, 2); every wave runs a loop of 8 independent-accumulator MFMAs with V independent
// v_fma_f32 (other registers) behind each; prints cycles per MFMA per SIMD.  64.0 = the pipe rate; a slope of ~c cycles per filler =
// fillers and fp32 MFMAs serialise (the f32 MFMA runs at the f32 VECTOR rate: same units?), flat = they overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int KIND>   // KIND 0: v_fma_f32, 1: v_mov_b32 (no FP unit), 2: v_pk_fma_f32, 3: ds_read_b128 (LDS), 4: v_add_u32 (integer)
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ float4 sm[1024];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = f2{a + i, a - i};
  int q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = threadIdx.x + i;
  sm[threadIdx.x] = float4{a, a, a, a};
  sm[threadIdx.x + 512] = float4{b, b, b, b};
  __syncthreads();
  unsigned sacc = 0;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, -1, 0x00020000);
  float4 ld[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ld[i] = float4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int i = (m * V + v) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(b));
        else if (KIND == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(b));
        else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(g[i]) : "v"(g[(i + 1) & 7]));
        else if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i]) : "v"((unsigned)((threadIdx.x + i * 64) & 1023) * 16u));
        else if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
        else if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[i]) : "v"(out + (threadIdx.x & 63) * 4) : "memory");
        else if (KIND == 6) asm volatile("s_nop 0");
        else if (KIND == 10) asm volatile("global_load_dword %0, %1, off" : "=v"(ld[i].x) : "v"(out + (threadIdx.x & 63)) : "memory");
        else if (KIND == 11) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"((unsigned)(i * 1024)), "v"((unsigned)((threadIdx.x & 63) * 16)), "s"(rsrc) : "memory");
        else if (KIND == 12) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[i]) : "v"((unsigned)((threadIdx.x & 63) * 16)), "s"(rsrc) : "memory");
        else if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)");
        else if (KIND == 8) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g[i]) : "v"(g[(i + 1) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 5 || KIND >= 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)sacc + acc[i][0] + f[i] + g[i][0] + g[i][1] + (float)q[i] + ld[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int V, int KIND>
void run(const char* what, int waves_per_simd, float* d_out, unsigned long long* d_cyc) {
  const int iters = 2000, nb = 256, nt = waves_per_simd * 4 * 64;
  hipLaunchKernelGGL((probe<V, KIND>), dim3(nb), dim3(nt), 0, 0, d_out, d_cyc, iters);
  hipLaunchKernelGGL((probe<V, KIND>), dim3(nb), dim3(nt), 0, 0, d_out, d_cyc, iters);
  static unsigned long long h[256 * 16];
  hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;   // per workgroup: last wave's end - first wave's start
  for (int i = 0; i < nb; ++i) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nt / 64; ++w) { lo = h[(i * 8 + w) * 2] < lo ? h[(i * 8 + w) * 2] : lo; hi = h[(i * 8 + w) * 2 + 1] > hi ? h[(i * 8 + w) * 2 + 1] : hi; }
    avg += (double)(hi - lo);
  }
  avg /= nb;
  // per SIMD: waves_per_simd waves x iters x 8 MFMAs in `avg` cycles
  printf("  %-18s V=%2d  waves/SIMD %d:  %7.1f cycles per MFMA per SIMD (%5.1f per MFMA of one wave)\n", what, V, waves_per_simd, avg / (iters * 8.0 * waves_per_simd),
         avg / (iters * 8.0));
}

int main() {
  float* d_out;
  unsigned long long* d_cyc;
  hipMalloc(&d_out, 256 * 512 * 4);
  hipMalloc(&d_cyc, 256 * 16 * 8);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>("none", w, d_out, d_cyc);
    run<1, 0>("v_fma_f32", w, d_out, d_cyc);
    run<2, 0>("v_fma_f32", w, d_out, d_cyc);
    run<3, 0>("v_fma_f32", w, d_out, d_cyc);
    run<4, 0>("v_fma_f32", w, d_out, d_cyc);
    run<6, 0>("v_fma_f32", w, d_out, d_cyc);
    run<8, 0>("v_fma_f32", w, d_out, d_cyc);
    run<12, 0>("v_fma_f32", w, d_out, d_cyc);
    run<4, 1>("v_mov_b32", w, d_out, d_cyc);
    run<8, 1>("v_mov_b32", w, d_out, d_cyc);
    run<4, 4>("v_add_u32", w, d_out, d_cyc);
    run<8, 4>("v_add_u32", w, d_out, d_cyc);
    run<2, 2>("v_pk_fma_f32", w, d_out, d_cyc);
    run<4, 2>("v_pk_fma_f32", w, d_out, d_cyc);
    run<6, 2>("v_pk_fma_f32", w, d_out, d_cyc);
    run<1, 5>("global_load_x4", w, d_out, d_cyc);
    run<2, 5>("global_load_x4", w, d_out, d_cyc);
    run<1, 10>("global_load_dword", w, d_out, d_cyc);
    run<2, 10>("global_load_dword", w, d_out, d_cyc);
    run<1, 11>("buffer_load_x4 lds", w, d_out, d_cyc);
    run<2, 11>("buffer_load_x4 lds", w, d_out, d_cyc);
    run<1, 12>("buffer_load_x4", w, d_out, d_cyc);
    run<2, 6>("s_nop 0", w, d_out, d_cyc);
    run<4, 6>("s_nop 0", w, d_out, d_cyc);
    run<2, 7>("s_waitcnt", w, d_out, d_cyc);
    run<4, 8>("s_add_u32", w, d_out, d_cyc);
    run<4, 9>("v_pk_add_f32", w, d_out, d_cyc);
    run<1, 3>("ds_read_b128", w, d_out, d_cyc);
    run<2, 3>("ds_read_b128", w, d_out, d_cyc);
  }
  return 0;
}
