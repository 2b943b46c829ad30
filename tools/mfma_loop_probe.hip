// Probe: what costs MFMA issue slots in a k-loop shaped like the library's (16 MFMAs on 4
// accumulators per half-iteration, operands from registers)?  Variants add one ingredient each.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters, int one) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a = {1.f, 2.f, 3.f, 4.f}, b[4];
  for (int q = 0; q < 4; ++q) b[q] = f32x4{0.5f * q, 1.f, 2.f, 3.f + threadIdx.x * 1e-9f};
  __shared__ float lds[4096];
  if (V >= 4) { for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i; __syncthreads(); }
  if (V == 2 || V == 3) __builtin_amdgcn_s_setprio(1);
  for (int it = 0; it < iters; ++it) {
    if (V >= 4) {  // LDS operand reads like the scoring loop (4 ds_read_b128 per 16 MFMAs)
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<f32x4 *>(&lds[(q * 256 + (threadIdx.x & 63) * 4 + (it & 1) * 1024) & 4095]);
    }
    if (V >= 5) a = *reinterpret_cast<const f32x4 *>(in + ((it * one) & 1023) * 256 + (threadIdx.x & 63) * 4);
    if (V == 1 || V == 3 || V >= 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[q][e], acc[q], 0, 0, 0);
    if (V == 1 || V == 3 || V >= 4) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int V>
void run(const char *name) {
  float *out, *in;
  hipMalloc(&out, 4);
  hipMalloc(&in, 1024 * 256 * 4 + 4096);
  hipMemset(in, 0, 1024 * 256 * 4);
  const int iters = 40000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 0, 0, out, in, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  double flops = 256.0 * 8 * iters * 16.0 * 4096.0;
  printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, best, flops / best / 1e9);
}

int main() {
  run<0>("V0 plain 16-MFMA loop");
  run<1>("V1 + sched_barrier fences");
  run<2>("V2 + s_setprio(1)");
  run<3>("V3 + fences + setprio");
  run<4>("V4 + 4 ds_read_b128 per iteration (fenced)");
  run<5>("V5 + 1 global_load_dwordx4 per iteration");
  return 0;
}
