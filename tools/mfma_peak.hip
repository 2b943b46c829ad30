// Sustained fp32-MFMA throughput of this MI355X (power/clock limited reality vs the
// 157.3 TF datasheet peak): v_mfma_f32_32x32x2_f32 on register operands only.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x * 1e-9f, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run(int waves_per_cu, const char *name) {
  float *out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  dim3 grid(256), block(64 * waves_per_cu);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * waves_per_cu * iters * 8.0 * NACC * 4096.0;
    printf("%s waves/CU=%d acc=%d: %.3f ms  %.1f TFLOP/s\n", name, waves_per_cu, NACC, ms, flops / ms / 1e9);
  }
}

int main() {
  run<4>(4, "f32 32x32x2");
  run<4>(8, "f32 32x32x2");
  run<2>(8, "f32 32x32x2");
  return 0;
}
