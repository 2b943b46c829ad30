// Issue rate of the fp32 MFMAs from ONE wave per SIMD as a function of the number of independent accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_rate_probe tools/mfma_rate_probe.hip && tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N 4096
template <int CH>
__global__ void k16(float *out, long long *cyc, float a, float b) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  const long long t0 = clock64();
  for (int i = 0; i < N / CH; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
__global__ void k32(float *out, long long *cyc, float a, float b) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  const long long t0 = clock64();
  for (int i = 0; i < N / CH; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <typename F>
static void run(const char *name, F f, int threads) {
  float *o;
  long long *c, h;
  hipMalloc(&o, 4096 * 4);
  hipMalloc(&c, 8);
  for (int r = 0; r < 2; ++r) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    f<<<1, threads>>>(o, c, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    if (r) printf("%-28s %2d waves: %6.1f clock64 ticks per MFMA per wave\n", name, threads / 64, (double)h / N);
  }
}
int main() {
  run("16x16x4 f32, 1 chain", k16<1>, 64);
  run("16x16x4 f32, 2 chains", k16<2>, 64);
  run("16x16x4 f32, 4 chains", k16<4>, 64);
  run("16x16x4 f32, 2 chains", k16<2>, 256);
  run("16x16x4 f32, 2 chains", k16<2>, 512);
  run("16x16x4 f32, 4 chains", k16<4>, 512);
  run("32x32x2 f32, 1 chain", k32<1>, 64);
  run("32x32x2 f32, 2 chains", k32<2>, 64);
  run("32x32x2 f32, 1 chain", k32<1>, 256);
  run("32x32x2 f32, 1 chain", k32<1>, 512);
  run("32x32x2 f32, 2 chains", k32<2>, 512);
  return 0;
}
