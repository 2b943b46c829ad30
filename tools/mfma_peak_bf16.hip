// What this MI355X sustains on v_mfma_f32_32x32x16_bf16 with register operands only (no LDS, no global traffic inside the loop):
// the power / clock limited ceiling of the bf16 candidate sweep, against the 2.5 PF datasheet peak.  Operands are random bf16
// data (16 A fragments x 4 B fragments per "tile", the sweep's shape: 64 MFMAs into 4 accumulators), not constants: the
// matrix pipe's power -- and so the clock the chip settles at -- depends on how much the operands toggle.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o /tmp/mfma_peak_bf16 && /tmp/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

template <int NB, bool ZEROS>
__global__ __launch_bounds__(512) void k(const f32x4 *__restrict__ frag, float *out, int tiles, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63;
  f32x4 a[16], b[NB];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = ZEROS ? f32x4{0, 0, 0, 0} : frag[(i * 64 + lane)];
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = ZEROS ? f32x4{0, 0, 0, 0} : frag[((16 + i) * 64 + lane)];
  f32x16 acc[NB];
  float s = 0;
  const unsigned long long t0 = clock64();
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int kg = 0; kg < 16; ++kg)
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[kg]), __builtin_bit_cast(bf16x8_t, b[q]),
                                                         kg == 0 ? z : acc[q], 0, 0, 0);
      }
    // keep the accumulators live without an epilogue worth mentioning
#pragma unroll
    for (int q = 0; q < NB; ++q) s += acc[q][t & 15];
    // rotate the fragments so that consecutive tiles do not repeat operands
    const f32x4 tmp = a[0];
#pragma unroll
    for (int i = 0; i < 15; ++i) a[i] = a[i + 1];
    a[15] = tmp;
  }
  const unsigned long long t1 = clock64();
  if (s == 12345.f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NB, bool ZEROS>
static void run(const f32x4 *frag, int waves_per_cu, int tiles, const char *name) {
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, 4);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, ZEROS>), dim3(256), dim3(64 * waves_per_cu), 0, 0, frag, out, tiles, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * waves_per_cu * (double)tiles * 16.0 * NB * 32768.0;
    // clock64 (s_memtime) counts shader clocks: ticks / time = the clock the chip ran at; at 100 % busy a SIMD needs
    // waves/4 * tiles * 64 MFMAs * 32 clocks
    const double need = waves_per_cu / 4.0 * tiles * 64.0 * 32.0;
    printf("%-28s waves/CU=%d tiles=%d: %.3f ms  %.0f TFLOP/s = %.2f of 2.5 PF   clock %.2f GHz, matrix pipe %.0f %% busy\n", name,
           waves_per_cu, tiles, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, c / ms / 1e6, 100.0 * need / c);
  }
  hipFree(out);
  hipFree(cyc);
}

int main() {
  std::vector<unsigned short> hb(20 * 512);
  srand(1);
  for (auto &v : hb) {  // random bf16 in (-1, 1): sign, exponent 119..126, random mantissa
    const unsigned sign = rand() & 1u, e = 119u + (rand() % 8u), m = rand() & 0x7Fu;
    v = (unsigned short)((sign << 15) | (e << 7) | m);
  }
  f32x4 *frag;
  hipMalloc(&frag, hb.size() * 2);
  hipMemcpy(frag, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  const int tiles = 2500;  // ~ the sweep's tiles per wave (8192 x 1.25 M: 2441 per SIMD pair)
  run<4, false>(frag, 8, tiles, "random operands");
  run<4, false>(frag, 4, tiles * 2, "random operands");
  run<4, true>(frag, 8, tiles, "zero operands");
  run<4, false>(frag, 8, tiles * 4, "random operands, 4x longer");
  return 0;
}
