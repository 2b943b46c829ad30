// Which fp32 summation order do the gfx950 fp32 MFMAs use?  C = A.B (M = N = 32 or 16, K = 64) by v_mfma_f32_32x32x2_f32,
// v_mfma_f32_16x16x4_f32 and a scalar ascending-k fmaf chain, compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_chain_probe tools/mfma_chain_probe.hip && tools/bin/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define K 64
__global__ void k32(const float *A, const float *B, float *C) {  // A [32][K], B [K][32]
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) {
    const float a = A[(lane & 31) * K + k + (lane >> 5)], b = B[(k + (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) C[((r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * 32 + (lane & 31)] = acc[r];
}
__global__ void k16(const float *A, const float *B, float *C) {  // the 16 x 16 corner of the same product
  const int lane = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int k = 0; k < K; k += 4) {
    const float a = A[(lane & 15) * K + k + (lane >> 4)], b = B[(k + (lane >> 4)) * 32 + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 32 + (lane & 15)] = acc[r];
}
int main() {
  std::vector<float> A(32 * K), B(K * 32), C32(1024), C16(1024, 0.f), R(1024);
  srand(1);
  for (auto &v : A) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto &v : B) v = (float)rand() / RAND_MAX - 0.5f;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[k * 32 + j], s);
      R[i * 32 + j] = s;
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k32<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C32.data(), dC, 4096, hipMemcpyDeviceToHost);
  hipMemset(dC, 0, 4096);
  k16<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C16.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad32 = 0, bad16 = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      bad32 += memcmp(&C32[i * 32 + j], &R[i * 32 + j], 4) != 0;
      if (i < 16 && j < 16) bad16 += memcmp(&C16[i * 32 + j], &R[i * 32 + j], 4) != 0;
    }
  printf("32x32x2 vs ascending fmaf chain: %d of 1024 differ; 16x16x4: %d of 256 differ\n", bad32, bad16);
  return 0;
}
