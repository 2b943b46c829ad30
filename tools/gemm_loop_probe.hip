// Probe: what limits the k-loop of the BPTT / forward kernels?  Every workgroup runs STEPS x GROUPS k-groups; per k-group a wave
// reads ONE 1 KiB A fragment from LDS (the dG / h tile), ONE 1 KiB B fragment from an L2-resident weight matrix through a ring
// of PF buffer loads, and issues 4 v_mfma_f32_32x32x2_f32 into one accumulator.  Variants: waves per workgroup (8 = two per
// SIMD, 4 = one per SIMD), ring depth, with / without the L2 stream, with / without the LDS reads.  Prints clock64 cycles per
// k-group per wave (4 MFMAs = 256 cycles of the matrix pipe).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_loop_probe tools/gemm_loop_probe.hip && tools/bin/gemm_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GROUPS 128
#define STEPS 16

template <int PF, bool L2, bool LDSR, int CH, bool WALK = false, bool STORES = false>
__global__ __launch_bounds__(512) void loop_kernel(const float *w, float *out, long long *cyc, int waves, int KGl = 32, int KGg = 32, float *dump = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < GROUPS * 256; i += blockDim.x) tile[i] = 1e-3f * (i & 15);
  __syncthreads();
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w) + (size_t)(wv & 7) * GROUPS * 256, 0, GROUPS * 1024, 0x00020000);
  auto ld = [&](int g) -> f32x4 {
    if constexpr (L2) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 1024, 0));
    return f32x4{1.0f, 0.5f, 0.25f, 2.0f};
  };
  const float *la = tile + lane * 4;
  struct Walk {
    int l, base;
  };
  auto adv = [&](Walk &x) {  // the BPTT kernel's (gate base, group in gate) walk
    const int l1 = x.l + 1;
    const bool wrap = l1 == KGl;
    const int b1 = x.base + KGg;
    x.base = wrap ? (b1 == 4 * KGg ? 0 : b1) : x.base;
    x.l = wrap ? 0 : l1;
  };
  const long long t0 = clock64();
  for (int s = 0; s < STEPS; ++s) {
    if constexpr (STORES) {  // the step's dG stores (16 x 1 KiB per wave, streamed to HBM) in front of the loop
      float *d = dump + ((size_t)(s * 256 + blockIdx.x) * 8 + wv) * 16 * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4 *>(d + i * 256) = f32x4{acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
      __syncthreads();
    }
    f32x4 bq[PF], aq[2];
    Walk wa{0, 0}, wb{0, 0};
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      bq[p] = ld(WALK ? wb.base + wb.l : p);
      if constexpr (WALK) adv(wb);
      __builtin_amdgcn_sched_barrier(0);
    }
    aq[0] = LDSR ? *reinterpret_cast<const f32x4 *>(la) : f32x4{1, 2, 3, 4};
    if constexpr (WALK) adv(wa);
    __builtin_amdgcn_s_setprio(1);
    for (int g = 0; g < GROUPS; g += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int gn = WALK ? wa.base + wa.l : (g + p + 1) & (GROUPS - 1);
        if constexpr (WALK) adv(wa);
        aq[(p + 1) & 1] = LDSR ? *reinterpret_cast<const f32x4 *>(la + gn * 256) : f32x4{1, 2, 3, 4};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[p][e], aq[p & 1][e], acc[e % CH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bq[p] = ld(WALK ? wb.base + wb.l : (g + p + PF) & (GROUPS - 1));
        if constexpr (WALK) adv(wb);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  const long long t1 = clock64();
  float sum = 0;
  for (int c = 0; c < CH; ++c) sum += acc[c][0] + acc[c][15];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && lane == 0) cyc[wv] = t1 - t0;
}

static float *g_dump = nullptr;
template <typename K>
static void run(const char *name, K kern, int waves, const float *w, float *out, long long *cyc) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GROUPS * 1024);
  long long h[8];
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), GROUPS * 1024, 0, w, out, cyc, waves, 32, 32, g_dump);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double lo = 1e30, hi = 0;
  for (int i = 0; i < waves; ++i) {
    const double v = (double)h[i] / (STEPS * GROUPS);
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  printf("%-44s %d waves/WG: %6.1f .. %6.1f cycles per k-group per wave (pipe: 256 per group and wave; %d waves per SIMD -> %d)\n", name, waves,
         lo, hi, waves / 4, 256 * (waves / 4));
}

int main() {
  float *w, *out;
  long long *cyc;
  hipMalloc(&w, (size_t)8 * GROUPS * 1024);
  hipMemset(w, 0, (size_t)8 * GROUPS * 1024);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  hipMalloc(&g_dump, (size_t)STEPS * 256 * 8 * 16 * 1024);
  for (int waves = 8; waves >= 4; waves -= 4) {
    run("L2 ring 4 + LDS + BPTT walk, 1 chain", loop_kernel<4, true, true, 1, true>, waves, w, out, cyc);
    run("L2 ring 4 + LDS + walk + dG stores", loop_kernel<4, true, true, 1, true, true>, waves, w, out, cyc);
    run("L2 ring 4 + LDS + dG stores (no walk)", loop_kernel<4, true, true, 1, false, true>, waves, w, out, cyc);
    run("registers only (no LDS, no L2), 1 chain", loop_kernel<4, false, false, 1>, waves, w, out, cyc);
    run("LDS reads only, 1 chain", loop_kernel<4, false, true, 1>, waves, w, out, cyc);
    run("L2 ring 4 + LDS reads, 1 chain", loop_kernel<4, true, true, 1>, waves, w, out, cyc);
    run("L2 ring 8 + LDS reads, 1 chain", loop_kernel<8, true, true, 1>, waves, w, out, cyc);
    run("L2 ring 4 + LDS reads, 2 chains", loop_kernel<4, true, true, 2>, waves, w, out, cyc);
    run("L2 ring 8 + LDS reads, 2 chains", loop_kernel<8, true, true, 2>, waves, w, out, cyc);
    run("L2 ring 8, no LDS reads, 1 chain", loop_kernel<8, true, false, 1>, waves, w, out, cyc);
  }
  return 0;
}
