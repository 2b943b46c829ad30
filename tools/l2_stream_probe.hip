// Probe: how fast can EVERY compute unit stream the same (L2-resident) weight matrix at once?  The training kernels
// (lstm_fwd_x3 TRAIN, lstm_bwd X3) have each of their 8 waves read a private 128 KiB slice of a 1 MiB matrix per step,
// all 256 workgroups the same matrix.  Prints bytes / clock / CU for: one shared copy, COPIES copies (workgroup -> copy
// blockIdx / 8 % COPIES), ring depths 2 / 4 / 8, and 4 vs 8 waves loading.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2probe tools/l2_stream_probe.hip && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4 *w, int copies, size_t copy_elems, int groups, int reps, int waves,
                                                      int stagger, u32x4 *out, long long *cycles) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv >= waves) return;
  const u32x4 *base = w + (size_t)((blockIdx.x >> 3) % copies) * copy_elems + (size_t)wv * groups * 128 + lane;
  u32x4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    u32x4 ring[DEPTH][2];
    const int rot = stagger ? (wv * (groups / 8) + blockIdx.x * 3) % groups : 0;
    auto idx = [&](int g) { int j = g + rot; return j >= groups ? j - groups : j; };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      ring[d][0] = base[(size_t)idx(d) * 128];
      ring[d][1] = base[(size_t)idx(d) * 128 + 64];
    }
    for (int g = 0; g < groups; g += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc ^= ring[d][0];
        acc ^= ring[d][1];
        const int gn = idx(g + d + DEPTH < groups ? g + d + DEPTH : g + d);
        ring[d][0] = base[(size_t)gn * 128];
        ring[d][1] = base[(size_t)gn * 128 + 64];
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int groups = 64, waves_max = 8, reps = 32, blocks = 256;
  const size_t copy_elems = (size_t)waves_max * groups * 128;  // u32x4 elements = 1 MiB
  const int max_copies = 8;
  u32x4 *w, *out;
  long long *cyc;
  (void)hipMalloc(&w, max_copies * copy_elems * 16);
  (void)hipMemset(w, 1, max_copies * copy_elems * 16);
  (void)hipMalloc(&out, blocks * 512 * 16);
  (void)hipMalloc(&cyc, blocks * 8);
  std::vector<long long> h(blocks);
  auto run = [&](int depth, int copies, int waves, int stagger) {
    for (int it = 0; it < 2; ++it) {
      if (depth == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(512), 0, 0, w, copies, copy_elems, groups, reps, waves, stagger, out, cyc);
      if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(blocks), dim3(512), 0, 0, w, copies, copy_elems, groups, reps, waves, stagger, out, cyc);
      if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(512), 0, 0, w, copies, copy_elems, groups, reps, waves, stagger, out, cyc);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto c : h) s += (double)c;
    s /= blocks;
    const double bytes = (double)reps * waves * groups * 2048;
    printf("depth %d copies %d waves %d stagger %d: %.1f B/clk/CU  (%.0f cycles per MiB per CU)\n", depth, copies, waves, stagger,
           bytes / s, s / (bytes / 1048576.0));
  };
  for (int stagger = 0; stagger < 2; ++stagger)
    for (int depth : {2, 4, 8})
      for (int copies : {1, 4})
        for (int waves : {4, 8}) run(depth, copies, waves, stagger);
  return 0;
}
