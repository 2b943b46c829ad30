// LSTM sequence encoder forward for FEW sequences (demo / web queries, evaluator tails; sse_demo.py:121-125,
// webserver.py:144-147, sse_index.py:66-68 last batch) on gfx950.
//
// The throughput kernel (lstm_fwd.hip) is built from 32-row MFMA tiles: a step costs ~38 us whether the tile holds 32
// sequences or one (B = 1: 818 seq/s, 1.22 ms per call at T = 32 in round 1).  For a handful of rows the gate
// "GEMM" is a GEMV: 2*(E+H)*4H flops per row against (E+H)*4H*4 bytes of weights, i.e. bound by streaming the
// 1.25 MB kernel matrix from L2 every step, and the vector ALUs (64 fma / 4 cycles / SIMD) are 16x the useful rate of
// a 32-row matrix tile at one row.  So: one workgroup per RB = 4 sequences, a thread owns 4 consecutive gate columns
// of the TF kernel matrix [(E+H)][4H] (read row-major straight from the master variable: for a fixed k a wave reads
// 1 KiB of consecutive bytes), x_t | h_{t-1} sit in LDS and are broadcast, accumulators in registers, a deep register
// ring of weights in flight (the L2 stream is the bound).
//
// Arithmetic: the same fp32 fma chain, in the same k order, as the matrix path -- within a k-group of 8 the MFMA
// consumes k = 0,4,1,5,2,6,3,7; the bias enters as the product 1.0 * b at k = E; zero-padding terms are exact no-ops and
// skipped -- then the same gate formulas (v_exp_f32 / v_rcp_f32), projection order and sum-of-squares tree.
#include <cstdlib>

#include "sse_kernels.h"

#define LS_RB 4  // sequences per workgroup

__device__ __forceinline__ float ls_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float ls_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

// ring slots d = 0 .. D-1 of one pass over the ring, by template recursion: the slot index must be a compile-time
// constant (a rolled loop indexes the register ring dynamically, i.e. through scratch memory)
template <int CPT> struct ColVec;
template <> struct ColVec<4> { typedef f32x4 type; };
template <> struct ColVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int d, int D, int CPT, typename WLoad>
__device__ __forceinline__ void ring_steps(typename ColVec<CPT>::type (&ring)[D][8], typename ColVec<CPT>::type (&acc)[LS_RB],
                                           const float *av, int KA, int g0, int NG, WLoad &wload) {
  if constexpr (d < D) {
    const int g = g0 + d;
    __builtin_amdgcn_sched_barrier(0);  // one slot at a time: hoisting every slot's operands costs 100+ registers
    f32x4 lo[LS_RB], hi[LS_RB];
#pragma unroll
    for (int b = 0; b < LS_RB; ++b) {
      lo[b] = *reinterpret_cast<const f32x4 *>(av + b * KA + g * 8);       // broadcast reads
      hi[b] = *reinterpret_cast<const f32x4 *>(av + b * KA + g * 8 + 4);
    }
    // the matrix path's fma chain: inside a k-group k = 0,4,1,5,2,6,3,7, each product one fma onto the accumulator
    // (v_mfma_f32_32x32x2_f32 == fma(a[k0], b[k0], c) then fma(a[k1], b[k1], .), checked bit for bit on the GPU)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < LS_RB; ++b)
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          acc[b][j] = __builtin_fmaf(lo[b][e], ring[d][e][j], acc[b][j]);
          acc[b][j] = __builtin_fmaf(hi[b][e], ring[d][4 + e][j], acc[b][j]);
        }
    // refill the slot D groups on (wrapping into the next step: the weights do not change)
    wload(g + D < NG ? g + D : g + D - NG, ring[d]);
    __builtin_amdgcn_sched_barrier(0);
    ring_steps<d + 1, D, CPT>(ring, acc, av, KA, g0, NG, wload);
  }
}

// One thread owns CPT consecutive gate columns (8- or 16-byte weight loads).  D = depth of the weight ring in k-groups:
// the L2 round trip under this access pattern is ~1 us, and 1.25 MB per ~10 us step needs >= 128 KB in flight per CU
// (a first version with dword loads and one k-group of prefetch ran 48 us per step, slower than the matrix kernel;
// now 256 threads x 4 columns x D = 4: 17 us).
template <int NT, int D, int CPT>
__global__ __launch_bounds__(NT) void lstm_small_kernel(LstmSmallArgs a) {
  typedef typename ColVec<CPT>::type colv;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int E = a.E, H = a.H, T = a.T, S = a.S, N4 = 4 * H;
  // unified k space, whole k-groups of 8: [x_t (E) | 1.0 (the bias column, as in the packed matrix kernel) | 0.. ] of
  // KX = round_up(E + 1, 8), then [h_{t-1} (H) | 0..] of KH = round_up(H, 8); padded with all-zero groups to a
  // multiple of D so that the ring slots are compile-time registers (their products are exact no-ops)
  const int KX = (E + 8) & ~7, KH = (H + 7) & ~7;
  const int NG = ((KX + KH) / 8 + D - 1) / D * D, KA = NG * 8;
  float *av = sm;                          // [RB][KA] operand rows
  float *gs = av + LS_RB * KA;             // [RB][N4] gate pre-activations; later [RB][S] raw encodings
  float *red = gs + LS_RB * (N4 > S ? N4 : S);  // [RB][16] per-tile sums of squares / scratch
  const int b0 = blockIdx.x * LS_RB;
  const int nb = min(LS_RB, a.B - b0);

  // left-pad prefix skip (exact, as in the matrix kernel): start at the smallest leading-PAD count of the rows
  int t0 = 0;
  if (a.pad_h != nullptr) {
    int lead = T;
    const int r = tid >> 6;  // wave r scans row r
    if (r < nb) {
      const int32_t *row = a.ids + (size_t)(b0 + r) * T;
      for (int t = lane; t < T; t += 64)
        if (row[t] != 0) {
          lead = t;
          break;
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    }
    if (lane == 0 && r < LS_RB) red[r] = __int_as_float(lead);
    __syncthreads();
    lead = T;
    for (int r2 = 0; r2 < nb; ++r2) lead = min(lead, __float_as_int(red[r2]));
    t0 = min(lead, T - 1);
    __syncthreads();
  }
  auto fetch_id = [&](int b, int t) -> int {
    int id = (b < nb) ? a.ids[(size_t)(b0 + b) * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  // x_{t0}, the constant 1, and the state after t0 PAD steps
  for (int i = tid; i < LS_RB * KA; i += NT) {
    const int b = i / KA, k = i - b * KA;
    float v = 0.0f;
    if (k < E) v = a.emb[(size_t)fetch_id(b, t0) * E + k];
    else if (k == E) v = 1.0f;
    else if (k >= KX && k - KX < H && t0 > 0) v = a.pad_h[(size_t)t0 * a.pad_stride + (k - KX)];
    av[i] = v;
  }
  // cell state: element e = tid + j * NT of [RB][H]
  constexpr int CE = (LS_RB * 512 + NT - 1) / NT;  // H <= 512
  float c[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) {
    const int e = tid + j * NT;
    c[j] = (e < LS_RB * H && t0 > 0) ? a.pad_c[(size_t)t0 * a.pad_stride + e % H] : 0.0f;
  }
  // this thread's CPT columns n0 .. (4H is a multiple of 4)
  const int n0 = tid * CPT;
  __syncthreads();

  // weights: the augmented matrix Waug [KA][4H] (pack_lstm_small: x rows, the bias row at k = E, zero rows, h rows,
  // zero rows -- the unified k space spelled out, so a k-group is 8 consecutive rows and no load is conditional)
  // through a buffer descriptor: per-lane byte offset 16*tid (constant), the row as a scalar offset; threads beyond 4H
  // point past the descriptor's range and read 0
  const __amdgpu_buffer_rsrc_t krs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Waug), 0, KA * N4 * 4, 0x00020000);
  const int kvo = (n0 < N4) ? n0 * 4 : KA * N4 * 4;
  const int rowb = N4 * 4;
  auto wload = [&](int g, colv (&w)[8]) {  // k-group g (wave-uniform)
    const int base = g * 8 * rowb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (CPT == 4) w[i] = __builtin_bit_cast(colv, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo, base + i * rowb, 0));
      else w[i] = __builtin_bit_cast(colv, __builtin_amdgcn_raw_buffer_load_b64(krs, kvo, base + i * rowb, 0));
    }
  };
  colv ring[D][8];
#pragma unroll
  for (int d = 0; d < D; ++d) wload(d, ring[d]);

  for (int t = t0; t < T; ++t) {
    colv acc[LS_RB];
#pragma unroll
    for (int b = 0; b < LS_RB; ++b)
#pragma unroll
      for (int j = 0; j < CPT; ++j) acc[b][j] = 0.0f;
    for (int g0 = 0; g0 < NG; g0 += D) ring_steps<0, D, CPT>(ring, acc, av, KA, g0, NG, wload);
    if (n0 < N4) {
#pragma unroll
      for (int b = 0; b < LS_RB; ++b) *reinterpret_cast<colv *>(gs + b * N4 + n0) = acc[b];
    }
    __syncthreads();  // gates complete; nobody reads the operand rows any more
    // elementwise: element e = (b, unit)
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      const int e = tid + j * NT;
      if (e < LS_RB * H) {
        const int b = e / H, unit = e - b * H;
        const float *g = gs + b * N4;
        const float si = ls_sigmoid(g[unit]);
        const float tj = ls_tanh(g[H + unit]);
        const float sf = ls_sigmoid(g[2 * H + unit]);
        const float so = ls_sigmoid(g[3 * H + unit]);
        const float pij = __fmul_rn(si, tj);           // the matrix kernel parks this product (rounded) between its two passes
        const float cn = __builtin_fmaf(c[j], sf, pij);
        c[j] = cn;
        const float hv = ls_tanh(cn) * so;
        av[b * KA + KX + unit] = hv;
        if (a.rec_h != nullptr && blockIdx.x == 0 && b == 0) {  // sequence 0: the pad-prefix table of THIS kernel
          a.rec_h[(size_t)(t + 1) * a.pad_stride + unit] = hv;
          a.rec_c[(size_t)(t + 1) * a.pad_stride + unit] = cn;
        }
      }
    }
    if (t + 1 < T) {
      for (int i = tid; i < LS_RB * E; i += NT) {
        const int b = i / E, k = i - b * E;
        av[b * KA + k] = a.emb[(size_t)fetch_id(b, t + 1) * E + k];
      }
    }
    __syncthreads();
  }

  // projection out[b][s] = sum_unit h[b][unit] * M[unit][s] in the matrix path's k order, then tf.nn.l2_normalize
  for (int i = tid; i < LS_RB * S; i += NT) {
    const int b = i / S, s = i - b * S;
    const float *hb = av + b * KA + KX;
    float acc = 0.0f;
    for (int kb = 0; kb < KH; kb += 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = kb + e, k1 = kb + 4 + e;
        if (k0 < H) acc = __builtin_fmaf(hb[k0], a.M[(size_t)k0 * S + s], acc);
        if (k1 < H) acc = __builtin_fmaf(hb[k1], a.M[(size_t)k1 * S + s], acc);
      }
    }
    gs[b * S + s] = acc;
  }
  __syncthreads();
  const int NTS = (S + 31) / 32;
  if (a.normalize) {
    // per 32-column tile: squares summed by the xor-shuffle tree of the matrix kernel, tiles then added in order
    for (int i = tid; i < LS_RB * NTS * 32; i += NT) {  // 32-lane groups stay whole
      const int col = i & 31, tile = (i >> 5) % NTS, b = i / (32 * NTS);
      const int s = tile * 32 + col;
      float v = (s < S) ? gs[b * S + s] : 0.0f;
      v = v * v;
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if (col == 0) red[b * 16 + tile] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < LS_RB * S; i += NT) {
    const int b = i / S, s = i - b * S;
    if (b >= nb) continue;
    float scale = 1.0f;
    if (a.normalize) {
      float tot = 0.0f;
      for (int j = 0; j < NTS; ++j) tot += red[b * 16 + j];
      scale = 1.0f / sqrtf(fmaxf(tot, 1e-12f));
    }
    a.out[(size_t)(b0 + b) * S + s] = gs[i] * scale;
  }
}

static int ls_groups(int E, int H, int D) {
  const int KX = (E + 8) & ~7, KH = (H + 7) & ~7;
  return ((KX + KH) / 8 + D - 1) / D * D;
}

// Waug [KA][4H] from the TF kernel [(E+H)][4H] and bias [4H] (forget_bias 1.0 folded into the f block)
__global__ void pack_lstm_small_kernel(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int KX, int KA,
                                       float *__restrict__ out) {
  const int N4 = 4 * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)KA * N4; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / N4), n = (int)(i - (int64_t)k * N4);
    float v = 0.0f;
    if (k < E) v = K[(size_t)k * N4 + n];
    else if (k == E) v = b[n] + ((n >= 2 * H && n < 3 * H) ? 1.0f : 0.0f);
    else if (k >= KX && k - KX < H) v = K[(size_t)(E + k - KX) * N4 + n];
    out[i] = v;
  }
}

size_t lstm_small_waug_floats(int E, int H) { return (size_t)ls_groups(E, H, 4) * 8 * 4 * H; }

hipError_t launch_pack_lstm_small(const float *K, const float *b, int E, int H, float *out, hipStream_t stream) {
  const int KX = (E + 8) & ~7, KA = ls_groups(E, H, 4) * 8;
  const int64_t n = (int64_t)KA * 4 * H;
  hipLaunchKernelGGL(pack_lstm_small_kernel, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, stream, K, b,
                     E, H, KX, KA, out);
  return hipGetLastError();
}

size_t lstm_small_lds_bytes(int E, int H, int S) {
  const int D = 4;
  const size_t g = (size_t)LS_RB * (4 * H > S ? 4 * H : S);
  return ((size_t)LS_RB * ls_groups(E, H, D) * 8 + g + LS_RB * 16) * sizeof(float);
}

hipError_t launch_lstm_small(const LstmSmallArgs &a, hipStream_t stream) {
  if (a.H > 512 || a.S > 512 || a.B < 1) return hipErrorInvalidValue;
  const size_t lds = lstm_small_lds_bytes(a.E, a.H, a.S);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const dim3 grid((a.B + LS_RB - 1) / LS_RB);
  // 4H <= 1024: 256 threads (one wave per SIMD) x 4 columns, ring depth 4 = 128 KB of weights in flight: 17 us per step
  // at H = 256 (measured; 512 threads x 2 columns x depth 8 keeps twice the bytes in flight but spills: 22 us)
  if (4 * a.H <= 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_small_kernel<256, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((lstm_small_kernel<256, 4, 4>), grid, dim3(256), lds, stream, a);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_small_kernel<512, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((lstm_small_kernel<512, 2, 4>), grid, dim3(512), lds, stream, a);
  }
  return hipGetLastError();
}
