// LSTM sequence encoder forward for gfx950 (MI355X): embedding gather + T
// BasicLSTMCell steps + projection + optional row L2-normalise, one launch.
//
// Replaces the TF graph built by sse_model.py:163-164 (embedding_lookup),
// :240-242/:248-250/:262-263/:273 (static_rnn over BasicLSTMCell, last step),
// :245/:254/:267/:275 (projection) and :282-283 (l2_normalize).
//
// Design (DESIGN.md "K2"): one 512-thread workgroup owns 64 sequences for all
// T steps.  h lives in LDS in MFMA A-fragment order, c in registers, the
// per-step gate GEMM [64,(Ep+Hp)] x [(Ep+Hp),4Hp] runs on
// v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF chip peak).  The kernel matrix is
// pre-packed so every B operand is one coalesced 1 KiB wave load from L2; the
// gate non-linearities are applied straight on the accumulators.
// Wave w: wn = w & 3 picks the hidden-unit range, wm = w >> 2 the 32-row half
// (waves w and w+4 share a SIMD and the same weights -> L1 reuse).
#include "sse_kernels.h"

#define LSTM_THREADS 512
#define LSTM_BM 64

size_t lstm_fwd_lds_bytes(int KGx, int KGh) {
  // xbuf[2 bufs][2 mt][KGx][256] + hbuf[2 mt][KGh][256] + red[64][4]
  return (size_t)(2 * 2 * KGx + 2 * KGh) * 256 * sizeof(float) + 64 * 4 * sizeof(float);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - __fdividef(2.0f, 1.0f + __expf(2.0f * x)); }

template <int UB, bool TRAIN>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_fwd_kernel(LstmFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w & 3, wm = w >> 2;
  const int KGx = a.KGx, KGh = a.KGh, KG = KGx + KGh, T = a.T;
  float *xbuf = smem;                      // [2][2][KGx][256]
  float *hbuf = smem + 4 * KGx * 256;      // [2][KGh][256]
  float *red = hbuf + 2 * KGh * 256;       // [64][4]
  const int b0 = blockIdx.x * LSTM_BM;

  // --- x gather assignment: 8 threads per sequence row, 8 floats (one k-group) each
  const int xr = tid >> 3, xq = tid & 7;
  const bool row_ok = (b0 + xr) < a.B;
  const int32_t *id_row = a.ids + (size_t)(row_ok ? (b0 + xr) : 0) * T;
  auto fetch_id = [&](int t) -> int {
    int id = row_ok ? id_row[t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  auto x_store = [&](int buf, int kg, f32x4 lo, f32x4 hi) {
    float *dst = xbuf + ((size_t)((buf * 2 + (xr >> 5)) * KGx + kg)) * 256;
    *reinterpret_cast<f32x4 *>(dst + (xr & 31) * 4) = lo;         // k%8 in 0..3 -> lane half 0
    *reinterpret_cast<f32x4 *>(dst + (32 + (xr & 31)) * 4) = hi;  // k%8 in 4..7 -> lane half 1
  };

  // --- prologue: x_0 -> xbuf[0], h_0 = 0
  {
    const int id = fetch_id(0);
    const float *src = a.emb + (size_t)id * a.Ep;
    for (int kg = xq; kg < KGx; kg += 8) {
      f32x4 lo = *reinterpret_cast<const f32x4 *>(src + kg * 8);
      f32x4 hi = *reinterpret_cast<const f32x4 *>(src + kg * 8 + 4);
      x_store(0, kg, lo, hi);
    }
    for (int i = tid; i < 2 * KGh * 64; i += LSTM_THREADS) reinterpret_cast<f32x4 *>(hbuf)[i] = f32x4{0, 0, 0, 0};
  }

  float bias[UB][4];
#pragma unroll
  for (int u = 0; u < UB; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[u][g] = a.bias[((wn * UB + u) * 4 + g) * 32 + (lane & 31)];

  f32x16 c[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[u][r] = 0.0f;

  __syncthreads();

  // weights of this wave: Wp[wn][u][kg][gate][256]
  const float *wbase = a.Wp + (size_t)wn * UB * KG * 1024 + lane * 4;

  for (int t = 0; t < T; ++t) {
    // prefetch the embedding rows of step t+1 into registers (one k-group per
    // thread covers E <= 64; wider embeddings loop below after the GEMM)
    const bool have_next = (t + 1) < T;
    int nid = 0;
    f32x4 nlo = {0, 0, 0, 0}, nhi = {0, 0, 0, 0};
    if (have_next) {
      nid = fetch_id(t + 1);
      if (xq < KGx) {
        const float *src = a.emb + (size_t)nid * a.Ep + xq * 8;
        nlo = *reinterpret_cast<const f32x4 *>(src);
        nhi = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    }

    const float *xa = xbuf + (size_t)(((t & 1) * 2 + wm) * KGx) * 256 + lane * 4;
    const float *ha = hbuf + (size_t)(wm * KGh) * 256 + lane * 4;

    if constexpr (TRAIN) {
      // A-tape for the weight-gradient GEMM: [x_t | h_{t-1}] of this tile, stored as
      // frag32 blocks with rows = k' (x: 0..63, h: 64 + unit) and reduction index
      // r = (t*NT32 + tile32)*32 + b, i.e. AT[(r/8)*KT + k'/32][256].
      const int KT = 2 + KGh / 4;
      const int nf4 = (64 + KGh * 8) * 16;  // float4s per step: k' count x 16 groups of 4 rows
      for (int i = tid; i < nf4; i += LSTM_THREADS) {
        const int kp = i % (64 + KGh * 8), b4 = i / (64 + KGh * 8);  // b4: rows 4*b4 .. 4*b4+3 of the 64
        const int mt = b4 >> 3, bl = (b4 & 7) * 4;
        f32x4 v = {0, 0, 0, 0};
        if (kp >= 64) {
          const int unit = kp - 64;
          const float *src = hbuf + (size_t)(mt * KGh + (unit >> 3)) * 256 + ((((unit >> 2) & 1) * 32 + bl) << 2) + (unit & 3);
          v = f32x4{src[0], src[4], src[8], src[12]};
        } else if (kp < KGx * 8) {
          const float *src = xbuf + (size_t)(((t & 1) * 2 + mt) * KGx + (kp >> 3)) * 256 + ((((kp >> 2) & 1) * 32 + bl) << 2) + (kp & 3);
          v = f32x4{src[0], src[4], src[8], src[12]};
        }
        const size_t rg = ((size_t)t * (gridDim.x * 2) + blockIdx.x * 2 + mt) * 4 + (bl >> 3);
        float *dst = a.tape_a + (rg * KT + (kp >> 5)) * 256 + ((((bl >> 2) & 1) * 32 + (kp & 31)) << 2);
        *reinterpret_cast<f32x4 *>(dst) = v;
      }
    }
    const int kend = (t == 0) ? KGx : KG;  // h_0 = 0: skip the recurrent part of step 0

    f32x16 hnew[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      f32x16 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = bias[u][g];

      const float *wp = wbase + (size_t)u * KG * 1024;
      f32x4 bcur[4], bnxt[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bcur[g] = *reinterpret_cast<const f32x4 *>(wp + g * 256);
      for (int kg = 0; kg < kend; ++kg) {
        const int kn = (kg + 1 < kend) ? kg + 1 : kg;
#pragma unroll
        for (int g = 0; g < 4; ++g) bnxt[g] = *reinterpret_cast<const f32x4 *>(wp + (size_t)kn * 1024 + g * 256);
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(kg < KGx ? xa + kg * 256 : ha + (kg - KGx) * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bcur[g][e], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) bcur[g] = bnxt[g];
      }
      // gates: acc[0]=i acc[1]=j acc[2]=f(+1 folded) acc[3]=o   (BasicLSTMCell, TF 1.x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float si = fast_sigmoid(acc[0][r]);
        const float tj = fast_tanh(acc[1][r]);
        const float sf = fast_sigmoid(acc[2][r]);
        const float so = fast_sigmoid(acc[3][r]);
        const float cn = c[u][r] * sf + si * tj;
        c[u][r] = cn;
        hnew[u][r] = fast_tanh(cn) * so;
        if constexpr (TRAIN) {
          // gate tape in accumulator layout: [t][tile32][wn][u][q][reg][lane], q = si,tj,sf,so,c
          float *tp = a.tape_g + ((((size_t)t * (gridDim.x * 2) + blockIdx.x * 2 + wm) * 4 + wn) * UB + u) * 5 * 1024 + r * 64 + lane;
          tp[0] = si;
          tp[1024] = tj;
          tp[2048] = sf;
          tp[3072] = so;
          tp[4096] = cn;
        }
      }
    }

    // stage x_{t+1} (its buffer was last read in step t-1)
    if (have_next) {
      if (xq < KGx) x_store((t + 1) & 1, xq, nlo, nhi);
      for (int kg = xq + 8; kg < KGx; kg += 8) {
        const float *src = a.emb + (size_t)nid * a.Ep + kg * 8;
        x_store((t + 1) & 1, kg, *reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
      }
    }
    __syncthreads();  // every wave is done reading h_{t-1}
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int unit = (wn * UB + u) * 32 + (lane & 31);
      float *dst = hbuf + (size_t)(wm * KGh + (unit >> 3)) * 256 + (unit & 3);
      const int half = (unit >> 2) & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(half * 32 + mfma_row(r, lane)) * 4] = hnew[u][r];
    }
    __syncthreads();  // h_t visible
  }

  if constexpr (TRAIN) {
    // h_T, row-major [Bp][Hp], for dM = h_T^T . d(out)
    const int Hp = KGh * 8;
    for (int i = tid; i < LSTM_BM * Hp; i += LSTM_THREADS) {
      const int unit = i % Hp, b = i / Hp;
      a.h_last[(size_t)(b0 + b) * Hp + unit] =
          hbuf[(size_t)((b >> 5) * KGh + (unit >> 3)) * 256 + ((((unit >> 2) & 1) * 32 + (b & 31)) << 2) + (unit & 3)];
    }
  }

  // --- projection  out = h_T . M   (+ optional l2_normalize), N tiles nt = wn, wn+4, ...
  constexpr int PT = 4;  // up to Sp = 512
  const float *ha = hbuf + (size_t)(wm * KGh) * 256 + lane * 4;
  f32x16 pacc[PT];
  float ss[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ss[r] = 0.0f;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + 4 * i;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
    if (nt < a.NTS) {
      const float *mp = a.Mp + (size_t)nt * KGh * 256 + lane * 4;
      for (int kg = 0; kg < KGh; ++kg) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ha + kg * 256);
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(mp + kg * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], pacc[i], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ss[r] += pacc[i][r] * pacc[i][r];
    }
  }
  float scale[16];
  if (a.normalize) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = ss[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if ((lane & 31) == 0) red[(wm * 32 + mfma_row(r, lane)) * 4 + wn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x4 p = *reinterpret_cast<const f32x4 *>(red + (wm * 32 + mfma_row(r, lane)) * 4);
      const float tot = (p[0] + p[1]) + (p[2] + p[3]);
      scale[r] = 1.0f / sqrtf(fmaxf(tot, 1e-12f));  // tf.nn.l2_normalize epsilon
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.0f;
  }
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + 4 * i;
    const int col = nt * 32 + (lane & 31);
    if (nt < a.NTS && col < a.S) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = b0 + wm * 32 + mfma_row(r, lane);
        if (row < a.B) a.out[(size_t)row * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

template <int UB, bool TRAIN>
static hipError_t launch_one(const LstmFwdArgs &a, size_t lds, dim3 grid, dim3 block, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_fwd_kernel<UB, TRAIN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((lstm_fwd_kernel<UB, TRAIN>), grid, block, lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_lstm_fwd(const LstmFwdArgs &a, int Hp, hipStream_t stream) {
  const size_t lds = lstm_fwd_lds_bytes(a.KGx, a.KGh);
  const dim3 grid((a.B + LSTM_BM - 1) / LSTM_BM), block(LSTM_THREADS);
  const bool train = a.tape_g != nullptr;
  if (Hp == 128) return train ? launch_one<1, true>(a, lds, grid, block, stream) : launch_one<1, false>(a, lds, grid, block, stream);
  if (Hp == 256) return train ? launch_one<2, true>(a, lds, grid, block, stream) : launch_one<2, false>(a, lds, grid, block, stream);
  return hipErrorInvalidValue;
}
