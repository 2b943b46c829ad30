// The LSTM encoder for ANY shape (gfx950): forward with tapes, BPTT, weight gradients -- the path behind the fast kernels.
//
// The reference builds its graph for whatever --src_cell_size / --tgt_cell_size / --embedding_size / --encoding_size the
// user passes (sse_train.py:60-74; sse_model.py:113-126,236-275).  The fused kernels of this library are laid out for cell
// sizes <= 512 (training <= 256), embeddings that fit their LDS tile (training <= 64 columns) and encodings <= 512; up to
// round 4 anything else was rejected with an error.  This file removes the rejection: the same
// arithmetic -- BasicLSTMCell over static_rnn, zero initial state, all T steps (sse_model.py:240-242); i, j, f, o gate order,
// forget_bias 1.0 added at run time -- as a per-step sequence of plain kernels:
//     forward   x part of A gathered for all steps at once;  per step:  G_t = A_t . K  (fp32 MFMA GEMM),  gates -> c_t, h_t
//               (h_t lands in the h columns of A_{t+1}), tapes si, sf, so, tj, tc, c_{t-1};
//     backward  per step (T-1 .. 0):  dG_t from (dh_t, dc_t, tape_t);  dA_t = dG_t . K^T  (GEMM);  dX_t = x columns of dA_t
//               -> scatter-add into d word_embedding (+ sum of squares of the raw slices), dh_{t-1} = h columns;
//               then  dK = A^T dG  over all T * rows at once (GEMM with the batch as the reduction), db = column sums of dG.
// Two launches per step and direction: latency-bound (a 128-row step at T = 80 is ~650 launches) -- correctness first; every
// shape the fused kernels accept keeps them.  GEMMs run on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation).
// The oracle these kernels are checked against is the same as the fused path's (tests/test_gpu_generic.py).
#include "sse_kernels.h"
#include "train.h"

namespace {

// C[m][n] (+)= sum_k A[m][k] * B[n][k]: both operands row-major along the reduction (lda, ldb, K multiples of 4 floats ->
// 16-byte aligned float4 reads; K % 8 == 0).  Lane (row, k half) reads one float4 of its row at k0 + 4 * half; component
// u of the two halves is the k pair of the u-th MFMA (both operands permute k identically).  One 32 x 32 tile per wave.
__global__ __launch_bounds__(256) void gen_gemm_nt_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                         float *__restrict__ Cm, int ldc, int M, int N, int K) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mt = blockIdx.y, nt = blockIdx.x * 4 + w;
  if (nt * 32 >= N) return;  // (no barriers in this kernel)
  const int row = lane & 31, kh = lane >> 5;
  const int m = mt * 32 + row, n = nt * 32 + row;
  const bool mok = m < M, nok = n < N;
  const float *pa = A + (size_t)(mok ? m : 0) * lda + 4 * kh;
  const float *pb = B + (size_t)(nok ? n : 0) * ldb + 4 * kh;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  constexpr int U = 4;
  int k0 = 0;
  for (; k0 + 8 * U <= K; k0 += 8 * U) {
    f32x4 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = *reinterpret_cast<const f32x4 *>(pa + k0 + 8 * u);
      bv[u] = *reinterpret_cast<const f32x4 *>(pb + k0 + 8 * u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mok ? av[u][e] : 0.0f, nok ? bv[u][e] : 0.0f, acc, 0, 0, 0);
  }
  for (; k0 < K; k0 += 8) {
    const f32x4 av = *reinterpret_cast<const f32x4 *>(pa + k0), bv = *reinterpret_cast<const f32x4 *>(pb + k0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mok ? av[e] : 0.0f, nok ? bv[e] : 0.0f, acc, 0, 0, 0);
  }
  const int nc = nt * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = mt * 32 + mfma_row(r, lane);
    if (mr < M && nc < N) Cm[(size_t)mr * ldc + nc] = acc[r];
  }
}

// part[chunk][m][n] = sum over the chunk's rows r of A[r][m] * B[r][n]: the ROW index is the reduction, both operands read as
// they lie in memory (lane (column, k half) takes one float of row r + k half: coalesced 128-byte row segments).
__global__ __launch_bounds__(256) void gen_gemm_tn_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                         float *__restrict__ part, int R, int M, int N, int chunk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mt = blockIdx.y, nt = blockIdx.x * 4 + w;
  if (nt * 32 >= N) return;
  const int r0 = blockIdx.z * chunk, r1 = min(R, r0 + chunk);
  const int m = mt * 32 + (lane & 31), n = nt * 32 + (lane & 31), kk = lane >> 5;
  const bool mok = m < M, nok = n < N;
  const float *pa = A + (size_t)(r0 + kk) * lda + (mok ? m : 0);
  const float *pb = B + (size_t)(r0 + kk) * ldb + (nok ? n : 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  constexpr int U = 8;  // chunk and R are multiples of 16
  for (int r = r0; r < r1; r += 2 * U) {
    float av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = pa[(size_t)(2 * u) * lda];
      bv[u] = pb[(size_t)(2 * u) * ldb];
    }
    pa += (size_t)2 * U * lda;
    pb += (size_t)2 * U * ldb;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mok ? av[u] : 0.0f, nok ? bv[u] : 0.0f, acc, 0, 0, 0);
  }
  float *out = part + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = mt * 32 + mfma_row(r, lane);
    if (mr < M && nok) out[(size_t)mr * N + n] = acc[r];
  }
}

// dK[k][g*H + u] (+)= sum_chunks part[c][k][g*Hq + u]  (k < E + H);  db[g*H + u] (+)= sum_chunks part[c][Kp - 1 ...]: see below
__global__ void gen_dk_reduce_kernel(const float *__restrict__ part, int nch, int Kp, int Nq, int EH, int H, int Hq, int accumulate,
                                     float *__restrict__ dK) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)EH * 4 * H) return;
  const int k = (int)(i / (4 * H)), n = (int)(i % (4 * H)), g = n / H, u = n % H;
  float acc = accumulate ? dK[i] : 0.0f;
  for (int c = 0; c < nch; ++c) acc += part[((size_t)c * Kp + k) * Nq + g * Hq + u];
  dK[i] = acc;
}

// db[g*H + u] (+)= sum over all rows of dG[r][g*Hq + u]   (one workgroup per 64 columns, fixed order: deterministic)
__global__ __launch_bounds__(256) void gen_db_kernel(const float *__restrict__ dG, int64_t R, int Nq, int H, int Hq, int accumulate,
                                                     float *__restrict__ db) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float acc = 0.0f;
  if (col < Nq)
    for (int64_t r = part; r < R; r += 4) acc += dG[r * Nq + col];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && col < Nq) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int g = col / Hq, u = col % Hq;
    if (u < H) db[g * H + u] = (accumulate ? db[g * H + u] : 0.0f) + v;
  }
}

// K [E+H][4H] -> KT [4Hq][Kp] (KT[g*Hq + u][k] = K[k][g*H + u]) and Kq [Kp][4Hq] (zero padding)
__global__ void gen_pack_kernel(const float *__restrict__ K, int EH, int H, int Hq, int Kp, float *__restrict__ KT, float *__restrict__ Kq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)4 * Hq * Kp;
  if (i >= total) return;
  {
    const int n = (int)(i / Kp), k = (int)(i % Kp), g = n / Hq, u = n % Hq;
    KT[i] = (k < EH && u < H) ? K[(size_t)k * 4 * H + g * H + u] : 0.0f;
  }
  {
    const int k = (int)(i / (4 * Hq)), n = (int)(i % (4 * Hq)), g = n / Hq, u = n % Hq;
    Kq[i] = (k < EH && u < H) ? K[(size_t)k * 4 * H + g * H + u] : 0.0f;
  }
}

// M [H][S] -> MT [S][Hq] (zero padded k)
__global__ void gen_pack_proj_kernel(const float *__restrict__ Mv, int H, int Hq, int S, float *__restrict__ MT) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * Hq) return;
  const int s = (int)(i / Hq), j = (int)(i % Hq);
  MT[i] = j < H ? Mv[(size_t)j * S + s] : 0.0f;
}

// A[t][b][0..E) = emb[ids[b][t]][0..E) for all steps; the h columns of step 0 and every padding column = 0
__global__ void gen_gather_x_kernel(const int32_t *__restrict__ ids, const float *__restrict__ emb, int B, int Bp, int T, int V, int E,
                                    int Kp_h /* E + Hq */, int Kp, float *__restrict__ A, int32_t *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)T * Bp * Kp;
  if (i >= total) return;
  const int k = (int)(i % Kp);
  const int64_t tb = i / Kp;
  const int b = (int)(tb % Bp), t = (int)(tb / Bp);
  float v = 0.0f;
  if (k < E && b < B) {
    int id = ids[(size_t)b * T + t];
    if (id < 0 || id >= V) {
      atomicOr(err, 1);
      id = 0;
    }
    v = emb[(size_t)id * E + k];
  } else if (k >= E && k < Kp_h && t > 0) {
    return;  // h columns of later steps: written by the gates kernel of the step before (the pad columns behind them: 0)
  }
  A[i] = v;
}

__device__ __forceinline__ float gen_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates of one step: G [Bp][4Hq] pre-activations (no bias yet) -> c, h; tape (TRAIN): si, sf, so, tj, tc, c_prev
struct GenGatesArgs {
  const float *G;      // [Bp][4Hq]
  const float *bias;   // [4H]
  float *c;            // [Bp][Hq] in / out
  float *h_next;       // h columns of the next step's A rows (ld = ldh), or the h_last buffer at the last step
  float *tape;         // TRAIN: [6][Bp][Hq] of this step, or null
  int32_t Bp, H, Hq, ldh;
};
__global__ void gen_gates_fwd_kernel(GenGatesArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.Bp * a.Hq) return;
  const int b = (int)(i / a.Hq), u = (int)(i % a.Hq);
  float cn = 0.0f, hn = 0.0f, si = 0.0f, sf = 0.0f, so = 0.0f, tj = 0.0f, tc = 0.0f;
  const float cp = a.c[i];
  if (u < a.H) {
    const float *g = a.G + (size_t)b * 4 * a.Hq + u;
    si = gen_sigmoid(g[0] + a.bias[u]);
    tj = tanhf(g[a.Hq] + a.bias[a.H + u]);
    sf = gen_sigmoid(g[2 * a.Hq] + a.bias[2 * a.H + u] + 1.0f);  // forget_bias = 1.0 (BasicLSTMCell default, sse_model.py:240)
    so = gen_sigmoid(g[3 * a.Hq] + a.bias[3 * a.H + u]);
    cn = cp * sf + si * tj;
    tc = tanhf(cn);
    hn = tc * so;
  }
  a.c[i] = cn;
  a.h_next[(size_t)b * a.ldh + u] = hn;
  if (a.tape) {
    const size_t n = (size_t)a.Bp * a.Hq;
    a.tape[i] = si;
    a.tape[n + i] = sf;
    a.tape[2 * n + i] = so;
    a.tape[3 * n + i] = tj;
    a.tape[4 * n + i] = tc;
    a.tape[5 * n + i] = cp;
  }
}

// BPTT of one step (oracle: _lstm_backward): dh, dc, tape -> dG [Bp][4Hq] (i, j, f, o), dc <- dc_total * sf
struct GenGatesBwdArgs {
  const float *dh;     // [Bp][ldh] (h columns of dA of the step after, or dh_last)
  float *dc;           // [Bp][Hq] in / out
  const float *tape;   // [6][Bp][Hq]
  float *dG;           // [Bp][4Hq]
  int32_t Bp, H, Hq, ldh;
};
__global__ void gen_gates_bwd_kernel(GenGatesBwdArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.Bp * a.Hq) return;
  const int b = (int)(i / a.Hq), u = (int)(i % a.Hq);
  float di = 0.0f, dj = 0.0f, df = 0.0f, dO = 0.0f, dcn = 0.0f;
  if (u < a.H) {
    const size_t n = (size_t)a.Bp * a.Hq;
    const float si = a.tape[i], sf = a.tape[n + i], so = a.tape[2 * n + i], tj = a.tape[3 * n + i], tc = a.tape[4 * n + i],
                cp = a.tape[5 * n + i];
    const float dh = a.dh[(size_t)b * a.ldh + u];
    const float dct = a.dc[i] + dh * so * (1.0f - tc * tc);
    dO = dh * tc * so * (1.0f - so);
    di = dct * tj * si * (1.0f - si);
    dj = dct * si * (1.0f - tj * tj);
    df = dct * cp * sf * (1.0f - sf);
    dcn = dct * sf;
  }
  a.dc[i] = dcn;
  float *g = a.dG + (size_t)b * 4 * a.Hq + u;
  g[0] = di;
  g[a.Hq] = dj;
  g[2 * a.Hq] = df;
  g[3 * a.Hq] = dO;
}

// dX_t = x columns of dA_t: one wave per row -- scatter-add into the dense d word_embedding, sum of squares of the raw slice
// (a token id outside [0, V) adds nothing: the forward raised the error flag for it and the update is cancelled on the device)
__global__ void gen_dx_scatter_kernel(const float *__restrict__ dA, int lda, const int32_t *__restrict__ ids, int t, int T, int B, int E,
                                      int V, float *__restrict__ d_emb, float *__restrict__ sq) {
  const int lane = threadIdx.x & 63, b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= B) return;
  const int id = ids[(size_t)b * T + t];
  const bool ok = id >= 0 && id < V;
  float acc = 0.0f;
  for (int e = lane; e < E; e += 64) {
    const float v = dA[(size_t)b * lda + e];
    acc += v * v;
    if (ok && v != 0.0f) atomicAdd(d_emb + (size_t)id * E + e, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) sq[(size_t)t * B + b] = acc;
}

void gemm_nt(const float *A, int lda, const float *B, int ldb, float *Cm, int ldc, int M, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL(gen_gemm_nt_kernel, dim3(((N + 31) / 32 + 3) / 4, (M + 31) / 32), dim3(256), 0, st, A, lda, B, ldb, Cm, ldc, M, N, K);
}

}  // namespace

GenLstmDims gen_lstm_dims(int B, int T, int E, int H) {
  GenLstmDims d;
  d.B = B;
  d.Bp = round_up(B, 32);
  d.T = T;
  d.E = E;
  d.H = H;
  d.Hq = round_up(H, 8);
  d.Kp = round_up(E + d.Hq, 8);  // [x (E) | h (Hq) | pad]: the h columns start at E
  return d;
}
size_t gen_lstm_a_floats(const GenLstmDims &d) { return (size_t)d.T * d.Bp * d.Kp; }
size_t gen_lstm_tape_floats(const GenLstmDims &d) { return (size_t)d.T * 6 * d.Bp * d.Hq; }
size_t gen_lstm_dg_floats(const GenLstmDims &d) { return (size_t)d.T * d.Bp * 4 * d.Hq; }
size_t gen_lstm_kt_floats(const GenLstmDims &d) { return (size_t)4 * d.Hq * d.Kp; }
int gen_lstm_dk_chunks(const GenLstmDims &d) {
  const int64_t R = (int64_t)d.T * d.Bp;
  int n = (int)((R + 2047) / 2048);
  return n < 1 ? 1 : n > 64 ? 64 : n;
}
size_t gen_lstm_dk_part_floats(const GenLstmDims &d) { return (size_t)gen_lstm_dk_chunks(d) * d.Kp * 4 * d.Hq; }

hipError_t launch_gen_pack(const float *K, const float *Mv, const GenLstmDims &d, int S, float *KT, float *Kq, float *MT, hipStream_t st) {
  const int64_t total = (int64_t)4 * d.Hq * d.Kp;
  // (the packed K rows are [x | h]: row k of the master is column k of A as long as the h columns start at E: E + u <-> row E + u)
  hipLaunchKernelGGL(gen_pack_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, K, d.E + d.H, d.H, d.Hq, d.Kp, KT, Kq);
  if (Mv && MT) hipLaunchKernelGGL(gen_pack_proj_kernel, dim3((int)(((int64_t)S * d.Hq + 255) / 256)), dim3(256), 0, st, Mv, d.H, d.Hq, S, MT);
  return hipGetLastError();
}

// forward: ids [B][T] -> h_last [Bp][Hq]; tape != null: training (tapes + A kept for the backward pass)
hipError_t launch_gen_forward(const int32_t *ids, const float *emb, int V, const float *KT, const float *bias, const GenLstmDims &d,
                              float *A, float *G, float *c, float *tape, float *h_last, int32_t *err, hipStream_t st) {
  const int64_t na = (int64_t)d.T * d.Bp * d.Kp, nc = (int64_t)d.Bp * d.Hq;
  hipLaunchKernelGGL(gen_gather_x_kernel, dim3((int)((na + 255) / 256)), dim3(256), 0, st, ids, emb, d.B, d.Bp, d.T, V, d.E, d.E + d.Hq, d.Kp, A, err);
  hipError_t e = hipMemsetAsync(c, 0, (size_t)nc * sizeof(float), st);
  if (e != hipSuccess) return e;
  for (int t = 0; t < d.T; ++t) {
    const float *At = A + (size_t)t * d.Bp * d.Kp;
    gemm_nt(At, d.Kp, KT, d.Kp, G, 4 * d.Hq, d.Bp, 4 * d.Hq, d.Kp, st);
    GenGatesArgs ga;
    ga.G = G;
    ga.bias = bias;
    ga.c = c;
    const bool last = t + 1 == d.T;
    ga.h_next = last ? h_last : A + (size_t)(t + 1) * d.Bp * d.Kp + d.E;
    ga.ldh = last ? d.Hq : d.Kp;
    ga.tape = tape ? tape + (size_t)t * 6 * d.Bp * d.Hq : nullptr;
    ga.Bp = d.Bp;
    ga.H = d.H;
    ga.Hq = d.Hq;
    hipLaunchKernelGGL(gen_gates_fwd_kernel, dim3((int)((nc + 255) / 256)), dim3(256), 0, st, ga);
  }
  return hipGetLastError();
}

// raw [Bp][S] = h_last [Bp][Hq] . M  (MT = M^T [S][Hq])
hipError_t launch_gen_project(const float *h_last, const float *MT, const GenLstmDims &d, int S, float *raw, hipStream_t st) {
  gemm_nt(h_last, d.Hq, MT, d.Hq, raw, S, d.Bp, S, d.Hq, st);
  return hipGetLastError();
}

// backward: dh_last [Bp][ldh_last] -> dG for all steps, d word_embedding (+ sq[t*B + b]), then dK, db
hipError_t launch_gen_backward(const int32_t *ids, const float *Kq, const GenLstmDims &d, const float *A, const float *tape,
                               const float *dh_last, int ldh_last, float *dG, float *dA, float *dc, float *dk_part, int accumulate,
                               float *dK, float *db, float *d_emb, float *sq, int V, hipStream_t st) {
  const int64_t nc = (int64_t)d.Bp * d.Hq;
  hipError_t e = hipMemsetAsync(dc, 0, (size_t)nc * sizeof(float), st);
  if (e != hipSuccess) return e;
  for (int t = d.T - 1; t >= 0; --t) {
    GenGatesBwdArgs ba;
    const bool last = t + 1 == d.T;
    ba.dh = last ? dh_last : dA + d.E;
    ba.ldh = last ? ldh_last : d.Kp;
    ba.dc = dc;
    ba.tape = tape + (size_t)t * 6 * d.Bp * d.Hq;
    ba.dG = dG + (size_t)t * d.Bp * 4 * d.Hq;
    ba.Bp = d.Bp;
    ba.H = d.H;
    ba.Hq = d.Hq;
    hipLaunchKernelGGL(gen_gates_bwd_kernel, dim3((int)((nc + 255) / 256)), dim3(256), 0, st, ba);
    // dA_t [Bp][Kp] = dG_t [Bp][4Hq] . Kq^T   (Kq [Kp][4Hq]: both along the reduction)
    gemm_nt(ba.dG, 4 * d.Hq, Kq, 4 * d.Hq, dA, d.Kp, d.Bp, d.Kp, 4 * d.Hq, st);
    hipLaunchKernelGGL(gen_dx_scatter_kernel, dim3((d.B + 3) / 4), dim3(256), 0, st, dA, d.Kp, ids, t, d.T, d.B, d.E, V, d_emb, sq);
  }
  // dK = A^T dG over all T * Bp rows (the batch and the steps are the reduction); db = column sums of dG
  const int64_t R = (int64_t)d.T * d.Bp;
  const int nch = gen_lstm_dk_chunks(d);
  const int chunk = (int)(((R + nch - 1) / nch + 15) / 16 * 16);
  hipLaunchKernelGGL(gen_gemm_tn_kernel, dim3(((4 * d.Hq + 31) / 32 + 3) / 4, (d.Kp + 31) / 32, nch), dim3(256), 0, st, A, d.Kp, dG, 4 * d.Hq,
                     dk_part, (int)R, d.Kp, 4 * d.Hq, chunk);
  const int64_t nk = (int64_t)(d.E + d.H) * 4 * d.H;
  // (columns E .. E+H of A are h: rows E .. E+H of K -- the same index)
  hipLaunchKernelGGL(gen_dk_reduce_kernel, dim3((int)((nk + 255) / 256)), dim3(256), 0, st, dk_part, nch, d.Kp, 4 * d.Hq, d.E + d.H, d.H,
                     d.Hq, accumulate, dK);
  hipLaunchKernelGGL(gen_db_kernel, dim3((4 * d.Hq + 63) / 64), dim3(256), 0, st, dG, R, 4 * d.Hq, d.H, d.Hq, accumulate, db);
  return hipGetLastError();
}
