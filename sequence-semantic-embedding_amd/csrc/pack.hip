// Layout and elementwise helper kernels (HBM-bound): frag32 packing of row-major
// matrices, f64->f32 conversion, row L2-normalise (tf.nn.l2_normalize,
// sse_model.py:282-283).
#include "sse_kernels.h"

// rows [R][C] -> frag32 [RT][KG][256]; one thread per output float4
// (lane = half*32 + r, 4 consecutive k) -> coalesced 1 KiB stores per wave.
__global__ void pack_rows_kernel(const float *__restrict__ rows, int64_t R, int C, int KG, int64_t total4,
                                 f32x4 *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg = (int)(blk % KG);
    const int64_t rt = blk / KG;
    const int64_t r = rt * 32 + (l & 31);
    const int k0 = kg * 8 + (l >> 5) * 4;
    f32x4 v = {0, 0, 0, 0};
    if (r < R) {
      const float *src = rows + (size_t)r * C + k0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < C) v[e] = src[e];
    }
    out[i] = v;
  }
}

// Same re-layout for C % 4 == 0, at HBM speed: a wave moves 8 rows x 8 float4 columns per step -- loads are
// 128-byte row segments, and because a float4 of 4 consecutive k of one row stays a float4 in fragment order
// ((k%8)/4 picks the half, k/8 the block) the stores are 128-byte runs too; no LDS transpose.  A wave owns its 8
// rows for all columns, so the row norms needed for the scoring error bound (max over rows of sum x^2) come for free
// (norm_bits != nullptr: atomicMax on the float bits).
__global__ __launch_bounds__(256) void pack_rows_vec_kernel(const f32x4 *__restrict__ rows, int64_t R, int C4, int KG,
                                                            int64_t groups, f32x4 *__restrict__ out,
                                                            unsigned *__restrict__ norm_bits) {
  const int lane = threadIdx.x & 63, rs = lane >> 3, qq = lane & 7;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int Q4 = KG * 2;
  float best = 0.0f;
  for (int64_t g = wave0; g < groups; g += nwaves) {
    const int64_t r = g * 8 + rs;
    const int64_t rt = r >> 5;
    const int rl = (int)(r & 31);
    float ss = 0.0f;
#pragma unroll 8
    for (int q = qq; q < Q4; q += 8) {
      f32x4 v = {0, 0, 0, 0};
      if (r < R && q < C4) v = rows[(size_t)r * C4 + q];
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      out[((size_t)rt * KG + (q >> 1)) * 64 + (q & 1) * 32 + rl] = v;
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    best = fmaxf(best, ss);
  }
  if (norm_bits) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if (lane == 0) atomicMax(norm_bits, __float_as_uint(best));
  }
}

hipError_t launch_pack_rows_norm(const float *rows, int64_t R, int C, float *out, float *norm_bits, hipStream_t stream) {
  const int KG = (C + 7) / 8;
  const int64_t RT = (R + 31) / 32;
  if (RT == 0) return hipSuccess;
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {
    const int64_t groups = RT * 4;  // 8-row groups, padding rows of the last tile included (zero-filled)
    const int64_t blocks = (groups + 3) / 4;
    hipLaunchKernelGGL(pack_rows_vec_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4 *>(rows), R, C / 4, KG, groups, reinterpret_cast<f32x4 *>(out),
                       reinterpret_cast<unsigned *>(norm_bits));
    return hipGetLastError();
  }
  hipError_t e = launch_pack_rows(rows, R, C, out, stream);
  if (e != hipSuccess || !norm_bits) return e;
  return launch_row_norm2_max(rows, R, C, norm_bits, stream);
}

hipError_t launch_pack_rows(const float *rows, int64_t R, int C, float *out, hipStream_t stream) {
  const int KG = (C + 7) / 8;
  const int64_t RT = (R + 31) / 32;
  const int64_t total4 = RT * KG * 64;
  if (total4 == 0) return hipSuccess;
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) return launch_pack_rows_norm(rows, R, C, out, nullptr, stream);
  const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, stream, rows, R, C, KG, total4,
                     reinterpret_cast<f32x4 *>(out));
  return hipGetLastError();
}

__global__ void f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

hipError_t launch_f64_to_f32(const double *in, float *out, int64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(grid), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// one wave per row: x * rsqrt(max(sum(x^2), 1e-12))
__global__ void l2_normalize_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const float *src = x + (size_t)r * cols;
    float ss = 0.0f;
    for (int c = lane; c < cols; c += 64) ss += src[c] * src[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float sc = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int c = lane; c < cols; c += 64) out[(size_t)r * cols + c] = src[c] * sc;
  }
}

// cols % 4 == 0 and cols <= 1024: LPR = lanes per row (power of two, cols/4 rounded up, <= 64), 64/LPR rows per wave,
// NV float4 per lane; every element is read once (16-byte loads) and kept in registers.
template <int NV>
__global__ __launch_bounds__(256) void l2_normalize_vec_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ out,
                                                               int64_t rows, int C4, int LPR) {
  const int lane = threadIdx.x & 63, sub = lane & (LPR - 1), rsub = lane / LPR, RPW = 64 / LPR;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r0 = wave0 * RPW; r0 < rows; r0 += nwaves * RPW) {
    const int64_t r = r0 + rsub;
    f32x4 v[NV];
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = sub + j * LPR;
      v[j] = (r < rows && c < C4) ? x[(size_t)r * C4 + c] : f32x4{0, 0, 0, 0};
      ss += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
    }
    for (int o = LPR >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float sc = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = sub + j * LPR;
      if (r < rows && c < C4) out[(size_t)r * C4 + c] = f32x4{v[j][0] * sc, v[j][1] * sc, v[j][2] * sc, v[j][3] * sc};
    }
  }
}

hipError_t launch_l2_normalize(const float *x, float *out, int64_t rows, int cols, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  if ((cols & 3) == 0 && cols <= 1024 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const int C4 = cols / 4;
    int LPR = 1;
    while (LPR < C4 && LPR < 64) LPR <<= 1;
    const int NV = (C4 + LPR - 1) / LPR;  // 1 unless cols > 256
    const int64_t waves = (rows + 64 / LPR - 1) / (64 / LPR), blk = (waves + 3) / 4;
    const dim3 grid((int)(blk < 32768 ? blk : 32768));
    const f32x4 *xi = reinterpret_cast<const f32x4 *>(x);
    f32x4 *xo = reinterpret_cast<f32x4 *>(out);
    if (NV == 1) hipLaunchKernelGGL(l2_normalize_vec_kernel<1>, grid, dim3(256), 0, stream, xi, xo, rows, C4, LPR);
    else if (NV == 2) hipLaunchKernelGGL(l2_normalize_vec_kernel<2>, grid, dim3(256), 0, stream, xi, xo, rows, C4, LPR);
    else hipLaunchKernelGGL(l2_normalize_vec_kernel<4>, grid, dim3(256), 0, stream, xi, xo, rows, C4, LPR);
    return hipGetLastError();
  }
  const int64_t blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((int)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, x, out,
                     rows, cols);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Weight re-layouts for the LSTM forward kernel (run whenever variables change).

// BasicLSTMCell kernel [(E+H)][4H] (TF 1.x: rows [x | h], columns [i | j | f | o])
// -> Wp[wn][u][kg][gate][256]: for hidden-unit block ub = wn*UB + u the four gate
// tiles of one k-group are contiguous (4 KiB), k = [x padded to Ep | h padded to Hp].
// k-row E (the first padding row of the x part, Ep > E always) carries the BIAS: the padded embedding table
// holds a constant 1.0 in column E, so the gate GEMM adds it -- forget_bias = 1.0 folded into the f block
// (BasicLSTMCell adds it at run time, it is not stored in the variable).
__device__ __forceinline__ void pack_lstm_body(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int Ep,
                                               int Hp, int64_t total4, f32x4 *__restrict__ out) {
  const int KG = (Ep + Hp) / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    int64_t blk = i >> 6;
    const int g = (int)(blk & 3);
    blk >>= 2;
    const int kg = (int)(blk % KG);
    const int ub = (int)(blk / KG);  // = wn*UB + u
    const int unit = ub * 32 + (l & 31);
    f32x4 v = {0, 0, 0, 0};
    if (unit < H) {
      const int col = g * H + unit;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = kg * 8 + (l >> 5) * 4 + e;
        int row = -1;
        if (kk < Ep) {
          if (kk < E) row = kk;
          else if (kk == E) v[e] = b[col] + (g == 2 ? 1.0f : 0.0f);
        } else if (kk - Ep < H) {
          row = E + (kk - Ep);
        }
        if (row >= 0) v[e] = K[(size_t)row * 4 * H + col];
      }
    }
    out[i] = v;
  }
}
__global__ void pack_lstm_kernel_k(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int Ep,
                                   int Hp, int UB, int64_t total4, f32x4 *__restrict__ out) {
  pack_lstm_body(K, b, E, H, Ep, Hp, total4, out);
  (void)UB;
}

// X[K][N] row-major -> frag32 with rows = n (columns of X), k = rows of X, zero padded to KGp groups
__device__ __forceinline__ void pack_kn_body(const float *__restrict__ X, int K, int N, int KGp, int64_t total4,
                                             f32x4 *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg = (int)(blk % KGp);
    const int nt = (int)(blk / KGp);
    const int n = nt * 32 + (l & 31);
    f32x4 v = {0, 0, 0, 0};
    if (n < N) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kg * 8 + (l >> 5) * 4 + e;
        if (k < K) v[e] = X[(size_t)k * N + n];
      }
    }
    out[i] = v;
  }
}

__global__ void pack_kn_kernel_k(const float *__restrict__ X, int K, int N, int KGp, int64_t total4,
                                 f32x4 *__restrict__ out) {
  pack_kn_body(X, K, N, KGp, total4, out);
}

// rows [R][C] -> [R][Cp] zero padded; one_col >= 0: that padding column holds 1.0 (the LSTM bias rides on it)
__device__ __forceinline__ void pad_rows_body(const float *__restrict__ in, int64_t R, int C, int Cp, int one_col, float *__restrict__ out) {
  const int64_t total = R * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const int64_t r = i / Cp;
    out[i] = (c < C) ? in[r * C + c] : (c == one_col ? 1.0f : 0.0f);
  }
}
__global__ void pad_rows_kernel_k(const float *__restrict__ in, int64_t R, int C, int Cp, int one_col, float *__restrict__ out) {
  pad_rows_body(in, R, C, Cp, one_col, out);
}

static inline int grid_for(int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

hipError_t launch_pack_lstm(const float *K, const float *b, int E, int H, int Ep, int Hp, int UB, float *Wp,
                            hipStream_t stream) {
  if (Ep <= E) return hipErrorInvalidValue;  // the bias needs a padding k-row
  const int KG = (Ep + Hp) / 8;
  const int64_t total4 = (int64_t)(Hp / 32) * KG * 4 * 64;
  hipLaunchKernelGGL(pack_lstm_kernel_k, dim3(grid_for(total4)), dim3(256), 0, stream, K, b, E, H, Ep, Hp, UB, total4,
                     reinterpret_cast<f32x4 *>(Wp));
  return hipGetLastError();
}

hipError_t launch_pack_kn(const float *X, int K, int N, int KGp, float *out, hipStream_t stream) {
  const int64_t total4 = (int64_t)((N + 31) / 32) * KGp * 64;
  hipLaunchKernelGGL(pack_kn_kernel_k, dim3(grid_for(total4)), dim3(256), 0, stream, X, K, N, KGp, total4,
                     reinterpret_cast<f32x4 *>(out));
  return hipGetLastError();
}

hipError_t launch_pad_rows(const float *in, int64_t R, int C, int Cp, int one_col, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(pad_rows_kernel_k, dim3(grid_for(R * Cp)), dim3(256), 0, stream, in, R, C, Cp, one_col, out);
  return hipGetLastError();
}

// max over rows of sum(x^2) (float bits are order-preserving for non-negative values)
__global__ void row_norm2_max_kernel_k(const float *__restrict__ x, int64_t rows, int cols, unsigned *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float best = 0.0f;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    float ss = 0.0f;
    for (int c = lane; c < cols; c += 64) ss += x[(size_t)r * cols + c] * x[(size_t)r * cols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    best = fmaxf(best, ss);
  }
  if (lane == 0) atomicMax(out, __float_as_uint(best));
}

hipError_t launch_row_norm2_max(const float *x, int64_t rows, int cols, float *out_bits, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const int64_t blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(row_norm2_max_kernel_k, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, x, rows,
                     cols, reinterpret_cast<unsigned *>(out_bits));
  return hipGetLastError();
}

__global__ void fill_kernel_k(float *p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t launch_fill(float *p, int64_t n, float v, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_kernel_k, dim3(grid_for(n)), dim3(256), 0, stream, p, n, v);
  return hipGetLastError();
}

// Transposed slices of the LSTM kernel for the backward GEMMs:
// out = frag32(rows = i in [0, RT*32), red = n in [0, 4*Hp)),
// value = K[row0 + i][g*H + unit] for i < nrows, unit < H, with n = g*Hp + unit (gate-major, padded)
__device__ __forceinline__ void pack_kT_body(const float *__restrict__ K, int row0, int nrows, int H, int Hp, int64_t total4,
                                             f32x4 *__restrict__ out) {
  const int KGn = Hp / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg = (int)(blk % KGn);
    const int rt = (int)(blk / KGn);
    const int row = rt * 32 + (l & 31);
    f32x4 v = {0, 0, 0, 0};
    if (row < nrows) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = kg * 8 + (l >> 5) * 4 + e;
        const int g = n / Hp, unit = n % Hp;
        if (unit < H) v[e] = K[(size_t)(row0 + row) * 4 * H + g * H + unit];
      }
    }
    out[i] = v;
  }
}

__global__ void pack_kT_kernel_k(const float *__restrict__ K, int row0, int nrows, int H, int Hp, int64_t total4,
                                 f32x4 *__restrict__ out) {
  pack_kT_body(K, row0, nrows, H, Hp, total4, out);
}

// Every layout a train step derives from the variables in ONE launch (a step used to end in 9 - 13 pack launches of 2 - 8 us
// with ~5 us between them: 0.1 ms on the critical path between the optimizer and the next forward).  All workgroups walk all
// jobs grid-stride; the jobs are independent.
__global__ __launch_bounds__(256) void pack_multi_kernel(PackJobs jobs) {
  for (int j = 0; j < jobs.n; ++j) {
    const PackJob &q = jobs.job[j];
    switch (q.type) {
      case PACK_JOB_PAD_ROWS: pad_rows_body(q.a, q.total, q.i0, q.i1, q.i2, q.out); break;
      case PACK_JOB_LSTM: pack_lstm_body(q.a, q.b, q.i0, q.i1, q.i2, q.i3, q.total, reinterpret_cast<f32x4 *>(q.out)); break;
      case PACK_JOB_KN: pack_kn_body(q.a, q.i0, q.i1, q.i2, q.total, reinterpret_cast<f32x4 *>(q.out)); break;
      case PACK_JOB_KT: pack_kT_body(q.a, q.i0, q.i1, q.i2, q.i3, q.total, reinterpret_cast<f32x4 *>(q.out)); break;
      default: break;
    }
  }
}

void pack_job_pad_rows(PackJobs &js, const float *in, int64_t R, int C, int Cp, int one_col, float *out) {
  js.job[js.n++] = PackJob{PACK_JOB_PAD_ROWS, in, nullptr, out, R, C, Cp, one_col, 0};
}
void pack_job_lstm(PackJobs &js, const float *K, const float *b, int E, int H, int Ep, int Hp, float *Wp) {
  js.job[js.n++] = PackJob{PACK_JOB_LSTM, K, b, Wp, (int64_t)(Hp / 32) * ((Ep + Hp) / 8) * 4 * 64, E, H, Ep, Hp};
}
void pack_job_kn(PackJobs &js, const float *X, int K, int N, int KGp, float *out) {
  js.job[js.n++] = PackJob{PACK_JOB_KN, X, nullptr, out, (int64_t)((N + 31) / 32) * KGp * 64, K, N, KGp, 0};
}
void pack_job_kT(PackJobs &js, const float *K, int row0, int nrows, int RT, int H, int Hp, float *out) {
  js.job[js.n++] = PackJob{PACK_JOB_KT, K, nullptr, out, (int64_t)RT * (Hp / 2) * 64, row0, nrows, H, Hp};
}
hipError_t launch_pack_multi(const PackJobs &js, hipStream_t stream) {
  if (js.n == 0) return hipSuccess;
  int64_t mx = 0;
  for (int j = 0; j < js.n; ++j) {
    const int64_t items = js.job[j].type == PACK_JOB_PAD_ROWS ? js.job[j].total * js.job[j].i1 : js.job[j].total;
    mx = items > mx ? items : mx;
  }
  hipLaunchKernelGGL(pack_multi_kernel, dim3(grid_for(mx) < 2048 ? grid_for(mx) : 2048), dim3(256), 0, stream, js);
  return hipGetLastError();
}

hipError_t launch_pack_kT(const float *K, int row0, int nrows, int RT, int H, int Hp, float *out, hipStream_t stream) {
  const int64_t total4 = (int64_t)RT * (Hp / 2) * 64;
  hipLaunchKernelGGL(pack_kT_kernel_k, dim3(grid_for(total4)), dim3(256), 0, stream, K, row0, nrows, H, Hp, total4,
                     reinterpret_cast<f32x4 *>(out));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// PAD-prefix bucketing of a device-resident id matrix (sse_encode_dev; sse_index.py:79-85 left-pads every row).  What the
// host-buffer entry point does with a counting sort on the host (sse_api.hip, encode_host_ids_locked) for ids that are already
// in HBM: the row numbers ordered by leading-PAD count so that every row tile of the matrix kernel can skip its whole common
// prefix -- SHORTEST prefix first: the tiles with the most steps left are dispatched first and the short ones fill the tail
// (longest-first left the T-step tiles for last: real crosslingual queries 0.85 -> 1.04 ms, profiles/r06_notes.txt).  Two launches, no host round trip:
//   pad_lead_kernel   one thread per row (walks its row to the first token), lead[b] kept, a histogram over lead
//                     through LDS; the LAST workgroup to finish (ticket counter) turns the histogram into bucket starts,
//                     re-zeroes histogram + ticket for the next call, and stores the batch's statistics to `stat`
//   pad_scatter_kernel  order[start[lead[b]]++] = b  (the order inside a bucket is whatever the atomics yield: every
//                     row's result is independent of its tile mates, bit for bit -- tests/test_gpu_encode.py)
// hist: [T + 2] int32 zero on entry (and on exit) | ticket: one int32, same | start: [T + 2] | stat (pinned host memory, may be
// null): {call number, 0 dense / 1 some padding / 2 mean prefix >= T / 4}.
__global__ void __launch_bounds__(256) pad_lead_kernel(const int32_t *__restrict__ ids, int B, int T, int32_t *__restrict__ lead,
                                                        int32_t *hist, int32_t *ticket, int32_t *__restrict__ start,
                                                        volatile int32_t *stat, int32_t seq) {
  extern __shared__ int32_t lh[];  // [T + 2]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < T + 2; i += 256) lh[i] = 0;
  __syncthreads();
  // one THREAD per row (second version).  The first had one wavefront per row on up to 1024 workgroups: a dense 16384-row batch
  // then paid 1024 same-address histogram atomics + 1024 ticket atomics, 69 us (rocprofv3, profiles/r06_notes.txt).  A lane walks
  // its own row until the first token: consecutive words of one cache line, at most T iterations of a wave whatever the mix;
  // the wave's rows are counted into the LDS histogram one DISTINCT prefix length at a time (one LDS atomic per length per wave).
  const int b = blockIdx.x * 256 + tid;
  int l = -1;
  if (b < B) {
    const int32_t *row = ids + (size_t)b * T;
    l = 0;
    while (l < T && row[l] == 0) ++l;
    lead[b] = l;
  }
  unsigned long long todo = __ballot(l >= 0);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int lv = __shfl(l, first);
    const unsigned long long same = __ballot(l == lv) & todo;
    if (lane == first) atomicAdd(&lh[lv + 1], __popcll(same));  // bucket lv (no padding = bucket 0), shifted by one for the exclusive scan
    todo &= ~same;
  }
  __syncthreads();
  for (int i = tid; i < T + 2; i += 256)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
  __threadfence();
  __shared__ int last;
  __syncthreads();
  if (tid == 0) last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last workgroup: exclusive scan of hist[0 .. T+1] (hist[0] is 0) -> start[], statistics, and the zero invariant back
  for (int i = tid; i < T + 2; i += 256) lh[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (w == 0) {
    int carry = 0;
    long long lead_sum = 0;
    for (int i0 = 0; i0 < T + 2; i0 += 64) {
      const int i = i0 + lane;
      const int v = i < T + 2 ? lh[i] : 0;
      if (i >= 1 && i < T + 2) lead_sum += (long long)v * (i - 1);  // rows of bucket i-1 have lead = i-1
      int s = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(s, o);
        if (lane >= o) s += u;
      }
      if (i < T + 2) start[i] = carry + s;  // inclusive over the shifted histogram = exclusive bucket starts
      carry += __shfl(s, 63);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lead_sum += __shfl_xor(lead_sum, o);
    if (lane == 0 && stat) {
      stat[1] = lead_sum == 0 ? 0 : (lead_sum * 4 >= (long long)B * T ? 2 : 1);
      __threadfence_system();
      stat[0] = seq;
    }
  }
  for (int i = tid; i < T + 2; i += 256) hist[i] = 0;
  if (tid == 0) *ticket = 0;
}

// The cheap check for batches expected to be dense (the latest completed call of that side saw no padding): one thread per
// row looks at the row's FIRST token only; nothing is written unless a padded row shows up -- no atomics, no second launch
// (the two-launch bucketing cost the dense 16384-row headline batch 0.07 ms of 2.4, profiles/r06_notes.txt).
__global__ void __launch_bounds__(256) pad_detect_kernel(const int32_t *__restrict__ ids, int B, int T, volatile int32_t *stat,
                                                          int32_t seq) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const bool padded = b < B && ids[(size_t)b * T] == 0;
  if (__ballot(padded) != 0 && (threadIdx.x & 63) == 0) {  // (racing waves store the same two words)
    stat[1] = 1;
    __threadfence_system();
    stat[0] = seq;
  }
}

// (wave-aggregated: the rows of a wave that share a prefix length take their slots with ONE atomic -- real batches have a few
// heavy buckets, a dense one a single bucket: 16384 same-address atomics took 188 us, rocprofv3, profiles/r06_notes.txt)
__global__ void __launch_bounds__(256) pad_scatter_kernel(const int32_t *__restrict__ lead, int B, int T, int32_t *start,
                                                           int32_t *__restrict__ order) {
  const int b = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int l = b < B ? lead[b] : -1;
  unsigned long long todo = __ballot(l >= 0);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const int lv = __shfl(l, first);
    const unsigned long long same = __ballot(l == lv) & todo;
    int base = 0;
    if (lane == first) base = atomicAdd(&start[lv], __popcll(same));
    base = __shfl(base, first);
    if (l == lv) order[base + __popcll(same & ((1ull << lane) - 1ull))] = b;
    todo &= ~same;
  }
}

// zeroed: int32 [SSE_PAD_SORT_MAX_T + 3] = hist [T + 2] ... | ticket (last word) -- zero on first use (the caller memsets the
// allocation once), zero again when the launch has finished, for any T.  work: int32 [T + 2 + B] = start | lead.
// scatter = false: pad_detect_kernel only (the caller saw dense batches and wants them re-checked).
size_t pad_sort_zeroed_words() { return (size_t)SSE_PAD_SORT_MAX_T + 3; }
size_t pad_sort_work_words(int B, int T) { return (size_t)(T + 2) + (size_t)B; }
hipError_t launch_pad_sort(const int32_t *ids, int B, int T, int32_t *zeroed, int32_t *work, int32_t *order, int32_t *stat_pinned,
                           int32_t seq, bool scatter, hipStream_t stream) {
  if (T > SSE_PAD_SORT_MAX_T) return hipErrorInvalidValue;
  if (!scatter) {
    hipLaunchKernelGGL(pad_detect_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ids, B, T, stat_pinned, seq);
    return hipGetLastError();
  }
  int32_t *hist = zeroed, *ticket = zeroed + SSE_PAD_SORT_MAX_T + 2, *start = work, *lead = work + (T + 2);
  const int grid = (B + 255) / 256;
  hipLaunchKernelGGL(pad_lead_kernel, dim3(grid), dim3(256), (size_t)(T + 2) * sizeof(int32_t), stream, ids, B, T, lead, hist,
                     ticket, start, stat_pinned, seq);
  hipLaunchKernelGGL(pad_scatter_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, lead, B, T, start, order);
  return hipGetLastError();
}
