// Training kernels for gfx950: pairwise cosine loss, BPTT through the LSTM
// encoders, gradient global norm + clip, Adagrad.  Together with the TRAIN
// instantiation of lstm_fwd_kernel they replace
//   session.run([model.train, model.loss, model.train_acc], feed)   (sse_train.py:170-172)
// i.e. sse_model.py:279-302 (_def_loss) and :355-364 (_def_optimize:
// tf.gradients -> clip_by_global_norm(5.0) -> AdagradOptimizer.apply_gradients).
//
// Tapes written by the forward pass (layouts in sse_kernels.h / lstm_fwd.hip):
//   tape_g  gate activations + c per step, accumulator layout (lane-private)
//   tape_a  [x_t | h_{t-1}]  as frag32(rows = k', red = r)
// Buffers produced here:
//   dg_a    d(pre-activation gates) as frag32(rows = b, red = n): A operand of
//           the recurrent GEMM dh = dg . Kh^T and of dX = dg . Kx^T
//   dg_b    the same values as frag32(rows = n, red = r): B operand of
//           dK = A^T . dG
#include <cstdio>
#include <cstdlib>

#include "train.h"

#include "sse_kernels.h"


__device__ __forceinline__ float fast_tanh_t(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }
__device__ __forceinline__ float fast_sigmoid_t(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }

// ---------------------------------------------------------------------------
// Loss: one wave per pair row.  sse_model.py:282-283,290,298,302.
struct LossArgs {
  const float *src_raw, *tgt_raw;  // [B][S] un-normalised encodings
  const float *labels;             // [B]
  float *d_src, *d_tgt;            // [Bp][S] gradients w.r.t. the raw encodings (rows >= B zeroed)
  float *row_loss, *row_acc;       // [B]
  int32_t B, Bp, S;
  float inv_rows;                  // 1 / rows of the (global) batch: the loss is a mean over all ranks' rows
};

__global__ void loss_kernel(LossArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= a.Bp) return;
  float *ds = a.d_src + (size_t)row * a.S, *dt = a.d_tgt + (size_t)row * a.S;
  if (row >= a.B) {
    for (int d = lane; d < a.S; d += 64) {
      ds[d] = 0.0f;
      dt[d] = 0.0f;
    }
    return;
  }
  const float *s = a.src_raw + (size_t)row * a.S, *t = a.tgt_raw + (size_t)row * a.S;
  float ss = 0.0f, tt = 0.0f, st = 0.0f;
  for (int d = lane; d < a.S; d += 64) {
    const float x = s[d], y = t[d];
    ss += x * x;
    tt += y * y;
    st += x * y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o);
    tt += __shfl_xor(tt, o);
    st += __shfl_xor(st, o);
  }
  const float rs = 1.0f / sqrtf(fmaxf(ss, 1e-12f)), rt = 1.0f / sqrtf(fmaxf(tt, 1e-12f));
  const float cosv = st * rs * rt;           // reduce_sum(ns * nt)
  const float x = 64.0f * cosv;
  const float z = a.labels[row];
  const float sg = fast_sigmoid_t(x);
  if (lane == 0) {
    // weighted_cross_entropy_with_logits, pos_weight = 1
    a.row_loss[row] = (1.0f - z) * x + log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);
    a.row_acc[row] = z * floorf(sg + 0.1f) + (1.0f - z) * floorf(1.1f - sg);
  }
  const float dcos = 64.0f * (sg - z) * a.inv_rows;
  // l2_normalize backward: d raw = r * (dn - n * (n . dn)), dn_s = dcos * nt; the max(., eps)
  // clamp passes no gradient to the norm when sum(x^2) < eps
  const bool cs = ss < 1e-12f, ct = tt < 1e-12f;
  for (int d = lane; d < a.S; d += 64) {
    const float ns = s[d] * rs, nt = t[d] * rt;
    ds[d] = rs * dcos * (nt - (cs ? 0.0f : ns * cosv));
    dt[d] = rt * dcos * (ns - (ct ? 0.0f : nt * cosv));
  }
}

// sum over rows / rows_global, fixed order (deterministic); out[0] = loss, out[1] = train_acc
__global__ void loss_reduce_kernel(const float *row_loss, const float *row_acc, int B, float inv_rows, float *out) {
  __shared__ float sl[256], sa[256];
  float l = 0.0f, c = 0.0f;
  for (int i = threadIdx.x; i < B; i += 256) {
    l += row_loss[i];
    c += row_acc[i];
  }
  sl[threadIdx.x] = l;
  sa[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sl[threadIdx.x] += sl[threadIdx.x + o];
      sa[threadIdx.x] += sa[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sl[0] * inv_rows;
    out[1] = sa[0] * inv_rows;
  }
}

// ---------------------------------------------------------------------------
// Projection backward (two small GEMMs, LDS-tiled 64x64 output tiles, 4x4 per thread):
//   dM[j][s] = sum_b hT[b][j] * d[b][s]     (reduction over the batch, split into chunks ->
//                                            per-chunk partials, summed in fixed order)
//   dh[b][j] = sum_s d[b][s] * M[j][s]
#define PB_TILE 64
#define PB_K 16

__global__ __launch_bounds__(256) void proj_bwd_dm_kernel(const float *__restrict__ hT, const float *__restrict__ d,
                                                          int Bp, int H, int Hp, int S, int chunk, float *__restrict__ part) {
  __shared__ float As[PB_K][PB_TILE + 4], Bs[PB_K][PB_TILE + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j0 = blockIdx.y * PB_TILE, s0 = blockIdx.x * PB_TILE;
  const int b_begin = blockIdx.z * chunk, b_end = min(Bp, b_begin + chunk);
  float acc[4][4] = {};
  for (int b0 = b_begin; b0 < b_end; b0 += PB_K) {
    for (int i = threadIdx.x; i < PB_K * PB_TILE; i += 256) {
      const int kk = i / PB_TILE, c = i % PB_TILE, b = b0 + kk;
      As[kk][c] = (b < b_end && j0 + c < H) ? hT[(size_t)b * Hp + j0 + c] : 0.0f;
      Bs[kk][c] = (b < b_end && s0 + c < S) ? d[(size_t)b * S + s0 + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < PB_K; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = As[kk][ty * 4 + u];
        bv[u] = Bs[kk][tx * 4 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
    }
    __syncthreads();
  }
  float *out = part + (size_t)blockIdx.z * H * S;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + ty * 4 + u, sidx = s0 + tx * 4 + v;
      if (j < H && sidx < S) out[(size_t)j * S + sidx] = acc[u][v];
    }
}

__global__ void proj_bwd_dm_reduce_kernel(const float *part, int nchunks, int n, float *dM) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
  for (int c = 0; c < nchunks; ++c) acc += part[(size_t)c * n + i];
  dM[i] = acc;
}

// The same partials on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation): the batch is
// the reduction dimension, so both operands are read as they lie in memory -- lane (column, k half) takes one float of
// row b + k half, a coalesced 128-byte row segment per half wave -- no LDS.  Workgroup = one 32-unit tile x four 32-column
// tiles (a wave each; the h fragment is the same for the four waves and comes from L1), grid.z = batch chunks.
// (The VALU kernel above took 0.11 ms per encoder at 8192 rows, a third of a forward pass on the bf16 pipe.)
__global__ __launch_bounds__(256) void proj_bwd_dm_mfma_kernel(const float *__restrict__ hT, const float *__restrict__ d, int Bp,
                                                               int H, int Hp, int S, int chunk, float *__restrict__ part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int jt = blockIdx.y, st = blockIdx.x * 4 + w;
  if (st * 32 >= S) return;  // (no barriers in this kernel)
  const int b0 = blockIdx.z * chunk, b1 = min(Bp, b0 + chunk);
  const int j = jt * 32 + (lane & 31), sc = st * 32 + (lane & 31), kk = lane >> 5;
  const bool jok = j < Hp, sok = sc < S;
  const float *pa = hT + (size_t)(b0 + kk) * Hp + (jok ? j : 0);
  const float *pb = d + (size_t)(b0 + kk) * S + (sok ? sc : 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  constexpr int U = 8;  // 8 row pairs (16 loads) in flight; chunk and Bp are multiples of 16
  for (int b = b0; b < b1; b += 2 * U) {
    float av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = pa[(size_t)(2 * u) * Hp];
      bv[u] = pb[(size_t)(2 * u) * S];
    }
    pa += (size_t)2 * U * Hp;
    pb += (size_t)2 * U * S;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(jok ? av[u] : 0.0f, sok ? bv[u] : 0.0f, acc, 0, 0, 0);
  }
  float *out = part + (size_t)blockIdx.z * H * S;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int jr = jt * 32 + mfma_row(r, lane);
    if (jr < H && sok) out[(size_t)jr * S + sc] = acc[r];
  }
}

__global__ __launch_bounds__(256) void proj_bwd_dh_kernel(const float *__restrict__ d, const float *__restrict__ M, int Bp,
                                                          int H, int Hp, int S, float *__restrict__ dh) {
  __shared__ float As[PB_K][PB_TILE + 4], Bs[PB_K][PB_TILE + 4];  // [s][b] and [s][j]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int b0 = blockIdx.y * PB_TILE, j0 = blockIdx.x * PB_TILE;
  float acc[4][4] = {};
  for (int s0 = 0; s0 < S; s0 += PB_K) {
    for (int i = threadIdx.x; i < PB_K * PB_TILE; i += 256) {
      const int r = i / PB_K, kk = i % PB_K;  // 16 consecutive s of one row: 64-byte segments
      As[kk][r] = (b0 + r < Bp && s0 + kk < S) ? d[(size_t)(b0 + r) * S + s0 + kk] : 0.0f;
      Bs[kk][r] = (j0 + r < H && s0 + kk < S) ? M[(size_t)(j0 + r) * S + s0 + kk] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < PB_K; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = As[kk][ty * 4 + u];
        bv[u] = Bs[kk][tx * 4 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int b = b0 + ty * 4 + u, j = j0 + tx * 4 + v;
      if (b < Bp && j < Hp) dh[(size_t)b * Hp + j] = acc[u][v];  // j >= H: weights read as 0 -> dh = 0
    }
}

// ---------------------------------------------------------------------------
// Sequential BPTT over one 32-row tile: 4 waves, wave wn owns hidden units
// [wn*UB*32, (wn+1)*UB*32) exactly as in the forward kernel, so the gate tape
// is read back lane-privately and dc / dh stay in registers across steps.
struct LstmBwdArgs {
  const float *tape_g;  // forward gate tape
  const float *dh_last; // [Bp][Hp]
  const float *KhT;     // frag32(rows = j (Hp), red = n (4Hp)):  Kh^T
  float *dg_a;          // [T][NT32][KGn][256]
  float *dg_b;          // [(T*NT32*4)][NTn][256]
  float *db_part;       // [NT32][4*Hp] per-tile bias-gradient partials
  int32_t T, NT32, Hp;
  int32_t H;            // real cell size: dG columns of padded units are exactly 0, their k-groups are skipped
  const unsigned short *KhT16;  // X3: Kh^T as split frag16 blocks [Hp/32][4Hp/16][hi|lo][512] (launch_pack_kT16)
  int32_t dg_b_split;   // 1: dg_b is written as split bf16 frag16 blocks [(T*NT32*2)][NTn][hi|lo][512] (same bytes) for the
                        // dK GEMM on the bf16 matrix pipe (dk_x3_kernel)
  int32_t NT_tape;      // 32-row tiles the gate tape holds per step: NT32, or NT32/2 when the batch is (pos, neg) pairs
                        // that share their source sequence -- tiles j and j + NT_tape then read the same tape
  // X3: dX_t = dG_t . Kx^T is taken from the dG tile while it sits in LDS (no dg_a dump, no dx_kernel) and scattered
  // into the dense embedding gradient here
  const unsigned short *KxT16;  // [4 e-tiles of 16][4Hp/32][hi|lo][512]: B fragments of v_mfma_f32_16x16x32_bf16 (launch_pack_kxT16)
  const int32_t *ids;           // [B][T]
  float *d_emb;                 // [V][E], zero-initialised
  float *sq_part;               // [NT32][NW] sum of dx^2 over OCCURRENCES (tf.global_norm sees IndexedSlices.values raw)
  float *hot_part;              // [T*NT32*2][2][64] per (step, tile, row half) sums for the ids 0 (PAD) and 1 (EOS)
  int32_t B, E, V;
  long long *clk;               // -DSSE_BWD_CLOCK builds only: per-wave phase cycle sums of tile 0 ([wave][8])
};

#ifndef BWD_ROT  // measurement builds override these (tools/)
#define BWD_ROT 1
#endif
#ifndef BWD_NT   // cache policy of the tape loads: 2 = nt (streamed once; the 1.25 MiB of weights every step re-reads should
#define BWD_NT 2 // stay in the XCD's 4 MiB L2 instead of being flushed by ~9 MiB of tape / dump traffic per step)
#endif
#ifdef SSE_BWD_CLOCK  // measurement builds (tools/): cycles per phase of the BPTT step, summed over the steps
#define BWD_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define BWD_CLK(i)                 \
  {                                \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;           \
    ck_t = n_;                     \
  }
#else
#define BWD_CLK_DECL
#define BWD_CLK(i)
#endif

// X3 (with SPLIT): the dG tile lives in LDS as split bf16 frag16 blocks [4Hp/16][hi|lo][1 KiB] (lane (row, half) owns 8
// consecutive n; the same bytes), written by 2-byte scatters, and the recurrent GEMM runs as three
// v_mfma_f32_32x32x16_bf16 per 16 n (see sse_kernels.h) against Kh^T in the same form.
template <int UB, int NW, bool SPLIT, bool X3>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_kernel(LstmBwdArgs a) {
  static_assert(!X3 || SPLIT, "the split-operand BPTT feeds the split-operand dK GEMM");
  extern __shared__ __attribute__((aligned(16))) float dgs[];  // [KGn][256]: dg tile, frag32(rows = b, red = n)
  unsigned char *dgb = reinterpret_cast<unsigned char *>(dgs);
  constexpr int NTHR = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR): the tape descriptor depends on it
  constexpr int Hp = 32 * UB * NW;  // (= a.Hp: the launcher picks the instantiation by it) compile-time, so that the tile
                                    // offsets below are instruction immediates instead of address registers
  constexpr int KGn = Hp / 2, NTn = Hp / 8;
  const int tile = blockIdx.x, T = a.T;
  const int half = lane >> 5;

  f32x16 dh[UB], dc[UB];
  float dbacc[UB][4];
  // gate tape of the step about to be processed, held in registers and refilled (for step t-1) while the
  // recurrent GEMM of step t runs: the HBM latency of the tape never sits on the critical path
  float tg[UB][4][16], tcn[UB][16], tcp[UB][16];
  // tape reads go through a buffer descriptor: address = SGPR base + SGPR offset + (16*lane), so the 20
  // loads of a refill cost no address VGPRs (per-lane 64-bit pointers spilled this kernel)
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.tape_g + ((size_t)(tile % a.NT_tape) * NW + wn) * UB * 5 * 1024), 0, 0x7fffffff, 0x00020000);
  const int tvo = lane * 16;  // 16-byte pieces: registers 4q .. 4q+3 of a quantity are one buffer_load_dwordx4 (20 per refill)
  const int tstep = a.NT_tape * NW * UB * 5 * 1024 * 4;  // bytes between consecutive steps (T*tstep < 2^31 checked by the launcher)
  auto tld4 = [&](int t, int u, int qty, int q4) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, tvo, t * tstep + (u * 5 * 1024 + qty * 1024 + q4 * 256) * 4, BWD_NT));
  };
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int unit = (wn * UB + u) * 32 + (lane & 31);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 v[6];
#pragma unroll
      for (int g = 0; g < 5; ++g) v[g] = tld4(T - 1, u, g, q4);
      v[5] = tld4(T > 1 ? T - 2 : 0, u, 4, q4);  // unused when T == 1
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = q4 * 4 + e;
        dh[u][r] = a.dh_last[(size_t)(tile * 32 + mfma_row(r, lane)) * Hp + unit];
        dc[u][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) tg[u][g][r] = v[g][e];
        tcn[u][r] = v[4][e];
        tcp[u][r] = v[5][e];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) dbacc[u][g] = 0.0f;
  }

  float xsq = 0.0f;  // X3: this lane's share of sum(dx^2) over all steps
  BWD_CLK_DECL
  for (int t = T - 1; t >= 0; --t) {
    // ---- elementwise gate backward, results into the LDS dg tile
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int unit = (wn * UB + u) * 32 + (lane & 31);
      float gq[4][2];  // X3: the even row's values, waiting for the odd row
      (void)gq;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float si = tg[u][0][r], tj = tg[u][1][r], sf = tg[u][2][r], so = tg[u][3][r];
        const float cn = tcn[u][r];
        const float cprev = (t > 0) ? tcp[u][r] : 0.0f;
        const float tc = fast_tanh_t(cn);
        const float dhv = dh[u][r];
        const float dov = dhv * tc;
        const float dcv = dc[u][r] + dhv * so * (1.0f - tc * tc);
        const float g_i = dcv * tj * si * (1.0f - si);
        const float g_j = dcv * si * (1.0f - tj * tj);
        const float g_f = dcv * cprev * sf * (1.0f - sf);
        const float g_o = dov * so * (1.0f - so);
        dc[u][r] = dcv * sf;
        dbacc[u][0] += g_i;
        dbacc[u][1] += g_j;
        dbacc[u][2] += g_f;
        dbacc[u][3] += g_o;
        const int b = mfma_row(r, lane);
        if constexpr (X3) {
          // element (b, n = g*Hp + unit) -> group n/16, slot ((n/8)&1)*32 + b, piece n%8; hi block, lo block 1 KiB on.
          // Registers r, r+1 (rows b, b+1) are split together on the hardware converter (v_cvt_pk_bf16_f32) and leave as
          // the low / high half-words of the packed results
          gq[0][r & 1] = g_i;
          gq[1][r & 1] = g_j;
          gq[2][r & 1] = g_f;
          gq[3][r & 1] = g_o;
          if (r & 1) {
            unsigned char *dst = dgb + (size_t)(unit >> 4) * 2048 + (size_t)((((unit >> 3) & 1) * 32 + (b - 1)) * 16) + (unit & 7) * 2;
            constexpr int GS = (Hp / 16) * 2048;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
              unsigned hi, lo;
              sse_split2(gq[gi][0], gq[gi][1], hi, lo);
              *reinterpret_cast<unsigned short *>(dst + gi * GS) = (unsigned short)hi;
              *reinterpret_cast<unsigned short *>(dst + gi * GS + 16) = (unsigned short)(hi >> 16);
              *reinterpret_cast<unsigned short *>(dst + gi * GS + 1024) = (unsigned short)lo;
              *reinterpret_cast<unsigned short *>(dst + gi * GS + 1024 + 16) = (unsigned short)(lo >> 16);
            }
          }
        } else {
          // element (b, n = g*Hp + unit) -> dgs[n/8][((n%8)/4*32 + b)*4 + n%4]; Hp % 8 == 0 so n%8 == unit%8
          float *dst = dgs + (size_t)(unit >> 3) * 256 + ((((unit >> 2) & 1) * 32 + b) << 2) + (unit & 3);
          dst[(size_t)(0 * Hp / 8) * 256] = g_i;
          dst[(size_t)(1 * Hp / 8) * 256] = g_j;
          dst[(size_t)(2 * Hp / 8) * 256] = g_f;
          dst[(size_t)(3 * Hp / 8) * 256] = g_o;
        }
        if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bound the interleaving to a pair of rows (register pressure: 0 B scratch)
      }
    }
    // refill the tape registers for step t-1 (c_{t-1} is already here: it was this step's c_prev); the fence
    // keeps the scheduler from hoisting these loads above the last use of the old values (two live copies spill).
    // Requested AFTER the recurrent GEMM and the dX product (in flight under the dump and the wait at the barrier): with
    // the tape registers pending the GEMM had 16 registers for its operand ring, spilled, and waited for L2 on every
    // k-group (clock64: 27 - 35 k cycles per step for 6 k cycles of MFMA); now both loops keep 8 groups of weights in flight.
    auto refill = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      // unconditional (the last step re-reads step 0 for nothing): behind an `if (t > 0)` the old values stay live through
      // the GEMM for the not-taken path and cost 64 registers
      const int tp = t > 0 ? t - 1 : 0, tpp = t > 1 ? t - 2 : 0;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          f32x4 v[5];
#pragma unroll
          for (int g = 0; g < 4; ++g) v[g] = tld4(tp, u, g, q4);
          v[4] = tld4(tpp, u, 4, q4);  // step 0 ignores it (c_{-1} = 0)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = q4 * 4 + e;
#pragma unroll
            for (int g = 0; g < 4; ++g) tg[u][g][r] = v[g][e];
            tcn[u][r] = tcp[u][r];
            tcp[u][r] = v[4][e];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    BWD_CLK(0)
    __syncthreads();
    BWD_CLK(1)

    // ---- dump the tile: linear copy (A-operand layout) and transposed (B-operand layout).  Both the dump
    // and the recurrent GEMM only READ the LDS tile, so the two waves that share a SIMD run them in
    // opposite order (waves < NW/2: dump then GEMM; the others: GEMM then dump): one wave's global stores
    // overlap its partner's MFMA stream.  Each group dumps its half of the tile.
    auto dump = [&]() {
      constexpr int GTHR = NTHR / 2;
      int td = tid;
      asm volatile("" : "+v"(td));  // opaque copy: dump addresses are recomputed per step (register pressure)
      const int grp = (wn >= NW / 2) ? 1 : 0, gt = td - grp * GTHR;
      if constexpr (X3) {
        // dg_b for the dK GEMM: the stored hi / lo pieces regrouped by row octets (dX is computed below from the tile
        // itself: no A-operand copy leaves the workgroup)
        unsigned short *gb = reinterpret_cast<unsigned short *>(a.dg_b);
        const size_t g0 = ((size_t)t * a.NT32 + tile) * 2;
        const int nb = 4 * Hp * 4, hb = nb / 2;
#pragma unroll 1
        for (int i = grp * hb + gt; i < (grp + 1) * hb; i += GTHR) {
          const int n = i % (4 * Hp), oc = i / (4 * Hp);  // rows 8*oc .. 8*oc+7
          const unsigned char *src = dgb + (size_t)(n >> 4) * 2048 + (size_t)((((n >> 3) & 1) * 32 + oc * 8) * 16) + (n & 7) * 2;
          sse_u32x4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = (unsigned)*reinterpret_cast<const unsigned short *>(src + (2 * e) * 16) |
                    ((unsigned)*reinterpret_cast<const unsigned short *>(src + (2 * e + 1) * 16) << 16);
            lo[e] = (unsigned)*reinterpret_cast<const unsigned short *>(src + 1024 + (2 * e) * 16) |
                    ((unsigned)*reinterpret_cast<const unsigned short *>(src + 1024 + (2 * e + 1) * 16) << 16);
          }
          unsigned short *dst = gb + (((g0 + (oc >> 1)) * NTn + (n >> 5)) * 2) * 512 + ((oc & 1) * 32 + (n & 31)) * 8;
          __builtin_nontemporal_store(hi, reinterpret_cast<sse_u32x4 *>(dst));  // streamed once: do not displace the weights in L2
          __builtin_nontemporal_store(lo, reinterpret_cast<sse_u32x4 *>(dst + 512));
        }
        return;
      }
      f32x4 *ga = reinterpret_cast<f32x4 *>(a.dg_a + ((size_t)t * a.NT32 + tile) * KGn * 256);
      const f32x4 *ls = reinterpret_cast<const f32x4 *>(dgs);
      const int n4 = KGn * 64, h4 = n4 / 2;
#pragma unroll 4
      for (int i = grp * h4 + gt; i < (grp + 1) * h4; i += GTHR) ga[i] = ls[i];
      if constexpr (SPLIT) {
        // (n, 8 consecutive rows) -> one hi and one lo octet of block (16-row group, n/32)
        unsigned short *gb = reinterpret_cast<unsigned short *>(a.dg_b);
        const size_t g0 = ((size_t)t * a.NT32 + tile) * 2;
        const int nb = 4 * Hp * 4, hb = nb / 2;
#pragma unroll 1
        for (int i = grp * hb + gt; i < (grp + 1) * hb; i += GTHR) {
          const int n = i % (4 * Hp), oc = i / (4 * Hp);  // rows 8*oc .. 8*oc+7
          const float *src = dgs + (size_t)(n >> 3) * 256 + ((((n >> 2) & 1) * 32 + oc * 8) << 2) + (n & 3);
          float v8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] = src[4 * e];
          sse_u32x4 hi, lo;
          sse_split8(v8, hi, lo);
          unsigned short *dst = gb + (((g0 + (oc >> 1)) * NTn + (n >> 5)) * 2) * 512 + ((oc & 1) * 32 + (n & 31)) * 8;
          __builtin_nontemporal_store(hi, reinterpret_cast<sse_u32x4 *>(dst));  // streamed once: do not displace the weights in L2
          __builtin_nontemporal_store(lo, reinterpret_cast<sse_u32x4 *>(dst + 512));
        }
      } else {
        // (n, 4 consecutive rows) -> one float4 of block (rg, n/32)
        const size_t rg0 = ((size_t)t * a.NT32 + tile) * 4;
        const int nb = 4 * Hp * 8, hb = nb / 2;
#pragma unroll 4
        for (int i = grp * hb + gt; i < (grp + 1) * hb; i += GTHR) {
          const int n = i % (4 * Hp), b4 = i / (4 * Hp);  // rows 4*b4 .. 4*b4+3
          const int bl = b4 * 4;
          const float *src = dgs + (size_t)(n >> 3) * 256 + ((((n >> 2) & 1) * 32 + bl) << 2) + (n & 3);
          const f32x4 v = {src[0], src[4], src[8], src[12]};
          float *dst = a.dg_b + ((rg0 + (bl >> 3)) * NTn + (n >> 5)) * 256 + ((((bl >> 2) & 1) * 32 + (n & 31)) << 2);
          *reinterpret_cast<f32x4 *>(dst) = v;
        }
      }
    };
    const bool dump_first = wn < NW / 2;
    if (dump_first) dump();
    BWD_CLK(2)

    // ---- recurrent GEMM: dh_{t-1}[b][j] = sum_n dg[b][n] * Kh[j][n]
    if (X3 && t > 0) {
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[u][r] = 0.0f;
      typedef short bw_bf16x8 __attribute__((ext_vector_type(8)));
      int lg = lane;
      asm volatile("" : "+v"(lg));
      const unsigned char *la = dgb + lg * 16;
      const int KG16 = 4 * Hp / 16, KGg = Hp / 16, KGl = min(KGg, (a.H + 15) / 16), NL = 4 * KGl;  // live groups of 16 n
      // Kh^T fragments through a buffer descriptor (SGPR base + SGPR offset + 16 * lane): no per-lane 64-bit pointers
      const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<unsigned short *>(a.KhT16) + (size_t)(wn * UB) * KG16 * 1024, 0, UB * KG16 * 2048, 0x00020000);
      // The walk over the 4 x KGl live k-groups starts at a different group per wave and tile and wraps around.  It is
      // kept as (gate base, group in gate) counters that advance by compare-and-select: `i / KGl` per k-group cost ~60
      // scalar instructions an iteration and its branches made the compiler drain the operand ring (vmcnt(0)).
      struct Walk {
        int l, base;
      };
      const int rot = BWD_ROT ? (wn * (NL / NW) + tile * 3) % NL : 0;
      const Walk w0{rot % KGl, (rot / KGl) * KGg};
      auto adv = [&](Walk &w) {
        const int l1 = w.l + 1;
        const bool wrap = l1 == KGl;
        const int b1 = w.base + KGg;
        w.base = wrap ? (b1 == 4 * KGg ? 0 : b1) : w.base;
        w.l = wrap ? 0 : l1;
      };
      constexpr int PF = 4;  // Kh^T groups in flight per wave; NL = 4 * KGl is a multiple of it (no tail)
      bw_bf16x8 bq[PF][UB][2], aq[2][2];
      auto bld = [&](int u, int g16, int hl) -> bw_bf16x8 {
        return __builtin_bit_cast(bw_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(krs, lg * 16, ((u * KG16 + g16) * 2 + hl) * 1024, 0));
      };
      Walk wb = w0, wa = w0;  // wb: next group to request from L2, wa: next dG fragment to read from LDS
#pragma unroll
      for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          bq[p][u][0] = bld(u, wb.base + wb.l, 0);
          bq[p][u][1] = bld(u, wb.base + wb.l, 1);
        }
        adv(wb);
      }
      aq[0][0] = *reinterpret_cast<const bw_bf16x8 *>(la + (size_t)(wa.base + wa.l) * 2048);
      aq[0][1] = *reinterpret_cast<const bw_bf16x8 *>(la + (size_t)(wa.base + wa.l) * 2048 + 1024);
      adv(wa);
      __builtin_amdgcn_s_setprio(1);
      for (int kg = 0; kg < NL; kg += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          // (past the end both walks wrap to the start: valid addresses, values unused)
          aq[(p + 1) & 1][0] = *reinterpret_cast<const bw_bf16x8 *>(la + (size_t)(wa.base + wa.l) * 2048);
          aq[(p + 1) & 1][1] = *reinterpret_cast<const bw_bf16x8 *>(la + (size_t)(wa.base + wa.l) * 2048 + 1024);
          adv(wa);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            dh[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[p & 1][1], bq[p][u][0], dh[u], 0, 0, 0);  // dg_lo * K_hi
            dh[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[p & 1][0], bq[p][u][1], dh[u], 0, 0, 0);  // dg_hi * K_lo
            dh[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[p & 1][0], bq[p][u][0], dh[u], 0, 0, 0);  // dg_hi * K_hi
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            bq[p][u][0] = bld(u, wb.base + wb.l, 0);
            bq[p][u][1] = bld(u, wb.base + wb.l, 1);
          }
          adv(wb);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (!X3 && t > 0) {
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[u][r] = 0.0f;
      int lg = lane;
      asm volatile("" : "+v"(lg));  // opaque copy: these addresses are recomputed per step (register pressure)
      const float *la = dgs + lg * 4;
      // Kh^T fragments through a buffer descriptor (SGPR base + SGPR offset + 16 * lane): no per-lane 64-bit pointers
      const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(a.KhT) + (size_t)(wn * UB) * KGn * 256, 0, UB * KGn * 1024, 0x00020000);
      auto kld = [&](int u, int kg) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, lg * 16, (u * KGn + kg) * 1024, 0));
      };
      // Kh^T fragments come from L2: PF k-groups (PF*4*UB MFMAs) of them in flight in a register ring; the dg fragments
      // come from LDS one k-group ahead.  Reduction index n = gate*Hp + unit: only the k-groups of units < H can be
      // non-zero -> a division-free walk over 4 gates x KGl live k-groups (NL = 4 * KGl is a multiple of PF), fixed
      // order: the exact path keeps one summation order for every tile
      constexpr int PF = 4;  // (the tape registers are refilled after this loop: room for a deeper ring than round 2's 2)
      f32x4 bq[PF][UB], aq[2];
      const int KGg = Hp / 8, KGl = min(KGg, (a.H + 7) / 8), NL = 4 * KGl;
      struct Walk {
        int l, base;
      };
      auto adv = [&](Walk &w) {
        const int l1 = w.l + 1;
        const bool wrap = l1 == KGl;
        const int b1 = w.base + KGg;
        w.base = wrap ? (b1 == 4 * KGg ? 0 : b1) : w.base;
        w.l = wrap ? 0 : l1;
      };
      Walk wb{0, 0}, wa{0, 0};
#pragma unroll
      for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int u = 0; u < UB; ++u) bq[p][u] = kld(u, wb.base + wb.l);
        adv(wb);
      }
      aq[0] = *reinterpret_cast<const f32x4 *>(la);
      adv(wa);
      __builtin_amdgcn_s_setprio(1);
      for (int kg = 0; kg < NL; kg += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          aq[(p + 1) & 1] = *reinterpret_cast<const f32x4 *>(la + (wa.base + wa.l) * 256);  // (wraps to group 0 past the end: unused)
          adv(wa);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < UB; ++u) dh[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[p & 1][e], bq[p][u][e], dh[u], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UB; ++u) bq[p][u] = kld(u, wb.base + wb.l);
          adv(wb);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    BWD_CLK(3)
    if constexpr (X3) {
      // ---- dX_t[b][e] = sum_n dG[b][n] * Kx[e][n] on v_mfma_f32_16x16x32_bf16 (three per product, split operands):
      // 2 row halves x 4 e-tiles of 16 = 8 output tiles, wave = (row half wn & 1, e-tile(s) wn >> 1 (+ NW/2)); the A
      // fragment of lane (row l & 15, n octet q = l >> 4) is the 16-byte slot (group 2 ks + (q >> 1), half q & 1, row)
      // of the dG tile, B comes from L2 (KxT16, 256 KiB per encoder)
      typedef short bx_bf16x8 __attribute__((ext_vector_type(8)));
      constexpr int NE = 8 / NW;
      const int rh = wn & 1, et0 = wn >> 1;
      int lx = lane;
      asm volatile("" : "+v"(lx));  // opaque copy: the addresses below are recomputed every step, not kept across the phases
      const int KS = 4 * Hp / 32, KSg = Hp / 32, KSl = min(KSg, (a.H + 31) / 32), NS = 4 * KSl;  // live 32-n steps
      // (staggered, division-free walk over the 4 x KSl live 32-n steps, as in the recurrent GEMM)
      struct WalkX {
        int l, base;
      };
      const int rots = BWD_ROT ? (wn * (NS / NW) + tile * 3) % NS : 0;
      const WalkX x0{rots % KSl, (rots / KSl) * KSg};
      auto advx = [&](WalkX &w) {
        const int l1 = w.l + 1;
        const bool wrap = l1 == KSl;
        const int b1 = w.base + KSg;
        w.base = wrap ? (b1 == 4 * KSg ? 0 : b1) : w.base;
        w.l = wrap ? 0 : l1;
      };
      const unsigned char *lax = dgb + (size_t)(lx >> 5) * 2048 + (size_t)(((lx >> 4) & 1) * 32 + rh * 16 + (lx & 15)) * 16;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.KxT16), 0, 4 * KS * 2048, 0x00020000);
      // outputs through descriptors as well (the launcher's caller keeps V * E * 4 < 2 GiB on this path: 32-bit offsets)
      const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(a.d_emb, 0, a.V * a.E * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.hot_part, 0, T * a.NT32 * 2 * 512, 0x00020000);
      // token ids of the four rows whose dX this lane will hold: requested here, used after the k-loop (buffer loads:
      // SGPR base + one lane offset + an SGPR per row; rows >= B fall outside the descriptor and read 0, masked below)
      int xid[4];
      {
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.ids), 0, a.B * T * 4, 0x00020000);
        const int b0 = tile * 32 + rh * 16 + 4 * (lx >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int id = __builtin_amdgcn_raw_buffer_load_b32(irs, b0 * T * 4, (i * T + t) * 4, 0);
          xid[i] = (b0 + i < a.B && id >= 0 && id < a.V) ? id : -1;  // out-of-range ids were flagged by the forward pass
        }
      }
      f32x4 xacc[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) xacc[j] = f32x4{0, 0, 0, 0};
      bool elive[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) elive[j] = (et0 + j * (NW / 2)) * 16 < a.E;
      // B fragments: a ring of PFX 32-n steps in flight (L2; one step is 3 MFMAs of 16 cycles: without the ring this loop
      // waited 25 k cycles per BPTT step, clock64); A fragments (LDS) one step ahead.  NS = 4 * KSl is a multiple of PFX.
      constexpr int PFX = 4;
      bx_bf16x8 xb[PFX][NE][2], xa[2][2];
      auto xbld = [&](int j, int ks, int hl) -> bx_bf16x8 {
        return __builtin_bit_cast(bx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, lx * 16, (((et0 + j * (NW / 2)) * KS + ks) * 2 + hl) * 1024, 0));
      };
      WalkX xwb = x0, xwa = x0;
#pragma unroll
      for (int p = 0; p < PFX; ++p) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          xb[p][j][0] = xbld(j, xwb.base + xwb.l, 0);
          xb[p][j][1] = xbld(j, xwb.base + xwb.l, 1);
        }
        advx(xwb);
      }
      xa[0][0] = *reinterpret_cast<const bx_bf16x8 *>(lax + (size_t)(xwa.base + xwa.l) * 4096);
      xa[0][1] = *reinterpret_cast<const bx_bf16x8 *>(lax + (size_t)(xwa.base + xwa.l) * 4096 + 1024);
      advx(xwa);
      for (int i0 = 0; i0 < NS; i0 += PFX) {
#pragma unroll
        for (int p = 0; p < PFX; ++p) {
          xa[(p + 1) & 1][0] = *reinterpret_cast<const bx_bf16x8 *>(lax + (size_t)(xwa.base + xwa.l) * 4096);
          xa[(p + 1) & 1][1] = *reinterpret_cast<const bx_bf16x8 *>(lax + (size_t)(xwa.base + xwa.l) * 4096 + 1024);
          advx(xwa);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NE; ++j)
            if (elive[j]) {
              xacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[p & 1][1], xb[p][j][0], xacc[j], 0, 0, 0);  // dg_lo * K_hi
              xacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[p & 1][0], xb[p][j][1], xacc[j], 0, 0, 0);  // dg_hi * K_lo
              xacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[p & 1][0], xb[p][j][0], xacc[j], 0, 0, 0);  // dg_hi * K_hi
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NE; ++j) {
            xb[p][j][0] = xbld(j, xwb.base + xwb.l, 0);
            xb[p][j][1] = xbld(j, xwb.base + xwb.l, 1);
          }
          advx(xwb);
        }
      }
      BWD_CLK(5)
      // scatter-add into the dense embedding gradient (duplicate ids summed, as TF's sparse Adagrad does); lane holds
      // rows rh*16 + 4 (l >> 4) + i, column e.  PAD (0) and EOS (1) fill most rows of a left-padded batch: their sums
      // go to hot_part without atomics and dx_hot_reduce_kernel adds the blocks in fixed order.  sum(dx^2) is taken per
      // OCCURRENCE (un-deduplicated IndexedSlices, sse_model.py:359-362).
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (!elive[j]) continue;
        const int e = (et0 + j * (NW / 2)) * 16 + (lx & 15);
        float h0 = 0.0f, h1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = xacc[j][i];
          const int id = xid[i];
          if (id >= 0) xsq += v * v;  // (columns >= E multiply zero weights: v = 0)
          h0 += (id == 0) ? v : 0.0f;
          h1 += (id == 1) ? v : 0.0f;
          if (id >= 2 && e < a.E) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, ers, (id * a.E + e) * 4, 0, 0);
        }
        h0 += __shfl_xor(h0, 16);
        h1 += __shfl_xor(h1, 16);
        h0 += __shfl_xor(h0, 32);
        h1 += __shfl_xor(h1, 32);
        if (lx < 16) {
          const int so = ((t * a.NT32 + tile) * 2 + rh) * 512;  // wave-uniform
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h0), hrs, e * 4, so, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h1), hrs, e * 4, so + 256, 0);
        }
      }
    }
    BWD_CLK(6)
    refill();
    BWD_CLK(4)
    if (!dump_first) dump();
    BWD_CLK(2)
    __syncthreads();
    BWD_CLK(7)
  }
#ifdef SSE_BWD_CLOCK
  if (a.clk && tile == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) a.clk[wn * 8 + i] = ck_[i];
#endif
  if constexpr (X3) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xsq += __shfl_xor(xsq, o);
    if (lane == 0) a.sq_part[tile * NW + wn] = xsq;
  }

  // bias-gradient partials of this tile (sum over its 32 rows and all steps)
#pragma unroll
  for (int u = 0; u < UB; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = dbacc[u][g];
      v += __shfl_xor(v, 32);
      if (half == 0) a.db_part[(size_t)tile * 4 * Hp + g * Hp + (wn * UB + u) * 32 + lane] = v;
    }
}

// ---------------------------------------------------------------------------
// dK partials: out[slice][k'][n] = sum_{r in slice} A[r][k'] * dG[r][n]   (the time-batched A^T dG GEMM,
// 2*(E+H)*4H*T*B flop: as much work as the forward pass).
// Workgroup = 8 waves = all KT k'-tiles x 8 consecutive n-tiles; wave w owns n-tile w for every k'-tile
// (KT accumulators), so per r-group it loads KT A fragments -- the same ones its 7 siblings load, served
// by L1 -- and ONE dG fragment for 4*KT MFMAs; dG is read from HBM exactly once, A once per 8 n-tiles.
// The r loop is software-pipelined by hand (two named operand sets).  grid = (NTn/8, slices).
struct DkArgs {
  const float *tape_a;  // [(RG)][KT][256]
  const float *dg_b;    // [(RG)][NTn][256]
  float *part;          // [SL][KT*32][NTn*32]
  int32_t RG, KT, NTn, SL;
  uint32_t live_k;      // bit i: k'-tile i has non-padding rows (x: 2 tiles of 32 up to E; h: one per 32 units up to H)
  int32_t pair_rg;      // 0, or r-groups per step of tape_a (= 4 * NT32/2) when the batch is (pos, neg) pairs sharing
                        // their source: tape_a holds RG r-groups of the shared rows, dg_b 2*RG (per step: the pos tiles,
                        // then the neg tiles) and A^T dG_pos + A^T dG_neg = A^T (dG_pos + dG_neg): half the MFMAs
};

// (n-group, r-slice) of a workgroup, XCD-aware: the gridDim.x n-groups of ONE slice read the same A-tape rows, and workgroup
// ids are dealt round-robin to the 8 XCDs (each with its own L2) -- in plain (x, y) order the n-groups of a slice land on
// gridDim.x different XCDs and every one of them pulls the A-tape from HBM (FETCH_SIZE 2.4 GB against 1.4 GB of operands at
// 8192 x 32 rows, profiles/r03x_hbm_pmc.txt).  Remapped, they run back to back on one XCD: one HBM read, the rest L2 hits.
__device__ __forceinline__ void dk_block_map(int &ngrp, int &slice) {
  const int NG = gridDim.x, SL = gridDim.y;
  ngrp = blockIdx.x;
  slice = blockIdx.y;
  if ((SL & 7) == 0) {
    const int p = blockIdx.x + NG * blockIdx.y, xcd = p & 7, j = p >> 3;
    ngrp = j % NG;
    slice = (j / NG) * 8 + xcd;
  }
}

// KT fragments of one LDS stage, software-pipelined one fragment ahead of its 4 MFMAs (used twice in the kernel)
#define DK_COMPUTE(CUR)                                                                                       \
  {                                                                                                             \
    const float *sa = &stage[CUR][0][lane * 4];                                                                 \
    f32x4 ax = *reinterpret_cast<const f32x4 *>(sa), ay;                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                              \
    _Pragma("unroll") for (int i = 0; i < KT; i += 2) {                                                         \
      if (i + 1 < KT) ay = *reinterpret_cast<const f32x4 *>(sa + (i + 1) * 256);                                \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if ((live_k >> i) & 1) {                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bc[e], acc[i], 0, 0, 0);                       \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if (i + 2 < KT) ax = *reinterpret_cast<const f32x4 *>(sa + (i + 2) * 256);                                \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if (i + 1 < KT && ((live_k >> (i + 1)) & 1)) {                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
            acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], bc[e], acc[i + 1], 0, 0, 0);               \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                              \
  }

template <int KT, bool PAIR>
__global__ __launch_bounds__(512) void dk_gemm_kernel(DkArgs a) {
  // The KT A fragments of an r-group are the same for all 8 waves: they are fetched ONCE per workgroup
  // (each wave brings 1-2 of the KT 1-KB blocks) into a double-buffered LDS stage, one barrier per r-group;
  // without it every wave pulled them through L1/L2 itself (17 B/clk/CU of L2 traffic, kernel at 50 %).
  __shared__ __attribute__((aligned(16))) float stage[2][KT][256];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t live_k = a.live_k;  // all-padding k'-tiles have all-zero A rows: their accumulators stay 0
  int ngrp, slice;
  dk_block_map(ngrp, slice);
  const int nt = min(ngrp * 8 + w, a.NTn - 1);  // surplus waves recompute the last tile (no divergent barriers)
  const bool live = ngrp * 8 + w < a.NTn;
  const int per = ((a.RG + a.SL - 1) / a.SL + 1) & ~1;  // even; RG = T*NT32*4 is a multiple of 4: every slice is even
  const int rg0 = slice * per, rg1 = min(a.RG, rg0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const float *pa = a.tape_a + lane * 4;
  const float *pb = a.dg_b + (size_t)nt * 256 + lane * 4;
  const int pair_rg = a.pair_rg;
  // dG fragment of r-group rg (an index into tape_a): with pairs, the sum of the two rows' fragments
  auto gload_b = [&](int rg) -> f32x4 {
    if constexpr (!PAIR) return *reinterpret_cast<const f32x4 *>(pb + (size_t)rg * a.NTn * 256);
    const int t = rg / pair_rg, d1 = rg + t * pair_rg;  // (t * 2 * pair_rg + rg % pair_rg)
    const f32x4 u = *reinterpret_cast<const f32x4 *>(pb + (size_t)d1 * a.NTn * 256);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(pb + (size_t)(d1 + pair_rg) * a.NTn * 256);
    return u + v;
  };
  constexpr bool TWO = KT > 8;  // waves 0 .. KT-9 bring a second block
  const bool second = TWO && (8 + w < KT);
  auto gload_a = [&](int rg, f32x4 &s0, f32x4 &s1) {
    if (w < KT) s0 = *reinterpret_cast<const f32x4 *>(pa + ((size_t)rg * KT + w) * 256);
    if (second) s1 = *reinterpret_cast<const f32x4 *>(pa + ((size_t)rg * KT + 8 + w) * 256);
  };
  auto stash = [&](int buf, const f32x4 &s0, const f32x4 &s1) {
    if (w < KT) *reinterpret_cast<f32x4 *>(&stage[buf][w][lane * 4]) = s0;
    if (second) *reinterpret_cast<f32x4 *>(&stage[buf][8 + w][lane * 4]) = s1;
  };
  if (rg0 < rg1) {
    // Operands are requested TWO r-groups ahead (two named register sets, X and Y, used alternately): under the
    // streaming load of this kernel an HBM access takes longer than one r-group of MFMAs (clock64: the hand-over
    // below waited 20-30 % of the kernel when the distance was one).
    f32x4 xs0 = {0, 0, 0, 0}, xs1 = {0, 0, 0, 0}, xb, ys0 = {0, 0, 0, 0}, ys1 = {0, 0, 0, 0}, yb, bc;
    auto clampr = [&](int rg) { return rg < rg1 ? rg : rg1 - 1; };
    gload_a(rg0, xs0, xs1);
    bc = gload_b(rg0);
    stash(0, xs0, xs1);
    gload_a(clampr(rg0 + 1), xs0, xs1);
    xb = gload_b(clampr(rg0 + 1));
    gload_a(clampr(rg0 + 2), ys0, ys1);
    yb = gload_b(clampr(rg0 + 2));
    __syncthreads();
    for (int rg = rg0; rg < rg1; rg += 2) {
      // even step: stage 0 holds r-group rg; set X holds rg+1 (to stage 1), then refetches rg+3
      DK_COMPUTE(0)
      if (rg + 1 < rg1) stash(1, xs0, xs1);
      bc = xb;
      gload_a(clampr(rg + 3), xs0, xs1);
      xb = gload_b(clampr(rg + 3));
      __syncthreads();
      // odd step: stage 1 holds rg+1; set Y holds rg+2 (to stage 0), then refetches rg+4
      DK_COMPUTE(1)
      if (rg + 2 < rg1) stash(0, ys0, ys1);
      bc = yb;
      gload_a(clampr(rg + 4), ys0, ys1);
      yb = gload_b(clampr(rg + 4));
      __syncthreads();
    }
  }
  if (!live) return;
  const int ldn = a.NTn * 32;
  float *out = a.part + (size_t)slice * KT * 32 * ldn;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(i * 32 + mfma_row(r, lane)) * ldn + nt * 32 + (lane & 31)] = acc[i][r];
}

// ---------------------------------------------------------------------------
// dk_gemm_kernel with TWO r-groups per barrier: four LDS stages (two in use, two being filled), the operands of a pair of
// r-groups requested one pair ahead.  Half the barriers: after every barrier the matrix pipe waits for the first A fragment
// to come back from LDS and for the slowest of the 8 waves (clock64: ~8 % of the kernel with one r-group per barrier).
template <int KT, bool PAIR>
__global__ __launch_bounds__(512) void dk_gemm2_kernel(DkArgs a) {
  __shared__ __attribute__((aligned(16))) float stage[4][KT][256];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t live_k = a.live_k;
  int ngrp, slice;
  dk_block_map(ngrp, slice);
  const int nt = min(ngrp * 8 + w, a.NTn - 1);
  const bool live = ngrp * 8 + w < a.NTn;
  const int per = ((a.RG + a.SL - 1) / a.SL + 1) & ~1;  // even (see dk_gemm_kernel): every slice is a whole number of pairs
  const int rg0 = slice * per, rg1 = min(a.RG, rg0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const float *pa = a.tape_a + lane * 4;
  const float *pb = a.dg_b + (size_t)nt * 256 + lane * 4;
  const int pair_rg = a.pair_rg;
  auto gload_b = [&](int rg) -> f32x4 {
    if constexpr (!PAIR) return *reinterpret_cast<const f32x4 *>(pb + (size_t)rg * a.NTn * 256);
    const int t = rg / pair_rg, d1 = rg + t * pair_rg;
    const f32x4 u = *reinterpret_cast<const f32x4 *>(pb + (size_t)d1 * a.NTn * 256);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(pb + (size_t)(d1 + pair_rg) * a.NTn * 256);
    return u + v;
  };
  constexpr bool TWO = KT > 8;
  const bool second = TWO && (8 + w < KT);
  struct Set {
    f32x4 s0[2], s1[2], b[2];
  };
  auto gload = [&](int rg, Set &x) {  // operands of r-groups rg, rg + 1 (rg even relative to rg0; rg + 1 < rg1 by construction)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (w < KT) x.s0[j] = *reinterpret_cast<const f32x4 *>(pa + ((size_t)(rg + j) * KT + w) * 256);
      if (second) x.s1[j] = *reinterpret_cast<const f32x4 *>(pa + ((size_t)(rg + j) * KT + 8 + w) * 256);
      x.b[j] = gload_b(rg + j);
    }
  };
  auto stash = [&](int buf, const Set &x) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (w < KT) *reinterpret_cast<f32x4 *>(&stage[buf * 2 + j][w][lane * 4]) = x.s0[j];
      if (second) *reinterpret_cast<f32x4 *>(&stage[buf * 2 + j][8 + w][lane * 4]) = x.s1[j];
    }
  };
  auto compute = [&](int st, const f32x4 &bc) {
    const float *sa = &stage[st][0][lane * 4];
    f32x4 ax = *reinterpret_cast<const f32x4 *>(sa), ay;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < KT; i += 2) {
      if (i + 1 < KT) ay = *reinterpret_cast<const f32x4 *>(sa + (i + 1) * 256);
      __builtin_amdgcn_sched_barrier(0);
      if ((live_k >> i) & 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bc[e], acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < KT) ax = *reinterpret_cast<const f32x4 *>(sa + (i + 2) * 256);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < KT && ((live_k >> (i + 1)) & 1)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], bc[e], acc[i + 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (rg0 < rg1) {  // (rg1 - rg0 is even: `per` and RG are)
    auto clampr = [&](int rg) { return rg < rg1 ? rg : rg1 - 2; };
    Set x, y;
    f32x4 bc[2];
    x.s1[0] = x.s1[1] = y.s1[0] = y.s1[1] = f32x4{0, 0, 0, 0};
    gload(rg0, x);
    stash(0, x);
    bc[0] = x.b[0];
    bc[1] = x.b[1];
    gload(clampr(rg0 + 2), x);
    gload(clampr(rg0 + 4), y);
    __syncthreads();
    for (int rg = rg0; rg < rg1; rg += 4) {
      // even pair: stages 0, 1 hold r-groups rg, rg + 1; set X holds rg + 2, rg + 3 (to stages 2, 3), then refetches rg + 6, rg + 7
      compute(0, bc[0]);
      compute(1, bc[1]);
      if (rg + 2 < rg1) stash(1, x);
      bc[0] = x.b[0];
      bc[1] = x.b[1];
      gload(clampr(rg + 6), x);
      __syncthreads();
      if (rg + 2 >= rg1) break;
      // odd pair: stages 2, 3 hold rg + 2, rg + 3; set Y holds rg + 4, rg + 5 (to stages 0, 1), then refetches rg + 8, rg + 9
      compute(2, bc[0]);
      compute(3, bc[1]);
      if (rg + 4 < rg1) stash(0, y);
      bc[0] = y.b[0];
      bc[1] = y.b[1];
      gload(clampr(rg + 8), y);
      __syncthreads();
    }
  }
  if (!live) return;
  const int ldn = a.NTn * 32;
  float *out = a.part + (size_t)slice * KT * 32 * ldn;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(i * 32 + mfma_row(r, lane)) * ldn + nt * 32 + (lane & 31)] = acc[i][r];
}

// ---------------------------------------------------------------------------
// dK partials, third structure: the A-tape blocks go global -> LDS by LDS-DMA (global_load_lds_dwordx4: a frag32 block IS the
// lane-linear 1 KiB image the DMA writes), so the stage costs no registers and no ds_write pass and can be G = 4 r-groups
// deep: ONE barrier per 4 r-groups (160 MFMAs per wave).  Two halves of the stage alternate; the DMA of the next set and
// the next set's dG fragments (registers) are issued right after the barrier and waited for (vmcnt(0), ~20 k cycles later)
// in front of the next one.  Raw s_barrier: __syncthreads() would drain the DMA queue at once (cdna_hip_programming.md).
// Measured at 8192 x 32 rows, H = 256: one r-group per barrier 1.54 ms, two 1.43 ms, this kernel: see profiles/r04_notes.txt.
template <int KT, bool PAIR>
__global__ __launch_bounds__(512) void dk_gemm3_kernel(DkArgs a) {
  constexpr int G = 4;
  // [2][G][KT][256] floats, dynamic: the only writer of this array is the DMA engine, and a static __shared__ array without a
  // visible store let the compiler fold every read of it (and with them the whole GEMM) to undef
  extern __shared__ __attribute__((aligned(16))) float dk_stage[];
  auto stg = [&](int buf, int g, int kt) -> float * { return dk_stage + (size_t)((buf * G + g) * KT + kt) * 256; };
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t live_k = a.live_k;
  int ngrp, slice;
  dk_block_map(ngrp, slice);
  const int nt = min(ngrp * 8 + w, a.NTn - 1);
  const bool live = ngrp * 8 + w < a.NTn;
  const int per = ((a.RG + a.SL - 1) / a.SL + 1) & ~1;
  const int rg0 = slice * per, rg1 = min(a.RG, rg0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const float *pa = a.tape_a + lane * 4;
  const float *pb = a.dg_b + (size_t)nt * 256 + lane * 4;
  const int pair_rg = a.pair_rg;
  auto gload_b = [&](int rg) -> f32x4 {
    if constexpr (!PAIR) return *reinterpret_cast<const f32x4 *>(pb + (size_t)rg * a.NTn * 256);
    const int t = rg / pair_rg, d1 = rg + t * pair_rg;
    const f32x4 u = *reinterpret_cast<const f32x4 *>(pb + (size_t)d1 * a.NTn * 256);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(pb + (size_t)(d1 + pair_rg) * a.NTn * 256);
    return u + v;
  };
  // the G * KT blocks of a set, dealt to the 8 waves (5 each at KT = 10, 3 at KT = 6); r-groups past the slice are clamped
  // (their products are skipped below)
  constexpr int NB = G * KT, PER_W = (NB + 7) / 8;
  auto dma = [&](int buf, int rg) {
#pragma unroll
    for (int j = 0; j < PER_W; ++j) {
      const int blk = w + 8 * j;
      if (blk < NB) {
        const int g = blk / KT, kt = blk - g * KT;
        const int rgc = min(rg + g, rg1 - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pa + ((size_t)rgc * KT + kt) * 256),
                                         (__attribute__((address_space(3))) void *)stg(buf, g, kt), 16, 0, 0);
      }
    }
  };
  auto compute = [&](int buf, int g, const f32x4 &bc) {
    const float *sa = stg(buf, g, 0) + lane * 4;
    f32x4 ax = *reinterpret_cast<const f32x4 *>(sa), ay;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < KT; i += 2) {
      if (i + 1 < KT) ay = *reinterpret_cast<const f32x4 *>(sa + (i + 1) * 256);
      __builtin_amdgcn_sched_barrier(0);
      if ((live_k >> i) & 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bc[e], acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < KT) ax = *reinterpret_cast<const f32x4 *>(sa + (i + 2) * 256);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < KT && ((live_k >> (i + 1)) & 1)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], bc[e], acc[i + 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // every wave waits for ITS loads (the DMA pieces it issued and its dG fragments), then the barrier publishes the stage.
  // The fragments pass through the asm as outputs: the compiler then treats them as ready and puts no wait of its own in
  // front of their first use (with a DMA in flight that wait would be vmcnt(0) at the top of the compute phase)
  auto land_and_sync = [&](f32x4 (&bn)[G]) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3])::"memory");
  };
  if (rg0 < rg1) {
    f32x4 bc[G], bn[G];
    auto clampr = [&](int rg) { return rg < rg1 ? rg : rg1 - 1; };
#pragma unroll
    for (int g = 0; g < G; ++g) bn[g] = gload_b(clampr(rg0 + g));
    dma(0, rg0);
    land_and_sync(bn);
    int buf = 0;
    for (int rg = rg0; rg < rg1; rg += G) {
#pragma unroll
      for (int g = 0; g < G; ++g) bc[g] = bn[g];
      if (rg + G < rg1) {  // next set: dG fragments first (plain loads), then the DMA pieces into the other half of the stage
#pragma unroll
        for (int g = 0; g < G; ++g) bn[g] = gload_b(clampr(rg + G + g));
        dma(buf ^ 1, rg + G);
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (rg + g < rg1) compute(buf, g, bc[g]);
      land_and_sync(bn);
      buf ^= 1;
    }
  }
  if (!live) return;
  const int ldn = a.NTn * 32;
  float *out = a.part + (size_t)slice * KT * 32 * ldn;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(i * 32 + mfma_row(r, lane)) * ldn + nt * 32 + (lane & 31)] = acc[i][r];
}

// ---------------------------------------------------------------------------
// The same dK partials on the bf16 matrix pipe with split operands (see sse_kernels.h): tape_a and dG arrive as hi / lo
// frag16 blocks (written that way by the forward and BPTT kernels: 16 reduction rows per block, same bytes as fp32), a
// product costs three v_mfma_f32_32x32x16_bf16 (32 cycles each, 16 r) instead of eight v_mfma_f32_32x32x2_f32 (64 cycles).
// Relative error of a product ~4e-6: below the fp32 summation-order noise of a 262144-term gradient sum.
// Same decomposition: workgroup = all KT k'-tiles x 8 n-tiles, wave w owns n-tile w; the 2*KT A blocks of a group are
// fetched once per workgroup into a double-buffered LDS stage; operands requested two groups ahead.
// PAIR: A^T dG_pos + A^T dG_neg = A^T (dG_pos + dG_neg): the two rows' dG fragments are rebuilt in fp32, added and split again.
typedef short dk_bf16x8 __attribute__((ext_vector_type(8)));
struct DkX3Args {
  const unsigned short *tape_a;  // [(G)][KT][hi|lo][512]
  const unsigned short *dg_b;    // [(G or 2G)][NTn][hi|lo][512]
  float *part;                   // [SL][KT*32][NTn*32]
  int32_t G, KT, NTn, SL;        // G = 16-row groups of tape_a
  uint32_t live_k;
  int32_t pair_g;                // 0, or 16-row groups per step of tape_a (= 2 * NT32/2)
};

template <int KT, bool PAIR>
__global__ __launch_bounds__(512) void dk_x3_kernel(DkX3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned short stage[2][2 * KT][512];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t live_k = a.live_k;
  int ngrp, slice;
  dk_block_map(ngrp, slice);
  const int nt = min(ngrp * 8 + w, a.NTn - 1);
  const bool live = ngrp * 8 + w < a.NTn;
  const int per = (a.G + a.SL - 1) / a.SL;
  const int g0 = slice * per, g1 = min(a.G, g0 + per);
  f32x16 acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const unsigned short *pa = a.tape_a + lane * 8;
  const unsigned short *pb = a.dg_b + (size_t)nt * 1024 + lane * 8;
  const int pair_g = a.pair_g;
  constexpr int NBLK = 2 * KT;                   // A blocks per group
  constexpr int PER_W = (NBLK + 7) / 8;          // blocks a wave brings (<= 3)
  struct Set {
    sse_u32x4 s[PER_W];
    dk_bf16x8 bh, bl;
  };
  auto gload = [&](int g, Set &x) {
#pragma unroll
    for (int j = 0; j < PER_W; ++j)
      if (w + 8 * j < NBLK) x.s[j] = *reinterpret_cast<const sse_u32x4 *>(pa + ((size_t)g * NBLK + w + 8 * j) * 512);
    if constexpr (!PAIR) {
      const unsigned short *p = pb + (size_t)g * a.NTn * 1024;
      x.bh = *reinterpret_cast<const dk_bf16x8 *>(p);  // (nt loads measured 10 % slower here)
      x.bl = *reinterpret_cast<const dk_bf16x8 *>(p + 512);
    } else {
      const int t = g / pair_g, d1 = g + t * pair_g;
      const unsigned short *p1 = pb + (size_t)d1 * a.NTn * 1024, *p2 = pb + (size_t)(d1 + pair_g) * a.NTn * 1024;
      const sse_u32x4 h1 = *reinterpret_cast<const sse_u32x4 *>(p1), l1 = *reinterpret_cast<const sse_u32x4 *>(p1 + 512);
      const sse_u32x4 h2 = *reinterpret_cast<const sse_u32x4 *>(p2), l2 = *reinterpret_cast<const sse_u32x4 *>(p2 + 512);
      float v8[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v8[2 * e] = (__uint_as_float(h1[e] << 16) + __uint_as_float(l1[e] << 16)) + (__uint_as_float(h2[e] << 16) + __uint_as_float(l2[e] << 16));
        v8[2 * e + 1] = (__uint_as_float(h1[e] & 0xffff0000u) + __uint_as_float(l1[e] & 0xffff0000u)) +
                        (__uint_as_float(h2[e] & 0xffff0000u) + __uint_as_float(l2[e] & 0xffff0000u));
      }
      sse_u32x4 hi, lo;
      sse_split8(v8, hi, lo);
      x.bh = __builtin_bit_cast(dk_bf16x8, hi);
      x.bl = __builtin_bit_cast(dk_bf16x8, lo);
    }
  };
  auto stash = [&](int buf, const Set &x) {
#pragma unroll
    for (int j = 0; j < PER_W; ++j)
      if (w + 8 * j < NBLK) *reinterpret_cast<sse_u32x4 *>(&stage[buf][w + 8 * j][lane * 8]) = x.s[j];
  };
  auto compute = [&](int buf, const dk_bf16x8 &bh, const dk_bf16x8 &bl) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      if ((live_k >> i) & 1) {
        const dk_bf16x8 ah = *reinterpret_cast<const dk_bf16x8 *>(&stage[buf][2 * i][lane * 8]);
        const dk_bf16x8 al = *reinterpret_cast<const dk_bf16x8 *>(&stage[buf][2 * i + 1][lane * 8]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (g0 < g1) {
    auto clampg = [&](int g) { return g < g1 ? g : g1 - 1; };
    Set x, y;
    dk_bf16x8 bh, bl;
    gload(g0, x);
    stash(0, x);
    bh = x.bh;
    bl = x.bl;
    gload(clampg(g0 + 1), x);
    gload(clampg(g0 + 2), y);
    __syncthreads();
    for (int g = g0; g < g1; g += 2) {
      // even step: stage 0 holds group g; set X holds g+1 (to stage 1), then refetches g+3
      compute(0, bh, bl);
      if (g + 1 < g1) stash(1, x);
      bh = x.bh;
      bl = x.bl;
      gload(clampg(g + 3), x);
      __syncthreads();
      if (g + 1 >= g1) break;
      // odd step: stage 1 holds g+1; set Y holds g+2 (to stage 0), then refetches g+4
      compute(1, bh, bl);
      if (g + 2 < g1) stash(0, y);
      bh = y.bh;
      bl = y.bl;
      gload(clampg(g + 4), y);
      __syncthreads();
    }
  }
  if (!live) return;
  const int ldn = a.NTn * 32;
  float *out = a.part + (size_t)slice * KT * 32 * ldn;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(i * 32 + mfma_row(r, lane)) * ldn + nt * 32 + (lane & 31)] = acc[i][r];
}

// sum the slices and write/accumulate into the variable-shaped gradient dK [(E+H)][4H]
// db (optional): d(bias) = column sums of dG = row k' = E of the partials -- the A-tape carries the constant-1 column that
// feeds the bias through the forward GEMM (E < 64), so the weight-gradient GEMM produces the bias gradient for free
__global__ void dk_reduce_kernel(const float *part, int SL, int KT, int NTn, int E, int H, int Hp, int accumulate,
                                 float *dK, float *db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = E + H, cols = 4 * H;
  if (i >= (rows + (db ? 1 : 0)) * cols) return;
  const int col = i % cols, row = i / cols;
  const bool is_db = row == rows;
  const int kp = is_db ? E : (row < E) ? row : 64 + (row - E);
  const int g = col / H, unit = col % H;
  const int n = g * Hp + unit;
  const size_t ldn = (size_t)NTn * 32, plane = (size_t)KT * 32 * ldn;
  float acc = 0.0f;
  for (int s = 0; s < SL; ++s) acc += part[s * plane + (size_t)kp * ldn + n];
  float *out = is_db ? db + col : dK + i;
  *out = accumulate ? *out + acc : acc;
}

__global__ void db_reduce_kernel(const float *db_part, int NT32, int H, int Hp, int accumulate, float *db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * H) return;
  const int g = i / H, unit = i % H;
  // 8 loads in flight (a plain loop waited one memory round trip per tile: 0.1 ms at 256 tiles); fixed summation order
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t0 = 0; t0 < NT32; t0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      part[j] += (t0 + j < NT32) ? db_part[(size_t)(t0 + j) * 4 * Hp + g * Hp + unit] : 0.0f;
  }
  const float acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  db[i] = accumulate ? db[i] + acc : acc;
}

// ---------------------------------------------------------------------------
// dX[r][e] = sum_n dg[r][n] * Kx[e][n]; scatter-add into the dense embedding
// gradient (duplicate ids summed, as TF's sparse Adagrad does) and accumulate
// sum(dx^2) over OCCURRENCES (tf.global_norm takes IndexedSlices.values raw).
struct DxArgs {
  const float *dg_a;   // [T][NT32][KGn][256]
  const float *KxT;    // frag32(rows = e (64), red = n): Kx^T, 2 tiles
  const int32_t *ids;  // [B][T]
  float *d_emb;        // [V][E] zero-initialised
  float *sq_part;      // [T*NT32] partial sums of dx^2
  int32_t T, NT32, KGn, B, E, V;
  int32_t KGl;         // live k-groups per gate = ceil(H/8) (dG columns of padded units are 0)
  float *hot_part;     // [T*NT32][2][64] this block's sums for the ids 0 (PAD) and 1 (EOS): no atomics for them
};

__global__ __launch_bounds__(64) void dx_kernel(DxArgs a) {
  const int lane = threadIdx.x;
  const int t = blockIdx.x / a.NT32, tile = blockIdx.x % a.NT32;
  const float *pa = a.dg_a + (size_t)blockIdx.x * a.KGn * 256 + lane * 4;
  const float *pb = a.KxT + lane * 4;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  // the dG tile streams from HBM: a ring of 4 k-groups of operands in flight (KGn = Hp/2 is a multiple of 4)
  constexpr int RING = 4;
  f32x4 ra[RING], rb0[RING], rb1[RING];
  // 4 gates x KGl live k-groups (padding skipped), walked with (gate base, group) counters instead of a division per
  // k-group; NL is a multiple of 4
  const int KGg = a.KGn / 4, KGl = a.KGl, NL = 4 * KGl;
  int wl_ = 0, wbase = 0;
  auto adv = [&]() {
    const int l1 = wl_ + 1;
    const bool wrap = l1 == KGl;
    const int b1 = wbase + KGg;
    wbase = wrap ? (b1 == 4 * KGg ? 0 : b1) : wbase;
    wl_ = wrap ? 0 : l1;
  };
#pragma unroll
  for (int d = 0; d < RING; ++d) {
    const int k = wbase + wl_;
    adv();
    ra[d] = *reinterpret_cast<const f32x4 *>(pa + k * 256);
    rb0[d] = *reinterpret_cast<const f32x4 *>(pb + k * 256);
    rb1[d] = *reinterpret_cast<const f32x4 *>(pb + (size_t)(a.KGn + k) * 256);
  }
  for (int kg0 = 0; kg0 < NL; kg0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][e], rb0[d][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][e], rb1[d][e], acc[1], 0, 0, 0);
      }
      const int kn = wbase + wl_;  // (wraps to the start past the end: a harmless reload)
      adv();
      ra[d] = *reinterpret_cast<const f32x4 *>(pa + kn * 256);
      rb0[d] = *reinterpret_cast<const f32x4 *>(pb + kn * 256);
      rb1[d] = *reinterpret_cast<const f32x4 *>(pb + (size_t)(a.KGn + kn) * 256);
    }
  }
  // Scatter-add into the dense embedding gradient.  Real batches are left-padded and end in EOS: at most steps most
  // rows of a tile carry the same id (PAD = 0, EOS = 1), and 8192 rows x 50 columns of float atomics on ONE embedding
  // row per step serialise in L2 (measured: 0.36 ms with random ids, 0.7 - 2.1 ms with just an EOS column).  So
  //  * rows of this lane's half that repeat an earlier id are first added onto that row's value (selects, no dynamic
  //    register indexing): one atomic per distinct id of the half;
  //  * ids 0 and 1 take no atomics at all: the block's sum for each goes to hot_part[block] and dx_hot_reduce_kernel
  //    adds the blocks in fixed order.
  // The squared norm is taken over the un-merged rows (TF: un-deduplicated IndexedSlices).
  float sq = 0.0f;
  int idr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = tile * 32 + mfma_row(r, lane);
    int id = -1;
    if (b < a.B) {
      id = a.ids[(size_t)b * a.T + t];
      if (id < 0 || id >= a.V) id = -1;  // flagged by the forward pass
    }
    idr[r] = id;
    if (id >= 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j * 32 + (lane & 31) < a.E) sq += acc[j][r] * acc[j][r];
    }
  }
  // leaders: row r is merged into the first earlier row r2 of this half with the same id
#pragma unroll
  for (int r = 15; r > 0; --r) {
    bool merged = false;
#pragma unroll
    for (int r2 = 0; r2 < r; ++r2) {
      const bool hit = !merged && idr[r] >= 0 && idr[r2] == idr[r];
      acc[0][r2] += hit ? acc[0][r] : 0.0f;
      acc[1][r2] += hit ? acc[1][r] : 0.0f;
      merged = merged || hit;
    }
    if (merged) idr[r] = -1;
  }
#pragma unroll
  for (int hid = 0; hid < 2; ++hid) {
    float h0 = 0.0f, h1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool is = idr[r] == hid;  // at most one leader per half
      h0 += is ? acc[0][r] : 0.0f;
      h1 += is ? acc[1][r] : 0.0f;
      if (is) idr[r] = -1;
    }
    h0 += __shfl_xor(h0, 32);
    h1 += __shfl_xor(h1, 32);
    if (lane < 32) {
      float *hp = a.hot_part + ((size_t)blockIdx.x * 2 + hid) * 64;
      hp[lane] = h0;
      hp[32 + lane] = h1;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (idr[r] >= 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = j * 32 + (lane & 31);
        if (e < a.E) atomicAdd(a.d_emb + (size_t)idr[r] * a.E + e, acc[j][r]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0) a.sq_part[blockIdx.x] = sq;
}

// d_emb[hid][e] += sum over blocks of hot_part[block][hid][e]: grid (2 hot ids, DXH_SLICES), one float atomic per
// (slice, column) at the end (a single workgroup per hot id took 0.17 ms at 8192 blocks)
#define DXH_SLICES 32
__global__ __launch_bounds__(256) void dx_hot_reduce_kernel(const float *hot_part, int nblocks, int E, int V, float *d_emb) {
  __shared__ float red[4][64];
  const int hid = blockIdx.x, e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int per = (nblocks + DXH_SLICES - 1) / DXH_SLICES;
  const int b_begin = blockIdx.y * per, b_end = min(nblocks, b_begin + per);
  float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 8 loads in flight per thread
  for (int b0 = b_begin + sl; b0 < b_end; b0 += 32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + 4 * j;
      part[j] += (b < b_end) ? hot_part[((size_t)b * 2 + hid) * 64 + e] : 0.0f;
    }
  }
  red[sl][e] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  __syncthreads();
  if (sl == 0 && e < E && hid < V) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (v != 0.0f) atomicAdd(d_emb + (size_t)hid * E + e, v);
  }
}

// ---------------------------------------------------------------------------
// global norm + clip scale + Adagrad
__global__ void sumsq_kernel(const float *g, int64_t n, float *part /*[gridDim.x]*/) {
  __shared__ float sh[256];
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += g[i] * g[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// out[0] = sum of n floats (fixed order), out[3] = tag
__global__ void sum_kernel(const float *part, int n, float tag, float *out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)sh[0];
    out[3] = tag;
  }
}

// scal[0] = global norm, scal[1] = clip scale = clip * min(1/norm, 1/clip)
__global__ void clip_scale_kernel(const float *part, int n, float clip, float *scal) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float gn = (float)sqrt(sh[0]);
    scal[0] = gn;
    scal[1] = (gn > 0.0f) ? clip * fminf(1.0f / gn, 1.0f / clip) : 1.0f;
  }
}

// acc += g^2 ; w -= lr * g / sqrt(acc)   with g = scale * grad  (TF ApplyAdagrad)
__global__ void adagrad_kernel(float *w, float *accum, const float *grad, const float *scal, float lr, int64_t n) {
  const float sc = scal[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = grad[i] * sc;
    const float ac = accum[i] + g * g;
    accum[i] = ac;
    w[i] -= lr * g / sqrtf(ac);
  }
}

// All tensors of the model in ONE launch each (a train step used to end in 6 sumsq + 7 adagrad launches of ~5 us):
// blockIdx.y selects the tensor.
__global__ void sumsq_multi_kernel(MultiTensor mt, float *part /*[n][gridDim.x]*/) {
  __shared__ float sh[256];
  const float *g = mt.grad[blockIdx.y];
  const int64_t n = mt.count[blockIdx.y];
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += g[i] * g[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

// scal[0] = global norm over part[0..n) plus *extra (the raw embedding-slice norm), scal[1] = clip scale; a pending
// device error (err != 0: token id out of range seen by this step's kernels) turns the update into a no-op: scal[1] = 0
// and scal[2] = 1 (adagrad_multi_kernel returns), so the host may check the flag AFTER queueing the whole step
__global__ void clip_scale_multi_kernel(const float *part, int n, const float *extra, float clip, const int32_t *err, float *scal) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float gn = (float)sqrt(sh[0] + (double)*extra);
    const bool bad = err != nullptr && (*err & 11) != 0;  // id / row out of range, sparse-exchange overflow; bit 4 is the encoders' cluster give-up, not a train error
    scal[0] = gn;
    scal[1] = bad ? 0.0f : ((gn > 0.0f) ? clip * fminf(1.0f / gn, 1.0f / clip) : 1.0f);
    scal[2] = bad ? 1.0f : 0.0f;
  }
}

__global__ void adagrad_multi_kernel(MultiTensor mt, const float *scal, float lr) {
  if (scal[2] != 0.0f) return;  // step cancelled (see clip_scale_multi_kernel)
  float *w = mt.w[blockIdx.y], *accum = mt.slot[blockIdx.y];
  const float *grad = mt.grad[blockIdx.y];
  const int64_t n = mt.count[blockIdx.y];
  const float sc = scal[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = grad[i] * sc;
    const float ac = accum[i] + g * g;
    accum[i] = ac;
    w[i] -= lr * g / sqrtf(ac);
  }
}

// ---------------------------------------------------------------------------
// host-side launchers
hipError_t launch_sumsq_multi(const MultiTensor &mt, float *part, int nblocks, hipStream_t st) {
  if (mt.n == 0) return hipSuccess;
  hipLaunchKernelGGL(sumsq_multi_kernel, dim3(nblocks, mt.n), dim3(256), 0, st, mt, part);
  return hipGetLastError();
}
hipError_t launch_clip_scale_multi(const float *part, int n, const float *extra, float clip, const int32_t *err, float *scal,
                                   hipStream_t st) {
  hipLaunchKernelGGL(clip_scale_multi_kernel, dim3(1), dim3(256), 0, st, part, n, extra, clip, err, scal);
  return hipGetLastError();
}
hipError_t launch_adagrad_multi(const MultiTensor &mt, const float *scal, float lr, hipStream_t st) {
  if (mt.n == 0) return hipSuccess;
  int64_t mx = 0;
  for (int i = 0; i < mt.n; ++i) mx = mt.count[i] > mx ? mt.count[i] : mx;
  const int bx = (int)((mx + 255) / 256 < 1024 ? (mx + 255) / 256 : 1024);
  hipLaunchKernelGGL(adagrad_multi_kernel, dim3(bx, mt.n), dim3(256), 0, st, mt, scal, lr);
  return hipGetLastError();
}

static inline int gridn(int64_t n, int cap = 4096) { return (int)((n + 255) / 256 < cap ? (n + 255) / 256 : cap); }

hipError_t launch_loss(const float *src_raw, const float *tgt_raw, const float *labels, float *d_src, float *d_tgt,
                       float *row_loss, float *row_acc, float *out2, int B, int Bp, int S, float inv_rows, hipStream_t st) {
  LossArgs a{src_raw, tgt_raw, labels, d_src, d_tgt, row_loss, row_acc, B, Bp, S, inv_rows};
  hipLaunchKernelGGL(loss_kernel, dim3((Bp + 3) / 4), dim3(256), 0, st, a);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, row_acc, B, inv_rows, out2);
  return hipGetLastError();
}

// dh on the fp32 matrix pipe (round 5; the 64 x 64 LDS-tiled VALU kernel above took 0.12 ms for 8192 x 576 x 512, 40 TFLOP/s):
// D[b][j] = sum_s d[b][s] * M[j][s] -- both operands are read along s as they lie in memory: lane (row, k half) takes one
// float4 of its row, s0 + 4 * (k half) .. + 3, and component u of the two halves is the k pair of the u-th
// v_mfma_f32_32x32x2_f32 (both operands permute s identically).  A wave owns one 32 x 32 tile, a workgroup four j tiles of
// one b tile (the d fragment is the same for the four waves and comes from L1).  Needs S % 8 == 0 (16-byte aligned rows).
__global__ __launch_bounds__(256) void proj_bwd_dh_mfma_kernel(const float *__restrict__ d, const float *__restrict__ M, int Bp,
                                                               int H, int Hp, int S, float *__restrict__ dh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bt = blockIdx.y, jt = blockIdx.x * 4 + w;
  if (jt * 32 >= Hp) return;  // (no barriers in this kernel)
  const int row = lane & 31, kh = lane >> 5;
  const int b = bt * 32 + row, j = jt * 32 + row;
  const bool bok = b < Bp, jok = j < H;  // j >= H: weights read as 0 -> dh = 0
  const float *pa = d + (size_t)(bok ? b : 0) * S + 4 * kh;
  const float *pb = M + (size_t)(jok ? j : 0) * S + 4 * kh;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  constexpr int U = 4;  // 8 float4 loads in flight
  int s0 = 0;
  for (; s0 + 8 * U <= S; s0 += 8 * U) {
    f32x4 av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = *reinterpret_cast<const f32x4 *>(pa + s0 + 8 * u);
      bv[u] = *reinterpret_cast<const f32x4 *>(pb + s0 + 8 * u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bok ? av[u][e] : 0.0f, jok ? bv[u][e] : 0.0f, acc, 0, 0, 0);
  }
  for (; s0 < S; s0 += 8) {
    const f32x4 av = *reinterpret_cast<const f32x4 *>(pa + s0), bv = *reinterpret_cast<const f32x4 *>(pb + s0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bok ? av[e] : 0.0f, jok ? bv[e] : 0.0f, acc, 0, 0, 0);
  }
  const int jc = jt * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int br = bt * 32 + mfma_row(r, lane);
    if (br < Bp && jc < Hp) dh[(size_t)br * Hp + jc] = acc[r];
  }
}

int proj_bwd_chunks(int Bp) { return Bp <= 256 ? 1 : (Bp + 255) / 256 > 32 ? 32 : (Bp + 255) / 256; }

hipError_t launch_proj_bwd(const float *hT, const float *d, const float *M, int Bp, int H, int Hp, int S, float *dM,
                           float *dh, float *dm_part, hipStream_t st) {
  const int nch = proj_bwd_chunks(Bp);
  const int chunk = ((Bp + nch - 1) / nch + PB_K - 1) / PB_K * PB_K;
  if (Bp % 16 == 0)  // (always: rows are padded to whole 64-row tiles)
    hipLaunchKernelGGL(proj_bwd_dm_mfma_kernel, dim3((S + 127) / 128, (H + 31) / 32, nch), dim3(256), 0, st, hT, d, Bp, H, Hp, S,
                       chunk, nch == 1 ? dM : dm_part);
  else
    hipLaunchKernelGGL(proj_bwd_dm_kernel, dim3((S + PB_TILE - 1) / PB_TILE, (H + PB_TILE - 1) / PB_TILE, nch), dim3(256), 0, st,
                       hT, d, Bp, H, Hp, S, chunk, nch == 1 ? dM : dm_part);
  if (nch > 1)
    hipLaunchKernelGGL(proj_bwd_dm_reduce_kernel, dim3((H * S + 255) / 256), dim3(256), 0, st, dm_part, nch, H * S, dM);
  static const bool dh_valu = getenv("SSE_PROJ_DH_VALU") != nullptr;  // measurement aid: the LDS-tiled VALU kernel
  if (S % 8 == 0 && !dh_valu && (reinterpret_cast<uintptr_t>(d) & 15) == 0 && (reinterpret_cast<uintptr_t>(M) & 15) == 0)
    hipLaunchKernelGGL(proj_bwd_dh_mfma_kernel, dim3(((Hp + 31) / 32 + 3) / 4, (Bp + 31) / 32), dim3(256), 0, st, d, M, Bp, H, Hp, S, dh);  // (Hp is any multiple of 8 on the any-shape path)
  else
    hipLaunchKernelGGL(proj_bwd_dh_kernel, dim3((Hp + PB_TILE - 1) / PB_TILE, (Bp + PB_TILE - 1) / PB_TILE), dim3(256), 0, st, d,
                       M, Bp, H, Hp, S, dh);
  return hipGetLastError();
}

hipError_t launch_lstm_bwd(const float *tape_g, const float *dh_last, const float *KhT, float *dg_a, float *dg_b,
                           float *db_part, int T, int NT32, int NT_tape, int Hp, int H, int dg_b_split,
                           const unsigned short *KhT16, const BwdDxArgs *dx, hipStream_t st) {
  LstmBwdArgs a{tape_g, dh_last, KhT, dg_a, dg_b, db_part, T, NT32, Hp, H, KhT16, dg_b_split, NT_tape > 0 ? NT_tape : NT32,
                nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
#ifdef SSE_BWD_CLOCK
  static long long *clk_dev = nullptr;
  if (!clk_dev) (void)hipMalloc((void **)&clk_dev, 64 * sizeof(long long));
  (void)hipMemsetAsync(clk_dev, 0, 64 * sizeof(long long), st);
  a.clk = clk_dev;
  struct Report {
    long long *p;
    hipStream_t st;
    int T;
    ~Report() {
      long long v[64];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(v, p, sizeof v, hipMemcpyDeviceToHost);
      static int n = 0;
      if (n++ % 8 < 2)
        for (int w = 0; w < 8; w += 7)
          fprintf(stderr, "[bwd clock] wave %d cycles/step: elementwise %lld | barrier1 %lld | dump %lld | gemm %lld | refill-issue %lld | dX-loop %lld | dX-epilogue %lld | barrier2 %lld\n",
                  w, v[w * 8 + 0] / T, v[w * 8 + 1] / T, v[w * 8 + 2] / T, v[w * 8 + 3] / T, v[w * 8 + 4] / T, v[w * 8 + 5] / T, v[w * 8 + 6] / T, v[w * 8 + 7] / T);
    }
  } report{clk_dev, st, T};
#endif
  if (KhT16 != nullptr) {  // split-operand BPTT: dX is part of the kernel
    if (!dx || !dx->KxT16 || !dx->ids || !dx->d_emb || !dx->sq_part || !dx->hot_part || dx->E > 64) return hipErrorInvalidValue;
    a.KxT16 = dx->KxT16;
    a.ids = dx->ids;
    a.d_emb = dx->d_emb;
    a.sq_part = dx->sq_part;
    a.hot_part = dx->hot_part;
    a.B = dx->B;
    a.E = dx->E;
    a.V = dx->V;
  }
  const size_t lds = (size_t)(Hp / 2) * 256 * sizeof(float);
  if ((size_t)T * NT32 * (Hp / 32) * 5 * 1024 * sizeof(float) >= ((size_t)1 << 31)) return hipErrorInvalidValue;  // 32-bit tape offsets
  auto go = [&](auto kern, int threads) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(NT32), dim3(threads), lds, st, a);
    return hipGetLastError();
  };
  if (KhT16 != nullptr && !dg_b_split) return hipErrorInvalidValue;
  if (Hp == 128) {
    if (KhT16) return go(lstm_bwd_kernel<1, 4, true, true>, 256);
    return dg_b_split ? go(lstm_bwd_kernel<1, 4, true, false>, 256) : go(lstm_bwd_kernel<1, 4, false, false>, 256);
  }
  if (Hp == 256) {
    if (KhT16) return go(lstm_bwd_kernel<1, 8, true, true>, 512);
    return dg_b_split ? go(lstm_bwd_kernel<1, 8, true, false>, 512) : go(lstm_bwd_kernel<1, 8, false, false>, 512);
  }
  return hipErrorInvalidValue;
}

// Kh^T as split frag16 blocks for the split-operand BPTT: out[ub][g16][hi|lo][lane][i], lane (j = 32 ub + (lane & 31),
// half), reduction index n = 16 g16 + 8 half + i = gate * Hp + unit: K[(E + j)][gate * H + unit]
__global__ void pack_kT16_kernel(const float *__restrict__ K, int E, int H, int Hp, int64_t total, unsigned short *__restrict__ out) {
  const int KG16 = 4 * Hp / 16;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const int64_t blk = idx >> 9;
    const int g16 = (int)(blk % KG16), ub = (int)(blk / KG16);
    const int j = ub * 32 + (lane & 31), n = g16 * 16 + (lane >> 5) * 8 + i;
    const int gate = n / Hp, unit = n % Hp;
    const float v = (j < H && unit < H) ? K[(size_t)(E + j) * 4 * H + gate * H + unit] : 0.0f;
    const unsigned short hi = sse_bf16_rne(v), lo = sse_bf16_rne(v - sse_bf16_f32(hi));
    const int64_t o = (blk * 2) * 512 + lane * 8 + i;
    out[o] = hi;
    out[o + 512] = lo;
  }
}

// Kx^T as B fragments of v_mfma_f32_16x16x32_bf16 for the dX product inside the split-operand BPTT kernel:
// out[et][ks][hi|lo][lane][i], lane (e = 16 et + (lane & 15), n octet lane >> 4), n = 32 ks + 8 (lane >> 4) + i = gate * Hp + unit
__global__ void pack_kxT16_kernel(const float *__restrict__ K, int E, int H, int Hp, int64_t total, unsigned short *__restrict__ out) {
  const int KS = 4 * Hp / 32;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const int64_t blk = idx >> 9;
    const int ks = (int)(blk % KS), et = (int)(blk / KS);
    const int e = et * 16 + (lane & 15), n = ks * 32 + (lane >> 4) * 8 + i;
    const int gate = n / Hp, unit = n % Hp;
    const float v = (e < E && unit < H) ? K[(size_t)e * 4 * H + gate * H + unit] : 0.0f;
    const unsigned short hi = sse_bf16_rne(v), lo = sse_bf16_rne(v - sse_bf16_f32(hi));
    const int64_t o = (blk * 2) * 512 + lane * 8 + i;
    out[o] = hi;
    out[o + 512] = lo;
  }
}
size_t kxT16_elems(int Hp) { return (size_t)4 * (4 * Hp / 32) * 2 * 512; }
hipError_t launch_pack_kxT16(const float *K, int E, int H, int Hp, unsigned short *out, hipStream_t stream) {
  const int64_t total = (int64_t)4 * (4 * Hp / 32) * 512;
  hipLaunchKernelGGL(pack_kxT16_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, K, E, H, Hp, total, out);
  return hipGetLastError();
}

size_t kT16_elems(int Hp) { return (size_t)(Hp / 32) * (4 * Hp / 16) * 2 * 512; }

hipError_t launch_pack_kT16(const float *K, int E, int H, int Hp, unsigned short *out, hipStream_t stream) {
  const int64_t total = (int64_t)(Hp / 32) * (4 * Hp / 16) * 512;
  hipLaunchKernelGGL(pack_kT16_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, K, E, H, Hp, total, out);
  return hipGetLastError();
}

int dk_slices(int RG) {
  int sl = RG / 16;  // >= 16 r-groups (128 rows) per slice
  if (sl < 1) sl = 1;
  if (sl > 64) sl = 64;  // x NTn/8 n-groups (4 at H=256) = 256 workgroups
  return sl;
}

hipError_t launch_dk(const float *tape_a, const float *dg_b, float *part, int RG, int KT, int NTn, int SL, int E, int H,
                     int Hp, int accumulate, float *dK, int pair_rg, hipStream_t st, float *db) {
  if (db && E >= 64) return hipErrorInvalidValue;  // the constant-1 column sits at k' = E of the 64 x columns
  uint32_t live = 0;
  for (int i = 0; i < KT; ++i) {
    const bool on = (i < 2) ? (i * 32 < E + (db ? 1 : 0)) : ((i - 2) * 32 < H);
    if (on) live |= 1u << i;
  }
  DkArgs a{tape_a, dg_b, part, RG, KT, NTn, SL, live, pair_rg};
  const dim3 grid((NTn + 7) / 8, SL);
  static const bool one = getenv("SSE_DK_ONE") != nullptr;  // measurement aid: one r-group per barrier (the round-1 kernel)
  if (one) {
    if (KT == 10 && pair_rg) hipLaunchKernelGGL((dk_gemm_kernel<10, true>), grid, dim3(512), 0, st, a);
    else if (KT == 10) hipLaunchKernelGGL((dk_gemm_kernel<10, false>), grid, dim3(512), 0, st, a);
    else if (KT == 6 && pair_rg) hipLaunchKernelGGL((dk_gemm_kernel<6, true>), grid, dim3(512), 0, st, a);
    else if (KT == 6) hipLaunchKernelGGL((dk_gemm_kernel<6, false>), grid, dim3(512), 0, st, a);
    else return hipErrorInvalidValue;
  } else if (static const bool two = getenv("SSE_DK_TWO") != nullptr; !pair_rg && !two) {  // (pairs: the two-r-group kernel measured 0.76 against 0.78 ms)
    const int lds3 = 2 * 4 * KT * 1024;
    auto go3 = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
      hipLaunchKernelGGL(kern, grid, dim3(512), lds3, st, a);
    };
    if (KT == 10) go3(dk_gemm3_kernel<10, false>);  // (no pairs in this branch: the PAIR = true bodies are not instantiated)
    else if (KT == 6) go3(dk_gemm3_kernel<6, false>);
    else return hipErrorInvalidValue;
  } else {
    if (KT == 10 && pair_rg) hipLaunchKernelGGL((dk_gemm2_kernel<10, true>), grid, dim3(512), 0, st, a);
    else if (KT == 10) hipLaunchKernelGGL((dk_gemm2_kernel<10, false>), grid, dim3(512), 0, st, a);
    else if (KT == 6 && pair_rg) hipLaunchKernelGGL((dk_gemm2_kernel<6, true>), grid, dim3(512), 0, st, a);
    else if (KT == 6) hipLaunchKernelGGL((dk_gemm2_kernel<6, false>), grid, dim3(512), 0, st, a);
    else return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(dk_reduce_kernel, dim3(((E + H + (db ? 1 : 0)) * 4 * H + 255) / 256), dim3(256), 0, st, part, SL, KT, NTn, E, H, Hp,
                     accumulate, dK, db);
  return hipGetLastError();
}

// dK on the bf16 matrix pipe: tape_a / dg_b in split frag16 form (tape_a_split / dg_b_split of the producing kernels);
// G = 16-row groups of tape_a, pair_g as pair_rg of launch_dk in 16-row groups
hipError_t launch_dk_x3(const void *tape_a, const void *dg_b, float *part, int G, int KT, int NTn, int SL, int E, int H,
                        int Hp, int accumulate, float *dK, int pair_g, hipStream_t st) {
  uint32_t live = 0;
  for (int i = 0; i < KT; ++i) {
    const bool on = (i < 2) ? (i * 32 < E) : ((i - 2) * 32 < H);
    if (on) live |= 1u << i;
  }
  DkX3Args a{(const unsigned short *)tape_a, (const unsigned short *)dg_b, part, G, KT, NTn, SL, live, pair_g};
  const dim3 grid((NTn + 7) / 8, SL);
  if (KT == 10 && pair_g) hipLaunchKernelGGL((dk_x3_kernel<10, true>), grid, dim3(512), 0, st, a);
  else if (KT == 10) hipLaunchKernelGGL((dk_x3_kernel<10, false>), grid, dim3(512), 0, st, a);
  else if (KT == 6 && pair_g) hipLaunchKernelGGL((dk_x3_kernel<6, true>), grid, dim3(512), 0, st, a);
  else if (KT == 6) hipLaunchKernelGGL((dk_x3_kernel<6, false>), grid, dim3(512), 0, st, a);
  else return hipErrorInvalidValue;
  hipLaunchKernelGGL(dk_reduce_kernel, dim3(((E + H) * 4 * H + 255) / 256), dim3(256), 0, st, part, SL, KT, NTn, E, H, Hp,
                     accumulate, dK, (float *)nullptr);
  return hipGetLastError();
}

hipError_t launch_db_reduce(const float *db_part, int NT32, int H, int Hp, int accumulate, float *db, hipStream_t st) {
  hipLaunchKernelGGL(db_reduce_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, db_part, NT32, H, Hp, accumulate, db);
  return hipGetLastError();
}

hipError_t launch_dx(const float *dg_a, const float *KxT, const int32_t *ids, float *d_emb, float *sq_part,
                     float *hot_part /* [T*NT32][2][64] */, int T, int NT32, int KGn, int B, int E, int V, int H,
                     hipStream_t st) {
  const int KGg = KGn / 4;
  DxArgs a{dg_a, KxT, ids, d_emb, sq_part, T, NT32, KGn, B, E, V, (H + 7) / 8 < KGg ? (H + 7) / 8 : KGg, hot_part};
  hipLaunchKernelGGL(dx_kernel, dim3(T * NT32), dim3(64), 0, st, a);
  hipLaunchKernelGGL(dx_hot_reduce_kernel, dim3(2, DXH_SLICES), dim3(256), 0, st, hot_part, T * NT32, E, V, d_emb);
  return hipGetLastError();
}

hipError_t launch_dx_hot_reduce(const float *hot_part, int nblocks, int E, int V, float *d_emb, hipStream_t st) {
  hipLaunchKernelGGL(dx_hot_reduce_kernel, dim3(2, DXH_SLICES), dim3(256), 0, st, hot_part, nblocks, E, V, d_emb);
  return hipGetLastError();
}

hipError_t launch_sumsq(const float *g, int64_t n, float *part, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks), dim3(256), 0, st, g, n, part);
  return hipGetLastError();
}

hipError_t launch_sum(const float *part, int n, float tag, float *out, hipStream_t st) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, part, n, tag, out);
  return hipGetLastError();
}

hipError_t launch_clip_scale(const float *part, int n, float clip, float *scal, hipStream_t st) {
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(256), 0, st, part, n, clip, scal);
  return hipGetLastError();
}

hipError_t launch_adagrad(float *w, float *accum, const float *grad, const float *scal, float lr, int64_t n,
                          hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(adagrad_kernel, dim3(gridn(n)), dim3(256), 0, st, w, accum, grad, scal, lr, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// batch token ids from the device-resident corpus: one wave per batch row
__global__ void gather_id_rows_kernel(const int32_t *__restrict__ corpus, const int32_t *__restrict__ rows, int B, int T,
                                      int64_t N, int32_t *__restrict__ out, int32_t *err) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  int64_t r = rows[b];
  if (r < 0 || r >= N) {
    if (lane == 0) atomicOr(err, 2);
    r = 0;
  }
  for (int t = lane; t < T; t += 64) out[(size_t)b * T + t] = corpus[(size_t)r * T + t];
}

hipError_t launch_gather_id_rows(const int32_t *corpus, const int32_t *rows, int B, int T, int64_t N, int32_t *out,
                                 int32_t *err, hipStream_t st) {
  hipLaunchKernelGGL(gather_id_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, st, corpus, rows, B, T, N, out, err);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Data-parallel exchange of the word-embedding gradient as (row id, gradient row) pairs (SURVEY 8e "Training": all-gather
// of the sparse embedding gradients instead of all-reducing the dense [V,E] block; data_parallel.py).
// packed = [count (int32), 0, 0, 0 | ids[cap4] (int32) | rows[cap][E]], cap4 = cap rounded up to 4.
// Pack: one wave per embedding row; a row with any non-zero gradient takes the next slot (atomic counter: the order of the
// slots is not deterministic, the SET is, and the ids of one rank are distinct -- the sums below do not depend on it).
__global__ __launch_bounds__(256) void emb_grad_pack_kernel(const float *__restrict__ g, int V, int E, int cap, float *__restrict__ packed,
                                                            int32_t *err) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= V) return;
  const float *src = g + (size_t)row * E;
  bool nz = false;
  for (int e = lane; e < E; e += 64) nz = nz || src[e] != 0.0f;
  if (__ballot(nz) == 0) return;
  int slot = 0;
  if (lane == 0) slot = atomicAdd(reinterpret_cast<int32_t *>(packed), 1);
  slot = __shfl(slot, 0);
  if (slot >= cap) {  // cannot happen while cap >= min(V, ids of this rank's batch); never write past the buffer
    if (lane == 0) atomicOr(err, 8);
    return;
  }
  const int cap4 = (cap + 3) & ~3;
  if (lane == 0) reinterpret_cast<int32_t *>(packed)[4 + slot] = row;
  float *dst = packed + 4 + cap4 + (size_t)slot * E;
  for (int e = lane; e < E; e += 64) dst[e] = src[e];
}

// Unpack one rank's buffer: g[id] += row for its count slots (distinct ids: no atomics).  Launched once per rank, in rank
// order, after the block has been zeroed: the same sums in the same order on every rank.
__global__ __launch_bounds__(256) void emb_grad_unpack_kernel(const float *__restrict__ packed, int V, int E, int cap, float *__restrict__ g,
                                                              int32_t *err) {
  const int sent = reinterpret_cast<const int32_t *>(packed)[0];
  // a peer whose pack overflowed (count > cap) raised bit 8 on ITS err word only and cancels its own update: raise it here
  // too, so every rank cancels the step together instead of applying a truncated gradient and drifting apart
  if ((sent > cap || sent < 0) && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, 8);
  const int count = sent < 0 ? 0 : min(sent, cap);
  const int cap4 = (cap + 3) & ~3;
  const int64_t n = (int64_t)count * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(i / E), e = (int)(i - (int64_t)slot * E);
    const int id = reinterpret_cast<const int32_t *>(packed)[4 + slot];
    if (id < 0 || id >= V) {
      atomicOr(err, 8);
      continue;
    }
    g[(size_t)id * E + e] += packed[4 + cap4 + (size_t)slot * E + e];
  }
}

int64_t emb_grad_packed_floats(int E, int cap) { return 4 + (int64_t)((cap + 3) & ~3) + (int64_t)cap * E; }

hipError_t launch_emb_grad_pack(const float *g, int V, int E, int cap, float *packed, int32_t *err, hipStream_t st) {
  hipError_t e = hipMemsetAsync(packed, 0, 16, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(emb_grad_pack_kernel, dim3((V + 3) / 4), dim3(256), 0, st, g, V, E, cap, packed, err);
  return hipGetLastError();
}

hipError_t launch_emb_grad_unpack(const float *gathered, int world, int V, int E, int cap, float *g, int32_t *err, hipStream_t st) {
  hipError_t e = hipMemsetAsync(g, 0, (size_t)V * E * sizeof(float), st);
  if (e != hipSuccess) return e;
  const int64_t stride = emb_grad_packed_floats(E, cap);
  const int64_t work = (int64_t)cap * E;
  const int blocks = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  for (int r = 0; r < world; ++r)
    hipLaunchKernelGGL(emb_grad_unpack_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, gathered + (size_t)r * stride, V, E, cap, g, err);
  return hipGetLastError();
}
