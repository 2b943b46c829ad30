// Backward pass of the text-CNN encoder (network_mode 'source_only_cnn') for gfx950.
//
// BUILDER-DEFINED training path (BASELINE configs[4]): the reference's CNN graph does not build
// (sse_model.py:206) and its loss is ill-shaped for this mode (:213-214,290); the pair loss of
// :279-302 is applied with the target side read as embedding_lookup(tgt_seq_embedding, rows)
// (oracle/sse_oracle.py::_cnn_gradients is the restatement these kernels are checked against).
//
// After max-over-time pooling only ONE position per (sequence, filter) carries gradient, so the
// convolution backward is not a GEMM: it is 576 weighted window copies per sequence (~100 k MAC,
// 2 % of the forward).  These kernels are gather/scatter-shaped and LDS/HBM-latency bound:
//   cnn_dw_kernel   dW[k][f] = sum_b g[b][f] * window(b, pos[b][f])[k], db[f] = sum_b g[b][f];
//                   a thread owns (filter, k-range) accumulators in registers, sequences are staged
//                   through LDS (double-buffered), the batch is split into chunks whose partials are
//                   summed in fixed order (deterministic);
//   cnn_dx_kernel   dX[b][t][:] += g[b][f] * W[:, f] over the winning windows -> LDS tile (ds_add_f32),
//                   then one float atomicAdd per element into the dense d word_embedding, and
//                   sum dX^2 per sequence for TF's global norm over the raw IndexedSlices;
//   rows_gather / rows_scatter: target-table lookup and its gradient.
#include "sse_kernels.h"
#include "train.h"
#include <cstdlib>

namespace {

__constant__ int b_fs[4] = {2, 3, 4, 5};
__constant__ int b_foff[4] = {0, 256, 384, 512};

struct CnnBwdArgs {
  const int32_t *ids;   // [B][T]
  const float *emb;     // [V][E] master copy
  const float *dfeat;   // [Bp][576] d loss / d pooled features
  const float *feat;    // [Bp][576] pooled features (ReLU mask: > 0)
  const int32_t *pos;   // [B][576] arg-max positions
  const float *W[4];    // master filters, row-major [fs*E][nf]
  const float *Wt[4];   // transposed copies [nf][fs*E] (made per step by launch_cnn_bwd)
  float *dw_part[4];    // [NCH][fs*E*nf]
  float *db_part;       // [NCH][576]
  float *d_emb;         // [V][E] dense embedding gradient (zeroed by the caller)
  float *sq_part;       // [B]
  float *hot_part;      // cnn_dx_kernel: [B][2][64] per-sequence dX sums for token ids 0 (PAD) and 1 (EOS), or null
  int32_t B, T, E, NCH, V;  // V: token ids outside [0, V) read row 0 / add nothing (the forward raised the error flag: the update is cancelled)
  int32_t bf16;         // option cnn_bf16: the forward ran on bf16-rounded embeddings / filters; the backward
                        // differentiates THAT function (dW from the rounded windows, dX from the rounded filters)
};

__device__ __forceinline__ float bf16_rne(float f) {  // nearest bfloat16 (ties to even), as fp32 -- cnn_fwd_bf16.hip
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xFFFF0000u);
}

// Software pipeline of the per-sequence staging (round 5; the arithmetic and its order are unchanged -- results bit-identical
// to the round-2 kernel): the gathered embedding values of sequence b+1 and the token ids of sequence b+2 are in flight in
// registers while sequence b is multiplied out of LDS, so the two dependent global loads (ids -> embedding row) of a
// sequence no longer sit between two barriers.  0.55 -> 0.1x ms at 8192 sequences (profiles/r05_notes.txt).
constexpr int DW_THREADS = 512, DW_WAVES = DW_THREADS / 64;
constexpr int DW_G = 2;  // sequences staged per barrier

template <int FS, int NF, int RMAX, bool X16>
__device__ __forceinline__ void dw_body(const CnnBwdArgs &a, float *xs) {
  // Round 5, second version: LANE = k.  A wave owns FPW = NF / 8 filters; for filter f of sequence b its 64 lanes read 64
  // CONSECUTIVE elements of the winning window (one conflict-free LDS read per 64 k; g and the position are wave-uniform:
  // v_readlane) and add g * x into acc[filter][k chunk].  The first version (thread = filter, registers = k) had 64 lanes
  // reading 64 different rows: bank conflicts set the pace (0.20 - 0.27 ms at 8192 sequences).  Each (k, f) is still summed over
  // the chunk's sequences in order and the chunks in order: results bit-identical to every earlier version.
  constexpr int FPW = NF / DW_WAVES;                 // 32, 16, 16, 8
  constexpr int EPL = X16 ? 2 : 1;                   // elements per lane and read (a dword of the bf16 tile holds two)
  constexpr int NCK = (FS * 64 + 64 * EPL - 1) / (64 * EPL);  // k chunks of 64 lanes (E <= 64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int T = a.T, E = a.E, K = FS * E, TE = T * E;
  // X16 (option cnn_bf16, even E): the tile holds the bf16-rounded embeddings AS bf16 -- half the LDS bytes; a dword unpacks
  // into two exact fp32 values.
  unsigned short *xs16 = reinterpret_cast<unsigned short *>(xs);
  const int wi = FS - 2;
  const int chunk = blockIdx.x, per = (a.B + a.NCH - 1) / a.NCH;
  const int b_begin = chunk * per, b_end = min(a.B, b_begin + per);
  float acc[FPW][NCK][EPL];
#pragma unroll
  for (int i = 0; i < FPW; ++i)
#pragma unroll
    for (int c = 0; c < NCK; ++c)
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[i][c][e] = 0.0f;
  float bsum = 0.0f;  // lane fl < FPW: d bias of this wave's filter fl
  // Staging, third version.  Ablation of the second (one sequence per barrier, 16 waves; profiles/r05_notes.txt): with the
  // multiply AND every global load switched off the kernel still took half its time -- the per-sequence barrier and the loop
  // skeleton, ~1 us per iteration.  Now DW_G (= 2, see above; 4 was measured first) sequences per barrier: the chunk's token ids sit in LDS (staged once,
  // validated), the next group's embedding values and gradient / mask / position words are loaded at the top of a group step
  // and stored / taken over at its bottom, four multiplies later.
  // This thread's elements of a [T][E] tile: column ce = lane (idle for ce >= E), tokens tq, tq + 8, .. with tq = its wave.
  const int ce = tid & 63, tq = __builtin_amdgcn_readfirstlane(tid >> 6), w = tq;
  constexpr int RM = RMAX > 0 ? RMAX : 1;
  int *s_ids = reinterpret_cast<int *>(xs + (2 * DW_G * TE) / (X16 ? 2 : 1) + 384);  // [per][T] (RMAX > 0 only; TE even in X16 mode)
  const int fo = b_foff[wi] + w * FPW + (lane < FPW ? lane : 0);    // lane fl holds filter fl's gradient / position
  // (Slots past the window -- k >= K -- read whatever follows in LDS, the next rows / tiles / the pad behind them, into
  // accumulators that are never stored: no per-slot range test, which once made the compiler emit one ds_read -> s_waitcnt ->
  // v_fmac block PER SLOT.  No "if (g != 0)" either: a branch per filter exposes one LDS round trip per filter; a masked filter
  // adds 0 * (finite window data).)
  auto multiply = [&](int tile, float g_l, float f_l, int pl) {
    const float gl = (f_l > 0.0f && lane < FPW) ? g_l : 0.0f;
    bsum += gl;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl), i));
      const int p = __builtin_amdgcn_readlane(pl, i);
      if constexpr (X16) {
        const unsigned *xw = reinterpret_cast<const unsigned *>(xs16 + tile * TE + p * E) + lane;  // (E even: dword aligned)
#pragma unroll
        for (int c = 0; c < NCK; ++c) {
          const unsigned v = xw[c * 64];
          acc[i][c][0] += g * __uint_as_float(v << 16);
          acc[i][c][1] += g * __uint_as_float(v & 0xFFFF0000u);
        }
      } else {
        const float *xw = xs + tile * TE + p * E + lane;
#pragma unroll
        for (int c = 0; c < NCK; ++c) acc[i][c][0] += g * xw[c * 64];
      }
    }
  };
  if constexpr (RMAX == 0) {
    // any T (more than DW_WAVES * 12 = 96 tokens) or a chunk whose ids do not fit LDS: the plain staging loop, one
    // sequence between two barriers
    int buf = 0;
    for (int b = b_begin; b < b_end; ++b, buf ^= 1) {
      for (int i = tid; i < TE; i += DW_THREADS) {
        int id = a.ids[(size_t)b * T + i / E];
        if (id < 0 || id >= a.V) id = 0;
        const float x = a.emb[(size_t)id * E + i % E];
        if constexpr (X16) xs16[buf * TE + i] = (unsigned short)(__float_as_uint(bf16_rne(x)) >> 16);
        else xs[buf * TE + i] = a.bf16 ? bf16_rne(x) : x;
      }
      const float g_l = a.dfeat[(size_t)b * 576 + fo], f_l = a.feat[(size_t)b * 576 + fo];
      const int pl = a.pos[(size_t)b * 576 + fo];
      __syncthreads();
      multiply(buf, g_l, f_l, pl);
    }
  } else {
    const int nseq = b_end - b_begin, NG = (nseq + DW_G - 1) / DW_G;
    for (int i = tid; i < nseq * T; i += DW_THREADS) {
      const int id = a.ids[(size_t)b_begin * T + i];
      s_ids[i] = (id < 0 || id >= a.V) ? 0 : id;  // (the forward raised the error flag: the update is cancelled)
    }
    __syncthreads();
    float vq[DW_G][RM], gc[DW_G], fc[DW_G], gn[DW_G], fn[DW_G];
    int pc[DW_G], pn[DW_G];
    auto load_group = [&](int grp) {  // values of the group's sequences + their gradient words, all in flight together
#pragma unroll
      for (int q = 0; q < DW_G; ++q) {
        const int sq = grp * DW_G + q;
        if (sq < nseq) {
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (tq + DW_WAVES * r < T && ce < E) vq[q][r] = a.emb[(size_t)s_ids[sq * T + tq + DW_WAVES * r] * E + ce];
          gn[q] = a.dfeat[(size_t)(b_begin + sq) * 576 + fo];
          fn[q] = a.feat[(size_t)(b_begin + sq) * 576 + fo];
          pn[q] = a.pos[(size_t)(b_begin + sq) * 576 + fo];
        } else {
          gn[q] = 0.0f;
          fn[q] = 0.0f;
          pn[q] = 0;
        }
      }
    };
    auto store_group = [&](int grp, int buf) {
#pragma unroll
      for (int q = 0; q < DW_G; ++q) {
        if (grp * DW_G + q < nseq) {
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (tq + DW_WAVES * r < T && ce < E) {
              const int at = (buf * DW_G + q) * TE + (tq + DW_WAVES * r) * E + ce;
              if constexpr (X16) xs16[at] = (unsigned short)(__float_as_uint(bf16_rne(vq[q][r])) >> 16);
              else xs[at] = a.bf16 ? bf16_rne(vq[q][r]) : vq[q][r];
            }
        }
        gc[q] = gn[q];
        fc[q] = fn[q];
        pc[q] = pn[q];
      }
    };
    if (NG > 0) {
      load_group(0);
      store_group(0, 0);
    }
    __syncthreads();
    int buf = 0;
    for (int grp = 0; grp < NG; ++grp, buf ^= 1) {
      if (grp + 1 < NG) load_group(grp + 1);
      float g0[DW_G], f0[DW_G];
      int p0[DW_G];
#pragma unroll
      for (int q = 0; q < DW_G; ++q) {
        g0[q] = gc[q];
        f0[q] = fc[q];
        p0[q] = pc[q];
      }
#pragma unroll
      for (int q = 0; q < DW_G; ++q)
        if (grp * DW_G + q < nseq) multiply(buf * DW_G + q, g0[q], f0[q], p0[q]);
      if (grp + 1 < NG) store_group(grp + 1, buf ^ 1);  // (those tiles were last read before the previous barrier)
      __syncthreads();
    }
  }
  // partials [chunk][filter][k] (k contiguous: coalesced stores; cnn_reduce_kernel reads them the same way)
  float *out = a.dw_part[wi] + (size_t)chunk * K * NF;
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    float *row = out + (size_t)(w * FPW + i) * K;
#pragma unroll
    for (int c = 0; c < NCK; ++c)
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int k = (c * 64 + lane) * EPL + e;
        if (k < K) row[k] = acc[i][c][e];
      }
  }
  if (lane < FPW) a.db_part[(size_t)chunk * 576 + b_foff[wi] + w * FPW + lane] = bsum;
}

// (the four bodies are inlined: as calls they took the argument block through scratch and spilled around the call)
template <int RMAX, bool X16>
__global__ __launch_bounds__(DW_THREADS) void cnn_dw_kernel(CnnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [2][DW_G][T*E] fp32 (or bf16: X16) + 384 floats of pad (read, never used: see multiply) + ids
  switch (blockIdx.y) {
    case 0: dw_body<2, 256, RMAX, X16>(a, xs); break;
    case 1: dw_body<3, 128, RMAX, X16>(a, xs); break;
    case 2: dw_body<4, 128, RMAX, X16>(a, xs); break;
    default: dw_body<5, 64, RMAX, X16>(a, xs); break;
  }
}

// dW (four widths) and db from the chunk partials in ONE launch (round 5: eight launches before, four of them single
// workgroups): thread i sums its element over the chunks in chunk order -- the order of the kernels it replaces.
struct CnnReduceArgs {
  const float *dw_part[4];
  float *dW[4];
  const float *db_part;
  float *db[4];
  int32_t n[4], nch;
};
__global__ void cnn_reduce_kernel(CnnReduceArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (i < a.n[w]) {  // i = f * K + k in the partials' layout [chunk][filter][k]; dW is [k][filter]
      float acc = 0.0f;
      for (int c = 0; c < a.nch; ++c) acc += a.dw_part[w][(size_t)c * a.n[w] + i];
      const int nf = w == 0 ? 256 : w == 3 ? 64 : 128, K = a.n[w] / nf;
      a.dW[w][(size_t)(i % K) * nf + i / K] = acc;
      return;
    }
    i -= a.n[w];
  }
  if (i < 576) {
    float acc = 0.0f;
    for (int c = 0; c < a.nch; ++c) acc += a.db_part[(size_t)c * 576 + i];
    const int w = i < 256 ? 0 : i < 384 ? 1 : i < 512 ? 2 : 3;
    a.db[w][i - b_foff[w]] = acc;
  }
}

// dX[t][e] = sum over the filters whose winning window covers t.  LDS float atomics run ~1 lane/clock on
// gfx950 (a scatter formulation took 3.5 ms at 8192 sequences), so this is a GATHER: the 576 filters are
// bucketed by winning position (ascending filter order: deterministic), thread (t,e) walks the buckets of
// positions t, t-1, .. t-4 and reads the TRANSPOSED filters Wt[f][j*E+e] (lanes = e: coalesced).
__global__ __launch_bounds__(256) void cnn_dx_kernel(CnnBwdArgs a) {
  __shared__ float g_s[576];
  __shared__ int pos_s[576], list_s[576];
  extern __shared__ int start_s[];  // [T + 1] bucket starts (dynamic: any T)
  __shared__ int woff_s[576], fs_s[576];  // per list entry: row offset into the transposed filters, width
  __shared__ float gl_s[576];
  __shared__ float red[256];
  const int tid = threadIdx.x, b = blockIdx.x, T = a.T, E = a.E;
  for (int f = tid; f < 576; f += 256) {
    float g = a.dfeat[(size_t)b * 576 + f];
    if (!(a.feat[(size_t)b * 576 + f] > 0.0f)) g = 0.0f;
    g_s[f] = g;
    pos_s[f] = (g != 0.0f) ? a.pos[(size_t)b * 576 + f] : -1;
  }
  __syncthreads();
  // bucket sizes, then (thread 0) their exclusive prefix, then the fill -- every bucket in ascending filter order
  for (int p = tid; p < T; p += 256) {
    int cnt = 0;
    for (int f = 0; f < 576; ++f) cnt += (pos_s[f] == p);
    start_s[p + 1] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    start_s[0] = 0;
    for (int p = 0; p < T; ++p) start_s[p + 1] += start_s[p];
  }
  __syncthreads();
  for (int p = tid; p < T; p += 256) {
    int st = start_s[p];
    for (int f = 0; f < 576; ++f)
      if (pos_s[f] == p) list_s[st++] = f;
  }
  __syncthreads();
  const int n_list = start_s[T];
  for (int x = tid; x < n_list; x += 256) {
    const int f = list_s[x];
    const int wi = f < 256 ? 0 : f < 384 ? 1 : f < 512 ? 2 : 3;
    woff_s[x] = (int)(a.Wt[wi] - a.Wt[0]) + (f - b_foff[wi]) * (b_fs[wi] * E);
    fs_s[x] = b_fs[wi];
    gl_s[x] = g_s[f];
  }
  __syncthreads();
  // PAD (0) / EOS (1) rows (most rows of a left-padded batch, the last of every sequence): summed per sequence in LDS, added
  // to the dense gradient by dx_hot_reduce_kernel in fixed order -- as global atomics they all hit the same 2 x E addresses
  __shared__ float s_hot[2][64];
  if (tid < 128) s_hot[tid >> 6][tid & 63] = 0.0f;
  __syncthreads();
  const float *Wt = a.Wt[0];
  float sq = 0.0f;
  for (int i = tid; i < T * E; i += 256) {
    const int t = i / E, e = i % E;
    float acc = 0.0f;
    for (int j = 0; j < 5 && j <= t; ++j) {
      const int p = t - j, i1 = start_s[p + 1], je = j * E + e;
#pragma unroll 4
      for (int x = start_s[p]; x < i1; ++x) {  // branch-free: filters narrower than j+1 contribute 0 * (a valid element)
        const bool in = j < fs_s[x];
        acc += (in ? gl_s[x] : 0.0f) * Wt[woff_s[x] + (in ? je : e)];
      }
    }
    sq += acc * acc;
    const int id = a.ids[(size_t)b * T + t];
    if (acc != 0.0f && id >= 0 && id < a.V) {
      if (id < 2 && a.hot_part) atomicAdd(&s_hot[id][e], acc);
      else atomicAdd(a.d_emb + (size_t)id * E + e, acc);
    }
  }
  red[tid] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) a.sq_part[b] = red[0];
  if (a.hot_part && tid < 128) a.hot_part[((size_t)b * 2 + (tid >> 6)) * 64 + (tid & 63)] = s_hot[tid >> 6][tid & 63];  // (after the reduction's barriers)
}

// Wt[f][k] = W[k][f]  (rounded to bf16 when the forward used the rounded filters)
__global__ void transpose_kernel(const float *W, int K, int NF, int bf16, float *Wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K * NF) Wt[(size_t)(i % NF) * K + i / NF] = bf16 ? bf16_rne(W[i]) : W[i];
}

// out[b][:] = table[rows[b]][:] (b < B), 0 for the padding rows; one wave per row
__global__ void rows_gather_kernel(const float *table, const int32_t *rows, int B, int Bp, int N, int S, float *out,
                                   int32_t *err) {
  const int lane = threadIdx.x & 63, b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= Bp) return;
  int r = (b < B) ? rows[b] : -1;
  if (b < B && (r < 0 || r >= N)) {
    if (lane == 0) atomicOr(err, 1);
    r = -1;
  }
  for (int s = lane; s < S; s += 64) out[(size_t)b * S + s] = (r >= 0) ? table[(size_t)r * S + s] : 0.0f;
}

// d_table[rows[b]][:] += d[b][:]; sq[b] = |d[b]|^2 (raw slice norm)
__global__ void rows_scatter_kernel(const float *d, const int32_t *rows, int B, int S, int N, float *d_table, float *sq) {
  const int lane = threadIdx.x & 63, b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (b >= B) return;
  const int r = rows[b];
  float acc = 0.0f;
  for (int s = lane; s < S; s += 64) {
    const float v = d[(size_t)b * S + s];
    acc += v * v;
    if (r >= 0 && r < N) atomicAdd(d_table + (size_t)r * S + s, v);  // (a row out of range raised the error flag in rows_gather)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) sq[b] = acc;
}

}  // namespace

int cnn_bwd_chunks(int B) {
  int n = (B + 31) / 32;
  return n < 1 ? 1 : n > 128 ? 128 : n;
}

size_t cnn_dw_part_floats(int E, int B) { return (size_t)cnn_bwd_chunks(B) * (size_t)E * 1728; }  // sum fs*nf = 1728

hipError_t launch_rows_gather(const float *table, const int32_t *rows, int B, int Bp, int N, int S, float *out,
                              int32_t *err, hipStream_t st) {
  hipLaunchKernelGGL(rows_gather_kernel, dim3((Bp + 3) / 4), dim3(256), 0, st, table, rows, B, Bp, N, S, out, err);
  return hipGetLastError();
}

hipError_t launch_rows_scatter(const float *d, const int32_t *rows, int B, int S, int N, float *d_table, float *sq,
                               hipStream_t st) {
  hipLaunchKernelGGL(rows_scatter_kernel, dim3((B + 3) / 4), dim3(256), 0, st, d, rows, B, S, N, d_table, sq);
  return hipGetLastError();
}

hipError_t launch_cnn_bwd(const int32_t *ids, const float *emb, const float *dfeat, const float *feat, const int32_t *pos,
                          const float *const W[4], float *const dW[4], float *const db[4], float *dw_part,
                          float *db_part, float *wt_scratch /* [E*1728] */, unsigned short *wct_scratch /* cnn_wct_elems(E) */,
                          float *d_emb, float *sq_part, float *hot_part /* [max(B, cnn_dx_mfma_blocks(B))][2][64] */, int B, int T, int E, int V, int bf16,
                          hipStream_t st) {
  static const int fs[4] = {2, 3, 4, 5}, nf[4] = {256, 128, 128, 64};
  if (E > 64) return hipErrorInvalidValue;
  CnnBwdArgs a;
  a.ids = ids;
  a.emb = emb;
  a.dfeat = dfeat;
  a.feat = feat;
  a.pos = pos;
  a.db_part = db_part;
  a.d_emb = d_emb;
  a.sq_part = sq_part;
  a.hot_part = hot_part;
  a.B = B;
  a.T = T;
  a.E = E;
  a.NCH = cnn_bwd_chunks(B);
  a.V = V;
  a.bf16 = bf16;
  // bf16 mode: dX as a dense contraction on the bf16 matrix pipe (cnn_bwd_mfma.hip); fp32 mode (and sequences longer than
  // its three t tiles): the gather kernel over the transposed filters
  static const bool no_mfma = getenv("SSE_CNN_DX_GATHER") != nullptr;  // measurement aid: the gather kernel in bf16 mode too
  const bool dx_mfma = bf16 && wct_scratch && hot_part && cnn_dx_mfma_ok(T, E) && !no_mfma;
  CnnReduceArgs ra;
  size_t off = 0, toff = 0;
  int total = 576;
  for (int i = 0; i < 4; ++i) {
    const int n = fs[i] * E * nf[i];
    a.W[i] = W[i];
    a.Wt[i] = wt_scratch + toff;
    if (!dx_mfma)
      hipLaunchKernelGGL(transpose_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W[i], fs[i] * E, nf[i], bf16, wt_scratch + toff);
    toff += n;
    a.dw_part[i] = dw_part + off;
    off += (size_t)a.NCH * n;
    ra.dw_part[i] = a.dw_part[i];
    ra.dW[i] = dW[i];
    ra.db[i] = db[i];
    ra.n[i] = n;
    total += n;
  }
  ra.db_part = db_part;
  ra.nch = a.NCH;
  const int per = (B + a.NCH - 1) / a.NCH;
  // LDS: [2 buffers][DW_G tiles][T*E] (fp32, or bf16 in X16 mode: the fp32 size is reserved either way) + 384 floats of pad (a
  // window read runs up to 5 * 64 elements past its start) + the chunk's token ids
  const bool x16 = bf16 && (E & 1) == 0;
  const size_t lds_tiles = ((size_t)2 * DW_G * T * E / (x16 ? 2 : 1) + 384) * sizeof(float), lds_ids = (size_t)per * T * sizeof(int32_t);
  const bool ids_fit = lds_tiles + lds_ids <= (size_t)150 * 1024;
  const size_t lds = ids_fit ? lds_tiles + lds_ids : ((size_t)2 * T * E + 384) * sizeof(float);
  // the dW kernel's own LDS need, checked here (a T * E that passed the forward's check but needs more than 160 KiB here
  // used to surface as a bare launch error); cnn_train_grads_locked reports it with the limit in the message
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  hipError_t attr_err = hipSuccess;
  auto go = [&](auto kern) {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_err == hipSuccess) hipLaunchKernelGGL(kern, dim3(a.NCH, 4), dim3(DW_THREADS), lds, st, a);
  };
  if (x16) {
    if (ids_fit && T <= DW_WAVES * 8) go(cnn_dw_kernel<8, true>);
    else if (ids_fit && T <= DW_WAVES * 12) go(cnn_dw_kernel<12, true>);
    else go(cnn_dw_kernel<0, true>);
  } else {
    if (ids_fit && T <= DW_WAVES * 8) go(cnn_dw_kernel<8, false>);
    else if (ids_fit && T <= DW_WAVES * 12) go(cnn_dw_kernel<12, false>);
    else go(cnn_dw_kernel<0, false>);
  }
  if (attr_err != hipSuccess) return attr_err;
  hipLaunchKernelGGL(cnn_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ra);
  if (dx_mfma) return launch_cnn_dx_mfma(ids, dfeat, feat, pos, W, wct_scratch, d_emb, sq_part, hot_part, B, T, E, V, st);
  hipLaunchKernelGGL(cnn_dx_kernel, dim3(B), dim3(256), (size_t)(T + 1) * sizeof(int), st, a);
  if (hot_part) return launch_dx_hot_reduce(hot_part, B, E, V, d_emb, st);
  return hipGetLastError();
}
