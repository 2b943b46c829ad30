// LSTM sequence encoder forward for MID-SIZE batches (33 .. 1024 sequences): the evaluator's batches of 600
// (sse_evaluator.py:104-109), the index builder's batches of 1000 (sse_train.py:226, sse_index.py:66,90-92).
//
// Why a third small-batch kernel.  A 32-row matrix tile (lstm_fwd.hip) puts the whole [(E+H) x 4H] gate GEMM of its rows on
// ONE compute unit: 21 MFLOP per step at the fp32 MFMA rate of one CU (0.61 TF) = 34 us per step -- 1.2 ms for T = 32
// whether the launch holds 33 rows or 8192, with 600 rows keeping 19 of the 256 CUs busy.  lstm_small.hip streams the kernel
// matrix through every workgroup of 4 rows (17 us per step, 0.55 ms).  Here the hidden units of a row tile are spread over a
// CLUSTER of 16 compute units, the MFMA version of lstm_persist.hip:
//   * a cluster serves 64 sequences; workgroup p owns hidden units [p*UW, (p+1)*UW), UW = Hp/16, all four gates of them, and
//     keeps their weight fragments in LDS for the whole call (78 KiB at E = 50, H = 256): a step reads no weights from memory;
//   * per step each of the 4 waves multiplies one [32 (gate, unit) rows] x [K] weight tile into one 32-sequence tile of
//     [x_t | 1 | h_{t-1}] (LDS, 78 KiB): 156 v_mfma_f32_32x32x2_f32 per wave = 10 k cycles, 1/16 of the row tile's GEMM;
//     the weights are the MFMA's A operand with rows ordered (gate, unit), so an accumulator lane holds all four gates of
//     four units of ONE sequence: the gate formulas run lane-locally on all 64 lanes, c stays in 4 registers;
//   * h_t crosses workgroups as {value, tag} 64-bit words (agent-scope atomic store / load, tag = (call epoch, step), two
//     alternating buffers, bounded spin -> error flag instead of a hang): the exchange protocol of lstm_persist.hip, 16 KiB
//     published and 128 KiB read per workgroup and step;
//   * projection + l2-normalise: workgroup p computes the 32-column tile p of h_T . M on MFMA (the matrix kernel's tail), the
//     per-tile row sums of squares travel the same way and are added in tile order.
// Arithmetic: per output element the SAME fp32 fma chain as lstm_fwd.hip (frag32 operands, k = [x | bias row | h] in
// k-group order, v_mfma_f32_32x32x2_f32 = fma(a[k0], b[k0], c) then fma(a[k1], b[k1], .)), the same gate formulas, projection
// order and sum-of-squares tree: results are BIT-IDENTICAL to lstm_fwd.hip / lstm_small.hip / lstm_persist.hip
// (tests/test_gpu_encode.py), and the pad-prefix table of lstm_small.hip serves the exact left-PAD skip here too.
// All 16 workgroups of a cluster must be resident together (one per CU: the kernel uses 157 of 160 KiB of LDS); clusters are
// dealt to the XCDs (blockIdx % 8), two per XCD at most: 16 clusters = 1024 sequences fill the chip.
#include "sse_kernels.h"

#define LC_NWG 16
#define LC_ROWS 64
#define LC_NT 256
#define LC_SPIN_LIMIT (1 << 22)

__device__ __forceinline__ float lc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float lc_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

typedef unsigned int lc_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lc_publish(unsigned long long *p, float v, unsigned int tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lc_peek(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float lc_await(const unsigned long long *p, unsigned int tag, int32_t *err) {
  unsigned long long w = lc_peek(p);
  int spins = 0;
  while ((unsigned int)(w >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    w = lc_peek(p);
    if (++spins > LC_SPIN_LIMIT) {  // the producing workgroup never ran: report (bit 2), do not hang
      atomicOr(err, 4);
      break;
    }
  }
  return __uint_as_float((unsigned int)w);
}

#ifdef SSE_LC_CLOCK  // measurement builds (tools/): cycles per phase of a step, summed over the steps, wave 0 of workgroup 0
#define LC_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define LC_CLK(i)                   \
  {                                 \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;            \
    ck_t = n_;                      \
  }
#else
#define LC_CLK_DECL
#define LC_CLK(i)
#endif

__global__ __launch_bounds__(LC_NT) void lstm_cluster_kernel(LstmClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lcs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cluster = xcd + 8 * (jj >> 4), p = jj & 15;
  if (cluster >= a.NCL) return;  // launched only to keep a cluster's workgroups on one XCD
  const int T = a.T, H = a.H, Hp = a.Hp, S = a.S, KGx = a.KGx, KGh = a.KGh, KG = KGx + KGh, Ep = KGx * 8;
  const int UW = Hp / LC_NWG, CT = UW / 8;           // units per workgroup (8 | 16), 8-unit weight tiles per workgroup
  const int KGhe = min(KGh, (H + 7) / 8);            // h k-groups that can be non-zero
  float *Wl = lcs;                                   // [CT][KG][256] weight fragments (A operand, rows = (gate, unit))
  float *Xl = Wl + (size_t)CT * KG * 256;            // [2 row tiles][KG][256] [x_t | 1 | h_{t-1}] (B operand, rows = sequences)
  int *red = reinterpret_cast<int *>(Xl + (size_t)2 * KG * 256);  // [64]
  const int b0 = cluster * LC_ROWS, nb = min(LC_ROWS, a.B - b0);
  unsigned long long *hx = a.hx + (size_t)cluster * 2 * LC_ROWS * Hp;  // [2][64][Hp] {h, tag}
  unsigned long long *sx = a.sx + (size_t)cluster * 16 * LC_ROWS;      // [16 tiles][64] {sum of squares, tag}
  const unsigned int epoch = a.epoch << 12;                            // tag = epoch | step + 1 (T < 4095)

  // left-pad prefix skip, exactly as lstm_small.hip: the cluster starts at t0 = min leading-PAD count of its rows
  int t0 = 0;
  if (a.pad_h != nullptr) {
    int lead = T;
    if (tid < LC_ROWS && tid < nb) {
      const int32_t *row = a.ids + (size_t)(b0 + tid) * T;
      lead = 0;
      while (lead < T && row[lead] == 0) ++lead;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    if (lane == 0) red[wv] = lead;
    __syncthreads();
    t0 = min(min(min(red[0], red[1]), min(red[2], red[3])), T - 1);
    __syncthreads();
  }
  auto fetch_id = [&](int s, int t) -> int {
    int id = (s < nb) ? a.ids[(size_t)(b0 + s) * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  // x_t of the 64 rows into the x part of the operand tile: 16-byte piece q of a padded embedding row = k 4q .. 4q+3
  auto gather_x = [&](int t) {
    for (int i = tid; i < LC_ROWS * 2 * KGx; i += LC_NT) {
      const int s = i & 63, q = i >> 6;  // a wave = one piece of 64 sequences: its LDS stores are contiguous
      const f32x4 v = *reinterpret_cast<const f32x4 *>(a.emb + (size_t)fetch_id(s, t) * Ep + q * 4);
      *reinterpret_cast<f32x4 *>(Xl + ((size_t)(s >> 5) * KG + (q >> 1)) * 256 + ((q & 1) * 32 + (s & 31)) * 4) = v;
    }
  };

  // ---- this workgroup's weight fragments into LDS, once (contiguous in the packed array)
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.Wc) + (size_t)p * CT * KG * 64;
    f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
    for (int i = tid; i < CT * KG * 64; i += LC_NT) dst[i] = src[i];
  }
  // h_{t0-1}: zero, or the pad-prefix state (the same for every row)
  for (int i = tid; i < LC_ROWS * (Hp / 4); i += LC_NT) {
    const int s = i & 63, uq = i >> 6;
    f32x4 v = {0, 0, 0, 0};
    if (t0 > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (uq * 4 + e < H) v[e] = a.pad_h[(size_t)t0 * a.pad_stride + uq * 4 + e];
    }
    *reinterpret_cast<f32x4 *>(Xl + ((size_t)(s >> 5) * KG + KGx + (uq >> 1)) * 256 + ((uq & 1) * 32 + (s & 31)) * 4) = v;
  }
  gather_x(t0);
  // wave -> (row tile, weight tile); an accumulator lane holds sequence rt*32 + (lane & 31), register 4g + j = gate g of
  // unit p*UW + ct*8 + 4*(lane >> 5) + j
  const bool active = wv < 2 * CT;
  const int rt = active ? wv / CT : 0, ct = active ? wv % CT : 0;
  const int seq = rt * 32 + (lane & 31), unit0 = p * UW + ct * 8 + 4 * (lane >> 5);
  float c[4] = {0, 0, 0, 0};
  if (active && t0 > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (unit0 + j < H) c[j] = a.pad_c[(size_t)t0 * a.pad_stride + unit0 + j];
  }
  __syncthreads();

  // x_{t+1} is requested BEFORE the MFMA phase of step t and parked in registers; it goes into the operand tile once every
  // wave has finished reading x_t (after the barrier).  Its token ids were requested a step earlier (ids of step t+2 ride
  // along), so neither of the two dependent round trips (id -> embedding row) is waited for in front of the MFMAs.
  constexpr int XPT = 4;  // 16-byte pieces per thread: 64 rows x 2*KGx <= 1024 pieces (KGx <= 8)
  f32x4 xr[XPT];
  int idn[XPT];           // token id of this thread's pieces at the step after next
  auto prefetch_ids = [&](int t) {
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int i = tid + u * LC_NT;
      idn[u] = (i < LC_ROWS * 2 * KGx && t < T) ? fetch_id(i & 63, t) : 0;
    }
  };
  auto prefetch_x = [&]() {  // embedding pieces of the ids in idn
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int i = tid + u * LC_NT;
      if (i < LC_ROWS * 2 * KGx) xr[u] = *reinterpret_cast<const f32x4 *>(a.emb + (size_t)idn[u] * Ep + (i >> 6) * 4);
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int i = tid + u * LC_NT;
      if (i < LC_ROWS * 2 * KGx) {
        const int s = i & 63, q = i >> 6;
        *reinterpret_cast<f32x4 *>(Xl + ((size_t)(s >> 5) * KG + (q >> 1)) * 256 + ((q & 1) * 32 + (s & 31)) * 4) = xr[u];
      }
    }
  };
  // The h exchange is 16 KiB written and 128 KiB read per workgroup and step.  Write-through (sc1) stores are visible to every
  // XCD but drop the line from the writer's L2, so all 32 MiB a step would come back from the memory side; when the 16
  // workgroups of the cluster sit on ONE XCD (what the blockIdx % 8 dealing gives in practice -- HIP does not promise it, so
  // it is checked, through the write-through path) plain stores leave the lines in the L2 all 16 readers share, and their sc1
  // loads (L1 bypassed) are served from there.
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * LC_ROWS * Hp * 8, 0x00020000);
  bool wthrough = true;
  if (!a.write_through) {
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    const unsigned int tag0 = epoch;  // step tags start at epoch | 1
    if (tid == 0) lc_publish(sx + (size_t)p * LC_ROWS, __uint_as_float(xcc), tag0);
    if (tid < LC_NWG) {
      const unsigned int other = __float_as_uint(lc_await(sx + (size_t)tid * LC_ROWS, tag0, a.err));
      const unsigned long long same = __ballot(other == xcc);
      if (tid == 0) red[8] = ((same & 0xFFFFull) == 0xFFFFull) ? 0 : 1;
    }
    __syncthreads();
    wthrough = red[8] != 0;
  }
  prefetch_ids(t0 + 1);
  LC_CLK_DECL
  for (int t = t0; t < T; ++t) {
    if (t + 1 < T) prefetch_x();  // x_{t+1}: its ids arrived during the previous step
    prefetch_ids(t + 2);
    // ---- gate pre-activations of this wave's tile: the matrix kernel's fma chain (k-groups of x, bias row, h in order)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (active) {
      const float *wp = Wl + (size_t)ct * KG * 256 + lane * 4;
      const float *xp = Xl + (size_t)rt * KG * 256 + lane * 4;
      const int kend = KGx + KGhe;
      // two operand register sets: the LDS reads of k-group kg+1 are in flight under the four MFMAs of k-group kg
      auto ldw = [&](int kg) { return *reinterpret_cast<const f32x4 *>(wp + (size_t)kg * 256); };
      auto ldx = [&](int kg) { return *reinterpret_cast<const f32x4 *>(xp + (size_t)kg * 256); };
      f32x4 w0 = ldw(0), x0 = ldx(0), w1, x1;
      int kg = 0;
      for (; kg + 2 <= kend; kg += 2) {
        w1 = ldw(kg + 1);
        x1 = ldx(kg + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x0[e], acc, 0, 0, 0);
        const int kn = min(kg + 2, kend - 1);
        w0 = ldw(kn);
        x0 = ldx(kn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], x1[e], acc, 0, 0, 0);
      }
      if (kg < kend) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x0[e], acc, 0, 0, 0);
      }
    }
    LC_CLK(0)
    __syncthreads();  // every wave has read the operand tile of step t
    LC_CLK(1)
    if (active) {
      // BasicLSTMCell gates (forget bias folded into the packed bias row), lane-local
      const unsigned int tag = epoch | (unsigned)(t + 1);
      float hv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float si = lc_sigmoid(acc[j]), tj = lc_tanh(acc[4 + j]), sf = lc_sigmoid(acc[8 + j]), so = lc_sigmoid(acc[12 + j]);
        const float pij = __fmul_rn(si, tj);  // the matrix kernel parks this product (rounded) between its two passes
        c[j] = __builtin_fmaf(c[j], sf, pij);
        hv[j] = lc_tanh(c[j]) * so;
      }
      // 32 bytes per lane, 1 KiB contiguous per half wave: whole cache lines leave the CU
      const int off = ((((t + 1) & 1) * (Hp / 4) + (unit0 >> 2)) * LC_ROWS + seq) * 32;
      const lc_u32x4 w0 = {__float_as_uint(hv[0]), tag, __float_as_uint(hv[1]), tag}, w1 = {__float_as_uint(hv[2]), tag, __float_as_uint(hv[3]), tag};
      if (wthrough) {
        __builtin_amdgcn_raw_buffer_store_b128(w0, hrs, off, 0, 16);  // aux 16 = sc1: write-through, visible to every XCD
        __builtin_amdgcn_raw_buffer_store_b128(w1, hrs, off + 16, 0, 16);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(w0, hrs, off, 0, 0);   // the line stays in the cluster's own L2
        __builtin_amdgcn_raw_buffer_store_b128(w1, hrs, off + 16, 0, 0);
      }
    }
    LC_CLK(2)
    if (t + 1 < T) store_x();
    LC_CLK(3)
    // ---- h_t of the whole cluster into the h part of the operand tile.  ALL of a thread's pieces are requested at once
    // (one memory round trip per step instead of four) as 16-byte loads of two {value, tag} words that bypass the L1
    // (sc1: the producers are other CUs); an element whose tag is not this step's yet is re-read until it is.
    {
      const unsigned int tag = epoch | (unsigned)(t + 1);
      const int pbase = ((t + 1) & 1) * LC_ROWS * Hp * 8;
      constexpr int QPT = 16;  // 4-unit pieces per thread at Hp = 256 (8 at Hp = 128)
      const int nq = LC_ROWS * (Hp / 4), npt = nq / LC_NT;
      lc_u32x4 w[QPT][2];
      unsigned stale = (npt >= 32) ? 0xFFFFFFFFu : ((1u << npt) - 1u);  // pieces still to be (re)read
      int spins = 0;
      while (stale != 0u) {
#pragma unroll
        for (int u = 0; u < QPT; ++u)
          if ((stale >> u) & 1u) {
            const int i = tid + u * LC_NT;
            w[u][0] = __builtin_amdgcn_raw_buffer_load_b128(hrs, i * 32, pbase, 16);  // aux 16 = sc1: served by the L2
            w[u][1] = __builtin_amdgcn_raw_buffer_load_b128(hrs, i * 32 + 16, pbase, 16);
          }
        unsigned still = 0u;
#pragma unroll
        for (int u = 0; u < QPT; ++u)
          if ((stale >> u) & 1u) {
            if (w[u][0][1] != tag || w[u][0][3] != tag || w[u][1][1] != tag || w[u][1][3] != tag) {
              still |= 1u << u;
            } else {
              const int i = tid + u * LC_NT;
              const int s = i & 63, uq = i >> 6;
              *reinterpret_cast<f32x4 *>(Xl + ((size_t)(s >> 5) * KG + KGx + (uq >> 1)) * 256 + ((uq & 1) * 32 + (s & 31)) * 4) =
                  f32x4{__uint_as_float(w[u][0][0]), __uint_as_float(w[u][0][2]), __uint_as_float(w[u][1][0]), __uint_as_float(w[u][1][2])};
            }
          }
        stale = still;
        if (stale != 0u) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > LC_SPIN_LIMIT) {  // a producing workgroup never ran: report (bit 2), do not hang
            atomicOr(a.err, 4);
            break;
          }
        }
      }
    }
    LC_CLK(4)
    __syncthreads();
    LC_CLK(5)
  }
#ifdef SSE_LC_CLOCK
  if (blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 3))
    printf("[cluster clock] wave %d cycles/step: mfma %lld | barrier1 %lld | gates+publish %lld | gather x %lld | exchange read %lld | barrier2 %lld\n",
           wv, ck_[0] / (T - t0), ck_[1] / (T - t0), ck_[2] / (T - t0), ck_[3] / (T - t0), ck_[4] / (T - t0), ck_[5] / (T - t0));
#endif

  // ---- projection: column tile p of out = h_T . M, both row tiles (waves 0, 1), the matrix kernel's tail
  const int NTS = a.NTS;
  if (p >= NTS || wv >= 2) return;
  const int prt = wv, nt = p;
  f32x16 pacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) pacc[r] = 0.0f;
  {
    const float *hp = Xl + ((size_t)prt * KG + KGx) * 256 + lane * 4;
    const float *mp = a.Mp + (size_t)nt * KGh * 256 + lane * 4;
    for (int kg = 0; kg < KGhe; ++kg) {
      const f32x4 ax = *reinterpret_cast<const f32x4 *>(hp + (size_t)kg * 256), bx = *reinterpret_cast<const f32x4 *>(mp + (size_t)kg * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc, 0, 0, 0);
    }
  }
  float scale[16];
  if (a.normalize) {
    const unsigned int tag = epoch | (unsigned)(T + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = pacc[r] * pacc[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if ((lane & 31) == 0) lc_publish(sx + (size_t)nt * LC_ROWS + prt * 32 + mfma_row(r, lane), v, tag);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = prt * 32 + mfma_row(r, lane);
      float tot = 0.0f;
      for (int j = 0; j < NTS; ++j) tot += lc_await(sx + (size_t)j * LC_ROWS + row, tag, a.err);
      scale[r] = 1.0f / sqrtf(fmaxf(tot, 1e-12f));  // tf.nn.l2_normalize epsilon
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.0f;
  }
  const int col = nt * 32 + (lane & 31);
  if (col < S) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = prt * 32 + mfma_row(r, lane);
      if (row < nb) a.out[(size_t)(b0 + row) * S + col] = pacc[r] * scale[r];
    }
  }
}

// Wc[p][ct][kg][lane][e]: weight fragment rows i = lane & 31 = (gate i >> 3, unit p*UW + ct*8 + (i & 7)), k = kg*8 + (lane >> 5)*4 + e
// in the matrix kernel's k space [x padded to Ep | h]: k-row E = bias (+1 for the forget gate), padding rows / units zero
__global__ void pack_lstm_cluster_kernel(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int Ep, int Hp,
                                         int64_t total4, f32x4 *__restrict__ out) {
  const int KG = (Ep + Hp) / 8, UW = Hp / LC_NWG, CT = UW / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    int64_t blk = i >> 6;
    const int kg = (int)(blk % KG);
    blk /= KG;
    const int ct = (int)(blk % CT), p = (int)(blk / CT);
    const int row = l & 31, g = row >> 3, unit = p * UW + ct * 8 + (row & 7);
    f32x4 v = {0, 0, 0, 0};
    if (unit < H) {
      const int col = g * H + unit;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = kg * 8 + (l >> 5) * 4 + e;
        if (kk < E) v[e] = K[(size_t)kk * 4 * H + col];
        else if (kk == E) v[e] = b[col] + (g == 2 ? 1.0f : 0.0f);
        else if (kk >= Ep && kk - Ep < H) v[e] = K[(size_t)(E + kk - Ep) * 4 * H + col];
      }
    }
    out[i] = v;
  }
}

static size_t lc_lds_bytes(int Ep, int Hp) {
  const int KG = (Ep + Hp) / 8, CT = Hp / LC_NWG / 8;
  return ((size_t)(CT + 2) * KG * 256 + 64) * sizeof(float);
}

// 1: the shape runs on the cluster kernel (cell sizes up to 256, the operand tiles fit 160 KiB of LDS)
int lstm_cluster_ok(int E, int H, int S) {
  if (H < 1 || H > 256 || S < 1 || S > 512) return 0;
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  return lc_lds_bytes(Ep, Hp) <= 160 * 1024 ? 1 : 0;
}
int lstm_cluster_max_rows() { return 16 * LC_ROWS; }
size_t lstm_cluster_weight_floats(int E, int H) {
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  return (size_t)LC_NWG * (Hp / LC_NWG / 8) * ((Ep + Hp) / 8) * 256;
}
size_t lstm_cluster_hx_words(int H) { return (size_t)16 * 2 * LC_ROWS * (H <= 128 ? 128 : 256); }
size_t lstm_cluster_sx_words() { return (size_t)16 * 16 * LC_ROWS; }

hipError_t launch_pack_lstm_cluster(const float *K, const float *b, int E, int H, float *Wc, hipStream_t stream) {
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  const int64_t total4 = (int64_t)lstm_cluster_weight_floats(E, H) / 4;
  hipLaunchKernelGGL(pack_lstm_cluster_kernel, dim3((int)((total4 + 255) / 256)), dim3(256), 0, stream, K, b, E, H, Ep, Hp, total4,
                     reinterpret_cast<f32x4 *>(Wc));
  return hipGetLastError();
}

// a.epoch must differ from the epoch of every earlier launch on the same exchange buffers (20 bits; the buffers start
// zeroed and epoch 0 is never used)
hipError_t launch_lstm_cluster(const LstmClusterArgs &a_in, hipStream_t stream) {
  LstmClusterArgs a = a_in;
  if (!lstm_cluster_ok(a.E, a.H, a.S) || a.Ep != ((a.E + 8) & ~7) || a.B < 1 || a.B > lstm_cluster_max_rows() || a.T < 1 || a.T > 4094 || a.epoch == 0 ||
      a.epoch >= (1u << 20) || (a.Hp != 128 && a.Hp != 256) || a.KGx * 8 != a.Ep || a.KGh * 8 != a.Hp)
    return hipErrorInvalidValue;
  a.NCL = (a.B + LC_ROWS - 1) / LC_ROWS;
  const size_t lds = lc_lds_bytes(a.Ep, a.Hp);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_cluster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // 8 XCDs x 32 slots: cluster c = xcd + 8 * (slot / 16); with at most 8 clusters only the first 16 slots are launched
  const int grid = a.NCL <= 8 ? 8 * LC_NWG : 16 * LC_NWG;
  hipLaunchKernelGGL(lstm_cluster_kernel, dim3(grid), dim3(LC_NT), lds, stream, a);
  return hipGetLastError();
}
