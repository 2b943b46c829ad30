// LSTM sequence encoder forward for the SMALL cells (H <= 128: the reference's default 96 of every makefile recipe, 64, 128;
// sse_train.py:68-69), inference, gfx950 -- "gate-split, two phase-shifted row groups".
//
// Same job and same arithmetic as lstm_fwd_kernel<2,1,1> (lstm_fwd.hip: embedding gather + T BasicLSTMCell steps +
// projection + optional l2-normalise for 64 sequences per 512-thread workgroup; sse_model.py:163-164, :240-275, :282-283),
// same packed operands, bit-identical results -- a different decomposition of the step.
//
// Why (profiles/r04_notes.txt clock64 tables): with one workgroup-wide barrier per step every wave of the
// CU runs its gate epilogue, the barrier and the operand-ring refill of the next step at the same time -- ~8-11 k cycles of a
// 40 k-cycle step at H = 96 during which the matrix pipe idles (0.64 of peak; 0.51 at H = 64 where four of the eight waves own
// nothing but padding).  A 16384-row batch is ONE 64-row tile per CU, so a second workgroup cannot hide that tail.  Here the two
// 32-row tiles of a workgroup are two independent GROUPS of four waves that never meet at a barrier inside the time loop; group 1
// starts half a step late, and from then on one group's serial tail runs under the other group's GEMM: the pipe sees back-to-back
// MFMAs from whichever group is in its GEMM phase.
//
// Decomposition: group g = w >> 2 owns row tile g; wave q = w & 3 of a group sits on SIMD q and computes GATE q (i, j, f, o)
// of ALL live unit blocks (NB = ceil(H / 32) accumulators of 32 units x 32 sequences): twelve gate tiles on four waves at
// H = 96, eight at H = 64 -- balanced for every NB, where the unit-block mapping leaves waves idle.  One x/h fragment read from
// LDS feeds NB weight tiles.  The four gates of a unit meet through LDS: every wave applies ITS non-linearity, keeps register
// quad q (units 8 q .. 8 q + 7 of every block) and hands the other three quads to their owners; the owner combines
// c' = c sigmoid(f) + sigmoid(i) tanh(j), h' = tanh(c') sigmoid(o) for its quad (c lives in its registers) and writes its piece
// of h_t in A-fragment order.  Group-local synchronisation is two sets of per-wave step flags in LDS (activations published /
// h_t published); h needs a single buffer per group (h_t is written after every wave of the group has finished reading h_{t-1}),
// x keeps two.
#include <cstdlib>

#include "sse_kernels.h"

#define GS_THREADS 512
// Wave priority: 1 in the GEMM phase; in the tail 3 where the tail is the longer leg of a group's step (NB = 2: 0.528 -> 0.490 ms
// at H = 64, NB = 4: 2.10 -> 2.08 at H = 128), 0 at NB = 3 (H = 96: 1.393 vs 1.419 ms) -- profiles/r06_notes.txt.  Priorities
// move little: the fp32 MFMA executes at the vector rate on the SIMD's own lanes (157.3 TF = the v_fma_f32 peak), so another
// wave's activation arithmetic is NOT hidden under a running MFMA stream (activations of a group 2.4 k cycles alone, 7.0 k beside
// the other group's GEMM) -- what the two groups hide from each other is latency: flag waits, LDS round trips, operand refills.
#ifndef GS_PRIO_GEMM
#define GS_PRIO_GEMM 1
#endif
#ifndef GS_PRIO_TAIL
#define GS_PRIO_TAIL (NB == 3 ? 0 : 3)
#endif

#ifdef SSE_GS_CLOCK  // measurement builds: cycles per phase of a step, summed over the steps, workgroup 0
#include <cstdio>
__device__ long long g_gs_clk[8 * 8];
#define GS_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define GS_CLK(i)                   \
  {                                 \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;            \
    ck_t = n_;                      \
  }
#else
#define GS_CLK_DECL
#define GS_CLK(i)
#endif

namespace {

__device__ __forceinline__ float gs_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float gs_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

__device__ __forceinline__ f32x4 gs_wload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Step flags in LDS.  Relaxed workgroup-scope ATOMICS on plain int pointers, not volatile accesses: the compiler keeps a volatile
// access through a pointer computed from the dynamic LDS base in the GENERIC address space -- every flag poll was four
// flat_load_dword sc0 sc1, each followed by s_waitcnt vmcnt(0) (which also drained the embedding prefetch in flight); the atomics
// become ds_read_b32 / ds_write_b32 with one lgkmcnt wait (profiles/r06_notes.txt).
__device__ __forceinline__ int gs_flag_min4(const int *f) {
  const int f0 = __hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int f1 = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int f2 = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int f3 = __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return min(min(f0, f1), min(f2, f3));
}
__device__ __forceinline__ void gs_flag_set(int *f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// slot of quad j of the activations of gate qw in the exchange area of a (group, block): the writer keeps its own quad
__device__ __forceinline__ int gs_slot(int j, int qw) { return j - (j > qw ? 1 : 0); }

// Q = this wave's gate and owned register quad (compile-time: the accumulator registers are indexed by it)
template <int NB, int Q>
__device__ __forceinline__ void gs_tail(const f32x16 (&acc)[NB], f32x4 (&c)[NB], float *ex /* exchange area of the group */,
                                        float *hb /* h tile of the group */, int *fl_act, int *fl_h, int lane,
                                        int step1 /* t + 1 */
#ifdef SSE_GS_CLOCK
                                        , long long (&ck_)[8], long long &ck_t
#endif
) {
  // 1. this wave's non-linearity on all of its gate; the three foreign quads go to their owners
  f32x4 own[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (Q == 1) ? gs_tanh(acc[b][j * 4 + e]) : gs_sigmoid(acc[b][j * 4 + e]);
      if (j == Q) own[b] = v;
      else *reinterpret_cast<f32x4 *>(ex + ((Q * NB + b) * 3 + gs_slot(j, Q)) * 256 + lane * 4) = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS operations of a wave complete in order, the flag follows the data
  if (lane == 0) gs_flag_set(fl_act + Q, step1);
  GS_CLK(4)
  // 2. wait for the other three gates of this group
  while (gs_flag_min4(fl_act) < step1) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
  GS_CLK(5)
  // 3. combine for the owned quad of every block (BasicLSTMCell, TF 1.x; the forget bias rides in the packed bias row)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    f32x4 gv[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (qq == Q) gv[qq] = own[b];
      else gv[qq] = *reinterpret_cast<const f32x4 *>(ex + ((qq * NB + b) * 3 + gs_slot(Q, qq)) * 256 + lane * 4);
    }
    f32x4 hv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pij = gv[0][e] * gv[1][e];
      asm volatile("" : "+v"(pij));  // a rounded product, as the other kernels park it in LDS: c' = fma(c, sf, pij)
      const float cn = c[b][e] * gv[2][e] + pij;
      c[b][e] = cn;
      hv[e] = gs_tanh(cn) * gv[3][e];
    }
    *reinterpret_cast<f32x4 *>(hb + (4 * b + Q) * 256 + lane * 4) = hv;  // h_t, A-fragment order: k-group 4 b + Q
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) gs_flag_set(fl_h + Q, step1);
  GS_CLK(6)
}

}  // namespace

size_t lstm_fwd_gs_lds_bytes(int KGx, int NB) {
  return (size_t)(4 * KGx + 2 * 4 * NB + 2 * 4 * NB * 3) * 256 * sizeof(float) + 64 * sizeof(int);
}
bool lstm_fwd_gs_ok(int KGx, int KGh, int H) {
  if (KGh != 16 || H < 1 || H > 128 || KGx > 8) return false;
  return lstm_fwd_gs_lds_bytes(KGx, (H + 31) / 32) <= 160 * 1024;
}

template <int NB>
__global__ __launch_bounds__(GS_THREADS) void lstm_fwd_gs_kernel(LstmFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and known to be: scalar branches / buffer offsets
  const int g = w >> 2, q = w & 3, tg = tid & 255;
  const int KGx = a.KGx, KGh = a.KGh, KG = KGx + KGh, T = a.T;
  const int KGhe = (a.KGhe > 0 && a.KGhe < KGh) ? a.KGhe : KGh;
  constexpr int KGl = 4 * NB;  // k-groups of h kept in LDS (live unit blocks)
  // LDS: x [2 bufs][2 groups][KGx][256] | h [2 groups][KGl][256] | exchange [2 groups][4 gates][NB][3 quads][256] | flags
  float *xbase = smem;
  float *hb = smem + (size_t)4 * KGx * 256 + (size_t)g * KGl * 256;
  float *ex = smem + (size_t)4 * KGx * 256 + (size_t)2 * KGl * 256 + (size_t)g * 12 * NB * 256;
  int *flags = reinterpret_cast<int *>(smem + (size_t)4 * KGx * 256 + (size_t)2 * KGl * 256 + (size_t)24 * NB * 256);
  int *fl_act = flags + g * 8, *fl_h = flags + g * 8 + 4;  // [group][activations | h][wave]
  int *red = flags + 16;                                   // [8 waves] prologue reductions
  auto xptr = [&](int buf) -> float * { return xbase + (size_t)((buf * 2 + g) * KGx) * 256; };

  // --- x gather assignment inside the group: 8 threads per sequence row, one k-group each (KGx <= 8)
  const int b0 = blockIdx.x * 64 + g * 32;
  const int xr = tg & 31, xq = tg >> 5;
  const bool row_ok = (b0 + xr) < a.B;
  const int32_t *id_row = a.ids + (size_t)(row_ok ? (a.row_map ? a.row_map[b0 + xr] : b0 + xr) : 0) * T;
  auto check_id = [&](int id) -> int {
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  auto fetch_id = [&](int t) -> int { return check_id(row_ok ? id_row[t] : 0); };
  auto x_store = [&](int buf, f32x4 lo, f32x4 hi) {
    float *dst = xptr(buf) + (size_t)xq * 256;
    *reinterpret_cast<f32x4 *>(dst + xr * 4) = lo;         // k%8 in 0..3 -> lane half 0
    *reinterpret_cast<f32x4 *>(dst + (32 + xr) * 4) = hi;  // k%8 in 4..7 -> lane half 1
  };

  if (tid < 16) flags[tid] = 0;

  // --- left-pad prefix skip, per GROUP: first step its 32 rows have to compute
  int t0 = 0;
  if (a.pad_h != nullptr) {
    int lead = T;
    if (row_ok) {
      for (int t = xq; t < T; t += 8)
        if (id_row[t] != 0) {
          lead = t;
          break;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    if (lane == 0) red[w] = lead;
    __syncthreads();
    lead = min(min(red[g * 4], red[g * 4 + 1]), min(red[g * 4 + 2], red[g * 4 + 3]));
    t0 = __builtin_amdgcn_readfirstlane(min(lead, T - 1));  // uniform: the step loop and the k-group offsets stay scalar
  }

  // --- prologue: x_{t0} -> x buffer (t0 & 1); h_{t0-1} = state after t0 PAD steps (0 when t0 = 0)
  {
    const int id = fetch_id(t0);
    if (xq < KGx) {
      const float *src = a.emb + (size_t)id * a.Ep + xq * 8;
      x_store(t0 & 1, *reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
    }
    const int Hp = KGh * 8;
    for (int i = tg; i < KGl * 256; i += 256) {
      const int un = (i >> 8) * 8 + ((i >> 7) & 1) * 4 + (i & 3);  // k index of element i of a frag32 row tile
      hb[i] = (t0 > 0) ? a.pad_h[(size_t)t0 * Hp + un] : 0.0f;
    }
  }
  // cell state of the owned quad: lane = sequence, element e = unit 32 b + 8 q + 4 (lane >> 5) + e
  f32x4 c[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int un = b * 32 + q * 8 + (lane >> 5) * 4 + e;
      c[b][e] = (t0 > 0) ? a.pad_c[(size_t)t0 * (KGh * 8) + un] : 0.0f;
    }
  __syncthreads();  // the only workgroup barrier before the projection tail

  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Wp), 0, (KGh / 4) * KG * 4096, 0x00020000);
  const int wvoff = lane * 16 + q * 1024;  // gate q of Wp[unit block][kg][gate][256]

  // group 1 starts when group 0 has published its first activations: from then on the two groups alternate between GEMM and
  // tail instead of running both in step (which would idle the matrix pipe during both tails)
  if (g == 1) {
    int spins = 0;
    while (gs_flag_min4(flags) < 1 && ++spins <= 4096) __builtin_amdgcn_s_sleep(8);  // (group 0, activations)
  }

  constexpr int R = 3;
  // token ids travel TWO steps ahead: the id of step t + 1 is in a register when step t starts (its embedding row is requested at
  // once, no load-to-load wait at the top of a step: that wait was 0.6 - 2.7 k cycles of every step), the id of t + 2 is requested
  // here and looked at when the step ends
  int id_next = (t0 + 1 < T) ? fetch_id(t0 + 1) : 0;
  GS_CLK_DECL
  for (int t = t0; t < T; ++t) {
    GS_CLK(7)
    const bool have_next = (t + 1) < T;
    const int id_raw2 = (t + 2 < T && row_ok) ? id_row[t + 2] : 0;
    f32x4 nlo = {0, 0, 0, 0}, nhi = {0, 0, 0, 0};
    if (have_next) {
      const int nid = id_next;
      if (xq < KGx) {
        const float *src = a.emb + (size_t)nid * a.Ep + xq * 8;
        nlo = *reinterpret_cast<const f32x4 *>(src);
        nhi = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    }
    const int cur = t & 1;
    const float *xa = xptr(cur) + lane * 4, *ha = hb + lane * 4;
    const int kend = (t == 0) ? KGx : KGx + KGhe;  // h_{-1} = 0: no recurrent part in step 0; padding units stay 0
    const int klast = kend - 1;

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;

    GS_CLK(0)
    // h_{t-1} complete? (every wave of the group has published its quads of step t-1)
    if (t > t0) {
      while (gs_flag_min4(fl_h) < t) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }

    GS_CLK(1)
    auto a_frag = [&](int kg) -> f32x4 {
      return *reinterpret_cast<const f32x4 *>(kg < KGx ? xa + kg * 256 : ha + (kg - KGx) * 256);
    };
    f32x4 wt[R][NB], af[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const int kg = s < kend ? s : klast;
#pragma unroll
      for (int b = 0; b < NB; ++b) wt[s][b] = gs_wload(wr, wvoff, (b * KG + kg) * 4096);
      af[s] = a_frag(kg);
      __builtin_amdgcn_sched_barrier(0);
    }
    auto mfmas = [&](int s) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[s][b][e], af[s][e], acc[b], 0, 0, 0);
    };
    __builtin_amdgcn_s_setprio(GS_PRIO_GEMM);
    int kg = 0;
    for (; kg + R <= kend; kg += R) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        mfmas(s);
        __builtin_amdgcn_sched_barrier(0);
        const int kn = (kg + s + R < kend) ? kg + s + R : klast;
#pragma unroll
        for (int b = 0; b < NB; ++b) wt[s][b] = gs_wload(wr, wvoff, (b * KG + kn) * 4096);
        af[s] = a_frag(kn);
      }
    }
#pragma unroll
    for (int s = 0; s < R - 1; ++s)
      if (kg + s < kend) mfmas(s);
    __builtin_amdgcn_s_setprio(GS_PRIO_TAIL);
    GS_CLK(2)

    // x_{t+1}: its buffer was last read in step t-1 (every wave of the group is past that GEMM: it published h_{t-1})
    if (have_next && xq < KGx) x_store(cur ^ 1, nlo, nhi);
    GS_CLK(3)

#ifdef SSE_GS_CLOCK
#define GS_TAIL_EXTRA , ck_, ck_t
#else
#define GS_TAIL_EXTRA
#endif
    switch (q) {
      case 0: gs_tail<NB, 0>(acc, c, ex, hb, fl_act, fl_h, lane, t + 1 GS_TAIL_EXTRA); break;
      case 1: gs_tail<NB, 1>(acc, c, ex, hb, fl_act, fl_h, lane, t + 1 GS_TAIL_EXTRA); break;
      case 2: gs_tail<NB, 2>(acc, c, ex, hb, fl_act, fl_h, lane, t + 1 GS_TAIL_EXTRA); break;
      default: gs_tail<NB, 3>(acc, c, ex, hb, fl_act, fl_h, lane, t + 1 GS_TAIL_EXTRA); break;
    }
    id_next = check_id(id_raw2);
  }
#ifdef SSE_GS_CLOCK
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) g_gs_clk[w * 8 + i] = ck_[i];
#endif
  __syncthreads();  // both groups: h_T complete

  // --- projection  out = h_T . M  (+ optional l2_normalize): the four waves of a group share its row tile,
  // wave -> N tiles nt = q, q + 4, ...   (same k order and the same fixed-order row sums as lstm_fwd_kernel)
  constexpr int PT = 4;  // up to Sp = 512
  const float *hp = hb + lane * 4;
  f32x16 pacc[PT];
  float *ssq = ex;  // [32 rows][16]  (the exchange area of the group is free now)
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = q + 4 * i;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
    if (nt < a.NTS) {
      const float *mp = a.Mp + (size_t)nt * KGh * 256 + lane * 4;
      f32x4 ax = *reinterpret_cast<const f32x4 *>(hp), bx = *reinterpret_cast<const f32x4 *>(mp), ay, by;
      int kg = 0;
      for (; kg + 1 < KGhe; kg += 2) {
        ay = *reinterpret_cast<const f32x4 *>(hp + (kg + 1) * 256);
        by = *reinterpret_cast<const f32x4 *>(mp + (kg + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < KGhe) ? kg + 2 : kg;
        ax = *reinterpret_cast<const f32x4 *>(hp + k2 * 256);
        bx = *reinterpret_cast<const f32x4 *>(mp + k2 * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], by[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < KGhe) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
      }
      if (a.normalize) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = pacc[i][r] * pacc[i][r];
          v += __shfl_xor(v, 1);
          v += __shfl_xor(v, 2);
          v += __shfl_xor(v, 4);
          v += __shfl_xor(v, 8);
          v += __shfl_xor(v, 16);
          if ((lane & 31) == 0) ssq[mfma_row(r, lane) * 16 + nt] = v;
        }
      }
    }
  }
  float scale[16];
  if (a.normalize) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *pr = ssq + mfma_row(r, lane) * 16;
      float tot = 0.0f;
      for (int j = 0; j < a.NTS; ++j) tot += pr[j];
      scale[r] = 1.0f / sqrtf(fmaxf(tot, 1e-12f));  // tf.nn.l2_normalize epsilon
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.0f;
  }
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = q + 4 * i;
    const int col = nt * 32 + (lane & 31);
    if (nt < a.NTS && col < a.S) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = b0 + mfma_row(r, lane);
        if (row < a.B) a.out[(size_t)(a.row_map ? a.row_map[row] : row) * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

template <int NB>
static hipError_t gs_launch(const LstmFwdArgs &a, hipStream_t stream) {
  const size_t lds = lstm_fwd_gs_lds_bytes(a.KGx, NB);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_fwd_gs_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((lstm_fwd_gs_kernel<NB>), dim3((a.B + 63) / 64), dim3(GS_THREADS), lds, stream, a);
#ifdef SSE_GS_CLOCK
  {
    static int n = 0;
    if (a.B >= 1024 && n++ % 16 == 4) {
      long long v[64];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_gs_clk), sizeof v);
      for (int w = 0; w < 8; ++w)
        fprintf(stderr, "[gs clock NB=%d B=%d H=%d T=%d] wave %d (group %d gate %d) cycles/step: id+emb issue %lld | wait h %lld | gemm %lld | x store %lld | act+publish %lld | wait act %lld | combine+h %lld | loop %lld | total %lld\n",
                NB, a.B, a.H, a.T, w, w >> 2, w & 3, v[w * 8 + 0] / a.T, v[w * 8 + 1] / a.T, v[w * 8 + 2] / a.T, v[w * 8 + 3] / a.T, v[w * 8 + 4] / a.T,
                v[w * 8 + 5] / a.T, v[w * 8 + 6] / a.T, v[w * 8 + 7] / a.T,
                (v[w * 8 + 0] + v[w * 8 + 1] + v[w * 8 + 2] + v[w * 8 + 3] + v[w * 8 + 4] + v[w * 8 + 5] + v[w * 8 + 6] + v[w * 8 + 7]) / a.T);
    }
  }
#endif
  return hipGetLastError();
}

// inference only, 64-row tiles, Hp = 128 (H <= 128), Sp <= 512; the caller checked lstm_fwd_gs_ok()
hipError_t launch_lstm_fwd_gs(const LstmFwdArgs &a, hipStream_t stream) {
  if (a.tape_g != nullptr || a.rec_h != nullptr || !lstm_fwd_gs_ok(a.KGx, a.KGh, a.H) || a.NTS > 16) return hipErrorInvalidValue;
  switch ((a.H + 31) / 32) {
    case 1: return gs_launch<1>(a, stream);
    case 2: return gs_launch<2>(a, stream);
    case 3: return gs_launch<3>(a, stream);
    default: return gs_launch<4>(a, stream);
  }
}
