// BPTT of the fp32 train step, second generation (the library default since round 4: sse_train_step computes in fp32 like
// the reference's tf.float32 graph, sse_model.py:355-364 -- tf.gradients through static_rnn over BasicLSTMCell).
//
// One workgroup walks one 32-row tile backwards through the T steps (as lstm_bwd_kernel in train.hip does); what changed:
//  * ORIENTATION.  The recurrent GEMM dh_{t-1}[b][j] = sum_n dG_t[b][n] Kh[j][n] takes Kh^T as the MFMA's A operand and the dG
//    tile as its B operand, so an accumulator lane holds ONE row b and 16 hidden units -- the layout lstm_fwd_kernel<.., TSW>
//    writes its gate tape in.  Registers 4q .. 4q+3 of a lane are then 4 consecutive units = 4 consecutive reduction indices
//    n of ONE row: exactly one 16-byte piece of the dG operand tile in LDS.  The gate backward leaves the wave as 16
//    conflict-free ds_write_b128 per step (lstm_bwd_kernel: 64 dword scatters into 8 banks).
//  * dG FOR THE WEIGHT GRADIENT straight from registers: the dK GEMM wants (one n, 4 consecutive rows) pieces; the rows of a
//    quad of lanes are consecutive, so a 4 x 4 quad transpose (DPP) turns the same registers into those pieces: 16 global
//    stores of 16 bytes per lane and step, 64-byte runs.  (lstm_bwd_kernel re-read its LDS tile for that: dword reads that
//    hit 4 banks -- "dump", a quarter of that kernel.)
//  * dX INSIDE: dX_t = dG_t Kx^T is taken from the tile while it sits in LDS -- the dG fragment a wave has just read for the
//    recurrent GEMM is also the A operand of its share of dX (wave = e-tile x one of NW/2 k-ranges, walked FIRST; partial
//    sums meet in LDS) -- and scattered into the dense embedding gradient one step later by two waves, behind their GEMM.
//    The A-operand copy of dG (dg_a: 1 GB written and read back per encoder at 8192 x 32 rows) and dx_kernel are gone.
//  * NO BIAS ACCUMULATORS: d(bias) is row E of the weight-gradient GEMM (the A-tape carries the constant-1 column that
//    feeds the bias through the forward GEMM), see dk_reduce_kernel.
#include <type_traits>

#include "sse_kernels.h"
#include "train.h"

__device__ __forceinline__ float b2_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

struct LstmBwd2Args {
  const float *tape_g;   // forward gate tape, lane = sequence accumulator layout (LstmFwdArgs::tape_swap)
  const float *dh_last;  // [Bp][Hp]
  const float *KhT;      // frag32(rows = j (Hp), red = n (4Hp)): Kh^T  [Hp/32][KGn][256]
  const float *KxT;      // frag32(rows = e (64), red = n): Kx^T  [2][KGn][256]
  float *dg_b;           // [(T*NT32*4)][NTn][256]: dG as frag32(rows = n, red = r), the dK GEMM's B operand
  float *dx;             // [T][NT32*32][64]: dX_t rows (columns >= E: unspecified when E <= 32), scattered by dx_scatter_kernel
  int32_t T, NT32, NT_tape, H, E;
  long long *clk;        // -DSSE_BWD_CLOCK builds only: per-wave phase cycle sums of tile 0 ([wave][8])
};

#ifdef SSE_BWD_CLOCK  // measurement builds (tools/): cycles per phase of the BPTT step, summed over the steps
#include <cstdio>
#define B2_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define B2_CLK(i)                   \
  {                                 \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;            \
    ck_t = n_;                      \
  }
#else
#define B2_CLK_DECL
#define B2_CLK(i)
#endif

#ifndef B2_PFV  // operand ring depth of the fused loops and the order of their first loads (measurement builds override them)
#define B2_PFV 4
#endif
#ifndef B2_INTERLEAVE
#define B2_INTERLEAVE 0
#endif
#ifndef BWD2_NT
#define BWD2_NT 2  // cache policy of the tape loads: nt (streamed once)
#endif

// LIN: every hidden unit is live (H == Hp): the live k-groups are 0 .. 4 KGg - 1 and the walk is one masked increment (the
// general (gate base, group in gate) walk costs ~7 scalar instructions per step and walk: 6 % of the kernel at H = 256).
template <int NW, bool LIN>
__global__ __launch_bounds__(NW * 64) void lstm_bwd2_kernel(LstmBwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) float dgs[];  // [KGn][256] dG tile, frag32(rows = b, red = n); then the dX partials
  constexpr int Hp = 32 * NW, KGn = Hp / 2, KGg = Hp / 8, NTn = Hp / 8;
  // dX: every wave forms the partial product of ONE e-tile over ONE of KQ k-ranges (GPQ gates each) inside its GEMM loop --
  // both products read the same dG fragment -- so that the two waves of a SIMD carry the same number of MFMAs and neither
  // runs alone (a separate dX phase on four waves left them issuing one MFMA per ~125 cycles while their partners waited at
  // the barrier: 112 k cycles per step for 82 k of MFMA).  Two waves (one per e-tile) keep their own partial in
  // registers, add the others' (parked in LDS after the barrier that ends the step) and scatter ONE STEP LATER, behind their
  // GEMM -- off the critical path (first version: in front of the next step's gate backward, 18 k cycles with everyone else
  // waiting at the barrier).  (The keeper / scatterer is k-range 1, see `scat` below.)
  constexpr int KQ = NW / 2, GPQ = 4 / KQ;
  float *xpart = dgs + (size_t)KGn * 256;  // [KQ - 1][2 e-tiles][16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, T = a.T;
  const int b = lane & 31, half = lane >> 5;

  f32x16 dh, dc;
  float tg[4][16], tcn[16], tcp[16];
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.tape_g + ((size_t)(tile % a.NT_tape) * NW + wn) * 5 * 1024), 0, 0x7fffffff, 0x00020000);
  const int tvo = lane * 16;
  const int tstep = a.NT_tape * NW * 5 * 1024 * 4;  // bytes between consecutive steps (T*tstep < 2^31 checked by the launcher)
  auto tld4 = [&](int t, int qty, int q4) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, tvo, t * tstep + (qty * 1024 + q4 * 256) * 4, BWD2_NT));
  };
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    f32x4 v[6];
#pragma unroll
    for (int g = 0; g < 5; ++g) v[g] = tld4(T - 1, g, q4);
    v[5] = tld4(T > 1 ? T - 2 : 0, 4, q4);  // unused when T == 1
    const f32x4 d4 = *reinterpret_cast<const f32x4 *>(a.dh_last + (size_t)(tile * 32 + b) * Hp + wn * 32 + q4 * 8 + half * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = q4 * 4 + e;
      dh[r] = d4[e];
      dc[r] = 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) tg[g][r] = v[g][e];
      tcn[r] = v[4][e];
      tcp[r] = v[5][e];
    }
  }

  const int KGl = min(KGg, (a.H + 7) / 8);  // live k-groups per gate (dG columns of padded units are exactly 0)
  const int et = wn & 1, kq = wn >> 1;      // dX: this wave's e-tile and k-range (gates kq * GPQ .. + GPQ - 1)
  const bool elive = et * 32 < a.E;
  // the scatter falls to the waves of k-range 1 (waves 2 / 3): measured, they leave the MFMA loops ~14 k cycles before waves
  // 0 / 1 and their SIMD partners, so the ~6 k cycles of the scatter stay off the step's critical path
  const bool scat = kq == 1 && elive;       // this wave scatters e-tile et
  // dX_t leaves the kernel as plain rows (during step t-1: own partial from registers + the other k-ranges' from LDS, fixed
  // order; lane = column e, register = row: 128 contiguous bytes per register and row half) and dx_scatter_kernel adds them
  // into the embedding gradient afterwards.  First version: buffer atomics from here -- agent-scope float atomics complete
  // slowly, and the in-order vmcnt of the wave that issued them then held back its tape and weight loads (its MFMA loop
  // ran 16 k cycles longer than its neighbours').
  auto dx_store = [&](int t, const f32x16 &own) {
    const float *xp = xpart + ((size_t)et * 16) * 64 + lane;
    float *dst = a.dx + ((size_t)t * a.NT32 + tile) * 32 * 64 + et * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = own[r];
#pragma unroll
      for (int p = 0; p < KQ - 1; ++p) v += xp[((size_t)p * 2 * 16 + r) * 64];
      dst[mfma_row(r, lane) * 64] = v;
    }
  };

  // division-free walk over the live k-groups: (gate base, group in gate), wrapping from the last gate to the first
  struct Walk {
    int l, base;
  };
  auto adv = [&](Walk &w) {
    if constexpr (LIN) {
      w.base = (w.base + 1) & (4 * KGg - 1);  // (l stays 0)
      return;
    }
    const int l1 = w.l + 1;
    const bool wrap = l1 == KGl;
    const int b1 = w.base + KGg;
    w.base = wrap ? (b1 == 4 * KGg ? 0 : b1) : w.base;
    w.l = wrap ? 0 : l1;
  };
  const int NL = 4 * KGl;    // live k-groups of the recurrent GEMM: a multiple of 4
  const int NA = GPQ * KGl;  // ... of which the first NA (this wave's own gates: the walk starts there) also feed its dX partial
  const bool fused = (NA & (B2_PFV - 1)) == 0 && (NL & (B2_PFV - 1)) == 0;  // rings of 8 k-groups; other cell sizes take the plain loops (rings of 4)
  const int g0 = kq * GPQ * KGg;

  f32x16 xacc, xprev;
  B2_CLK_DECL
  for (int t = T - 1; t >= 0; --t) {
    // ---- elementwise gate backward: 16-byte pieces into the LDS operand tile, their quad transposes to dg_b
    {
      const size_t rg = ((size_t)t * a.NT32 + tile) * 4 + (b >> 3);
      float *gb = a.dg_b + (rg * NTn + wn) * 256 + ((((b >> 2) & 1) * 32 + half * 4 + (b & 3)) << 2);
      float *ls = dgs + (size_t)(wn * 4) * 256 + lane * 4;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 g4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = q4 * 4 + e;
          const float si = tg[0][r], tj = tg[1][r], sf = tg[2][r], so = tg[3][r];
          const float cprev = (t > 0) ? tcp[r] : 0.0f;
          const float tc = b2_tanh(tcn[r]);
          const float dhv = dh[r];
          const float dov = dhv * tc;
          const float dcv = dc[r] + dhv * so * (1.0f - tc * tc);
          g4[0][e] = dcv * tj * si * (1.0f - si);
          g4[1][e] = dcv * si * (1.0f - tj * tj);
          g4[2][e] = dcv * cprev * sf * (1.0f - sf);
          g4[3][e] = dov * so * (1.0f - so);
          dc[r] = dcv * sf;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<f32x4 *>(ls + (size_t)(g * KGg + q4) * 256) = g4[g];  // n = g Hp + 32 wn + 8 q4 + 4 half + e
          f32x4 tr = g4[g];
          sse_quad_transpose(tr, lane);  // -> rows (b & ~3) .. +3 of column n = g Hp + 32 wn + 8 q4 + 4 half + (b & 3)
#ifndef B2_NOSTORE  // (measurement builds: what the dG stream costs the loops behind it)
          *reinterpret_cast<f32x4 *>(gb + (size_t)(g * (Hp / 32)) * 256 + q4 * 32) = tr;
#else
          if (a.H < 0) *reinterpret_cast<f32x4 *>(gb + (size_t)(g * (Hp / 32)) * 256 + q4 * 32) = tr;
#endif
        }
        __builtin_amdgcn_sched_barrier(0);  // bound the interleaving to one quarter (register pressure)
      }
    }
    B2_CLK(0)
    __syncthreads();
    B2_CLK(1)

    const bool scatter_now = scat && t < T - 1;  // dX of step t+1: own partial in xprev, the others' parked in LDS

    // ---- recurrent GEMM dh_{t-1}[j][b] = sum_n Kh[j][n] dG[b][n] (A = Kh^T fragments from L2, B = the LDS tile) and, over this
    // wave's own gates, dX_t[b][e] += sum_n dG[b][n] Kx[e][n] (A = the SAME tile fragment: lane = column e, register = row,
    // the layout the scatter wants; B = Kx^T fragments from L2)
    int lg = lane;
    asm volatile("" : "+v"(lg));  // opaque copy: the addresses below are recomputed per step (register pressure)
    const float *la = dgs + lg * 4;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.KhT) + (size_t)wn * KGn * 256, 0, KGn * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.KxT) + (size_t)et * KGn * 256, 0, KGn * 1024, 0x00020000);
    auto kld = [&](int kg) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, lg * 16, kg * 1024, 0)); };
    auto xld = [&](int kg) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, lg * 16, kg * 1024, 0)); };
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[r] = 0.0f;
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[r] = 0.0f;
    }
    // (the last step, t = 0, still runs the loops for its dX partial; its dh is not used)
    // PF = k-groups of weights in flight per ring (L2).  A wave whose SIMD partner is in another phase has only its own ring to
    // cover the L2 latency: 4 groups (1 k cycles of MFMAs) were not enough -- the wave that lost the arbitration while both
    // were in the loop then finished its remainder at half rate.  8 where the group counts allow it.
    auto mfma_loops = [&](auto pf_tag, bool dxl) {
      constexpr int PF = decltype(pf_tag)::value;
      f32x4 bq[PF], xb[PF], aq[2];
      Walk wb{0, g0}, wa{0, g0}, xw{0, g0};
      if (dxl) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          bq[p] = kld(wb.base + wb.l);
          adv(wb);
          __builtin_amdgcn_sched_barrier(0);
#if B2_INTERLEAVE
          xb[p] = xld(xw.base + xw.l);
          adv(xw);
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#if !B2_INTERLEAVE
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          xb[p] = xld(xw.base + xw.l);
          adv(xw);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      } else {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          bq[p] = kld(wb.base + wb.l);
          adv(wb);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      aq[0] = *reinterpret_cast<const f32x4 *>(la + (wa.base + wa.l) * 256);
      adv(wa);
      __builtin_amdgcn_s_setprio(1);
      int kg = 0;
      if (dxl) {
        for (; kg < NA; kg += PF) {
#pragma unroll
          for (int p = 0; p < PF; ++p) {
            aq[(p + 1) & 1] = *reinterpret_cast<const f32x4 *>(la + (wa.base + wa.l) * 256);
            adv(wa);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              dh = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[p][e], aq[p & 1][e], dh, 0, 0, 0);
              xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[p & 1][e], xb[p][e], xacc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            bq[p] = kld(wb.base + wb.l);
            adv(wb);
            xb[p] = xld(xw.base + xw.l);  // (past the own gates the ring walks on into valid, unused fragments)
            adv(xw);
          }
        }
      }
      for (; kg < NL; kg += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          aq[(p + 1) & 1] = *reinterpret_cast<const f32x4 *>(la + (wa.base + wa.l) * 256);  // (wraps past the end: unused)
          adv(wa);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) dh = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[p][e], aq[p & 1][e], dh, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bq[p] = kld(wb.base + wb.l);
          adv(wb);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };
    if (fused) {
      mfma_loops(std::integral_constant<int, B2_PFV>{}, elive);
    } else {
      mfma_loops(std::integral_constant<int, 4>{}, false);
      if (elive) {  // plain dX loop for the odd shapes (no operand ring: rare, small cells)
        Walk w{0, g0};
        for (int i = 0; i < NA; ++i) {
          const f32x4 av = *reinterpret_cast<const f32x4 *>(la + (w.base + w.l) * 256);
          const f32x4 bv = xld(w.base + w.l);
          adv(w);
#pragma unroll
          for (int e = 0; e < 4; ++e) xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], xacc, 0, 0, 0);
        }
      }
    }
    B2_CLK(2)
    if (scatter_now) dx_store(t + 1, xprev);
    B2_CLK(3)
    // ---- refill the tape registers for step t-1 (c_{t-1} is already here: it was this step's c_prev); unconditional (the
    // last step re-reads step 0 for nothing) so that the old values are not kept alive on a not-taken path
    {
      __builtin_amdgcn_sched_barrier(0);
      const int tp = t > 0 ? t - 1 : 0, tpp = t > 1 ? t - 2 : 0;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 v[5];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = tld4(tp, g, q4);
        v[4] = tld4(tpp, 4, q4);  // step 0 ignores it (c_{-1} = 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = q4 * 4 + e;
#pragma unroll
          for (int g = 0; g < 4; ++g) tg[g][r] = v[g][e];
          tcn[r] = tcp[r];
          tcp[r] = v[4][e];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    B2_CLK(4)
    __syncthreads();
    B2_CLK(5)
    if (elive) {  // this step's partial: parked for the scatter wave (read after the next barrier), or kept
      if (kq != 1) {
        float *xp = xpart + ((size_t)((kq > 1 ? kq - 1 : 0) * 2 + et) * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[r * 64] = xacc[r];
      } else {
        xprev = xacc;
      }
    }
    B2_CLK(6)
  }
#ifdef SSE_BWD_CLOCK
  if (a.clk && tile == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) a.clk[wn * 8 + i] = ck_[i];
#endif
  __syncthreads();
  if (scat) dx_store(0, xprev);  // dX of step 0
}

// ---------------------------------------------------------------------------
// Scatter-add of the dX rows into the dense embedding gradient (duplicate ids summed, as TF's sparse Adagrad does): one wave
// per (row, step), lane = column e.  PAD (0) and EOS (1) fill most rows of a left-padded batch: their rows are summed per
// workgroup in LDS and go to hot_part without atomics (dx_hot_reduce_kernel adds the blocks in fixed order).  sum(dx^2) is
// taken per OCCURRENCE (tf.global_norm sees the un-deduplicated IndexedSlices.values, sse_model.py:359-362) into sq_part.
#define DXS_ROWS 16  // (row, step) pairs per 1024-thread workgroup
__global__ __launch_bounds__(DXS_ROWS * 64) void dx_scatter_kernel(const float *__restrict__ dx, const int32_t *__restrict__ ids, int B,
                                                                   int Bp, int T, int E, int V, float *__restrict__ d_emb,
                                                                   float *__restrict__ sq_part, float *__restrict__ hot_part) {
  __shared__ float hot[2][DXS_ROWS][64], sqs[DXS_ROWS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long item = (long long)blockIdx.x * DXS_ROWS + w;  // = t * Bp + row
  const int t = (int)(item / Bp), row = (int)(item - (long long)t * Bp);
  float v = 0.0f;
  int id = -1;
  if (t < T && row < B) {
    id = ids[(size_t)row * T + t];
    if (id < 0 || id >= V) id = -1;  // flagged by the forward pass
    if (lane < E) v = dx[(size_t)item * 64 + lane];
  }
  if (id < 0) v = 0.0f;
  hot[0][w][lane] = (id == 0) ? v : 0.0f;
  hot[1][w][lane] = (id == 1) ? v : 0.0f;
  float sq = v * v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0) sqs[w] = sq;
  if (id >= 2 && lane < E) atomicAdd(d_emb + (size_t)id * E + lane, v);
  __syncthreads();
  if (threadIdx.x < 128) {
    const int hid = threadIdx.x >> 6;
    float h = 0.0f;
#pragma unroll
    for (int i = 0; i < DXS_ROWS; ++i) h += hot[hid][i][lane];
    hot_part[((size_t)blockIdx.x * 2 + hid) * 64 + lane] = h;
  } else if (threadIdx.x == 128) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < DXS_ROWS; ++i) s += sqs[i];
    sq_part[blockIdx.x] = s;
  }
}

int dx_scatter_blocks(int T, int Bp) { return (int)(((long long)T * Bp + DXS_ROWS - 1) / DXS_ROWS); }


size_t lstm_bwd2_lds_bytes(int Hp) {
  const int KQ = Hp / 64;  // k-ranges of the dX product = NW / 2
  return ((size_t)(Hp / 2) * 256 + (size_t)(KQ - 1) * 2 * 16 * 64) * sizeof(float);
}

hipError_t launch_lstm_bwd2(const float *tape_g, const float *dh_last, const float *KhT, const float *KxT, float *dg_b,
                            float *dx, const int32_t *ids, float *d_emb, float *sq_part, float *hot_part, int T, int NT32,
                            int NT_tape, int Hp, int H, int B, int E, int V, hipStream_t st) {
  LstmBwd2Args a{tape_g, dh_last, KhT, KxT, dg_b, dx, T, NT32, NT_tape > 0 ? NT_tape : NT32, H, E, nullptr};
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
        for (int w = 0; w < 8; w += 1)
          fprintf(stderr, "[bwd2 clock] wave %d cycles/step: elementwise %lld | barrier1 %lld | gemm+dX %lld | dX store %lld | refill-issue %lld | barrier2 %lld | park %lld\n",
                  w, v[w * 8 + 0] / T, v[w * 8 + 1] / T, v[w * 8 + 2] / T, v[w * 8 + 3] / T, v[w * 8 + 4] / T, v[w * 8 + 5] / T, v[w * 8 + 6] / T);
    }
  } report{clk_dev, st, T};
#endif
  if (E > 64 || (Hp != 128 && Hp != 256)) return hipErrorInvalidValue;
  if ((size_t)T * NT32 * (Hp / 32) * 5 * 1024 * sizeof(float) >= ((size_t)1 << 31)) return hipErrorInvalidValue;  // 32-bit tape offsets
  const size_t lds = lstm_bwd2_lds_bytes(Hp);
  auto go = [&](auto kern, int threads) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(NT32), dim3(threads), lds, st, a);
    return hipGetLastError();
  };
  const bool lin = H == Hp;
  hipError_t e = Hp == 128 ? (lin ? go(lstm_bwd2_kernel<4, true>, 256) : go(lstm_bwd2_kernel<4, false>, 256))
                           : (lin ? go(lstm_bwd2_kernel<8, true>, 512) : go(lstm_bwd2_kernel<8, false>, 512));
  if (e != hipSuccess) return e;
  const int Bp = NT32 * 32;
  hipLaunchKernelGGL(dx_scatter_kernel, dim3(dx_scatter_blocks(T, Bp)), dim3(DXS_ROWS * 64), 0, st, dx, ids, B, Bp, T, E, V, d_emb, sq_part,
                     hot_part);
  return hipGetLastError();
}
