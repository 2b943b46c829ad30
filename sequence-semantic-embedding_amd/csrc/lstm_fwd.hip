// LSTM sequence encoder forward for gfx950 (MI355X): embedding gather + T
// BasicLSTMCell steps + projection + optional row L2-normalise, one launch.
//
// Replaces the TF graph built by sse_model.py:163-164 (embedding_lookup),
// :240-242/:248-250/:262-263/:273 (static_rnn over BasicLSTMCell, last step),
// :245/:254/:267/:275 (projection) and :282-283 (l2_normalize).
//
// Design (DESIGN.md "K2"): one 512-thread workgroup owns 64 sequences for all
// T steps.  h lives in LDS in MFMA A-fragment order (double-buffered: one
// barrier per step), c in registers, the per-step gate GEMM
// [64,(Ep+Hp)] x [(Ep+Hp),4Hp] runs on v_mfma_f32_32x32x2_f32 (exact fp32,
// 157 TF chip peak).  The kernel matrix is pre-packed so every B operand is one
// coalesced 1 KiB wave load from L2; gate non-linearities are applied straight
// on the accumulators.
//
// Wave decomposition (template MT = M-tiles per wave):
//   MT = 2 (Hp = 256): wave w owns hidden units [32w, 32w+32) for ALL 64 rows --
//       every weight tile is loaded by exactly one wave per CU and feeds two
//       MFMAs (both row halves): 16 MFMAs per (2 global loads + 2 LDS reads).
//   MT = 1 (Hp = 128): wave w: units [32(w&3), +32), row half w>>2.
// Per unit block the four gates are computed in two passes (i,j then f,o) so
// that only 2*MT accumulators are live.
//
// Operand roles (inference, SWAP): the WEIGHT fragment is the MFMA's A operand (M = hidden
// units) and the x/h fragment its B operand (N = sequences), so an accumulator lane holds ONE
// sequence (lane & 31) and 16 units -- registers 4g..4g+3 are four consecutive units, i.e. exactly
// one float4 of the frag32 h tile: h_t (and the parked sigmoid(i)*tanh(j)) leave the wave as four
// contiguous, conflict-free ds_write_b128 per row tile.  (The other orientation -- lane = unit,
// registers = sequences -- scatters 16 dwords per lane into 4 banks: 8-way conflicts,
// SQ_LDS_BANK_CONFLICT = 9.4e7 per 16384-sequence launch in round 1.)  Both operands use the same
// frag32 layout, so the swap is free.  The training forward keeps the old orientation: its gate tape
// is re-read lane-privately by the BPTT kernel in that layout.
// The bias is not added separately: the padded embedding table carries a constant 1.0 in column E
// and the packed kernel the bias (forget-bias folded in) in k-row E, so it rides in the GEMM.
#include <cstdlib>

#include "sse_kernels.h"

#define LSTM_THREADS 512
#define LSTM_BM 64

#ifdef SSE_FWD_CLOCK  // measurement builds (tools/): cycles per phase of a step, summed over the steps, workgroup 0
#include <cstdio>
__device__ long long g_fwd_clk[16 * 8];
#define FW_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define FW_CLK(i)                   \
  {                                 \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;            \
    ck_t = n_;                      \
  }
#else
#define FW_CLK_DECL
#define FW_CLK(i)
#endif

// LDS: xbuf[XD ? 2 : 1][RT][KGx][256] + hbuf[2 bufs][RT][KGh][256] + red[256 floats].
// h is double-buffered (step t reads buffer t&1, writes (t+1)&1: one barrier per
// step); x is double-buffered too when it fits in the 160 KiB (XD).
// (RT = 32-row tiles per workgroup: 2, or 1 for the low-latency / Hp = 512 configurations)
bool lstm_fwd_x_double(int KGx, int KGh, int RT) { return (size_t)(2 * RT * (KGx + KGh) + 1) * 1024 <= 160 * 1024; }
size_t lstm_fwd_lds_bytes(int KGx, int KGh, int RT) {
  const int xb = lstm_fwd_x_double(KGx, KGh, RT) ? 2 : 1;
  return (size_t)(xb * RT * KGx + 2 * RT * KGh) * 256 * sizeof(float) + 64 * 4 * sizeof(float);
}

// v_exp_f32 (2^x) + v_rcp_f32 (1 ulp); plain `/` or __fdividef would expand to the ~10-instruction IEEE division
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// B operand: one 1 KiB wave load through a buffer descriptor -- the address is
// (SGPR descriptor) + (SGPR soffset) + (per-lane voffset = 16*lane, constant): no 64-bit
// per-lane pointer arithmetic, no address VGPRs.
__device__ __forceinline__ f32x4 wload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// One GEMM pass of a wave: two gate tiles (B operands at byte offsets soff and soff+1024 of
// the packed kernel, k-group stride 4096 B) times MT row tiles, over k-groups [0, kend).
// A fragments: LIN -> x and h parts are contiguous in LDS (xa[m] + kg*256); otherwise x tile
// for kg < KGx, h tile after.  Operands travel through a ring of R k-groups (weights from L2, A fragments from LDS), issued in
// consumption order so that every wait is a counted vmcnt: the operands of k-groups kg+1 .. kg+R-1 are in flight while kg's
// 8*MT MFMAs issue.  (Rounds 1-3: two named sets = one group ahead.  Enough while both waves of a SIMD are in a GEMM pass --
// the partner's MFMAs double the cover -- but a wave that runs its pass while the partner is in its gate epilogue or waits at
// the barrier had 8*MT*64 cycles to hide an L2 round trip: clock64 showed such passes at 73 % of the matrix rate at MT = 1.)
// The accumulation order is k-group by k-group as before: results are bit-identical.
// kbeg: first k-group.
template <int MT, bool LIN, bool SWAP, int R = (MT == 1 ? 4 : 3), bool PRIO = true>
__device__ __forceinline__ void gemm_pass(__amdgpu_buffer_rsrc_t wr, int voff, int soff, const float *const (&xa)[MT],
                                          const float *const (&ha)[MT], int KGx, int kend, f32x16 (&acc)[MT][2], int kbeg = 0) {
  auto a_frag = [&](int m, int kg) -> f32x4 {
    if constexpr (LIN) return *reinterpret_cast<const f32x4 *>(xa[m] + kg * 256);
    return *reinterpret_cast<const f32x4 *>(kg < KGx ? xa[m] + kg * 256 : ha[m] + (kg - KGx) * 256);
  };
  f32x4 p0[R], p1[R], af[R][MT];
  const int klast = kend - 1;
#pragma unroll
  for (int s = 0; s < R; ++s) {
    const int kg = kbeg + s < kend ? kbeg + s : klast;  // (shorter passes: harmless reloads of the last group)
    p0[s] = wload(wr, voff, soff + kg * 4096);
    p1[s] = wload(wr, voff + 1024, soff + kg * 4096);
#pragma unroll
    for (int m = 0; m < MT; ++m) af[s][m] = a_frag(m, kg);
    __builtin_amdgcn_sched_barrier(0);
  }
  auto mfmas = [&](int s) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][0] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(p0[s][e], af[s][m][e], acc[m][0], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][m][e], p0[s][e], acc[m][0], 0, 0, 0);
        acc[m][1] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(p1[s][e], af[s][m][e], acc[m][1], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][m][e], p1[s][e], acc[m][1], 0, 0, 0);
      }
  };
  if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);  // the MFMA stream outranks the partner wave's epilogue VALU
  int kg = kbeg;
  for (; kg + R <= kend; kg += R) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      mfmas(s);
      __builtin_amdgcn_sched_barrier(0);
      const int kn = (kg + s + R < kend) ? kg + s + R : klast;  // clamped: harmless reload past the end
      p0[s] = wload(wr, voff, soff + kn * 4096);
      p1[s] = wload(wr, voff + 1024, soff + kn * 4096);
#pragma unroll
      for (int m = 0; m < MT; ++m) af[s][m] = a_frag(m, kn);
    }
  }
  // the last kend % R groups: their operands are stages 0 .. already loaded
#pragma unroll
  for (int s = 0; s < R - 1; ++s)
    if (kg + s < kend) mfmas(s);
  if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
}

// Kernel configurations (RT = 32-row tiles per workgroup, MT = row tiles per wave, UBW = unit
// blocks of 32 hidden units a wave processes one after the other); always 8 waves:
//   <2,2,1>  Hp = 256, 64 rows: wave w owns unit block w for both row tiles      (throughput)
//   <2,1,1>  Hp = 128, 64 rows: wave w: unit block w&3, row tile w>>2
//   <1,1,1>  Hp = 256, 32 rows: wave w owns unit block w      (half the per-step latency: used
//            when the batch cannot fill the chip with 64-row tiles)
//   <1,1,2>  Hp = 512, 32 rows: wave w owns unit blocks w and w+8
template <int RT, int MT, int UBW, bool TRAIN, bool LIN, bool SPL = false, bool TSW = false>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_fwd_kernel(LstmFwdArgs a) {
  static_assert(!TSW || TRAIN, "TSW is the training forward in the inference orientation");
  constexpr int NTHR = LSTM_THREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ROWS = RT * 32;                 // sequences per workgroup
  constexpr int TPR = NTHR / ROWS;              // threads per sequence row in the x gather
  constexpr int NWR = 8 / RT;                   // waves sharing one row tile (projection tail)
  // weights as the MFMA A operand (see the file header).  TSW: the training forward in that orientation too -- the h_t /
  // parked-product stores are conflict-free 16-byte pieces instead of 8-way conflicting dword scatters, the gate tape is
  // written in the lane = sequence accumulator layout (lstm_bwd2_kernel reads it back the same way), and the A-tape pieces
  // (one k', 4 consecutive rows) come from registers through 4 x 4 quad transposes instead of 40 conflicting LDS reads per
  // thread and step.  Same fma chains as the other orientation: bit-identical values.
  constexpr bool SWAP = !TRAIN || TSW;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // first unit block owned by this wave (further ones at +8), and its first row tile
  // <2,1,1>: wave w -> unit block w>>1, row tile w&1.  Waves are placed on SIMD w%4, so the live unit blocks of a
  // small cell (H <= 64: blocks 0,1 -> waves 0..3) land on four different SIMDs and the padding-only blocks, which
  // are not computed at all in inference, leave no SIMD with two busy waves.
  // Three live unit blocks (64 < H <= 96 -- the reference's default cell size, every makefile recipe): 3 blocks x 2
  // row tiles = 6 wave jobs on 8 waves leave two SIMDs with two jobs and two with one.  Instead waves 0..3 take
  // blocks 0,1 whole and block 2 is split BY PASS: wave 4 / 6 compute the i,j pass of row tile 0 / 1 and hand
  // sigmoid(i)*tanh(j) over through the (already used) parking slots in LDS, wave 5 / 7 compute the f,o pass, own the
  // cell state and finish the step: 12 pass-units, 3 per SIMD (waves w and w+4 share one).
  // SPL (<1,1,1> with Hp = 128: ONE 32-row tile per workgroup, half the per-step latency of the 64-row mapping --
  // for batches that cannot fill the chip, e.g. the reference's default training shape of 128 pair rows): 4 unit
  // blocks on 8 waves, every block split by pass in the same way: wave ub computes its i,j pass, wave ub + 4 its f,o
  // pass and owns the cell state.
  const bool split3 = !TRAIN && RT == 2 && MT == 1 && a.H > 64 && a.H <= 96;
  int ub0 = (RT == 2 && MT == 1) ? (w >> 1) : w;
  int mt0 = (RT == 2 && MT == 1) ? (w & 1) : 0;
  bool do_a = true, do_b = true;
  if (split3 && w >= 4) {
    ub0 = 2;
    mt0 = (w - 4) >> 1;
    do_a = ((w & 1) == 0);
    do_b = !do_a;
  }
  if constexpr (SPL) {
    static_assert(RT == 1 && MT == 1 && UBW == 1, "pass-split mapping is defined for one 32-row tile");
    ub0 = w & 3;
    do_a = w < 4;
    do_b = !do_a;
  }
  const int fidx = SPL ? ub0 : mt0;  // hand-over flag of this wave's (unit block | row tile)
  const int KGx = a.KGx, KGh = a.KGh, KG = KGx + KGh, T = a.T;
  const int KGhe = (a.KGhe > 0 && a.KGhe < KGh) ? a.KGhe : KGh;
  // LDS.  LIN (x double-buffered, fits 160 KiB): A tiles [2 bufs][RT][KG][256], the x part of
  // a row tile followed by its h part, so a k-loop walks one linear array.  Otherwise:
  // x [RT][KGx][256] single-buffered, then h [2 bufs][RT][KGh][256].
  constexpr bool XD = LIN;
  auto xptr = [&](int buf, int mt) -> float * {
    return LIN ? smem + (size_t)((buf * RT + mt) * KG) * 256 : smem + (size_t)(mt * KGx) * 256;
  };
  auto hptr = [&](int buf, int mt) -> float * {
    return LIN ? smem + (size_t)((buf * RT + mt) * KG + KGx) * 256
               : smem + (size_t)(RT * KGx + (buf * RT + mt) * KGh) * 256;
  };
  float *red = smem + (size_t)(LIN ? 2 * RT * KG : RT * KGx + 2 * RT * KGh) * 256;  // [ROWS][NWR] (>= 8 floats)
  const int b0 = blockIdx.x * ROWS;
  int *pass_flag = reinterpret_cast<int *>(red + 16);  /* LDS atomics, not volatile: see gs_flag_min4 in lstm_fwd_gs.hip */  // split3: [row tile] = steps whose i,j pass is parked
  if (tid < 8) pass_flag[tid] = 0;

  // --- x gather assignment: TPR threads per sequence row, 8 floats (one k-group) each.  Consecutive lanes take
  // consecutive ROWS of the same k-group, so a 16-byte x_store of 8 adjacent lanes covers 128 contiguous bytes
  // (with tid / TPR the 8 lanes of a store group wrote k-groups 1 KiB apart: the same banks, 8-way conflict)
  const int xr = tid % ROWS, xq = tid / ROWS;
  const bool row_ok = (b0 + xr) < a.B;
  const int32_t *id_row = a.ids + (size_t)(row_ok ? (a.row_map ? a.row_map[b0 + xr] : b0 + xr) : 0) * T;
  auto fetch_id = [&](int t) -> int {
    int id = row_ok ? id_row[t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  auto x_store = [&](int buf, int kg, f32x4 lo, f32x4 hi) {
    float *dst = xptr(buf, xr >> 5) + (size_t)kg * 256;
    *reinterpret_cast<f32x4 *>(dst + (xr & 31) * 4) = lo;         // k%8 in 0..3 -> lane half 0
    *reinterpret_cast<f32x4 *>(dst + (32 + (xr & 31)) * 4) = hi;  // k%8 in 4..7 -> lane half 1
  };

  const int NT32 = a.NT32 > 0 ? a.NT32 : gridDim.x * RT;  // 32-row tiles in the launch (tape indexing)
  // TSW: the x part of the A-tape of step `ts` from the gathered embedding row (registers): after the quad transposes this
  // lane holds, for k' = 8 kg (+4) + (xr & 3), rows (xr & ~3) .. +3 -- one float4 of AT[(r/8)*KT + k'/32][256] each
  auto tape_x = [&](int ts, int kg, f32x4 lo, f32x4 hi) {
    if constexpr (TSW) {
      if (kg >= 8) return;  // the x part of the A-tape is 64 columns wide
      sse_quad_transpose(lo, lane);
      sse_quad_transpose(hi, lane);
      const int KT = 2 + KGh / 4, b = xr & 31;
      const size_t rg = ((size_t)ts * NT32 + blockIdx.x * RT + (xr >> 5)) * 4 + (b >> 3);
      const int k0 = kg * 8 + (xr & 3);
      float *dst = a.tape_a + (rg * KT + (k0 >> 5)) * 256 + ((((b >> 2) & 1) * 32 + (k0 & 31)) << 2);
      __builtin_nontemporal_store(lo, reinterpret_cast<f32x4 *>(dst));
      __builtin_nontemporal_store(hi, reinterpret_cast<f32x4 *>(dst + 16));  // k' + 4: four lanes further
    }
  };

  // --- left-pad prefix skip: first step this tile has to compute (0 when disabled / training)
  int t0 = 0;
  if (!TRAIN && a.pad_h != nullptr) {
    // leading-PAD count of row xr (TPR threads per row scan interleaved positions), min over the
    // tile; rows beyond B count as all-PAD
    int lead = T;
    if (row_ok) {
      for (int t = xq; t < T; t += TPR)
        if (id_row[t] != 0) {
          lead = t;
          break;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    if (lane == 0) red[w] = __int_as_float(lead);
    __syncthreads();
    lead = T;
    for (int i = 0; i < NTHR / 64; ++i) lead = min(lead, __float_as_int(red[i]));
    t0 = min(lead, T - 1);  // every real sequence ends in EOS, but stay safe: at least one step
    __syncthreads();
  }

  // --- prologue: x_{t0} -> x buffer (t0 & 1), h_{t0-1} = state after t0 PAD steps (0 when t0 = 0)
  {
    const int id = fetch_id(t0);
    const float *src = a.emb + (size_t)id * a.Ep;
    for (int kg = xq; kg < KGx; kg += TPR) {
      f32x4 lo = *reinterpret_cast<const f32x4 *>(src + kg * 8);
      f32x4 hi = *reinterpret_cast<const f32x4 *>(src + kg * 8 + 4);
      x_store(t0 & 1, kg, lo, hi);
    }
    if constexpr (TSW) {  // A-tape of step t0: x columns (zeros past the padded embedding width: whole quads take this path together)
      for (int kg = xq; kg < 8; kg += TPR) {
        f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (kg < KGx) {
          lo = *reinterpret_cast<const f32x4 *>(src + kg * 8);
          hi = *reinterpret_cast<const f32x4 *>(src + kg * 8 + 4);
        }
        tape_x(t0, kg, lo, hi);
      }
    }
    const int Hp = KGh * 8;
    for (int i = tid; i < RT * KGh * 256; i += NTHR) {  // all row tiles of buffer t0 & 1
      const int mt = i / (KGh * 256), e = i % (KGh * 256);
      const int un = (e >> 8) * 8 + ((e >> 7) & 1) * 4 + (e & 3);  // k index of element e of a frag32 row tile
      hptr(t0 & 1, mt)[e] = (t0 > 0) ? a.pad_h[(size_t)t0 * Hp + un] : 0.0f;
    }
  }

  // cell state, accumulator layout: SWAP: lane = sequence, register r = unit mfma_row(r, lane) of the block;
  // otherwise lane = unit, register = sequence
  f32x16 c[UBW][MT];
#pragma unroll
  for (int u = 0; u < UBW; ++u) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int un = (ub0 + 8 * u) * 32 + (SWAP ? mfma_row(r, lane) : (lane & 31));
      const float c0 = (t0 > 0) ? a.pad_c[(size_t)t0 * (KGh * 8) + un] : 0.0f;
#pragma unroll
      for (int m = 0; m < MT; ++m) c[u][m][r] = c0;
    }
  }

  if constexpr (TSW) {
    // h_{-1} = 0: the h columns of the first step's A-tape (every later step's are written with h_t, see pass B)
    if (do_b) {
      const int KT = 2 + KGh / 4, b = lane & 31;
#pragma unroll
      for (int u = 0; u < UBW; ++u)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const size_t rg = ((size_t)t0 * NT32 + blockIdx.x * RT + mt0 + m) * 4 + (b >> 3);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            float *dst = a.tape_a + (rg * KT + 2 + ub0 + 8 * u) * 256 + ((((b >> 2) & 1) * 32 + q4 * 8 + (lane >> 5) * 4 + (b & 3)) << 2);
            __builtin_nontemporal_store(f32x4{0, 0, 0, 0}, reinterpret_cast<f32x4 *>(dst));
          }
        }
    }
  }
  __syncthreads();

  // weights: Wp[unit block][kg][gate][256], read through a buffer descriptor
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.Wp), 0, (KGh / 4) * KG * 4096, 0x00020000);
  const int wvoff = lane * 16;

  FW_CLK_DECL
  for (int t = t0; t < T; ++t) {
    FW_CLK(0)
    // prefetch the embedding rows of step t+1 into registers (one k-group per
    // thread covers E <= 8*TPR; wider embeddings are completed at the store)
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

    const int cur = t & 1, nxt = (t + 1) & 1;  // h_{t-1} (and x_t when XD) live in buffer `cur`
    const float *xa[MT], *ha[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xa[m] = xptr(cur, mt0 + m) + lane * 4;
      ha[m] = hptr(cur, mt0 + m) + lane * 4;
    }

    if constexpr (TRAIN && !TSW) {
      // A-tape for the weight-gradient GEMM: [x_t | h_{t-1}] of this tile, stored as
      // frag32 blocks with rows = k' (x: 0..63, h: 64 + unit) and reduction index
      // r = (t*NT32 + tile32)*32 + b, i.e. AT[(r/8)*KT + k'/32][256].
      const int KT = 2 + KGh / 4;
      const int nk = 64 + KGh * 8;
      auto a_elem = [&](int mt, int kp, int b) -> float {  // element (row b, k' = kp) of row tile mt, 0 beyond the x columns
        if (kp >= 64) {
          const int un = kp - 64;
          return hptr(cur, mt)[(size_t)(un >> 3) * 256 + ((((un >> 2) & 1) * 32 + b) << 2) + (un & 3)];
        }
        if (kp < KGx * 8) return xptr(cur, mt)[(size_t)(kp >> 3) * 256 + ((((kp >> 2) & 1) * 32 + b) << 2) + (kp & 3)];
        return 0.0f;
      };
      if (a.tape_a_split) {
        // split bf16 operands for the dK GEMM on the bf16 matrix pipe: per 16-row group and k'-tile a hi and a lo frag16
        // block; lane (k' & 31, half) owns rows 16 j + 8 half .. + 7 of the tile
        unsigned short *ta = reinterpret_cast<unsigned short *>(a.tape_a);
        for (int i = tid; i < nk * (ROWS / 8); i += NTHR) {
          const int kp = i % nk, o = i / nk;  // o: rows 8*o .. 8*o+7 of the workgroup
          const int mt = o >> 2, oc = o & 3;
          float v8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] = a_elem(mt, kp, oc * 8 + e);
          sse_u32x4 hi, lo;
          sse_split8(v8, hi, lo);
          const size_t g16 = ((size_t)t * NT32 + blockIdx.x * RT + mt) * 2 + (oc >> 1);
          unsigned short *dst = ta + ((g16 * KT + (kp >> 5)) * 2) * 512 + ((oc & 1) * 32 + (kp & 31)) * 8;
          *reinterpret_cast<sse_u32x4 *>(dst) = hi;
          *reinterpret_cast<sse_u32x4 *>(dst + 512) = lo;
        }
      } else {
        for (int i = tid; i < nk * (ROWS / 4); i += NTHR) {
          const int kp = i % nk, b4 = i / nk;  // b4: rows 4*b4 .. 4*b4+3 of the tile
          const int mt = b4 >> 3, bl = (b4 & 7) * 4;
          const f32x4 v = {a_elem(mt, kp, bl), a_elem(mt, kp, bl + 1), a_elem(mt, kp, bl + 2), a_elem(mt, kp, bl + 3)};
          const size_t rg = ((size_t)t * NT32 + blockIdx.x * RT + mt) * 4 + (bl >> 3);
          float *dst = a.tape_a + (rg * KT + (kp >> 5)) * 256 + ((((bl >> 2) & 1) * 32 + (kp & 31)) << 2);
          *reinterpret_cast<f32x4 *>(dst) = v;
        }
      }
    }
    // h_{-1} = 0: skip the recurrent part of step 0; hidden units >= H are padding whose state stays exactly 0,
    // so their k-groups are skipped in every step (H = 96 in 128 slots: 19 instead of 23 k-groups)
    const int kend = (t == 0) ? KGx : KGx + KGhe;

#pragma unroll
    for (int u = 0; u < UBW; ++u) {
      const int ub = ub0 + 8 * u;
      // a unit block made only of padding (units >= H) keeps c = h = 0 and its h slots are never read (k-groups
      // >= KGhe are skipped): nothing to compute.  Training keeps it (the tapes cover all Hp units).
      if (!TRAIN && a.H > 0 && ub * 32 >= a.H) {
        // (the wave still delivers its share of the x_{t+1} gather, which normally rides between the two passes)
        if (u == 0 && XD && have_next) {
          if (xq < KGx) x_store(nxt, xq, nlo, nhi);
          for (int kg = xq + TPR; kg < KGx; kg += TPR) {
            const float *src = a.emb + (size_t)nid * a.Ep + kg * 8;
            x_store(nxt, kg, *reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
          }
        }
        continue;
      }
      const int unit = ub * 32 + (lane & 31);   // !SWAP: this lane's hidden unit
      // !SWAP: h element (row 0, k = unit) in a row tile.  SWAP: this lane's float4 (4 consecutive units of its
      // sequence) in k-group 4*ub of a row tile; registers 4g..4g+3 go to k-group 4*ub + g.
      const int hoff = SWAP ? (ub * 4) * 256 + lane * 4
                            : (unit >> 3) * 256 + ((((unit >> 2) & 1) * 32) << 2) + (unit & 3);
      const int wsoff = __builtin_amdgcn_readfirstlane(ub) * KG * 4096;
      // gate tape, accumulator layout: [t][tile32][unit block][q][reg / 4][lane][reg % 4], q = si,tj,sf,so,c: a lane's
      // registers 4 q4 .. 4 q4 + 3 of a quantity are one 16-byte piece (one store here, one load in the BPTT kernel)
      float *tp[MT];
      if constexpr (TRAIN) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          tp[m] = a.tape_g + (((size_t)t * NT32 + blockIdx.x * RT + mt0 + m) * (KGh / 4) + ub) * 5 * 1024 + lane * 4;
      }

      // pass A: gates i, j  ->  pij = sigmoid(i) * tanh(j)      (BasicLSTMCell, TF 1.x)
      // pij is parked in the (still unused) h_t slots of the other h buffer -- same (row, unit)
      // coordinates -- instead of 16*MT registers held across pass B.
      // (accumulators start at 0: the bias arrives through the constant-1 column of x, see the file header)
      f32x16 g[MT][2];
      float *hdst[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        hdst[m] = hptr(nxt, mt0 + m) + hoff;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          g[m][0][r] = 0.0f;
          g[m][1][r] = 0.0f;
        }
      }
      FW_CLK(1)
      if (do_a) gemm_pass<MT, LIN, SWAP>(wr, wvoff, wsoff, xa, ha, KGx, kend, g);
      FW_CLK(2)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!do_a) break;
        if constexpr (SWAP) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 pij, si4, tj4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              si4[e] = fast_sigmoid(g[m][0][q4 * 4 + e]);
              tj4[e] = fast_tanh(g[m][1][q4 * 4 + e]);
              pij[e] = si4[e] * tj4[e];
            }
            *reinterpret_cast<f32x4 *>(hdst[m] + q4 * 256) = pij;
            if constexpr (TSW) {
              __builtin_nontemporal_store(si4, reinterpret_cast<f32x4 *>(tp[m] + q4 * 256));
              __builtin_nontemporal_store(tj4, reinterpret_cast<f32x4 *>(tp[m] + 1024 + q4 * 256));
            }
          }
        } else {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 si4, tj4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = q4 * 4 + e;
              si4[e] = fast_sigmoid(g[m][0][r]);
              tj4[e] = fast_tanh(g[m][1][r]);
              hdst[m][mfma_row(r, lane) << 2] = si4[e] * tj4[e];
            }
            if constexpr (TRAIN) {
              __builtin_nontemporal_store(si4, reinterpret_cast<f32x4 *>(tp[m] + q4 * 256));
              __builtin_nontemporal_store(tj4, reinterpret_cast<f32x4 *>(tp[m] + 1024 + q4 * 256));
            }
          }
        }
      }
      if (do_a && !do_b) {  // split3: publish the parked products of this step (LDS operations of a wave complete in order)
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        if (lane == 0) __hip_atomic_store(pass_flag + fidx, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (u == 0 && XD && have_next) {
        // x_{t+1}: its buffer was last read in step t-1, so it can be written as soon as the
        // prefetch has landed (frees the staging registers before pass B)
        if (xq < KGx) x_store(nxt, xq, nlo, nhi);
        for (int kg = xq + TPR; kg < KGx; kg += TPR) {
          const float *src = a.emb + (size_t)nid * a.Ep + kg * 8;
          x_store(nxt, kg, *reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
        }
        tape_x(t + 1, xq, nlo, nhi);  // TSW: x columns of the next step's A-tape (zeros where xq >= KGx)
      }
      // pass B: gates f (+1 folded into the bias), o -> c' = c*sigmoid(f) + pij ; h' = tanh(c')*sigmoid(o)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          g[m][0][r] = 0.0f;
          g[m][1][r] = 0.0f;
        }
      FW_CLK(3)
      if (do_b) gemm_pass<MT, LIN, SWAP>(wr, wvoff, wsoff + 2048, xa, ha, KGx, kend, g);
      FW_CLK(4)
      if (do_b && !do_a) {  // split3: the i,j pass of this (block, row tile) comes from the partner wave
        while (__hip_atomic_load(pass_flag + fidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
      }
      FW_CLK(5)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!do_b) break;
        if constexpr (SWAP) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 pij = *reinterpret_cast<const f32x4 *>(hdst[m] + q4 * 256);
            f32x4 hv4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = q4 * 4 + e;
              const float sf = fast_sigmoid(g[m][0][r]);
              const float so = fast_sigmoid(g[m][1][r]);
              const float cn = c[u][m][r] * sf + pij[e];
              c[u][m][r] = cn;
              hv4[e] = fast_tanh(cn) * so;
              if (a.rec_h != nullptr && blockIdx.x == 0 && mt0 + m == 0 && (lane & 31) == 0) {
                // sequence 0 of the launch: state after t+1 steps (used to build the pad-prefix table)
                a.rec_h[(size_t)(t + 1) * (KGh * 8) + ub * 32 + mfma_row(r, lane)] = hv4[e];
                a.rec_c[(size_t)(t + 1) * (KGh * 8) + ub * 32 + mfma_row(r, lane)] = cn;
              }
            }
            *reinterpret_cast<f32x4 *>(hdst[m] + q4 * 256) = hv4;  // h_t, A-fragment order: one 16-byte piece per k-group
            if constexpr (TSW) {
              f32x4 sf4, so4, cn4;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = q4 * 4 + e;
                sf4[e] = fast_sigmoid(g[m][0][r]);  // (the values computed above: common subexpressions)
                so4[e] = fast_sigmoid(g[m][1][r]);
                cn4[e] = c[u][m][r];
              }
              __builtin_nontemporal_store(sf4, reinterpret_cast<f32x4 *>(tp[m] + 2048 + q4 * 256));
              __builtin_nontemporal_store(so4, reinterpret_cast<f32x4 *>(tp[m] + 3072 + q4 * 256));
              __builtin_nontemporal_store(cn4, reinterpret_cast<f32x4 *>(tp[m] + 4096 + q4 * 256));
              if (have_next) {
                // h_t is the h part of the NEXT step's A-tape: k' = 64 + 32 ub + 8 q4 + 4 (lane >> 5) + e.  After the quad
                // transpose this lane holds k' = .. + (b & 3) for rows (b & ~3) .. +3: one float4 of AT[(r/8)*KT + k'/32][256]
                f32x4 ht = hv4;
                sse_quad_transpose(ht, lane);
                const int KT = 2 + KGh / 4, b = lane & 31;
                const size_t rg = ((size_t)(t + 1) * NT32 + blockIdx.x * RT + mt0 + m) * 4 + (b >> 3);
                float *dst = a.tape_a + (rg * KT + 2 + ub) * 256 + ((((b >> 2) & 1) * 32 + q4 * 8 + (lane >> 5) * 4 + (b & 3)) << 2);
                __builtin_nontemporal_store(ht, reinterpret_cast<f32x4 *>(dst));
              }
            }
          }
        } else {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 sf4, so4, cn4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = q4 * 4 + e;
              const float sf = fast_sigmoid(g[m][0][r]);
              const float so = fast_sigmoid(g[m][1][r]);
              const float cn = c[u][m][r] * sf + hdst[m][mfma_row(r, lane) << 2];
              c[u][m][r] = cn;
              const float hv = fast_tanh(cn) * so;
              hdst[m][mfma_row(r, lane) << 2] = hv;  // h_t, A-fragment order
              sf4[e] = sf;
              so4[e] = so;
              cn4[e] = cn;
            }
            if constexpr (TRAIN) {
              __builtin_nontemporal_store(sf4, reinterpret_cast<f32x4 *>(tp[m] + 2048 + q4 * 256));
              __builtin_nontemporal_store(so4, reinterpret_cast<f32x4 *>(tp[m] + 3072 + q4 * 256));
              __builtin_nontemporal_store(cn4, reinterpret_cast<f32x4 *>(tp[m] + 4096 + q4 * 256));
            }
          }
        }
      }
    }

    FW_CLK(6)
    // end of step: with a double-buffered x tile x_{t+1} is already in place; with a single
    // buffer it must wait until every wave finished step t
    if (XD) {
      __syncthreads();  // h_t complete and visible; h_{t-1} / x_t no longer needed
      FW_CLK(7)
    } else {
      __syncthreads();
      if (have_next) {
        if (xq < KGx) x_store(0, xq, nlo, nhi);
        for (int kg = xq + TPR; kg < KGx; kg += TPR) {
          const float *src = a.emb + (size_t)nid * a.Ep + kg * 8;
          x_store(0, kg, *reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
        }
        tape_x(t + 1, xq, nlo, nhi);
      }
      __syncthreads();
    }
  }

#ifdef SSE_FWD_CLOCK
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) g_fwd_clk[w * 8 + i] = ck_[i];
#endif
  if constexpr (TRAIN) {
    // h_T, row-major [Bp][Hp], for dM = h_T^T . d(out)
    const int Hp = KGh * 8;
    for (int i = tid; i < ROWS * Hp; i += NTHR) {
      const int un = i % Hp, b = i / Hp;
      a.h_last[(size_t)(b0 + b) * Hp + un] =
          hptr(T & 1, b >> 5)[(size_t)(un >> 3) * 256 + ((((un >> 2) & 1) * 32 + (b & 31)) << 2) + (un & 3)];
    }
  }

  // --- projection  out = h_T . M   (+ optional l2_normalize): NWR waves per row tile,
  // wave -> row tile wm, N tiles nt = wn, wn + NWR, ...
  const int wn = w % NWR, wm = w / NWR;
  constexpr bool pwave = true;
  constexpr int PT = 16 / NWR;  // up to Sp = 512
  const float *hp = hptr(T & 1, wm) + lane * 4;  // h_T
  f32x16 pacc[PT];
  // per-(row, N-tile) sums of squares go to LDS (the h buffer that is NOT h_T is free now) and
  // are added in fixed tile order, so a row's result does not depend on the wave decomposition
  float *ssq = hptr((T + 1) & 1, 0);  // [ROWS][16]
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + NWR * i;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
    if (nt < a.NTS && pwave) {
      const float *mp = a.Mp + (size_t)nt * KGh * 256 + lane * 4;
      // two named operand sets (as in gemm_pass): M fragments come from L2, keep the next one in flight
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
          if ((lane & 31) == 0) ssq[(wm * 32 + mfma_row(r, lane)) * 16 + nt] = v;
        }
      }
    }
  }
  float scale[16];
  if (a.normalize) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *pr = ssq + (wm * 32 + mfma_row(r, lane)) * 16;
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
    const int nt = wn + NWR * i;
    const int col = nt * 32 + (lane & 31);
    if (nt < a.NTS && col < a.S && pwave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = b0 + wm * 32 + mfma_row(r, lane);
        if (row < a.B) a.out[(size_t)(a.row_map ? a.row_map[row] : row) * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

template <int RT, int MT, int UBW, bool TRAIN, bool LIN, bool SPL, bool TSW = false>
static hipError_t launch_one(const LstmFwdArgs &a, size_t lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_fwd_kernel<RT, MT, UBW, TRAIN, LIN, SPL, TSW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const dim3 grid(a.NT32 > 0 ? a.NT32 / RT : (a.B + RT * 32 - 1) / (RT * 32)), block(LSTM_THREADS);
  hipLaunchKernelGGL((lstm_fwd_kernel<RT, MT, UBW, TRAIN, LIN, SPL, TSW>), grid, block, lds, stream, a);
#ifdef SSE_FWD_CLOCK
  {
    static int n = 0;
    if (a.B >= 1024 && n++ % 16 == 4) {
      long long v[128];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_fwd_clk), sizeof v);
      for (int w = 0; w < 8; ++w)
        fprintf(stderr, "[fwd clock <%d,%d,%d> B=%d H=%d] wave %d cycles/step: prefetch %lld | setup %lld | passA gemm %lld | epiA+x %lld | passB gemm %lld | flag wait %lld | epiB %lld | barrier %lld\n",
                RT, MT, UBW, a.B, a.H, w, v[w * 8 + 0] / a.T, v[w * 8 + 1] / a.T, v[w * 8 + 2] / a.T, v[w * 8 + 3] / a.T, v[w * 8 + 4] / a.T, v[w * 8 + 5] / a.T,
                v[w * 8 + 6] / a.T, v[w * 8 + 7] / a.T);
    }
  }
#endif
  return hipGetLastError();
}

template <int RT, int MT, int UBW, bool SPL = false>
static hipError_t launch_cfg(const LstmFwdArgs &a_in, hipStream_t stream) {
  LstmFwdArgs a = a_in;
  a.xdouble = lstm_fwd_x_double(a.KGx, a.KGh, RT) ? 1 : 0;
  const size_t lds = lstm_fwd_lds_bytes(a.KGx, a.KGh, RT);
  const bool train = a.tape_g != nullptr;
  if (train && a.tape_swap) {
    if (a.tape_a_split) return hipErrorInvalidValue;  // the register-built A-tape is fp32
    return a.xdouble ? launch_one<RT, MT, UBW, true, true, SPL, true>(a, lds, stream)
                     : launch_one<RT, MT, UBW, true, false, SPL, true>(a, lds, stream);
  }
  if (a.xdouble)
    return train ? launch_one<RT, MT, UBW, true, true, SPL>(a, lds, stream) : launch_one<RT, MT, UBW, false, true, SPL>(a, lds, stream);
  return train ? launch_one<RT, MT, UBW, true, false, SPL>(a, lds, stream) : launch_one<RT, MT, UBW, false, false, SPL>(a, lds, stream);
}

// rows per workgroup the launcher will use for (Hp, B): 64, or 32 when 64-row tiles cannot fill
// the 256 CUs (half the per-step latency, all tiles still resident) and for Hp = 512
int lstm_fwd_rows_per_wg(int Hp, int B, int tiles_elsewhere, int cus) {
  if (cus <= 0) cus = 256;  // MI355X
  if (Hp == 512) return 32;
  if ((Hp == 256 || Hp == 128) && (B + 31) / 32 + tiles_elsewhere <= cus) return 32;
  if (tiles_elsewhere == 0 && (Hp == 256 || Hp == 128)) {
    // Rounds (round 5; the CU count is the device's since round 6).  64-row tiles: one workgroup per CU,
    // ceil(n64 / cus) rounds; a round with two workgroups costs as much as a full one -- 16,491 queries = 258 tiles ran 7.53 ms
    // where 16,384 run 3.7 on 256 CUs.  32-row tiles: two workgroups per CU, a full round of 2 * cus costs 1.006 (Hp = 256) /
    // 1.06 (Hp = 128) of a 64-row round (profiles/r02_notes.txt; cost ratios measured on MI355X), a last round of <= cus
    // workgroups (one per CU: half the per-step latency) 0.55.  Take 32 rows when the model says > 7 % less:
    // 16,491 rows 7.53 -> 5.74 ms measured (profiles/r05_notes.txt); 16,384 and 32,060 rows keep 64.
    const int n64 = (B + 63) / 64, n32 = (B + 31) / 32;
    const double full = Hp == 256 ? 1.006 : 1.06;
    const int rem = n32 % (2 * cus);
    const double t64 = (double)((n64 + cus - 1) / cus), t32 = (n32 / (2 * cus)) * full + (rem == 0 ? 0.0 : rem <= cus ? 0.55 : full);
    if (t32 < 0.93 * t64) return 32;
  }
  return 64;
}

hipError_t launch_lstm_fwd(const LstmFwdArgs &a, int Hp, hipStream_t stream) {
  int rows = lstm_fwd_rows_per_wg(Hp, a.B, a.tiles_elsewhere, a.cu_count);
  if (Hp <= 256 && (a.force_rows == 32 || a.force_rows == 64)) rows = a.force_rows;
  if (const char *ev = getenv("SSE_FWD_ROWS")) {  // measurement aid
    const int r = atoi(ev);
    if (Hp <= 256 && (r == 32 || r == 64)) rows = r;
  }
  if (rows == 64 && a.NT32 > 0 && (a.NT32 & 1)) return hipErrorInvalidValue;  // tapes are laid out per 32-row tile
  if (Hp == 128 && rows == 64 && a.gate_split && a.tape_g == nullptr && a.rec_h == nullptr && a.NTS <= 16 &&
      lstm_fwd_gs_ok(a.KGx, a.KGh, a.H > 0 ? a.H : Hp)) {
    const char *ev = getenv("SSE_FWD_GS");  // measurement aid: 0 = lstm_fwd_kernel<2,1,1>
    if (!(ev && atoi(ev) == 0)) {
      LstmFwdArgs g = a;
      if (g.H <= 0) g.H = Hp;
      return launch_lstm_fwd_gs(g, stream);
    }
  }
  if (Hp == 128) return rows == 32 ? launch_cfg<1, 1, 1, true>(a, stream) : launch_cfg<2, 1, 1>(a, stream);
  if (Hp == 256) return rows == 32 ? launch_cfg<1, 1, 1>(a, stream) : launch_cfg<2, 2, 1>(a, stream);
  if (Hp == 512) return launch_cfg<1, 1, 2>(a, stream);
  return hipErrorInvalidValue;
}
