// LSTM sequence encoder forward for MID-SIZE batches (33 .. 1024 sequences): the evaluator's batches of 600
// (sse_evaluator.py:104-109), the index builder's batches of 1000 (sse_train.py:226, sse_index.py:66,90-92).
//
// Why a third small-batch kernel.  A 32-row matrix tile (lstm_fwd.hip) puts the whole [(E+H) x 4H] gate GEMM of its rows on
// ONE compute unit: 21 MFLOP per step at the fp32 MFMA rate of one CU (0.61 TF) = 34 us per step -- 1.2 ms for T = 32
// whether the launch holds 33 rows or 8192, with 600 rows keeping 19 of the 256 CUs busy.  lstm_small.hip streams the kernel
// matrix through every workgroup of 4 rows (17 us per step, 0.55 ms).  Here the hidden units of a row tile are spread over a
// CLUSTER of 16 compute units, the MFMA version of lstm_persist.hip:
//   * a cluster serves 64 sequences as four independent GROUPS of 16; workgroup p owns hidden units [p*UW, (p+1)*UW), UW =
//     Hp/16, all four gates of them, and keeps their weight fragments in LDS for the whole call (78 KiB at E = 50, H = 256):
//     a step reads no weights from memory;
//   * a workgroup is 16 waves: waves 4G .. 4G+3 step group G, each group at its own pace (its own exchange buffers, its own
//     LDS arrival counter instead of s_barrier).  A SIMD holds one wave of each group, so while one group waits for its h_t
//     to arrive from the 15 other workgroups (~3 us) the other groups' MFMAs have the matrix pipe: the exchange latency that
//     made the single-group version spend 1/3 of every step idle is mostly covered (two groups of 32: 0.29 ms at 1024 rows,
//     four of 16: 0.27 ms);
//   * per step a wave multiplies one [16 (unit, gate) rows] x [K] weight fragment into the group's 16-sequence tile of
//     [x_t | 1 | h_{t-1}] (LDS, 19.5 KiB per group): 78 v_mfma_f32_16x16x4_f32 = 2.5 k cycles, 1/64 of the cluster's GEMM (the
//     instruction issues back to back at 32 cycles with a single accumulator chain: tools/mfma_rate_probe.hip).  The weights
//     are the MFMA's A operand with rows ordered (unit, gate), so an accumulator lane holds the four gates of one unit of ONE
//     sequence: the gate formulas run lane-locally, c stays in 1 register;
//   * h_t crosses workgroups as {value, tag} 64-bit words (tag = (call epoch, step), two alternating buffers, waits that
//     give up -> error flag instead of a hang), stored 8 bytes per lane and read 16 bytes at a time (the unit pair of one LDS
//     slot), laid out so that a wave's loads are contiguous KiBs and a loaded piece is one ds_write_b64 into the operand
//     tile; 16 KiB published and 128 KiB read per workgroup and step.  When the cluster's workgroups share an XCD (checked)
//     the stores are plain and the lines stay in that XCD's L2;
//   * two group barriers per step: after the h part of the MFMAs + gates + publish + the x_{t+1} stores (nobody reads the x
//     part of the tile then), and after h_t has been written into the tile; the x part of step t+1 is multiplied between
//     them, while h_t is on its way;
//   * projection + l2-normalise: h_T is re-laid into the matrix kernel's operand layout (LDS) and workgroup p computes the
//     32-column tile p of h_T . M with the matrix kernel's tail; the per-tile row sums of squares travel like h and are added
//     in tile order.
// Arithmetic: per output element the SAME fp32 fma chain as lstm_fwd.hip (k = [x | bias row | h] ascending;
// v_mfma_f32_16x16x4_f32 = four fmas in ascending k like v_mfma_f32_32x32x2_f32's two -- tools/mfma_chain_probe.hip), the
// same gate formulas, projection order and sum-of-squares tree: results are BIT-IDENTICAL to lstm_fwd.hip / lstm_small.hip /
// lstm_persist.hip (tests/test_gpu_encode.py), and the pad-prefix table of lstm_small.hip serves the exact left-PAD skip.
// All 16 workgroups of a cluster must be resident together (one per CU: the kernel uses 158 of 160 KiB of LDS); clusters are
// dealt to the XCDs (blockIdx % 8), two per XCD at most: 16 clusters = 1024 sequences fill the chip.
#include <cstdlib>

#include "sse_kernels.h"

#define LC_NWG 16
#define LC_ROWS 64   // sequences per cluster: four groups of 16
#define LC_NT 1024   // 16 waves: group = wave >> 2
#define LC_GT 256    // threads per group
#define LC_WAIT_TICKS 1000000  // 10 ms of the 100 MHz wall clock without the awaited word: give up (error bit 2)

typedef unsigned int lc_u32x4 __attribute__((ext_vector_type(4)));
typedef float lc_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int lc_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float lc_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

__device__ __forceinline__ void lc_publish(unsigned long long *p, float v, unsigned int tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lc_peek(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait gives up when its word has not come for 10 ms, or as soon as anybody else has given up (the error word is polled
// every 64 spins): a cluster that cannot become resident costs one time-out, not one per step and workgroup.
__device__ __forceinline__ bool lc_give_up(int &spins, long long &since, int32_t *err) {
  if ((++spins & 63) != 0) return false;
  if (since == 0) since = wall_clock64();
  if ((__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) == 0 && wall_clock64() - since < LC_WAIT_TICKS) return false;
  atomicOr(err, 4);
  return true;
}
__device__ __forceinline__ float lc_await(const unsigned long long *p, unsigned int tag, int32_t *err) {
  unsigned long long w = lc_peek(p);
  int spins = 0;
  long long since = 0;
  while ((unsigned int)(w >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    w = lc_peek(p);
    if (lc_give_up(spins, since, err)) break;
  }
  return __uint_as_float((unsigned int)w);
}

// k order.  The matrix kernel's operand layout feeds v_mfma_f32_32x32x2_f32 number e of a k-group of 8 with k offsets (e, e + 4):
// its fma chain runs 0, 4, 1, 5, 2, 6, 3, 7 inside every k-group.  v_mfma_f32_16x16x4_f32 adds its four k slots (kq = lane >> 4)
// in order 0..3, so slot kq of MFMA e (two per k-group) carries k offset 2e + (kq >> 1) + 4*(kq & 1): the same chain.
__host__ __device__ __forceinline__ int lc_koff(int e, int kq) { return 2 * e + (kq >> 1) + 4 * (kq & 1); }
__host__ __device__ __forceinline__ int lc_kq_of(int o) { return 2 * (o & 1) + (o >> 2); }  // slot of k offset o; its e = (o & 3) >> 1

// The four waves of a group meet on an LDS counter (s_barrier would tie the two groups together).  A wave's LDS reads and
// writes are executed by the LDS in program order ahead of its ds_add, so a wave that sees the count knows they are done.
__device__ __forceinline__ void lc_group_barrier(int *cnt, int &target, int lane) {
  asm volatile("" ::: "memory");
  target += 4;
  if (lane == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

#ifdef SSE_LC_CLOCK  // measurement builds (tools/): cycles per phase of a step, summed over the steps
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
  const int tid = threadIdx.x, lane = tid & 63, tg = tid & (LC_GT - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), G = wv >> 2, g = wv & 3;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cluster = xcd + 8 * (jj >> 4), p = jj & 15;
  if (cluster >= a.NCL) return;  // launched only to keep a cluster's workgroups on one XCD
  if (a.drop_wg && cluster == 0 && p == LC_NWG - 1) return;  // testing aid: a workgroup that never arrives
  const int T = a.T, H = a.H, Hp = a.Hp, S = a.S, KGx = a.KGx, KGh = a.KGh, KG = KGx + KGh, Ep = KGx * 8;
  const int UW = Hp / LC_NWG, NQ = UW / 4;           // units per workgroup (8 | 16), 4-unit weight fragments per workgroup
  const int KGhe = min(KGh, (H + 7) / 8);            // h k-groups that can be non-zero
  float *Wl = lcs;                                   // [NQ][KG][64 lanes][2]: A operand, rows (unit, gate), k = kg*8 + lc_koff(e, lane >> 4)
  float *Xl = Wl + (size_t)NQ * KG * 128;            // [group][KG][64 lanes][2] (+ pad): B operand [x_t | 1 | h_{t-1}] of 16 sequences
  const int XS = KG * 128 + 32;  // floats per group operand tile (+ 128 bytes: consecutive tiles start on different LDS banks)
  int *red = reinterpret_cast<int *>(Xl + (size_t)4 * XS);  // [32]: 0..3 lead of a group; 8 publish mode; 10..13 arrival counters; 14..17 gave up
  float *H32 = (2 * KGh * 256 <= NQ * KG * 128) ? Wl : reinterpret_cast<float *>(red + 32 + 3 * 128);  // h_T in the matrix kernel's layout
  const int b0 = cluster * LC_ROWS, nb = min(LC_ROWS, a.B - b0);
  const int gb0 = b0 + G * 16, gnb = max(0, min(16, nb - G * 16));     // this group's sequences
  unsigned long long *hx = a.hx + (size_t)cluster * 2 * LC_ROWS * Hp;  // [2 steps][group][Hp/8][4][16] x 2 {h, tag}
  unsigned long long *sx = a.sx + (size_t)cluster * 16 * LC_ROWS;      // [16 tiles][64] {sum of squares, tag}
  const unsigned int epoch = a.epoch << 12;                            // tag = epoch | step + 1 (T < 4095)
  auto xslot = [&](int s16, int kg, int kq) -> float * {  // the 2 floats (k = kg*8 + lc_koff(0, kq), + 2) of sequence s16 of this group
    return Xl + (size_t)G * XS + ((size_t)kg * 64 + kq * 16 + s16) * 2;
  };

  if (tid < 32) red[tid] = (tid < 4) ? T : 0;
  __syncthreads();
  // left-pad prefix skip, exactly as lstm_small.hip: a group starts at t0 = min leading-PAD count of its rows
  int t0 = 0;
  if (a.pad_h != nullptr) {
    if (g == 0) {  // the group's 16 rows sit in the first lanes of its first wave
      int lead = T;
      if (lane < gnb) {
        const int32_t *row = a.ids + (size_t)(gb0 + lane) * T;
        lead = 0;
        while (lead < T && row[lead] == 0) ++lead;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
      if (lane == 0) red[G] = lead;
    }
    __syncthreads();
    t0 = min(red[G], T - 1);
  }
  auto fetch_id = [&](int s16, int t) -> int {
    int id = (s16 < gnb) ? a.ids[(size_t)(gb0 + s16) * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };

  // ---- this workgroup's weight fragments into LDS, once (contiguous in the packed array)
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.Wc) + (size_t)p * NQ * KG * 32;
    f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
    for (int i = tid; i < NQ * KG * 32; i += LC_NT) dst[i] = src[i];
  }
  // h_{t0-1}: zero, or the pad-prefix state (the same for every row).  Piece i of a group = units (u, u + 2), u = (i >> 6)*8 +
  // lc_koff(0, (i >> 4) & 3), of sequence i & 15: the unit pair one LDS slot holds (two waves produce its halves).
  const int npc = (Hp / 8) * 4 * 16;  // pieces per group and step
  for (int i = tg; i < npc; i += LC_GT) {
    const int s16 = i & 15, kq = (i >> 4) & 3, kgh = i >> 6, u = kgh * 8 + lc_koff(0, kq);
    lc_f32x2 v = {0, 0};
    if (t0 > 0) {
      if (u < H) v[0] = a.pad_h[(size_t)t0 * a.pad_stride + u];
      if (u + 2 < H) v[1] = a.pad_h[(size_t)t0 * a.pad_stride + u + 2];
    }
    *reinterpret_cast<lc_f32x2 *>(xslot(s16, KGx + kgh, kq)) = v;
  }
  // x_t: 16-byte piece q of a padded embedding row = k 4q .. 4q+3 = k offsets 4*(q & 1) + j of k-group q >> 1
  constexpr int XPT = 1;  // pieces per thread: 16 rows x 2*KGx <= 256 pieces (KGx <= 8)
  auto put_x = [&](int i, const f32x4 &v) {
    const int s16 = i & 15, q = i >> 4;
    float *dst = xslot(s16, q >> 1, q & 1);  // offset 4*(q & 1) + j: slot kq = 2*(j & 1) + (q & 1), e = j >> 1
    dst[0] = v[0];
    dst[64] = v[1];
    dst[1] = v[2];
    dst[65] = v[3];
  };
  for (int i = tg; i < 16 * 2 * KGx; i += LC_GT)
    put_x(i, *reinterpret_cast<const f32x4 *>(a.emb + (size_t)fetch_id(i & 15, t0) * Ep + (i >> 4) * 4));

  // wave g of a group: weight fragment g = units p*UW + 8*(g >> 1) + lc_koff(g & 1, .); an accumulator lane holds the four
  // gates (register = gate) of unit uW = that with . = lane >> 4, for sequence lane & 15 of the group
  const bool active = g < NQ && gnb > 0;
  const int qW = active ? g : 0, s16w = lane & 15, uW = p * UW + 8 * (qW >> 1) + lc_koff(qW & 1, lane >> 4);
  float cW = 0.0f;
  if (active && t0 > 0 && uW < H) cW = a.pad_c[(size_t)t0 * a.pad_stride + uW];

  // The h exchange is 16 KiB written and 128 KiB read per workgroup and step.  Write-through (sc1) stores are visible to every
  // XCD but drop the line from the writer's L2, so all 32 MiB a step would come back from the memory side; when the 16
  // workgroups of the cluster sit on ONE XCD (what the blockIdx % 8 dealing gives in practice -- HIP does not promise it, so
  // it is checked, through the write-through path) plain stores leave the lines in the L2 all 16 readers share, and their sc1
  // loads (L1 bypassed) are served from there.
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * LC_ROWS * Hp * 8, 0x00020000);
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
  } else if (tid == 0) {
    red[8] = 1;
  }
  __syncthreads();  // weights, h, x, the publish mode and the zeroed arrival counters are in LDS
  const bool wthrough = red[8] != 0;
  int *arrive = red + 10 + G;  // (10 .. 13)
  int arrived = 0;

  // x_{t+1} is requested BEFORE the MFMA phase of step t and parked in registers; it goes into the operand tile once every
  // wave of the group has finished reading x_t.  Its token ids were requested a step earlier (ids of step t+2 ride along),
  // so neither of the two dependent round trips (id -> embedding row) is waited for in front of the MFMAs.
  f32x4 xr[XPT];
  int idn[XPT];
  auto prefetch_ids = [&](int t) {  // raw: the range check waits for the value, so it is made where the id is used
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int i = tg + u * LC_GT;
      idn[u] = (i < 16 * 2 * KGx && t < T && (i & 15) < gnb) ? a.ids[(size_t)(gb0 + (i & 15)) * T + t] : 0;
    }
  };
  auto prefetch_x = [&](const int (&idc)[XPT]) {
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int i = tg + u * LC_GT;
      if (i < 16 * 2 * KGx) xr[u] = *reinterpret_cast<const f32x4 *>(a.emb + (size_t)idc[u] * Ep + (i >> 4) * 4);
    }
  };
  // Gate pre-activations of this wave: the matrix kernel's fma chain (k-groups of x, bias row, h in order), one accumulator.
  // The k-groups [k0, k1) of a step go through a ring of four operand register sets: the LDS reads of k-group kg+3 go out
  // under the MFMAs of k-group kg, 3 x 64 cycles ahead of their use.  A 16x16x4 fp32 MFMA holds the pipe for 32 cycles and the
  // wave issues in order, so everything else in the loop must fit in those shadows (or run while another group's wave has
  // the pipe): one read after each MFMA, immediate offsets from two pointers that advance once per four k-groups, no
  // clamping (the reads run up to 3 k-groups past k1: inside the tile, or into the padding behind the last one).
  f32x4 acc = {0, 0, 0, 0};
  auto mfma_range = [&](int k0, int k1) {
    const float *wa = Wl + (((size_t)qW * KG + k0) * 64 + lane) * 2;
    const float *xb = Xl + (size_t)G * XS + ((size_t)k0 * 64 + lane) * 2;
    auto ld = [&](const float *q, int kg) { return *reinterpret_cast<const lc_f32x2 *>(q + (size_t)kg * 128); };
#define LC_STEP(C, N, off)                                                  \
  N##a = ld(wa, off);                                                       \
  __builtin_amdgcn_sched_barrier(0);                                        \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C##a[0], C##x[0], acc, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);                                        \
  N##x = ld(xb, off);                                                       \
  __builtin_amdgcn_sched_barrier(0);                                        \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C##a[1], C##x[1], acc, 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
#define LC_LAST(C)                                                          \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C##a[0], C##x[0], acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C##a[1], C##x[1], acc, 0, 0, 0);
#define LC_PRE(N, off)                 \
  N##a = ld(wa, off);                  \
  __builtin_amdgcn_sched_barrier(0);   \
  N##x = ld(xb, off);                  \
  __builtin_amdgcn_sched_barrier(0);
    lc_f32x2 r0a, r0x, r1a, r1x, r2a, r2x, r3a, r3x;
    LC_PRE(r0, 0)  // in the loop's order: its waits count outstanding reads
    LC_PRE(r1, 1)
    LC_PRE(r2, 2)
    int kg = k0;
    for (; kg + 4 <= k1; kg += 4) {
      LC_STEP(r0, r3, 3)
      LC_STEP(r1, r0, 4)
      LC_STEP(r2, r1, 5)
      LC_STEP(r3, r2, 6)
      wa += 512;
      xb += 512;
    }
    if (kg < k1) { LC_LAST(r0) }
    if (kg + 1 < k1) { LC_LAST(r1) }
    if (kg + 2 < k1) { LC_LAST(r2) }
#undef LC_STEP
#undef LC_LAST
#undef LC_PRE
  };
  if (gnb > 0) {
    prefetch_ids(t0 + 1);
    if (active) mfma_range(0, KGx);  // x part of the first step
    lc_group_barrier(arrive, arrived, lane);  // (the first x_{t+1} store below must not overtake another wave's x part)
    LC_CLK_DECL
    for (int t = t0; t < T; ++t) {
      // x_{t+1}: its ids arrived during the previous step.  Order matters: take the ids out of their registers first, THEN
      // issue the loads of this step -- a wait placed after them (the id registers are reused) would wait for them
      int idc[XPT];
      bool bad = false;
#pragma unroll
      for (int u = 0; u < XPT; ++u) {
        idc[u] = idn[u];
        if (idc[u] < 0 || idc[u] >= a.V) {
          bad = true;
          idc[u] = 0;
        }
      }
      if (bad) atomicOr(a.err, 1);
      prefetch_ids(t + 2);
      if (t + 1 < T) prefetch_x(idc);
      // ---- gate pre-activations, h part (the x part of this step went in while the previous h was on its way)
      if (active) mfma_range(KGx, KGx + KGhe);
      LC_CLK(0)
      const unsigned int tag = epoch | (unsigned)(t + 1);
      const int pbase = ((t + 1) & 1) * LC_ROWS * Hp * 8 + G * npc * 16;  // this step's, this group's pieces
      if (active) {
        // BasicLSTMCell gates (forget bias folded into the packed bias row), lane-local; i*j is rounded before it meets
        // c*f as in the matrix kernel (which parks the product between its two passes)
        const float pW = __fmul_rn(lc_sigmoid(acc[0]), lc_tanh(acc[1]));
        cW = __builtin_fmaf(cW, lc_sigmoid(acc[2]), pW);
        const float hW = lc_tanh(cW) * lc_sigmoid(acc[3]);
        // half (unit offset + 2 or not) of the 16-byte piece (k-group, slot lane >> 4, sequence): 8 bytes per lane
        const int off = pbase + ((((uW >> 3) * 4 + (lane >> 4)) * 16) + s16w) * 16 + (qW & 1) * 8;
        const lc_u32x2 w = {__float_as_uint(hW), tag};
        if (wthrough)
          __builtin_amdgcn_raw_buffer_store_b64(w, hrs, off, 0, 16);  // aux 16 = sc1: write-through, visible to every XCD
        else
          __builtin_amdgcn_raw_buffer_store_b64(w, hrs, off, 0, 0);   // the line stays in the cluster's own L2
      }
      LC_CLK(1)
      // x_{t+1} into the x part of the tile: nobody reads that part now (the x part of step t was multiplied before the
      // previous step's last barrier, the h part is all that the MFMAs above read)
      if (t + 1 < T) {
#pragma unroll
        for (int u = 0; u < XPT; ++u)
          if (tg + u * LC_GT < 16 * 2 * KGx) put_x(tg + u * LC_GT, xr[u]);
      }
      lc_group_barrier(arrive, arrived, lane);  // every wave of the group has read h_{t-1} and stored its pieces of x_{t+1}
      LC_CLK(2)
      if (t + 1 < T) {
        // the x part of step t+1 while h_t is on its way: out of the next step's critical path, and the first read attempt
        // below comes late enough to find most of h_t
        acc = f32x4{0, 0, 0, 0};
        if (active) mfma_range(0, KGx);
      }
      LC_CLK(3)
      // ---- h_t of the group from the whole cluster into the h part of the operand tile.  ALL of a thread's pieces are
      // requested at once (one memory round trip per attempt) with loads that bypass the L1 (sc1: the producers are other
      // CUs); pieces whose tags are not this step's yet are requested again.
      {
        constexpr int QPT = 8;  // pieces per thread at Hp = 256 (4 at Hp = 128)
        const int npt = npc / LC_GT;
        lc_u32x4 w[QPT];
        unsigned stale = (1u << npt) - 1u;
        int spins = 0;
        long long since = 0;
        while (stale != 0u) {
#pragma unroll
          for (int u = 0; u < QPT; ++u)
            if ((stale >> u) & 1u) w[u] = __builtin_amdgcn_raw_buffer_load_b128(hrs, (tg + u * LC_GT) * 16, pbase, 16);
          unsigned still = 0u;
#pragma unroll
          for (int u = 0; u < QPT; ++u)
            if ((stale >> u) & 1u) {
              if (w[u][1] != tag || w[u][3] != tag) {
                still |= 1u << u;
              } else {
                const int i = tg + u * LC_GT;
                *reinterpret_cast<lc_f32x2 *>(xslot(i & 15, KGx + (i >> 6), (i >> 4) & 3)) = lc_f32x2{__uint_as_float(w[u][0]), __uint_as_float(w[u][2])};
              }
            }
          stale = still;
          if (stale != 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (lc_give_up(spins, since, a.err)) {  // a producing workgroup never ran: report (bit 2), do not hang
              red[14 + G] = 1;                      // and leave the step loop, the whole group together
              break;
            }
          }
        }
      }
      LC_CLK(4)
      lc_group_barrier(arrive, arrived, lane);
      LC_CLK(5)
      if (__hip_atomic_load(red + 14 + G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;  // written before the barrier
    }
#ifdef SSE_LC_CLOCK
    if (blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 15))
      printf("[cluster clock] wave %d cycles/step: mfma %lld | gates+publish %lld | arrive1 %lld | store x %lld | exchange read %lld | arrive2 %lld\n",
             wv, ck_[0] / (T - t0), ck_[1] / (T - t0), ck_[2] / (T - t0), ck_[3] / (T - t0), ck_[4] / (T - t0), ck_[5] / (T - t0));
#endif
  }
  __syncthreads();  // both groups are through their last step

  // ---- h_T of the 64 rows into the matrix kernel's A-operand layout [row tile][kg][64 lanes][4] (over the weights)
  for (int i = tid; i < LC_ROWS * Hp; i += LC_NT) {
    const int s = i & 63, u = i >> 6;
    const float v = Xl[(size_t)(s >> 4) * XS + (((size_t)KGx + (u >> 3)) * 64 + lc_kq_of(u & 7) * 16 + (s & 15)) * 2 + ((u & 3) >> 1)];
    H32[((size_t)(s >> 5) * KGh + (u >> 3)) * 256 + (((u >> 2) & 1) * 32 + (s & 31)) * 4 + (u & 3)] = v;
  }
  __syncthreads();

  // ---- projection: column tile p of out = h_T . M, both row tiles (waves 0, 1), the matrix kernel's tail
  const int NTS = a.NTS;
  if (p >= NTS || wv >= 2) return;
  const int prt = wv, nt = p;
  f32x16 pacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) pacc[r] = 0.0f;
  {
    const float *hp = H32 + (size_t)prt * KGh * 256 + lane * 4;
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

// Wc[p][q][kg][lane][e]: weight fragment rows i = lane & 15 = (unit p*UW + 8*(q >> 1) + lc_koff(q & 1, i >> 2), gate i & 3),
// k = kg*8 + lc_koff(e, lane >> 4)
// in the matrix kernel's k space [x padded to Ep | h]: k-row E = bias (+1 for the forget gate), padding rows / units zero
__global__ void pack_lstm_cluster_kernel(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int Ep, int Hp,
                                         int64_t total2, lc_f32x2 *__restrict__ out) {
  const int KG = (Ep + Hp) / 8, UW = Hp / LC_NWG, NQ = UW / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    int64_t blk = i >> 6;
    const int kg = (int)(blk % KG);
    blk /= KG;
    const int q = (int)(blk % NQ), p = (int)(blk / NQ);
    const int row = l & 15, g = row & 3, unit = p * UW + 8 * (q >> 1) + lc_koff(q & 1, row >> 2);
    lc_f32x2 v = {0, 0};
    if (unit < H) {
      const int col = g * H + unit;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kk = kg * 8 + lc_koff(e, l >> 4);
        if (kk < E) v[e] = K[(size_t)kk * 4 * H + col];
        else if (kk == E) v[e] = b[col] + (g == 2 ? 1.0f : 0.0f);
        else if (kk >= Ep && kk - Ep < H) v[e] = K[(size_t)(E + kk - Ep) * 4 * H + col];
      }
    }
    out[i] = v;
  }
}

static size_t lc_lds_bytes(int Ep, int Hp) {
  const int KG = (Ep + Hp) / 8, NQ = Hp / LC_NWG / 4, KGh = Hp / 8;
  size_t fl = (size_t)(NQ + 4) * KG * 128 + 4 * 32 + 32 + 3 * 128;  // + the operand ring's reads past the last k-group
  if (2 * KGh * 256 > NQ * KG * 128) fl += (size_t)2 * KGh * 256;  // h_T re-laid for the projection does not fit over the weights
  return fl * sizeof(float);
}

// 1: the shape runs on the cluster kernel (cell sizes up to 256, the operand tiles fit 160 KiB of LDS)
int lstm_cluster_ok(int E, int H, int S) {
  if (H < 1 || H > 256 || S < 1 || S > 512) return 0;
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  if (Ep > 64) return 0;  // two x pieces per thread
  return lc_lds_bytes(Ep, Hp) <= 160 * 1024 ? 1 : 0;
}
int lstm_cluster_max_rows() { return 16 * LC_ROWS; }
size_t lstm_cluster_weight_floats(int E, int H) {
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  return (size_t)LC_NWG * (Hp / LC_NWG / 4) * ((Ep + Hp) / 8) * 128;
}
size_t lstm_cluster_hx_words(int H) { return (size_t)16 * 2 * LC_ROWS * (H <= 128 ? 128 : 256); }
size_t lstm_cluster_sx_words() { return (size_t)16 * 16 * LC_ROWS; }

hipError_t launch_pack_lstm_cluster(const float *K, const float *b, int E, int H, float *Wc, hipStream_t stream) {
  const int Ep = (E + 8) & ~7, Hp = H <= 128 ? 128 : 256;
  const int64_t total2 = (int64_t)lstm_cluster_weight_floats(E, H) / 2;
  hipLaunchKernelGGL(pack_lstm_cluster_kernel, dim3((int)((total2 + 255) / 256)), dim3(256), 0, stream, K, b, E, H, Ep, Hp, total2,
                     reinterpret_cast<lc_f32x2 *>(Wc));
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
  // COOPERATIVE launch: the workgroups of a cluster hand h_t to each other every step, so all of them must be resident at
  // once; hipLaunchCooperativeKernel makes the runtime guarantee that (the grid fits the device by construction: at most
  // one workgroup per CU is asked for) instead of leaving it to the dispatcher's mood on a busy device.  The bounded spin
  // with its give-up flag stays as a belt.  (A runtime without cooperative launches falls back to the plain launch.)
  // Measured cost: +20 us per launch (single query 0.122 -> 0.142 ms); option lstm_cluster_coop = 0 takes the plain launch.
  static const bool no_coop = getenv("SSE_NO_COOP") != nullptr;  // measurement aid: plain launches
  if (!no_coop && !a.plain_launch) {
    void *args[] = {(void *)&a};
    hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(lstm_cluster_kernel), dim3(grid), dim3(LC_NT), args, (unsigned)lds, stream);
    if (ce == hipSuccess) return hipGetLastError();
    (void)hipGetLastError();  // not supported / too large for this device: plain launch, and say so (counter lstm_coop_refused)
    lstm_note_coop_refused();
  }
  hipLaunchKernelGGL(lstm_cluster_kernel, dim3(grid), dim3(LC_NT), lds, stream, a);
  return hipGetLastError();
}
