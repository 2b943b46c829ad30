// Query x index cosine scoring with fused top-k for gfx950 (MI355X).
//
// Replaces np.dot(sourceEncodings, targetEncodings.T) + getSortedResults
// (sse_evaluator.py:110-111, data_utils.py:263-267, sse_demo.py:126-127): the
// reference materialises the [Q,N] float64 score matrix and fully argsorts each
// row although only the first <= 10 columns are consumed
// (sse_evaluator.py:95,112; sse_demo.py:128-129).  Here:
//   1. score_topk_kernel: [N,S] x [S,Q] on v_mfma_f32_32x32x2_f32 (exact fp32),
//      index rows as the MFMA M dimension so that every lane owns ONE query
//      column and keeps a private sorted top-KC list in registers; the [Q,N]
//      matrix is never written.
//   2. rescore_kernel: the union of the per-lane lists is re-scored exactly in
//      float64 (the reference's arithmetic), ordered (score desc, row id asc),
//      and CERTIFIED: if a row outside the candidate set could still reach the
//      exact top-k (fp32 bound), the query is flagged and
//   3. exact_topk_kernel recomputes flagged queries by float64 brute force.
#include "sse_kernels.h"

#define SC_THREADS 512
#define SC_KC 16
#define NEG_INF (-__builtin_inff())

// ---------------------------------------------------------------------------
template <int KC>
__device__ __forceinline__ void list_insert(float (&ls)[KC], int (&li)[KC], float s, int id, bool take) {
  // insert (s,id) into a descending list: once the insertion point is found every later entry
  // shifts down by one (sticky flag -- re-comparing the displaced entry would reorder equal
  // scores); lanes with take == false keep their list
  bool ins = false;
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const bool gt = ins || (take && (s > ls[i]));
    ins = gt;
    const float tv = ls[i];
    const int ti = li[i];
    ls[i] = gt ? s : tv;
    li[i] = gt ? id : ti;
    s = gt ? tv : s;
    id = gt ? ti : id;
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool entry_before(float sa, int ia, float sb, int ib) {
  return (sa > sb) || (sa == sb && (unsigned)ia < (unsigned)ib);  // score desc, then lower row (empty = -1 last)
}

// top-KC of two descending lists (A in registers, B given) -> A.  Bitonic: max(A[i], B[KC-1-i])
// is a bitonic sequence holding the KC best; log2(KC) compare-exchange stages sort it.
template <int KC>
__device__ __forceinline__ void merge_lists(float (&ls)[KC], int (&li)[KC], const float (&bs)[KC], const int (&bi)[KC]) {
#pragma unroll
  for (int i = 0; i < KC; ++i) {
    const bool tb = entry_before(bs[KC - 1 - i], bi[KC - 1 - i], ls[i], li[i]);
    ls[i] = tb ? bs[KC - 1 - i] : ls[i];
    li[i] = tb ? bi[KC - 1 - i] : li[i];
  }
#pragma unroll
  for (int stride = KC / 2; stride > 0; stride >>= 1)
#pragma unroll
    for (int i = 0; i < KC; ++i)
      if ((i & stride) == 0) {
        const bool sw = entry_before(ls[i + stride], li[i + stride], ls[i], li[i]);
        const float ts = ls[i];
        const int ti = li[i];
        ls[i] = sw ? ls[i + stride] : ts;
        li[i] = sw ? li[i + stride] : ti;
        ls[i + stride] = sw ? ts : ls[i + stride];
        li[i + stride] = sw ? ti : li[i + stride];
      }
}

// 8 waves per workgroup = 2 per SIMD: there is no barrier in the sweep, so the waves drift
// apart and one wave's top-k epilogue (VALU) overlaps its partner's MFMA stream.
// Wave tile: 1 index tile (32 rows, M) x NQ query tiles (N): NQ accumulators + NQ private lists.
//   NQ = 4: 128-query block (MFMA-bound regime).  NQ = 1: <= 32 queries (demo / web, Q = 1):
//   a quarter of the MFMA work per index byte, so the sweep runs at HBM speed.
//   MERGE: the 16 lists per query of a workgroup are merged to one (in-wave via shuffles,
//   across waves through LDS) so that many index splits stay cheap to re-score.
// BF: the candidate pass runs on v_mfma_f32_32x32x16_bf16 -- index and queries are bf16 copies in the same 1-KiB
// block / 16-byte-per-lane fragment scheme (a block now holds 32 rows x 16 k, a.KG counts 16-k groups), ONE MFMA per
// (k-group, query tile).  Only candidate SELECTION sees bf16: the float64 re-scoring pass works on the fp32 / f64
// rows with an error bound widened to the bf16 rounding, so the results stay exact (or fall back, certified).
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
template <int KC, int NQ, bool MERGE, bool BF>
__global__ __launch_bounds__(SC_THREADS) void score_topk_kernel(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // query block [NQ][KG][256]; later merge scratch
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KG = a.KG;

  // XCD-aware decode: workgroups of one XCD (blockIdx % 8) sweep the same index
  // range so that the range is fetched into that XCD's L2 once.
  int split, qb;
  {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    if (a.NSPLIT <= 8) {
      const int per = 8 / a.NSPLIT;
      split = xcd / per;
      qb = j * per + xcd % per;
    } else {
      const int m = a.NSPLIT >> 3;
      split = xcd + 8 * (j % m);
      qb = j / m;
    }
  }
  const int QB = (a.QT + NQ - 1) / NQ;
  if (qb >= QB) return;
  if (a.skip_cert) {  // fp32 second chance after a bf16 candidate pass: only blocks with an uncertified query run
    int open_q = 0;
    for (int i = tid; i < NQ * 32; i += SC_THREADS) {
      const int qq = qb * NQ * 32 + i;
      if (qq < a.Q && a.skip_cert[qq] == 0) open_q = 1;
    }
    if (!__syncthreads_or(open_q)) return;
  }

  // stage the query block (already frag32-packed) into LDS
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.qp) + (size_t)qb * NQ * KG * 64;
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    const int valid = min(NQ, a.QT - qb * NQ) * KG * 64;
    for (int i = tid; i < NQ * KG * 64; i += SC_THREADS) dst[i] = (i < valid) ? src[i] : f32x4{0, 0, 0, 0};
  }
  __syncthreads();

  float ls[NQ][KC];
  int li[NQ][KC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      ls[q][i] = NEG_INF;
      li[q][i] = -1;
    }

  // Shared per-query insertion thresholds (LDS, order-preserving int encoding of the float):
  // max over the workgroup's 16 lists of a query of their current minima.  A list minimum is
  // the KC-th best of a subset of the rows, hence a lower bound of the workgroup's KC-th best:
  // rows below it can never reach the merged top-KC, so every lane may use it as its
  // threshold -- the 16 lists of a query share their progress and insertions (a
  // wave-divergent ~100-instruction path) become ~16x rarer.
  int *thr_s = reinterpret_cast<int *>(smem + (size_t)a.thr_off);
  auto enc = [](float f) -> int { const int i = __float_as_int(f); return i >= 0 ? i : (i ^ 0x7FFFFFFF); };
  auto dec = [](int i) -> float { return __int_as_float(i >= 0 ? i : (i ^ 0x7FFFFFFF)); };
  for (int i = tid; i < NQ * 32; i += SC_THREADS) thr_s[i] = enc(NEG_INF);
  // A second, much tighter lower bound of the workgroup's KC-th best per query: the SMALLEST of its 16 lists' best
  // entries (16 distinct rows score at least that).  Every list keeps its current best in mx_s[query][list]
  // (monotone, plain stores); wave 0 folds the minimum into thr_s every few tiles.  Racy reads only see older,
  // smaller -- still valid -- values.  With 16 lists of 1/16 of the rows each, the list minima alone let a lane insert
  // on practically every tile (~150 instructions per round): at bf16 matrix speed that cost more than the MFMAs.
  float *mx_s = reinterpret_cast<float *>(thr_s + NQ * 32);  // [NQ*32][16]
  for (int i = tid; i < NQ * 32 * 16; i += SC_THREADS) mx_s[i] = NEG_INF;
  __syncthreads();

  const int tps = (a.NT + a.NSPLIT - 1) / a.NSPLIT;  // n-tiles per split
  const int t0 = split * tps, t1 = min(a.NT, t0 + tps);
  const float *qs = smem + lane * 4;
  const int voff = lane * 16;

  // NQ == 1 (<= 32 queries) is an HBM-bound sweep: 4 MFMAs per KiB of index.  Two loads in flight per wave
  // (the pipelined loop below) cover only ~4 MB chip-wide where 8 TB/s x ~1.5 us wants >= 12 MB, so this variant
  // keeps a ring of RING k-group fragments per wave in registers (32 VGPRs), running across tile boundaries.
  constexpr int RING = 8;
  constexpr int WSTEP = SC_THREADS / 64;
  const bool use_ring = (NQ == 1) && (KG % RING == 0) && ((int64_t)tps * KG * 1024 < ((int64_t)1 << 31));
  f32x4 ring[RING];
  const int tile0 = t0 + w;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.idxp) + (size_t)__builtin_amdgcn_readfirstlane(tile0 < t1 ? tile0 : t0) * KG * 256, 0,
      (tile0 < t1 ? (t1 - tile0) : 0) * KG * 1024, 0x00020000);  // this wave's whole tile stream (< 2 GiB, else use_ring is off)
  if (use_ring && tile0 < t1) {
#pragma unroll
    for (int d = 0; d < RING; ++d)
      ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, d * 1024, 0));
  }

  for (int tile = t0 + w; tile < t1; tile += SC_THREADS / 64) {
    // index tile through a buffer descriptor (base = this tile: stays below the 4 GiB
    // descriptor range for any index size); per-lane offset is the constant 16*lane
    // The descriptor also spans this wave's NEXT tile (8 tiles on) so that the loop can touch index bytes PD
    // k-groups ahead of their use: the fragment load proper is issued only one k-group (16 MFMAs, ~1-2 k cycles)
    // early, which does not cover an HBM miss; the touch pulls the lines into L2 first.
    const int utile = __builtin_amdgcn_readfirstlane(tile);
    const int span = min(a.NT - utile, SC_THREADS / 64 + 1);
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.idxp) + (size_t)utile * KG * 256, 0, span * KG * 1024, 0x00020000);
    auto iload = [&](int kg) -> f32x4 {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ir, voff, kg * 1024, 0));
    };
    constexpr int PD = 8;
    auto touch = [&](int kg) -> int {  // one dword per 16-byte segment of k-group kg+PD (next tile past the end)
      const int kp = kg + PD;
      const int off = (kp < KG) ? kp : kp + (SC_THREADS / 64 - 1) * KG;
      return __builtin_amdgcn_raw_buffer_load_b32(ir, voff, off * 1024, 0);
    };
    int tch0 = 0, tch1 = 0;
    f32x16 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

    if (use_ring) {
      if constexpr (NQ == 1) {
        const int tbase = (tile - tile0) * KG;                       // k-group index of this tile in the wave's stream
        const bool more = tile + WSTEP < t1;                          // a next tile exists
        f32x4 bq = *reinterpret_cast<const f32x4 *>(qs), bqn;
        __builtin_amdgcn_s_setprio(1);
        for (int kg0 = 0; kg0 < KG; kg0 += RING) {
#pragma unroll
          for (int d = 0; d < RING; ++d) {
            const int kg = kg0 + d;
            bqn = *reinterpret_cast<const f32x4 *>(qs + ((kg + 1 < KG) ? kg + 1 : 0) * 256);
            if constexpr (BF) {
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ring[d]), __builtin_bit_cast(bf16x8_t, bq), acc[0], 0, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d][e], bq[e], acc[0], 0, 0, 0);
            }
            // refill the slot with the fragment RING k-groups on: same tile, or the head of this wave's next tile
            const int kn = kg + RING;
            const int off = (kn < KG) ? tbase + kn : (more ? tbase + WSTEP * KG + (kn - KG) : tbase + kg);
            ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, off * 1024, 0));
            bq = bqn;
          }
        }
      }
    } else {
      // k-loop, hand software-pipelined with two named operand sets (X / Y): the index
      // fragment of k-group kg+1 (global) and the query fragments (LDS) are in flight while
      // kg's 4*NQ MFMAs issue; no register copies.
      f32x4 ax = iload(0), ay;
      f32x4 bx[NQ], by[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) bx[q] = *reinterpret_cast<const f32x4 *>(qs + (q * KG) * 256);
      __builtin_amdgcn_s_setprio(1);
      int kg = 0;
      for (; kg + 1 < KG; kg += 2) {
        asm volatile("" ::"v"(tch0), "v"(tch1));  // last iteration's touches (long since returned: in-order)
        ay = iload(kg + 1);
        if constexpr (NQ > 1 && !BF) {  // the single-query-tile sweep is HBM-bound already: extra requests only cost
          tch0 = touch(kg);
          tch1 = touch(kg + 1);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) by[q] = *reinterpret_cast<const f32x4 *>(qs + (q * KG + kg + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BF) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bx[q]), acc[q], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[q][e], acc[q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < KG) ? kg + 2 : kg;
        ax = iload(k2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) bx[q] = *reinterpret_cast<const f32x4 *>(qs + (q * KG + k2) * 256);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BF) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ay), __builtin_bit_cast(bf16x8_t, by[q]), acc[q], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], by[q][e], acc[q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < KG) {
        if constexpr (BF) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bx[q]), acc[q], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[q][e], acc[q], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);

    // fused top-k: lane owns query column (lane & 31) of each q-tile and sees 16 index rows
    // per n-tile.  Branch-lean: one max tree per q-tile against the (shared) threshold; only
    // when some lane beats it, lanes repeatedly extract their best remaining score and
    // insert it (1-2 rounds in practice) -- no per-score branches.
    const int nrow0 = tile * 32;
    const bool tail = (nrow0 + 32) > a.N;  // only the last tile has rows >= N (zero padding)
    if (w == 0 && (((tile - t0) / (SC_THREADS / 64)) & 3) == 3 && lane < 32) {
#pragma nounroll
      for (int q = 0; q < NQ; ++q) {  // rolled on purpose: runs once per 4 tiles, must not cost registers
        const float *mp = mx_s + (q * 32 + lane) * 16;
        float f = mp[0];
#pragma nounroll
        for (int j = 1; j < 16; ++j) f = fminf(f, mp[j]);
        if (f > NEG_INF) atomicMax(&thr_s[q * 32 + lane], enc(f));
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (tail) {
        // only the last tile of the index: a real (wave-uniform) branch -- the empty asm keeps the compiler from
        // if-converting it into 16 selects per query tile on EVERY tile (it did: ~300 instructions per tile)
        asm volatile("");
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = (nrow0 + mfma_row(r, lane) >= a.N) ? NEG_INF : acc[q][r];
      }
      float m = NEG_INF;
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[q][r]);
      float thr = fmaxf(ls[q][KC - 1], dec(thr_s[q * 32 + (lane & 31)]));
      if (__any(m > thr)) {
        for (;;) {
          // this lane's best remaining score and its register index (first one on ties: rows
          // ascend with r, so equal scores are taken in row order)
          const bool take = m > thr;
          int ridx = 0;
          bool found = false;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool hit = !found && (acc[q][r] == m);
            ridx = hit ? r : ridx;
            acc[q][r] = (hit && take) ? NEG_INF : acc[q][r];
            found = found || hit;
          }
          list_insert<KC>(ls[q], li[q], m, nrow0 + (ridx & 3) + 8 * (ridx >> 2) + 4 * (lane >> 5), take);
          thr = fmaxf(thr, ls[q][KC - 1]);
          m = NEG_INF;
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[q][r]);
          if (!__any(m > thr)) break;
        }
        atomicMax(&thr_s[q * 32 + (lane & 31)], enc(ls[q][KC - 1]));  // publish this list's minimum
        mx_s[(q * 32 + (lane & 31)) * 16 + w * 2 + (lane >> 5)] = ls[q][0];  // and its best
      }
    }
  }

  constexpr int WAVES = SC_THREADS / 64;
  if constexpr (MERGE) {
    // (1) the two lane halves hold lists of the same query over different rows: merge into lanes 0-31
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float os[KC];
      int oi[KC];
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        os[i] = __shfl_xor(ls[q][i], 32);
        oi[i] = __shfl_xor(li[q][i], 32);
      }
      merge_lists<KC>(ls[q], li[q], os, oi);
    }
    // (2) tree over the 8 waves through LDS (the query block is no longer needed):
    // scratch [wave][q][entry][32 queries], one (score, id) plane pair per sender wave
    __syncthreads();
    float *ms = smem;
    int *mi = reinterpret_cast<int *>(smem) + (WAVES / 2) * NQ * KC * 32;
    for (int half = WAVES / 2; half >= 1; half >>= 1) {
      if (w >= half && w < 2 * half && lane < 32) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
          for (int i = 0; i < KC; ++i) {
            ms[(((w - half) * NQ + q) * KC + i) * 32 + lane] = ls[q][i];
            mi[(((w - half) * NQ + q) * KC + i) * 32 + lane] = li[q][i];
          }
      }
      __syncthreads();
      if (w < half && lane < 32) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float os[KC];
          int oi[KC];
#pragma unroll
          for (int i = 0; i < KC; ++i) {
            os[i] = ms[((w * NQ + q) * KC + i) * 32 + lane];
            oi[i] = mi[((w * NQ + q) * KC + i) * 32 + lane];
          }
          merge_lists<KC>(ls[q], li[q], os, oi);
        }
      }
      __syncthreads();
    }
    if (w == 0 && lane < 32) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int query = (qb * NQ + q) * 32 + lane;
        if (query < a.Q) {
          float *ps = a.part_scores + ((size_t)query * a.NSPLIT + split) * KC;
          int32_t *pi = a.part_ids + ((size_t)query * a.NSPLIT + split) * KC;
#pragma unroll
          for (int i = 0; i < KC; i += 4) {
            *reinterpret_cast<f32x4 *>(ps + i) = f32x4{ls[q][i], ls[q][i + 1], ls[q][i + 2], ls[q][i + 3]};
            *reinterpret_cast<int4 *>(pi + i) = int4{li[q][i], li[q][i + 1], li[q][i + 2], li[q][i + 3]};
          }
        }
      }
    }
  } else {
    // partial lists -> global: candidate slot (split, wave, lane half)
    const int slot = (split * WAVES + w) * 2 + (lane >> 5);
    const int nslots = a.NSPLIT * WAVES * 2;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int query = (qb * NQ + q) * 32 + (lane & 31);
      if (query < a.Q) {
        float *ps = a.part_scores + ((size_t)query * nslots + slot) * KC;
        int32_t *pi = a.part_ids + ((size_t)query * nslots + slot) * KC;
#pragma unroll
        for (int i = 0; i < KC; i += 4) {
          *reinterpret_cast<f32x4 *>(ps + i) = f32x4{ls[q][i], ls[q][i + 1], ls[q][i + 2], ls[q][i + 3]};
          *reinterpret_cast<int4 *>(pi + i) = int4{li[q][i], li[q][i + 1], li[q][i + 2], li[q][i + 3]};
        }
      }
    }
  }
}

// candidate lists per query and index split
int score_slots_per_split(int merge) { return merge ? 1 : (SC_THREADS / 64) * 2; }

template <int NQ, bool MERGE, bool BF = false>
static hipError_t launch_score_variant(const ScoreArgs &a_in, hipStream_t stream) {
  size_t lds = (size_t)NQ * a_in.KG * 256 * sizeof(float);
  const size_t merge_lds = (size_t)(SC_THREADS / 128) * NQ * SC_KC * 32 * 8;
  if (MERGE && merge_lds > lds) lds = merge_lds;
  ScoreArgs a = a_in;
  a.thr_off = (int32_t)(lds / sizeof(float));  // shared thresholds live behind the query block / merge scratch
  lds += (size_t)NQ * 32 * sizeof(int) + (size_t)NQ * 32 * 16 * sizeof(float);  // thresholds + per-list best entries
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int QB = (a.QT + NQ - 1) / NQ;
  int grid;
  if (a.NSPLIT <= 8) {
    const int per = 8 / a.NSPLIT;
    grid = (QB + per - 1) / per * 8;
  } else {
    grid = QB * a.NSPLIT;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(score_topk_kernel<SC_KC, NQ, MERGE, BF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((score_topk_kernel<SC_KC, NQ, MERGE, BF>), dim3(grid), dim3(SC_THREADS), lds, stream, a);
  return hipGetLastError();
}

// diagnostic: number of queries of this call left uncertified by the bf16 pass, accumulated on the device
__global__ void count_uncert_kernel(const int32_t *cert, int Q, unsigned long long *count) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long m = __ballot(q < Q && cert[q] == 0);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}
hipError_t launch_count_uncert(const int32_t *cert, int Q, unsigned long long *count, hipStream_t st) {
  hipLaunchKernelGGL(count_uncert_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, cert, Q, count);
  return hipGetLastError();
}

// rows [R][C] fp32 -> bf16 fragment blocks [ceil(R/32)][ceil(C/16)][512 bf16]: lane (k half, row) owns 8 consecutive k
__global__ void pack_rows_bf16_kernel(const float *__restrict__ rows, int64_t R, int C, int KG16, int64_t total8,
                                      f32x4 *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg = (int)(blk % KG16);
    const int64_t r = (blk / KG16) * 32 + (l & 31);
    const int k0 = kg * 16 + (l >> 5) * 8;
    unsigned w[4] = {0, 0, 0, 0};
    if (r < R) {
      const float *src = rows + (size_t)r * C + k0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned short b = 0;
        if (k0 + j < C) {
          unsigned u = __float_as_uint(src[j]);
          u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even
          b = (unsigned short)(u >> 16);
        }
        w[j >> 1] |= (unsigned)b << ((j & 1) * 16);
      }
    }
    out[i] = __builtin_bit_cast(f32x4, u32x4{w[0], w[1], w[2], w[3]});
  }
}

hipError_t launch_pack_rows_bf16(const float *rows, int64_t R, int C, void *out, hipStream_t stream) {
  const int KG16 = (C + 15) / 16;
  const int64_t total8 = ((R + 31) / 32) * KG16 * 64;
  if (total8 == 0) return hipSuccess;
  const int64_t blocks = (total8 + 255) / 256;
  hipLaunchKernelGGL(pack_rows_bf16_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, rows, R, C, KG16,
                     total8, reinterpret_cast<f32x4 *>(out));
  return hipGetLastError();
}

// the resident fp32 fragment index [NT][KG][256 floats] -> bf16 fragment copy [NT][ceil(KG/2)][1 KiB]
__global__ void frag32_to_bf16_kernel(const f32x4 *__restrict__ in, int64_t NT, int KG, int KG16, int64_t total8,
                                      f32x4 *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg16 = (int)(blk % KG16);
    const int64_t tile = blk / KG16;
    const int kg = kg16 * 2 + (l >> 5), r = l & 31;
    f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
    if (kg < KG) {
      lo = in[((size_t)tile * KG + kg) * 64 + r];
      hi = in[((size_t)tile * KG + kg) * 64 + 32 + r];
    }
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = (j < 2) ? lo[2 * j] : hi[2 * (j - 2)], b = (j < 2) ? lo[2 * j + 1] : hi[2 * (j - 2) + 1];
      unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
      ua += 0x7FFFu + ((ua >> 16) & 1u);
      ub += 0x7FFFu + ((ub >> 16) & 1u);
      w[j] = (ua >> 16) | (ub & 0xFFFF0000u);
    }
    out[i] = __builtin_bit_cast(f32x4, u32x4{w[0], w[1], w[2], w[3]});
  }
}

hipError_t launch_frag32_to_bf16(const float *idxp, int64_t NT, int KG, void *out, hipStream_t stream) {
  const int KG16 = (KG + 1) / 2;
  const int64_t total8 = NT * KG16 * 64;
  if (total8 == 0) return hipSuccess;
  const int64_t blocks = (total8 + 255) / 256;
  hipLaunchKernelGGL(frag32_to_bf16_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream,
                     reinterpret_cast<const f32x4 *>(idxp), NT, KG, KG16, total8, reinterpret_cast<f32x4 *>(out));
  return hipGetLastError();
}

hipError_t launch_score_topk(const ScoreArgs &a, hipStream_t stream) {
  if (a.KC != SC_KC) return hipErrorInvalidValue;
  if (a.NSPLIT > 8 && (a.NSPLIT & 7)) return hipErrorInvalidValue;
  if (a.BF) {  // bf16 candidate pass: merged lists only
    if (!a.MERGE) return hipErrorInvalidValue;
    if (a.NQ == 1) return launch_score_variant<1, true, true>(a, stream);
    if (a.NQ == 4) return launch_score_variant<4, true, true>(a, stream);
    return hipErrorInvalidValue;
  }
  if (a.NQ == 1) return a.MERGE ? launch_score_variant<1, true>(a, stream) : launch_score_variant<1, false>(a, stream);
  if (a.NQ == 4) return a.MERGE ? launch_score_variant<4, true>(a, stream) : launch_score_variant<4, false>(a, stream);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// exact float64 score of query row q against index row n (frag32-packed f32 rows
// or row-major f64 rows), computed by one wave; result valid in every lane.
__device__ __forceinline__ double wave_exact_dot(const float *qrow, const float *idxp, const double *idx64,
                                                 int64_t n, int S, int KG, int lane) {
  double acc = 0.0;
  if (idx64) {
    const double *row = idx64 + (size_t)n * S;
    for (int d = lane; d < S; d += 64) acc += (double)qrow[d] * row[d];
  } else {
    const float *blk = idxp + (size_t)(n >> 5) * KG * 256;
    const int r = (int)(n & 31);
    for (int j = lane; j < KG * 2; j += 64) {  // j = kg*2 + half -> 4 consecutive dims
      const int kg = j >> 1, half = j & 1;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(blk + kg * 256 + (half * 32 + r) * 4);
      const int d0 = kg * 8 + half * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < S) acc += (double)qrow[d0 + e] * (double)v[e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  return acc;
}

__device__ __forceinline__ bool before(double sa, int64_t ia, double sb, int64_t ib) {
  return (sa > sb) || (sa == sb && ia < ib);  // score descending, then lower row id
}

// One 256-thread workgroup per query.  NC candidates (f32 score, local row id).
#define RS_THREADS 256
#define RS_MAXWIN 256
#define RS_MAXNC 4096  // candidates per query the re-scoring pass accepts
__global__ __launch_bounds__(RS_THREADS) void rescore_kernel(RescoreArgs a) {
  __shared__ int s_cnt;
  __shared__ int s_win[RS_MAXWIN];
  __shared__ double s_ex[RS_MAXWIN];
  __shared__ float s_m[RS_THREADS / 64];
  __shared__ double s_qn[RS_THREADS / 64];
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];  // [NC] (dynamic: keeps occupancy for small NC)
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (a.skip && a.skip[q] != 0) return;  // (uniform per workgroup) already final
  const float *ps = a.part_scores + (size_t)q * a.NC;
  const int32_t *pi = a.part_ids + (size_t)q * a.NC;
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;

  // |q| for the fp32 error bound eps_q = eps * |q|
  {
    double v = 0.0;
    for (int d = tid; d < a.S; d += RS_THREADS) v += (double)qrow[d] * qrow[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_qn[w] = v;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const float eps_q = a.eps * (float)sqrt(s_qn[0] + s_qn[1] + s_qn[2] + s_qn[3]);

  // k-th largest fp32 candidate: stage one sortable 64-bit key per candidate in LDS
  // (monotone score bits << 32 | inverted row id, 0 = empty slot), then every thread
  // ranks its candidates by counting larger keys (LDS broadcast reads).
  for (int c = tid; c < a.NC; c += RS_THREADS) {
    const int id = pi[c];
    unsigned u = __float_as_uint(ps[c]);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    s_key[c] = (id < 0) ? 0ull : (((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)id));
  }
  __syncthreads();
  // k-th largest key: every slot of KC candidates is a sorted list (descending), so k rounds of "largest list head
  // wins and advances" find it -- O(k * slots / 256) instead of ranking all NC candidates against each other
  // (NC = 4096 for a single query over 256 index splits: that ranking was a third of the whole pass).
  float kth = NEG_INF;
  if (a.NC <= 256) {
    // few candidates (many-queries launches: 16 per index split): every thread ranks its candidate by counting
    // larger keys (LDS broadcast reads) -- cheaper than the barriers of the tournament below
    const unsigned long long key = (tid < a.NC) ? s_key[tid] : 0ull;
    int rank = 0;
    for (int j = 0; j < a.NC; ++j) rank += (s_key[j] > key) ? 1 : 0;
    float mine = (key != 0ull && rank == a.k - 1) ? ps[tid] : NEG_INF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = fmaxf(mine, __shfl_xor(mine, o));
    __shared__ float s_thr[RS_THREADS / 64];
    if (lane == 0) s_thr[w] = mine;
    __syncthreads();
    kth = fmaxf(fmaxf(s_thr[0], s_thr[1]), fmaxf(s_thr[2], s_thr[3]));
  } else {
    __shared__ unsigned long long s_best[RS_THREADS / 64];
    const int nslots = a.NC / SC_KC;
    int head[RS_MAXNC / SC_KC / RS_THREADS];   // this thread's slots: tid, tid + 256, ...
#pragma unroll
    for (int j = 0; j < RS_MAXNC / SC_KC / RS_THREADS; ++j) head[j] = 0;
    unsigned long long kkey = 0ull;
    for (int it = 0; it < a.k; ++it) {
      unsigned long long best = 0ull;
#pragma unroll
      for (int j = 0; j < RS_MAXNC / SC_KC / RS_THREADS; ++j) {
        const int slot = tid + j * RS_THREADS;
        if (slot < nslots && head[j] < SC_KC) {
          const unsigned long long key = s_key[slot * SC_KC + head[j]];
          best = key > best ? key : best;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
      }
      if (lane == 0) s_best[w] = best;
      __syncthreads();
      unsigned long long win = s_best[0];
#pragma unroll
      for (int i = 1; i < RS_THREADS / 64; ++i) win = s_best[i] > win ? s_best[i] : win;
      __syncthreads();
      kkey = win;
      if (win == 0ull) break;                  // fewer than k candidates
#pragma unroll
      for (int j = 0; j < RS_MAXNC / SC_KC / RS_THREADS; ++j) {  // keys are unique (row id): exactly one head matches
        const int slot = tid + j * RS_THREADS;
        if (slot < nslots && head[j] < SC_KC && s_key[slot * SC_KC + head[j]] == win) ++head[j];
      }
    }
    if (kkey != 0ull) {  // decode the score bits of the k-th key
      unsigned u = (unsigned)(kkey >> 32);
      u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
      kth = __uint_as_float(u);
    }
  }

  // window: candidates whose fp32 score is within 2*eps of the k-th (the only ones
  // that can be in the exact top-k); largest slot minimum M over full slots
  float mmax = NEG_INF;
  const int KCc = SC_KC;
  for (int c = tid; c < a.NC; c += RS_THREADS) {
    const int id = pi[c];
    if (id < 0) continue;
    const float s = ps[c];
    if (s >= kth - 2.0f * eps_q) {
      const int p = atomicAdd(&s_cnt, 1);
      if (p < RS_MAXWIN) s_win[p] = c;
    }
    if ((c % KCc) == KCc - 1) mmax = fmaxf(mmax, s);  // slot is full: rows outside it score <= s
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mmax = fmaxf(mmax, __shfl_xor(mmax, o));
  if (lane == 0) s_m[w] = mmax;
  __syncthreads();
  mmax = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  const int nwin_all = s_cnt;
  const int nwin = min(nwin_all, RS_MAXWIN);

  // exact float64 re-score of the window, one wave per candidate
  for (int i = w; i < nwin; i += RS_THREADS / 64) {
    const double ex = wave_exact_dot(qrow, a.idx32, a.idx64, pi[s_win[i]], a.S, KG, lane);
    if (lane == 0) s_ex[i] = ex;
  }
  __syncthreads();

  // rank the window exactly; emit the first k
  double theta = -__builtin_inf();
  for (int i = tid; i < nwin; i += RS_THREADS) {
    const double s = s_ex[i];
    const int64_t id = pi[s_win[i]];
    int rank = 0;
    for (int j = 0; j < nwin; ++j) rank += before(s_ex[j], (int64_t)pi[s_win[j]], s, id);
    if (rank < a.k) {
      a.out_scores[(size_t)q * a.k + rank] = s;
      a.out_ids[(size_t)q * a.k + rank] = a.id_base + id;
    }
    if (rank == a.k - 1) theta = s;
  }
  // certificate: every row outside the candidate set has fp32 score <= mmax, hence
  // exact score <= mmax + eps_q; it cannot displace the exact k-th if that is < theta.
  {
    double t = theta;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = fmax(t, __shfl_xor(t, o));
    __syncthreads();
    if (lane == 0) s_ex[w] = t;
    __syncthreads();
    if (tid == 0) {
      t = fmax(fmax(s_ex[0], s_ex[1]), fmax(s_ex[2], s_ex[3]));
      const bool ok = (nwin_all <= RS_MAXWIN) && (nwin >= a.k) && ((double)mmax + (double)eps_q < t);
      a.cert[q] = ok ? 1 : 0;
    }
  }
}

// Same pass for NC <= 64 candidates per query (many-queries launches: 16 per index split, <= 4 splits), one WAVE per
// query and everything in registers -- no LDS, no barriers: lane c owns candidate c; ranks by shuffling every key past
// every lane; the float64 dots use the same wave_exact_dot as above (bit-identical scores).  A workgroup per query
// cost 0.10 ms per 16384 queries, mostly idle threads and barriers.
__global__ __launch_bounds__(256) void rescore_small_kernel(RescoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= a.Q) return;
  if (a.skip && a.skip[q] != 0) return;  // already final (uniform per wave)
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;
  double qn = 0.0;
  for (int d = lane; d < a.S; d += 64) qn += (double)qrow[d] * qrow[d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  const float eps_q = a.eps * (float)sqrt(qn);

  const bool have = lane < a.NC;
  const int id = have ? a.part_ids[(size_t)q * a.NC + lane] : -1;
  const float sc = have ? a.part_scores[(size_t)q * a.NC + lane] : NEG_INF;
  unsigned u = __float_as_uint(sc);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  const unsigned long long key = (id < 0) ? 0ull : (((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)id));
  int rank = 0;
  for (int j = 0; j < a.NC; ++j) rank += (__shfl(key, j) > key) ? 1 : 0;
  // fp32 score of the k-th best candidate
  float kth = (key != 0ull && rank == a.k - 1) ? sc : NEG_INF;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kth = fmaxf(kth, __shfl_xor(kth, o));
  // window: candidates within 2*eps of the k-th; M: largest minimum of a full slot (rows outside score <= M)
  const bool in_win = (id >= 0) && (sc >= kth - 2.0f * eps_q);
  float mmax = (id >= 0 && (lane % SC_KC) == SC_KC - 1) ? sc : NEG_INF;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mmax = fmaxf(mmax, __shfl_xor(mmax, o));
  const unsigned long long wmask = __ballot(in_win);
  const int nwin = __popcll(wmask);
  // exact float64 scores of the window members, one at a time with the whole wave
  double ex = -__builtin_inf();
  for (unsigned long long m = wmask; m; m &= m - 1) {
    const int src = __ffsll((long long)m) - 1;
    const double v = wave_exact_dot(qrow, a.idx32, a.idx64, __shfl(id, src), a.S, KG, lane);
    if (lane == src) ex = v;
  }
  // exact rank inside the window (score descending, then lower row id)
  int r2 = 0;
  for (unsigned long long m = wmask; m; m &= m - 1) {
    const int src = __ffsll((long long)m) - 1;
    r2 += before(__shfl(ex, src), (int64_t)__shfl(id, src), ex, (int64_t)id) ? 1 : 0;
  }
  if (in_win && r2 < a.k) {
    a.out_scores[(size_t)q * a.k + r2] = ex;
    a.out_ids[(size_t)q * a.k + r2] = a.id_base + id;
  }
  double theta = (in_win && r2 == a.k - 1) ? ex : -__builtin_inf();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) theta = fmax(theta, __shfl_xor(theta, o));
  if (lane == 0) a.cert[q] = ((nwin >= a.k) && ((double)mmax + (double)eps_q < theta)) ? 1 : 0;
}

hipError_t launch_rescore(const RescoreArgs &a, hipStream_t stream) {
  if (a.NC > RS_MAXNC) return hipErrorInvalidValue;
  if (a.NC <= 64) {
    hipLaunchKernelGGL(rescore_small_kernel, dim3((a.Q + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(rescore_kernel, dim3(a.Q), dim3(RS_THREADS), (size_t)a.NC * sizeof(unsigned long long), stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// float64 brute force for queries whose certificate failed (cert[q] == 0), or
// for every query when cert == nullptr.  One workgroup per query; each thread
// keeps a top-k list (k <= 16) over rows n = tid (mod 256); then a serial merge.
#define EX_THREADS 256
struct ExactArgs {
  const float *q;
  const float *idxp;
  const double *idx64;
  const int32_t *cert;
  double *out_scores;
  int64_t *out_ids;
  int64_t id_base, N;
  int32_t Q, S, k;  // k = entries emitted by this pass (<= 16)
  int32_t k_off, k_total;  // they land at columns [k_off, k_off + k) of rows of k_total columns; when
                           // k_off > 0 only rows ranked AFTER column k_off-1 are considered (next page)
};

__global__ __launch_bounds__(EX_THREADS) void exact_topk_kernel(ExactArgs a) {
  __shared__ double s_sc[EX_THREADS / 64][SC_KC];
  __shared__ int64_t s_id[EX_THREADS / 64][SC_KC];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (a.cert && a.cert[q]) return;
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;
  double ls[SC_KC];
  int64_t li[SC_KC];
#pragma unroll
  for (int i = 0; i < SC_KC; ++i) {
    ls[i] = -__builtin_inf();
    li[i] = -1;
  }
  // every wave scans rows n = w, w+4, ...; the dot is wave-cooperative, the list is
  // replicated in all lanes of the wave
  double cut_s = __builtin_inf();
  int64_t cut_id = -1;
  if (a.k_off > 0) {
    cut_s = a.out_scores[(size_t)q * a.k_total + a.k_off - 1];
    cut_id = a.out_ids[(size_t)q * a.k_total + a.k_off - 1] - a.id_base;
  }
  for (int64_t n = w; n < a.N; n += EX_THREADS / 64) {
    double s = wave_exact_dot(qrow, a.idxp, a.idx64, n, a.S, KG, lane);
    const bool after_cut = (s < cut_s) || (s == cut_s && n > cut_id);
    if (after_cut && s > ls[SC_KC - 1]) {
      int64_t id = n;
      bool ins = false;
#pragma unroll
      for (int i = 0; i < SC_KC; ++i) {
        const bool gt = ins || (s > ls[i]);
        ins = gt;
        const double tv = ls[i];
        const int64_t ti = li[i];
        ls[i] = gt ? s : tv;
        li[i] = gt ? id : ti;
        s = gt ? tv : s;
        id = gt ? ti : id;
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < SC_KC; ++i) {
      s_sc[w][i] = ls[i];
      s_id[w][i] = li[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    int pos[EX_THREADS / 64] = {0, 0, 0, 0};
    for (int o = 0; o < a.k; ++o) {
      int best = -1;
      for (int p = 0; p < EX_THREADS / 64; ++p) {
        if (pos[p] >= SC_KC || s_id[p][pos[p]] < 0) continue;
        if (best < 0 || before(s_sc[p][pos[p]], s_id[p][pos[p]], s_sc[best][pos[best]], s_id[best][pos[best]])) best = p;
      }
      a.out_scores[(size_t)q * a.k_total + a.k_off + o] = s_sc[best][pos[best]];
      a.out_ids[(size_t)q * a.k_total + a.k_off + o] = a.id_base + s_id[best][pos[best]];
      ++pos[best];
    }
  }
}

hipError_t launch_exact_topk(const float *q, const float *idxp, const double *idx64, const int32_t *cert,
                             double *out_scores, int64_t *out_ids, int64_t id_base, int64_t N, int Q, int S,
                             int k, hipStream_t stream) {
  // k <= 16: one pass (the certified-failure path).  Larger k: pages of 16, each pass a full
  // float64 sweep restricted to rows ranked after the previous page (exact, slow, rarely used:
  // the reference's consumers read <= 10 columns, sse_evaluator.py:95,112)
  for (int off = 0; off < k; off += SC_KC) {
    ExactArgs a{q, idxp, idx64, cert, out_scores, out_ids, id_base, N, Q, S, (k - off < SC_KC) ? k - off : SC_KC, off, k};
    hipLaunchKernelGGL(exact_topk_kernel, dim3(Q), dim3(EX_THREADS), 0, stream, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k-way merge of P sorted lists per query: in [P][Q][k] -> out [Q][k].
__global__ void merge_topk_kernel(const double *in_s, const int64_t *in_i, int P, int Q, int k, double *out_s,
                                  int64_t *out_i) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  // rank of element (p,j) in the merged order = #elements before it; lists are
  // sorted, so count with a scan over the other lists (P*k is small)
  for (int p = 0; p < P; ++p)
    for (int j = 0; j < k; ++j) {
      const double s = in_s[((size_t)p * Q + q) * k + j];
      const int64_t id = in_i[((size_t)p * Q + q) * k + j];
      int rank = j;
      for (int p2 = 0; p2 < P && rank < k; ++p2) {
        if (p2 == p) continue;
        for (int j2 = 0; j2 < k; ++j2) {
          const double s2 = in_s[((size_t)p2 * Q + q) * k + j2];
          const int64_t id2 = in_i[((size_t)p2 * Q + q) * k + j2];
          if (before(s2, id2, s, id)) ++rank; else break;
        }
      }
      if (rank < k) {
        out_s[(size_t)q * k + rank] = s;
        out_i[(size_t)q * k + rank] = id;
      }
    }
}

hipError_t launch_merge_topk(const double *in_s, const int64_t *in_i, int P, int Q, int k, double *out_s,
                             int64_t *out_i, hipStream_t stream) {
  hipLaunchKernelGGL(merge_topk_kernel, dim3((Q + 127) / 128), dim3(128), 0, stream, in_s, in_i, P, Q, k, out_s,
                     out_i);
  return hipGetLastError();
}
