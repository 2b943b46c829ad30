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
#include <cstdlib>

#include "sse_kernels.h"

#define SC_THREADS 512
#define SC_KC 16
#ifndef SC_PRIO_K  // wave priority inside the k-loop / in the top-k epilogue
#define SC_PRIO_K 1
#define SC_PRIO_E 0
#endif

#ifdef SSE_SCORE_CLOCK  // measurement builds (tools/): cycles per phase of a wave tile, summed over the sweep, workgroup 0
#include <cstdio>
__device__ long long g_score_clk[8 * 8];
#define SC_CLK_DECL long long ck_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = clock64();
#define SC_CLK(i)                   \
  {                                 \
    const long long n_ = clock64(); \
    ck_[i] += n_ - ck_t;            \
    ck_t = n_;                      \
  }
#else
#define SC_CLK_DECL
#define SC_CLK(i)
#endif
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

// merge of two sorted (descending) 8-lists held by a lane pair into one sorted 16-list: [A0..A7, B7..B0] is bitonic
// (in the list order), four compare-exchange stages sort it
__device__ __forceinline__ void merge8_to16(const float (&as)[8], const int (&ai)[8], const float (&bs)[8], const int (&bi)[8],
                                            float (&os)[16], int (&oi)[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    os[i] = as[i];
    oi[i] = ai[i];
    os[8 + i] = bs[7 - i];
    oi[8 + i] = bi[7 - i];
  }
#pragma unroll
  for (int stride = 8; stride > 0; stride >>= 1)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((i & stride) == 0) {
        const bool sw = entry_before(os[i + stride], oi[i + stride], os[i], oi[i]);
        const float ts = os[i];
        const int ti = oi[i];
        os[i] = sw ? os[i + stride] : ts;
        oi[i] = sw ? oi[i + stride] : ti;
        os[i + stride] = sw ? ts : os[i + stride];
        oi[i + stride] = sw ? ti : oi[i + stride];
      }
}

// deferred insertion: all lanes insert their parked entries (pend[p * 64], p < cnt) of one query tile, oldest first
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KL>
__device__ __forceinline__ void drain_parked(float (&ls)[KL], int (&li)[KL], int &cnt, const f32x2 *pend) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const bool has = cnt > p;
    if (__ballot(has) != 0ull) {
      asm volatile("");
      const f32x2 e = pend[p * 64];
      list_insert<KL>(ls, li, e[0], __float_as_int(e[1]), has);
    }
  }
  cnt = 0;
}

// 8 waves per workgroup = 2 per SIMD: there is no barrier in the sweep, so the waves drift
// apart and one wave's top-k epilogue (VALU) overlaps its partner's MFMA stream.
// Wave tile: 1 index tile (32 rows, M) x NQ query tiles (N): NQ accumulators + NQ private lists.
//   NQ = 4: 128-query block (MFMA-bound regime).  NQ = 1: <= 32 queries (demo / web, Q = 1):
//   a quarter of the MFMA work per index byte, so the sweep runs at HBM speed.
// Lists.  A lane keeps a sorted top-KL (8) list per query tile over the rows it sees (1/16 of the workgroup's rows);
//   the 16 lane lists of a query are merged in the kernel (lane pairs via shuffles into a 16-list, then an LDS tree
//   over the waves) into ONE list of KC = 16 candidates per (query, index split), together with a BOUND: every row of
//   the split that is not in the merged list scores <= bound = max(16th merged entry, the 8th entry of any FULL lane
//   list): a row a lane list evicted or refused is <= that list's final 8th entry (monotone), a row refused by the
//   shared threshold is <= the workgroup's 16th best, a row dropped by a merge is <= the merged 16th.  The re-scoring
//   pass certifies against that bound, so 8-entry lane lists cost no exactness: when one of them overflows with rows
//   that mattered (probability ~1e-6 per query on unordered data) the certificate fails and the query is recomputed.
//   (Round 1 kept 16 sorted entries per lane: 128 list registers at NQ = 4 and ~250 instructions per insertion round.)
// Epilogue of a wave tile (per query tile: 16 accumulator registers x 64 lanes against the lanes' thresholds).  It is an
//   instruction-count problem: beside the partner wave's MFMA stream an epilogue instruction costs ~12 cycles, and the
//   partner's k-loop (64 MFMAs) is all there is to hide it behind (clock64 table: profiles/r04_notes.txt).  So: the lane's
//   threshold lives in a register (shared part re-read every 4th tile); the maxima of the four register groups and of the
//   tile are 9 v_max3_f32; one compare + branch when nothing beats the threshold (60 % of the query tiles).  Otherwise
//   the groups, then the registers of a group with a hit, are tested -- the no-hit case falls through, a hit runs out of
//   line (~6 hits per wave tile in ~1.6 of the 4 query tiles).
// Deferred insertion (DEFER = the bf16 variants, whose query block leaves LDS room): a hit is almost always ONE lane's,
//   yet a select-based insertion makes all 64 lanes execute ~75 instructions.  Instead the lane parks (score, row) in
//   its own 4-entry LDS slot row (one exec-masked ds_write_b64) and keeps only its best score current (that is what the
//   shared threshold is made of); the parked entries are inserted 64 lanes in parallel when some lane's row has filled
//   up (about every 20th tile of a query tile) and before the final merge; takers that found their row full (start of a
//   sweep, rows sorted by score) are inserted directly right after that.  Entries enter the lists in row order, exactly as
//   they would have; bounds and candidates are unchanged.  The fp32 variants insert with selects (list_insert): their
//   MFMAs are 4x longer per tile and hide it.
// Tiles are dealt dynamically (LDS counter) and the threshold fold is spread over the waves: see the tile loop.
// The accumulators start from the MFMA's zero C operand (first k-group), not from 16 moves per query tile.
// BF: the candidate pass runs on v_mfma_f32_32x32x16_bf16 -- index and queries are bf16 copies in the same 1-KiB
// block / 16-byte-per-lane fragment scheme (a block now holds 32 rows x 16 k, a.KG counts 16-k groups), ONE MFMA per
// (k-group, query tile).  Only candidate SELECTION sees bf16: the float64 re-scoring pass works on the fp32 / f64
// rows with an error bound widened to the bf16 rounding, so the results stay exact (or fall back, certified).
// COLLECT: no lists at all -- every row whose score reaches the query's threshold a.col_thr[query] is appended to the
// query's buffer (a.col_buf[slot], atomic counter a.col_cnt[slot]); queries with a.col_slot[query] < 0 are skipped.
// This is the grid-wide exact path for what the lists cannot certify (ties / duplicates at the k-th score, k > 16):
// the threshold is a proven lower bound of the fp32 score of every exact top-k row, so the buffer provably holds the
// exact top-k (select_topk_kernel re-scores and sorts it in float64).
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
#define SC_KL 8
#define SC_PEND 4  // parked entries per (wave, query tile, lane)
// Final merge of a workgroup's lane lists (shared by the sweep kernels): (1) lane pairs 8 + 8 -> 16, (2) LDS tree over the 8
// waves, (3) wave 0 writes the KC candidates of (query, split) and the bound every other row of the split stays under.
template <int NQ>
__device__ __forceinline__ void merge_lane_lists(const ScoreArgs &a, float (&ls)[NQ][SC_KL], int (&li)[NQ][SC_KL], float *smem, int w,
                                                 int lane, int qb, int split) {
  constexpr int KC = SC_KC, KL = SC_KL;
  constexpr int WAVES = SC_THREADS / 64;
  // (1) the two lane halves hold lists of the same query over different rows: merge into a 16-list in lanes 0-31;
  //     bnd = largest 8th entry of a full lane list (see the header comment)
  float ms16[NQ][KC];
  int mi16[NQ][KC];
  float bnd[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float os[KL];
    int oi[KL];
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      os[i] = __shfl_xor(ls[q][i], 32);
      oi[i] = __shfl_xor(li[q][i], 32);
    }
    merge8_to16(ls[q], li[q], os, oi, ms16[q], mi16[q]);
    bnd[q] = fmaxf(ls[q][KL - 1], os[KL - 1]);
  }
  // (2) tree over the 8 waves through LDS (the query block is no longer needed):
  // scratch [wave][q][entry][32 queries], one (score, id) plane pair per sender wave, then the bounds
  __syncthreads();
  float *ms = smem;
  int *mi = reinterpret_cast<int *>(smem) + (WAVES / 2) * NQ * KC * 32;
  float *mb = smem + 2 * (WAVES / 2) * NQ * KC * 32;  // [wave][q][32]
  for (int half = WAVES / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half && lane < 32) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int i = 0; i < KC; ++i) {
          ms[(((w - half) * NQ + q) * KC + i) * 32 + lane] = ms16[q][i];
          mi[(((w - half) * NQ + q) * KC + i) * 32 + lane] = mi16[q][i];
        }
        mb[((w - half) * NQ + q) * 32 + lane] = bnd[q];
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
        merge_lists<KC>(ms16[q], mi16[q], os, oi);
        bnd[q] = fmaxf(bnd[q], mb[(w * NQ + q) * 32 + lane]);
      }
    }
    __syncthreads();
  }
  const int Qeff = a.q_count ? min(a.Q, *a.q_count) : a.Q;
  if (w == 0 && lane < 32) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int query = (qb * NQ + q) * 32 + lane;
      if (query < Qeff) {
        float *ps = a.part_scores + ((size_t)query * a.NSPLIT + split) * KC;
        int32_t *pi = a.part_ids + ((size_t)query * a.NSPLIT + split) * KC;
#pragma unroll
        for (int i = 0; i < KC; i += 4) {
          *reinterpret_cast<f32x4 *>(ps + i) = f32x4{ms16[q][i], ms16[q][i + 1], ms16[q][i + 2], ms16[q][i + 3]};
          *reinterpret_cast<int4 *>(pi + i) = int4{mi16[q][i], mi16[q][i + 1], mi16[q][i + 2], mi16[q][i + 3]};
        }
        // every row of this split outside the 16 candidates scores <= this
        a.part_bnd[(size_t)query * a.NSPLIT + split] = fmaxf(bnd[q], ms16[q][KC - 1]);
      }
    }
  }
}

struct TagZ { static constexpr bool value = true; };   // "accumulators start from zero" / "accumulate" tags of mma()
struct TagA { static constexpr bool value = false; };
template <int NQ, bool BF, bool COLLECT, bool RINGED>
__global__ __launch_bounds__(SC_THREADS) void score_topk_kernel(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // query block [NQ][KG][256]; later merge scratch
  constexpr int KL = SC_KL;
  constexpr bool DEFER = BF && !COLLECT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR): tile numbers and load offsets derive from it
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
  // q_count (second chance of the bf16 pass): the queries are a compacted set whose size is only known on the device; the
  // launch is sized for the worst case and the workgroups past the set return at once
  const int Qeff = a.q_count ? min(a.Q, *a.q_count) : a.Q;
  if (qb * NQ * 32 >= Qeff) return;
  if (COLLECT && a.lane_max == nullptr) {  // collect pass: only blocks with an open query run
    int open_q = 0;
    for (int i = tid; i < NQ * 32; i += SC_THREADS) {
      const int qq = qb * NQ * 32 + i;
      if (qq < a.Q && a.col_slot[qq] >= 0) open_q = 1;
    }
    if (!__syncthreads_or(open_q)) return;
  }

  // stage the query block (already fragment-packed, [query tile][k-group][1 KiB]) into LDS as [k-group][query tile]:
  // the NQ fragments of one k-group are 1 KiB apart and consecutive k-groups NQ KiB, so every read in the unrolled
  // k-loop is ONE base register + an immediate offset (with [q][kg] the NQ*KG addresses each took a VGPR)
  if ((NQ == 1 || COLLECT) && a.q_rows != nullptr) {
    // latency path (and every collect sweep): fragments straight from the row-major fp32 queries (same values as
    // launch_pack_rows / launch_pack_rows_bf16 produce: lane (half, row) owns 4 / 8 consecutive k, bf16 rounded to nearest
    // even); LDS order [k-group][query tile]
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    const int Sd = a.S;
    for (int i = tid; i < NQ * KG * 64; i += SC_THREADS) {
      const int kg = (i >> 6) / NQ, l = i & 63, row = (qb * NQ + (i >> 6) % NQ) * 32 + (l & 31);
      f32x4 v = {0, 0, 0, 0};
      if (row < Qeff) {
        if constexpr (BF) {
          const int k0 = kg * 16 + (l >> 5) * 8;
          const float *src = a.q_rows + (size_t)row * Sd + k0;
          unsigned wd[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            unsigned b = 0;
            if (k0 + j < Sd) {
              unsigned u = __float_as_uint(src[j]);
              u += 0x7FFFu + ((u >> 16) & 1u);
              b = u >> 16;
            }
            wd[j >> 1] |= b << ((j & 1) * 16);
          }
          v = __builtin_bit_cast(f32x4, u32x4{wd[0], wd[1], wd[2], wd[3]});
        } else {
          const int k0 = kg * 8 + (l >> 5) * 4;
          const float *src = a.q_rows + (size_t)row * Sd + k0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + e < Sd) v[e] = src[e];
        }
      }
      dst[i] = v;
    }
  } else {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.qp) + (size_t)qb * NQ * KG * 64;
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    const int valid = min(NQ, (Qeff + 31) / 32 - qb * NQ) * KG * 64;
    for (int i = tid; i < NQ * KG * 64; i += SC_THREADS) {
      const int blk = i >> 6, qt = blk / KG, kg = blk - qt * KG;
      dst[(kg * NQ + qt) * 64 + (i & 63)] = (i < valid) ? src[i] : f32x4{0, 0, 0, 0};
    }
  }

  float ls[NQ][KL];
  int li[NQ][KL];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      ls[q][i] = NEG_INF;
      li[q][i] = -1;
    }
  // COLLECT: this lane's query threshold and buffer slot per query tile
  float cthr[NQ];
  int cslot[NQ];
  // MAX-ONLY mode of the collect variant (a.lane_max set; first pass of the two-pass path for mid-size indexes, see
  // launch_lane_max_threshold): no lists, no buffers -- a lane keeps the largest score it has seen per query tile.
  float lmax[NQ];
  if constexpr (COLLECT) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int query = (qb * NQ + q) * 32 + (lane & 31);
      lmax[q] = NEG_INF;
      cslot[q] = (query < a.Q && a.lane_max == nullptr) ? a.col_slot[query] : -1;
      cthr[q] = (cslot[q] >= 0) ? a.col_thr[query] : __builtin_inff();
    }
  }

  // Shared per-query insertion threshold (LDS, order-preserving int encoding of the float): a lower bound of the
  // workgroup's 16th best score -- the SMALLEST of the 16 lane lists' best entries (16 distinct rows score at least
  // that).  Every list keeps its current best in mx_s[q-tile][list][query] (monotone, plain stores); wave 0 folds the minimum
  // into thr_s every few tiles.  Racy reads only see older, smaller -- still valid -- values.  Rows below it can never
  // reach the merged top-16, so every lane may use it as its threshold: the 16 lists of a query share their progress.
  // (16 words in front of them: the workgroup's tile counter, see the tile loop)
  int *tile_ctr = reinterpret_cast<int *>(smem + (size_t)a.thr_off);
  int *thr_s = tile_ctr + 16;
  auto enc = [](float f) -> int { const int i = __float_as_int(f); return i >= 0 ? i : (i ^ 0x7FFFFFFF); };
  auto dec = [](int i) -> float { return __int_as_float(i >= 0 ? i : (i ^ 0x7FFFFFFF)); };
  float *mx_s = reinterpret_cast<float *>(thr_s + NQ * 32);  // [NQ][16 lists][32 queries]: lanes of a store / of the fold
                                                               // hit 32 different banks ([query][list] was a 16-way conflict)
  if constexpr (!COLLECT) {
    for (int i = tid; i < NQ * 32; i += SC_THREADS) thr_s[i] = enc(NEG_INF);
    for (int i = tid; i < 2 * NQ * 32 * 16; i += SC_THREADS) mx_s[i] = NEG_INF;  // (both planes: bests and second bests)
  }
  if (tid == 0) *tile_ctr = SC_THREADS / 64;  // tiles 0 .. 7 of the split are the waves' first ones
  __syncthreads();
  // DEFER: parked hits [wave][q-tile][entry][lane] (score, row), this lane's fill count and best score per query tile
  float *mx2_s = mx_s + NQ * 32 * 16;  // the lists' second-best entries, same layout (the tight threshold below)
  f32x2 *pend_s = reinterpret_cast<f32x2 *>(mx2_s + NQ * 32 * 16) + (size_t)w * NQ * SC_PEND * 64 + lane;
  int pcnt[NQ];
  float lbest[NQ], lbest2[NQ];  // best and second-best score among the rows this lane took (two distinct rows)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    pcnt[q] = 0;
    lbest[q] = NEG_INF;
    lbest2[q] = NEG_INF;
  }

  const int tps = (a.NT + a.NSPLIT - 1) / a.NSPLIT;  // n-tiles per split
  const int t0 = split * tps, t1 = min(a.NT, t0 + tps);
  const float *qs = smem + lane * 4;
  const int voff = lane * 16;

  // k-group product: acc (+)= A fragment x the NQ query fragments; ZERO: the accumulators start from the MFMA's
  // inline-constant C operand
  auto mma = [&](f32x16 (&acc)[NQ], const f32x4 &av, const f32x4 (&bv)[NQ], auto zero_tag) {
    constexpr bool ZERO = decltype(zero_tag)::value;
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (BF) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv[q]),
                                                         ZERO ? z : acc[q], 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[q][e], (ZERO && e == 0) ? z : acc[q], 0, 0, 0);
    }
  };

  // Tiles are handed out dynamically: a wave takes the split's next tile from an LDS counter (one tile ahead: the tile after
  // the one it works on is known when it starts -- the index ring below runs into it).  With the tiles dealt round-robin
  // the two waves of a SIMD did not finish together: the older one wins the matrix pipe whenever both want it and was done
  // ~12 % before its partner, which then ran alone (instrumented build: 5.7 k against 6.5 k cycles per tile).  A wave's
  // tiles still ascend, so a lane's rows still ascend (equal scores keep the earlier row).
  // Index fragments: a ring of RING k-groups per wave in registers, running ACROSS tile boundaries (the slot used for
  // k-group kg is refilled with k-group kg + RING of the same tile, or the head of this wave's next tile), through one
  // buffer descriptor for the tile and one for the next (per-lane offset: the constant 16*lane).  RING k-groups in
  // flight cover an HBM miss at either matrix rate: fp32 16 MFMAs (1 k cycles) per k-group, bf16 4 MFMAs (128 cycles).
  // Needs KG % RING == 0 (RINGED, chosen by the launcher); other shapes (small encodings) use the two-set pipelined
  // loop below.  A template parameter, not a run-time branch: with both loops in one kernel the compiler's wait-count
  // bookkeeping merges their pending loads and drains the ring (vmcnt(0)) in front of every epilogue.
  constexpr int RING = 8;
  f32x4 ring[RING];
  const int tile0 = t0 + w;
  const int tail_tile = (a.N & 31) ? (int)(a.N >> 5) : -1;
  if (RINGED && tile0 < t1) {
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.idxp) + (size_t)__builtin_amdgcn_readfirstlane(tile0) * KG * 256, 0, KG * 1024, 0x00020000);
#pragma unroll
    for (int d = 0; d < RING; ++d)
      ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, voff, d * 1024, 0));
  }
  auto grab_tile = [&]() -> int {  // lane 0's value counts: t0 + readfirstlane() where it is needed (not here: no wait here)
    int got = 0;
    if (lane == 0) got = atomicAdd(tile_ctr, 1);
    return got;
  };

  // RINGED: the query fragments of the k-group in work; loaded here for the first tile, then carried from tile to tile (the
  // last k-group of a tile requests the first of the next).
  // (Requesting them two k-groups ahead -- a third fragment set for half of the query tiles, 251 registers -- measured
  // 4.57 against 4.59 ms: not kept.)
  f32x4 bq[NQ];
  if constexpr (RINGED) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(qs + q * 256);
  }
  float thr_r[NQ];  // this lane's insertion thresholds (see the epilogue)
#pragma unroll
  for (int q = 0; q < NQ; ++q) thr_r[q] = NEG_INF;
  SC_CLK_DECL
  int nxt = (tile0 < t1) ? t0 + __builtin_amdgcn_readfirstlane(grab_tile()) : t1;
  int it = 0;  // this wave's tile count
  int after = 0;
  for (int tile = tile0; tile < t1; tile = nxt, nxt = t0 + __builtin_amdgcn_readfirstlane(after), ++it) {
    SC_CLK(0)
    after = grab_tile();  // (needed at the end of this tile only: the LDS round trip hides)
    if constexpr (!COLLECT) {
      // Tight threshold: the 16th largest of the 32 values {best, second best of the 16 lists} of a query is a lower bound
      // of the workgroup's 16th best score too (two distinct rows per list) and -- unlike the minimum of the 16 bests,
      // which sits near the 54th best -- nearly always IS it: a third of the hits.  It costs a 32-value sort per query (two
      // queries at a time across the wave, shuffles), so it runs on a doubling schedule: wave w for query tile w % NQ
      // at its tile counts 4, 8, 16, ... (the second wave of a query tile at 6, 12, 24, ...).  Here, in front of the k-loop:
      // the accumulators are dead, registers are free.
      const int itq = (w / NQ) & 1 ? it / 3 : it;
      if (it >= 4 && (itq & (itq - 1)) == 0 && ((w / NQ) & 1 ? itq * 3 == it : true)) {
        const int fq = w % NQ, j = lane & 31;
        const float *plane = (j < 16 ? mx_s : mx2_s) + (fq * 16 + (j & 15)) * 32 + (lane >> 5);
#pragma nounroll
        for (int x0 = 0; x0 < 32; x0 += 2) {
          float v = plane[x0];  // value j of query x0 + (lane >> 5)
          // bitonic sort, descending, across the 32 lanes of a half
#pragma unroll
          for (int k = 2; k <= 32; k <<= 1)
#pragma unroll
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
              const float o = __shfl_xor(v, jj);
              const bool keep_max = ((j & jj) == 0) == ((j & k) == 0 || k == 32);
              v = keep_max ? fmaxf(v, o) : fminf(v, o);
            }
          if (j == 15 && v > NEG_INF) atomicMax(&thr_s[fq * 32 + x0 + (lane >> 5)], enc(v));
        }
      }
    }
    const int utile = __builtin_amdgcn_readfirstlane(tile);
    const bool more = nxt < t1;  // this wave has a next tile
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.idxp) + (size_t)utile * KG * 256, 0, KG * 1024, 0x00020000);
    // the next tile (the ring's refills of the last RING k-groups; without one: this tile again, never used)
    const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.idxp) + (size_t)(more ? nxt : utile) * KG * 256, 0, KG * 1024, 0x00020000);
    auto iload = [&](int off_kg) -> f32x4 {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ir, voff, off_kg * 1024, 0));
    };
    f32x16 acc[NQ];

    if constexpr (RINGED) {
      f32x4 bqn[NQ];  // (bq: the query fragments of k-group 0 arrive with the previous tile's last k-group, see above the loop)
      __builtin_amdgcn_s_setprio(SC_PRIO_K);
      auto ring_block = [&](int kg0, auto zero_tag) {
        // the tile's last RING k-groups: the refills are the next tile's first RING (one descriptor select per block)
        const bool last = kg0 + RING >= KG;
        const __amdgpu_buffer_rsrc_t rr = last ? nr : ir;
        const int rbase_kg = last ? 0 : kg0 + RING;
        const float *qcur = qs + (size_t)kg0 * NQ * 256;            // query fragments of k-group kg0
        const float *qnext = last ? qs : qcur + RING * NQ * 256;  // ... of the next block (or the next tile's first)
#pragma unroll
        for (int d = 0; d < RING; ++d) {
          // next k-group's query fragments (LDS) first, then this k-group's MFMAs, then the refill of the slot
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            bqn[q] = *reinterpret_cast<const f32x4 *>(d + 1 < RING ? qcur + ((d + 1) * NQ + q) * 256 : qnext + q * 256);
          __builtin_amdgcn_sched_barrier(0);
          if (d == 0) mma(acc, ring[d], bq, zero_tag);
          else mma(acc, ring[d], bq, TagA{});
          __builtin_amdgcn_sched_barrier(0);
          // refill the slot with the fragment RING k-groups on: same tile, or the head of this wave's next tile
          ring[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, voff, (rbase_kg + d) * 1024, 0));
#pragma unroll
          for (int q = 0; q < NQ; ++q) bq[q] = bqn[q];
        }
      };
      ring_block(0, TagZ{});
      for (int kg0 = RING; kg0 < KG; kg0 += RING) ring_block(kg0, TagA{});
    } else {
      // k-loop, hand software-pipelined with two named operand sets (X / Y): the index
      // fragment of k-group kg+1 (global) and the query fragments (LDS) are in flight while
      // kg's MFMAs issue; no register copies.
      f32x4 ax = iload(0), ay;
      f32x4 bx[NQ], by[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) bx[q] = *reinterpret_cast<const f32x4 *>(qs + q * 256);
      __builtin_amdgcn_s_setprio(SC_PRIO_K);
      int kg = 0;
      if (KG >= 2) {  // first pair: starts the accumulators from zero
        ay = iload(1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) by[q] = *reinterpret_cast<const f32x4 *>(qs + (NQ + q) * 256);
        __builtin_amdgcn_sched_barrier(0);
        mma(acc, ax, bx, TagZ{});
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (2 < KG) ? 2 : 0;
        ax = iload(k2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) bx[q] = *reinterpret_cast<const f32x4 *>(qs + (k2 * NQ + q) * 256);
        __builtin_amdgcn_sched_barrier(0);
        mma(acc, ay, by, TagA{});
        __builtin_amdgcn_sched_barrier(0);
        kg = 2;
        for (; kg + 1 < KG; kg += 2) {
          ay = iload(kg + 1);
#pragma unroll
          for (int q = 0; q < NQ; ++q) by[q] = *reinterpret_cast<const f32x4 *>(qs + ((kg + 1) * NQ + q) * 256);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc, ax, bx, TagA{});
          __builtin_amdgcn_sched_barrier(0);
          const int k3 = (kg + 2 < KG) ? kg + 2 : kg;
          ax = iload(k3);
#pragma unroll
          for (int q = 0; q < NQ; ++q) bx[q] = *reinterpret_cast<const f32x4 *>(qs + (k3 * NQ + q) * 256);
          __builtin_amdgcn_sched_barrier(0);
          mma(acc, ay, by, TagA{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kg < KG) mma(acc, ax, bx, TagA{});
      } else {
        mma(acc, ax, bx, TagZ{});
      }
    }
    __builtin_amdgcn_s_setprio(SC_PRIO_E);
    SC_CLK(1)

    const int nrow0 = tile * 32;
    if constexpr (COLLECT) {
      if (a.lane_max != nullptr) {
        // max-only: five v_max3_f32 per query tile; the zero padding rows of the index's last tile are masked (wave-uniform branch)
        const bool tail = (tile == tail_tile);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (tail) {
            int nb = nrow0 + 4 * (lane >> 5);
            asm volatile("" : "+v"(nb));
            const int nlim = (int)a.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = (nb + (r & 3) + 8 * (r >> 2) >= nlim) ? NEG_INF : acc[q][r];
          }
          float m = lmax[q];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            m = fmaxf(fmaxf(m, acc[q][4 * g4]), acc[q][4 * g4 + 1]);
            m = fmaxf(fmaxf(m, acc[q][4 * g4 + 2]), acc[q][4 * g4 + 3]);
          }
          lmax[q] = m;
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float m = NEG_INF;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[q][r]);
        if (__any(m >= cthr[q])) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = nrow0 + mfma_row(r, lane);
            if (acc[q][r] >= cthr[q] && row < a.N) {  // rows >= N are the zero padding of the last tile
              const int pos = atomicAdd(a.col_cnt + cslot[q], 1);
              if (pos < a.col_cap) a.col_buf[(size_t)cslot[q] * a.col_cap + pos] = row;
            }
          }
        }
      }
    } else {
      // fused top-k: lane owns query column (lane & 31) of each q-tile and sees 16 index rows per n-tile
      const bool tail = (tile == tail_tile);  // only the last tile of the index can have rows >= N (zero padding)
      // threshold fold, spread over the waves: wave w folds query tile w % NQ every fourth tile of its own (the two waves
      // of a query tile two tiles apart), all 64 lanes: lane (half, query) takes the minimum over 8 of the 16 list bests,
      // four LDS reads in flight at a time, the halves meet through one shuffle.  (One wave folding everything -- 64
      // dependent LDS round trips every fourth tile -- cost that wave 1.8 k cycles per tile: with the static tile
      // assignment it finished ~10 % after the others.)
      if (((it + 2 * (w / NQ)) & 3) == 1) {
        const int fq = w % NQ;
        const float *mp = mx_s + (fq * 16 + (lane >> 5) * 8) * 32 + (lane & 31);
        float f = fminf(fminf(mp[0], mp[32]), fminf(mp[64], mp[96]));
        f = fminf(f, fminf(fminf(mp[128], mp[160]), fminf(mp[192], mp[224])));
        f = fminf(f, __shfl_xor(f, 32));
        if (lane < 32 && f > NEG_INF) atomicMax(&thr_s[fq * 32 + lane], enc(f));
      }
      const int rbase = nrow0 + 4 * (lane >> 5);  // row of accumulator register r: rbase + (r & 3) + 8 * (r >> 2)
      // This lane's thresholds live in registers (thr_r: the larger of its list's last entry and the shared threshold);
      // the shared part is re-read every fourth tile only -- thresholds only rise, an older one is a valid, lower one.
      // The epilogue is an instruction-count problem: beside the partner wave's MFMA stream a wave issues ~5 instructions
      // per 32-cycle MFMA, so every instruction here is ~6 - 8 cycles of a phase that the partner's k-loop (64 MFMAs)
      // has to cover (tools: -DSSE_SCORE_CLOCK).
      if ((it & 3) == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) thr_r[q] = fmaxf(thr_r[q], dec(thr_s[q * 32 + (lane & 31)]));
      }
      SC_CLK(2)
#ifdef SSE_SCORE_MEASURE  // measurement builds only (tools/): k-loop without the top-k epilogue, results are garbage
      if (a.dbg & 1) continue;
#endif
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (tail) {
          // only the last tile of the index: a real (wave-uniform) branch -- the asm makes the row base opaque, which
          // keeps the compiler from if-converting the masking into 16 selects per query tile on EVERY tile and from
          // hoisting the 16 row numbers (and their registers) out of the branch
          int nb = rbase;
          asm volatile("" : "+v"(nb));
          const int nlim = (int)a.N;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] = (nb + (r & 3) + 8 * (r >> 2) >= nlim) ? NEG_INF : acc[q][r];
        }
        // maxima of the four register groups (rows 0-3, 8-11, 16-19, 24-27 of this lane's half), then of the tile: ten
        // instructions when nothing beats the threshold.  (Written as chains from a constant so that every step becomes
        // one v_max3_f32 on the raw MFMA results: pairwise fmaxf() makes the compiler quiet both inputs first.)
        float gm[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          gm[g4] = fmaxf(fmaxf(NEG_INF, acc[q][4 * g4]), acc[q][4 * g4 + 1]);
          gm[g4] = fmaxf(fmaxf(gm[g4], acc[q][4 * g4 + 2]), acc[q][4 * g4 + 3]);
        }
        const float m = fmaxf(fmaxf(fmaxf(gm[0], gm[1]), gm[2]), gm[3]);
        SC_CLK(5)
        if (__ballot(m > thr_r[q]) != 0ull) {
#ifdef SSE_SCORE_CLOCK
          ck_[7] += 1;
#endif
          asm volatile("");
#ifdef SSE_SCORE_MEASURE
          if (a.dbg & 2) continue;
#endif
          // (~6 hits per wave tile, in ~1.6 of its 4 query tiles.)  Registers ascend with the row number, '>' keeps the
          // earlier row on equal scores.
          float thr = thr_r[q];
          int rb = rbase;
          asm volatile("" : "+v"(rb));  // (keeps the 16 row numbers from being formed in front of the branch, on every tile)
          if constexpr (DEFER) {
            // every taker with room in its parked row parks; a lane whose row is full skips (the rows are looked at below)
            const int pc0 = pcnt[q];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              if (__builtin_expect(__ballot(gm[g4] > thr) == 0ull, 1)) continue;  // (the usual case falls through)
              asm volatile("");  // keep it a branch
#pragma unroll
              for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
                const bool take = acc[q][r] > thr;
                if (__builtin_expect(__ballot(take) == 0ull, 1)) continue;
                asm volatile("");
#ifdef SSE_SCORE_MEASURE
                if (a.dbg & 4) continue;
#endif
                if (take && pcnt[q] < SC_PEND) {
                  pend_s[(q * SC_PEND + pcnt[q]) * 64] = f32x2{acc[q][r], __int_as_float(rb + (r & 3) + 8 * (r >> 2))};
                  pcnt[q] += 1;
                  lbest2[q] = fmaxf(lbest2[q], fminf(fminf(lbest[q], -NEG_INF), acc[q][r]));
                  lbest[q] = fmaxf(fmaxf(lbest[q], NEG_INF), acc[q][r]);
                }
              }
            }
            if (__builtin_expect(__ballot(pcnt[q] >= SC_PEND) != 0ull, 0)) {
              // some parked row is full now (about every 20th tile of a query tile; at the start of a sweep, or over rows
              // sorted by score, with takers left over): everybody inserts what is parked; then the takers that did not fit
              // -- a lane's takers beyond the `quota` it parked just now, counted against the threshold they were parked
              // under -- go into the lists directly.  Every entry once, in row order.
              asm volatile("");
              const int quota = pcnt[q] - pc0;
              const float thr0 = thr;
              drain_parked<KL>(ls[q], li[q], pcnt[q], pend_s + q * SC_PEND * 64);
              thr = fmaxf(thr, ls[q][KL - 1]);
              int seen = 0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const bool tk = acc[q][r] > thr0;
                seen += tk ? 1 : 0;
                const bool ins = tk && seen > quota && acc[q][r] > thr;
                if (__ballot(ins) != 0ull) {
                  asm volatile("");
                  list_insert<KL>(ls[q], li[q], acc[q][r], rb + (r & 3) + 8 * (r >> 2), ins);
                  thr = fmaxf(thr, ls[q][KL - 1]);
                  lbest[q] = fmaxf(lbest[q], ls[q][0]);  // (after the drain the list's first two ARE the lane's best two rows)
                  lbest2[q] = fmaxf(lbest2[q], ls[q][1]);
                }
              }
            }
          } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              if (__builtin_expect(__ballot(gm[g4] > thr) == 0ull, 1)) continue;  // (the usual case falls through)
              asm volatile("");  // keep it a branch
#pragma unroll
              for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
                const bool take = acc[q][r] > thr;
                if (__builtin_expect(__ballot(take) == 0ull, 1)) continue;
                asm volatile("");
                list_insert<KL>(ls[q], li[q], acc[q][r], rb + (r & 3) + 8 * (r >> 2), take);
                thr = fmaxf(thr, ls[q][KL - 1]);
              }
            }
          }
          thr_r[q] = thr;
          // publish this list's best
#ifdef SSE_SCORE_MEASURE
          if (!(a.dbg & 8))
#endif
          mx_s[(q * 16 + w * 2 + (lane >> 5)) * 32 + (lane & 31)] = DEFER ? lbest[q] : ls[q][0];
          mx2_s[(q * 16 + w * 2 + (lane >> 5)) * 32 + (lane & 31)] = DEFER ? lbest2[q] : ls[q][1];
          SC_CLK(6)
        }
      }
    }
  }
  if constexpr (COLLECT) {
    if (a.lane_max != nullptr) {
      // [query][split][16 lists]: list = (wave, lane half) -- the rows a lane saw are disjoint from every other lane's of its query
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int query = (qb * NQ + q) * 32 + (lane & 31);
        if (query < Qeff) a.lane_max[((size_t)query * a.NSPLIT + split) * 16 + w * 2 + (lane >> 5)] = lmax[q];
      }
    }
    return;
  }
#ifdef SSE_SCORE_CLOCK
  SC_CLK(3)
  if (blockIdx.x == 0 && lane == 0 && NQ == 4) {
    ck_[4] = it;
    for (int i = 0; i < 8; ++i) g_score_clk[w * 8 + i] = ck_[i];
  }
#endif
  if constexpr (DEFER) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) drain_parked<KL>(ls[q], li[q], pcnt[q], pend_s + q * SC_PEND * 64);
  }

  merge_lane_lists<NQ>(a, ls, li, smem, w, lane, qb, split);
}

// dynamic LDS of one workgroup: query block (re-used as merge scratch) + shared thresholds + per-list bests (+ parked hits)
static size_t score_lds_layout(int NQ, int KG, bool BF, bool COLLECT, int32_t *thr_off) {
  size_t lds = (size_t)NQ * KG * 256 * sizeof(float);
  const size_t merge_lds = (size_t)(SC_THREADS / 128) * NQ * (SC_KC * 2 + 1) * 32 * 4;  // (score, id) planes + bounds
  if (!COLLECT && merge_lds > lds) lds = merge_lds;
  if (thr_off) *thr_off = (int32_t)(lds / sizeof(float));  // shared thresholds live behind the query block / merge scratch
  lds += 64 + (size_t)NQ * 32 * sizeof(int) + 2 * (size_t)NQ * 32 * 16 * sizeof(float);  // tile counter, thresholds + per-list best and second-best entries
  if (BF && !COLLECT) lds += (size_t)(SC_THREADS / 64) * NQ * SC_PEND * 64 * 8;  // parked hits (deferred insertion)
  return lds;
}
size_t score_lds_bytes(int NQ, int KG, int BF, int COLLECT) { return score_lds_layout(NQ, KG, BF != 0, COLLECT != 0, nullptr); }

// Query tiles per workgroup for an index of dimension S.  <= 32 queries: one tile (HBM-bound sweep).  Otherwise the
// largest of 4 / 2 / 1 whose query block fits the 160 KiB of LDS in EVERY variant a call may launch (fp32 candidates,
// fp32 collect, and -- with_bf16 -- the bf16 candidate pass with its parked-hit rows): 4 up to S = 296, 2 up to S = 616
// (BASELINE configs[4]: S = 512), 1 up to SSE_MAX_INDEX_DIM.  0: S does not fit at all.
int score_pick_nq(int Q, int S, int with_bf16) {
  const int KG = (S + 7) / 8, KG16 = (S + 15) / 16;
  const size_t cap = 160 * 1024;
  for (int nq = (Q <= 32) ? 1 : 4; nq >= 1; nq >>= 1) {
    if (score_lds_bytes(nq, KG, 0, 0) > cap || score_lds_bytes(nq, KG, 0, 1) > cap) continue;
    if (with_bf16 && score_lds_bytes(nq, KG16, 1, 0) > cap) continue;
    return nq;
  }
  return 0;
}

template <int NQ, bool BF, bool COLLECT, bool RINGED>
static hipError_t launch_score_ringed(const ScoreArgs &a_in, hipStream_t stream) {
  ScoreArgs a = a_in;
  const size_t lds = score_lds_layout(NQ, a.KG, BF, COLLECT, &a.thr_off);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int QB = (a.QT + NQ - 1) / NQ;
  int grid;
  if (a.NSPLIT <= 8) {
    const int per = 8 / a.NSPLIT;
    grid = (QB + per - 1) / per * 8;
  } else {
    grid = QB * a.NSPLIT;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(score_topk_kernel<NQ, BF, COLLECT, RINGED>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((score_topk_kernel<NQ, BF, COLLECT, RINGED>), dim3(grid), dim3(SC_THREADS), lds, stream, a);
#ifdef SSE_SCORE_CLOCK
  if (NQ == 4 && !COLLECT) {
    static int n = 0;
    if (n++ % 8 == 3) {
      long long v[64];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_score_clk), sizeof v);
      for (int w = 0; w < 8; ++w) {
        const double t = (double)(v[w * 8 + 4] > 0 ? v[w * 8 + 4] : 1);
        fprintf(stderr, "[score clock <%d,%d> KG=%d] wave %d: %lld tiles, cycles per tile: loop top %.0f | k-loop %.0f | fold + thresholds %.0f | compares %.0f | hit paths %.0f (%.2f query tiles with hits per wave tile) | sum %.0f\n",
                NQ, (int)BF, a.KG, w, v[w * 8 + 4], v[w * 8 + 0] / t, v[w * 8 + 1] / t, v[w * 8 + 2] / t, v[w * 8 + 5] / t, v[w * 8 + 6] / t, v[w * 8 + 7] / t,
                (v[w * 8 + 0] + v[w * 8 + 1] + v[w * 8 + 2] + v[w * 8 + 5] + v[w * 8 + 6]) / t);
      }
    }
  }
#endif
  return hipGetLastError();
}

template <int NQ, bool BF, bool COLLECT>
static hipError_t launch_score_variant(const ScoreArgs &a, hipStream_t stream) {
  return (a.KG % 8 == 0) ? launch_score_ringed<NQ, BF, COLLECT, true>(a, stream) : launch_score_ringed<NQ, BF, COLLECT, false>(a, stream);
}

// Second chance of the bf16 candidate pass: the uncertified queries are gathered into a dense set (slot -> query in qmap,
// size in *count, rows copied to qc[slot][S], zero rows up to a whole 32-query tile behind the set) so that the fp32 sweep
// costs what THEY cost -- with the certified queries only skipped per 64- / 128-query block, 4 % of them scattered over the
// blocks (index dimension 512, random rows) re-ran the whole fp32 sweep: 81 ms for a 10 ms bf16 pass.
__global__ void compact_uncert_kernel(const int32_t *cert, int Q, int32_t *qmap, int32_t *count) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool open_q = q < Q && cert[q] == 0;
  const unsigned long long m = __ballot(open_q);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (open_q) qmap[base + __popcll(m & ((1ull << lane) - 1ull))] = q;
}
__global__ void gather_query_rows_kernel(const float *q, const int32_t *qmap, const int32_t *count, int Q, int S, float *qc) {
  const int slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = *count, padded = min(Q, (n + 31) & ~31);
  if (slot >= padded) return;
  const float *src = slot < n ? q + (size_t)qmap[slot] * S : nullptr;
  for (int d = lane; d < S; d += 64) qc[(size_t)slot * S + d] = src ? src[d] : 0.0f;
}
hipError_t launch_compact_uncert(const float *q, const int32_t *cert, int Q, int S, int32_t *qmap, int32_t *count, float *qc,
                                 hipStream_t st) {
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(compact_uncert_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, cert, Q, qmap, count);
  hipLaunchKernelGGL(gather_query_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, q, qmap, count, Q, S, qc);
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

// ---------------------------------------------------------------------------
// Small-index scorer (see SmallIndexArgs).  16384 queries x 571 targets -- the scoring half of the headline step, the
// evaluator's shape (sse_evaluator.py:104-112) -- took 110 us in the list sweep (+ 7 us of packing) for ~30 us of fp32 MFMA
// work: 18 index tiles are 9 tiles on the 8 waves of a workgroup, and every cold tile pays ~50 k cycles of select-insertions
// into empty lists.  Here a 4-wave workgroup owns 32 queries, holds their fragments in REGISTERS (32 k-groups x 4), forms all
// N scores into an LDS tile [32 queries][NP] (74 KiB at 571 rows: two workgroups per CU, one's selection under the other's
// MFMAs) and selects per query.
// Selection (a wave: four queries side by side): a lane holds rows lane, lane + 64, ... of a query's score row; the rows that
// reach a threshold 16 .. 24 of the 64 lanes' maxima reach (>= 16 rows, ~25 as a rule) are moved to lanes 0 .. n-1 with ds_permute,
// ranked against each other (score descending, lower row first) and the best 16 written in order.
// (A first version -- per-lane sorted lists and 16 rounds of a wave-wide arg-max -- cost ~1300 VALU instructions per query,
// as much as the list sweep's insertions: 110 us.  profiles/r04_notes.txt)
__device__ __forceinline__ int si_key(float f) {  // signed-integer order == float order
  const int i = __float_as_int(f);
  return i >= 0 ? i : (i ^ 0x7FFFFFFF);
}
__device__ __forceinline__ float si_val(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7FFFFFFF)); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float x) {  // lanes the row mask leaves out (or without a source lane) see their own value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_f(float x) {  // row_ror 8 / 4 / 2 / 1, row_bcast15 / 31: lane 63 holds the result
  x = fmaxf(x, dpp_f<0x128, 0xF>(x));
  x = fmaxf(x, dpp_f<0x124, 0xF>(x));
  x = fmaxf(x, dpp_f<0x122, 0xF>(x));
  x = fmaxf(x, dpp_f<0x121, 0xF>(x));
  x = fmaxf(x, dpp_f<0x142, 0xA>(x));
  x = fmaxf(x, dpp_f<0x143, 0xC>(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
template <int JMAX>  // score registers per lane this instance holds (J <= JMAX)
#ifdef SSE_SCORE_CLOCK
#define SI_CLK_PARAMS , long long (&ck_)[8], long long &ck_t
#define SI_CLK_ARGS , ck_, ck_t
#else
#define SI_CLK_PARAMS
#define SI_CLK_ARGS
#endif
__device__ __forceinline__ void si_select4(const SmallIndexArgs &a, const float *sc, int NP, int NT, int qt, int q0, int lane SI_CLK_PARAMS) {
  constexpr int QW = 4;
  const int J = (NT * 32 + 63) / 64;  // score registers per lane (uniform, <= 16)
  float v[QW][JMAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int row = lane + 64 * j;
#pragma unroll
    for (int c = 0; c < QW; ++c) v[c][j] = (j < J && row < NT * 32) ? sc[(q0 + c) * NP + row] : NEG_INF;
  }
  // Threshold: ANY value that 16 .. 24 of the 64 lanes' maxima reach.  That many lanes hold a row >= it, so it is a lower bound
  // of the 16th best score, and a tight one (the best 16 rows sit in ~14 different lanes): ~20 - 30 rows reach it, and they
  // are ranked exactly below.  Found by bisection on the order-preserving integer keys between the smallest and the largest
  // lane maximum, ONE compare + ballot count per step, ~6 steps.  (Bisecting on all rows for the exact 16th best -- nine
  // compares and a DPP sum per step -- and bisecting on the lane maxima to convergence, 32 steps, were each half of the
  // selection.)  Fewer than 16 lanes with a row (N < 16): it ends on -inf, every row is taken.
  int lok[QW], hik[QW];
  float lm[QW];
  bool done[QW];
#pragma unroll
  for (int c = 0; c < QW; ++c) {
    float m = NEG_INF;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) m = fmaxf(m, v[c][j]);
    lm[c] = m;
    lok[c] = si_key(-wave_max_f(-m));    // the smallest lane maximum: all 64 reach it
    hik[c] = si_key(wave_max_f(m)) + 1;  // nothing reaches this (exclusive; scores are finite: no overflow)
    done[c] = false;
  }
  SC_CLK(5)
  for (;;) {
    bool all_done = true;
#pragma unroll
    for (int c = 0; c < QW; ++c) {
      const unsigned width = (unsigned)hik[c] - (unsigned)lok[c];
      const int mid = lok[c] + (int)(width >> 1);
      const int cnt = __popcll(__ballot(lm[c] >= si_val(mid)));
      const bool open = !done[c] && width > 1u;
      const bool ge = cnt >= SC_KC;
      lok[c] = (open && ge) ? mid : lok[c];
      hik[c] = (open && !ge) ? mid : hik[c];
      done[c] = !open || (ge && cnt <= 24);
      all_done = all_done && done[c];
    }
    if (all_done) break;
  }
  SC_CLK(6)
#pragma unroll
  for (int c = 0; c < QW; ++c) {
    const int query = qt * 32 + q0 + c;
    if (query < a.Q) {  // (uniform)
      const float T = si_val(lok[c]);
      // the winners (>= T, masked rows left out) to lanes 0 .. n-1, in row order; non-winners push to lane 63 (unused)
      float cs = NEG_INF;
      int cr = -1, base = 0;
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        const bool take = (j < J) & (v[c][j] >= T) & (v[c][j] > NEG_INF);
        const unsigned long long mask = __ballot(take);
        const int n = __popcll(mask);
        // (no branch on n: the nine pushes and their waits then batch; with no taker nobody receives)
        const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        const int dst = (take && slot < 63) ? slot : 63;
        const int gs = __builtin_amdgcn_ds_permute(dst * 4, __float_as_int(v[c][j]));
        const int gr = __builtin_amdgcn_ds_permute(dst * 4, lane + 64 * j);
        const bool recv = (lane >= base) & (lane < base + n) & (lane < 63);
        cs = recv ? __int_as_float(gs) : cs;
        cr = recv ? gr : cr;
        base += n;
      }
      const int nw = base < 63 ? base : 63;  // (more than 63 rows tied into the selection: the bound says so below)
      int rank = 0;
      for (int l = 0; l < nw; ++l) {
        const float os = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), l));
        const int orow = __builtin_amdgcn_readlane(cr, l);
        rank += ((os > cs) | ((os == cs) & (orow < cr))) ? 1 : 0;
      }
      const bool mine = lane < nw;
      if (mine && rank < SC_KC) {
        a.part_scores[(size_t)query * SC_KC + rank] = cs;
        a.part_ids[(size_t)query * SC_KC + rank] = cr;
      }
      if (lane >= nw && lane < SC_KC) {  // fewer than 16 rows in the index: empty slots
        a.part_scores[(size_t)query * SC_KC + lane] = NEG_INF;
        a.part_ids[(size_t)query * SC_KC + lane] = -1;
      }
      const unsigned long long m16 = __ballot(mine && rank == SC_KC - 1);
      const float last = m16 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), __ffsll((long long)m16) - 1)) : NEG_INF;
      if (lane == 0) a.part_bnd[query] = (base > 63) ? __builtin_inff() : last;
    }
  }
}

template <int KGC>
__global__ __launch_bounds__(256) void score_small_index_kernel(SmallIndexArgs a) {
  extern __shared__ __attribute__((aligned(16))) float si_sc[];  // scores [32 queries][NP]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = a.NT, NP = NT * 32 + 1;  // odd row stride: the 32 queries of a store hit 32 banks
  const int qt = blockIdx.x;
  float *sc = si_sc;
  SC_CLK_DECL
  // this lane's query fragments (lane (half, query): 4 consecutive k per k-group), straight from the fp32 rows
  f32x4 b[KGC];
  {
    const int row = qt * 32 + (lane & 31), kb = (lane >> 5) * 4;
    const float *src = a.q_rows + (size_t)(row < a.Q ? row : 0) * a.S + kb;
    const bool vec = (a.S & 3) == 0 && (reinterpret_cast<uintptr_t>(a.q_rows) & 15) == 0 && KGC * 8 <= a.S;  // (uniform)
#pragma unroll
    for (int kg = 0; kg < KGC; ++kg) {
      f32x4 v = {0, 0, 0, 0};
      if (vec) {
        v = *reinterpret_cast<const f32x4 *>(src + kg * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kg * 8 + kb + e < a.S) v[e] = src[kg * 8 + e];
      }
      if (row >= a.Q) v = f32x4{0, 0, 0, 0};
      b[kg] = v;
    }
  }
  SC_CLK(0)
  // all N scores of the 32 queries: wave w takes tiles w, w + 4, ...; index fragments from L2 four k-groups ahead (pinned:
  // the compiler otherwise sinks the refills behind the block and waits for them at once); a lane ends up with query
  // (lane & 31) and 16 rows of the tile, as in the list sweep
  for (int t = w; t < NT; t += 4) {
    const f32x4 *ap = reinterpret_cast<const f32x4 *>(a.idxp) + (size_t)t * KGC * 64 + lane;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 ar[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) ar[d] = ap[(d < KGC ? d : 0) * 64];
#pragma unroll
    for (int kg = 0; kg < KGC; ++kg) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kg & 3][e], b[kg][e], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ar[kg & 3] = ap[(kg + 4 < KGC ? kg + 4 : kg) * 64];  // (the last ones re-read themselves: never used)
    }
    const int rb = t * 32 + 4 * (lane >> 5);
    float *col = sc + (lane & 31) * NP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb + (r & 3) + 8 * (r >> 2);
      col[row] = (row < a.N) ? acc[r] : NEG_INF;  // (the zero padding of the last tile)
    }
  }
  SC_CLK(1)
  __syncthreads();
  SC_CLK(2)
  if (NT * 32 <= 640) {  // (571 rows: 9 registers per lane and query instead of 16)
    si_select4<10>(a, sc, NP, NT, qt, w * 8, lane SI_CLK_ARGS);
    SC_CLK(3)
    si_select4<10>(a, sc, NP, NT, qt, w * 8 + 4, lane SI_CLK_ARGS);
    SC_CLK(4)
#ifdef SSE_SCORE_CLOCK
    if (blockIdx.x == gridDim.x / 2 && lane == 0)
      for (int i = 0; i < 8; ++i) g_score_clk[w * 8 + i] = ck_[i];
#endif
  } else {
    si_select4<16>(a, sc, NP, NT, qt, w * 8, lane SI_CLK_ARGS);
    si_select4<16>(a, sc, NP, NT, qt, w * 8 + 4, lane SI_CLK_ARGS);
  }
}
bool score_small_index_applies(int Q, int KG, int64_t NT) {
  // many queries only: a single query's workgroup would walk all tiles alone (0.09 ms against 0.05 for the list sweep)
  // index dimensions 249 .. 256 (configs[1]), 57 .. 64 (the reference's default encoding_size, sse_train.py:67), 49 .. 56 (its
  // crosslingual recipe, makefile:42): the k-groups are a template parameter (query fragments in registers)
  return Q >= 1024 && (KG == 32 || KG == 8 || KG == 7) && NT * 32 <= 1024 && (size_t)32 * (NT * 32 + 1) * sizeof(float) <= (size_t)144 * 1024;  // (1024 rows: 128.1 KiB of the CU's 160 KiB -> one workgroup per CU; up to 608 rows two fit)
}
hipError_t launch_score_small_index(const SmallIndexArgs &a, hipStream_t stream) {
  if (!score_small_index_applies(a.Q, a.KG, a.NT)) return hipErrorInvalidValue;
  const size_t lds = (size_t)32 * (a.NT * 32 + 1) * sizeof(float);
  const dim3 grid((a.Q + 31) / 32), block(256);
  auto go = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);
    return hipGetLastError();
  };
  if (a.KG == 8) return go(score_small_index_kernel<8>);
  if (a.KG == 7) return go(score_small_index_kernel<7>);
  const hipError_t e32 = go(score_small_index_kernel<32>);
  if (e32 != hipSuccess) return e32;
#ifdef SSE_SCORE_CLOCK
  if (a.Q >= 8192) {
    static int n = 0;
    if (n++ % 8 == 3) {
      long long v[64];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_score_clk), sizeof v);
      for (int w = 0; w < 4; w += 3)
        fprintf(stderr, "[small index clock Q=%d NT=%d] wave %d cycles: query fragments %lld | scores (MFMA + LDS stores) %lld | barrier %lld | selection: loads + bounds %lld | bisection %lld | winners + rank + stores %lld (both batches)\n",
                a.Q, a.NT, w, v[w * 8 + 0], v[w * 8 + 1], v[w * 8 + 2], v[w * 8 + 5], v[w * 8 + 6], v[w * 8 + 3] + v[w * 8 + 4]);
    }
  }
#endif
  return hipGetLastError();
}

hipError_t launch_score_topk(const ScoreArgs &a_in, hipStream_t stream) {
  ScoreArgs a = a_in;
#ifdef SSE_SCORE_MEASURE
  {
    static const int dbg = getenv("SSE_SCORE_DBG") ? atoi(getenv("SSE_SCORE_DBG")) : 0;
    a.dbg = dbg;
  }
#endif
  if (a.KC != SC_KC) return hipErrorInvalidValue;
  if (a.q_rows && ((a.NQ != 1 && !a.COLLECT) || a.S < 1)) return hipErrorInvalidValue;
  if (a.NSPLIT > 8 && (a.NSPLIT & 7)) return hipErrorInvalidValue;
  if (a.NSPLIT < 8 && (8 % a.NSPLIT)) return hipErrorInvalidValue;
  if (a.COLLECT) {
    // collect pass.  fp32 scores (the thresholds are fp32 bounds of the exact k-th score), or -- the two-pass path for mid-size
    // indexes, NQ = 4 -- bf16 scores against thresholds widened by the bf16 bound; lane_max set: its max-only first pass
    if (!a.lane_max && (!a.col_thr || !a.col_slot || !a.col_cnt || !a.col_buf)) return hipErrorInvalidValue;
    if (a.BF) return a.NQ == 4 ? launch_score_variant<4, true, true>(a, stream) : hipErrorInvalidValue;
    if (a.NQ == 1) return launch_score_variant<1, false, true>(a, stream);
    if (a.NQ == 2) return launch_score_variant<2, false, true>(a, stream);
    if (a.NQ == 4) return launch_score_variant<4, false, true>(a, stream);
    return hipErrorInvalidValue;
  }
  if (!a.part_bnd) return hipErrorInvalidValue;
  if (a.BF) {
    if (a.NQ == 1) return launch_score_variant<1, true, false>(a, stream);
    if (a.NQ == 2) return launch_score_variant<2, true, false>(a, stream);
    if (a.NQ == 4) return launch_score_variant<4, true, false>(a, stream);
    return hipErrorInvalidValue;
  }
  if (a.NQ == 1) return launch_score_variant<1, false, false>(a, stream);
  if (a.NQ == 2) return launch_score_variant<2, false, false>(a, stream);
  if (a.NQ == 4) return launch_score_variant<4, false, false>(a, stream);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// exact float64 scores of query row q against NB index rows n[0..NB) (frag32-packed f32 rows or row-major f64 rows),
// computed by one wave; results valid in every lane.  The rows' loads are independent and in flight together (a window of
// ~12 candidates re-scored one row at a time was a chain of 12 HBM round trips); every row's sum is formed in exactly the
// order of the single-row form below, which is this template with NB = 1: all passes produce bit-identical scores.
template <int NB>
__device__ __forceinline__ void wave_exact_dot_n(const float *qrow, const float *idxp, const double *idx64, const int64_t (&n)[NB],
                                                 int S, int KG, int lane, double (&out)[NB]) {
  double acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0.0;
  if (idx64) {
    for (int d = lane; d < S; d += 64) {
      double rv[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) rv[b] = idx64[(size_t)n[b] * S + d];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] += (double)qrow[d] * rv[b];
    }
  } else {
    for (int j = lane; j < KG * 2; j += 64) {  // j = kg*2 + half -> 4 consecutive dims
      const int kg = j >> 1, half = j & 1;
      f32x4 v[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        v[b] = *reinterpret_cast<const f32x4 *>(idxp + (size_t)(n[b] >> 5) * KG * 256 + kg * 256 + (half * 32 + (int)(n[b] & 31)) * 4);
      const int d0 = kg * 8 + half * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < S) {
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] += (double)qrow[d0 + e] * (double)v[b][e];
        }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] += __shfl_xor(acc[b], o);
#pragma unroll
  for (int b = 0; b < NB; ++b) out[b] = acc[b];
}
__device__ __forceinline__ double wave_exact_dot(const float *qrow, const float *idxp, const double *idx64,
                                                 int64_t n, int S, int KG, int lane) {
  const int64_t nn[1] = {n};
  double out[1];
  wave_exact_dot_n<1>(qrow, idxp, idx64, nn, S, KG, lane, out);
  return out[0];
}

// The same NB sums with the cross-lane reduction PACKED: the first log2(NB) butterfly steps (xor 32, 16, ...) also halve
// the number of values a lane carries -- a lane keeps the half its lane bit selects and adds what its partner sends of it --
// so NB values cost (NB - 1) + (6 - log2 NB) shuffles instead of 6 NB.  Every partial sum is the one the plain xor butterfly
// (32, 16, 8, 4, 2, 1) forms, from the same two operands (IEEE addition commutes): the totals are bit-identical to
// wave_exact_dot's.  Returns the total of value packed_value_of_lane<NB>(lane) (all lanes of that value agree).
template <int NB>
__device__ __forceinline__ int packed_value_of_lane(int lane) {
  int v = 0;
#pragma unroll
  for (int cnt = NB, bit = 5; cnt > 1; cnt >>= 1, --bit) v = v * 2 + ((lane >> bit) & 1);
  return v;
}
template <int NB>
__device__ __forceinline__ int packed_lane_of_value(int v) {  // the first lane that ends up with value v
  int lane = 0;
#pragma unroll
  for (int cnt = NB, bit = 5; cnt > 1; cnt >>= 1, --bit) lane |= ((v / (cnt / 2)) & 1) << bit;
  return lane;
}
template <int NB>
__device__ __forceinline__ double wave_exact_dot_packed(const float *qrow, const float *idxp, const double *idx64, const int64_t (&n)[NB],
                                                        int S, int KG, int lane, const float *idx_rm = nullptr) {
  static_assert(NB == 2 || NB == 4 || NB == 8, "power of two");
  double acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0.0;
  if (idx64) {
    for (int d = lane; d < S; d += 64) {
      double rv[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) rv[b] = idx64[(size_t)n[b] * S + d];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] += (double)qrow[d] * rv[b];
    }
  } else {
    for (int j = lane; j < KG * 2; j += 64) {  // j = kg*2 + half -> 4 consecutive dims
      const int kg = j >> 1, half = j & 1;
      f32x4 v[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        v[b] = idx_rm ? *reinterpret_cast<const f32x4 *>(idx_rm + (size_t)n[b] * S + j * 4)  // (dimensions 4j .. 4j+3 either way)
                      : *reinterpret_cast<const f32x4 *>(idxp + (size_t)(n[b] >> 5) * KG * 256 + kg * 256 + (half * 32 + (int)(n[b] & 31)) * 4);
      const int d0 = kg * 8 + half * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < S) {
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] += (double)qrow[d0 + e] * (double)v[b][e];
        }
    }
  }
  int m = 32;
#pragma unroll
  for (int cnt = NB; cnt > 1; cnt >>= 1, m >>= 1) {
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < cnt / 2; ++i) {
      const double send = up ? acc[i] : acc[i + cnt / 2];
      const double keep = up ? acc[i + cnt / 2] : acc[i];
      acc[i] = keep + __shfl_xor(send, m);
    }
  }
#pragma unroll
  for (; m > 0; m >>= 1) acc[0] += __shfl_xor(acc[0], m);
  return acc[0];
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
  __shared__ int s_wid[RS_MAXWIN];
  __shared__ double s_ex[RS_MAXWIN];
  __shared__ float s_m[RS_THREADS / 64];
  __shared__ double s_qn[RS_THREADS / 64];
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];  // [NC] (dynamic: keeps occupancy for small NC)
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (a.q_count && q >= *a.q_count) return;  // compacted second chance: slots past the set (uniform per workgroup)
  const int qo = a.qmap ? a.qmap[q] : q;     // where this slot's results go
  const float *ps = a.part_scores + (size_t)q * a.NC;
  const int32_t *pi = a.part_ids + (size_t)q * a.NC;
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;

  // this thread's candidates (c = tid, tid + 256, ...) and split bounds into registers: every global read of the pass
  // that does not depend on another one is issued here, in front of the first wait (a single query's call is a chain of
  // dependent round trips: each one removed is ~2 us of a ~0.15 ms call)
  constexpr int PER = RS_MAXNC / RS_THREADS;
  float rs[PER];
  int ri[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = tid + j * RS_THREADS;
    ri[j] = (c < a.NC) ? pi[c] : -1;
    rs[j] = (c < a.NC) ? ps[c] : NEG_INF;
  }
  // M = largest per-split bound (rows outside the candidates score <= M)
  float mmax = NEG_INF;
  for (int c = tid; c < a.NC / SC_KC; c += RS_THREADS) mmax = fmaxf(mmax, a.part_bnd[(size_t)q * (a.NC / SC_KC) + c]);

  // |q| for the fp32 error bound eps_q = eps * |q|
  {
    double v = 0.0;
    for (int d = tid; d < a.S; d += RS_THREADS) v += (double)qrow[d] * qrow[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_qn[w] = v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mmax = fmaxf(mmax, __shfl_xor(mmax, o));
  if (lane == 0) s_m[w] = mmax;
  if (tid == 0) s_cnt = 0;

  // k-th largest fp32 candidate: stage one sortable 64-bit key per candidate in LDS
  // (monotone score bits << 32 | inverted row id, 0 = empty slot)
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = tid + j * RS_THREADS;
    if (c < a.NC) {
      unsigned u = __float_as_uint(rs[j]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      s_key[c] = (ri[j] < 0) ? 0ull : (((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ri[j]));
    }
  }
  __syncthreads();
  const float qnorm = (float)sqrt(s_qn[0] + s_qn[1] + s_qn[2] + s_qn[3]);
  const float eps_q = a.eps * qnorm;
  mmax = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
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
    float mine = (key != 0ull && rank == a.k - 1) ? rs[0] : NEG_INF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = fmaxf(mine, __shfl_xor(mine, o));
    __shared__ float s_thr[RS_THREADS / 64];
    if (lane == 0) s_thr[w] = mine;
    __syncthreads();
    kth = fmaxf(fmaxf(s_thr[0], s_thr[1]), fmaxf(s_thr[2], s_thr[3]));
  } else {
    // every wave plays the tournament over ITS slots on its own (shuffles, no barrier), k rounds; the k winners of the
    // four waves meet in LDS and are ranked against each other once: the k-th largest of those is the k-th largest of all
    __shared__ unsigned long long s_best[(RS_THREADS / 64) * SC_KC];
    const int nslots = a.NC / SC_KC;
    // slots of this thread: w * 64 + lane + j * 256 -- consecutive slots per wave; any partition works
    int head[RS_MAXNC / SC_KC / RS_THREADS];
#pragma unroll
    for (int j = 0; j < RS_MAXNC / SC_KC / RS_THREADS; ++j) head[j] = 0;
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
      unsigned long long win = best;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(win, o);
        win = other > win ? other : win;
      }
      if (lane == 0) s_best[w * SC_KC + it] = win;
      if (win == 0ull) {  // this wave's slots are exhausted (wave-uniform)
        for (int r = it + 1; r < a.k; ++r)
          if (lane == 0) s_best[w * SC_KC + r] = 0ull;
        break;
      }
#pragma unroll
      for (int j = 0; j < RS_MAXNC / SC_KC / RS_THREADS; ++j) {  // keys are unique (row id): exactly one head matches
        const int slot = tid + j * RS_THREADS;
        if (slot < nslots && head[j] < SC_KC && s_key[slot * SC_KC + head[j]] == win) ++head[j];
      }
    }
    __syncthreads();
    // rank the 4 * k wave winners (<= 64 keys, one per lane of every wave -- all waves compute the same)
    const int nk = (RS_THREADS / 64) * a.k;
    const unsigned long long key = (lane < nk) ? s_best[(lane / a.k) * SC_KC + lane % a.k] : 0ull;
    int rank = 0;
    for (int j = 0; j < nk; ++j) rank += (__shfl(key, j) > key) ? 1 : 0;
    unsigned long long kkey = (key != 0ull && rank == a.k - 1) ? key : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(kkey, o);
      kkey = other > kkey ? other : kkey;
    }
    if (kkey != 0ull) {  // decode the score bits of the k-th key
      unsigned u = (unsigned)(kkey >> 32);
      u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
      kth = __uint_as_float(u);
    }
  }

  // window: candidates whose fp32 score is within 2*eps of the k-th (the only ones
  // that can be in the exact top-k)
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (ri[j] >= 0 && rs[j] >= kth - 2.0f * eps_q) {
      const int p = atomicAdd(&s_cnt, 1);
      if (p < RS_MAXWIN) s_wid[p] = ri[j];
    }
  }
  __syncthreads();
  const int nwin_all = s_cnt;
  const int nwin = min(nwin_all, RS_MAXWIN);

  // exact float64 re-score of the window: a wave takes 4 candidates at a time
  for (int i0 = w * 4; i0 < nwin; i0 += (RS_THREADS / 64) * 4) {
    int64_t n[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) n[c] = s_wid[min(i0 + c, nwin - 1)];
    const double tot = wave_exact_dot_packed<4>(qrow, a.idx32, a.idx64, n, a.S, KG, lane, a.idx_rm);
    const int c = packed_value_of_lane<4>(lane);
    if ((lane & 15) == 0 && i0 + c < nwin) s_ex[i0 + c] = tot;
  }
  __syncthreads();

  // rank the window exactly; emit the first k
  double theta = -__builtin_inf();
  for (int i = tid; i < nwin; i += RS_THREADS) {
    const double s = s_ex[i];
    const int64_t id = s_wid[i];
    int rank = 0;
    for (int j = 0; j < nwin; ++j) rank += before(s_ex[j], (int64_t)s_wid[j], s, id);
    if (rank < a.k) {
      a.out_scores[(size_t)qo * a.k + rank] = s;
      a.out_ids[(size_t)qo * a.k + rank] = a.id_base + id;
      if (a.host_flag) {
        a.host_scores[(size_t)qo * a.k + rank] = s;
        a.host_ids[(size_t)qo * a.k + rank] = a.id_base + id;
      }
    }
    if (rank == a.k - 1) theta = s;
  }
  if (a.host_flag) __threadfence_system();  // this thread's mirror stores are out before the barriers in front of the flag
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
      a.cert[qo] = ok ? 1 : 0;
      // t = exact score of the k-th best candidate (-inf with fewer than k): a lower bound of the true k-th best, so
      // every exact top-k row has fp32 score >= t - eps32*|q| (the collect pass gathers exactly those)
      if (a.col_thr) a.col_thr[qo] = ok ? __builtin_inff() : __double2float_rd(t - (double)(a.eps32 * qnorm));
      if (a.col_slot) {
        a.col_slot[qo] = ok ? -1 : qo;
        a.col_cnt[qo] = 0;
      }
      if (a.host_flag) {
        a.host_cert[qo] = ok ? 1 : 0;
        if (qo == 0 && a.host_err) *a.host_err = a.err_in ? *a.err_in : 0;
        __hip_atomic_store(a.host_flag + qo, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Same pass for NC <= 64 candidates per query (many-queries launches: 16 per index split, <= 4 splits), one WAVE per
// query and everything in registers -- no LDS arrays, no barriers: lane c owns candidate c.  Round 4 (16384 queries x 571
// targets, the evaluator's shape, spent 76 us here = 3 % of the headline step): wave-uniform lane indices are read with
// v_readlane instead of shuffles (the rank loops, the k-th candidate, the bounds, theta); the window members are moved to
// lanes 0 .. nwin-1 with one ds_permute; their float64 dots are formed RS_NB rows at a time with the packed reduction
// (wave_exact_dot_packed: 7 shuffles per 4 rows instead of 24) and land in the members' lanes with one more shuffle.
// The float64 scores are bit-identical to wave_exact_dot's (same sums, same order).
#ifndef RS_NB
#define RS_NB 4  // rows per float64 dot batch of rescore_small_kernel (8: 130 registers, half the occupancy)
#endif
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ __launch_bounds__(256) void rescore_small_kernel(RescoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= a.Q) return;
  if (a.q_count && q >= *a.q_count) return;  // compacted second chance: slots past the set (uniform per wave)
  const int qo = a.qmap ? a.qmap[q] : q;
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;
#ifdef SSE_SCORE_CLOCK
  long long rk_[6] = {0, 0, 0, 0, 0, 0}, rk_t = clock64();
#define RS_CLK(i) { const long long n_ = clock64(); rk_[i] += n_ - rk_t; rk_t = n_; }
#else
#define RS_CLK(i)
#endif
  // candidates and bounds first: their loads are in flight under the norm
  const bool have = lane < a.NC;
  const int id = have ? a.part_ids[(size_t)q * a.NC + lane] : -1;
  const float sc = have ? a.part_scores[(size_t)q * a.NC + lane] : NEG_INF;
  const float bnd = (lane < a.NC / SC_KC) ? a.part_bnd[(size_t)q * (a.NC / SC_KC) + lane] : NEG_INF;
  double qn = 0.0;
  for (int d = lane; d < a.S; d += 64) qn += (double)qrow[d] * qrow[d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  const float qnorm = (float)sqrt(qn);
  const float eps_q = a.eps * qnorm;
  RS_CLK(0)

  unsigned u = __float_as_uint(sc);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  const unsigned long long key = (id < 0) ? 0ull : (((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)id));
  const int key_lo = (int)(unsigned)key, key_hi = (int)(unsigned)(key >> 32);
  int rank = 0;
  for (int j = 0; j < a.NC; ++j) {
    const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(key_hi, j) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readlane(key_lo, j);
    rank += (kj > key) ? 1 : 0;
  }
  // fp32 score of the k-th best candidate (keys are unique: one lane at most holds rank k - 1)
  const unsigned long long kmask = __ballot(key != 0ull && rank == a.k - 1);
  const float kth = kmask ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), __ffsll((long long)kmask) - 1)) : NEG_INF;
  // window: candidates within 2*eps of the k-th; M: largest per-split bound (rows outside the candidates score <= M)
  const bool in_win = (id >= 0) && (sc >= kth - 2.0f * eps_q);
  float mmax = NEG_INF;
  for (int j = 0; j < a.NC / SC_KC; ++j) mmax = fmaxf(mmax, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bnd), j)));
  const unsigned long long wmask = __ballot(in_win);
  const int nwin = __popcll(wmask);
  // the window members to lanes 0 .. nwin-1, in candidate order
  const int mi = __popcll(wmask & ((1ull << lane) - 1ull));
  const int idm = __builtin_amdgcn_ds_permute((in_win ? mi : 63) * 4, id);  // (lane 63 is a member's only with all 64 in the window)
  RS_CLK(1)
  // exact float64 scores, RS_NB members at a time (padding: member 0 again)
  double exm = -__builtin_inf();
  for (int m0 = 0; m0 < nwin; m0 += RS_NB) {
    int64_t n[RS_NB];
#pragma unroll
    for (int c = 0; c < RS_NB; ++c) n[c] = __builtin_amdgcn_readlane(idm, (m0 + c < nwin) ? m0 + c : 0);
    const double tot = wave_exact_dot_packed<RS_NB>(qrow, a.idx32, a.idx64, n, a.S, KG, lane, a.idx_rm);
    const int want = lane - m0;  // this lane's member of the batch
    const double got = __shfl(tot, packed_lane_of_value<RS_NB>(want & (RS_NB - 1)));
    if (want >= 0 && want < RS_NB && lane < nwin) exm = got;
  }
  RS_CLK(2)
  // exact rank inside the window (score descending, then lower row id)
  int r2 = 0;
  for (int j = 0; j < nwin; ++j)
    r2 += before(readlane_f64(exm, j), (int64_t)__builtin_amdgcn_readlane(idm, j), exm, (int64_t)idm) ? 1 : 0;
  const bool member = lane < nwin;
  if (member && r2 < a.k) {
    a.out_scores[(size_t)qo * a.k + r2] = exm;
    a.out_ids[(size_t)qo * a.k + r2] = a.id_base + idm;
    if (a.host_flag) {
      a.host_scores[(size_t)qo * a.k + r2] = exm;
      a.host_ids[(size_t)qo * a.k + r2] = a.id_base + idm;
    }
  }
  const unsigned long long tmask = __ballot(member && r2 == a.k - 1);
  const double theta = tmask ? readlane_f64(exm, __ffsll((long long)tmask) - 1) : -__builtin_inf();
  if (a.host_flag) __threadfence_system();
  RS_CLK(3)
#ifdef SSE_SCORE_CLOCK
  if (q == a.Q / 2 && lane == 0) {
    rk_[4] = nwin;
    for (int i = 0; i < 6; ++i) g_score_clk[i] = rk_[i];
  }
#endif
  if (lane == 0) {
    const bool ok = (nwin >= a.k) && ((double)mmax + (double)eps_q < theta);
    a.cert[qo] = ok ? 1 : 0;
    if (a.col_thr) a.col_thr[qo] = ok ? __builtin_inff() : __double2float_rd(theta - (double)(a.eps32 * qnorm));
    if (a.col_slot) {
      a.col_slot[qo] = ok ? -1 : qo;
      a.col_cnt[qo] = 0;
    }
    if (a.host_flag) {
      a.host_cert[qo] = ok ? 1 : 0;
      if (qo == 0 && a.host_err) *a.host_err = a.err_in ? *a.err_in : 0;
      __hip_atomic_store(a.host_flag + qo, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

hipError_t launch_rescore(const RescoreArgs &a, hipStream_t stream) {
  if (a.NC > RS_MAXNC) return hipErrorInvalidValue;
  if (a.NC <= 64) {
    hipLaunchKernelGGL(rescore_small_kernel, dim3((a.Q + 3) / 4), dim3(256), 0, stream, a);
#ifdef SSE_SCORE_CLOCK
    if (a.Q >= 8192) {
      static int n = 0;
      if (n++ % 8 == 3) {
        long long v[8];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_score_clk), sizeof v);
        fprintf(stderr, "[rescore_small clock Q=%d NC=%d] cycles: query norm %lld | candidates, rank, k-th, window %lld | float64 dots %lld (window of %lld) | exact rank + outputs %lld\n",
                a.Q, a.NC, v[0], v[1], v[2], v[4], v[3]);
      }
    }
#endif
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
  unsigned long long *served;  // diagnostic counter (device) or nullptr: +1 per query computed here (first page)
};

__global__ __launch_bounds__(EX_THREADS) void exact_topk_kernel(ExactArgs a) {
  __shared__ double s_sc[EX_THREADS / 64][SC_KC];
  __shared__ int64_t s_id[EX_THREADS / 64][SC_KC];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (a.cert && a.cert[q]) return;
  if (a.served && tid == 0 && a.k_off == 0) atomicAdd(a.served, 1ull);
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
                             int k, hipStream_t stream, unsigned long long *served) {
  // k <= 16: one pass (the certified-failure path).  Larger k: pages of 16, each pass a full
  // float64 sweep restricted to rows ranked after the previous page (exact, slow, rarely used:
  // the reference's consumers read <= 10 columns, sse_evaluator.py:95,112)
  for (int off = 0; off < k; off += SC_KC) {
    ExactArgs a{q, idxp, idx64, cert, out_scores, out_ids, id_base, N, Q, S, (k - off < SC_KC) ? k - off : SC_KC, off, k, served};
    hipLaunchKernelGGL(exact_topk_kernel, dim3(Q), dim3(EX_THREADS), 0, stream, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Collect path (exact results where the lists cannot certify them: ties / duplicates at the k-th score, k > 16).

__global__ void assign_slots_kernel(const int32_t *cert, int Q, int slots, int32_t *col_slot, int32_t *counter) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  int slot = -1;
  if (cert[q] == 0) {
    slot = atomicAdd(counter, 1);
    if (slot >= slots) slot = -1;  // pool exhausted: the float64 brute force serves this query
  }
  col_slot[q] = slot;
}
hipError_t launch_assign_slots(const int32_t *cert, int Q, int slots, int32_t *col_slot, int32_t *counter, hipStream_t st) {
  hipLaunchKernelGGL(assign_slots_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, cert, Q, slots, col_slot, counter);
  return hipGetLastError();
}

// bitonic sort, descending, of n2 (power of two) 64-bit keys in LDS by one workgroup
__device__ __forceinline__ void lds_sort_desc_u64(unsigned long long *key, int n2, int tid, int nthr) {
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = tid; i < (n2 >> 1); i += nthr) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = key[lo], y = key[hi];
        if ((x < y) == desc) {
          key[lo] = y;
          key[hi] = x;
        }
      }
    }
  __syncthreads();
}

// k > 16: threshold from the k-th best fp32 candidate.  One workgroup per query; NC <= RS_MAXNC candidates.
__global__ __launch_bounds__(256) void kth_bound_kernel(const float *q, const float *part_scores, const int32_t *part_ids, int Q,
                                                        int S, int NC, int k, float eps, float *col_thr, int32_t *col_slot) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long kb_key[];  // [n2]
  __shared__ double s_qn[4];
  const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int n2 = 1;
  while (n2 < NC) n2 <<= 1;
  double v = 0.0;
  for (int d = tid; d < S; d += 256) v += (double)q[(size_t)qi * S + d] * q[(size_t)qi * S + d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) s_qn[w] = v;
  for (int c = tid; c < n2; c += 256) {
    unsigned long long key = 0ull;  // empty slots sort last
    if (c < NC && part_ids[(size_t)qi * NC + c] >= 0) {
      unsigned u = __float_as_uint(part_scores[(size_t)qi * NC + c]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      key = ((unsigned long long)u << 32) | 1ull;
    }
    kb_key[c] = key;
  }
  lds_sort_desc_u64(kb_key, n2, tid, 256);
  if (tid == 0) {
    float thr = -__builtin_inff();
    if (k <= n2 && kb_key[k - 1] != 0ull) {
      unsigned u = (unsigned)(kb_key[k - 1] >> 32);
      u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
      const float eps_q = eps * (float)sqrt(s_qn[0] + s_qn[1] + s_qn[2] + s_qn[3]);
      // k rows score (fp32) >= kth, so the exact k-th best is >= kth - eps and every exact top-k row has fp32 score
      // >= kth - 2 eps
      thr = __double2float_rd((double)__uint_as_float(u) - 2.0 * (double)eps_q);
    }
    col_thr[qi] = thr;
    col_slot[qi] = qi;
  }
}
hipError_t launch_kth_bound(const float *q, const float *part_scores, const int32_t *part_ids, int Q, int S, int NC, int k,
                            float eps, float *col_thr, int32_t *col_slot, hipStream_t st) {
  if (NC > RS_MAXNC) return hipErrorInvalidValue;
  int n2 = 1;
  while (n2 < NC) n2 <<= 1;
  hipLaunchKernelGGL(kth_bound_kernel, dim3(Q), dim3(256), (size_t)n2 * sizeof(unsigned long long), st, q, part_scores, part_ids,
                     Q, S, NC, k, eps, col_thr, col_slot);
  return hipGetLastError();
}

// Two-pass path for mid-size indexes (10^4 .. 10^5 rows under thousands of queries -- the reference's real evaluation sizes,
// sse_evaluator.py:104-112 on rawdata-crosslingual: 16,491 x 32,060).  The list sweep is built for long streams: a lane list
// settles after ~30 tiles of its own, and with 64 independent thresholds per wave a register holds a hit in SOME lane in most
// tiles until the workgroup has seen ~16 k rows -- at 8 k rows per split the insertion passes cost 5x the MFMAs (0.83 ms for
// 0.16 ms of matrix work, profiles/r06_notes.txt).  Instead:
//   pass 1  the same sweep with no lists at all (score_topk_kernel<.., COLLECT> in max-only mode): every lane keeps the largest
//           bf16 score it saw; the NSPLIT x 16 lane maxima of a query belong to DISJOINT row sets, so their 16th largest is
//           reached by 16 distinct rows: theta_q <= the 16th best bf16 score of the whole index (in practice the ~18th best);
//   here    theta_q - 2 E, E = (bf16 + fp32 accumulation bound) |q| max|t|: 16 rows have an exact score >= theta - E, so has the
//           exact 16th best, so every exact top-k row (k <= 16), whose bf16 score is therefore >= theta - 2 E;
//   pass 2  the collect sweep on bf16 scores gathers every such row (~20 - 30 per query), select_topk_kernel re-scores them in
//           float64 and sorts (score descending, lower row first): provably the exact top-k; a buffer overflow (more than
//           col_cap rows within 2 E of the 16th best: crowded scores) leaves the query to the float64 brute force.
// One wave per query, NV = NSPLIT * 16 <= 256 maxima.
__global__ __launch_bounds__(256) void lane_max_threshold_kernel(const float *q, const float *lane_max, int Q, int S, int NV, float eps,
                                                                  float *col_thr, int32_t *col_slot) {
  const int lane = threadIdx.x & 63, qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= Q) return;
  double qn = 0.0;
  for (int d = lane; d < S; d += 64) qn += (double)q[(size_t)qi * S + d] * q[(size_t)qi * S + d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (j * 64 + lane < NV) ? lane_max[(size_t)qi * NV + j * 64 + lane] : -__builtin_inff();
  float kth = -__builtin_inff();
  for (int round = 0; round < 16; ++round) {  // the 16th largest: 16 times "take the maximum out"
    float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    kth = m;
    const unsigned long long has = __ballot(v[0] == m || v[1] == m || v[2] == m || v[3] == m);
    if (has == 0ull) break;  // (NaN scores: give up, threshold -inf below)
    if (lane == __ffsll((long long)has) - 1) {
      if (v[0] == m) v[0] = -__builtin_inff();
      else if (v[1] == m) v[1] = -__builtin_inff();
      else if (v[2] == m) v[2] = -__builtin_inff();
      else v[3] = -__builtin_inff();
    }
  }
  if (lane == 0) {
    float thr = -__builtin_inff();
    if (kth > -__builtin_inff() && kth == kth) thr = __double2float_rd((double)kth - 2.0 * (double)eps * sqrt(qn));
    col_thr[qi] = thr;
    col_slot[qi] = qi;
  }
}
hipError_t launch_lane_max_threshold(const float *q, const float *lane_max, int Q, int S, int NV, float eps, float *col_thr,
                                     int32_t *col_slot, hipStream_t st) {
  if (NV < 16 || NV > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lane_max_threshold_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, q, lane_max, Q, S, NV, eps, col_thr, col_slot);
  return hipGetLastError();
}

// One workgroup per collected query: float64 scores of its rows (the reference's arithmetic, wave_exact_dot as in the
// re-scoring pass: bit-identical values), sorted (score descending, then lower row id), first k out.
__global__ __launch_bounds__(256) void select_topk_kernel(SelectArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sel_smem[];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int slot = a.col_slot[q];
  if (slot < 0) return;
  const int n = a.col_cnt[slot];
  if (n > a.col_cap || n < a.k) return;  // overflow (or an inconsistent threshold): left to the float64 brute force
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  // sort key: float64 score mapped to an order-preserving u64 is not enough (ties need the row id): sort indices by
  // (score, id) with a 64-bit score key in one array and the row in a second one that moves along
  unsigned long long *skey = sel_smem;                       // [n2] order-preserving score bits
  int *srow = reinterpret_cast<int *>(sel_smem + n2);        // [n2]
  const int32_t *rows = a.col_buf + (size_t)slot * a.col_cap;
  const float *qrow = a.q + (size_t)q * a.S;
  const int KG = (a.S + 7) / 8;
  for (int i = w; i < n2; i += 4) {
    if (i < n) {
      const int row = rows[i];
      const double ex = wave_exact_dot(qrow, a.idx32, a.idx64, row, a.S, KG, lane);
      if (lane == 0) {
        unsigned long long u = (unsigned long long)__double_as_longlong(ex);
        u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
        skey[i] = u;
        srow[i] = row;
      }
    } else if (lane == 0) {
      skey[i] = 0ull;  // below every real score (-inf maps to 0x000f...: real keys are > 0)
      srow[i] = 0x7FFFFFFF;
    }
  }
  // bitonic sort on (key descending, row ascending)
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = tid; i < (n2 >> 1); i += 256) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = skey[lo], y = skey[hi];
        const int rx = srow[lo], ry = srow[hi];
        const bool x_after_y = (x < y) || (x == y && rx > ry);  // x ranks after y
        if (x_after_y == desc) {
          skey[lo] = y;
          skey[hi] = x;
          srow[lo] = ry;
          srow[hi] = rx;
        }
      }
    }
  __syncthreads();
  for (int j = tid; j < a.k; j += 256) {
    unsigned long long u = skey[j];
    u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    a.out_scores[(size_t)q * a.k + j] = __longlong_as_double((long long)u);
    a.out_ids[(size_t)q * a.k + j] = a.id_base + srow[j];
  }
  if (tid == 0) {
    a.cert[q] = 1;
    if (a.served) atomicAdd(a.served, 1ull);
  }
}
hipError_t launch_select_topk(const SelectArgs &a, hipStream_t st) {
  if (a.col_cap > SSE_COLLECT_CAP) return hipErrorInvalidValue;
  const size_t lds = (size_t)a.col_cap * (sizeof(unsigned long long) + sizeof(int));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(select_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(select_topk_kernel, dim3(a.Q), dim3(256), lds, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k-way merge of P sorted lists per query: in [P][Q][k] -> out [Q][k].
__global__ void merge_topk_kernel(const double *in_s, const int64_t *in_i, int64_t stride, int P, int Q, int k, double *out_s,
                                  int64_t *out_i) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  // rank of element (p,j) in the merged order = #elements before it; lists are
  // sorted, so count with a scan over the other lists (P*k is small)
  for (int p = 0; p < P; ++p)
    for (int j = 0; j < k; ++j) {
      const double s = in_s[(size_t)p * stride + (size_t)q * k + j];
      const int64_t id = in_i[(size_t)p * stride + (size_t)q * k + j];
      int rank = j;
      for (int p2 = 0; p2 < P && rank < k; ++p2) {
        if (p2 == p) continue;
        for (int j2 = 0; j2 < k; ++j2) {
          const double s2 = in_s[(size_t)p2 * stride + (size_t)q * k + j2];
          const int64_t id2 = in_i[(size_t)p2 * stride + (size_t)q * k + j2];
          if (before(s2, id2, s, id)) ++rank; else break;
        }
      }
      if (rank < k) {
        out_s[(size_t)q * k + rank] = s;
        out_i[(size_t)q * k + rank] = id;
      }
    }
}

hipError_t launch_merge_topk(const double *in_s, const int64_t *in_i, int64_t stride, int P, int Q, int k, double *out_s,
                             int64_t *out_i, hipStream_t stream) {
  hipLaunchKernelGGL(merge_topk_kernel, dim3((Q + 127) / 128), dim3(128), 0, stream, in_s, in_i, stride, P, Q, k, out_s,
                     out_i);
  return hipGetLastError();
}
