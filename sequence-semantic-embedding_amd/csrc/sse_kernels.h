// Internal declarations shared by the gfx950 kernels and the C-ABI layer.
// Not part of the public interface (that is include/sse_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// MFMA fragment packing ("frag32" layout), used for every operand of
// v_mfma_f32_32x32x2_f32 in this library.
//
// A matrix X[R][K] (R = the MFMA's 32-wide M or N dimension, K = reduction) is
// stored as blocks of 32 rows x 8 k = 256 floats = 1 KiB:
//     Xp[(rt * KG + kg) * 256 + lane * 4 + e],   rt = r / 32, kg = k / 8,
//     lane = ((k % 8) / 4) * 32 + (r % 32),      e = k % 4.
// Lane l of a wave reads ONE float4 at block + 4*l (a fully coalesced 1 KiB
// wave access, or a conflict-free ds_read_b128) and uses component e as the
// MFMA operand of sub-step e: A[i = l&31][kk = l>>5] (or B[kk][j = l&31]) =
// X[rt*32 + (l&31)][kg*8 + (l>>5)*4 + e].  Both operands permute k the same
// way, so four MFMAs cover k = kg*8 .. kg*8+7 exactly.
// ---------------------------------------------------------------------------
__host__ __device__ static inline int frag32_off(int r_in_tile, int k_in_group) {
  return ((((k_in_group >> 2) << 5) + r_in_tile) << 2) + (k_in_group & 3);
}

// C/D layout of the 32x32 MFMA: acc[reg] holds row mfma_row(reg, lane), column lane & 31.
__device__ static inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------
// Split bf16 operands: an fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to 2^-18 relative) and a
// product a*b as the three bf16 MFMAs a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (lstm_fwd_x3.hip, the
// training dK GEMM).  "frag16" blocks: 32 rows x 16 k bf16 = 1 KiB, lane (row & 31, k octet = lane >> 5) owns 8
// consecutive k = 16 bytes; a hi block is followed by its lo block.
// ---------------------------------------------------------------------------
typedef unsigned int sse_u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ static inline unsigned short sse_bf16_rne(float f) {  // round to nearest even (finite inputs)
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ static inline float sse_bf16_f32(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
// two fp32 -> packed hi pair and packed lo pair (element 0 in the low half-word) on the hardware converter
// (v_cvt_pk_bf16_f32: round to nearest even, the same values as sse_bf16_rne for finite inputs): 5 instructions per pair
// where the integer restatement takes ~24
__device__ static inline void sse_split2(float a, float b, unsigned &hi, unsigned &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __bf16 sse_bf2 __attribute__((ext_vector_type(2)));
  typedef float sse_f2 __attribute__((ext_vector_type(2)));
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(sse_f2{a, b}, sse_bf2));
  const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
  hi = h;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(sse_f2{ra, rb}, sse_bf2));
#else  // (host pass of the single-source compile: never called)
  const unsigned short h0 = sse_bf16_rne(a), h1 = sse_bf16_rne(b);
  hi = (unsigned)h0 | ((unsigned)h1 << 16);
  lo = (unsigned)sse_bf16_rne(a - sse_bf16_f32(h0)) | ((unsigned)sse_bf16_rne(b - sse_bf16_f32(h1)) << 16);
#endif
}
// 8 fp32 -> the hi and the lo octet (16 bytes each)
__device__ static inline void sse_split8(const float (&v)[8], sse_u32x4 &hi, sse_u32x4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    sse_split2(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

// ---------------------------------------------------------------------------
// 4 x 4 transpose inside every quad of lanes (DPP quad_perm, no LDS): on entry lane i of a quad holds v[0..3], on exit
// v[j] of lane i is the old v[i] of lane j.  Used where an accumulator lane owns (one row, 4 consecutive k) -- the piece the
// LDS operand tiles want -- and the tapes want (one k, 4 consecutive rows): the rows of a quad of lanes are consecutive.
// Two butterfly stages (lane bit 0 <-> element bit 0, lane bit 1 <-> element bit 1), 16 VALU instructions.
// ---------------------------------------------------------------------------
__device__ static inline void sse_quad_transpose(f32x4 &v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  auto x1 = [](float x) -> float {  // quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
  };
  auto x2 = [](float x) -> float {  // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
  };
  {  // stage 1: elements (0,1) and (2,3) against lane bit 0
    const float r01 = x1(b0 ? v[0] : v[1]), r23 = x1(b0 ? v[2] : v[3]);
    const float n0 = b0 ? r01 : v[0], n1 = b0 ? v[1] : r01, n2 = b0 ? r23 : v[2], n3 = b0 ? v[3] : r23;
    v = f32x4{n0, n1, n2, n3};
  }
  {  // stage 2: elements (0,2) and (1,3) against lane bit 1
    const float r02 = x2(b1 ? v[0] : v[2]), r13 = x2(b1 ? v[1] : v[3]);
    const float n0 = b1 ? r02 : v[0], n2 = b1 ? v[2] : r02, n1 = b1 ? r13 : v[1], n3 = b1 ? v[3] : r13;
    v = f32x4{n0, n1, n2, n3};
  }
#else
  (void)v;
  (void)lane;
#endif
}

// ------------------------------ LSTM forward -------------------------------
struct LstmFwdArgs {
  const int32_t *ids;   // [B][T]
  const float *emb;     // [V][Ep] word_embedding, zero padded, column E = 1.0 (carries the bias through the GEMM)
  const float *Wp;      // packed kernel, see pack_lstm_kernel(): k-row E = bias, forget bias folded into the f block
  const float *Mp;      // packed projection [Sp/32][KGh][256]
  float *out;           // [B][S]
  int32_t *err;         // bit 0: token id out of range
  int32_t B, T, V, Ep, KGx, KGh, S, NTS, normalize;
  int32_t H = 0;        // real cell size (0: Hp); unit blocks made only of padding are not computed in inference
  int32_t KGhe = 0;     // k-groups of h that can be non-zero = ceil(H/8) (0: all KGh); units >= H stay exactly 0
  int32_t NT32 = 0;     // training: number of 32-row tiles the tapes are laid out for (0: from the grid)
  int32_t xdouble = 1;  // set by launch_lstm_fwd from lstm_fwd_x_double()
  int32_t tiles_elsewhere = 0;  // 32-row tiles of launches that run CONCURRENTLY on other streams (the other encoder of a
                                // train step): the 32- vs 64-row tile choice looks at the chip, not at this launch alone
  int32_t force_rows = 0;       // 32 / 64: override the tile choice (Hp = 256; measurement aid, option lstm_train_rows)
  int32_t cu_count = 0;         // compute units of the device (0: 256, MI355X): the tile policy counts rounds of them
  int32_t gate_split = 1;       // Hp = 128 inference at 64-row tiles: lstm_fwd_gs.hip (two phase-shifted row groups, one gate per
                                // wave) instead of lstm_fwd_kernel<2,1,1>; bit-identical results (option "lstm_gate_split")
  // Left-pad prefix skip (exact): the state after p leading PAD (id 0) steps does not depend on
  // the sequence, so a tile starts at t0 = min over its rows of the leading-PAD count with
  // (h, c) = pad_h/pad_c[t0].  pad_* [T+1][Hp] come from rec_* of an all-PAD launch by the same
  // kernel (bit-identical arithmetic).  nullptr disables either side.
  const float *pad_h = nullptr, *pad_c = nullptr;
  const int32_t *row_map = nullptr;  // optional: logical row b reads ids / writes out at row row_map[b]
  float *rec_h = nullptr, *rec_c = nullptr;
  // training only (nullptr for inference): tapes consumed by the backward kernels
  float *tape_g = nullptr;  // [T][NT32][4][UB][5][4][64][4] gate activations + c, accumulator layout: lane l, register r at word
                            // (r >> 2) * 256 + 4 l + (r & 3) of its 1024-word block (16-byte pieces per lane)
  float *tape_a = nullptr;  // [(T*NT32*4)][KT][256]  [x_t | h_{t-1}] as frag32(rows = k', red = r)
  int32_t tape_a_split = 0; // 1: tape_a holds split bf16 frag16 blocks instead: [(T*NT32*2)][KT][hi|lo][512] (same bytes)
  int32_t tape_swap = 0;    // 1 (fp32 tapes only): the training forward runs in the inference orientation (weights = MFMA A operand,
                            // accumulator lane = sequence): tape_g is in THAT accumulator layout (lstm_bwd2_kernel reads it back
                            // lane-privately) and tape_a is written from registers (quad transposes), not re-read from LDS
  float *h_last = nullptr;  // [Bp][Hp] h_T
};
// Hp = 128 * UB hidden units; 512 threads; dynamic LDS = lstm_fwd_lds_bytes()
size_t lstm_fwd_lds_bytes(int KGx, int KGh, int RT);
bool lstm_fwd_x_double(int KGx, int KGh, int RT);
int lstm_fwd_rows_per_wg(int Hp, int B, int tiles_elsewhere = 0, int cus = 0 /* 0: 256 */);
hipError_t launch_lstm_fwd(const LstmFwdArgs &a, int Hp, hipStream_t stream);
// small cells (H <= 128), inference, 64-row tiles: lstm_fwd_gs.hip
size_t lstm_fwd_gs_lds_bytes(int KGx, int NB);
bool lstm_fwd_gs_ok(int KGx, int KGh, int H);
hipError_t launch_lstm_fwd_gs(const LstmFwdArgs &a, hipStream_t stream);

// few-sequences LSTM forward (lstm_small.hip): one workgroup per 4 sequences, vector-ALU GEMV on the master variables
struct LstmSmallArgs {
  const int32_t *ids;    // [B][T]
  const float *emb;      // word_embedding [V][E] (master variable)
  const float *Waug;     // [KA][4H] kernel rows in the kernel's k space incl. the bias row (launch_pack_lstm_small)
  const float *M;        // projection [H][S] (master variable)
  float *out;            // [B][S]
  int32_t *err;
  int32_t B, T, V, E, H, S, normalize;
  const float *pad_h = nullptr, *pad_c = nullptr;  // [T+1][pad_stride] state after p leading PAD steps (this kernel's own)
  float *rec_h = nullptr, *rec_c = nullptr;        // table build: states of sequence 0 after every step
  int32_t pad_stride = 0;
};
size_t lstm_small_lds_bytes(int E, int H, int S);
size_t lstm_small_waug_floats(int E, int H);
hipError_t launch_pack_lstm_small(const float *K, const float *b, int E, int H, float *out, hipStream_t stream);
hipError_t launch_lstm_small(const LstmSmallArgs &a, hipStream_t stream);

// opt-in LSTM forward on the bf16 matrix pipe with split (hi + lo) operands, Hp = 128 / 256 (lstm_fwd_x3.hip)
struct LstmX3Args {
  const int32_t *ids;            // [B][T]
  const unsigned short *emb16;   // [V][hi|lo][KGX*16] bf16 split embedding table, column E = 1.0
  const unsigned short *Wx3;     // [Hp/32][KGX+Hp/16][4 gates][hi|lo][512] bf16 fragments (launch_pack_lstm_x3)
  const float *Mp;               // packed projection (fp32, as lstm_fwd.hip)
  float *out;                    // [B][S]
  int32_t *err;
  int32_t B, T, V, KGX, H, S, NTS, normalize;
  const int32_t *row_map = nullptr;
  const float *pad_h = nullptr, *pad_c = nullptr;  // [T+1][Hp] state after p leading PAD steps (recorded by this kernel)
  float *rec_h = nullptr, *rec_c = nullptr;        // table build: states of sequence 0 after every step
  // training forward (tape_g != nullptr): tapes as lstm_fwd.hip's, tape_a in the split frag16 form (tape_a_split = 1)
  float *tape_g = nullptr, *tape_a = nullptr, *h_last = nullptr;
  int32_t NT32 = 0;
  int32_t tiles_elsewhere = 0;  // 32-row tiles of the other encoder's forward running concurrently (tile-size choice)
};
int lstm_x3_kgx(int E);
size_t lstm_x3_weight_elems(int E, int Hp);
size_t lstm_x3_emb_elems(int64_t V, int E);
hipError_t launch_pack_lstm_x3(const float *K, const float *b, const float *emb /* master [V][E] */, int64_t V, int E, int H,
                               int Hp, unsigned short *Wx3, unsigned short *emb16 /* nullptr: weights only */,
                               int natural_k /* 1: h part in unit order (training forward) */, hipStream_t stream);
hipError_t launch_lstm_fwd_x3(const LstmX3Args &a, hipStream_t stream);

// a handful of sequences with the recurrent weights resident in LDS (lstm_persist.hip): a cluster of NWG workgroups per
// 4 sequences, h_t exchanged through global memory every step
struct LstmPersistArgs {
  const int32_t *ids;    // [B][T]
  const float *emb;      // word_embedding [V][E] (master variable)
  const float *Waug;     // [KA][4H] as for lstm_small.hip
  const float *M;        // projection [H][S] (master variable)
  float *out;            // [B][S]
  int32_t *err;          // bit 0: token id out of range; bit 2: a cluster workgroup never arrived (bounded spin)
  int32_t B, T, V, E, H, S, normalize;
  const float *pad_h = nullptr, *pad_c = nullptr;  // the pad-prefix table of lstm_small.hip (same arithmetic)
  int32_t pad_stride = 0;
  unsigned long long *hx = nullptr;    // [8][2][4][H] h_t exchange, {value, tag} words (lstm_persist_hx_words)
  unsigned long long *rawx = nullptr;  // [8][4][S] raw encodings exchange (lstm_persist_raw_words)
  uint32_t epoch = 0;           // 1 .. 2^20-1, different for every launch on the same exchange buffers
  int32_t NWG = 0;              // set by the launcher
  int32_t map_mode = 0;         // 0: cluster = blockIdx % 8 (one XCD per cluster); 1: consecutive blocks (measurement aid)
  int32_t write_through = 0;    // 1: always publish h with write-through stores (the any-placement path)
  int32_t plain_launch = 0;     // 1: hipLaunchKernelGGL instead of the cooperative launch (option lstm_cluster_coop = 0)
};
int lstm_persist_nwg(int E, int H, int S);   // workgroups per cluster, 0: shape not supported
int lstm_persist_max_rows();
int lstm_persist_max_steps();
size_t lstm_persist_hx_words(int H);
size_t lstm_persist_raw_words(int S);
hipError_t launch_lstm_persist(const LstmPersistArgs &a, hipStream_t stream);
// cooperative launches the runtime REFUSED (the launchers then took the plain launch, whose co-residency nobody promises):
// process-wide count, surfaced as counter "lstm_coop_refused"
void lstm_note_coop_refused();
long long lstm_coop_refused();

// mid-size batches (33 .. 1024 sequences) with the hidden units of a 64-row tile spread over a cluster of 16 workgroups that
// keep their weight fragments in LDS and exchange h_t every step -- lstm_persist.hip's layout on MFMA (lstm_cluster.hip)
struct LstmClusterArgs {
  const int32_t *ids;    // [B][T]
  const float *emb;      // padded embedding table [V][Ep], column E = 1.0 (the matrix kernel's)
  const float *Wc;       // launch_pack_lstm_cluster
  const float *Mp;       // packed projection [NTS][KGh][256] (the matrix kernel's)
  float *out;            // [B][S]
  int32_t *err;          // bit 0: token id out of range; bit 2: a cluster workgroup never arrived (bounded spin)
  int32_t B, T, V, E, Ep, KGx, KGh, H, Hp, S, NTS, normalize;
  const float *pad_h = nullptr, *pad_c = nullptr;  // the pad-prefix table of lstm_small.hip (same arithmetic)
  int32_t pad_stride = 0;
  unsigned long long *hx = nullptr;  // lstm_cluster_hx_words(H): h_t exchange, {value, tag} words
  unsigned long long *sx = nullptr;  // lstm_cluster_sx_words(): per-tile row sums of squares
  uint32_t epoch = 0;                // 1 .. 2^20-1, different for every launch on the same exchange buffers
  int32_t NCL = 0;                   // set by the launcher
  int32_t write_through = 0;         // 1: always publish h with write-through stores (the any-placement path; tests)
  int32_t plain_launch = 0;          // 1: hipLaunchKernelGGL instead of the cooperative launch (option lstm_cluster_coop = 0)
  int32_t drop_wg = 0;               // 1: the last workgroup of cluster 0 exits at once (tests of the give-up path)
};
int lstm_cluster_ok(int E, int H, int S);
int lstm_cluster_max_rows();
size_t lstm_cluster_weight_floats(int E, int H);
size_t lstm_cluster_hx_words(int H);
size_t lstm_cluster_sx_words();
hipError_t launch_pack_lstm_cluster(const float *K, const float *b, int E, int H, float *Wc, hipStream_t stream);
hipError_t launch_lstm_cluster(const LstmClusterArgs &a, hipStream_t stream);

// ------------------------------ scoring ------------------------------------
struct ScoreArgs {
  const float *idxp;     // packed index  [NT][KG][256]  (frag32, rows = targets)
  const float *qp;       // packed queries [QT][KG][256] (frag32, rows = queries)
  float *part_scores;    // [Q][NSPLIT][KC] per-split candidate scores (f32), sorted descending
  int32_t *part_ids;     // [Q][NSPLIT][KC] row numbers local to this index (or -1)
  float *part_bnd;       // [Q][NSPLIT] every row of the split outside its KC candidates scores <= this
  int64_t N;             // index rows
  int32_t Q, KG, NT, QT, NSPLIT, KC;
  int32_t NQ = 4;     // query tiles (of 32) per workgroup: 4 (128-query blocks), 2 (index dimension > 296) or 1 (Q <= 32, dimension > 616)
  int32_t thr_off = 0;  // set by the launcher: LDS offset (floats) of the shared per-query thresholds
  const int32_t *q_count = nullptr;  // compacted second-chance pass: the real number of queries (<= Q) lives on the device;
                                     // workgroups and lanes past it do nothing
  int32_t BF = 0;       // 1: idxp / qp are bf16 fragment copies, KG counts 16-k groups (candidate pass on bf16 MFMA)
  // collect pass (COLLECT = 1, fp32 only): rows with score >= col_thr[query] are appended to buffer col_slot[query]
  int32_t COLLECT = 0;
  const float *col_thr = nullptr;     // [Q]
  const int32_t *col_slot = nullptr;  // [Q] buffer slot, < 0: query not collected
  int32_t *col_cnt = nullptr;         // [slots] rows appended (may exceed col_cap: overflow)
  int32_t *col_buf = nullptr;         // [slots][col_cap] local row numbers
  int32_t col_cap = 0;
  float *lane_max = nullptr;          // COLLECT variant, max-only mode: [Q][NSPLIT][16] largest score per (query, split, lane list); no
                                      // lists, no buffers (first pass of the two-pass path for mid-size indexes)
  int32_t dbg = 0;  // builds with -DSSE_SCORE_MEASURE only (env SSE_SCORE_DBG): bit 0 = skip the top-k epilogue of the sweep
  // NQ == 1 (<= 32 queries, the latency path) and every COLLECT sweep: the workgroups build their query fragments from the row-major fp32 queries
  // themselves (q_rows [Q][S]; fp32 or rounded to bf16 exactly as launch_pack_rows / launch_pack_rows_bf16 would) -- qp is
  // not read and the pack launch in front of the sweep goes away
  const float *q_rows = nullptr;
  int32_t S = 0;
};
hipError_t launch_score_topk(const ScoreArgs &a, hipStream_t stream);
// two-pass path for mid-size indexes: threshold per query from the lane maxima of a max-only sweep (ScoreArgs::lane_max)
hipError_t launch_lane_max_threshold(const float *q, const float *lane_max, int Q, int S, int NV, float eps, float *col_thr,
                                     int32_t *col_slot, hipStream_t st);
// query tiles per workgroup (4 / 2 / 1) whose LDS query block fits for index dimension S in every variant of a call
// (0: none); dynamic LDS of one variant
#define SSE_MAX_INDEX_DIM 1024
int score_pick_nq(int Q, int S, int with_bf16);
size_t score_lds_bytes(int NQ, int KG, int BF, int COLLECT);
// rows [R][C] fp32 -> bf16 fragment blocks [ceil(R/32)][ceil(C/16)][1 KiB]; fp32 frag32 index -> the same
hipError_t launch_pack_rows_bf16(const float *rows, int64_t R, int C, void *out, hipStream_t stream);
hipError_t launch_count_uncert(const int32_t *cert, int Q, unsigned long long *count, hipStream_t st);
// uncertified queries (cert[q] == 0) -> dense set: qmap[slot] = q, *count = size, qc[slot][S] = their rows (zero rows up to
// a whole 32-query tile)
hipError_t launch_compact_uncert(const float *q, const int32_t *cert, int Q, int S, int32_t *qmap, int32_t *count, float *qc,
                                 hipStream_t st);
hipError_t launch_frag32_to_bf16(const float *idxp, int64_t NT, int KG, void *out, hipStream_t stream);

#define SSE_COLLECT_CAP 4096   // rows one query can collect (exact path); more -> float64 brute force
#define SSE_MAX_SELECT_K 1024  // largest k the collect path serves

struct RescoreArgs {
  const float *q;          // [Q][S] f32 row-major queries
  const float *idx32;      // [N][S] f32 rows (used when idx64 == nullptr)
  const double *idx64;     // [N][S] f64 rows or nullptr
  const float *part_scores;
  const int32_t *part_ids;
  const float *part_bnd;   // [Q][NSPLIT]
  double *out_scores;      // [Q][k]
  int64_t *out_ids;        // [Q][k]
  int32_t *cert;           // [Q] 1 = the f32 candidate set provably contains the exact top-k
  int64_t id_base, N;
  int32_t Q, S, NC, k;     // NC = NSPLIT*KC candidates per query
  float eps;               // bound on |candidate score - exact score| / |q|
  // compacted second-chance pass: q / part_* are indexed by slot, results (out_*, cert, col_thr) go to query qmap[slot];
  // slots >= *q_count do not exist
  const int32_t *qmap = nullptr, *q_count = nullptr;
  // optional: collect threshold of every processed query, +inf when certified, else (exact k-th of the
  // candidates) - eps32 * |q| rounded down: no exact top-k row has a smaller fp32 score
  float *col_thr = nullptr;
  float eps32 = 0.0f;      // fp32 bound used for col_thr (eps may be the wider bf16 one)
  // optional (calls of <= the collect pool's queries): every query owns collect slot `query`; the pass writes
  // col_slot[query] = certified ? -1 : query and zeroes col_cnt[query], which replaces launch_assign_slots + a memset
  int32_t *col_slot = nullptr, *col_cnt = nullptr;
  // optional host mirror (few queries, host-buffer entry points): scores / ids / certificates are ALSO stored through
  // these device-visible pinned host pointers, then -- after a system-scope fence -- host_flag[query] = seq.  The host
  // polls the flags instead of queueing three read-back copies and waiting for the stream.  host_err (query 0 only)
  // receives *err_in, the encoder's error flag of the same call.
  double *host_scores = nullptr;
  int64_t *host_ids = nullptr;
  int32_t *host_cert = nullptr, *host_flag = nullptr, *host_err = nullptr;
  const int32_t *err_in = nullptr;
  int32_t seq = 0;
  // optional row-major f32 copy of the index [N][S] (S % 4 == 0; kept for small indexes): the float64 re-scoring gathers a
  // row from 8 cache lines instead of the 64 the fragment order spreads it over.  A lane reads the same four dimensions
  // either way, so the sums -- and the float64 scores -- are bit-identical.
  const float *idx_rm = nullptr;
};
hipError_t launch_rescore(const RescoreArgs &a, hipStream_t stream);

// Small indexes under MANY queries (the evaluator's shape: 16384 x 571 x 256, the scoring half of the headline step): no
// lists, no splits -- a workgroup of 32 queries forms ALL N <= 1024 fp32 scores on the matrix pipe into an LDS tile and then
// selects each query's 16 best (score descending, lower row first) exactly, by a threshold search; part_* in the layout of
// launch_score_topk with ONE split (NC = 16), so launch_rescore follows unchanged.  part_bnd[q] = the 16th best (every
// other row scores <= that), +inf when more than 63 rows tie into the selection (the certificate then fails: next stage).
// Index dimensions 249 .. 256, 57 .. 64 and 49 .. 56 (the k-groups are a template parameter: the query fragments live in registers).
struct SmallIndexArgs {
  const float *q_rows;   // [Q][S] fp32 row-major
  const float *idxp;     // frag32 index [NT][KG][256]
  float *part_scores;    // [Q][16]
  int32_t *part_ids;     // [Q][16]
  float *part_bnd;       // [Q]
  int64_t N;
  int32_t Q, S, KG, NT;
};
bool score_small_index_applies(int Q, int KG, int64_t NT);
hipError_t launch_score_small_index(const SmallIndexArgs &a, hipStream_t stream);

// uncertified queries (cert[q] == 0) get collect-buffer slots 0, 1, ... (col_slot[q]; -1 when certified or the pool
// of `slots` is exhausted); *counter must be zero on entry
hipError_t launch_assign_slots(const int32_t *cert, int Q, int slots, int32_t *col_slot, int32_t *counter, hipStream_t st);
// k > 16: col_thr[q] = (k-th best fp32 candidate score) - 2 * eps * |q| rounded down (-inf with fewer than k
// candidates); col_slot[q] = q
hipError_t launch_kth_bound(const float *q, const float *part_scores, const int32_t *part_ids, int Q, int S, int NC, int k,
                            float eps, float *col_thr, int32_t *col_slot, hipStream_t st);
struct SelectArgs {
  const float *q;
  const float *idx32;
  const double *idx64;
  const int32_t *col_slot, *col_cnt, *col_buf;
  int32_t col_cap;
  double *out_scores;  // [Q][k]
  int64_t *out_ids;
  int32_t *cert;       // set to 1 for queries served here; left alone on overflow / fewer than k rows
  int64_t id_base;
  int32_t Q, S, k;
  unsigned long long *served = nullptr;  // diagnostic counter (device), +1 per query served here
};
// float64 re-score + sort of the collected rows, first k out (score descending, then lower row id)
hipError_t launch_select_topk(const SelectArgs &a, hipStream_t st);

// in_s / in_i: list (p, q, j) at p * stride + q * k + j (stride = Q*k for plain [P][Q][k] arrays)
hipError_t launch_merge_topk(const double *in_s, const int64_t *in_i, int64_t stride, int P, int Q, int k, double *out_s,
                             int64_t *out_i, hipStream_t stream);

// ------------------------------ packing / misc -----------------------------
// rows [R][C] f32 row-major -> frag32 [ceil(R/32)][ceil(C/8)][256], zero padded
hipError_t launch_pack_rows(const float *rows, int64_t R, int C, float *out, hipStream_t stream);
// pack + max over rows of sum(x^2) (float bits, atomicMax into *norm_bits which the caller zeroed) in one pass
hipError_t launch_pack_rows_norm(const float *rows, int64_t R, int C, float *out, float *norm_bits, hipStream_t stream);
hipError_t launch_f64_to_f32(const double *in, float *out, int64_t n, hipStream_t stream);
hipError_t launch_l2_normalize(const float *x, float *out, int64_t rows, int cols, hipStream_t stream);

hipError_t launch_pack_lstm(const float *K, const float *b, int E, int H, int Ep, int Hp, int UB, float *Wp,
                            hipStream_t stream);
// several re-layouts in one launch (the train step's: padded embedding table, packed kernels / projections, Kh^T / Kx^T)
enum { PACK_JOB_PAD_ROWS = 1, PACK_JOB_LSTM = 2, PACK_JOB_KN = 3, PACK_JOB_KT = 4 };
struct PackJob {
  int32_t type;
  const float *a, *b;
  float *out;
  int64_t total;  // PAD_ROWS: rows; the others: float4 items
  int32_t i0, i1, i2, i3;
};
#define SSE_MAX_PACK_JOBS 12
struct PackJobs {
  PackJob job[SSE_MAX_PACK_JOBS];
  int32_t n = 0;
};
void pack_job_pad_rows(PackJobs &js, const float *in, int64_t R, int C, int Cp, int one_col, float *out);
void pack_job_lstm(PackJobs &js, const float *K, const float *b, int E, int H, int Ep, int Hp, float *Wp);
void pack_job_kn(PackJobs &js, const float *X, int K, int N, int KGp, float *out);
void pack_job_kT(PackJobs &js, const float *K, int row0, int nrows, int RT, int H, int Hp, float *out);
hipError_t launch_pack_multi(const PackJobs &js, hipStream_t stream);
hipError_t launch_pack_kn(const float *X, int K, int N, int KGp, float *out, hipStream_t stream);
hipError_t launch_pad_rows(const float *in, int64_t R, int C, int Cp, int one_col, float *out, hipStream_t stream);
// PAD-prefix bucketing of device-resident ids (pack.hip): row numbers by leading-PAD count, longest prefix first
#define SSE_PAD_SORT_MAX_T 8192
size_t pad_sort_zeroed_words();
size_t pad_sort_work_words(int B, int T);
hipError_t launch_pad_sort(const int32_t *ids, int B, int T, int32_t *zeroed, int32_t *work, int32_t *order, int32_t *stat_pinned,
                           int32_t seq, bool scatter, hipStream_t stream);
hipError_t launch_row_norm2_max(const float *x, int64_t rows, int cols, float *out_bits, hipStream_t stream);
hipError_t launch_fill(float *p, int64_t n, float v, hipStream_t stream);
hipError_t launch_exact_topk(const float *q, const float *idxp, const double *idx64, const int32_t *cert,
                             double *out_scores, int64_t *out_ids, int64_t id_base, int64_t N, int Q, int S,
                             int k, hipStream_t stream, unsigned long long *served = nullptr);

// ------------------------------ CNN encoder --------------------------------
// bf16-storage variant (cnn_fwd_bf16.hip): embeddings / filters as bf16, fp32 accumulation, fp32 tail
size_t cnn_bf16_packed_weight_elems(int Ep);
size_t cnn_bf16_lds_bytes(int T, int Ep, int train);
hipError_t launch_cnn_bf16_pack(const float *emb, int64_t V, int E, int Ep, unsigned short *emb_bf16, const float *const W[4],
                                unsigned short *Wc, hipStream_t stream);
hipError_t launch_cnn_fwd_bf16(const int32_t *ids, const unsigned short *emb_bf16, const unsigned short *Wc, const float *bias,
                               float *featp, int32_t *err, int B, int T, int V, int Ep, float *feat_rm, int32_t *pos,
                               hipStream_t stream);
hipError_t launch_cnn_proj(const float *featp, const float *Mp, float *out, int B, int S, int normalize, hipStream_t stream);
// the same tail on split bf16 operands (option cnn_bf16; cnn_fwd_bf16.hip)
size_t cnn_proj_x3_elems(int S);
hipError_t launch_pack_cnn_proj_x3(const float *Mv, int S, unsigned short *Mx3, hipStream_t stream);
hipError_t launch_cnn_proj_x3(const float *featp, const unsigned short *Mx3, float *out, int B, int S, int normalize, hipStream_t stream);
size_t cnn_lds_bytes(int T, int Ep, int train);
size_t cnn_packed_weight_floats(int Ep);
hipError_t launch_pack_conv(const float *const W[4], int E, int Ep, float *out, hipStream_t stream);
hipError_t launch_cnn_fwd(const int32_t *ids, const float *emb, const float *Wc, const float *bias, const float *Mp,
                          float *featp, float *out, int32_t *err, int B, int T, int V, int Ep, int S, int normalize,
                          float *feat_rm /* training: [B][576], else NULL */, int32_t *pos /* training: [B][576] */,
                          hipStream_t stream);
