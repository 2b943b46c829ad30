// C-ABI layer of libsse_hip.so (see include/sse_hip.h): handle, variables,
// lazy weight re-layout, scratch management, kernel orchestration.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <link.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sse_hip.h"
#include "sse_kernels.h"
#include "train.h"

namespace {

thread_local std::string g_create_error;

struct Variable {
  std::string name;
  int rows = 0, cols = 0;  // logical 2-D view (count = rows*cols)
  int64_t count = 0;
  float *dev = nullptr;     // master copy, row-major, TF shape
  float *slot = nullptr;    // '<name>/Adagrad' accumulator
  float *grad = nullptr;    // gradient of the current step (allocated by the first train step)
};

struct Encoder {
  int kernel = -1, bias = -1, proj = -1;  // variable indices
  int H = 0, Hp = 0, UB = 0, KGx = 0, KGh = 0, Ep = 0;
  float *Wp = nullptr, *Mp = nullptr;
  float *Waug = nullptr;      // few-sequences kernel: kernel rows in its k space incl. the bias row
  bool waug_valid = false;
  float *Wc = nullptr;        // cluster kernel (lstm_cluster.hip): weight fragments per workgroup of a cluster
  bool wc_valid = false;
  unsigned short *Wx3 = nullptr;  // option lstm_x3: hi / lo bf16 fragment copies of the kernel matrix
  bool x3_valid = false;
  unsigned short *Wx3t = nullptr;  // option train_fwd_x3: the same copies with the h part in unit order (training forward)
  float *pad_h_x3 = nullptr, *pad_c_x3 = nullptr;  // pad-prefix table of the lstm_x3 path ([pad_T_x3+1][Hp])
  int pad_T_x3 = 0;
  bool pad_valid_x3 = false;
  int shares_lstm_with = -1;  // shared-encoder: target reuses the source LSTM packing
  bool generic = false;       // shape outside the fused kernels' layouts: every encode / train step of this encoder runs lstm_generic.hip
  // pad-prefix table: state after p leading PAD steps, p = 0..pad_T ([pad_T+1][Hp] each)
  float *pad_h = nullptr, *pad_c = nullptr;
  int pad_T = 0;
  bool pad_valid = false;
  // the same table for the few-sequences kernel (lstm_small.hip), produced by that kernel ([pad_T_small+1][H])
  float *pad_h_small = nullptr, *pad_c_small = nullptr;
  int pad_T_small = 0;
  bool pad_valid_small = false;
};

struct DevBuf {  // grow-only device scratch; freed with its owner (handle / TrainState)
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

// device buffers of the training path, grown on demand
struct TrainState {
  DevBuf ids[2], labels, raw[2], draw[2], tape_g[2], tape_a[2], h_last[2], dh_last[2], dg_a[2], dg_b[2], db_part[2],
      dk_part[2], dm_part[2], sq_part, norm_part, row_loss, row_acc, scal;
  DevBuf feat_rm, pos, dfeat, dw_part, dbias_part, wt, wct;  // text-CNN training
  uint64_t gen_ver[2] = {0, 0};  // weights_version the any-shape packs gen_KT / gen_Kq / gen_MT were built from
  DevBuf gen_A[2], gen_tape[2], gen_dG[2], gen_hl[2], gen_KT[2], gen_Kq[2], gen_MT[2], gen_G, gen_c, gen_dA, gen_dc, gen_dkp;  // any-shape LSTM path
  hipStream_t side[2] = {nullptr, nullptr};  // the two encoders run concurrently (forward and backward)
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  float *KhT[2] = {nullptr, nullptr}, *KxT[2] = {nullptr, nullptr};
  unsigned short *KhT16[2] = {nullptr, nullptr};  // option train_bwd_x3: Kh^T as split frag16 blocks
  unsigned short *KxT16[2] = {nullptr, nullptr};  // ... and Kx^T as 16x16x32 B fragments (dX inside the BPTT kernel)
  bool packed_dirty = true;
  bool fp32_dirty = true;  // the fp32 Kh^T / Kx^T fragment copies are stale (rebuilt only by steps that run fp32 BPTT kernels)
  // gradient arena: [grad of variable 0 | ... | grad of variable n-1 | tail[4]]; tail = {sum of squares of the
  // un-deduplicated embedding-gradient slices, loss, train_acc, rows}, every entry a plain sum over the
  // ranks of a data-parallel job (one flat all-reduce)
  float *arena = nullptr;
  bool arena_external = false, grads_ready = false;
  // corpus resident on the device (sse_corpus_upload): the *_rows train entry points ship row numbers, not token ids
  DevBuf corpus[2], rows[2];
  int64_t corpus_N[2] = {0, 0};
  int32_t corpus_T[2] = {0, 0};
  bool rows_mode = false;  // set by the *_rows entry points around train_grads_locked
  bool defer_err = false;  // set by the fused step entry points: the device error flag is read once, with the loss, after the
                           // whole step has been queued (a flagged step cancels its own update on the device)
  // paired batches (data.py:95-115: every source row appears twice, once with its positive and once with a negative
  // target): internal row order = [rows 0,2,4,.. | rows 1,3,5,..], the source encoder runs on the first half only
  DevBuf ids_raw[2], perm, hot_part[2];
  int perm_B = 0;
  std::vector<int32_t> h_perm, h_rows[2], h_tgt;
  std::vector<float> h_labels;
};

}  // namespace

struct sse_handle {
  sse_config cfg;
  std::mutex mu;
  std::string err;
  std::vector<Variable> vars;
  Encoder enc[2];
  bool packed_dirty = true;
  uint64_t weights_version = 1;  // bumped with every change of a variable: the any-shape path's packs (lstm_generic.hip) are cached against it
  bool mp_fresh = false;     // the packed projections match the variables although packed_dirty is set (ensure_proj_packed)
  bool pad_skip = true;      // option "pad_skip": exact left-pad prefix skip in inference encodes
  // option "pad_sort_dev": PAD-prefix bucketing of device-resident id matrices (sse_encode_dev) by two small kernels (pack.hip):
  // 0 off | 1 adaptive: every eligible call measures its batch; rows are bucketed while the latest completed call of that side
  // saw any padding, 32-row tiles while it saw a mean prefix >= T / 4 (no host round trip: an index build's batches look alike)
  // | 2 always bucket + 32-row tiles (deterministic: tests, tools)
  int pad_sort_dev = 1;
  // options "score_two_pass_min_rows" / "score_two_pass_rows": indexes of min .. max rows under >= 1024 queries (k <= 16, bf16
  // candidates) are ranked by a max-only sweep + a collect sweep instead of the list sweep (exact, same ids and scores); max 0 = off.
  // Window measured on well-spread unit vectors, S = 256 (tools/bench_c3.py, tools/bench_score.py, profiles/r06_notes.txt), two-pass /
  // list sweep in ms: 16,491 queries x 32 k rows 1.19 / 1.03, 64 k 1.58 / 1.72, 128 k 2.39 / 3.07, 256 k 4.13 / 5.51, 512 k 7.61 / 10.17;
  // 1024 x 300 k 0.44 / 0.57; but 8192 x 1.25 M 7.83 / 4.91 (a list-free sweep runs at 0.54 of the bf16 peak there, the list sweep at 0.43)
  int64_t score_two_pass_min_rows = 49152, score_two_pass_rows = 524288;
  int64_t two_pass_calls = 0;  // counter "score_two_pass_calls"
  DevBuf s_lmax;
  bool lstm_gate_split = true;  // option "lstm_gate_split": small cells (H <= 128) at 64-row tiles on lstm_fwd_gs.hip (bit-identical)
  DevBuf s_padzero, s_padwork, s_padorder;
  int32_t *pad_stat = nullptr;  // pinned host words the device stores to: [side] {call number, class 0 / 1 / 2}
  int32_t pad_seq = 0;
  int64_t pad_sorted_calls = 0;  // counter "pad_sorted_calls"
  bool lstm_x3 = false;       // option "lstm_x3": large inference encodes (Hp = 256) on the bf16 matrix pipe with split operands
  unsigned short *emb16 = nullptr;  // split embedding table of that path
  bool emb16_valid = false;
  int lstm_persist_rows = 32; // option "lstm_persist_rows": batches up to this many rows (<= 32) take the weights-in-LDS cluster kernel
  uint32_t persist_epoch = 0; // tag epoch of the cluster kernel's exchange buffers
  hipStream_t stream = nullptr;  // sse_set_stream: the stream the train-step entry points enqueue on (default: the null stream)
  bool persist_inject = false;   // testing aid (option lstm_persist_inject_miss): report a missing cluster workgroup after every cluster launch
  int64_t persist_fallbacks = 0; // host-buffer encodes re-run on lstm_small because a cluster workgroup did not arrive
  int cu_count = 0;           // compute units of the device (co-residency check of the cluster kernel)
  int lstm_cluster_rows = 1024; // option "lstm_cluster_rows": batches above lstm_persist_rows up to this many rows (<= 1024) take the MFMA cluster kernel
  uint32_t cluster_epoch = 0;   // tag epoch of that kernel's exchange buffers
  int lstm_cluster_chunks = 3;  // option "lstm_cluster_chunks": batches of up to this many times lstm_cluster_rows go through that kernel in launches of lstm_cluster_rows
  int cluster_backoff = -1;     // option "lstm_cluster_backoff": after a cluster-kernel launch gave up, this many following eligible calls go straight to the kernels that need no co-residency.  -1 (default) = automatic: no back-off is ARMED until a give-up has been observed (cooperative launches make co-residency the runtime's promise), but an observed give-up -- another process on the device, or a cooperative launch the runtime refused (counter lstm_coop_refused) -- arms 16 calls, so a busy device pays the 10 ms give-up once in 17 calls, not on every call
  int cluster_skip[2] = {0, 0}; // calls still to skip: [0] single-query kernel (lstm_persist), [1] mid-batch kernel (lstm_cluster)
  int lstm_cluster_coop = 1;    // option "lstm_cluster_coop" (default 1): the cluster kernels are launched with hipLaunchCooperativeKernel (co-residency guaranteed by the runtime; +20 us per launch measured); 0 = plain launches
  int lstm_cluster_wt = 0;      // option "lstm_cluster_write_through": force the any-placement publish path (tests)
  int lstm_cluster_drop = 0;    // option "lstm_cluster_drop_wg": one workgroup of the cluster kernel exits at once (tests)
  int lstm_small_rows = 1024; // option "lstm_small_rows": batches up to this many rows take the few-sequences LSTM kernel
  bool score_small_index = true;  // option "score_small_index": many queries against <= 1024 rows skip the list sweep (launch_score_small_index)
  bool score_bf16 = true;    // option "score_bf16" (default on): candidate pass on the bf16 matrix pipe; results stay exact
  void *idxp16 = nullptr;    // bf16 fragment copy of the index (built on demand)
  size_t idxp16_cap = 0;
  bool idxp16_valid = false;
  bool fb_cnt_init = false;
  bool cnn_bf16 = false;     // option "cnn_bf16": source_only_cnn inference with bf16 storage / fp32 accumulation
  unsigned short *emb_bf16 = nullptr, *cnn_Wc16 = nullptr, *cnn_Mx3 = nullptr;
  int lstm_train_rows = 0;   // option "lstm_train_rows": 0 = automatic, 32 / 64 = rows per workgroup of the training forward (Hp = 256)
  // The train step computes in fp32 like the reference (tf.float32 graph, sse_model.py:355-364): the three split-operand
  // options below are OPT-IN (a default narrower than the reference's arithmetic earns no credit).
  bool train_bwd_x3 = false;  // option "train_bwd_x3": recurrent GEMM of BPTT on the bf16 matrix pipe with split operands (needs train_dk_x3)
  bool train_fwd_x3 = false;  // option "train_fwd_x3": forward of the LSTM train step on the bf16 matrix pipe with split operands
  bool train_dk_x3 = false;   // option "train_dk_x3": weight-gradient GEMM of the LSTM train step on the bf16 matrix pipe with split operands
  bool train_generic = false;  // option "train_generic": the LSTM train step on the any-shape path (lstm_generic.hip) whatever the shape (tests, A/B)
  bool train_gen1 = false;    // option "train_gen1": the fp32 train step on the first-generation kernels (lstm_bwd_kernel + dx_kernel + db partials; A/B timing and tests)
  bool train_pair_dedup = true; // option "train_pair_dedup": run the source encoder once per (pos, neg) pair of rows that share it
  bool train_serial = false; // option "train_serial": both encoders on one stream (profiling: isolated kernel times)
  float *emb_pad = nullptr;  // [V][Ep]
  // source_only_cnn
  int cnn_W[4] = {-1, -1, -1, -1}, cnn_b[4] = {-1, -1, -1, -1}, cnn_M = -1, tgt_table = -1;
  float *cnn_Wc = nullptr, *cnn_bias = nullptr, *cnn_Mp = nullptr;
  int32_t *err_flag = nullptr;
  // index
  float *idxp = nullptr;
  size_t idxp_cap = 0;
  double *idx64 = nullptr;
  DevBuf idx_rm;            // row-major f32 copy of a SMALL index (re-scoring gathers, RescoreArgs::idx_rm); valid: idx_rm_valid
  bool idx_rm_valid = false;
  int64_t idx_N = 0, idx_base = 0;
  int idx_S = 0;
  float idx_norm_max = 1.0f;
  // scratch
  DevBuf s_ids, s_out, s_q, s_qp, s_ps, s_pi, s_cert, s_os, s_oi, s_tmp, s_tmp2, s_feat, s_zero, s_map;
  DevBuf s_qmap, s_qc;     // fp32 second chance of the bf16 candidate pass: the uncertified queries as a dense set
  DevBuf s_qp32, s_fb_cnt;  // fp32 second chance of the bf16 candidate pass: fp32 query fragments, counter
  DevBuf g_A, g_G, g_c, g_hl, g_raw;  // any-shape LSTM encode (lstm_generic.hip)
  DevBuf g_KT[2], g_Kq[2], g_MT[2];    // ... its packed weights per side, valid while g_ver[side] == weights_version (they
  uint64_t g_ver[2] = {0, 0};          //     were rebuilt by every call)
  DevBuf s_xchg;     // sse_score_topk_sharded_dev: [local lists | gathered lists] of the RCCL exchange
  DevBuf s_persist;  // lstm_persist.hip: h_t / raw-encoding exchange buffers and arrival counters
  DevBuf s_cluster;  // lstm_cluster.hip: h_t / sum-of-squares exchange buffers
  int32_t *pin_small = nullptr;  // 64 pinned host words: error flag / loss read-backs (a pageable target makes the copy a blocking one)
  void *pin = nullptr;  // pinned host staging of the host-buffer scoring entry points: [scores | ids | certificates]
  size_t pin_cap = 0;
  int32_t score_seq = 0;  // call number the re-scoring pass stores into the pinned completion flags (ScoreMirror)
  DevBuf s_pb, s_cthr, s_cslot, s_ccnt, s_cbuf;  // per-split bounds; collect path: thresholds, slots, counters, row buffers
  const int32_t *cur_row_map = nullptr;  // set by sse_encode around its launch
  bool cur_padded_hint = false;          // ... together with this: the batch's mean leading-PAD count is >= T / 4
  // training
  float lr = 0.9f;
  int64_t global_step = 0;
  TrainState *train = nullptr;
  std::vector<hipEvent_t> events;
};

namespace {

int fail(sse_handle *h, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return 1;
}

#define HIPCHECK(h, expr)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Wait for a stream: poll for up to ~3 ms (the latency-critical calls finish in 0.1 - 0.5 ms; the runtime's blocking wait
// was seen to add 1 - 2 ms to some of them: an 8-token query against a 1.25 M-row index took 2.0 ms end to end with 0.3 ms
// of device work), then block.
hipError_t sync_stream(hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) return hipStreamSynchronize(st);
  }
}

int reserve(sse_handle *h, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) HIPCHECK(h, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  const size_t cap = bytes + bytes / 4 + 256;
  HIPCHECK(h, hipMalloc(&b.p, cap));
  b.cap = cap;
  return 0;
}

int add_var(sse_handle *h, const std::string &name, int rows, int cols) {
  Variable v;
  v.name = name;
  v.rows = rows;
  v.cols = cols;
  v.count = (int64_t)rows * cols;
  h->vars.push_back(v);
  return (int)h->vars.size() - 1;
}

int find_var(sse_handle *h, const char *name, bool *is_slot) {
  std::string n(name);
  *is_slot = false;
  const std::string suffix = "/Adagrad";
  if (n.size() > suffix.size() && n.compare(n.size() - suffix.size(), suffix.size(), suffix) == 0) {
    n.resize(n.size() - suffix.size());
    *is_slot = true;
  }
  for (size_t i = 0; i < h->vars.size(); ++i)
    if (h->vars[i].name == n) return (int)i;
  return -1;
}

int setup_encoder(sse_handle *h, Encoder &e, const std::string &scope, const std::string &proj, int H, int proj_rows) {
  const sse_config &c = h->cfg;
  e.H = H;
  e.kernel = add_var(h, scope + "/rnn/basic_lstm_cell/kernel", c.embedding_size + H, 4 * H);
  e.bias = add_var(h, scope + "/rnn/basic_lstm_cell/bias", 1, 4 * H);
  e.proj = add_var(h, proj, proj_rows, c.encoding_size);
  return 0;
}

int geometry(sse_handle *h, Encoder &e) {
  const sse_config &c = h->cfg;
  if (e.H <= 0) return 0;
  e.Hp = e.H <= 128 ? 128 : e.H <= 256 ? 256 : round_up(e.H, 512);
  e.UB = e.Hp / 128;
  e.Ep = round_up(c.embedding_size + 1, 8);  // at least one padding column: it holds the constant 1 of the bias row
  e.KGx = e.Ep / 8;
  e.KGh = e.Hp / 8;
  // Shapes the fused kernels are not laid out for -- cell size > 512, an embedding too wide for the LDS tile, encoding_size >
  // 512 -- run the any-shape path (lstm_generic.hip) instead of being rejected (round 5; the reference accepts any size,
  // sse_train.py:60-74).  (The scorer's own limit, index dimension <= 1024, is separate.)
  e.generic = e.H > 512 || c.encoding_size > 512 || lstm_fwd_lds_bytes(e.KGx, e.KGh, e.Hp == 512 ? 1 : 2) > 160 * 1024;
  return 0;
}

// row stride of the padded embedding table: the LSTM kernels consume k in groups of 8 and need one padding column
// for the constant 1 that carries the bias; the CNN only needs 16-byte aligned window starts (E = 50: 52 instead of
// 56 -> 7 % fewer MFMAs)
static int emb_cols(const sse_config &c) {
  return c.network_mode == SSE_MODE_SOURCE_ONLY_CNN ? round_up(c.embedding_size, 4) : round_up(c.embedding_size + 1, 8);
}

// (re)build the kernel-facing layouts from the master variables
// the packed projections alone (what a train step whose LSTM kernels all run on split operands reads of these layouts):
// the embedding pad, the fp32 kernel fragments and the CNN layouts stay stale (packed_dirty) until an encode needs them
int ensure_proj_packed(sse_handle *h, hipStream_t st) {
  if (!h->packed_dirty || h->mp_fresh) return 0;
  const sse_config &c = h->cfg;
  for (int s = 0; s < 2; ++s) {
    Encoder &e = h->enc[s];
    if (e.H <= 0 || e.kernel < 0 || e.generic) continue;
    const int NTS = (c.encoding_size + 31) / 32;
    if (!e.Mp) HIPCHECK(h, hipMalloc((void **)&e.Mp, (size_t)NTS * e.KGh * 256 * sizeof(float)));
    HIPCHECK(h, launch_pack_kn(h->vars[e.proj].dev, e.H, c.encoding_size, e.KGh, e.Mp, st));
  }
  h->mp_fresh = true;
  return 0;
}

int ensure_packed(sse_handle *h, hipStream_t st) {
  if (!h->packed_dirty) return 0;
  const sse_config &c = h->cfg;
  const int Ep = emb_cols(c);
  h->emb16_valid = false;
  if (!h->emb_pad) HIPCHECK(h, hipMalloc((void **)&h->emb_pad, (size_t)c.vocab_size * Ep * sizeof(float)));
  HIPCHECK(h, launch_pad_rows(h->vars[0].dev, c.vocab_size, c.embedding_size, Ep,
                              c.network_mode == SSE_MODE_SOURCE_ONLY_CNN ? -1 : c.embedding_size, h->emb_pad, st));
  for (int s = 0; s < 2; ++s) {
    Encoder &e = h->enc[s];
    if (e.H <= 0 || e.kernel < 0 || e.generic) continue;
    e.pad_valid = false;
    e.pad_valid_small = false;
    e.waug_valid = false;
    e.wc_valid = false;
    e.x3_valid = false;
    e.pad_valid_x3 = false;
    const int KG = e.KGx + e.KGh;
    if (e.shares_lstm_with < 0) {
      if (!e.Wp) HIPCHECK(h, hipMalloc((void **)&e.Wp, (size_t)(e.Hp / 32) * KG * 4 * 256 * sizeof(float)));
      HIPCHECK(h, launch_pack_lstm(h->vars[e.kernel].dev, h->vars[e.bias].dev, c.embedding_size, e.H, e.Ep, e.Hp, e.UB,
                                   e.Wp, st));
    } else {
      e.Wp = h->enc[e.shares_lstm_with].Wp;
    }
    const int NTS = (c.encoding_size + 31) / 32;
    if (!e.Mp) HIPCHECK(h, hipMalloc((void **)&e.Mp, (size_t)NTS * e.KGh * 256 * sizeof(float)));
    HIPCHECK(h, launch_pack_kn(h->vars[e.proj].dev, e.H, c.encoding_size, e.KGh, e.Mp, st));
  }
  if (c.network_mode == SSE_MODE_SOURCE_ONLY_CNN) {
    if (c.encoding_size > 512) return fail(h, "encoding_size %d > 512 not supported", c.encoding_size);
    if (!h->cnn_Wc) HIPCHECK(h, hipMalloc((void **)&h->cnn_Wc, cnn_packed_weight_floats(Ep) * sizeof(float)));
    if (!h->cnn_bias) HIPCHECK(h, hipMalloc((void **)&h->cnn_bias, 576 * sizeof(float)));
    const int NTS = (c.encoding_size + 31) / 32;
    if (!h->cnn_Mp) HIPCHECK(h, hipMalloc((void **)&h->cnn_Mp, (size_t)NTS * 72 * 256 * sizeof(float)));
    const float *W[4];
    static const int foff[4] = {0, 256, 384, 512}, nf[4] = {256, 128, 128, 64};
    for (int i = 0; i < 4; ++i) {
      W[i] = h->vars[h->cnn_W[i]].dev;
      HIPCHECK(h, hipMemcpyAsync(h->cnn_bias + foff[i], h->vars[h->cnn_b[i]].dev, nf[i] * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
    }
    HIPCHECK(h, launch_pack_conv(W, c.embedding_size, Ep, h->cnn_Wc, st));
    if (h->cnn_bf16) {
      const int Ep8 = round_up(c.embedding_size, 8);
      if (!h->emb_bf16) HIPCHECK(h, hipMalloc((void **)&h->emb_bf16, (size_t)c.vocab_size * Ep8 * sizeof(unsigned short)));
      if (!h->cnn_Wc16) HIPCHECK(h, hipMalloc((void **)&h->cnn_Wc16, cnn_bf16_packed_weight_elems(Ep8) * sizeof(unsigned short)));
      HIPCHECK(h, launch_cnn_bf16_pack(h->vars[0].dev, c.vocab_size, c.embedding_size, Ep8, h->emb_bf16, W, h->cnn_Wc16, st));
      if (!h->cnn_Mx3) HIPCHECK(h, hipMalloc((void **)&h->cnn_Mx3, cnn_proj_x3_elems(c.encoding_size) * sizeof(unsigned short)));
      HIPCHECK(h, launch_pack_cnn_proj_x3(h->vars[h->cnn_M].dev, c.encoding_size, h->cnn_Mx3, st));
    }
    HIPCHECK(h, launch_pack_kn(h->vars[h->cnn_M].dev, 576, c.encoding_size, 72, h->cnn_Mp, st));
  }
  h->packed_dirty = false;
  h->mp_fresh = true;
  return 0;
}

// The layouts a pure fp32 LSTM train step reads -- padded embedding table, packed kernels and projections (what ensure_packed
// builds for the LSTM modes) and the fp32 Kh^T / Kx^T fragment copies of the BPTT kernel -- in ONE launch (launch_pack_multi).
int pack_train_fp32(sse_handle *h, TrainState &ts, int nside, hipStream_t st) {
  const sse_config &c = h->cfg;
  PackJobs js;
  if (h->packed_dirty) {
    const int Ep = emb_cols(c);
    h->emb16_valid = false;
    if (!h->emb_pad) HIPCHECK(h, hipMalloc((void **)&h->emb_pad, (size_t)c.vocab_size * Ep * sizeof(float)));
    pack_job_pad_rows(js, h->vars[0].dev, c.vocab_size, c.embedding_size, Ep, c.embedding_size, h->emb_pad);
    for (int s = 0; s < 2; ++s) {
      Encoder &e = h->enc[s];
      if (e.H <= 0 || e.kernel < 0) continue;
      e.pad_valid = e.pad_valid_small = e.waug_valid = e.wc_valid = e.x3_valid = e.pad_valid_x3 = false;
      const int KG = e.KGx + e.KGh;
      if (e.shares_lstm_with < 0) {
        if (!e.Wp) HIPCHECK(h, hipMalloc((void **)&e.Wp, (size_t)(e.Hp / 32) * KG * 4 * 256 * sizeof(float)));
        pack_job_lstm(js, h->vars[e.kernel].dev, h->vars[e.bias].dev, c.embedding_size, e.H, e.Ep, e.Hp, e.Wp);
      } else {
        e.Wp = h->enc[e.shares_lstm_with].Wp;
      }
      const int NTS = (c.encoding_size + 31) / 32;
      if (!e.Mp) HIPCHECK(h, hipMalloc((void **)&e.Mp, (size_t)NTS * e.KGh * 256 * sizeof(float)));
      pack_job_kn(js, h->vars[e.proj].dev, e.H, c.encoding_size, e.KGh, e.Mp);
    }
  }
  if (ts.fp32_dirty) {
    for (int s = 0; s < nside; ++s) {
      Encoder &e = h->enc[s];
      if (e.shares_lstm_with >= 0) {
        ts.KhT[s] = ts.KhT[e.shares_lstm_with];
        ts.KxT[s] = ts.KxT[e.shares_lstm_with];
        continue;
      }
      if (!ts.KhT[s]) HIPCHECK(h, hipMalloc((void **)&ts.KhT[s], (size_t)(e.Hp / 32) * (e.Hp / 2) * 256 * sizeof(float)));
      if (!ts.KxT[s]) HIPCHECK(h, hipMalloc((void **)&ts.KxT[s], (size_t)2 * (e.Hp / 2) * 256 * sizeof(float)));
      pack_job_kT(js, h->vars[e.kernel].dev, c.embedding_size, e.H, e.Hp / 32, e.H, e.Hp, ts.KhT[s]);
      pack_job_kT(js, h->vars[e.kernel].dev, 0, c.embedding_size, 2, e.H, e.Hp, ts.KxT[s]);
    }
  }
  HIPCHECK(h, launch_pack_multi(js, st));
  h->packed_dirty = false;
  h->mp_fresh = true;
  ts.fp32_dirty = false;
  return 0;
}

void fill_fwd_args(sse_handle *h, Encoder &e, LstmFwdArgs &a) {
  const sse_config &c = h->cfg;
  a.emb = h->emb_pad;
  a.Wp = e.Wp;
  a.Mp = e.Mp;
  a.err = h->err_flag;
  a.V = c.vocab_size;
  a.Ep = e.Ep;
  a.KGx = e.KGx;
  a.KGh = e.KGh;
  a.H = e.H;
  a.KGhe = (e.H + 7) / 8;
  a.S = c.encoding_size;
  a.NTS = (c.encoding_size + 31) / 32;
  a.gate_split = h->lstm_gate_split ? 1 : 0;
  a.cu_count = h->cu_count;  // (set by sse_create)
}

// state after p leading PAD steps for p = 0..T: one all-PAD row through the SAME kernel
// (bit-identical arithmetic), recorded step by step
int ensure_pad_table(sse_handle *h, int side, int T, hipStream_t st) {
  Encoder &e = h->enc[side];
  Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
  if (!(own.pad_valid && own.pad_T >= T)) {
    const int Tt = T > h->cfg.max_seq_length ? T : h->cfg.max_seq_length;
    if (own.pad_T < Tt || !own.pad_h) {
      if (own.pad_h) HIPCHECK(h, hipFree(own.pad_h));
      if (own.pad_c) HIPCHECK(h, hipFree(own.pad_c));
      own.pad_h = own.pad_c = nullptr;
      HIPCHECK(h, hipMalloc((void **)&own.pad_h, (size_t)(Tt + 1) * own.Hp * sizeof(float)));
      HIPCHECK(h, hipMalloc((void **)&own.pad_c, (size_t)(Tt + 1) * own.Hp * sizeof(float)));
      own.pad_T = Tt;
    }
    HIPCHECK(h, hipMemsetAsync(own.pad_h, 0, (size_t)(Tt + 1) * own.Hp * sizeof(float), st));
    HIPCHECK(h, hipMemsetAsync(own.pad_c, 0, (size_t)(Tt + 1) * own.Hp * sizeof(float), st));
    if (reserve(h, h->s_zero, (size_t)Tt * sizeof(int32_t) + (size_t)h->cfg.encoding_size * sizeof(float))) return 1;
    HIPCHECK(h, hipMemsetAsync(h->s_zero.p, 0, (size_t)Tt * sizeof(int32_t), st));
    LstmFwdArgs a;
    fill_fwd_args(h, own, a);
    a.ids = (const int32_t *)h->s_zero.p;
    a.out = (float *)((char *)h->s_zero.p + (size_t)Tt * sizeof(int32_t));
    a.B = 1;
    a.T = Tt;
    a.normalize = 0;
    a.rec_h = own.pad_h;
    a.rec_c = own.pad_c;
    HIPCHECK(h, launch_lstm_fwd(a, own.Hp, st));
    own.pad_valid = true;
  }
  e.pad_h = own.pad_h;
  e.pad_c = own.pad_c;
  return 0;
}

int ensure_waug(sse_handle *h, int side, hipStream_t st) {
  Encoder &e = h->enc[side];
  Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
  if (own.waug_valid) return 0;
  const int E = h->cfg.embedding_size;
  if (!own.Waug) HIPCHECK(h, hipMalloc((void **)&own.Waug, lstm_small_waug_floats(E, own.H) * sizeof(float)));
  HIPCHECK(h, launch_pack_lstm_small(h->vars[own.kernel].dev, h->vars[own.bias].dev, E, own.H, own.Waug, st));
  own.waug_valid = true;
  return 0;
}

void fill_small_args(sse_handle *h, Encoder &e, LstmSmallArgs &a) {
  const sse_config &c = h->cfg;
  Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
  a.emb = h->vars[0].dev;
  a.Waug = own.Waug;
  a.M = h->vars[e.proj].dev;
  a.err = h->err_flag;
  a.V = c.vocab_size;
  a.E = c.embedding_size;
  a.H = e.H;
  a.S = c.encoding_size;
  a.pad_stride = e.H;
}

// pad-prefix table of the few-sequences kernel: one all-PAD row through THAT kernel (its own arithmetic, bit for bit)
int ensure_pad_table_small(sse_handle *h, int side, int T, hipStream_t st) {
  Encoder &e = h->enc[side];
  Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
  if (own.pad_valid_small && own.pad_T_small >= T) return 0;
  const int Tt = T > h->cfg.max_seq_length ? T : h->cfg.max_seq_length;
  if (own.pad_T_small < Tt || !own.pad_h_small) {
    if (own.pad_h_small) HIPCHECK(h, hipFree(own.pad_h_small));
    if (own.pad_c_small) HIPCHECK(h, hipFree(own.pad_c_small));
    own.pad_h_small = own.pad_c_small = nullptr;
    HIPCHECK(h, hipMalloc((void **)&own.pad_h_small, (size_t)(Tt + 1) * own.H * sizeof(float)));
    HIPCHECK(h, hipMalloc((void **)&own.pad_c_small, (size_t)(Tt + 1) * own.H * sizeof(float)));
    own.pad_T_small = Tt;
  }
  HIPCHECK(h, hipMemsetAsync(own.pad_h_small, 0, (size_t)(Tt + 1) * own.H * sizeof(float), st));
  HIPCHECK(h, hipMemsetAsync(own.pad_c_small, 0, (size_t)(Tt + 1) * own.H * sizeof(float), st));
  if (reserve(h, h->s_zero, (size_t)Tt * sizeof(int32_t) + (size_t)h->cfg.encoding_size * sizeof(float))) return 1;
  HIPCHECK(h, hipMemsetAsync(h->s_zero.p, 0, (size_t)Tt * sizeof(int32_t), st));
  if (ensure_waug(h, side, st)) return 1;
  LstmSmallArgs a;
  // the LSTM of `own` with any projection of that cell size (the table only records h, c)
  fill_small_args(h, own, a);
  a.ids = (const int32_t *)h->s_zero.p;
  a.out = (float *)((char *)h->s_zero.p + (size_t)Tt * sizeof(int32_t));
  a.B = 1;
  a.T = Tt;
  a.normalize = 0;
  a.rec_h = own.pad_h_small;
  a.rec_c = own.pad_c_small;
  HIPCHECK(h, launch_lstm_small(a, st));
  own.pad_valid_small = true;
  return 0;
}

// Batches the MFMA cluster kernel (lstm_cluster.hip) takes: 33 rows up to lstm_cluster_chunks launches of lstm_cluster_rows
// (<= 1024) rows.  A launch costs about the same 0.3 ms whatever it holds (T = 32), a 32-row tile of the matrix kernel 1.2 ms
// whether 33 or 8192 rows run beside it: three launches are still ahead of it, four are not.
// With the split-bf16 matrix kernel opted in (lstm_x3: ~0.45 ms for anything up to 8192 rows, ~1e-5 from fp32) a second
// launch no longer pays: one launch only.
static bool x3_applies(const sse_handle *h, const Encoder &e) {
  return h->lstm_x3 && e.Hp <= 256 && e.H >= 64 && h->cfg.embedding_size < 64;
}
static int cluster_row_limit(const sse_handle *h, const Encoder &e) {
  const int per = std::min(h->lstm_cluster_rows, lstm_cluster_max_rows());
  return per * (x3_applies(h, e) ? 1 : std::max(1, h->lstm_cluster_chunks));
}

static int ensure_cu_count(sse_handle *h) {
  if (h->cu_count == 0) {
    hipDeviceProp_t prop;
    HIPCHECK(h, hipGetDeviceProperties(&prop, h->cfg.device));
    h->cu_count = prop.multiProcessorCount;
  }
  return 0;
}

// Would encode_dev_locked hand a batch of this shape to the MFMA cluster kernel RIGHT NOW?  One test for the launch path and
// for the host-side pad-prefix row sort (which the cluster kernel does not want and the matrix kernel does): shape, the
// back-off counter after a give-up, the device's CU count and the LDS fit all enter (the sort used to be
// switched off by the shape alone, so a backed-off or small device lost the pad skip silently).
static bool cluster_takes(sse_handle *h, const Encoder &e, int B, int T) {
  const sse_config &c = h->cfg;
  if (h->cur_row_map || lstm_small_lds_bytes(c.embedding_size, e.H, c.encoding_size) > 160 * 1024) return false;
  if (!(B > 32 && B <= cluster_row_limit(h, e) && T <= lstm_persist_max_steps() &&
        lstm_cluster_ok(c.embedding_size, e.H, c.encoding_size)))
    return false;
  if (h->cluster_skip[1] > 0) return false;
  if (ensure_cu_count(h)) return false;
  const int per = std::min(h->lstm_cluster_rows, lstm_cluster_max_rows());
  const int ncl = (std::min(B, per) + 63) / 64;
  return (ncl <= 8 ? 128 : 256) <= h->cu_count;
}

// Inference encode of an encoder whose shape is outside the fused kernels' layouts: lstm_generic.hip, row chunks sized so
// that the per-step operand matrix A [T][rows][E + H] stays below ~1 GiB.  Same arithmetic, device ids [B][T] in, [B][S] out.
static int encode_generic_locked(sse_handle *h, Encoder &e, const int32_t *ids, int B, int T, int normalize, float *out, hipStream_t st) {
  const sse_config &c = h->cfg;
  const int E = c.embedding_size, S = c.encoding_size;
  const GenLstmDims d0 = gen_lstm_dims(32, T, E, e.H);
  const size_t per_row = (size_t)T * d0.Kp * sizeof(float);
  int rows = (int)std::min<size_t>(8192, std::max<size_t>(32, (((size_t)1 << 30) / per_row) / 32 * 32));
  const GenLstmDims dm = gen_lstm_dims(std::min(rows, B), T, E, e.H);
  if (reserve(h, h->g_A, gen_lstm_a_floats(dm) * sizeof(float)) || reserve(h, h->g_G, (size_t)dm.Bp * 4 * dm.Hq * sizeof(float)) ||
      reserve(h, h->g_c, (size_t)dm.Bp * dm.Hq * sizeof(float)) || reserve(h, h->g_hl, (size_t)dm.Bp * dm.Hq * sizeof(float)) ||
      reserve(h, h->g_raw, (size_t)dm.Bp * S * sizeof(float)))
    return 1;
  const int side = (&e == &h->enc[1]) ? 1 : 0;
  if (h->g_ver[side] != h->weights_version || !h->g_KT[side].p) {
    if (reserve(h, h->g_KT[side], gen_lstm_kt_floats(dm) * sizeof(float)) || reserve(h, h->g_Kq[side], gen_lstm_kt_floats(dm) * sizeof(float)) ||
        reserve(h, h->g_MT[side], (size_t)S * dm.Hq * sizeof(float)))
      return 1;
    HIPCHECK(h, launch_gen_pack(h->vars[e.kernel].dev, h->vars[e.proj].dev, dm, S, (float *)h->g_KT[side].p, (float *)h->g_Kq[side].p,
                                (float *)h->g_MT[side].p, st));
    h->g_ver[side] = h->weights_version;
  }
  for (int b0 = 0; b0 < B; b0 += rows) {
    const int nb = std::min(rows, B - b0);
    const GenLstmDims d = gen_lstm_dims(nb, T, E, e.H);
    HIPCHECK(h, launch_gen_forward(ids + (size_t)b0 * T, h->vars[0].dev, c.vocab_size, (const float *)h->g_KT[side].p, h->vars[e.bias].dev, d,
                                   (float *)h->g_A.p, (float *)h->g_G.p, (float *)h->g_c.p, nullptr, (float *)h->g_hl.p, h->err_flag, st));
    HIPCHECK(h, launch_gen_project((const float *)h->g_hl.p, (const float *)h->g_MT[side].p, d, S, (float *)h->g_raw.p, st));
    if (normalize) HIPCHECK(h, launch_l2_normalize((const float *)h->g_raw.p, out + (size_t)b0 * S, nb, S, st));
    else HIPCHECK(h, hipMemcpyAsync(out + (size_t)b0 * S, h->g_raw.p, (size_t)nb * S * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

// PAD-prefix bucketing for ids that are already on the device (sse_index.py:79-85 left-pads every row):
// what encode_host_ids_locked does with a host counting sort, as two launches on the caller's stream.  *row_map / *padded
// keep the caller's values when the batch is not bucketed.
static int device_pad_sort(sse_handle *h, int side, const int32_t *ids, int B, int T, hipStream_t st, const int32_t **row_map,
                           bool *padded) {
  if (*row_map || !h->pad_skip || h->pad_sort_dev == 0 || T < 2 || T > SSE_PAD_SORT_MAX_T || B <= 64) return 0;
  volatile int32_t *stat = h->pad_stat + 2 * side;
  const int seen = h->pad_sort_dev == 2 ? 2 : (stat[0] != 0 ? (int)stat[1] : 1);  // nothing seen yet: bucket, keep the tile policy
  const bool scatter = seen != 0;
  if (!h->s_padzero.p) {
    if (reserve(h, h->s_padzero, pad_sort_zeroed_words() * sizeof(int32_t))) return 1;
    HIPCHECK(h, hipMemsetAsync(h->s_padzero.p, 0, h->s_padzero.cap, st));
  }
  if (reserve(h, h->s_padwork, pad_sort_work_words(B, T) * sizeof(int32_t))) return 1;
  if (scatter && reserve(h, h->s_padorder, (size_t)B * sizeof(int32_t))) return 1;
  h->pad_seq = h->pad_seq == INT32_MAX ? 1 : h->pad_seq + 1;
  HIPCHECK(h, launch_pad_sort(ids, B, T, (int32_t *)h->s_padzero.p, (int32_t *)h->s_padwork.p, (int32_t *)h->s_padorder.p,
                              h->pad_stat + 2 * side, h->pad_seq, scatter, st));
  if (scatter) {
    *row_map = (const int32_t *)h->s_padorder.p;
    *padded = seen == 2;
    h->pad_sorted_calls += 1;
  }
  return 0;
}

int encode_dev_locked(sse_handle *h, int side, const int32_t *ids, int B, int T, int normalize, float *out,
                      hipStream_t st) {
  const sse_config &c = h->cfg;
  if (side != SSE_SIDE_SOURCE && side != SSE_SIDE_TARGET) return fail(h, "side must be 0 (source) or 1 (target)");
  if (B < 0 || T < 1) return fail(h, "bad batch shape B=%d T=%d", B, T);
  if (B == 0) return 0;
  if (side == SSE_SIDE_TARGET && h->tgt_table >= 0) {
    // source-encoder-only / source_only_cnn: the "target encoder" is the free matrix
    // tgt_seq_embedding [N,S] (sse_model.py:214,233); norm_tgt_seq_embedding ignores the feed
    // and yields all N rows.  Honour that contract when the caller asks for exactly N rows.
    const Variable &tv = h->vars[h->tgt_table];
    if (B != tv.rows)
      return fail(h, "network mode has no target sequence encoder: tgt_seq_embedding is a [%d,%d] variable "
                     "(sse_model.py:214,233); request exactly %d rows", tv.rows, tv.cols, tv.rows);
    if (normalize) HIPCHECK(h, launch_l2_normalize(tv.dev, out, tv.rows, tv.cols, st));
    else HIPCHECK(h, hipMemcpyAsync(out, tv.dev, tv.count * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (c.network_mode == SSE_MODE_SOURCE_ONLY_CNN) {
    if (ensure_packed(h, st)) return 1;
    const int Ep = emb_cols(c);
    if (T < 5) return fail(h, "source_only_cnn needs max_seq_length >= 5 (widest filter)");
    if (cnn_lds_bytes(T, Ep, 0) > 160 * 1024)
      return fail(h, "source_only_cnn: T*E = %d*%d does not fit the LDS tile of the gfx950 kernel", T, c.embedding_size);
    if (reserve(h, h->s_feat, (size_t)((B + 31) / 32) * 72 * 256 * sizeof(float))) return 1;
    if (h->cnn_bf16) {
      const int Ep8 = round_up(c.embedding_size, 8);
      if (cnn_bf16_lds_bytes(T, Ep8, 0) > 160 * 1024)
        return fail(h, "source_only_cnn (bf16): T*E = %d*%d does not fit the LDS tile of the gfx950 kernel", T, c.embedding_size);
      HIPCHECK(h, launch_cnn_fwd_bf16(ids, h->emb_bf16, h->cnn_Wc16, h->cnn_bias, (float *)h->s_feat.p, h->err_flag, B, T,
                                      c.vocab_size, Ep8, nullptr, nullptr, st));
      HIPCHECK(h, launch_cnn_proj_x3((const float *)h->s_feat.p, h->cnn_Mx3, out, B, c.encoding_size, normalize ? 1 : 0, st));
      return 0;
    }
    HIPCHECK(h, launch_cnn_fwd(ids, h->emb_pad, h->cnn_Wc, h->cnn_bias, h->cnn_Mp, (float *)h->s_feat.p, out, h->err_flag, B,
                               T, c.vocab_size, Ep, c.encoding_size, normalize ? 1 : 0, nullptr, nullptr, st));
    return 0;
  }
  Encoder &e = h->enc[side];
  if (e.kernel < 0) return fail(h, "network mode has no %s sequence encoder (sse_model.py:231-233)", side ? "target" : "source");
  if (e.generic) return encode_generic_locked(h, e, ids, B, T, normalize, out, st);
  if (ensure_packed(h, st)) return 1;
  const bool small_ok = !h->cur_row_map && lstm_small_lds_bytes(c.embedding_size, e.H, c.encoding_size) <= 160 * 1024;
  bool persist_shape = small_ok && B <= h->lstm_persist_rows && B <= lstm_persist_max_rows() && T <= lstm_persist_max_steps();
  if (persist_shape && h->cluster_skip[0] > 0) {  // backing off after a launch that gave up
    --h->cluster_skip[0];
    persist_shape = false;
  }
  if (persist_shape) {
    // one to a few queries (sse_demo / webserver): a cluster of workgroups with the weights resident in LDS, see
    // lstm_persist.hip.  Needs every workgroup of the launch resident at once: at most half the CUs are asked for.
    const int nwg = lstm_persist_nwg(c.embedding_size, e.H, c.encoding_size);
    if (ensure_cu_count(h)) return 1;
    if (nwg > 0 && 8 * nwg * 2 <= h->cu_count) {
      if (ensure_waug(h, side, st)) return 1;
      LstmSmallArgs sa;
      fill_small_args(h, e, sa);
      LstmPersistArgs pa;
      pa.emb = sa.emb;
      pa.Waug = sa.Waug;
      pa.M = sa.M;
      pa.err = sa.err;
      pa.V = sa.V;
      pa.E = sa.E;
      pa.H = sa.H;
      pa.S = sa.S;
      pa.pad_stride = sa.pad_stride;
      pa.ids = ids;
      pa.out = out;
      pa.B = B;
      pa.T = T;
      pa.normalize = normalize ? 1 : 0;
      if (h->pad_skip && T > 1) {
        if (ensure_pad_table_small(h, side, T, st)) return 1;  // same arithmetic, same table
        Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
        pa.pad_h = own.pad_h_small;
        pa.pad_c = own.pad_c_small;
      }
      const size_t nhx = lstm_persist_hx_words(e.H), nraw = lstm_persist_raw_words(c.encoding_size);
      const size_t need = (nhx + nraw) * sizeof(unsigned long long);
      if (h->persist_epoch == 0 || h->persist_epoch >= (1u << 20) - 1 || need > h->s_persist.cap) {
        // fresh (or re-sized, or the 20-bit epoch ran out): all tags to "never written"
        if (reserve(h, h->s_persist, need)) return 1;
        HIPCHECK(h, hipMemsetAsync(h->s_persist.p, 0, h->s_persist.cap, st));
        h->persist_epoch = 0;
      }
      pa.epoch = ++h->persist_epoch;
      pa.write_through = h->lstm_cluster_wt;
      pa.plain_launch = h->lstm_cluster_coop ? 0 : 1;
      pa.hx = (unsigned long long *)h->s_persist.p;
      pa.rawx = pa.hx + nhx;
      HIPCHECK(h, launch_lstm_persist(pa, st));
      if (h->persist_inject) {  // testing aid: pretend a workgroup of the cluster never arrived
        static const int32_t four = 4;
        HIPCHECK(h, hipMemcpyAsync(h->err_flag, &four, sizeof four, hipMemcpyHostToDevice, st));
      }
      return 0;
    }
  }
  const bool cluster_shape = cluster_takes(h, e, B, T);
  if (!cluster_shape && h->cluster_skip[1] > 0 && small_ok && B > 32 && B <= cluster_row_limit(h, e) &&
      T <= lstm_persist_max_steps() && lstm_cluster_ok(c.embedding_size, e.H, c.encoding_size))
    --h->cluster_skip[1];  // an eligible call spent backing off
  if (cluster_shape) {
    // mid-size batches (the evaluator's 600, the index builder's 1000): the hidden units of every 64-row tile spread over a
    // cluster of 16 compute units, weights in LDS, h_t exchanged per step (lstm_cluster.hip); needs one CU per workgroup
    const int per = std::min(h->lstm_cluster_rows, lstm_cluster_max_rows());
    {
      Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
      if (!own.wc_valid) {
        if (!own.Wc) HIPCHECK(h, hipMalloc((void **)&own.Wc, lstm_cluster_weight_floats(c.embedding_size, own.H) * sizeof(float)));
        HIPCHECK(h, launch_pack_lstm_cluster(h->vars[own.kernel].dev, h->vars[own.bias].dev, c.embedding_size, own.H, own.Wc, st));
        own.wc_valid = true;
      }
      LstmClusterArgs ca;
      ca.ids = ids;
      ca.emb = h->emb_pad;
      ca.Wc = own.Wc;
      ca.Mp = e.Mp;
      ca.out = out;
      ca.err = h->err_flag;
      ca.B = B;
      ca.T = T;
      ca.V = c.vocab_size;
      ca.E = c.embedding_size;
      ca.Ep = e.Ep;
      ca.KGx = e.KGx;
      ca.H = e.H;
      ca.Hp = e.H <= 128 ? 128 : 256;
      ca.KGh = ca.Hp / 8;
      ca.S = c.encoding_size;
      ca.NTS = (c.encoding_size + 31) / 32;
      ca.normalize = normalize ? 1 : 0;
      if (h->pad_skip && T > 1) {
        if (ensure_pad_table_small(h, side, T, st)) return 1;  // lstm_small's table: the same arithmetic
        ca.pad_h = own.pad_h_small;
        ca.pad_c = own.pad_c_small;
        ca.pad_stride = e.H;
      }
      const size_t nhx = lstm_cluster_hx_words(e.H), nsx = lstm_cluster_sx_words();
      const size_t need = (nhx + nsx) * sizeof(unsigned long long);
      if (h->cluster_epoch == 0 || h->cluster_epoch >= (1u << 20) - 1 || need > h->s_cluster.cap) {
        if (reserve(h, h->s_cluster, need)) return 1;
        HIPCHECK(h, hipMemsetAsync(h->s_cluster.p, 0, h->s_cluster.cap, st));
        h->cluster_epoch = 0;
      }
      ca.write_through = h->lstm_cluster_wt;
      ca.plain_launch = h->lstm_cluster_coop ? 0 : 1;
      ca.drop_wg = h->lstm_cluster_drop;
      ca.hx = (unsigned long long *)h->s_cluster.p;
      ca.sx = ca.hx + nhx;
      for (int c0 = 0; c0 < B; c0 += per) {  // launches of the same stream reuse the exchange buffers under new epochs
        if (h->cluster_epoch >= (1u << 20) - 1) {
          HIPCHECK(h, hipMemsetAsync(h->s_cluster.p, 0, h->s_cluster.cap, st));
          h->cluster_epoch = 0;
        }
        ca.epoch = ++h->cluster_epoch;
        ca.ids = ids + (size_t)c0 * T;
        ca.out = out + (size_t)c0 * c.encoding_size;
        ca.B = std::min(per, B - c0);
        HIPCHECK(h, launch_lstm_cluster(ca, st));
      }
      if (h->persist_inject) {  // testing aid: pretend a workgroup of the cluster never arrived
        static const int32_t four = 4;
        HIPCHECK(h, hipMemcpyAsync(h->err_flag, &four, sizeof four, hipMemcpyHostToDevice, st));
      }
      return 0;
    }
  }
  if (B <= h->lstm_small_rows && small_ok) {
    // a handful of sequences (demo / web query, last batch of an index build): GEMV on the vector ALUs, see lstm_small.hip
    if (ensure_waug(h, side, st)) return 1;
    LstmSmallArgs sa;
    fill_small_args(h, e, sa);
    sa.ids = ids;
    sa.out = out;
    sa.B = B;
    sa.T = T;
    sa.normalize = normalize ? 1 : 0;
    if (h->pad_skip && T > 1) {
      if (ensure_pad_table_small(h, side, T, st)) return 1;
      Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
      sa.pad_h = own.pad_h_small;
      sa.pad_c = own.pad_c_small;
    }
    HIPCHECK(h, launch_lstm_small(sa, st));
    return 0;
  }
  if (x3_applies(h, e)) {  // (tiny cells: nothing to gain, and their raw
    // encodings can be small enough for the 2e-6 absolute error to matter after normalisation)
    // opt-in: the gate GEMMs on the bf16 matrix pipe with hi + lo split operands (lstm_fwd_x3.hip); ~1e-5 from the fp32 path
    Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
    const int E = c.embedding_size;
    if (!own.x3_valid || !h->emb16_valid) {
      if (!own.Wx3) HIPCHECK(h, hipMalloc((void **)&own.Wx3, lstm_x3_weight_elems(E, own.Hp) * sizeof(unsigned short)));
      if (!h->emb16) HIPCHECK(h, hipMalloc((void **)&h->emb16, lstm_x3_emb_elems(c.vocab_size, E) * sizeof(unsigned short)));
      HIPCHECK(h, launch_pack_lstm_x3(h->vars[own.kernel].dev, h->vars[own.bias].dev, h->vars[0].dev, c.vocab_size, E, own.H,
                                      own.Hp, own.Wx3, h->emb16_valid ? nullptr : h->emb16, 0, st));
      own.x3_valid = true;
      h->emb16_valid = true;
    }
    LstmX3Args xa;
    xa.ids = ids;
    xa.emb16 = h->emb16;
    xa.Wx3 = own.Wx3;
    xa.Mp = e.Mp;
    xa.out = out;
    xa.err = h->err_flag;
    xa.B = B;
    xa.T = T;
    xa.V = c.vocab_size;
    xa.KGX = lstm_x3_kgx(E);
    xa.H = e.H;
    xa.S = c.encoding_size;
    xa.NTS = (c.encoding_size + 31) / 32;
    xa.normalize = normalize ? 1 : 0;
    xa.row_map = h->cur_row_map;
    {
      bool padded_x3 = false;
      if (device_pad_sort(h, side, ids, B, T, st, &xa.row_map, &padded_x3)) return 1;
    }
    if (h->pad_skip && T > 1) {
      // pad-prefix table of THIS path: one all-PAD row through the same kernel, recorded step by step
      if (!(own.pad_valid_x3 && own.pad_T_x3 >= T)) {
        const int Tt = T > c.max_seq_length ? T : c.max_seq_length;
        if (own.pad_T_x3 < Tt || !own.pad_h_x3) {
          if (own.pad_h_x3) HIPCHECK(h, hipFree(own.pad_h_x3));
          if (own.pad_c_x3) HIPCHECK(h, hipFree(own.pad_c_x3));
          own.pad_h_x3 = own.pad_c_x3 = nullptr;
          HIPCHECK(h, hipMalloc((void **)&own.pad_h_x3, (size_t)(Tt + 1) * own.Hp * sizeof(float)));
          HIPCHECK(h, hipMalloc((void **)&own.pad_c_x3, (size_t)(Tt + 1) * own.Hp * sizeof(float)));
          own.pad_T_x3 = Tt;
        }
        HIPCHECK(h, hipMemsetAsync(own.pad_h_x3, 0, (size_t)(Tt + 1) * own.Hp * sizeof(float), st));
        HIPCHECK(h, hipMemsetAsync(own.pad_c_x3, 0, (size_t)(Tt + 1) * own.Hp * sizeof(float), st));
        if (reserve(h, h->s_zero, (size_t)Tt * sizeof(int32_t) + (size_t)c.encoding_size * sizeof(float))) return 1;
        HIPCHECK(h, hipMemsetAsync(h->s_zero.p, 0, (size_t)Tt * sizeof(int32_t), st));
        LstmX3Args ta = xa;
        ta.ids = (const int32_t *)h->s_zero.p;
        ta.out = (float *)((char *)h->s_zero.p + (size_t)Tt * sizeof(int32_t));
        ta.B = 1;
        ta.T = Tt;
        ta.normalize = 0;
        ta.row_map = nullptr;
        ta.rec_h = own.pad_h_x3;
        ta.rec_c = own.pad_c_x3;
        HIPCHECK(h, launch_lstm_fwd_x3(ta, st));
        own.pad_valid_x3 = true;
      }
      xa.pad_h = own.pad_h_x3;
      xa.pad_c = own.pad_c_x3;
    }
    HIPCHECK(h, launch_lstm_fwd_x3(xa, st));
    return 0;
  }
  LstmFwdArgs a;
  fill_fwd_args(h, e, a);
  a.ids = ids;
  a.out = out;
  a.B = B;
  a.T = T;
  a.normalize = normalize ? 1 : 0;
  a.row_map = h->cur_row_map;
  bool padded_hint = h->cur_padded_hint;
  if (device_pad_sort(h, side, ids, B, T, st, &a.row_map, &padded_hint)) return 1;
  if (h->pad_skip && T > 1) {
    if (ensure_pad_table(h, side, T, st)) return 1;
    a.pad_h = e.pad_h;
    a.pad_c = e.pad_c;
    // a heavily left-padded batch sorted by pad count (sse_encode's host path knows both): tiles differ in length, the finer
    // 32-row granularity with two workgroups per CU balances them better -- crosslingual index build 2.96 -> 2.64 ms, queries
    // 1.24 -> 0.84 ms (profiles/r05_notes.txt)
    if (padded_hint && a.row_map && e.Hp <= 256) a.force_rows = 32;
  }
  HIPCHECK(h, launch_lstm_fwd(a, e.Hp, st));
  return 0;
}

// bits: (optional) receives the raw flag; a flag that is exactly bit 2 (cluster kernel: a workgroup did not arrive) is then
// cleared and reported through *bits with rc 0, so that the caller can re-run the batch on another kernel
int check_err_flag(sse_handle *h, hipStream_t st, int32_t *bits = nullptr) {
  HIPCHECK(h, hipMemcpyAsync(h->pin_small, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHECK(h, sync_stream(st));
  const int32_t flag = h->pin_small[0];
  if (bits) *bits = flag;
  if (flag) {
    HIPCHECK(h, hipMemsetAsync(h->err_flag, 0, sizeof(int32_t), st));
    if (bits && flag == 4) return 0;
    if (flag & 1) return fail(h, "token id out of range [0, %d) (tf.gather would raise; sse_model.py:163-164)", h->cfg.vocab_size);
    if (flag & 2) return fail(h, "corpus row out of range in a train step by rows");
    if (flag & 8) return fail(h, "data-parallel embedding-gradient exchange: more touched rows than the packed buffer holds, or a row id "
                                 "out of range in a gathered buffer (sse_train_pack_embedding_grad / sse_train_unpack_embedding_grad)");
    if (flag & 4) return fail(h, "LSTM cluster kernel: a workgroup of a cluster did not arrive (device oversubscribed?); the "
                                 "host-buffer entry points fall back to the few-sequences kernel by themselves, for "
                                 "sse_encode_dev set option lstm_persist_rows to 0");
    return fail(h, "device error flag 0x%x", flag);
  }
  return 0;
}

int index_from_dev_rows(sse_handle *h, const float *rows_dev, int64_t N, int S, int64_t id_base, hipStream_t st) {
  if (N <= 0 || S <= 0) return fail(h, "empty index");
  if (S > SSE_MAX_INDEX_DIM) return fail(h, "index dimension %d > %d not supported by the scoring kernel", S, SSE_MAX_INDEX_DIM);
  if (N > (int64_t)2147483000) return fail(h, "index shard too large for int32 row ids");
  const int KG = (S + 7) / 8;
  const int64_t NT = (N + 31) / 32;
  // fragment-order copy of the index: grow-only (re-indexing with the same or a smaller shard reuses the allocation)
  const size_t need = (size_t)NT * KG * 256 * sizeof(float);
  if (need > h->idxp_cap) {
    if (h->idxp) HIPCHECK(h, hipFree(h->idxp));
    h->idxp = nullptr;
    h->idxp_cap = 0;
    HIPCHECK(h, hipMalloc((void **)&h->idxp, need));
    h->idxp_cap = need;
  }
  if (reserve(h, h->s_tmp2, 16)) return 1;
  HIPCHECK(h, hipMemsetAsync(h->s_tmp2.p, 0, 4, st));
  HIPCHECK(h, launch_pack_rows_norm(rows_dev, N, S, h->idxp, (float *)h->s_tmp2.p, st));
  float n2 = 0;
  HIPCHECK(h, hipMemcpyAsync(&n2, h->s_tmp2.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHECK(h, hipStreamSynchronize(st));
  h->idx_norm_max = std::sqrt(n2);
  // small indexes (the evaluator's 571 targets, anything up to 64 MiB) also keep their rows as they came: the float64
  // re-scoring of 16384 queries x 10 rows gathered every row from 64 cache lines of the fragment copy
  h->idx_rm_valid = false;
  if ((S & 3) == 0 && (size_t)N * S * sizeof(float) <= ((size_t)64 << 20) && (reinterpret_cast<uintptr_t>(rows_dev) & 15) == 0) {
    if (reserve(h, h->idx_rm, (size_t)N * S * sizeof(float))) return 1;
    HIPCHECK(h, hipMemcpyAsync(h->idx_rm.p, rows_dev, (size_t)N * S * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHECK(h, hipStreamSynchronize(st));  // (rows_dev may be the caller's, or scratch that is reused)
    h->idx_rm_valid = true;
  }
  h->idxp16_valid = false;
  h->idx_N = N;
  h->idx_S = S;
  h->idx_base = id_base;
  return 0;
}

// splits of the index range for a candidate / collect sweep (see score_dev_locked)
static int choose_nsplit(int NQ, int QB, int64_t NT) {
  // enough workgroups to fill the 256 CUs (2 waves of them when the sweep is long), at least 16 n-tiles (2 per
  // wave) per split
  int nsplit = 1;
  const int max_split = (NQ == 1) ? 256 : 128;  // 16 candidates per split, at most RS_MAXNC = 4096 per query
  while (nsplit < max_split && QB * nsplit < 512 && NT / (nsplit * 2) >= 16) nsplit *= 2;
  if (nsplit <= 8) {
    nsplit = 1;
    while (nsplit < 8 && QB * nsplit < 256 && NT / (nsplit * 2) >= 8) nsplit *= 2;
  }
  // one workgroup per CU is resident (the query block fills LDS): the launch runs in ceil(WGs/256) rounds and the
  // last round may be nearly empty (782 query blocks = 3.05 rounds ran at 76 % of the 64-block rate).  Split
  // further while that evens the rounds out by more than the extra per-split cost (~1 % per doubling).
  if (NQ >= 2) {
    auto eff = [&](int ns) {
      const double r = (double)QB * ns / 256.0;
      return r / std::ceil(r);
    };
    int best = nsplit;
    double best_score = eff(nsplit);
    // (splits down to 200 n-tiles -- 256 before round 5: the crosslingual index, 1002 tiles x 129 query blocks, stopped at 2 splits = 258
    // workgroups = one full round + one of 2 workgroups, and its 32 candidates per query left 3,850 of 16,491 crowded random-init
    // queries to the fp32 second chance: 1.60 ms per pass; 4 splits: 1.17 ms, no second chance; 8 splits 1.18; on well-spread
    // vectors 0.97 / 1.07 / 1.38 ms -- tools/bench_c3.py, profiles/r05_notes.txt)
    static const int min_tiles = getenv("SSE_SPLIT_MIN_TILES") ? atoi(getenv("SSE_SPLIT_MIN_TILES")) : 200;  // (env: measurement aid)
    for (int ns = nsplit * 2, d = 1; ns <= 128 && NT / ns >= min_tiles; ns *= 2, ++d) {
      const double sc = eff(ns) - 0.01 * d;
      if (sc > best_score + 0.01) {
        best = ns;
        best_score = sc;
      }
    }
    nsplit = best;
  }
  return nsplit;
}

// diagnostic counters on the device: [0] queries whose bf16-candidate result missed its certificate, [1] queries served
// by the collect path, [2] queries that fell through to the float64 brute force
static int ensure_counters(sse_handle *h, hipStream_t st) {
  if (reserve(h, h->s_fb_cnt, 4 * sizeof(unsigned long long))) return 1;
  if (!h->fb_cnt_init) {
    HIPCHECK(h, hipMemsetAsync(h->s_fb_cnt.p, 0, 4 * sizeof(unsigned long long), st));
    h->fb_cnt_init = true;
  }
  return 0;
}

// collect-path scratch: thresholds, slots, counters (+1: the slot allocator), row buffers
static int reserve_collect(sse_handle *h, int Q, int slots) {
  if (reserve(h, h->s_cthr, (size_t)Q * sizeof(float))) return 1;
  if (reserve(h, h->s_cslot, (size_t)Q * sizeof(int32_t))) return 1;
  if (reserve(h, h->s_ccnt, (size_t)(slots + 1) * sizeof(int32_t))) return 1;
  if (reserve(h, h->s_cbuf, (size_t)slots * SSE_COLLECT_CAP * sizeof(int32_t))) return 1;
  return 0;
}

// k > 16 (sse_demo.py:128-134 and webserver.py take a user-chosen nbest): candidate sweep with enough per-split
// lists that the k-th best candidate is a tight lower bound of the k-th best row, then a collect sweep of every row
// that can still be in the exact top-k and a float64 sort of those -- two grid-wide sweeps instead of the
// one-workgroup-per-query float64 brute force (which stays as the fallback: k > SSE_MAX_SELECT_K, or more than
// SSE_COLLECT_CAP rows within the bound of the k-th score).
static int score_select_locked(sse_handle *h, const float *q, int Q, int k, double *out_s, int64_t *out_i, hipStream_t st) {
  const int S = h->idx_S, KG = (S + 7) / 8;
  const int64_t NT = (h->idx_N + 31) / 32;
  const int POOL = 2048;
  const float eps32 = (float)(2.0 * (S + 2) * 5.97e-8 * h->idx_norm_max);
  if (ensure_counters(h, st)) return 1;
  unsigned long long *counters = (unsigned long long *)h->s_fb_cnt.p;
  for (int q0 = 0; q0 < Q; q0 += POOL) {
    const int Qc = std::min(POOL, Q - q0);
    const float *qc = q + (size_t)q0 * S;
    const int QT = (Qc + 31) / 32;
    const int NQ = score_pick_nq(Qc, S, 0);
    if (NQ == 0) return fail(h, "index dimension %d does not fit the scoring kernel's LDS query block", S);
    const int QB = (QT + NQ - 1) / NQ;
    int nsplit = choose_nsplit(NQ, QB, NT);
    // at least 4k candidates (k-th candidate close to the k-th row), at most RS_MAXNC = 4096, >= 2 tiles per split
    const int max_split = (NQ == 1) ? 256 : 128;
    while (nsplit < max_split && nsplit * 16 < 4 * k && NT / (nsplit * 2) >= 2) nsplit *= 2;
    const int NC = nsplit * 16;
    if (reserve(h, h->s_qp, (size_t)QB * NQ * KG * 256 * sizeof(float))) return 1;
    if (reserve(h, h->s_ps, (size_t)Qc * NC * sizeof(float))) return 1;
    if (reserve(h, h->s_pi, (size_t)Qc * NC * sizeof(int32_t))) return 1;
    if (reserve(h, h->s_pb, (size_t)Qc * nsplit * sizeof(float))) return 1;
    if (reserve(h, h->s_cert, (size_t)Qc * sizeof(int32_t))) return 1;
    if (reserve_collect(h, Qc, Qc)) return 1;
    HIPCHECK(h, launch_pack_rows(qc, Qc, S, (float *)h->s_qp.p, st));
    ScoreArgs a;
    a.idxp = h->idxp;
    a.qp = (const float *)h->s_qp.p;
    a.part_scores = (float *)h->s_ps.p;
    a.part_ids = (int32_t *)h->s_pi.p;
    a.part_bnd = (float *)h->s_pb.p;
    a.N = h->idx_N;
    a.Q = Qc;
    a.KG = KG;
    a.NT = (int)NT;
    a.QT = QT;
    a.NSPLIT = nsplit;
    a.KC = 16;
    a.NQ = NQ;
    HIPCHECK(h, launch_score_topk(a, st));
    HIPCHECK(h, launch_kth_bound(qc, a.part_scores, a.part_ids, Qc, S, NC, k, eps32, (float *)h->s_cthr.p,
                                 (int32_t *)h->s_cslot.p, st));
    HIPCHECK(h, hipMemsetAsync(h->s_ccnt.p, 0, (size_t)(Qc + 1) * sizeof(int32_t), st));
    HIPCHECK(h, hipMemsetAsync(h->s_cert.p, 0, (size_t)Qc * sizeof(int32_t), st));
    ScoreArgs c = a;
    c.COLLECT = 1;
    c.col_thr = (const float *)h->s_cthr.p;
    c.col_slot = (const int32_t *)h->s_cslot.p;
    c.col_cnt = (int32_t *)h->s_ccnt.p;
    c.col_buf = (int32_t *)h->s_cbuf.p;
    c.col_cap = SSE_COLLECT_CAP;
    HIPCHECK(h, launch_score_topk(c, st));
    SelectArgs sel{qc, h->idxp, h->idx64, c.col_slot, c.col_cnt, c.col_buf, SSE_COLLECT_CAP,
                   out_s + (size_t)q0 * k, out_i + (size_t)q0 * k, (int32_t *)h->s_cert.p, h->idx_base, Qc, S, k, counters + 1};
    HIPCHECK(h, launch_select_topk(sel, st));
    // whatever overflowed its buffer: float64 brute force (pages of 16)
    HIPCHECK(h, launch_exact_topk(qc, h->idxp, h->idx64, (const int32_t *)h->s_cert.p, out_s + (size_t)q0 * k,
                                  out_i + (size_t)q0 * k, h->idx_base, h->idx_N, Qc, S, k, st, counters + 2));
  }
  return 0;
}

// phase: SCORE_ALL queues every stage (the asynchronous *_dev entry point: no host sync, the follow-up stages return at once
// when every query is certified).  The host-buffer entry points split the call: SCORE_FIRST = candidate sweep + float64
// re-scoring (results and certificates final for every certified query), then -- only if the certificates they read back
// with the results say so -- SCORE_REST = second chance / collect / select / brute force: the common call saves five to nine
// empty launches (~4.5 us each: a third of a single-query call, profiles/r02z_demo_kernel_stats.csv).
enum { SCORE_ALL = 0, SCORE_FIRST = 1, SCORE_REST = 2 };
// Host mirror of a SCORE_FIRST call of few queries (see RescoreArgs): pinned, device-visible host pointers the re-scoring pass
// stores results, certificates, the error flag of the encoder in front of it and -- last, released at system scope -- one
// completion word per query through.  The host polls the words: no read-back copies, no stream synchronisation.
struct ScoreMirror {
  double *scores;
  int64_t *ids;
  int32_t *cert, *err, *flag;
  int32_t seq;
};
// *split (out, may be null): 1 when the call has a FIRST / REST split (k <= 16), 0 when SCORE_FIRST did everything
int score_dev_locked(sse_handle *h, const float *q, int Q, int k, double *out_s, int64_t *out_i, hipStream_t st,
                     int phase = SCORE_ALL, int *split = nullptr, const ScoreMirror *mirror = nullptr) {
  if (split) *split = 0;
  if (!h->idxp) return fail(h, "no index uploaded");
  if (Q < 0) return fail(h, "bad Q");
  if (Q == 0) return 0;
  if (k < 1 || k > h->idx_N) return fail(h, "k=%d must be in [1, N=%lld]", k, (long long)h->idx_N);
  if ((k > 16) && phase == SCORE_REST) return 0;
  if (k > SSE_MAX_SELECT_K) {
    // beyond the collect path's buffers: exact float64 paging (correct for any k <= N, one workgroup per query)
    HIPCHECK(h, launch_exact_topk(q, h->idxp, h->idx64, nullptr, out_s, out_i, h->idx_base, h->idx_N, Q, h->idx_S, k, st));
    return 0;
  }
  if (k > 16) return score_select_locked(h, q, Q, k, out_s, out_i, st);
  const int S = h->idx_S, KG = (S + 7) / 8;
  const int QT = (Q + 31) / 32;
  // bf16 candidate pass (option score_bf16): 16x the matrix rate for the 128-query-block variant, half the index
  // bytes for the HBM-bound few-queries sweep
  // (small indexes stay on the fp32 pass: nothing to win, and no bf16 copy / second-chance launches to pay for)
  const bool bf = h->score_bf16 && h->idx_N >= 8192;
  // <= 32 queries (demo / web): single query tile, HBM-bound sweep; else 128-query blocks while they fit LDS (index
  // dimension <= 296), 64-query blocks up to 616 (configs[4]: 512), 32-query blocks beyond
  const int NQ = score_pick_nq(Q, S, bf ? 1 : 0);
  if (NQ == 0) return fail(h, "index dimension %d does not fit the scoring kernel's LDS query block", S);
  const int QB = (QT + NQ - 1) / NQ;
  const int64_t NT = (h->idx_N + 31) / 32;
  // the 16 lane lists of a workgroup are always merged in-kernel (a few tens of microseconds per workgroup): the
  // re-scoring pass ranks 16 candidates per split
  const int nsplit = choose_nsplit(NQ, QB, NT);
  // the fp32 second chance of the bf16 pass sweeps for a dense set of FEW queries (typically a handful of query blocks): it
  // takes its parallelism from the index instead -- up to 128 splits -- or three workgroups would walk 1.25 M rows alone
  int nsplit2 = nsplit;
  // (up to 128 since round 6: with 32, the 30 uncertified of 16384 random queries against 1.25 M rows -- one query block -- kept 32
  // workgroups busy for 4 ms on an otherwise idle chip: 13.5 ms per pass, 9.75 with 128; env: measurement aid)
  static const int split2_max = getenv("SSE_SPLIT2_MAX") ? atoi(getenv("SSE_SPLIT2_MAX")) : 128;
  while (nsplit2 < split2_max && NT / (nsplit2 * 2 * 2) >= 16) nsplit2 *= 2;
  // many queries against a small index (the evaluator's 16384 x 571): one launch forms all N scores per query and selects the
  // 16 best exactly (launch_score_small_index) instead of the list sweep
  const bool small_idx = h->score_small_index && !bf && score_small_index_applies(Q, KG, NT);
  const int NC = small_idx ? 16 : nsplit * 16, NCmax = std::max(nsplit, nsplit2) * 16;
  const int KG16 = (S + 15) / 16;
  if (bf && !h->idxp16_valid) {
    const size_t need = (size_t)NT * KG16 * 1024;
    if (need > h->idxp16_cap) {
      if (h->idxp16) HIPCHECK(h, hipFree(h->idxp16));
      h->idxp16 = nullptr;
      h->idxp16_cap = 0;
      HIPCHECK(h, hipMalloc(&h->idxp16, need));
      h->idxp16_cap = need;
    }
    HIPCHECK(h, launch_frag32_to_bf16(h->idxp, NT, KG, h->idxp16, st));
    h->idxp16_valid = true;
  }
  if (ensure_counters(h, st)) return 1;
  unsigned long long *counters = (unsigned long long *)h->s_fb_cnt.p;
  // ---- mid-size indexes under many queries: max-only sweep -> per-query threshold -> collect sweep -> float64 select (see
  // lane_max_threshold_kernel in score_topk.hip for the argument).  Everything is final after it: no certificates to follow up.
  if (bf && NQ == 4 && Q >= 1024 && h->idx_N >= h->score_two_pass_min_rows && h->idx_N <= h->score_two_pass_rows && !mirror) {
    // Splits of the two list-free sweeps: the list sweep's choice weighs list warm-up against rounds; without lists only the rounds
    // count -- workgroups are one per CU (VGPRs), so QB x splits should sit just below a multiple of the CU count: 129 query blocks x 4
    // splits = 2.02 rounds run as 3, x 16 = 8.06 run as 9.  At most 16 (the threshold kernel takes 16 x 16 lane maxima per query),
    // at least 32 tiles per split.
    int ns2 = std::min(nsplit, 16);
    {
      static const int force = getenv("SSE_TWO_PASS_SPLITS") ? atoi(getenv("SSE_TWO_PASS_SPLITS")) : 0;  // (env: measurement aid)
      const int cus = h->cu_count > 0 ? h->cu_count : 256;
      double best = 0.0;
      for (int ns = 1; ns <= 16; ns *= 2) {
        if (ns > 1 && NT / ns < 32) break;
        const double r = (double)QB * ns / cus, eff = r / std::ceil(r);
        if (eff > best + 0.02) {
          best = eff;
          ns2 = ns;
        }
      }
      if (force == 1 || force == 2 || force == 4 || force == 8 || force == 16) ns2 = force;
    }
    if (phase == SCORE_REST) return 0;
    h->two_pass_calls += 1;
    constexpr int MID_CAP = 512;  // rows one query may collect (typically 20 - 30); more: float64 brute force for that query
    const float eps32 = (float)(2.0 * (S + 2) * 5.97e-8 * h->idx_norm_max);
    const float eps_bf = eps32 + (float)(1.02 * (1.0 / 256.0 + 1.0 / 262144.0) * h->idx_norm_max);
    if (reserve(h, h->s_qp, (size_t)QB * NQ * KG * 256 * sizeof(float))) return 1;
    if (reserve(h, h->s_lmax, (size_t)Q * ns2 * 16 * sizeof(float))) return 1;
    if (reserve(h, h->s_cert, (size_t)Q * sizeof(int32_t))) return 1;
    if (reserve(h, h->s_cthr, (size_t)Q * sizeof(float)) || reserve(h, h->s_cslot, (size_t)Q * sizeof(int32_t)) ||
        reserve(h, h->s_ccnt, (size_t)(Q + 1) * sizeof(int32_t)) || reserve(h, h->s_cbuf, (size_t)Q * MID_CAP * sizeof(int32_t)))
      return 1;
    HIPCHECK(h, launch_pack_rows_bf16(q, Q, S, h->s_qp.p, st));
    HIPCHECK(h, hipMemsetAsync(h->s_ccnt.p, 0, (size_t)(Q + 1) * sizeof(int32_t), st));
    HIPCHECK(h, hipMemsetAsync(h->s_cert.p, 0, (size_t)Q * sizeof(int32_t), st));
    ScoreArgs m1;
    m1.BF = 1;
    m1.COLLECT = 1;
    m1.idxp = (const float *)h->idxp16;
    m1.qp = (const float *)h->s_qp.p;
    m1.N = h->idx_N;
    m1.Q = Q;
    m1.KG = KG16;
    m1.NT = (int)NT;
    m1.QT = QT;
    m1.NSPLIT = ns2;
    m1.KC = 16;
    m1.NQ = NQ;
    m1.lane_max = (float *)h->s_lmax.p;
    HIPCHECK(h, launch_score_topk(m1, st));
    HIPCHECK(h, launch_lane_max_threshold(q, (const float *)h->s_lmax.p, Q, S, ns2 * 16, eps_bf, (float *)h->s_cthr.p,
                                          (int32_t *)h->s_cslot.p, st));
    ScoreArgs m2 = m1;
    m2.lane_max = nullptr;
    m2.col_thr = (const float *)h->s_cthr.p;
    m2.col_slot = (const int32_t *)h->s_cslot.p;
    m2.col_cnt = (int32_t *)h->s_ccnt.p;
    m2.col_buf = (int32_t *)h->s_cbuf.p;
    m2.col_cap = MID_CAP;
    HIPCHECK(h, launch_score_topk(m2, st));
    SelectArgs sel{q, h->idxp, h->idx64, m2.col_slot, m2.col_cnt, m2.col_buf, MID_CAP, out_s, out_i, (int32_t *)h->s_cert.p, h->idx_base,
                   Q, S, k, nullptr};
    HIPCHECK(h, launch_select_topk(sel, st));
    // a query whose buffer overflowed (or that saw fewer than k rows): float64 brute force
    HIPCHECK(h, launch_exact_topk(q, h->idxp, h->idx64, (const int32_t *)h->s_cert.p, out_s, out_i, h->idx_base, h->idx_N, Q, S, k, st,
                                  counters + 2));
    if (split) *split = 0;  // nothing is left for a SCORE_REST call
    return 0;
  }
  const int POOL = std::min(Q, 1024);  // collect-buffer slots for uncertified queries (the rest: float64 brute force)
  if (reserve(h, h->s_qp, (size_t)QB * NQ * KG * 256 * sizeof(float))) return 1;
  if (reserve(h, h->s_ps, (size_t)Q * (bf ? NCmax : NC) * sizeof(float))) return 1;
  if (reserve(h, h->s_pi, (size_t)Q * (bf ? NCmax : NC) * sizeof(int32_t))) return 1;
  if (reserve(h, h->s_pb, (size_t)Q * (bf ? NCmax / 16 : nsplit) * sizeof(float))) return 1;
  if (reserve(h, h->s_cert, (size_t)Q * sizeof(int32_t))) return 1;
  if (reserve_collect(h, Q, POOL)) return 1;
  if (split) *split = 1;
  const bool first = phase != SCORE_REST, rest = phase != SCORE_FIRST;
  // <= 32 queries (the latency path): the sweeps build their query fragments from the rows themselves -- no pack launch
  const bool rows_direct = NQ == 1;
  if (first && !rows_direct && !small_idx) {
    if (bf) HIPCHECK(h, launch_pack_rows_bf16(q, Q, S, h->s_qp.p, st));
    else HIPCHECK(h, launch_pack_rows(q, Q, S, (float *)h->s_qp.p, st));
  }
  ScoreArgs a;
  if (rows_direct) {
    a.q_rows = q;
    a.S = S;
  }
  a.BF = bf ? 1 : 0;
  a.idxp = bf ? (const float *)h->idxp16 : h->idxp;
  a.qp = (const float *)h->s_qp.p;
  a.part_scores = (float *)h->s_ps.p;
  a.part_ids = (int32_t *)h->s_pi.p;
  a.part_bnd = (float *)h->s_pb.p;
  a.N = h->idx_N;
  a.Q = Q;
  a.KG = bf ? KG16 : KG;
  a.NT = (int)NT;
  a.QT = QT;
  a.NSPLIT = nsplit;
  a.KC = 16;
  a.NQ = NQ;
  if (first) {
    if (small_idx) {
      SmallIndexArgs si{q, h->idxp, a.part_scores, a.part_ids, a.part_bnd, h->idx_N, Q, S, KG, (int)NT};
      HIPCHECK(h, launch_score_small_index(si, st));
    } else {
      HIPCHECK(h, launch_score_topk(a, st));
    }
  }
  RescoreArgs r;
  r.q = q;
  r.idx32 = h->idxp;
  r.idx64 = h->idx64;
  r.idx_rm = h->idx_rm_valid ? (const float *)h->idx_rm.p : nullptr;
  r.part_scores = a.part_scores;
  r.part_ids = a.part_ids;
  r.part_bnd = a.part_bnd;
  r.out_scores = out_s;
  r.out_ids = out_i;
  r.cert = (int32_t *)h->s_cert.p;
  r.id_base = h->idx_base;
  r.N = h->idx_N;
  r.Q = Q;
  r.S = S;
  r.NC = NC;
  r.k = k;
  // |fp32 fma-chain dot - exact| <= S * 2^-24 * |q||t| (+ the f32 rounding of f64 rows); use 2x margin
  const float eps32 = (float)(2.0 * (S + 2) * 5.97e-8 * h->idx_norm_max);
  r.eps = eps32;
  r.eps32 = eps32;
  r.col_thr = (float *)h->s_cthr.p;
  // every query has a collect slot of its own (the usual call): the re-scoring pass hands them out (slot = query for an
  // uncertified one) and zeroes their counters -- no memset / launch_assign_slots between the passes
  const bool own_slots = Q <= POOL;
  if (own_slots) {
    r.col_slot = (int32_t *)h->s_cslot.p;
    r.col_cnt = (int32_t *)h->s_ccnt.p;
  }
  // bf16 operands: |q^.t^ - q.t| <= ((1+u)^2 - 1) sum|q_i t_i| <= (2^-8 + 2^-18) |q||t|, u = 2^-9 (round to nearest)
  if (bf) r.eps += (float)(1.02 * (1.0 / 256.0 + 1.0 / 262144.0) * h->idx_norm_max);
  if (first) {
    RescoreArgs rf = r;
    if (mirror) {
      rf.host_scores = mirror->scores;
      rf.host_ids = mirror->ids;
      rf.host_cert = mirror->cert;
      rf.host_err = mirror->err;
      rf.host_flag = mirror->flag;
      rf.err_in = h->err_flag;
      rf.seq = mirror->seq;
    }
    HIPCHECK(h, launch_rescore(rf, st));
  }
  if (!rest) return 0;
  const float *qp32 = (const float *)h->s_qp.p;  // fp32 query fragments for the collect sweep
  // (<= 32 queries skip the second chance: for one query block it is the same fp32 sweep of the whole index as the collect
  // pass, which is final)
  if (bf && Q > 32) {
    // Second chance, entirely on the device (the call stays asynchronous): queries whose bf16-candidate result missed
    // its certificate (top scores packed closer than the bf16 bound) are swept again with fp32 candidates -- the same
    // kernels, where a workgroup whose whole query block is certified returns at once and a certified query is left
    // alone -- before anything falls through to the collect path.
    if (reserve(h, h->s_qp32, (size_t)QB * NQ * KG * 256 * sizeof(float))) return 1;
    if (reserve(h, h->s_qmap, (size_t)(Q + 1) * sizeof(int32_t))) return 1;
    if (reserve(h, h->s_qc, (size_t)QT * 32 * S * sizeof(float))) return 1;
    HIPCHECK(h, launch_count_uncert(r.cert, Q, counters, st));
    // the uncertified queries as a dense set: the sweep and the re-scoring below cost what THEY cost (launches sized for all
    // Q; workgroups past the set return at once)
    int32_t *qmap = (int32_t *)h->s_qmap.p, *qcount = qmap + Q;
    HIPCHECK(h, launch_compact_uncert(q, r.cert, Q, S, qmap, qcount, (float *)h->s_qc.p, st));
    if (!rows_direct) HIPCHECK(h, launch_pack_rows((const float *)h->s_qc.p, Q, S, (float *)h->s_qp32.p, st));
    ScoreArgs a2 = a;
    a2.BF = 0;
    a2.idxp = h->idxp;
    a2.qp = (const float *)h->s_qp32.p;
    if (rows_direct) a2.q_rows = (const float *)h->s_qc.p;
    a2.KG = KG;
    a2.q_count = qcount;
    a2.NSPLIT = nsplit2;
    HIPCHECK(h, launch_score_topk(a2, st));
    RescoreArgs r2 = r;
    r2.q = (const float *)h->s_qc.p;
    r2.NC = nsplit2 * 16;
    r2.eps = eps32;
    r2.qmap = qmap;
    r2.q_count = qcount;
    HIPCHECK(h, launch_rescore(r2, st));
    qp32 = (const float *)h->s_qp32.p;  // (unused by the collect sweep: it reads the rows)
  }
  // What is still uncertified has its k-th score tied with (or within the fp32 bound of) rows outside the candidate
  // lists -- e.g. an index holding many exact duplicates of a query's best rows.  Collect path: every row whose fp32
  // score reaches (exact k-th of the candidates) - bound is gathered by a grid-wide sweep and sorted in float64:
  // provably the exact top-k, at the cost of one more fp32 sweep for the query blocks concerned.  Launched
  // unconditionally (no host sync); with everything certified all workgroups return at once.
  if (!own_slots) {
    HIPCHECK(h, hipMemsetAsync(h->s_ccnt.p, 0, (size_t)(POOL + 1) * sizeof(int32_t), st));
    HIPCHECK(h, launch_assign_slots(r.cert, Q, POOL, (int32_t *)h->s_cslot.p, (int32_t *)h->s_ccnt.p + POOL, st));
  }
  ScoreArgs c = a;
  c.BF = 0;
  c.idxp = h->idxp;
  c.qp = qp32;
  c.q_rows = q;  // (the collect sweep builds its fp32 fragments from the rows: nothing packed for it)
  c.S = S;
  c.KG = KG;
  c.COLLECT = 1;
  c.col_thr = (const float *)h->s_cthr.p;
  c.col_slot = (const int32_t *)h->s_cslot.p;
  c.col_cnt = (int32_t *)h->s_ccnt.p;
  c.col_buf = (int32_t *)h->s_cbuf.p;
  c.col_cap = SSE_COLLECT_CAP;
  HIPCHECK(h, launch_score_topk(c, st));
  SelectArgs sel{q, h->idxp, h->idx64, c.col_slot, c.col_cnt, c.col_buf, SSE_COLLECT_CAP, out_s, out_i, r.cert, h->idx_base, Q, S, k,
                 counters + 1};
  HIPCHECK(h, launch_select_topk(sel, st));
  // last resort (more than SSE_COLLECT_CAP rows within the bound of the k-th score, or no buffer slot left)
  HIPCHECK(h, launch_exact_topk(q, h->idxp, h->idx64, r.cert, out_s, out_i, h->idx_base, h->idx_N, Q, S, k, st, counters + 2));
  return 0;
}

}  // namespace

extern "C" {

const char *sse_last_error(sse_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sse_create(const sse_config *cfg, sse_handle **out) {
  if (!cfg || !out) return fail(nullptr, "sse_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, "no HIP device available");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "device %d out of range (have %d)", cfg->device, ndev);
  if (cfg->vocab_size < 2 || cfg->embedding_size < 1 || cfg->encoding_size < 1)
    return fail(nullptr, "bad model sizes V=%d E=%d S=%d", cfg->vocab_size, cfg->embedding_size, cfg->encoding_size);
  sse_handle *h = new sse_handle();
  h->cfg = *cfg;
  h->lr = cfg->learning_rate;
#define CREATE_FAIL(...)        \
  do {                          \
    fail(nullptr, __VA_ARGS__); \
    sse_destroy(h);             \
    return 1;                   \
  } while (0)
  if (hipSetDevice(cfg->device) != hipSuccess) CREATE_FAIL("hipSetDevice(%d) failed", cfg->device);
  add_var(h, "word_embedding", cfg->vocab_size, cfg->embedding_size);
  switch (cfg->network_mode) {
    case SSE_MODE_DUAL_ENCODER:
      setup_encoder(h, h->enc[0], "source_encoder", "source_encoder/src_M", cfg->src_cell_size, cfg->src_cell_size);
      setup_encoder(h, h->enc[1], "target_encoder", "target_encoder/tgt_M", cfg->tgt_cell_size, cfg->tgt_cell_size);
      break;
    case SSE_MODE_SHARED_ENCODER:
      setup_encoder(h, h->enc[0], "shared_encoder", "shared_encoder/src_M", cfg->src_cell_size, cfg->src_cell_size);
      h->enc[1] = h->enc[0];
      h->enc[1].proj = add_var(h, "shared_encoder/tgt_M", cfg->src_cell_size, cfg->encoding_size);
      h->enc[1].shares_lstm_with = 0;
      break;
    case SSE_MODE_SOURCE_ENCODER_ONLY:
      setup_encoder(h, h->enc[0], "source_only_encoder", "source_only_encoder/src_M", cfg->src_cell_size, cfg->src_cell_size);
      h->tgt_table = add_var(h, "target_embedding/tgt_seq_embedding", cfg->target_space_size, cfg->encoding_size);
      break;
    case SSE_MODE_SOURCE_ONLY_CNN: {
      static const int fs[4] = {2, 3, 4, 5}, nf[4] = {256, 128, 128, 64};
      for (int i = 0; i < 4; ++i) {
        char nm[96];
        snprintf(nm, sizeof nm, "source_only_cnn/conv-maxpool-%d/W", fs[i]);
        h->cnn_W[i] = add_var(h, nm, fs[i] * cfg->embedding_size, nf[i]);
        snprintf(nm, sizeof nm, "source_only_cnn/conv-maxpool-%d/b", fs[i]);
        h->cnn_b[i] = add_var(h, nm, 1, nf[i]);
      }
      h->cnn_M = add_var(h, "source_only_cnn/src_M", 576, cfg->encoding_size);
      h->tgt_table = add_var(h, "target_embedding/tgt_seq_embedding", cfg->target_space_size, cfg->encoding_size);
      break;
    }
    default:
      CREATE_FAIL("Unsupported network mode %d (sse_model.py:175-177)", cfg->network_mode);
  }
  for (int s = 0; s < 2; ++s)
    if (h->enc[s].kernel >= 0 && geometry(h, h->enc[s])) {
      g_create_error = h->err;
      sse_destroy(h);
      return 1;
    }
  for (auto &v : h->vars) {
    if (hipMalloc((void **)&v.dev, v.count * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&v.slot, v.count * sizeof(float)) != hipSuccess)
      CREATE_FAIL("hipMalloc failed for variable %s", v.name.c_str());
    hipMemset(v.dev, 0, v.count * sizeof(float));
    launch_fill(v.slot, v.count, 0.1f, nullptr);  // AdagradOptimizer initial_accumulator_value
  }
  if (hipMalloc((void **)&h->err_flag, sizeof(int32_t)) != hipSuccess) CREATE_FAIL("hipMalloc failed");
  if (hipHostMalloc((void **)&h->pin_small, 64 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) CREATE_FAIL("hipHostMalloc failed");
  hipMemset(h->err_flag, 0, sizeof(int32_t));
  if (hipHostMalloc((void **)&h->pad_stat, 8 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) CREATE_FAIL("hipHostMalloc failed");
  {  // compute units of the device: the tile policy of the matrix LSTM kernel counts rounds of them, the cluster kernels check co-residency
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) h->cu_count = prop.multiProcessorCount;
  }
  if (const char *ev = getenv("SSE_PAD_SORT_DEV")) {  // measurement aid: the option's initial value
    const int v = atoi(ev);
    if (v >= 0 && v <= 2) h->pad_sort_dev = v;
  }
  memset(h->pad_stat, 0, 8 * sizeof(int32_t));
  if (hipDeviceSynchronize() != hipSuccess) CREATE_FAIL("device initialisation failed");
#undef CREATE_FAIL
  *out = h;
  return 0;
}

void sse_destroy(sse_handle *h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  hipDeviceSynchronize();
  for (auto &v : h->vars) {
    if (v.dev) hipFree(v.dev);
    if (v.slot) hipFree(v.slot);
  }
  if (h->train && h->train->arena && !h->train->arena_external) hipFree(h->train->arena);
  for (int s = 0; s < 2; ++s) {
    Encoder &e = h->enc[s];
    if (e.shares_lstm_with < 0) {
      if (e.Wp) hipFree(e.Wp);
      if (e.pad_h) (void)hipFree(e.pad_h);
      if (e.pad_c) (void)hipFree(e.pad_c);
      if (e.Waug) (void)hipFree(e.Waug);
      if (e.Wc) (void)hipFree(e.Wc);
      if (e.Wx3) (void)hipFree(e.Wx3);
      if (e.Wx3t) (void)hipFree(e.Wx3t);
      if (e.pad_h_x3) (void)hipFree(e.pad_h_x3);
      if (e.pad_c_x3) (void)hipFree(e.pad_c_x3);
      if (e.pad_h_small) (void)hipFree(e.pad_h_small);
      if (e.pad_c_small) (void)hipFree(e.pad_c_small);
    }
    if (e.Mp) hipFree(e.Mp);
  }
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->pin_small) (void)hipHostFree(h->pin_small);
  if (h->pad_stat) (void)hipHostFree(h->pad_stat);
  if (h->emb_pad) hipFree(h->emb_pad);
  if (h->emb16) (void)hipFree(h->emb16);
  if (h->err_flag) hipFree(h->err_flag);
  if (h->idxp) hipFree(h->idxp);
  if (h->idxp16) (void)hipFree(h->idxp16);
  if (h->idx64) hipFree(h->idx64);
  if (h->cnn_Wc) (void)hipFree(h->cnn_Wc);
  if (h->emb_bf16) (void)hipFree(h->emb_bf16);
  if (h->cnn_Wc16) (void)hipFree(h->cnn_Wc16);
  if (h->cnn_Mx3) (void)hipFree(h->cnn_Mx3);
  if (h->cnn_bias) (void)hipFree(h->cnn_bias);
  if (h->cnn_Mp) (void)hipFree(h->cnn_Mp);
  if (h->train) {
    TrainState *t = h->train;
    for (int s = 0; s < 2; ++s) {
      if (t->side[s]) (void)hipStreamDestroy(t->side[s]);
      if (t->ev_join[s]) (void)hipEventDestroy(t->ev_join[s]);
      if (t->KhT[s] && (s == 0 || t->KhT[s] != t->KhT[0])) (void)hipFree(t->KhT[s]);
      if (t->KxT[s] && (s == 0 || t->KxT[s] != t->KxT[0])) (void)hipFree(t->KxT[s]);
      if (t->KhT16[s] && (s == 0 || t->KhT16[s] != t->KhT16[0])) (void)hipFree(t->KhT16[s]);
      if (t->KxT16[s] && (s == 0 || t->KxT16[s] != t->KxT16[0])) (void)hipFree(t->KxT16[s]);
    }
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    delete t;
  }
  for (hipEvent_t e : h->events)
    if (e) (void)hipEventDestroy(e);
  delete h;
}

int sse_num_variables(sse_handle *h) { return h ? (int)h->vars.size() : 0; }

int sse_variable_info(sse_handle *h, int index, const char **name, int64_t *count, int32_t *rows, int32_t *cols) {
  if (!h) return 1;
  if (index < 0 || index >= (int)h->vars.size()) return fail(h, "variable index %d out of range", index);
  const Variable &v = h->vars[index];
  if (name) *name = v.name.c_str();
  if (count) *count = v.count;
  if (rows) *rows = v.rows;
  if (cols) *cols = v.cols;
  return 0;
}

int sse_set_variable(sse_handle *h, const char *name, const float *host, int64_t count) {
  if (!h || !name || !host) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  bool slot;
  const int i = find_var(h, name, &slot);
  if (i < 0) return fail(h, "unknown variable '%s'", name);
  if (count != h->vars[i].count) return fail(h, "variable '%s' has %lld elements, got %lld", name, (long long)h->vars[i].count, (long long)count);
  HIPCHECK(h, hipMemcpy(slot ? h->vars[i].slot : h->vars[i].dev, host, count * sizeof(float), hipMemcpyHostToDevice));
  if (!slot) {
    h->packed_dirty = true;
    h->weights_version += 1;
    h->mp_fresh = false;
    if (h->train) h->train->packed_dirty = h->train->fp32_dirty = true;
  }
  return 0;
}

int sse_get_variable(sse_handle *h, const char *name, float *host, int64_t count) {
  if (!h || !name || !host) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  bool slot;
  const int i = find_var(h, name, &slot);
  if (i < 0) return fail(h, "unknown variable '%s'", name);
  if (count != h->vars[i].count) return fail(h, "variable '%s' has %lld elements, got %lld", name, (long long)h->vars[i].count, (long long)count);
  HIPCHECK(h, hipDeviceSynchronize());
  HIPCHECK(h, hipMemcpy(host, slot ? h->vars[i].slot : h->vars[i].dev, count * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int sse_encode_dev(sse_handle *h, int side, const int32_t *ids_dev, int32_t B, int32_t T, int32_t normalize,
                   float *out_dev, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return encode_dev_locked(h, side, ids_dev, B, T, normalize, out_dev, (hipStream_t)stream);
}

// host ids -> encodings [B][S] in h->s_out (device); the handle mutex is held by the caller
// the handle's pinned read-back buffer, grown on demand
static int ensure_pin(sse_handle *h, size_t need) {
  if (need <= h->pin_cap) return 0;
  if (h->pin) HIPCHECK(h, hipHostFree(h->pin));
  h->pin = nullptr;
  h->pin_cap = 0;
  HIPCHECK(h, hipHostMalloc(&h->pin, need + need / 2 + 4096, hipHostMallocCoherent));  // (the device stores into it: ScoreMirror)
  h->pin_cap = need + need / 2 + 4096;
  memset(h->pin, 0, h->pin_cap);  // (the completion words of ScoreMirror must not hold a call number by accident)
  return 0;
}

// Re-run of a batch whose cluster-kernel launch reported a missing workgroup (error bit 2), on the kernels that need no
// co-residency.  The cluster kernels need their 16 - 32 workgroups per cluster resident together; a device busy with other
// work (a train step on another handle, four serving routes at once) can keep one from arriving within the give-up time.
// Nothing was written that the other kernels do not overwrite; results are bit-identical.
static int encode_fallback_locked(sse_handle *h, int side, int32_t B, int32_t T, int32_t normalize, hipStream_t st) {
  h->persist_fallbacks += 1;
  h->cluster_skip[B <= 32 ? 0 : 1] = h->cluster_backoff >= 0 ? h->cluster_backoff : 16;  // a time-out costs 10 ms: do not pay it on every call of a busy device
  const int keep = h->lstm_persist_rows, keep_c = h->lstm_cluster_rows;
  h->lstm_persist_rows = 0;
  h->lstm_cluster_rows = 0;
  const int rc = encode_dev_locked(h, side, (const int32_t *)h->s_ids.p, B, T, normalize, (float *)h->s_out.p, st);
  h->lstm_persist_rows = keep;
  h->lstm_cluster_rows = keep_c;
  if (rc) return 1;
  return check_err_flag(h, st);
}

// defer_check: leave the device error flag unread (no synchronisation here); the caller reads it with its own read-back
static int encode_host_ids_locked(sse_handle *h, int side, const int32_t *ids_host, int32_t B, int32_t T, int32_t normalize,
                                  bool defer_check = false) {
  const size_t S = h->cfg.encoding_size;
  if (reserve(h, h->s_ids, (size_t)B * T * sizeof(int32_t))) return 1;
  if (reserve(h, h->s_out, (size_t)B * S * sizeof(float))) return 1;
  hipStream_t st = nullptr;
  // Group rows by their leading-PAD count (left-padded inputs, sse_index.py:79-85) so that every
  // 64-row tile can skip its whole common PAD prefix; results are scattered back in caller order.
  const bool lstm_side = h->cfg.network_mode != SSE_MODE_SOURCE_ONLY_CNN && !(side == SSE_SIDE_TARGET && h->tgt_table >= 0);
  const int32_t *row_map_dev = nullptr;
  const bool to_cluster = lstm_side && h->enc[side].kernel >= 0 && cluster_takes(h, h->enc[side], B, T);  // (takes rows as they come)
  if (h->pad_skip && lstm_side && B > 64 && B > h->lstm_small_rows && !to_cluster && !h->enc[side].generic) {
    // counting sort of the row numbers by leading-PAD count, shortest prefix first (the tiles with the most steps left are
    // dispatched first; see pack.hip, pad_lead_kernel)
    std::vector<int32_t> lead(B), start(T + 2, 0), order(B);
    for (int b = 0; b < B; ++b) {
      const int32_t *row = ids_host + (size_t)b * T;
      int t = 0;
      while (t < T && row[t] == 0) ++t;
      lead[b] = t;
      ++start[t + 1];
    }
    for (int i = 1; i <= T + 1; ++i) start[i] += start[i - 1];
    for (int b = 0; b < B; ++b) order[start[lead[b]]++] = b;
    int64_t lead_sum = 0;
    for (int b = 0; b < B; ++b) lead_sum += lead[b];
    h->cur_padded_hint = lead_sum * 4 >= (int64_t)B * T;
    if (reserve(h, h->s_map, (size_t)B * sizeof(int32_t))) return 1;
    HIPCHECK(h, hipMemcpyAsync(h->s_map.p, order.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(h, hipStreamSynchronize(st));  // `order` is a local
    row_map_dev = (const int32_t *)h->s_map.p;
  }
  HIPCHECK(h, hipMemcpyAsync(h->s_ids.p, ids_host, (size_t)B * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
  h->cur_row_map = row_map_dev;
  int rc = encode_dev_locked(h, side, (const int32_t *)h->s_ids.p, B, T, normalize, (float *)h->s_out.p, st);
  h->cur_row_map = nullptr;
  h->cur_padded_hint = false;
  if (rc) return 1;
  if (defer_check) return 0;
  int32_t bits = 0;
  if (check_err_flag(h, st, &bits)) return 1;
  if (bits == 4) return encode_fallback_locked(h, side, B, T, normalize, st);
  return 0;
}

int sse_encode(sse_handle *h, int side, const int32_t *ids_host, int32_t B, int32_t T, int32_t normalize,
               float *out_host) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (B == 0) return 0;
  if (B < 0 || T < 1 || !ids_host || !out_host) return fail(h, "bad arguments to sse_encode");
  const size_t nbytes = (size_t)B * h->cfg.encoding_size * sizeof(float);
  if (nbytes <= (64u << 10)) {
    // small results (a few queries): the encodings and the error flag come back in ONE pinned read-back with one
    // synchronisation, 0.155 -> 0.144 ms for a query.  (Measured the other way for larger ones: 600 x 256 floats 0.340 ->
    // 0.364 ms, 1024 rows 0.364 -> 0.440 ms -- the runtime's staged copy into a pageable target beats pinned + memcpy.)
    if (encode_host_ids_locked(h, side, ids_host, B, T, normalize, /*defer_check=*/true)) return 1;
    const size_t need = nbytes + sizeof(int32_t);
    if (ensure_pin(h, need)) return 1;
    char *pin = (char *)h->pin;
    hipStream_t st = nullptr;
    HIPCHECK(h, hipMemcpyAsync(pin, h->s_out.p, nbytes, hipMemcpyDeviceToHost, st));
    HIPCHECK(h, hipMemcpyAsync(pin + nbytes, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(h, sync_stream(st));
    int32_t bits = *(const int32_t *)(pin + nbytes);
    if (bits == 0) {
      memcpy(out_host, pin, nbytes);
      return 0;
    }
    if (check_err_flag(h, st, &bits)) return 1;  // reports what is reportable, resets the flag
    if (bits == 4 && encode_fallback_locked(h, side, B, T, normalize, st)) return 1;
  } else if (encode_host_ids_locked(h, side, ids_host, B, T, normalize)) {
    return 1;
  }
  HIPCHECK(h, hipMemcpy(out_host, h->s_out.p, nbytes, hipMemcpyDeviceToHost));
  return 0;
}

// Scores + ids of Q queries (device rows q_dev) into host buffers: first phase, ONE pinned read-back of [scores | ids |
// certificates] with one synchronisation, and the follow-up stages only when some query is not certified.
// err_bits: (optional) the device error flag rides along in the first read-back (an encoder launched just before on the
// same stream with its check deferred); when it comes back non-zero the outputs are not written.
static int score_to_host_locked(sse_handle *h, const float *q_dev, int Q, int k, double *out_scores, int64_t *out_ids,
                                int32_t *err_bits = nullptr) {
  hipStream_t st = nullptr;
  if (reserve(h, h->s_os, (size_t)Q * k * sizeof(double))) return 1;
  if (reserve(h, h->s_oi, (size_t)Q * k * sizeof(int64_t))) return 1;
  // pinned block: [scores | ids | certificates | error flag | completion words]
  const size_t nb = (size_t)Q * k * 8, need = 2 * nb + (size_t)(2 * Q + 1) * sizeof(int32_t);
  if (ensure_pin(h, need)) return 1;
  char *pin = (char *)h->pin;
  int32_t *cert = (int32_t *)(pin + 2 * nb), *flag = cert + Q, *done = flag + 1;
  int split = 0;
  // few queries (the demo / web call): the re-scoring pass stores its results through the pinned pointers itself and the
  // host polls the completion words -- three read-back copies and the stream synchronisation were ~40 us of a 0.19 ms call
  const bool use_mirror = Q <= 64 && k <= 16;
  ScoreMirror hm{(double *)pin, (int64_t *)(pin + nb), cert, flag, done, 0};
  if (use_mirror) {
    h->score_seq = (h->score_seq == INT32_MAX) ? 1 : h->score_seq + 1;
    hm.seq = h->score_seq;
    // The completion words move inside the pinned block with (Q, k), and the block is shared with earlier calls of other
    // shapes (scores, int64 ids, certificates) and with sse_encode's read-back: a stale word could equal this call's
    // sequence number.  Nothing of this handle is in flight here (the previous call returned after its last poll / sync),
    // so the host clears them before the launch.
    for (int i = 0; i < Q; ++i) __atomic_store_n(done + i, 0, __ATOMIC_RELAXED);
    __atomic_store_n(flag, 0, __ATOMIC_RELEASE);
  }
  if (score_dev_locked(h, q_dev, Q, k, (double *)h->s_os.p, (int64_t *)h->s_oi.p, st, SCORE_FIRST, &split, use_mirror ? &hm : nullptr))
    return 1;
  int pass = 0;
  if (use_mirror && split) {
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (;;) {
      bool all = true;
      for (int i = 0; i < Q && all; ++i) all = __atomic_load_n(done + i, __ATOMIC_ACQUIRE) == hm.seq;
      if (all) break;
      if (synced) return fail(h, "the re-scoring pass did not report completion");
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) {
        HIPCHECK(h, hipStreamSynchronize(st));  // (a launch that failed shows up here)
        synced = true;
      }
    }
    if (err_bits) {
      *err_bits = *flag;
      if (*flag) return 0;  // the queries were not what the caller meant: it looks at the flag first
    }
    bool open_q = false;
    for (int i = 0; i < Q && !open_q; ++i) open_q = cert[i] == 0;
    if (open_q) {
      // rare: ties / near-ties at the k-th score, or scores packed closer than the bf16 bound
      if (score_dev_locked(h, q_dev, Q, k, (double *)h->s_os.p, (int64_t *)h->s_oi.p, st, SCORE_REST)) return 1;
      pass = 1;
    } else {
      pass = 2;
    }
  }
  for (; pass < 2; ++pass) {
    HIPCHECK(h, hipMemcpyAsync(pin, h->s_os.p, nb, hipMemcpyDeviceToHost, st));
    HIPCHECK(h, hipMemcpyAsync(pin + nb, h->s_oi.p, nb, hipMemcpyDeviceToHost, st));
    if (split && pass == 0) HIPCHECK(h, hipMemcpyAsync(cert, h->s_cert.p, (size_t)Q * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (err_bits && pass == 0) HIPCHECK(h, hipMemcpyAsync(flag, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(h, sync_stream(st));
    if (err_bits && pass == 0) {
      *err_bits = *flag;
      if (*flag) return 0;  // the queries were not what the caller meant: it looks at the flag first
    }
    bool open_q = false;
    if (split && pass == 0) {
      for (int i = 0; i < Q && !open_q; ++i) open_q = cert[i] == 0;
    }
    if (!open_q) break;
    // rare: ties / near-ties at the k-th score, or scores packed closer than the bf16 bound
    if (score_dev_locked(h, q_dev, Q, k, (double *)h->s_os.p, (int64_t *)h->s_oi.p, st, SCORE_REST)) return 1;
  }
  memcpy(out_scores, pin, nb);
  memcpy(out_ids, pin + nb, nb);
  return 0;
}

int sse_encode_score_topk(sse_handle *h, int side, const int32_t *ids_host, int32_t B, int32_t T, int32_t normalize,
                          int32_t k, double *out_scores, int64_t *out_ids, float *enc_out_host) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (B == 0) return 0;
  if (B < 0 || T < 1 || !ids_host || !out_scores || !out_ids) return fail(h, "bad arguments to sse_encode_score_topk");
  if (!h->idxp) return fail(h, "no index uploaded");
  if (h->cfg.encoding_size != h->idx_S)
    return fail(h, "index dimension %d != encoding_size %d", h->idx_S, h->cfg.encoding_size);
  // One synchronisation for the whole call: the encoder's error flag (token id out of range, a cluster workgroup that did
  // not arrive) is not waited for before the scorer is launched -- it comes back with the scores.
  if (encode_host_ids_locked(h, side, ids_host, B, T, normalize, /*defer_check=*/true)) return 1;
  // the encodings never leave the device between the encoder and the scorer
  int32_t bits = 0;
  if (score_to_host_locked(h, (const float *)h->s_out.p, B, k, out_scores, out_ids, &bits)) return 1;
  if (bits) {
    if (check_err_flag(h, nullptr, &bits)) return 1;  // reports what is reportable, resets the flag
    if (bits == 4) {
      if (encode_fallback_locked(h, side, B, T, normalize, nullptr)) return 1;
      if (score_to_host_locked(h, (const float *)h->s_out.p, B, k, out_scores, out_ids)) return 1;
    }
  }
  if (enc_out_host)
    HIPCHECK(h, hipMemcpy(enc_out_host, h->s_out.p, (size_t)B * h->cfg.encoding_size * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int sse_get_counter(sse_handle *h, const char *name, int64_t *value) {
  if (!h || !name || !value) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  if (strcmp(name, "lstm_persist_fallbacks") == 0) {
    *value = h->persist_fallbacks;
    return 0;
  }
  if (strcmp(name, "score_two_pass_calls") == 0) {  // sse_score_topk* calls ranked by the two-pass path for mid-size indexes
    *value = h->two_pass_calls;
    return 0;
  }
  if (strcmp(name, "pad_sorted_calls") == 0) {  // encodes of device-resident ids whose rows were bucketed by PAD prefix on the device
    *value = h->pad_sorted_calls;
    return 0;
  }
  if (strcmp(name, "lstm_coop_refused") == 0) {  // process-wide: cooperative launches refused by the runtime (plain launch taken)
    *value = (int64_t)lstm_coop_refused();
    return 0;
  }
  static const char *const names[3] = {"score_bf16_second_chance_queries", "score_collect_queries", "score_bruteforce_queries"};
  for (int i = 0; i < 3; ++i) {
    if (strcmp(name, names[i]) != 0) continue;
    unsigned long long v[4] = {0, 0, 0, 0};
    if (h->s_fb_cnt.p && h->fb_cnt_init) {
      HIPCHECK(h, hipSetDevice(h->cfg.device));
      HIPCHECK(h, hipDeviceSynchronize());
      HIPCHECK(h, hipMemcpy(v, h->s_fb_cnt.p, sizeof v, hipMemcpyDeviceToHost));
    }
    *value = (int64_t)v[i];
    return 0;
  }
  return fail(h, "unknown counter '%s'", name);
}

int sse_set_option(sse_handle *h, const char *name, int32_t value) {
  if (!h || !name) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  if (strcmp(name, "score_small_index") == 0) {
    h->score_small_index = value != 0;
    return 0;
  }
  if (strcmp(name, "score_bf16") == 0) {
    h->score_bf16 = value != 0;
    return 0;
  }
  if (strcmp(name, "cnn_bf16") == 0) {
    if (h->cfg.network_mode != SSE_MODE_SOURCE_ONLY_CNN) return fail(h, "option cnn_bf16 needs network_mode source_only_cnn");
    h->cnn_bf16 = value != 0;
    h->packed_dirty = true;  // (re)build the bf16 copies with the next encode
    h->mp_fresh = false;
    return 0;
  }
  if (strcmp(name, "lstm_train_rows") == 0) {
    if (value != 0 && value != 32 && value != 64) return fail(h, "lstm_train_rows must be 0, 32 or 64");
    h->lstm_train_rows = (int)value;
    return 0;
  }
  if (strcmp(name, "train_bwd_x3") == 0) {
    h->train_bwd_x3 = value != 0;
    return 0;
  }
  if (strcmp(name, "train_fwd_x3") == 0) {
    h->train_fwd_x3 = value != 0;
    return 0;
  }
  if (strcmp(name, "train_dk_x3") == 0) {
    h->train_dk_x3 = value != 0;
    return 0;
  }
  if (strcmp(name, "train_generic") == 0) {
    h->train_generic = value != 0;
    return 0;
  }
  if (strcmp(name, "train_gen1") == 0) {
    h->train_gen1 = value != 0;
    return 0;
  }
  if (strcmp(name, "train_pair_dedup") == 0) {
    h->train_pair_dedup = value != 0;
    return 0;
  }
  if (strcmp(name, "train_serial") == 0) {
    h->train_serial = value != 0;
    return 0;
  }
  if (strcmp(name, "pad_skip") == 0) {
    h->pad_skip = value != 0;
    return 0;
  }
  if (strcmp(name, "score_two_pass_min_rows") == 0) {
    if (value < 0) return fail(h, "score_two_pass_min_rows must be >= 0");
    h->score_two_pass_min_rows = value;
    return 0;
  }
  if (strcmp(name, "score_two_pass_rows") == 0) {
    if (value < 0) return fail(h, "score_two_pass_rows must be >= 0");
    h->score_two_pass_rows = value;
    return 0;
  }
  if (strcmp(name, "lstm_gate_split") == 0) {
    h->lstm_gate_split = value != 0;
    return 0;
  }
  if (strcmp(name, "pad_sort_dev") == 0) {
    if (value < 0 || value > 2) return fail(h, "pad_sort_dev must be 0 (off), 1 (adaptive) or 2 (always)");
    h->pad_sort_dev = (int)value;
    return 0;
  }
  if (strcmp(name, "lstm_x3") == 0) {
    h->lstm_x3 = value != 0;
    return 0;
  }
  if (strcmp(name, "lstm_persist_epoch") == 0) {  // testing aid: move the cluster kernel's 20-bit tag epoch (wrap-around path)
    if (value < 0 || value >= (1 << 20)) return fail(h, "lstm_persist_epoch must be in [0, 2^20)");
    h->persist_epoch = (uint32_t)value;
    return 0;
  }
  if (strcmp(name, "lstm_persist_inject_miss") == 0) {  // testing aid for the fallback path of the host-buffer encodes
    h->persist_inject = value != 0;
    return 0;
  }
  if (strcmp(name, "lstm_persist_rows") == 0) {
    if (value < 0) return fail(h, "lstm_persist_rows must be >= 0");
    h->lstm_persist_rows = (int)value;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_backoff") == 0) {
    if (value < -1) return fail(h, "lstm_cluster_backoff must be >= 0 (or -1: automatic)");
    h->cluster_backoff = (int)value;
    h->cluster_skip[0] = h->cluster_skip[1] = 0;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_chunks") == 0) {
    if (value < 1) return fail(h, "lstm_cluster_chunks must be >= 1");
    h->lstm_cluster_chunks = (int)value;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_drop_wg") == 0) {
    h->lstm_cluster_drop = value != 0;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_coop") == 0) {
    h->lstm_cluster_coop = value != 0;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_write_through") == 0) {
    h->lstm_cluster_wt = value != 0;
    return 0;
  }
  if (strcmp(name, "lstm_cluster_rows") == 0) {
    if (value < 0) return fail(h, "lstm_cluster_rows must be >= 0");
    h->lstm_cluster_rows = value;
    return 0;
  }
  if (strcmp(name, "lstm_small_rows") == 0) {
    if (value < 0) return fail(h, "lstm_small_rows must be >= 0");
    h->lstm_small_rows = value;
    return 0;
  }
  return fail(h, "unknown option '%s'", name);
}

int sse_l2_normalize_dev(sse_handle *h, const float *x_dev, float *out_dev, int64_t rows, int32_t cols, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, launch_l2_normalize(x_dev, out_dev, rows, cols, (hipStream_t)stream));
  return 0;
}

int sse_index_set_dev(sse_handle *h, const float *rows_dev, int64_t N, int32_t S, int64_t id_base, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (h->idx64) {
    HIPCHECK(h, hipFree(h->idx64));
    h->idx64 = nullptr;
  }
  return index_from_dev_rows(h, rows_dev, N, S, id_base, (hipStream_t)stream);
}

int sse_index_upload(sse_handle *h, const float *rows_host, int64_t N, int32_t S, int64_t id_base) {
  if (!h || !rows_host) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (N <= 0 || S <= 0) return fail(h, "empty index");
  if (reserve(h, h->s_tmp, (size_t)N * S * sizeof(float))) return 1;
  HIPCHECK(h, hipMemcpy(h->s_tmp.p, rows_host, (size_t)N * S * sizeof(float), hipMemcpyHostToDevice));
  if (h->idx64) {
    HIPCHECK(h, hipFree(h->idx64));
    h->idx64 = nullptr;
  }
  return index_from_dev_rows(h, (const float *)h->s_tmp.p, N, S, id_base, nullptr);
}

int sse_index_upload_f64(sse_handle *h, const double *rows_host, int64_t N, int32_t S, int64_t id_base) {
  if (!h || !rows_host) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (N <= 0 || S <= 0) return fail(h, "empty index");
  if (h->idx64) {
    HIPCHECK(h, hipFree(h->idx64));
    h->idx64 = nullptr;
  }
  HIPCHECK(h, hipMalloc((void **)&h->idx64, (size_t)N * S * sizeof(double)));
  HIPCHECK(h, hipMemcpy(h->idx64, rows_host, (size_t)N * S * sizeof(double), hipMemcpyHostToDevice));
  if (reserve(h, h->s_tmp, (size_t)N * S * sizeof(float))) return 1;
  HIPCHECK(h, launch_f64_to_f32(h->idx64, (float *)h->s_tmp.p, N * S, nullptr));
  return index_from_dev_rows(h, (const float *)h->s_tmp.p, N, S, id_base, nullptr);
}

int sse_score_topk_dev(sse_handle *h, const float *q_dev, int32_t Q, int32_t k, double *out_scores_dev,
                       int64_t *out_ids_dev, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return score_dev_locked(h, q_dev, Q, k, out_scores_dev, out_ids_dev, (hipStream_t)stream);
}

int sse_score_topk(sse_handle *h, const float *q_host, int32_t Q, int32_t k, double *out_scores, int64_t *out_ids) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (Q == 0) return 0;
  if (Q < 0 || !q_host || !out_scores || !out_ids) return fail(h, "bad arguments to sse_score_topk");
  if (!h->idxp) return fail(h, "no index uploaded");
  const size_t S = h->idx_S;
  if (reserve(h, h->s_q, (size_t)Q * S * sizeof(float))) return 1;
  HIPCHECK(h, hipMemcpy(h->s_q.p, q_host, (size_t)Q * S * sizeof(float), hipMemcpyHostToDevice));
  return score_to_host_locked(h, (const float *)h->s_q.p, Q, k, out_scores, out_ids);
}

int sse_merge_topk_strided_dev(sse_handle *h, const double *in_scores_dev, const int64_t *in_ids_dev, int64_t shard_stride,
                               int32_t P, int32_t Q, int32_t k, double *out_scores_dev, int64_t *out_ids_dev, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (P < 1 || Q < 0 || k < 1 || shard_stride < (int64_t)Q * k) return fail(h, "bad arguments to sse_merge_topk_dev");
  if (Q == 0) return 0;
  HIPCHECK(h, launch_merge_topk(in_scores_dev, in_ids_dev, shard_stride, P, Q, k, out_scores_dev, out_ids_dev, (hipStream_t)stream));
  return 0;
}

int sse_merge_topk_dev(sse_handle *h, const double *in_scores_dev, const int64_t *in_ids_dev, int32_t P, int32_t Q,
                       int32_t k, double *out_scores_dev, int64_t *out_ids_dev, void *stream) {
  return sse_merge_topk_strided_dev(h, in_scores_dev, in_ids_dev, (int64_t)Q * k, P, Q, k, out_scores_dev, out_ids_dev, stream);
}

// ---------------------------------------------------------------------------
// Torch-free exchange step of the row-sharded index (BASELINE configs[3]): RCCL straight from the C ABI.
// The library does not LINK librccl: the few entry points it needs are bound at first use, ALL from ONE library instance
// (a per-symbol dlsym(RTLD_DEFAULT) misses an RCCL the host loaded RTLD_LOCAL -- the Python default, torch's
// bundled copy -- and a second instance opened beside it would be handed the first one's ncclComm_t).  Order:
//   1. $SSE_RCCL_LIB, when set: that file (the host names its RCCL explicitly);
//   2. an RCCL already mapped into the process (dl_iterate_phdr, any visibility): dlopen(its path, RTLD_NOLOAD) returns THAT
//      instance, so a communicator the host created with it and this library's ncclAllGather agree;
//   3. librccl.so.1 / librccl.so from the loader path.
// sse_rccl_library_path() reports the choice.  libsse_hip.so still loads on a box without RCCL; every other entry point works.
namespace {
typedef struct { char internal[128]; } sse_nccl_uid;  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct RcclApi {
  int (*get_unique_id)(sse_nccl_uid *) = nullptr;
  int (*comm_init_rank)(void **, int, sse_nccl_uid, int) = nullptr;
  int (*comm_destroy)(void *) = nullptr;
  int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  const char *(*error_string)(int) = nullptr;
  const char *why = nullptr;
  std::string path;  // the library instance everything above came from
};
int find_mapped_rccl(struct dl_phdr_info *info, size_t, void *data) {
  const char *name = info->dlpi_name;
  if (!name || !*name) return 0;
  const char *base = strrchr(name, '/');
  base = base ? base + 1 : name;
  if (strncmp(base, "librccl.so", 10) != 0) return 0;
  *static_cast<std::string *>(data) = name;
  return 1;  // first match: stop
}
const RcclApi *rccl_api_ptr() {
  static const RcclApi api = [] {
    RcclApi a;
    void *lib = nullptr;
    if (const char *env = getenv("SSE_RCCL_LIB")) {
      if (*env) {
        lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
          a.why = "SSE_RCCL_LIB is set but that file does not load as a library";
          return a;
        }
        a.path = env;
      }
    }
    if (!lib) {
      std::string mapped;
      dl_iterate_phdr(find_mapped_rccl, &mapped);
      if (!mapped.empty() && (lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD)) != nullptr) a.path = mapped;
    }
    for (const char *cand : {"librccl.so.1", "librccl.so"}) {
      if (lib) break;
      if ((lib = dlopen(cand, RTLD_NOW | RTLD_LOCAL)) != nullptr) a.path = cand;
    }
    if (!lib) {
      a.why = "RCCL (librccl.so.1) is not loadable in this process";
      return a;
    }
    a.get_unique_id = reinterpret_cast<int (*)(sse_nccl_uid *)>(dlsym(lib, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<int (*)(void **, int, sse_nccl_uid, int)>(dlsym(lib, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommDestroy"));
    a.all_gather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(lib, "ncclAllGather"));
    a.group_start = reinterpret_cast<int (*)()>(dlsym(lib, "ncclGroupStart"));
    a.group_end = reinterpret_cast<int (*)()>(dlsym(lib, "ncclGroupEnd"));
    a.error_string = reinterpret_cast<const char *(*)(int)>(dlsym(lib, "ncclGetErrorString"));
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.group_start || !a.group_end)
      a.why = "the RCCL library found in this process lacks an entry point this library binds";
    return a;
  }();
  return &api;
}
constexpr int SSE_NCCL_INT64 = 4;  // ncclInt64 (rccl.h: ncclDataType_t)
int rccl_fail(sse_handle *h, const char *what, int rc) {
  const RcclApi &r = *rccl_api_ptr();
  return fail(h, "%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
}
}  // namespace

const char *sse_rccl_library_path(void) {
  const RcclApi &r = *rccl_api_ptr();
  return r.why ? nullptr : r.path.c_str();
}

/* ncclGroupStart / ncclGroupEnd of the bound RCCL: ONE thread that drives several handles (one per GPU) must bracket its
 * sse_rccl_comm_init_rank calls with these, or the first (blocking) call waits for ranks that the thread has not reached yet. */
int sse_rccl_group_start(void) {
  const RcclApi &r = *rccl_api_ptr();
  return r.why ? 1 : (r.group_start() != 0);
}
int sse_rccl_group_end(void) {
  const RcclApi &r = *rccl_api_ptr();
  return r.why ? 1 : (r.group_end() != 0);
}

int sse_rccl_get_unique_id(char *id128) {
  const RcclApi &r = *rccl_api_ptr();
  if (!id128 || r.why) return 1;
  sse_nccl_uid u;
  if (r.get_unique_id(&u) != 0) return 1;
  memcpy(id128, u.internal, sizeof u.internal);
  return 0;
}

int sse_rccl_comm_init_rank(sse_handle *h, void **comm, int32_t world, int32_t rank, const char *id128) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  const RcclApi &r = *rccl_api_ptr();
  if (r.why) return fail(h, "%s", r.why);
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return fail(h, "bad arguments to sse_rccl_comm_init_rank");
  HIPCHECK(h, hipSetDevice(h->cfg.device));  // the communicator binds the calling thread's current device
  sse_nccl_uid u;
  memcpy(u.internal, id128, sizeof u.internal);
  const int rc = r.comm_init_rank(comm, world, u, rank);
  return rc == 0 ? 0 : rccl_fail(h, "ncclCommInitRank", rc);
}

int sse_rccl_comm_destroy(sse_handle *h, void *comm) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  const RcclApi &r = *rccl_api_ptr();
  if (r.why) return fail(h, "%s", r.why);
  if (!comm) return 0;
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  const int rc = r.comm_destroy(comm);
  return rc == 0 ? 0 : rccl_fail(h, "ncclCommDestroy", rc);
}

// gather + merge of lists that are already on the device; the handle mutex is held by the caller
static int allgather_merge_locked(sse_handle *h, void *comm, int world, const double *loc_s, const int64_t *loc_i, int Q, int k,
                                  double *out_s, int64_t *out_i, hipStream_t st) {
  const RcclApi &r = *rccl_api_ptr();
  if (r.why) return fail(h, "%s", r.why);
  const size_t n = (size_t)Q * k;  // 64-bit words per list
  // exchange buffer: [ this rank's (scores | ids) : 2n words ][ gathered, rank-major: world x 2n words ]
  if (reserve(h, h->s_xchg, (size_t)(world + 1) * 2 * n * 8)) return 1;
  int64_t *loc = (int64_t *)h->s_xchg.p, *all = loc + 2 * n;
  HIPCHECK(h, hipMemcpyAsync(loc, loc_s, n * 8, hipMemcpyDeviceToDevice, st));  // (float64 scores travel as their bit patterns)
  HIPCHECK(h, hipMemcpyAsync(loc + n, loc_i, n * 8, hipMemcpyDeviceToDevice, st));
  const int rc = r.all_gather(loc, all, 2 * n, SSE_NCCL_INT64, comm, st);  // ONE collective for scores and ids
  if (rc != 0) return rccl_fail(h, "ncclAllGather", rc);
  HIPCHECK(h, launch_merge_topk((const double *)all, all + n, (int64_t)(2 * n), world, Q, k, out_s, out_i, st));
  return 0;
}

int sse_allgather_merge_topk_dev(sse_handle *h, void *nccl_comm, int32_t world, const double *local_scores_dev,
                                 const int64_t *local_ids_dev, int32_t Q, int32_t k, double *out_scores_dev, int64_t *out_ids_dev,
                                 void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!nccl_comm || world < 1 || Q < 0 || k < 1 || !local_scores_dev || !local_ids_dev || !out_scores_dev || !out_ids_dev)
    return fail(h, "bad arguments to sse_allgather_merge_topk_dev");
  if (Q == 0) return 0;
  return allgather_merge_locked(h, nccl_comm, world, local_scores_dev, local_ids_dev, Q, k, out_scores_dev, out_ids_dev, (hipStream_t)stream);
}

int sse_score_topk_sharded_dev(sse_handle *h, void *nccl_comm, int32_t world, const float *q_dev, int32_t Q, int32_t k,
                               double *out_scores_dev, int64_t *out_ids_dev, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!nccl_comm || world < 1 || Q < 0 || k < 1 || !q_dev || !out_scores_dev || !out_ids_dev)
    return fail(h, "bad arguments to sse_score_topk_sharded_dev");
  if (Q == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // this shard's lists go to the out buffers first (global row ids: id_base of sse_index_set_dev), the merged lists overwrite them
  if (score_dev_locked(h, q_dev, Q, k, out_scores_dev, out_ids_dev, st)) return 1;
  return allgather_merge_locked(h, nccl_comm, world, out_scores_dev, out_ids_dev, Q, k, out_scores_dev, out_ids_dev, st);
}

static int64_t grad_arena_count(sse_handle *h) {
  int64_t n = 4;
  for (auto &v : h->vars) n += v.count;
  return n;
}

static void bind_arena(sse_handle *h, float *p, bool external) {
  TrainState &ts = *h->train;
  ts.arena = p;
  ts.arena_external = external;
  ts.grads_ready = false;
  int64_t off = 0;
  for (auto &v : h->vars) {
    v.grad = p + off;
    off += v.count;
  }
}

int sse_train_grad_count(sse_handle *h, int64_t *count) {
  if (!h || !count) return 1;
  *count = grad_arena_count(h);
  return 0;
}

int sse_train_set_grad_arena(sse_handle *h, float *arena_dev, int64_t count) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train) h->train = new TrainState();
  TrainState &ts = *h->train;
  if (arena_dev && count != grad_arena_count(h))
    return fail(h, "gradient arena holds %lld floats, the model needs %lld", (long long)count, (long long)grad_arena_count(h));
  HIPCHECK(h, hipDeviceSynchronize());
  if (ts.arena && !ts.arena_external) HIPCHECK(h, hipFree(ts.arena));
  ts.arena = nullptr;
  if (arena_dev) {
    bind_arena(h, arena_dev, true);
  } else {
    for (auto &v : h->vars) v.grad = nullptr;
    ts.arena_external = false;
    ts.grads_ready = false;
  }
  return 0;
}

static int ensure_arena(sse_handle *h) {
  TrainState &ts = *h->train;
  if (!ts.arena) {
    float *p = nullptr;
    HIPCHECK(h, hipMalloc((void **)&p, grad_arena_count(h) * sizeof(float)));
    bind_arena(h, p, false);
  }
  ts.grads_ready = false;
  return 0;
}

// token ids of one side of the batch into ts.ids[side] ([B][T] on the device): copied from the host, or -- rows mode --
// gathered on the device from the resident corpus by B row numbers (B ints cross PCIe instead of B*T)
// perm (host, [B], or nullptr): internal row r holds the caller's row perm[r].
static int stage_ids(sse_handle *h, TrainState &ts, int side, const int32_t *host, int B, int T, const int32_t *perm,
                     hipStream_t st) {
  if (reserve(h, ts.ids[side], (size_t)B * T * sizeof(int32_t))) return 1;
  if (!ts.rows_mode) {
    if (!perm) {
      HIPCHECK(h, hipMemcpyAsync(ts.ids[side].p, host, (size_t)B * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
      return 0;
    }
    // the batch as handed over is the "corpus", the permutation the row numbers (ts.perm is uploaded by the caller)
    if (reserve(h, ts.ids_raw[side], (size_t)B * T * sizeof(int32_t))) return 1;
    HIPCHECK(h, hipMemcpyAsync(ts.ids_raw[side].p, host, (size_t)B * T * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(h, launch_gather_id_rows((const int32_t *)ts.ids_raw[side].p, (const int32_t *)ts.perm.p, B, T, B,
                                      (int32_t *)ts.ids[side].p, h->err_flag, st));
    return 0;
  }
  if (!ts.corpus[side].p || ts.corpus_T[side] != T)
    return fail(h, "train step by rows: no %s corpus with T = %d on the device (sse_corpus_upload)", side ? "target" : "source", T);
  if (reserve(h, ts.rows[side], (size_t)B * sizeof(int32_t))) return 1;
  if (perm) {
    ts.h_rows[side].resize(B);
    for (int r = 0; r < B; ++r) ts.h_rows[side][r] = host[perm[r]];
    host = ts.h_rows[side].data();
  }
  HIPCHECK(h, hipMemcpyAsync(ts.rows[side].p, host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIPCHECK(h, launch_gather_id_rows((const int32_t *)ts.corpus[side].p, (const int32_t *)ts.rows[side].p, B, T,
                                    ts.corpus_N[side], (int32_t *)ts.ids[side].p, h->err_flag, st));
  return 0;
}

// source_only_cnn (BUILDER-DEFINED, see cnn_bwd.hip): row b pairs source sequence b with row tgt_rows[b]
// of the free target matrix; same loss kernel, CNN forward with arg-max tape, gather/scatter backward.
static int cnn_train_grads_locked(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_rows_host,
                                  const float *labels_host, int32_t B, int32_t T, int64_t rows_global) {
  const sse_config &c = h->cfg;
  hipStream_t st = h->stream;
  TrainState &ts = *h->train;
  const int E = c.embedding_size, S = c.encoding_size, V = c.vocab_size, Ep = emb_cols(c);
  const int Bp = round_up(B, 64);
  if (T < 5) return fail(h, "source_only_cnn needs max_seq_length >= 5 (widest filter)");
  if (E > 64) return fail(h, "train step: embedding_size %d > 64 not supported yet", E);
  const int Ep8 = round_up(E, 8);
  if ((h->cnn_bf16 ? cnn_bf16_lds_bytes(T, Ep8, 1) : cnn_lds_bytes(T, Ep, 1)) > 160 * 1024)
    return fail(h, "source_only_cnn training: T*E = %d*%d does not fit the LDS tile of the gfx950 kernel", T, E);
  // the weight-gradient kernel stages two [T][E] fp32 tiles + 384 floats (cnn_bwd.hip): its own limit, named here
  if (((size_t)2 * T * E + 384) * sizeof(float) > (size_t)160 * 1024)
    return fail(h, "source_only_cnn training: T*E = %d*%d needs %zu bytes of LDS in the weight-gradient kernel (limit 160 KiB: T*E <= 20288)",
                T, E, ((size_t)2 * T * E + 384) * sizeof(float));
  if (ensure_arena(h)) return 1;
  if (ensure_packed(h, st)) return 1;
  float *tail = ts.arena + grad_arena_count(h) - 4;
  const float inv_rows = 1.0f / (float)rows_global;
  Variable &emb = h->vars[0], &table = h->vars[h->tgt_table], &M = h->vars[h->cnn_M];

  if (reserve(h, ts.ids[1], (size_t)B * sizeof(int32_t))) return 1;
  if (reserve(h, ts.labels, (size_t)B * sizeof(float))) return 1;
  if (stage_ids(h, ts, 0, src_ids_host, B, T, nullptr, st)) return 1;
  HIPCHECK(h, hipMemcpyAsync(ts.ids[1].p, tgt_rows_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIPCHECK(h, hipMemcpyAsync(ts.labels.p, labels_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
  for (int s = 0; s < 2; ++s) {
    if (reserve(h, ts.raw[s], (size_t)Bp * S * sizeof(float))) return 1;
    if (reserve(h, ts.draw[s], (size_t)Bp * S * sizeof(float))) return 1;
  }
  if (reserve(h, h->s_feat, (size_t)((B + 31) / 32) * 72 * 256 * sizeof(float))) return 1;
  if (reserve(h, ts.feat_rm, (size_t)Bp * 576 * sizeof(float))) return 1;
  if (reserve(h, ts.pos, (size_t)Bp * 576 * sizeof(int32_t))) return 1;
  if (reserve(h, ts.dfeat, (size_t)Bp * 576 * sizeof(float))) return 1;
  if (reserve(h, ts.dw_part, cnn_dw_part_floats(E, B) * sizeof(float))) return 1;
  if (reserve(h, ts.dbias_part, (size_t)cnn_bwd_chunks(B) * 576 * sizeof(float))) return 1;
  if (reserve(h, ts.wt, (size_t)E * 1728 * sizeof(float))) return 1;
  if (reserve(h, ts.wct, cnn_wct_elems(E) * sizeof(unsigned short))) return 1;
  if (reserve(h, ts.hot_part[0], (size_t)std::max(B, cnn_dx_mfma_blocks(B)) * 2 * 64 * sizeof(float))) return 1;
  if (reserve(h, ts.dm_part[0], (size_t)proj_bwd_chunks(Bp) * 576 * S * sizeof(float))) return 1;
  if (reserve(h, ts.sq_part, (size_t)2 * B * sizeof(float))) return 1;
  if (reserve(h, ts.row_loss, (size_t)B * sizeof(float))) return 1;
  if (reserve(h, ts.row_acc, (size_t)B * sizeof(float))) return 1;

  // ---- forward with the arg-max tape; target rows looked up from the free matrix
  HIPCHECK(h, hipMemsetAsync(ts.feat_rm.p, 0, (size_t)Bp * 576 * sizeof(float), st));  // padding rows feed the dM GEMM
  if (h->cnn_bf16) {
    // option cnn_bf16 (BASELINE configs[4]): convolution on the bf16 matrix pipe over bf16-rounded embeddings and
    // filters (fp32 masters, copies refreshed by ensure_packed after every update), fp32 accumulation; bias, ReLU,
    // pooling, projection, loss and the optimizer in fp32
    HIPCHECK(h, launch_cnn_fwd_bf16((const int32_t *)ts.ids[0].p, h->emb_bf16, h->cnn_Wc16, h->cnn_bias, (float *)h->s_feat.p,
                                    h->err_flag, B, T, V, Ep8, (float *)ts.feat_rm.p, (int32_t *)ts.pos.p, st));
    HIPCHECK(h, launch_cnn_proj_x3((const float *)h->s_feat.p, h->cnn_Mx3, (float *)ts.raw[0].p, B, S, 0, st));
  } else {
    HIPCHECK(h, launch_cnn_fwd((const int32_t *)ts.ids[0].p, h->emb_pad, h->cnn_Wc, h->cnn_bias, h->cnn_Mp,
                               (float *)h->s_feat.p, (float *)ts.raw[0].p, h->err_flag, B, T, V, Ep, S, 0,
                               (float *)ts.feat_rm.p, (int32_t *)ts.pos.p, st));
  }
  HIPCHECK(h, launch_rows_gather(table.dev, (const int32_t *)ts.ids[1].p, B, Bp, table.rows, S, (float *)ts.raw[1].p,
                                 h->err_flag, st));
  // (the backward kernels validate every token id / target row themselves: the fused step reads the flag once at its end --
  // train_apply_locked; a flagged step cancels its own update on the device)
  if (!ts.defer_err && check_err_flag(h, st)) return 1;
  HIPCHECK(h, launch_loss((const float *)ts.raw[0].p, (const float *)ts.raw[1].p, (const float *)ts.labels.p,
                          (float *)ts.draw[0].p, (float *)ts.draw[1].p, (float *)ts.row_loss.p, (float *)ts.row_acc.p,
                          tail + 1, B, Bp, S, inv_rows, st));

  // ---- backward
  HIPCHECK(h, hipMemsetAsync(emb.grad, 0, emb.count * sizeof(float), st));
  HIPCHECK(h, hipMemsetAsync(table.grad, 0, table.count * sizeof(float), st));
  HIPCHECK(h, launch_proj_bwd((const float *)ts.feat_rm.p, (const float *)ts.draw[0].p, M.dev, Bp, 576, 576, S, M.grad,
                              (float *)ts.dfeat.p, (float *)ts.dm_part[0].p, st));
  const float *W[4];
  float *dW[4], *db[4];
  for (int i = 0; i < 4; ++i) {
    W[i] = h->vars[h->cnn_W[i]].dev;
    dW[i] = h->vars[h->cnn_W[i]].grad;
    db[i] = h->vars[h->cnn_b[i]].grad;
  }
  float *sq = (float *)ts.sq_part.p;
  HIPCHECK(h, launch_cnn_bwd((const int32_t *)ts.ids[0].p, emb.dev, (const float *)ts.dfeat.p, (const float *)ts.feat_rm.p,
                             (const int32_t *)ts.pos.p, W, dW, db, (float *)ts.dw_part.p, (float *)ts.dbias_part.p,
                             (float *)ts.wt.p, (unsigned short *)ts.wct.p, emb.grad, sq, (float *)ts.hot_part[0].p, B, T, E, V, h->cnn_bf16 ? 1 : 0, st));
  HIPCHECK(h, launch_rows_scatter((const float *)ts.draw[1].p, (const int32_t *)ts.ids[1].p, B, S, table.rows, table.grad, sq + B, st));
  // tail[0]: both lookups are IndexedSlices -> raw slice norms
  HIPCHECK(h, launch_sum(sq, 2 * B, (float)B, tail, st));
  ts.grads_ready = true;
  return 0;
}

// The LSTM modes at shapes the fused training kernels are not laid out for (cell size > 256, embedding_size > 64): the same
// step through lstm_generic.hip -- per-step GEMM + gate kernels, latency-bound but exact fp32 -- so that no reference flag
// combination (sse_train.py:60-74) is rejected.  Same arena layout, same loss / projection-backward / optimizer kernels.
static int train_grads_generic_locked(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host,
                                      const float *labels_host, int32_t B, int32_t T, int64_t rows_global) {
  const sse_config &c = h->cfg;
  const bool table_tgt = c.network_mode == SSE_MODE_SOURCE_ENCODER_ONLY;
  const int nside = table_tgt ? 1 : 2;
  const bool shared = c.network_mode == SSE_MODE_SHARED_ENCODER;
  hipStream_t st = h->stream;
  TrainState &ts = *h->train;
  const int E = c.embedding_size, S = c.encoding_size, V = c.vocab_size;
  if (ensure_arena(h)) return 1;
  float *tail = ts.arena + grad_arena_count(h) - 4;
  const float inv_rows = 1.0f / (float)rows_global;
  const int Bp = round_up(B, 32);
  const int32_t *ids_host[2] = {src_ids_host, tgt_ids_host};
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && table_tgt) {
      if (reserve(h, ts.ids[s], (size_t)B * sizeof(int32_t))) return 1;
      HIPCHECK(h, hipMemcpyAsync(ts.ids[s].p, ids_host[s], (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    } else if (stage_ids(h, ts, s, ids_host[s], B, T, nullptr, st)) {
      return 1;
    }
  }
  if (reserve(h, ts.labels, (size_t)B * sizeof(float))) return 1;
  HIPCHECK(h, hipMemcpyAsync(ts.labels.p, labels_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
  Variable &emb = h->vars[0];
  GenLstmDims dims[2];
  size_t g_max = 0, c_max = 0, da_max = 0, dkp_max = 0;
  for (int s = 0; s < nside; ++s) {
    dims[s] = gen_lstm_dims(B, T, E, h->enc[s].H);
    const GenLstmDims &d = dims[s];
    g_max = std::max(g_max, (size_t)d.Bp * 4 * d.Hq);
    c_max = std::max(c_max, (size_t)d.Bp * d.Hq);
    da_max = std::max(da_max, (size_t)d.Bp * d.Kp);
    dkp_max = std::max(dkp_max, gen_lstm_dk_part_floats(d));
  }
  if (reserve(h, ts.gen_G, g_max * sizeof(float)) || reserve(h, ts.gen_c, c_max * sizeof(float)) ||
      reserve(h, ts.gen_dA, da_max * sizeof(float)) || reserve(h, ts.gen_dc, c_max * sizeof(float)) ||
      reserve(h, ts.gen_dkp, dkp_max * sizeof(float)))
    return 1;
  // ---- forward with tapes (un-normalised encodings; the loss kernel normalises)
  for (int s = 0; s < 2; ++s) {
    if (reserve(h, ts.raw[s], (size_t)Bp * S * sizeof(float))) return 1;
    if (reserve(h, ts.draw[s], (size_t)Bp * S * sizeof(float))) return 1;
  }
  if (table_tgt) {
    Variable &table = h->vars[h->tgt_table];
    HIPCHECK(h, launch_rows_gather(table.dev, (const int32_t *)ts.ids[1].p, B, Bp, table.rows, S, (float *)ts.raw[1].p, h->err_flag, st));
  }
  for (int s = 0; s < nside; ++s) {
    Encoder &e = h->enc[s];
    const GenLstmDims &d = dims[s];
    if (reserve(h, ts.gen_A[s], gen_lstm_a_floats(d) * sizeof(float)) || reserve(h, ts.gen_tape[s], gen_lstm_tape_floats(d) * sizeof(float)) ||
        reserve(h, ts.gen_dG[s], gen_lstm_dg_floats(d) * sizeof(float)) || reserve(h, ts.gen_hl[s], (size_t)d.Bp * d.Hq * sizeof(float)) ||
        reserve(h, ts.gen_KT[s], gen_lstm_kt_floats(d) * sizeof(float)) || reserve(h, ts.gen_Kq[s], gen_lstm_kt_floats(d) * sizeof(float)) ||
        reserve(h, ts.gen_MT[s], (size_t)S * d.Hq * sizeof(float)))
      return 1;
    if (ts.gen_ver[s] != h->weights_version) {  // (every step changes the weights; repeated sse_train_grads calls on one set do not)
      HIPCHECK(h, launch_gen_pack(h->vars[e.kernel].dev, h->vars[e.proj].dev, d, S, (float *)ts.gen_KT[s].p, (float *)ts.gen_Kq[s].p,
                                  (float *)ts.gen_MT[s].p, st));
      ts.gen_ver[s] = h->weights_version;
    }
    HIPCHECK(h, launch_gen_forward((const int32_t *)ts.ids[s].p, emb.dev, V, (const float *)ts.gen_KT[s].p, h->vars[e.bias].dev, d,
                                   (float *)ts.gen_A[s].p, (float *)ts.gen_G.p, (float *)ts.gen_c.p, (float *)ts.gen_tape[s].p,
                                   (float *)ts.gen_hl[s].p, h->err_flag, st));
    HIPCHECK(h, launch_gen_project((const float *)ts.gen_hl[s].p, (const float *)ts.gen_MT[s].p, d, S, (float *)ts.raw[s].p, st));
  }
  // (no host round trip here any more: gen_dx_scatter_kernel validates the ids itself, as rows_scatter_kernel and cnn_dx_kernel do;
  // the forward raised the error flag, the update is cancelled on the device and sse_train_apply reports it)
  // ---- loss, train accuracy, d(raw encodings)
  if (reserve(h, ts.row_loss, (size_t)B * sizeof(float))) return 1;
  if (reserve(h, ts.row_acc, (size_t)B * sizeof(float))) return 1;
  HIPCHECK(h, launch_loss((const float *)ts.raw[0].p, (const float *)ts.raw[1].p, (const float *)ts.labels.p, (float *)ts.draw[0].p,
                          (float *)ts.draw[1].p, (float *)ts.row_loss.p, (float *)ts.row_acc.p, tail + 1, B, Bp, S, inv_rows, st));
  // ---- backward
  HIPCHECK(h, hipMemsetAsync(emb.grad, 0, emb.count * sizeof(float), st));
  const int n_sq = nside * T * B + (table_tgt ? B : 0);
  if (reserve(h, ts.sq_part, (size_t)n_sq * sizeof(float))) return 1;
  if (table_tgt) {
    Variable &table = h->vars[h->tgt_table];
    HIPCHECK(h, hipMemsetAsync(table.grad, 0, table.count * sizeof(float), st));
    HIPCHECK(h, launch_rows_scatter((const float *)ts.draw[1].p, (const int32_t *)ts.ids[1].p, B, S, table.rows, table.grad,
                                    (float *)ts.sq_part.p + (size_t)nside * T * B, st));
  }
  for (int s = 0; s < nside; ++s) {
    Encoder &e = h->enc[s];
    const GenLstmDims &d = dims[s];
    if (reserve(h, ts.dh_last[s], (size_t)d.Bp * d.Hq * sizeof(float))) return 1;
    if (reserve(h, ts.dm_part[s], (size_t)proj_bwd_chunks(d.Bp) * e.H * S * sizeof(float))) return 1;
    HIPCHECK(h, launch_proj_bwd((const float *)ts.gen_hl[s].p, (const float *)ts.draw[s].p, h->vars[e.proj].dev, d.Bp, e.H, d.Hq, S,
                                h->vars[e.proj].grad, (float *)ts.dh_last[s].p, (float *)ts.dm_part[s].p, st));
    HIPCHECK(h, launch_gen_backward((const int32_t *)ts.ids[s].p, (const float *)ts.gen_Kq[s].p, d, (const float *)ts.gen_A[s].p,
                                    (const float *)ts.gen_tape[s].p, (const float *)ts.dh_last[s].p, d.Hq, (float *)ts.gen_dG[s].p,
                                    (float *)ts.gen_dA.p, (float *)ts.gen_dc.p, (float *)ts.gen_dkp.p, (shared && s == 1) ? 1 : 0,
                                    h->vars[e.kernel].grad, h->vars[e.bias].grad, emb.grad, (float *)ts.sq_part.p + (size_t)s * T * B, V, st));
  }
  HIPCHECK(h, launch_sum((const float *)ts.sq_part.p, n_sq, (float)B, tail, st));
  ts.grads_ready = true;
  return 0;
}

// Forward, loss and backward of one batch of pair rows; gradients are scaled by 1/rows_global and left in
// the arena (with the tail sums), nothing is updated.
static int train_grads_locked(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host,
                              const float *labels_host, int32_t B, int32_t T, int64_t rows_global) {
  const sse_config &c = h->cfg;
  if (B < 1 || T < 1 || !src_ids_host || !tgt_ids_host || !labels_host || rows_global < B)
    return fail(h, "bad arguments to the train step");
  if (!h->train) h->train = new TrainState();
  if (c.network_mode == SSE_MODE_SOURCE_ONLY_CNN)
    return cnn_train_grads_locked(h, src_ids_host, tgt_ids_host, labels_host, B, T, rows_global);
  // source-encoder-only (BUILDER-DEFINED like the CNN mode: the reference's loss is ill-shaped there,
  // sse_model.py:233,290): LSTM source encoder, target side = embedding_lookup(tgt_seq_embedding, rows);
  // tgt_ids_host is int32 [B] rows of the free target matrix.  nside = sequence encoders in the step.
  const bool table_tgt = c.network_mode == SSE_MODE_SOURCE_ENCODER_ONLY;
  const int nside = table_tgt ? 1 : 2;
  {
    // shapes outside the fused training kernels (cell size > 256, embedding_size > 64; option train_generic forces it: tests)
    bool generic = h->train_generic || c.embedding_size > 64;
    for (int s = 0; s < nside; ++s) generic = generic || h->enc[s].Hp > 256 || h->enc[s].generic;
    if (generic) return train_grads_generic_locked(h, src_ids_host, tgt_ids_host, labels_host, B, T, rows_global);
  }
  hipStream_t st = h->stream;
  TrainState &ts = *h->train;
  if (!ts.side[0]) {
    // The two encoders of a step run side by side on these two streams.  Streams of ONE priority share the runtime's small pool
    // of hardware queues, and in a process that has created and destroyed many streams both can land on the same queue: the
    // encoders then run one after the other (the qna recipe, T = 1000: 26 ms / step alone, 52 inside the whole test suite, equal
    // to option train_serial; profiles/r06_notes.txt).  Different priorities come from different queue pools.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (int s = 0; s < 2; ++s) {
      HIPCHECK(h, hipStreamCreateWithPriority(&ts.side[s], hipStreamNonBlocking, s == 0 ? prio_least : prio_greatest));
      HIPCHECK(h, hipEventCreateWithFlags(&ts.ev_join[s], hipEventDisableTiming));
    }
    HIPCHECK(h, hipEventCreateWithFlags(&ts.ev_fork, hipEventDisableTiming));
  }
  const int E = c.embedding_size, S = c.encoding_size, V = c.vocab_size;
  const int Bp = round_up(B, 64), NT32 = Bp / 32;
  const bool shared = c.network_mode == SSE_MODE_SHARED_ENCODER;
  if (ensure_arena(h)) return 1;
  float *tail = ts.arena + grad_arena_count(h) - 4;
  const float inv_rows = 1.0f / (float)rows_global;

  // which kernel families this step runs on split bf16 operands (options train_fwd_x3 / train_bwd_x3 / train_dk_x3)
  bool fwd_x3[2] = {false, false}, bwd_x3[2] = {false, false};
  bool all_x3 = true;
  for (int s = 0; s < nside; ++s) {
    const Encoder &e = h->enc[s];
    fwd_x3[s] = h->train_fwd_x3 && h->train_dk_x3 && e.Hp <= 256 && e.H >= 64 && E < 64;
    // (the in-kernel dX scatter addresses the dense embedding gradient through a 32-bit buffer descriptor: tables of
    // 2 GiB and more keep the fp32 dx kernel with its 64-bit pointers)
    bwd_x3[s] = h->train_bwd_x3 && h->train_dk_x3 && e.H >= 64 && e.Hp <= 256 && (int64_t)V * E * 4 < ((int64_t)1 << 31);
    all_x3 = all_x3 && fwd_x3[s] && bwd_x3[s];
  }
  // Pure fp32 step (the default): second-generation kernels -- the forward in the inference orientation with register-built
  // tapes (lstm_fwd_kernel<.., TSW>), lstm_bwd2_kernel (dX inside, dG for the weight gradient straight from registers, no
  // A-operand dG copy in HBM), d(bias) as row E of the weight-gradient GEMM.  Any split-operand option, an embedding of 64
  // columns (no room for the constant-1 column in the 64-column x part of the A-tape) or a >= 2 GiB embedding table (32-bit
  // scatter offsets) keeps the first-generation kernels.
  const bool bwd2 = !h->train_fwd_x3 && !h->train_bwd_x3 && !h->train_dk_x3 && !h->train_gen1 && E < 64 &&
                    (int64_t)V * E * 4 < ((int64_t)1 << 31);
  // layouts derived from the variables, rebuilt after every update: only the ones this step's kernels read (an all-split
  // step needs the projections, Kh^T / Kx^T in split form and -- below -- the split kernel matrix and embedding table;
  // the fp32 fragment copies wait for the next encode or fp32 step: 7 fewer launches on the critical path of a step)
  if (bwd2 ? pack_train_fp32(h, ts, nside, st) : all_x3 ? ensure_proj_packed(h, st) : ensure_packed(h, st)) return 1;
  bool any_bwd_x3 = false;
  for (int s = 0; s < nside; ++s) any_bwd_x3 = any_bwd_x3 || bwd_x3[s];
  // (the split Kh^T / Kx^T copies are only rebuilt by steps that run the split-operand BPTT: a pure fp32 step leaves them stale)
  if ((any_bwd_x3 && ts.packed_dirty) || (!all_x3 && ts.fp32_dirty)) {
    for (int s = 0; s < nside; ++s) {
      Encoder &e = h->enc[s];
      if (e.shares_lstm_with >= 0) {
        ts.KhT[s] = ts.KhT[e.shares_lstm_with];
        ts.KxT[s] = ts.KxT[e.shares_lstm_with];
        ts.KhT16[s] = ts.KhT16[e.shares_lstm_with];
        ts.KxT16[s] = ts.KxT16[e.shares_lstm_with];
        continue;
      }
      if (any_bwd_x3 && ts.packed_dirty) {
        if (!ts.KhT16[s]) HIPCHECK(h, hipMalloc((void **)&ts.KhT16[s], kT16_elems(e.Hp) * sizeof(unsigned short)));
        HIPCHECK(h, launch_pack_kT16(h->vars[e.kernel].dev, E, e.H, e.Hp, ts.KhT16[s], st));
        if (!ts.KxT16[s]) HIPCHECK(h, hipMalloc((void **)&ts.KxT16[s], kxT16_elems(e.Hp) * sizeof(unsigned short)));
        HIPCHECK(h, launch_pack_kxT16(h->vars[e.kernel].dev, E, e.H, e.Hp, ts.KxT16[s], st));
      }
      if (!all_x3) {
        if (!ts.KhT[s]) HIPCHECK(h, hipMalloc((void **)&ts.KhT[s], (size_t)(e.Hp / 32) * (e.Hp / 2) * 256 * sizeof(float)));
        if (!ts.KxT[s]) HIPCHECK(h, hipMalloc((void **)&ts.KxT[s], (size_t)2 * (e.Hp / 2) * 256 * sizeof(float)));
        HIPCHECK(h, launch_pack_kT(h->vars[e.kernel].dev, E, e.H, e.Hp / 32, e.H, e.Hp, ts.KhT[s], st));
        HIPCHECK(h, launch_pack_kT(h->vars[e.kernel].dev, 0, E, 2, e.H, e.Hp, ts.KxT[s], st));
      }
    }
    if (!all_x3) ts.fp32_dirty = false;
    if (any_bwd_x3) ts.packed_dirty = false;
  }
  for (int s = 0; s < nside; ++s) {
    // the BPTT kernels address the gate tape [T][rows/32][Hp/32][5][1024] floats through 32-bit offsets (both generations)
    const size_t tape_bytes = (size_t)T * NT32 * (h->enc[s].Hp / 32) * 5 * 1024 * sizeof(float);
    if (tape_bytes >= ((size_t)1 << 31))
      return fail(h, "train step: batch x T x H too large -- %d rows x %d steps x cell size %d need a %.2f GiB gate tape per encoder, "
                     "the limit is 2 GiB (<= %lld rows at this T and cell size): use a smaller batch, or sse_train_grads per micro-batch with "
                     "rows_global = the whole batch and the gradient arenas summed before sse_train_apply (the data-parallel recipe)",
                  B, T, h->enc[s].H, (double)tape_bytes / (double)((size_t)1 << 30),
                  (long long)(((((size_t)1 << 31) - 1) / ((size_t)T * (h->enc[s].Hp / 32) * 5 * 1024 * sizeof(float))) * 32 / 64 * 64));
  }

  // ---- inputs
  // Paired batch (data.py:95-115 builds every batch this way: source row 2i and 2i+1 are the same sequence, once with
  // its positive and once with a sampled negative target): the rows are taken in the order [0,2,4,.. | 1,3,5,..], the
  // source encoder runs on the first half only, and its tapes serve both halves of the backward pass.  The backward
  // itself stays per row: clip_by_global_norm sees the two rows' embedding slices separately (sse_model.py:359-362).
  bool paired = h->train_pair_dedup && B % 128 == 0;  // both halves whole 64-row tiles
  if (paired) {
    if (ts.rows_mode) {
      for (int i = 0; i < B && paired; i += 2) paired = src_ids_host[i] == src_ids_host[i + 1];
    } else {
      for (int i = 0; i < B && paired; i += 2)
        paired = memcmp(src_ids_host + (size_t)i * T, src_ids_host + (size_t)(i + 1) * T, (size_t)T * sizeof(int32_t)) == 0;
    }
  }
  const int32_t *perm = nullptr;
  if (paired) {
    if ((int)ts.h_perm.size() != B) {
      ts.h_perm.resize(B);
      for (int r = 0; r < B / 2; ++r) {
        ts.h_perm[r] = 2 * r;
        ts.h_perm[B / 2 + r] = 2 * r + 1;
      }
      ts.perm_B = 0;
    }
    perm = ts.h_perm.data();
    if (ts.perm_B != B) {
      if (reserve(h, ts.perm, (size_t)B * sizeof(int32_t))) return 1;
      HIPCHECK(h, hipMemcpyAsync(ts.perm.p, perm, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      ts.perm_B = B;
    }
    ts.h_labels.resize(B);
    for (int r = 0; r < B; ++r) ts.h_labels[r] = labels_host[perm[r]];
    labels_host = ts.h_labels.data();
  }
  const int32_t *ids_host[2] = {src_ids_host, tgt_ids_host};
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && table_tgt) {  // rows of the free target matrix
      if (reserve(h, ts.ids[s], (size_t)B * sizeof(int32_t))) return 1;
      const int32_t *src = ids_host[s];
      if (perm) {
        ts.h_tgt.resize(B);
        for (int r = 0; r < B; ++r) ts.h_tgt[r] = src[perm[r]];
        src = ts.h_tgt.data();
      }
      HIPCHECK(h, hipMemcpyAsync(ts.ids[s].p, src, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    } else if (stage_ids(h, ts, s, ids_host[s], B, T, perm, st)) {
      return 1;
    }
  }
  if (reserve(h, ts.labels, (size_t)B * sizeof(float))) return 1;
  HIPCHECK(h, hipMemcpyAsync(ts.labels.p, labels_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
  const int NT_half = NT32 / 2;  // paired: 32-row tiles of one half (B % 128 == 0: Bp = B, NT_half even)

  if (h->train_fwd_x3 && h->train_dk_x3 && E < 64) {  // split embedding table of this step's weights (both encoders read it)
    if (!h->emb16) HIPCHECK(h, hipMalloc((void **)&h->emb16, lstm_x3_emb_elems(V, E) * sizeof(unsigned short)));
    Encoder &e0 = h->enc[0];
    if (!e0.Wx3t) HIPCHECK(h, hipMalloc((void **)&e0.Wx3t, lstm_x3_weight_elems(E, e0.Hp) * sizeof(unsigned short)));
    HIPCHECK(h, launch_pack_lstm_x3(h->vars[e0.kernel].dev, h->vars[e0.bias].dev, h->vars[0].dev, V, E, e0.H, e0.Hp, e0.Wx3t,
                                    h->emb16, 1, st));
    h->emb16_valid = true;  // (ensure_packed above cleared it: the table matches the current weights again)
  }
  // ---- forward with tapes (un-normalised encodings; the loss kernel normalises); the two
  // encoders are independent: fork onto two side streams, join before the loss
  HIPCHECK(h, hipEventRecord(ts.ev_fork, st));
  if (table_tgt) {
    Variable &table = h->vars[h->tgt_table];
    if (reserve(h, ts.raw[1], (size_t)Bp * S * sizeof(float))) return 1;
    if (reserve(h, ts.draw[1], (size_t)Bp * S * sizeof(float))) return 1;
    HIPCHECK(h, launch_rows_gather(table.dev, (const int32_t *)ts.ids[1].p, B, Bp, table.rows, S, (float *)ts.raw[1].p,
                                   h->err_flag, st));
  }
  for (int s = 0; s < nside; ++s) {
    hipStream_t fs = h->train_serial ? ts.side[0] : ts.side[s];
    HIPCHECK(h, hipStreamWaitEvent(fs, ts.ev_fork, 0));
    Encoder &e = h->enc[s];
    const int KT = 2 + e.Hp / 32;
    if (reserve(h, ts.raw[s], (size_t)Bp * S * sizeof(float))) return 1;
    if (reserve(h, ts.draw[s], (size_t)Bp * S * sizeof(float))) return 1;
    const bool half = paired && s == 0;  // the source encoder of a paired batch: first half of the rows only
    const int NTf = half ? NT_half : NT32, Bf = half ? B / 2 : B;
    if (reserve(h, ts.tape_g[s], (size_t)T * NTf * 4 * e.UB * 5 * 1024 * sizeof(float))) return 1;
    if (reserve(h, ts.tape_a[s], (size_t)T * NTf * 4 * KT * 256 * sizeof(float))) return 1;
    if (reserve(h, ts.h_last[s], (size_t)Bp * e.Hp * sizeof(float))) return 1;
    LstmFwdArgs a;
    a.ids = (const int32_t *)ts.ids[s].p;
    a.emb = h->emb_pad;
    a.Wp = e.Wp;
      a.Mp = e.Mp;
    a.out = (float *)ts.raw[s].p;
    a.err = h->err_flag;
    a.B = Bf;
    a.T = T;
    a.V = V;
    a.Ep = e.Ep;
    a.KGx = e.KGx;
    a.KGh = e.KGh;
    a.KGhe = (e.H + 7) / 8;
    a.S = S;
    a.NTS = (S + 31) / 32;
    a.normalize = 0;
    a.NT32 = NTf;
    a.tiles_elsewhere = (nside == 2 && !h->train_serial) ? ((paired && s == 1) ? NT_half : NT32) : 0;  // the other encoder's forward runs beside this one
    a.force_rows = h->lstm_train_rows;
    a.tape_g = (float *)ts.tape_g[s].p;
    a.tape_a = (float *)ts.tape_a[s].p;
    a.tape_a_split = h->train_dk_x3 ? 1 : 0;
    a.tape_swap = bwd2 ? 1 : 0;
    a.h_last = (float *)ts.h_last[s].p;
    if (fwd_x3[s]) {
      // the gate GEMMs as three bf16 MFMAs on hi + lo split operands (lstm_fwd_x3.hip, TRAIN): same tapes, same outputs
      Encoder &own = e.shares_lstm_with >= 0 ? h->enc[e.shares_lstm_with] : e;
      if (e.shares_lstm_with < 0 && s != 0) {  // (side 0 was packed ahead of the fork together with the embedding table)
        if (!own.Wx3t) HIPCHECK(h, hipMalloc((void **)&own.Wx3t, lstm_x3_weight_elems(E, own.Hp) * sizeof(unsigned short)));
        if (!h->emb16) HIPCHECK(h, hipMalloc((void **)&h->emb16, lstm_x3_emb_elems(V, E) * sizeof(unsigned short)));
        // both depend on the weights of this step: repacked on the launch stream (the other side's stream waits for
        // ev_fork, recorded before; the split embedding table is shared, so it is made on `st` ahead of the fork)
        HIPCHECK(h, launch_pack_lstm_x3(h->vars[own.kernel].dev, h->vars[own.bias].dev, h->vars[0].dev, V, E, own.H, own.Hp,
                                        own.Wx3t, nullptr, 1, fs));
      }
      LstmX3Args xa;
      xa.ids = a.ids;
      xa.emb16 = h->emb16;
      xa.Wx3 = own.Wx3t;
      xa.Mp = e.Mp;
      xa.out = a.out;
      xa.err = h->err_flag;
      xa.B = Bf;
      xa.T = T;
      xa.V = V;
      xa.KGX = lstm_x3_kgx(E);
      xa.H = e.H;
      xa.S = S;
      xa.NTS = (S + 31) / 32;
      xa.normalize = 0;
      xa.tape_g = a.tape_g;
      xa.tape_a = a.tape_a;
      xa.h_last = a.h_last;
      xa.NT32 = NTf;
      xa.tiles_elsewhere = a.tiles_elsewhere;
      HIPCHECK(h, launch_lstm_fwd_x3(xa, fs));
    } else {
      HIPCHECK(h, launch_lstm_fwd(a, e.Hp, fs));
    }
    if (half) {  // rows B/2 .. B-1 are the same sequences: the loss and the projection backward read them per row
      HIPCHECK(h, hipMemcpyAsync((float *)ts.raw[s].p + (size_t)Bf * S, ts.raw[s].p, (size_t)Bf * S * sizeof(float),
                                 hipMemcpyDeviceToDevice, fs));
      HIPCHECK(h, hipMemcpyAsync((float *)ts.h_last[s].p + (size_t)Bf * e.Hp, ts.h_last[s].p, (size_t)Bf * e.Hp * sizeof(float),
                                 hipMemcpyDeviceToDevice, fs));
    }
    HIPCHECK(h, hipEventRecord(ts.ev_join[s], fs));
    HIPCHECK(h, hipStreamWaitEvent(st, ts.ev_join[s], 0));
  }
  // token ids / corpus rows out of range: the sequence-encoder backward kernels validate every id themselves, so the fused
  // step reads the flag once at its end (train_apply_locked; a flagged step cancels its own update on the device) instead
  // of stalling the queue here; the free-target-matrix scatter trusts its rows: checked now
  if (!ts.defer_err && check_err_flag(h, st)) return 1;  // (rows_scatter validates the target rows itself since round 5)

  // ---- loss, train accuracy, d(raw encodings)
  if (reserve(h, ts.row_loss, (size_t)B * sizeof(float))) return 1;
  if (reserve(h, ts.row_acc, (size_t)B * sizeof(float))) return 1;
  HIPCHECK(h, launch_loss((const float *)ts.raw[0].p, (const float *)ts.raw[1].p, (const float *)ts.labels.p,
                          (float *)ts.draw[0].p, (float *)ts.draw[1].p, (float *)ts.row_loss.p, (float *)ts.row_acc.p,
                          tail + 1, B, Bp, S, inv_rows, st));

  // ---- backward
  Variable &emb = h->vars[0];
  HIPCHECK(h, hipMemsetAsync(emb.grad, 0, emb.count * sizeof(float), st));
  // partial sums of dx^2 per side: one per (tile, wave) when dX comes out of the split-operand BPTT kernel, one per
  // (step, tile) from dx_kernel (+ one per target row in source-encoder-only mode)
  int sq_off[3] = {0, 0, 0};
  for (int s = 0; s < nside; ++s)
    sq_off[s + 1] = sq_off[s] + (bwd2 ? dx_scatter_blocks(T, NT32 * 32) : bwd_x3[s] ? NT32 * (h->enc[s].Hp / 32) : T * NT32);
  const int n_sq = sq_off[nside] + (table_tgt ? B : 0);
  if (reserve(h, ts.sq_part, (size_t)n_sq * sizeof(float))) return 1;
  if (table_tgt) {
    Variable &table = h->vars[h->tgt_table];
    HIPCHECK(h, hipMemsetAsync(table.grad, 0, table.count * sizeof(float), st));
    HIPCHECK(h, launch_rows_scatter((const float *)ts.draw[1].p, (const int32_t *)ts.ids[1].p, B, S, table.rows, table.grad,
                                    (float *)ts.sq_part.p + sq_off[1], st));
  }
  HIPCHECK(h, hipEventRecord(ts.ev_fork, st));  // loss + zeroed embedding gradient are ready
  for (int s = 0; s < nside; ++s) {
    Encoder &e = h->enc[s];
    const bool half = paired && s == 0;
    const int Hp = e.Hp, KGn = Hp / 2, NTn = Hp / 8, KT = 2 + Hp / 32, RG = T * NT32 * 4;
    const int RGa = half ? RG / 2 : RG;  // r-groups of tape_a: the dK GEMM adds the two halves' dG fragments
    const int SL = dk_slices(RGa);
    // dual-encoder: the two backward chains are independent -> side streams; shared-encoder: the
    // target side accumulates onto the source side's kernel/bias gradient -> one stream, in order
    hipStream_t bs = (shared || h->train_serial) ? ts.side[0] : ts.side[s];
    if (!shared || s == 0) HIPCHECK(h, hipStreamWaitEvent(bs, ts.ev_fork, 0));
    if (reserve(h, ts.dh_last[s], (size_t)Bp * Hp * sizeof(float))) return 1;
    if (!bwd_x3[s] && !bwd2 && reserve(h, ts.dg_a[s], (size_t)T * NT32 * KGn * 256 * sizeof(float))) return 1;
    if (reserve(h, ts.hot_part[s], (size_t)std::max(T * NT32 * 2, dx_scatter_blocks(T, NT32 * 32)) * 2 * 64 * sizeof(float))) return 1;
    if (bwd2 && reserve(h, ts.dg_a[s], (size_t)T * NT32 * 32 * 64 * sizeof(float))) return 1;  // (second generation: the dX rows)
    BwdDxArgs bdx{ts.KxT16[s], (const int32_t *)ts.ids[s].p, emb.grad, (float *)ts.sq_part.p + sq_off[s], (float *)ts.hot_part[s].p,
                  B, E, V};
    if (reserve(h, ts.dg_b[s], (size_t)RG * NTn * 256 * sizeof(float))) return 1;
    if (reserve(h, ts.db_part[s], (size_t)NT32 * 4 * Hp * sizeof(float))) return 1;
    if (reserve(h, ts.dk_part[s], (size_t)SL * KT * 32 * NTn * 32 * sizeof(float))) return 1;
    if (reserve(h, ts.dm_part[s], (size_t)proj_bwd_chunks(Bp) * e.H * S * sizeof(float))) return 1;
    HIPCHECK(h, launch_proj_bwd((const float *)ts.h_last[s].p, (const float *)ts.draw[s].p, h->vars[e.proj].dev, Bp, e.H, Hp,
                                S, h->vars[e.proj].grad, (float *)ts.dh_last[s].p, (float *)ts.dm_part[s].p, bs));
    if (bwd2) {
      HIPCHECK(h, launch_lstm_bwd2((const float *)ts.tape_g[s].p, (const float *)ts.dh_last[s].p, ts.KhT[s], ts.KxT[s],
                                   (float *)ts.dg_b[s].p, (float *)ts.dg_a[s].p, (const int32_t *)ts.ids[s].p, emb.grad,
                                   (float *)ts.sq_part.p + sq_off[s], (float *)ts.hot_part[s].p, T, NT32, half ? NT_half : NT32, Hp,
                                   e.H, B, E, V, bs));
      const int acc2 = (shared && s == 1) ? 1 : 0;
      HIPCHECK(h, launch_dk((const float *)ts.tape_a[s].p, (const float *)ts.dg_b[s].p, (float *)ts.dk_part[s].p, RGa, KT, NTn, SL,
                            E, e.H, Hp, acc2, h->vars[e.kernel].grad, half ? NT_half * 4 : 0, bs, h->vars[e.bias].grad));
      HIPCHECK(h, launch_dx_hot_reduce((const float *)ts.hot_part[s].p, dx_scatter_blocks(T, NT32 * 32), E, V, emb.grad, bs));
      if (!shared || s == 1) {
        HIPCHECK(h, hipEventRecord(ts.ev_join[s], bs));
        HIPCHECK(h, hipStreamWaitEvent(st, ts.ev_join[s], 0));
      }
      continue;
    }
    HIPCHECK(h, launch_lstm_bwd((const float *)ts.tape_g[s].p, (const float *)ts.dh_last[s].p, ts.KhT[s],
                                (float *)ts.dg_a[s].p, (float *)ts.dg_b[s].p, (float *)ts.db_part[s].p, T, NT32,
                                half ? NT_half : NT32, Hp, e.H, h->train_dk_x3 ? 1 : 0, bwd_x3[s] ? ts.KhT16[s] : nullptr, &bdx, bs));
    const int accumulate = (shared && s == 1) ? 1 : 0;
    if (h->train_dk_x3)  // the 8-row r-groups of the fp32 layout pair up into 16-row groups: the same bytes
      HIPCHECK(h, launch_dk_x3(ts.tape_a[s].p, ts.dg_b[s].p, (float *)ts.dk_part[s].p, RGa / 2, KT, NTn, SL, E, e.H, Hp, accumulate,
                               h->vars[e.kernel].grad, half ? NT_half * 2 : 0, bs));
    else
      HIPCHECK(h, launch_dk((const float *)ts.tape_a[s].p, (const float *)ts.dg_b[s].p, (float *)ts.dk_part[s].p, RGa, KT, NTn, SL,
                            E, e.H, Hp, accumulate, h->vars[e.kernel].grad, half ? NT_half * 4 : 0, bs));
    HIPCHECK(h, launch_db_reduce((const float *)ts.db_part[s].p, NT32, e.H, Hp, accumulate, h->vars[e.bias].grad, bs));
    if (bwd_x3[s])  // dX left the BPTT kernel already; only the PAD / EOS rows are still to be added
      HIPCHECK(h, launch_dx_hot_reduce((const float *)ts.hot_part[s].p, T * NT32 * 2, E, V, emb.grad, bs));
    else
      HIPCHECK(h, launch_dx((const float *)ts.dg_a[s].p, ts.KxT[s], (const int32_t *)ts.ids[s].p, emb.grad,
                            (float *)ts.sq_part.p + sq_off[s], (float *)ts.hot_part[s].p, T, NT32, KGn, B, E, V, e.H, bs));
    if (!shared || s == 1) {
      HIPCHECK(h, hipEventRecord(ts.ev_join[s], bs));
      HIPCHECK(h, hipStreamWaitEvent(st, ts.ev_join[s], 0));
    }
  }

  // tail[0] = sum of squares of the raw (un-deduplicated) embedding-gradient slices, tail[3] = rows
  HIPCHECK(h, launch_sum((const float *)ts.sq_part.p, n_sq, (float)B, tail, st));
  ts.grads_ready = true;
  return 0;
}

// clip_by_global_norm over the (possibly all-reduced) arena + Adagrad + global_step (sse_model.py:355-364)
static int train_apply_locked(sse_handle *h, float *loss, float *train_acc) {
  if (!h->train || !h->train->grads_ready) return fail(h, "train apply: no gradients pending (call sse_train_grads first)");
  TrainState &ts = *h->train;
  hipStream_t st = h->stream;
  const int NORM_BLOCKS = 64;
  float *tail = ts.arena + grad_arena_count(h) - 4;
  if (reserve(h, ts.scal, 4 * sizeof(float))) return 1;
  float *scal = (float *)ts.scal.p;  // [0] global norm [1] clip scale [2] update cancelled
  if (reserve(h, ts.norm_part, (size_t)(h->vars.size() * NORM_BLOCKS + 1) * sizeof(float))) return 1;
  float *np_ = (float *)ts.norm_part.p;
  if (h->vars.size() > SSE_MAX_TENSORS) return fail(h, "too many variables");
  // dense tensors enter the global norm whole; the lookup tables (word_embedding, tgt_seq_embedding) through the raw
  // slice norm in tail[0] (tf.clip_by_global_norm over IndexedSlices.values, sse_model.py:359-362)
  MultiTensor dense, all;
  dense.n = all.n = 0;
  for (size_t i = 0; i < h->vars.size(); ++i) {
    Variable &v = h->vars[i];
    all.w[all.n] = v.dev;
    all.slot[all.n] = v.slot;
    all.grad[all.n] = v.grad;
    all.count[all.n++] = v.count;
    if (i == 0 || (int)i == h->tgt_table) continue;
    dense.w[dense.n] = v.dev;
    dense.slot[dense.n] = v.slot;
    dense.grad[dense.n] = v.grad;
    dense.count[dense.n++] = v.count;
  }
  HIPCHECK(h, launch_sumsq_multi(dense, np_, NORM_BLOCKS, st));
  // a token id out of range seen by this step's kernels (deferred check of the fused step) cancels the update on the device
  HIPCHECK(h, launch_clip_scale_multi(np_, dense.n * NORM_BLOCKS, tail, 5.0f /* max_gradient_norm, sse_model.py:117 */, h->err_flag,
                                      scal, st));
  // ---- Adagrad (dense for every tensor; rows of word_embedding with zero gradient are unchanged)
  HIPCHECK(h, launch_adagrad_multi(all, scal, h->lr, st));
  h->packed_dirty = true;
  h->weights_version += 1;
  h->mp_fresh = false;
  ts.packed_dirty = ts.fp32_dirty = true;
  ts.grads_ready = false;

  float *out = reinterpret_cast<float *>(h->pin_small + 4);
  HIPCHECK(h, hipMemcpyAsync(out, tail, 4 * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHECK(h, hipMemcpyAsync(h->pin_small, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHECK(h, sync_stream(st));
  // bits 1 | 2 | 8 (token id / corpus row out of range, sparse gradient exchange overflow) belong to this step: the update was cancelled on the device, variables
  // unchanged.  Bit 4 (a cluster-kernel give-up of an asynchronous sse_encode_dev issued earlier) is not this step's: it
  // stays in the flag for sse_synchronize / the next encode to report.
  if (h->pin_small[0] & 11) return check_err_flag(h, st);  // (resets the flag)
  h->global_step += 1;
  if (loss) *loss = out[1];
  if (train_acc) *train_acc = out[2];
  return 0;
}

int sse_train_grads(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host, const float *labels_host,
                    int32_t B, int32_t T, int64_t rows_global) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return train_grads_locked(h, src_ids_host, tgt_ids_host, labels_host, B, T, rows_global);
}

int sse_corpus_upload(sse_handle *h, int side, const int32_t *ids_host, int64_t N, int32_t T) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if ((side != 0 && side != 1) || !ids_host || N < 1 || T < 1 || N > 2147483000) return fail(h, "bad arguments to sse_corpus_upload");
  if (!h->train) h->train = new TrainState();
  TrainState &ts = *h->train;
  HIPCHECK(h, hipDeviceSynchronize());
  if (reserve(h, ts.corpus[side], (size_t)N * T * sizeof(int32_t))) return 1;
  HIPCHECK(h, hipMemcpy(ts.corpus[side].p, ids_host, (size_t)N * T * sizeof(int32_t), hipMemcpyHostToDevice));
  ts.corpus_N[side] = N;
  ts.corpus_T[side] = T;
  return 0;
}

int sse_train_grads_rows(sse_handle *h, const int32_t *src_rows_host, const int32_t *tgt_rows_host, const float *labels_host,
                         int32_t B, int64_t rows_global) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train || !h->train->corpus[0].p) return fail(h, "train step by rows: no source corpus on the device (sse_corpus_upload)");
  h->train->rows_mode = true;
  const int rc = train_grads_locked(h, src_rows_host, tgt_rows_host, labels_host, B, h->train->corpus_T[0], rows_global);
  h->train->rows_mode = false;
  return rc;
}

int sse_train_step_rows(sse_handle *h, const int32_t *src_rows_host, const int32_t *tgt_rows_host, const float *labels_host,
                        int32_t B, float *loss, float *train_acc) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train || !h->train->corpus[0].p) return fail(h, "train step by rows: no source corpus on the device (sse_corpus_upload)");
  h->train->rows_mode = true;
  h->train->defer_err = true;
  const int rc = train_grads_locked(h, src_rows_host, tgt_rows_host, labels_host, B, h->train->corpus_T[0], B);
  h->train->rows_mode = false;
  h->train->defer_err = false;
  if (rc) return 1;
  return train_apply_locked(h, loss, train_acc);
}

int64_t sse_train_packed_embedding_floats(sse_handle *h, int32_t cap) {
  if (!h || cap < 1) return 0;
  return emb_grad_packed_floats(h->cfg.embedding_size, cap);
}

int sse_train_pack_embedding_grad(sse_handle *h, int32_t cap, float *packed_dev) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train || !h->train->grads_ready) return fail(h, "pack embedding gradient: no gradients pending (call sse_train_grads first)");
  if (cap < 1 || !packed_dev) return fail(h, "bad arguments to sse_train_pack_embedding_grad");
  HIPCHECK(h, launch_emb_grad_pack(h->vars[0].grad, h->cfg.vocab_size, h->cfg.embedding_size, cap, packed_dev, h->err_flag, h->stream));
  return 0;
}

int sse_train_unpack_embedding_grad(sse_handle *h, const float *gathered_dev, int32_t world, int32_t cap) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train || !h->train->grads_ready) return fail(h, "unpack embedding gradient: no gradients pending (call sse_train_grads first)");
  if (cap < 1 || world < 1 || !gathered_dev) return fail(h, "bad arguments to sse_train_unpack_embedding_grad");
  HIPCHECK(h, launch_emb_grad_unpack(gathered_dev, world, h->cfg.vocab_size, h->cfg.embedding_size, cap, h->vars[0].grad, h->err_flag,
                                     h->stream));
  return 0;
}

int sse_train_apply(sse_handle *h, float *loss, float *train_acc) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  return train_apply_locked(h, loss, train_acc);
}

int sse_train_step(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host, const float *labels_host,
                   int32_t B, int32_t T, float *loss, float *train_acc) {
  if (!h || !loss || !train_acc) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if (!h->train) h->train = new TrainState();
  h->train->defer_err = true;  // one host sync per step: the error flag is read with the loss
  const int rc = train_grads_locked(h, src_ids_host, tgt_ids_host, labels_host, B, T, B);
  h->train->defer_err = false;
  if (rc) return 1;
  return train_apply_locked(h, loss, train_acc);
}

int sse_get_learning_rate(sse_handle *h, float *lr) {
  if (!h || !lr) return 1;
  *lr = h->lr;
  return 0;
}
int sse_set_learning_rate(sse_handle *h, float lr) {
  if (!h) return 1;
  h->lr = lr;
  return 0;
}
int sse_decay_learning_rate(sse_handle *h) {
  if (!h) return 1;
  // learning_rate.assign(max(lr * decay, 1e-3)), float32 (sse_model.py:123-124)
  h->lr = fmaxf(h->lr * h->cfg.learning_rate_decay_factor, 1e-3f);
  return 0;
}
int sse_get_global_step(sse_handle *h, int64_t *step) {
  if (!h || !step) return 1;
  *step = h->global_step;
  return 0;
}
int sse_set_global_step(sse_handle *h, int64_t step) {
  if (!h) return 1;
  h->global_step = step;
  return 0;
}

int sse_set_stream(sse_handle *h, void *stream) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, hipStreamSynchronize(h->stream));  // nothing of an earlier step may still be in flight on the old stream
  h->stream = (hipStream_t)stream;
  return 0;
}

int sse_timer_record(sse_handle *h, int32_t slot, void *stream) {
  if (!h) return 1;
  if (slot < 0 || slot >= 256) return fail(h, "timer slot %d out of range [0,256)", slot);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  if ((int)h->events.size() <= slot) h->events.resize(slot + 1, nullptr);
  if (!h->events[slot]) HIPCHECK(h, hipEventCreate(&h->events[slot]));
  HIPCHECK(h, hipEventRecord(h->events[slot], (hipStream_t)stream));
  return 0;
}
int sse_timer_elapsed_ms(sse_handle *h, int32_t a, int32_t b, float *ms) {
  if (!h || !ms) return 1;
  if (a < 0 || b < 0 || a >= (int)h->events.size() || b >= (int)h->events.size() || !h->events[a] || !h->events[b])
    return fail(h, "timer slots %d/%d were not recorded", a, b);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, hipEventSynchronize(h->events[b]));
  HIPCHECK(h, hipEventElapsedTime(ms, h->events[a], h->events[b]));
  return 0;
}
int sse_synchronize(sse_handle *h) {
  if (!h) return 1;
  std::lock_guard<std::mutex> lk(h->mu);
  HIPCHECK(h, hipSetDevice(h->cfg.device));
  HIPCHECK(h, hipDeviceSynchronize());
  return check_err_flag(h, nullptr);
}

}  // extern "C"
