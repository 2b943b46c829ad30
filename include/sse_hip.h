/*
 * sse_hip.h -- C ABI of libsse_hip.so, the MI355X (gfx950) implementation of
 * the Sequence-Semantic-Embedding hot path.
 *
 * The reference (eBay/Sequence-Semantic-Embedding, pure Python on TensorFlow
 * 1.x) has no FFI; its de-facto boundary for this path is the `SSEModel`
 * object plus `tf.Session.run(fetches, feed_dict)`.  Each entry point below
 * names the reference call it replaces (file:line in the reference tree).
 * The ctypes binding a maintainer adds on the reference side is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error;
 *     sse_last_error(h) (or sse_last_error(NULL) for sse_create failures)
 *     returns a NUL-terminated message owned by the library;
 *   - "host" pointers are caller-owned numpy-style buffers, read/written only
 *     during the call; "dev" pointers are HIP device pointers on the handle's
 *     device, `stream` is a hipStream_t passed as void* (NULL = null stream);
 *   - token ids are row-major int32 [B,T] (sse_model.py:408-409,419-421,430,439),
 *     encodings row-major float32 [B,S];
 *   - a handle may be used from several threads for encode/score (calls are
 *     serialised internally, like tf.Session.run in webserver.py:108);
 *     sse_train_step is exclusive;
 *   - the internal mutex serialises the HOST side only: the *_dev entry points enqueue on the stream they are given
 *     and share the handle's packed-weight and scratch buffers, so ONE handle must be driven from ONE stream (or the
 *     caller orders its streams with events); create one handle per stream for concurrent device-side use.  The
 *     host-buffer entry points use the null stream and synchronise before returning; the train step runs on the
 *     stream set with sse_set_stream (default: the null stream).
 */
#ifndef SSE_HIP_H
#define SSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sse_handle sse_handle;

/* network_mode values (sse_model.py:167-177) */
enum {
  SSE_MODE_DUAL_ENCODER = 0,        /* 'dual-encoder'        sse_model.py:236 */
  SSE_MODE_SHARED_ENCODER = 1,      /* 'shared-encoder'      sse_model.py:258 */
  SSE_MODE_SOURCE_ENCODER_ONLY = 2, /* 'source-encoder-only' sse_model.py:217 */
  SSE_MODE_SOURCE_ONLY_CNN = 3      /* 'source_only_cnn'     sse_model.py:179 */
};
enum { SSE_SIDE_SOURCE = 0, SSE_SIDE_TARGET = 1 };

/* modelParams of SSEModel.__init__ (sse_model.py:94-126) */
typedef struct sse_config {
  int32_t network_mode;
  int32_t vocab_size;        /* V */
  int32_t embedding_size;    /* E */
  int32_t encoding_size;     /* S */
  int32_t src_cell_size;     /* H source */
  int32_t tgt_cell_size;     /* H target (ignored in shared-encoder, sse_model.py:269) */
  int32_t max_seq_length;    /* T */
  int32_t target_space_size; /* N targets (source-encoder-only / cnn modes) */
  int32_t device;            /* HIP device ordinal */
  float learning_rate;               /* sse_model.py:122 */
  float learning_rate_decay_factor;  /* sse_model.py:124 */
} sse_config;

/* SSEModel(modelParams) -- sse_model.py:94; sse_train.py:109 */
int sse_create(const sse_config *cfg, sse_handle **out);
void sse_destroy(sse_handle *h);
const char *sse_last_error(sse_handle *h);

/* Variables, addressed by their TF names ('word_embedding',
 * 'source_encoder/rnn/basic_lstm_cell/kernel', '.../bias', 'source_encoder/src_M',
 * ..., and '<name>/Adagrad' for the optimizer slots).  Replaces
 * saver.restore / saver.save (sse_model.py:138,379-386; sse_train.py:110-113). */
int sse_num_variables(sse_handle *h);
int sse_variable_info(sse_handle *h, int index, const char **name, int64_t *count, int32_t *rows, int32_t *cols);
int sse_set_variable(sse_handle *h, const char *name, const float *host, int64_t count);
int sse_get_variable(sse_handle *h, const char *name, float *host, int64_t count);

/* session.run([model.norm_src_seq_embedding | model.src_seq_embedding |
 * model.norm_tgt_seq_embedding], feed) -- sse_evaluator.py:107-109,
 * sse_index.py:90-92, sse_demo.py:121-125.  side: SSE_SIDE_*.  normalize=1
 * gives the l2-normalised encoding (sse_model.py:282-283). */
int sse_encode(sse_handle *h, int side, const int32_t *ids_host, int32_t B, int32_t T,
               int32_t normalize, float *out_host);
int sse_encode_dev(sse_handle *h, int side, const int32_t *ids_dev, int32_t B, int32_t T,
                   int32_t normalize, float *out_dev, void *stream);

/* Library options.  "pad_skip" (default 1): inference encodes skip the left-PAD prefix of
 * every 64-row tile exactly -- the LSTM state after p leading PAD (id 0) steps is
 * sequence-independent (sse_index.py:79-85 left-pads; sse_model.py:240-242 runs all T steps),
 * so it is precomputed per p with the same kernel; results are bit-identical to pad_skip = 0.
 * "pad_sort_dev" (default 1): with pad_skip, sse_encode_dev buckets the rows of a device-resident id matrix by their
 * leading-PAD count on the device (two small launches on the caller's stream, no host round trip) so that every row tile
 * skips its whole common prefix -- what sse_encode does with a host counting sort; outputs keep the caller's row order and
 * are bit-identical.  1 = adaptive (each eligible call measures its batch; rows are bucketed while the latest completed call
 * of that side saw padding, 32-row tiles while its mean prefix was >= T / 4), 2 = always, 0 = off.  Counter "pad_sorted_calls".
 * "lstm_gate_split" (default 1): inference encodes of cell sizes <= 128 at 64-row tiles (batches above 8192 rows) run the
 * gate-split kernel (one gate per wave, two phase-shifted 32-row groups per workgroup); bit-identical to 0.
 * "score_two_pass_min_rows" / "score_two_pass_rows" (defaults 49152 / 524288): sse_score_topk* on an index of that many rows
 * with >= 1024 queries, k <= 16 and bf16 candidates ranks by a max-only sweep -> per-query threshold -> collect sweep -> float64
 * select instead of the list sweep: the same exact ids and score bits; max 0 = off.  Counter "score_two_pass_calls".
 * "lstm_small_rows" (default 1024): LSTM encodes of at most this many rows (a demo / web query, an evaluator batch of
 * 600, the tail batch of an index build) run on the few-sequences kernel -- one workgroup per 4 rows, gate GEMV on the
 * vector ALUs streaming the kernel matrix from L2 instead of ~38 us per step for a 32-row matrix tile; the same fp32
 * fma chains in the same order as the matrix kernel: results are bit-identical (tests/test_gpu_encode.py).  0 disables.
 * "score_bf16" (default 1): the candidate pass of sse_score_topk* reads bf16 copies of the index and the queries on
 * the bf16 matrix pipe; the float64 re-scoring pass, its error bound widened to the bf16 rounding, still returns
 * exactly the reference's ids and scores (bit-identical to score_bf16 = 0), ~5x faster; +50 % index memory.  Queries
 * whose result misses the certificate (top scores packed closer than the bf16 bound) are swept again with fp32
 * candidates on the device before anything reaches the float64 brute force.  0 = fp32 candidates only.
 * "lstm_persist_rows" (default 32): encodes of at most this many rows (<= 32) run on a cluster of workgroups that keeps
 * the LSTM kernel matrix in LDS and exchanges h_t every step (single query 0.13 ms at H = 256, T = 32); bit-identical
 * to the other kernels; used only when the device has at least 2 x 8 x 16 CUs.  0 disables.  The cluster's workgroups
 * must be resident together: when one does not arrive within a bounded spin (device busy with other work) the
 * host-buffer entry points re-run the batch on the few-sequences kernel (counter "lstm_persist_fallbacks");
 * sse_encode_dev reports the condition through sse_synchronize.
 * "lstm_x3" (default 0): inference encodes of more than lstm_small_rows rows (cell sizes 64 .. 256, embedding < 64) run
 * their gate GEMMs as three bf16 MFMAs per product on hi + lo split fp32 operands (x = bf16(x) + bf16(x - bf16(x))):
 * NOT bit-identical to the fp32 path, ~2e-6 from it on normalised encodings, 2.2 - 2.9x faster.
 * "train_fwd_x3", "train_bwd_x3", "train_dk_x3" (default 0: sse_train_step* computes in fp32 MFMA throughout, the
 * reference's tf.float32 arithmetic; LSTM modes, cell sizes 64 .. 256): opt in to the forward, the BPTT recurrence (+ dX)
 * and the weight-gradient GEMM on split operands in the same way (~4e-6 relative per product; loss within ~1e-5 .. 1e-4
 * relative of the fp32 kernels; the forward and BPTT variants need train_dk_x3).
 * "train_pair_dedup" (default 1): a train batch whose rows 2i, 2i+1 carry the same source sequence (data.py:95-115 builds
 * every batch that way) runs the source encoder once per pair.  "lstm_train_rows" (0 | 32 | 64): fp32 training forward
 * tile rows (measurement aid).
 * "cnn_bf16" (default 0; source_only_cnn only): the convolution of inference encodes AND of the train step reads
 * embeddings and filters rounded to bf16 (fp32 masters, fp32 accumulation, fp32 bias/ReLU/pool/projection) on the bf16
 * matrix pipe -- the reference has no reduced-precision behaviour; BASELINE configs[4] names bf16.
 * "train_generic" (default 0): run the LSTM train step on the any-shape path whatever the shape (tests, A/B).
 * "train_serial" (default 0): run both encoders of a train step on one stream (profiling aid:
 * isolated kernel durations; same results).
 * "score_small_index" (default 1): >= 1024 queries against <= 1024 index rows (index dimensions 249 .. 256, 57 .. 64,
 * 49 .. 56: the evaluator's shape, sse_evaluator.py:104-112) are scored by one launch that forms all N scores per query and
 * selects the 16 best exactly, instead of the list sweep; identical results.  "lstm_cluster_coop" (default 1): the cluster
 * encoder kernels are launched cooperatively (co-residency guaranteed by the runtime; ~20 us per call on ROCm 7.2; 0 =
 * plain launch, the give-up / fall-back path alone).  "lstm_cluster_backoff" (default -1 = automatic): eligible calls that go
 * straight to the kernels needing no co-residency after a cluster launch gave up (a give-up costs 10 ms); automatic arms 16
 * calls once a give-up was observed (another process on the device, or a refused cooperative launch).  "train_gen1" (default 0): the first-generation fp32 training
 * kernels instead of the round-4 ones (A/B and test aid; same results to fp32 summation order). */
int sse_set_option(sse_handle *h, const char *name, int32_t value);
/* Diagnostic counters (cumulative).  "score_bf16_second_chance_queries": queries whose bf16-candidate result missed
 * its certificate and were re-run with fp32 candidates.  "score_collect_queries": queries served by the collect path
 * (k > 16, or a k-th score tied with rows outside the candidate lists: one more grid-wide sweep gathers every row that
 * can be in the exact top-k).  "score_bruteforce_queries": queries that fell through to the one-workgroup-per-query
 * float64 sweep (k > 1024, or more than 4096 rows within the fp32 bound of the k-th score).
 * "lstm_persist_fallbacks": host-buffer encodes re-run on the few-sequences kernel (see option lstm_persist_rows).
 * "lstm_coop_refused" (process-wide): cooperative launches the runtime refused -- the plain launch was taken instead. */
int sse_get_counter(sse_handle *h, const char *name, int64_t *value);

/* tf.nn.l2_normalize(x, dim=-1) on device rows (sse_model.py:282-283). */
int sse_l2_normalize_dev(sse_handle *h, const float *x_dev, float *out_dev, int64_t rows, int32_t cols,
                         void *stream);

/* The target index (Evaluator.__init__ builds it from targetEncodingIndex.tsv,
 * sse_evaluator.py:80-92; sse_demo.py:79-90).  Rows are uploaded once and stay
 * resident.  *_f64 keeps the float64 values the reference parses from text for
 * the exact re-scoring pass; id_base is added to row numbers in results (row
 * shards of one index).  sse_index_set_dev adopts rows already on
 * the device (copied). */
int sse_index_upload(sse_handle *h, const float *rows_host, int64_t N, int32_t S, int64_t id_base);
int sse_index_upload_f64(sse_handle *h, const double *rows_host, int64_t N, int32_t S, int64_t id_base);
int sse_index_set_dev(sse_handle *h, const float *rows_dev, int64_t N, int32_t S, int64_t id_base,
                      void *stream);

/* np.dot(srcEnc, tgtEnc.T) + data_utils.getSortedResults, first k columns only
 * (sse_evaluator.py:110-112, data_utils.py:263-267, sse_demo.py:126-129).
 * Scores are float64 like the reference's; ties rank the lower row id first.
 * out_scores [Q,k] float64, out_ids [Q,k] int64. */
int sse_score_topk(sse_handle *h, const float *q_host, int32_t Q, int32_t k, double *out_scores,
                   int64_t *out_ids);
int sse_score_topk_dev(sse_handle *h, const float *q_dev, int32_t Q, int32_t k, double *out_scores_dev,
                       int64_t *out_ids_dev, void *stream);

/* encode + score in one call, the encodings never leaving the device: session.run([src_seq_embedding | norm_...])
 * followed by np.dot + getSortedResults[:k] as sse_demo.py:121-129, webserver.py:144-151 (and the three other routes)
 * and sse_evaluator.py:107-111 do per query / batch.  enc_out_host (may be NULL) also receives the [B,S] encodings. */
int sse_encode_score_topk(sse_handle *h, int side, const int32_t *ids_host, int32_t B, int32_t T, int32_t normalize,
                          int32_t k, double *out_scores, int64_t *out_ids, float *enc_out_host);

/* k-way merge of P per-shard top-k lists per query (after the RCCL all-gather
 * of SURVEY 8e): in_* are [P,Q,k] (shard-major), out_* [Q,k]; same order rule. */
int sse_merge_topk_dev(sse_handle *h, const double *in_scores_dev, const int64_t *in_ids_dev, int32_t P,
                       int32_t Q, int32_t k, double *out_scores_dev, int64_t *out_ids_dev, void *stream);
/* The same merge on lists that sit `shard_stride` elements apart per shard (list (p, q, j) at p * shard_stride + q * k
 * + j): lets ONE all-gather carry scores and ids together ([P][2][Q][k] 64-bit words, shard_stride = 2*Q*k). */
int sse_merge_topk_strided_dev(sse_handle *h, const double *in_scores_dev, const int64_t *in_ids_dev,
                               int64_t shard_stride, int32_t P, int32_t Q, int32_t k, double *out_scores_dev,
                               int64_t *out_ids_dev, void *stream);

/* The exchange step WITHOUT torch: RCCL from the C ABI, so that the reference-side binding of
 * INTEGRATION.md can shard an index with ctypes alone.  `nccl_comm` is an ncclComm_t (one rank per GPU, created on the handle's
 * device) OF THE RCCL INSTANCE THIS LIBRARY BOUND: libsse_hip.so does not link RCCL, it binds every entry point it needs from
 * ONE library at first use -- $SSE_RCCL_LIB when set; else an RCCL already mapped into the process, whatever its visibility
 * (a host that imported torch: torch's copy); else librccl.so.1.  sse_rccl_library_path() names it (NULL: none loadable).
 * A communicator from sse_rccl_comm_init_rank is always of that instance; one the host made itself is valid here only if the
 * host's RCCL is the file sse_rccl_library_path() reports (two RCCL instances in one process do not share communicators).
 * Threads: one thread or process per rank.  ncclCommInitRank blocks until every rank has called it, so ONE thread driving
 * several handles must bracket its sse_rccl_comm_init_rank calls with sse_rccl_group_start / sse_rccl_group_end
 * (ncclGroupStart / ncclGroupEnd); the handle's mutex is held for the duration of the call.
 *   sse_allgather_merge_topk_dev  this rank's [Q][k] lists (global row ids) -> ONE ncclAllGather of the packed
 *       (float64 score bits | int64 ids) words on `stream` -> k-way merge: the unsharded result on every rank;
 *   sse_score_topk_sharded_dev    sse_score_topk_dev on this rank's shard (rows set with id_base = shard offset) + the above;
 *   sse_rccl_get_unique_id / sse_rccl_comm_init_rank / sse_rccl_comm_destroy: thin conveniences over ncclGetUniqueId /
 *       ncclCommInitRank / ncclCommDestroy for hosts without an RCCL binding of their own (id128: the 128-byte ncclUniqueId;
 *       rank 0 creates it and hands it to the other ranks by whatever channel the host has -- file, socket, MPI).
 * Replaces nothing in the reference (it has no distributed code); consumes what sse_evaluator.py:110-111 / data_utils.py:263-267
 * compute per shard. */
const char *sse_rccl_library_path(void);
int sse_rccl_group_start(void);
int sse_rccl_group_end(void);
int sse_rccl_get_unique_id(char *id128);
int sse_rccl_comm_init_rank(sse_handle *h, void **comm, int32_t world, int32_t rank, const char *id128);
int sse_rccl_comm_destroy(sse_handle *h, void *comm);
int sse_allgather_merge_topk_dev(sse_handle *h, void *nccl_comm, int32_t world, const double *local_scores_dev,
                                 const int64_t *local_ids_dev, int32_t Q, int32_t k, double *out_scores_dev,
                                 int64_t *out_ids_dev, void *stream);
int sse_score_topk_sharded_dev(sse_handle *h, void *nccl_comm, int32_t world, const float *q_dev, int32_t Q, int32_t k,
                               double *out_scores_dev, int64_t *out_ids_dev, void *stream);

/* session.run([model.train, model.loss, model.train_acc], feed) --
 * sse_train.py:170-172; loss/acc are evaluated before the update.  labels
 * float32 [B] (sse_model.py:420).  tgt_ids_host is int32 [B,T] token ids in the
 * dual- and shared-encoder modes; in source-encoder-only and source_only_cnn
 * (builder-defined training: the reference's loss is ill-shaped for the free
 * target matrix and its CNN graph does not build) it is int32 [B] rows of the
 * free target matrix.  LSTM modes accept ANY cell size / embedding_size / encoding_size (round 5): the fused kernels cover
 * cell sizes <= 512 (training <= 256), embeddings that fit their LDS tile (training <= 64 columns) and encodings <= 512;
 * every other shape runs the per-step any-shape path (csrc/lstm_generic.hip: same arithmetic, exact fp32, latency-bound).
 * Remaining limits (rejected with an error, never silently): source_only_cnn embedding_size <= 64 for training and
 * T x E within the LDS tile; index dimension <= 1024 for scoring. */
int sse_train_step(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host,
                   const float *labels_host, int32_t B, int32_t T, float *loss, float *train_acc);

/* Data-parallel training (SURVEY 8e "Training"): sse_train_step split at the
 * gradient exchange.  sse_train_grads runs forward + loss + backward on this
 * rank's B pair rows with the loss defined as the mean over rows_global rows
 * (the sum of B over all ranks) and leaves everything that has to be summed
 * across ranks in ONE flat float32 device buffer, the gradient arena:
 *   [ d word_embedding (dense [V,E]) | d <variable 1> | ... | tail[4] ]
 * in sse_variable_info order, tail = { sum of squares of the un-deduplicated
 * embedding-gradient slices (what tf.clip_by_global_norm sees for the
 * IndexedSlices of sse_model.py:355-359), loss, train_acc, rows }.
 * The caller all-reduces (sum) the arena over RCCL, then sse_train_apply
 * clips by the global norm of the reduced gradients and applies Adagrad --
 * bit-identical on every rank.  sse_train_set_grad_arena lets the caller own
 * the buffer (e.g. a torch tensor handed to torch.distributed.all_reduce);
 * NULL returns to a library-owned one.  sse_train_step == grads(rows_global =
 * B) + apply. */
/* The stream the train-step entry points (sse_train_step[_rows], sse_train_grads[_rows], sse_train_apply) enqueue on;
 * NULL (the default) is the null stream.  The two encoders of a step run on internal side streams forked from and joined
 * to this stream, so a caller that orders other work against it (torch.distributed orders a collective against torch's
 * current stream: data_parallel.py hands that stream over) needs no further synchronisation.  The call waits for the
 * previously set stream to drain.  The other entry points take their stream per call (*_dev) or use the null stream and
 * synchronise before returning (host buffers). */
int sse_set_stream(sse_handle *h, void *stream);
int sse_train_grad_count(sse_handle *h, int64_t *count);
int sse_train_set_grad_arena(sse_handle *h, float *arena_dev, int64_t count);
int sse_train_grads(sse_handle *h, const int32_t *src_ids_host, const int32_t *tgt_ids_host,
                    const float *labels_host, int32_t B, int32_t T, int64_t rows_global);
int sse_train_apply(sse_handle *h, float *loss, float *train_acc);
/* The word-embedding gradient of a data-parallel step as (row id, gradient row) pairs (SURVEY 8e "Training": "all-gather
 * of (row-id, grad-row) for the sparse embedding grads") -- for vocabularies where the dense [V,E] block dwarfs the rows a
 * step touches.  Between sse_train_grads and sse_train_apply, on the stream of sse_set_stream, no host synchronisation:
 *   sse_train_pack_embedding_grad   compacts the rows of the arena's dense block that carry any non-zero gradient into
 *       packed_dev = [ count (int32), 0, 0, 0 | ids (int32 x cap, padded to a multiple of 4) | rows (cap x E floats) ],
 *       sse_train_packed_embedding_floats(cap) floats in all; cap = an upper bound of the rows one rank can touch that is
 *       IDENTICAL on all ranks (min(V, 2 * max rows per rank * T)): the caller all-gathers the fixed-size buffers;
 *   sse_train_unpack_embedding_grad zeroes the dense block and adds the `world` gathered buffers in rank order -- the same
 *       sums in the same order on every rank; the rest of the arena is all-reduced as before.
 * A slot overflow or a row id out of range raises device error bit 8: sse_train_apply cancels the update and reports it. */
int64_t sse_train_packed_embedding_floats(sse_handle *h, int32_t cap);
int sse_train_pack_embedding_grad(sse_handle *h, int32_t cap, float *packed_dev);
int sse_train_unpack_embedding_grad(sse_handle *h, const float *gathered_dev, int32_t world, int32_t cap);
/* Batches by row number: the padded source / target corpora (the token-id matrices Data builds
 * from TrainPairs / targetIDs, data.py:95-115) are uploaded once with sse_corpus_upload (side 0 = source corpus
 * [N,T], side 1 = target corpus) and a step ships 2*B row numbers instead of 2*B*T token ids; the batch's id matrix
 * is gathered on the device and the step is otherwise sse_train_step / sse_train_grads.  In the modes whose target
 * side is already a row of the free target matrix, tgt_rows keeps that meaning and no target corpus is needed. */
int sse_corpus_upload(sse_handle *h, int side, const int32_t *ids_host, int64_t N, int32_t T);
int sse_train_step_rows(sse_handle *h, const int32_t *src_rows_host, const int32_t *tgt_rows_host,
                        const float *labels_host, int32_t B, float *loss, float *train_acc);
int sse_train_grads_rows(sse_handle *h, const int32_t *src_rows_host, const int32_t *tgt_rows_host,
                         const float *labels_host, int32_t B, int64_t rows_global);
/* model.learning_rate.eval(), model.global_step.eval(), learning_rate_decay_op
 * (sse_train.py:181,200; sse_model.py:122-125) */
int sse_get_learning_rate(sse_handle *h, float *lr);
int sse_set_learning_rate(sse_handle *h, float lr);
int sse_decay_learning_rate(sse_handle *h);
int sse_get_global_step(sse_handle *h, int64_t *step);
int sse_set_global_step(sse_handle *h, int64_t step);

/* Timing helper for bench.py: HIP events recorded on the stream the kernels are
 * launched on.  sse_timer_record stamps event `slot` (0..255); sse_timer_elapsed_ms
 * waits for event `b` and returns the time from event `a` to event `b`. */
int sse_timer_record(sse_handle *h, int32_t slot, void *stream);
int sse_timer_elapsed_ms(sse_handle *h, int32_t a, int32_t b, float *ms);
/* hipDeviceSynchronize + report deferred device-side errors (e.g. an id out of
 * range seen by an asynchronous sse_encode_dev). */
int sse_synchronize(sse_handle *h);


/* ---- targetEncodingIndex.tsv text I/O, host only (no handle, no GPU) --------
 * sse_format_rows_f32: the vector field of sse_index.py:93-95,
 *   ",".join([str(n) for n in row]) with n a numpy.float32, byte-identical
 *   (shortest round-trip digits; positional for 1e-4 <= |x| < 1e16, else
 *   scientific).  Row r is written at out + r * sse_format_rows_stride(S)
 *   (no terminator) and its length stored in lengths[r].  Multi-threaded.
 * sse_parse_rows_f64: the inverse as sse_evaluator.py:87 / sse_demo.py:87 do it,
 *   [float(f) for f in field.split(",")] -> float64; row r is
 *   text[offsets[r] .. offsets[r+1]) (trailing newline / blanks ignored) and
 *   must hold exactly S numbers, else the call returns 2 and *bad_row is the
 *   first offending row. */
int64_t sse_format_rows_stride(int32_t S);
int sse_format_rows_f32(const float *rows, int64_t n_rows, int32_t S, char *out, int64_t *lengths);
int sse_parse_rows_f64(const char *text, const int64_t *offsets, int64_t n_rows, int32_t S, double *out,
                       int64_t *bad_row);
/* CRC-32C (Castagnoli) of n bytes, host only: the checksum TensorFlow V2 checkpoints carry per table block and per
 * tensor (saver.restore of a reference-trained model directory, sse_train.py:110-113: tf_checkpoint.py verifies them).
 * seed = 0 for a fresh sum, or the value returned for the preceding bytes. */
uint32_t sse_crc32c(const void *data, int64_t n, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* SSE_HIP_H */
