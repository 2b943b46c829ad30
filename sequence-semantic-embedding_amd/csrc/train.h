// Training-path launchers (train.hip) and the per-handle training state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

hipError_t launch_loss(const float *src_raw, const float *tgt_raw, const float *labels, float *d_src, float *d_tgt,
                       float *row_loss, float *row_acc, float *out2, int B, int Bp, int S, float inv_rows,
                       hipStream_t st);
int proj_bwd_chunks(int Bp);
hipError_t launch_proj_bwd(const float *hT, const float *d, const float *M, int Bp, int H, int Hp, int S, float *dM,
                           float *dh, float *dm_part /* [proj_bwd_chunks(Bp)][H*S] */, hipStream_t st);
// embedding-gradient part of the split-operand BPTT kernel (KhT16 != nullptr): dX = dG . Kx^T is formed from the dG tile in
// LDS and scattered into d_emb by the kernel itself -- dg_a is not written and launch_dx is not used; the caller runs
// launch_dx_hot_reduce(hot_part, T * NT32 * 2, ...) afterwards and sums sq_part[NT32 * (Hp / 32)]
struct BwdDxArgs {
  const unsigned short *KxT16;  // launch_pack_kxT16
  const int32_t *ids;           // [B][T]
  float *d_emb;                 // [V][E] zero-initialised
  float *sq_part;               // [NT32 * Hp/32]
  float *hot_part;              // [T * NT32 * 2][2][64]
  int32_t B, E, V;
};
hipError_t launch_lstm_bwd(const float *tape_g, const float *dh_last, const float *KhT, float *dg_a, float *dg_b,
                           float *db_part, int T, int NT32, int NT_tape /* 0: NT32 */, int Hp, int H,
                           int dg_b_split /* 1: split bf16 frag16 blocks for launch_dk_x3 */,
                           const unsigned short *KhT16 /* non-null: split-operand recurrent GEMM (launch_pack_kT16) */,
                           const BwdDxArgs *dx /* with KhT16 */, hipStream_t st);
size_t kxT16_elems(int Hp);
hipError_t launch_pack_kxT16(const float *K, int E, int H, int Hp, unsigned short *out, hipStream_t stream);
hipError_t launch_dx_hot_reduce(const float *hot_part, int nblocks, int E, int V, float *d_emb, hipStream_t st);
size_t kT16_elems(int Hp);
hipError_t launch_pack_kT16(const float *K, int E, int H, int Hp, unsigned short *out, hipStream_t stream);
int dk_slices(int RG);
hipError_t launch_dk(const float *tape_a, const float *dg_b, float *part, int RG, int KT, int NTn, int SL, int E, int H,
                     int Hp, int accumulate, float *dK, int pair_rg /* 0: dg_b has RG r-groups */, hipStream_t st,
                     float *db = nullptr /* non-null (E < 64): d(bias) from row E of the partials (A-tape with the constant-1 column) */);
// fp32 BPTT, second generation (lstm_bwd2.hip): tape_g in the lane = sequence layout (LstmFwdArgs::tape_swap); dX formed
// inside the kernel, written as rows to dx [T][NT32*32][64] and scatter-added by dx_scatter_kernel (same launcher):
// sq_part / hot_part hold dx_scatter_blocks(T, NT32*32) entries / [..][2][64] floats (launch_dx_hot_reduce(.., that many
// blocks, ..) afterwards); dg_b for launch_dk; no bias partials (launch_dk's db)
int dx_scatter_blocks(int T, int Bp);
hipError_t launch_lstm_bwd2(const float *tape_g, const float *dh_last, const float *KhT, const float *KxT, float *dg_b,
                            float *dx, const int32_t *ids, float *d_emb, float *sq_part, float *hot_part, int T, int NT32,
                            int NT_tape /* 0: NT32 */, int Hp, int H, int B, int E, int V, hipStream_t st);
hipError_t launch_dk_x3(const void *tape_a, const void *dg_b, float *part, int G, int KT, int NTn, int SL, int E, int H,
                        int Hp, int accumulate, float *dK, int pair_g, hipStream_t st);
hipError_t launch_db_reduce(const float *db_part, int NT32, int H, int Hp, int accumulate, float *db, hipStream_t st);
hipError_t launch_dx(const float *dg_a, const float *KxT, const int32_t *ids, float *d_emb, float *sq_part,
                     float *hot_part /* [T*NT32][2][64] floats */, int T, int NT32, int KGn, int B, int E, int V, int H,
                     hipStream_t st);
// every tensor of the model in one launch (blockIdx.y = tensor)
#define SSE_MAX_TENSORS 16
struct MultiTensor {
  float *w[SSE_MAX_TENSORS], *slot[SSE_MAX_TENSORS];
  const float *grad[SSE_MAX_TENSORS];
  int64_t count[SSE_MAX_TENSORS];
  int32_t n;
};
hipError_t launch_sumsq_multi(const MultiTensor &mt, float *part /* [mt.n][nblocks] */, int nblocks, hipStream_t st);
// scal[0] = sqrt(sum(part[0..n)) + *extra), scal[1] = clip * min(1/norm, 1/clip); *err != 0 cancels the update (scal[1] = 0, scal[2] = 1)
hipError_t launch_clip_scale_multi(const float *part, int n, const float *extra, float clip, const int32_t *err, float *scal,
                                   hipStream_t st);
hipError_t launch_adagrad_multi(const MultiTensor &mt, const float *scal, float lr, hipStream_t st);
hipError_t launch_sumsq(const float *g, int64_t n, float *part, int nblocks, hipStream_t st);
hipError_t launch_sum(const float *part, int n, float tag, float *out /* out[0]=sum, out[3]=tag */, hipStream_t st);
hipError_t launch_clip_scale(const float *part, int n, float clip, float *scal, hipStream_t st);
hipError_t launch_adagrad(float *w, float *accum, const float *grad, const float *scal, float lr, int64_t n,
                          hipStream_t st);
hipError_t launch_pack_kT(const float *K, int row0, int nrows, int RT, int H, int Hp, float *out, hipStream_t stream);

// data-parallel exchange of the word-embedding gradient as (row id, gradient row) pairs: see train.hip
int64_t emb_grad_packed_floats(int E, int cap);
hipError_t launch_emb_grad_pack(const float *g, int V, int E, int cap, float *packed, int32_t *err, hipStream_t st);
hipError_t launch_emb_grad_unpack(const float *gathered, int world, int V, int E, int cap, float *g, int32_t *err, hipStream_t st);

// LSTM encoder for any shape (lstm_generic.hip): per-step GEMM + gate kernels behind the fused ones
struct GenLstmDims {
  int B, Bp, T, E, H, Hq, Kp;  // Bp = rows padded to 32, Hq = cell size padded to 8, Kp = padded width of an A row [x | h | pad]
};
GenLstmDims gen_lstm_dims(int B, int T, int E, int H);
size_t gen_lstm_a_floats(const GenLstmDims &d);
size_t gen_lstm_tape_floats(const GenLstmDims &d);
size_t gen_lstm_dg_floats(const GenLstmDims &d);
size_t gen_lstm_kt_floats(const GenLstmDims &d);
size_t gen_lstm_dk_part_floats(const GenLstmDims &d);
hipError_t launch_gen_pack(const float *K, const float *Mv, const GenLstmDims &d, int S, float *KT, float *Kq, float *MT, hipStream_t st);
hipError_t launch_gen_forward(const int32_t *ids, const float *emb, int V, const float *KT, const float *bias, const GenLstmDims &d,
                              float *A, float *G, float *c, float *tape, float *h_last, int32_t *err, hipStream_t st);
hipError_t launch_gen_project(const float *h_last, const float *MT, const GenLstmDims &d, int S, float *raw, hipStream_t st);
hipError_t launch_gen_backward(const int32_t *ids, const float *Kq, const GenLstmDims &d, const float *A, const float *tape,
                               const float *dh_last, int ldh_last, float *dG, float *dA, float *dc, float *dk_part, int accumulate,
                               float *dK, float *db, float *d_emb, float *sq, int V, hipStream_t st);

// text-CNN training path (cnn_bwd.hip)
int cnn_bwd_chunks(int B);
size_t cnn_dw_part_floats(int E, int B);
hipError_t launch_rows_gather(const float *table, const int32_t *rows, int B, int Bp, int N, int S, float *out,
                              int32_t *err, hipStream_t st);
hipError_t launch_rows_scatter(const float *d, const int32_t *rows, int B, int S, int N, float *d_table, float *sq,
                               hipStream_t st);
hipError_t launch_cnn_bwd(const int32_t *ids, const float *emb, const float *dfeat, const float *feat, const int32_t *pos,
                          const float *const W[4], float *const dW[4], float *const db[4], float *dw_part,
                          float *db_part, float *wt_scratch, unsigned short *wct_scratch, float *d_emb, float *sq_part,
                          float *hot_part, int B, int T, int E, int V, int bf16, hipStream_t st);
// dX of the text-CNN on the bf16 matrix pipe (cnn_bwd_mfma.hip; option cnn_bf16, T <= 96)
bool cnn_dx_mfma_ok(int T, int E);
size_t cnn_wct_elems(int E);
hipError_t launch_cnn_dx_mfma(const int32_t *ids, const float *dfeat, const float *feat, const int32_t *pos, const float *const W[4],
                              unsigned short *wct_scratch, float *d_emb, float *sq_part, float *hot_part, int B, int T, int E, int V,
                              hipStream_t st);
int cnn_dx_mfma_blocks(int B);
// out[b][0..T) = corpus[rows[b]][0..T)  (rows outside [0, N): error flag bit 1, row 0 used)
hipError_t launch_gather_id_rows(const int32_t *corpus, const int32_t *rows, int B, int T, int64_t N, int32_t *out,
                                 int32_t *err, hipStream_t st);
