// Convolution backward of the text-CNN encoder on the bf16 matrix pipe (option cnn_bf16, BASELINE configs[4]) for gfx950.
//
// BUILDER-DEFINED training path, as cnn_bwd.hip (the reference's CNN graph does not build, sse_model.py:206; the oracle
// is oracle/sse_oracle.py::_cnn_gradients(bf16=True)).  After max-over-time pooling ONE position per (sequence, filter)
// carries gradient:  G_b[p][f] = g[b][f] if p == pos[b][f] else 0.  cnn_bwd.hip walks that sparsity with gather loops
// (0.79 ms for 8192 sequences); here the embedding gradient is written as ONE dense contraction per sequence whose
// reduction index runs over (filter width, tap j, filter f) -- 1728 terms = 108 groups of 16:
//
//     dX_b[t][e] = sum_{wi, j, f}  A_b[t][(wi,j,f)] * Wr_wi[j*E + e][f],     A_b[t][(wi,j,f)] = g[b][f] * [pos[b][f] + j == t]
//
// (Wr = the bf16-rounded filters the forward multiplied with: the backward differentiates THAT function.)  The tap shift
// lives in the A operand, so the accumulator tile IS dX_b[t][e] -- no overlap-add, every element written once.
//   * B operand (filters): packed once per step in fragment order (pack_wct_bf16_kernel): one coalesced 1-KiB block per
//     (group, 32-column e tile), staged through LDS chunk by chunk for the eight sequences (waves) of a workgroup;
//   * A operand: built in registers from the sequence's pos (bytes) / g arrays in LDS -- a lane (row t, k octet) compares
//     8 filter positions with t - j and masks the 8 bf16 of g.  g is carried as hi + lo bf16 (g = hi + lo to 2^-17): two
//     MFMAs per product, products exact in fp32, fp32 accumulation -- the gradient stays an fp32 quantity as in the oracle
//     (a single bf16 g would move the Adagrad update by ~1e-3);
//   * epilogue: sum dX^2 per sequence (TF's global norm over the raw IndexedSlices) and one float atomicAdd per non-zero
//     element into the dense d word_embedding, as cnn_dx_kernel.
// Work: 108 groups x (TT x ET tiles) x 2 (hi, lo) v_mfma_f32_32x32x16_bf16 per sequence = 864 at T = 64, E = 50 -- the flops
// of two forward convolutions; the mask building (VALU) runs beside the matrix pipe.
#include "sse_kernels.h"
#include "train.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DX_WAVES = 8;    // one sequence per wave, eight per workgroup
constexpr int DX_TMAX = 96;    // three 32-row t tiles
constexpr int DX_STEPS = 108;  // reduction groups of 16: sum over the widths of taps x filters / 16
constexpr int DX_CH = 4;       // groups per staged chunk (27 chunks)

struct CnnDxArgs {
  const int32_t *ids;         // [B][T]
  const float *dfeat;         // [Bp][576] d loss / d pooled features
  const float *feat;          // [Bp][576] pooled features (ReLU mask: > 0)
  const int32_t *pos;         // [B][576] arg-max positions
  const unsigned short *WcT;  // [108][ET][512] bf16 filter fragments in step order (pack_wct_bf16_kernel)
  float *d_emb;               // [V][E] dense embedding gradient (zeroed by the caller)
  float *sq_part;             // [B]
  float *hot_part;            // [workgroups][2][64]: the workgroup's dX sums for token ids 0 (PAD) and 1 (EOS)
  int32_t B, T, E, V;
};

__device__ __forceinline__ unsigned short dx_bf16(float f) {  // nearest bfloat16, ties to even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// step -> (first filter of the group among the 576, tap): steps run width by width, filter group by filter group, tap by tap
// (32 + 24 + 32 + 20 = 108).  Wave-uniform: scalar arithmetic.
__device__ __forceinline__ void dx_step(int step, int &f0, int &j) {
  if (step < 32) {
    f0 = (step >> 1) * 16;
    j = step & 1;
  } else if (step < 56) {
    const int r = step - 32;
    f0 = 256 + (r / 3) * 16;
    j = r % 3;
  } else if (step < 88) {
    const int r = step - 56;
    f0 = 384 + (r >> 2) * 16;
    j = r & 3;
  } else {
    const int r = step - 88;
    f0 = 512 + (r / 5) * 16;
    j = r % 5;
  }
}

// One workgroup = 8 sequences, one per wave.  The filter fragments of the next chunk of four groups travel global -> registers ->
// LDS under the current chunk's MFMAs (one 16-byte load per thread), so every wave reads its B operands from LDS and the
// 216 KiB of fragments cross the L2 once per WORKGROUP, not once per sequence (one wave per two sequences with direct L2 loads
// was measured first: 154 VGPRs + 144 accumulators = one wave per SIMD, 0.32 ms at 8192 sequences).
template <int TT, int ET>
__global__ __launch_bounds__(DX_WAVES * 64) void cnn_dx_mfma_kernel(CnnDxArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_pos[DX_WAVES][576];  // arg-max position, 255 = no gradient
  __shared__ __attribute__((aligned(16))) unsigned short s_ghi[DX_WAVES][576], s_glo[DX_WAVES][576];
  __shared__ int s_ids[DX_WAVES][DX_TMAX];
  __shared__ __attribute__((aligned(16))) unsigned short s_B[2][DX_CH * ET * 512];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, E = a.E;
  const int b = blockIdx.x * DX_WAVES + w;
  constexpr int CHUNK_VEC = DX_CH * ET * 64;  // 16-byte vectors per chunk: 512 (ET = 2: one per thread) or 256
  const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.WcT);
  u32x4 stage = {0, 0, 0, 0};
  if (tid < CHUNK_VEC) stage = wsrc[tid];  // chunk 0
  for (int f = lane; f < 576; f += 64) {
    float g = 0.0f;
    int p = 255;
    if (b < a.B) {
      g = a.dfeat[(size_t)b * 576 + f];
      if (!(a.feat[(size_t)b * 576 + f] > 0.0f)) g = 0.0f;
      if (g != 0.0f) p = a.pos[(size_t)b * 576 + f];
    }
    const unsigned short hi = dx_bf16(g);
    s_pos[w][f] = (unsigned char)p;
    s_ghi[w][f] = hi;
    s_glo[w][f] = dx_bf16(g - __uint_as_float((unsigned)hi << 16));
  }
  for (int t = lane; t < T; t += 64) {  // (an id out of range: -1 = add nothing; the forward raised the error flag and the update is cancelled)
    const int id = (b < a.B) ? a.ids[(size_t)b * T + t] : -1;
    s_ids[w][t] = (id < 0 || id >= a.V) ? -1 : id;
  }
  if (tid < CHUNK_VEC) reinterpret_cast<u32x4 *>(s_B[0])[tid] = stage;
  __syncthreads();
  f32x16 acc[TT][ET];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][et][r] = 0.0f;
  const int h = lane >> 5, row = lane & 31;
  constexpr int NCHUNK = DX_STEPS / DX_CH;
  for (int c = 0; c < NCHUNK; ++c) {
    if (c + 1 < NCHUNK && tid < CHUNK_VEC) stage = wsrc[(size_t)(c + 1) * CHUNK_VEC + tid];
    const unsigned short *bsrc = s_B[c & 1];
#pragma unroll
    for (int st = 0; st < DX_CH; ++st) {
      const int step = c * DX_CH + st;
      int f0, j;
      dx_step(step, f0, j);
      const int f = f0 + 8 * h;
      const uint2 pw = *reinterpret_cast<const uint2 *>(&s_pos[w][f]);
      const u32x4 gh = *reinterpret_cast<const u32x4 *>(&s_ghi[w][f]);
      const u32x4 gl = *reinterpret_cast<const u32x4 *>(&s_glo[w][f]);
      bf16x8 bfr[ET];
#pragma unroll
      for (int et = 0; et < ET; ++et) bfr[et] = *reinterpret_cast<const bf16x8 *>(bsrc + (st * ET + et) * 512 + lane * 8);
      int pb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pb[i] = (int)(((i < 4 ? pw.x : pw.y) >> (8 * (i & 3))) & 0xFFu);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int tm = tt * 32 + row - j;  // the pooled position whose tap j lands on this lane's row t (negative: none)
        u32x4 m;
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (pb[2 * q] == tm ? 0x0000FFFFu : 0u) | (pb[2 * q + 1] == tm ? 0xFFFF0000u : 0u);
        const bf16x8 ahi = __builtin_bit_cast(bf16x8, gh & m), alo = __builtin_bit_cast(bf16x8, gl & m);
#pragma unroll
        for (int et = 0; et < ET; ++et) {
          acc[tt][et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bfr[et], acc[tt][et], 0, 0, 0);
          acc[tt][et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bfr[et], acc[tt][et], 0, 0, 0);
        }
      }
    }
    if (c + 1 < NCHUNK && tid < CHUNK_VEC) reinterpret_cast<u32x4 *>(s_B[(c + 1) & 1])[tid] = stage;  // (last read before the previous barrier)
    __syncthreads();
  }
  // PAD (0) and EOS (1) fill most rows of a left-padded batch (and every sequence ends in EOS): their gradient rows would all
  // be atomics on the same 2 x E addresses -- the kernel ran 0.41 instead of 0.23 ms on a batch with just an EOS column.  They
  // take no atomics: a wave sums its sequence's rows per hot id in registers, the eight waves meet in LDS and the workgroup's
  // sums go to hot_part[workgroup]; dx_hot_reduce_kernel (train.hip) adds the workgroups in fixed order.
  __shared__ float s_hot[DX_WAVES][2][64];
  float sq = 0.0f, h0[ET], h1[ET];
#pragma unroll
  for (int et = 0; et < ET; ++et) h0[et] = h1[et] = 0.0f;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tt * 32 + mfma_row(r, lane);
      const int id = (t < T && b < a.B) ? s_ids[w][t] : -1;
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        const int e = et * 32 + (lane & 31);
        const float v = acc[tt][et][r];
        sq += v * v;  // (rows t >= T and columns e >= E are exact zeros: no position matches / zero filter columns)
        h0[et] += (id == 0) ? v : 0.0f;
        h1[et] += (id == 1) ? v : 0.0f;
        if (v != 0.0f && id >= 2 && e < E) atomicAdd(a.d_emb + (size_t)id * E + e, v);
      }
    }
#pragma unroll
  for (int et = 0; et < ET; ++et) {
    h0[et] += __shfl_xor(h0[et], 32);
    h1[et] += __shfl_xor(h1[et], 32);
    if (lane < 32) {
      s_hot[w][0][et * 32 + lane] = h0[et];
      s_hot[w][1][et * 32 + lane] = h1[et];
    }
  }
  if (ET == 1 && lane < 32) s_hot[w][0][32 + lane] = s_hot[w][1][32 + lane] = 0.0f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0 && b < a.B) a.sq_part[b] = sq;
  __syncthreads();
  if (tid < 128) {
    const int hid = tid >> 6, e = tid & 63;
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < DX_WAVES; ++i) v += s_hot[i][hid][e];
    a.hot_part[((size_t)blockIdx.x * 2 + hid) * 64 + e] = v;
  }
}

// Filters W_wi [fs*E][nf] (fp32 masters) -> bf16 fragments of the dX contraction in STEP order: block (step, et), step = width,
// filter group fg, tap j as in dx_step; lane l (column e = et*32 + (l & 31), k octet l >> 5) owns the 8 filters
// fg*16 + 8*(l >> 5) + i of filter row j*E + e (0 for e >= E).
struct WctArgs {
  const float *W[4];
  unsigned short *out;
  int32_t E, ET;
};
__global__ void pack_wct_bf16_kernel(WctArgs a) {
  const int fsv[4] = {2, 3, 4, 5}, nfv[4] = {256, 128, 128, 64}, sb[5] = {0, 32, 56, 88, 108};
  const int64_t total = (int64_t)DX_STEPS * a.ET * 512;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int idx = (int)(i & 7), l = (int)((i >> 3) & 63);
    const int blk = (int)(i >> 9), et = blk % a.ET, step = blk / a.ET;
    int wi = 0;
    while (step >= sb[wi + 1]) ++wi;
    const int fg = (step - sb[wi]) / fsv[wi], j = (step - sb[wi]) % fsv[wi];
    const int e = et * 32 + (l & 31), f = fg * 16 + 8 * (l >> 5) + idx;
    a.out[i] = (e < a.E) ? dx_bf16(a.W[wi][(size_t)(j * a.E + e) * nfv[wi] + f]) : (unsigned short)0;
  }
}

}  // namespace

int cnn_dx_mfma_blocks(int B) { return (B + DX_WAVES - 1) / DX_WAVES; }
bool cnn_dx_mfma_ok(int T, int E) { return T >= 5 && T <= DX_TMAX && E >= 1 && E <= 64; }
size_t cnn_wct_elems(int E) { return (size_t)DX_STEPS * ((E + 31) / 32) * 512; }

// dX of the whole batch on the bf16 matrix pipe; wct_scratch: cnn_wct_elems(E) bf16, rebuilt here from the masters
hipError_t launch_cnn_dx_mfma(const int32_t *ids, const float *dfeat, const float *feat, const int32_t *pos, const float *const W[4],
                              unsigned short *wct_scratch, float *d_emb, float *sq_part, float *hot_part /* [ceil(B/8)][2][64] */, int B, int T, int E, int V,
                              hipStream_t st) {
  if (!cnn_dx_mfma_ok(T, E)) return hipErrorInvalidValue;
  const int ET = (E + 31) / 32, TT = (T + 31) / 32;
  WctArgs wa;
  for (int i = 0; i < 4; ++i) wa.W[i] = W[i];
  wa.out = wct_scratch;
  wa.E = E;
  wa.ET = ET;
  hipLaunchKernelGGL(pack_wct_bf16_kernel, dim3((int)((cnn_wct_elems(E) + 255) / 256)), dim3(256), 0, st, wa);
  CnnDxArgs a{ids, dfeat, feat, pos, wct_scratch, d_emb, sq_part, hot_part, B, T, E, V};
  const dim3 grid((B + DX_WAVES - 1) / DX_WAVES), block(DX_WAVES * 64);
#define DX_GO(tt, et) hipLaunchKernelGGL((cnn_dx_mfma_kernel<tt, et>), grid, block, 0, st, a)
  if (ET == 1) {
    if (TT == 1) DX_GO(1, 1);
    else if (TT == 2) DX_GO(2, 1);
    else DX_GO(3, 1);
  } else {
    if (TT == 1) DX_GO(1, 2);
    else if (TT == 2) DX_GO(2, 2);
    else DX_GO(3, 2);
  }
#undef DX_GO
  return launch_dx_hot_reduce(hot_part, (int)grid.x, E, V, d_emb, st);
}
