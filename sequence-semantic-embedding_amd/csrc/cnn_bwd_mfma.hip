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
//     (group, 32-column e tile), L2 resident (216 KiB), shared by the two sequences a wave works on;
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

constexpr int DX_WAVES = 4, DX_SG = 2, DX_SEQ = DX_WAVES * DX_SG;  // sequences per workgroup
constexpr int DX_TMAX = 96;                                        // three 32-row t tiles

struct CnnDxArgs {
  const int32_t *ids;         // [B][T]
  const float *dfeat;         // [Bp][576] d loss / d pooled features
  const float *feat;          // [Bp][576] pooled features (ReLU mask: > 0)
  const int32_t *pos;         // [B][576] arg-max positions
  const unsigned short *WcT;  // [108][ET][512] bf16 filter fragments (pack_wct_bf16_kernel)
  float *d_emb;               // [V][E] dense embedding gradient (zeroed by the caller)
  float *sq_part;             // [B]
  int32_t B, T, E, wbytes;
};

__device__ __forceinline__ unsigned short dx_bf16(float f) {  // nearest bfloat16, ties to even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// One filter width: FS taps x NF filters = FS * NF / 16 reduction groups, group (j, fg) at block KB + j * (NF / 16) + fg.
template <int FS, int NF, int FOFF, int KB, int TT, int ET>
__device__ __forceinline__ void dx_width(const __amdgpu_buffer_rsrc_t wr, const unsigned char (*s_pos)[576], const unsigned short (*s_ghi)[576],
                                         const unsigned short (*s_glo)[576], int sl, int lane, f32x16 (&acc)[DX_SG][TT][ET]) {
  const int h = lane >> 5, row = lane & 31, voff = lane * 16;
  auto wl = [&](int kg, int et) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, (kg * ET + et) * 1024, 0));
  };
  constexpr int NFG = NF / 16, NIT = NFG * FS;
  bf16x8 bc[ET], bn[ET];
#pragma unroll
  for (int et = 0; et < ET; ++et) bc[et] = wl(KB, et);
  for (int fg = 0; fg < NFG; ++fg) {
    const int f = FOFF + fg * 16 + 8 * h;
    int pb[DX_SG][8];
    u32x4 gh[DX_SG], gl[DX_SG];
#pragma unroll
    for (int s = 0; s < DX_SG; ++s) {
      const uint2 pw = *reinterpret_cast<const uint2 *>(&s_pos[sl + s][f]);
#pragma unroll
      for (int i = 0; i < 8; ++i) pb[s][i] = (int)(((i < 4 ? pw.x : pw.y) >> (8 * (i & 3))) & 0xFFu);
      gh[s] = *reinterpret_cast<const u32x4 *>(&s_ghi[sl + s][f]);
      gl[s] = *reinterpret_cast<const u32x4 *>(&s_glo[sl + s][f]);
    }
#pragma unroll
    for (int j = 0; j < FS; ++j) {
      // the next group's filter fragments are in flight under this group's MFMAs
      const int it = fg * FS + j;
      const int nj = (j + 1 < FS) ? j + 1 : 0, nfg = (j + 1 < FS) ? fg : fg + 1;
      const int kn = (it + 1 < NIT) ? KB + nj * NFG + nfg : KB + j * NFG + fg;
#pragma unroll
      for (int et = 0; et < ET; ++et) bn[et] = wl(kn, et);
#pragma unroll
      for (int s = 0; s < DX_SG; ++s) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int tm = tt * 32 + row - j;  // the pooled position whose tap j lands on this lane's row t (negative: none)
          u32x4 m;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            m[q] = (pb[s][2 * q] == tm ? 0x0000FFFFu : 0u) | (pb[s][2 * q + 1] == tm ? 0xFFFF0000u : 0u);
          const bf16x8 ahi = __builtin_bit_cast(bf16x8, gh[s] & m), alo = __builtin_bit_cast(bf16x8, gl[s] & m);
#pragma unroll
          for (int et = 0; et < ET; ++et) {
            acc[s][tt][et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bc[et], acc[s][tt][et], 0, 0, 0);
            acc[s][tt][et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bc[et], acc[s][tt][et], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int et = 0; et < ET; ++et) bc[et] = bn[et];
    }
  }
}

template <int TT, int ET>
__global__ __launch_bounds__(DX_WAVES * 64) void cnn_dx_mfma_kernel(CnnDxArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_pos[DX_SEQ][576];  // arg-max position, 255 = no gradient
  __shared__ __attribute__((aligned(16))) unsigned short s_ghi[DX_SEQ][576], s_glo[DX_SEQ][576];
  __shared__ int s_ids[DX_SEQ][DX_TMAX];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, E = a.E;
  const int sl = w * DX_SG, b0 = blockIdx.x * DX_SEQ + sl;
#pragma unroll
  for (int s = 0; s < DX_SG; ++s) {
    const int b = b0 + s;
    for (int f = lane; f < 576; f += 64) {
      float g = 0.0f;
      int p = 255;
      if (b < a.B) {
        g = a.dfeat[(size_t)b * 576 + f];
        if (!(a.feat[(size_t)b * 576 + f] > 0.0f)) g = 0.0f;
        if (g != 0.0f) p = a.pos[(size_t)b * 576 + f];
      }
      const unsigned short hi = dx_bf16(g);
      s_pos[sl + s][f] = (unsigned char)p;
      s_ghi[sl + s][f] = hi;
      s_glo[sl + s][f] = dx_bf16(g - __uint_as_float((unsigned)hi << 16));
    }
    for (int t = lane; t < T; t += 64) s_ids[sl + s][t] = (b < a.B) ? a.ids[(size_t)b * T + t] : 0;
  }
  __syncthreads();
  f32x16 acc[DX_SG][TT][ET];
#pragma unroll
  for (int s = 0; s < DX_SG; ++s)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][tt][et][r] = 0.0f;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.WcT), 0, a.wbytes, 0x00020000);
  dx_width<2, 256, 0, 0, TT, ET>(wr, s_pos, s_ghi, s_glo, sl, lane, acc);
  dx_width<3, 128, 256, 32, TT, ET>(wr, s_pos, s_ghi, s_glo, sl, lane, acc);
  dx_width<4, 128, 384, 56, TT, ET>(wr, s_pos, s_ghi, s_glo, sl, lane, acc);
  dx_width<5, 64, 512, 88, TT, ET>(wr, s_pos, s_ghi, s_glo, sl, lane, acc);
#pragma unroll
  for (int s = 0; s < DX_SG; ++s) {
    const int b = b0 + s;
    float sq = 0.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = tt * 32 + mfma_row(r, lane), e = et * 32 + (lane & 31);
          const float v = acc[s][tt][et][r];
          sq += v * v;  // (rows t >= T and columns e >= E are exact zeros: no position matches / zero filter columns)
          if (v != 0.0f && t < T && e < E && b < a.B) atomicAdd(a.d_emb + (size_t)s_ids[sl + s][t] * E + e, v);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0 && b < a.B) a.sq_part[b] = sq;
  }
}

// Filters W_wi [fs*E][nf] (fp32 masters) -> bf16 fragments of the dX contraction: block (kg, et) with kg = KB_wi + j * (nf/16) + fg;
// lane l (column e = et*32 + (l & 31), k octet l >> 5) owns the 8 filters fg*16 + 8*(l >> 5) + i of row j*E + e (0 for e >= E).
struct WctArgs {
  const float *W[4];
  unsigned short *out;
  int32_t E, ET;
};
__global__ void pack_wct_bf16_kernel(WctArgs a) {
  const int fsv[4] = {2, 3, 4, 5}, nfv[4] = {256, 128, 128, 64}, kb[5] = {0, 32, 56, 88, 108};
  const int64_t total = (int64_t)108 * a.ET * 512;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int idx = (int)(i & 7), l = (int)((i >> 3) & 63);
    const int blk = (int)(i >> 9), et = blk % a.ET, kg = blk / a.ET;
    int wi = 0;
    while (kg >= kb[wi + 1]) ++wi;
    const int nfg = nfv[wi] / 16, j = (kg - kb[wi]) / nfg, fg = (kg - kb[wi]) % nfg;
    const int e = et * 32 + (l & 31), f = fg * 16 + 8 * (l >> 5) + idx;
    a.out[i] = (e < a.E && j < fsv[wi]) ? dx_bf16(a.W[wi][(size_t)(j * a.E + e) * nfv[wi] + f]) : (unsigned short)0;
  }
}

}  // namespace

bool cnn_dx_mfma_ok(int T, int E) { return T >= 5 && T <= DX_TMAX && E >= 1 && E <= 64; }
size_t cnn_wct_elems(int E) { return (size_t)108 * ((E + 31) / 32) * 512; }

// dX of the whole batch on the bf16 matrix pipe; wct_scratch: cnn_wct_elems(E) bf16, rebuilt here from the masters
hipError_t launch_cnn_dx_mfma(const int32_t *ids, const float *dfeat, const float *feat, const int32_t *pos, const float *const W[4],
                              unsigned short *wct_scratch, float *d_emb, float *sq_part, int B, int T, int E, hipStream_t st) {
  if (!cnn_dx_mfma_ok(T, E)) return hipErrorInvalidValue;
  const int ET = (E + 31) / 32, TT = (T + 31) / 32;
  WctArgs wa;
  for (int i = 0; i < 4; ++i) wa.W[i] = W[i];
  wa.out = wct_scratch;
  wa.E = E;
  wa.ET = ET;
  hipLaunchKernelGGL(pack_wct_bf16_kernel, dim3((int)((cnn_wct_elems(E) + 255) / 256)), dim3(256), 0, st, wa);
  CnnDxArgs a{ids, dfeat, feat, pos, wct_scratch, d_emb, sq_part, B, T, E, (int32_t)(cnn_wct_elems(E) * sizeof(unsigned short))};
  const dim3 grid((B + DX_SEQ - 1) / DX_SEQ), block(DX_WAVES * 64);
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
  return hipGetLastError();
}
