// Text-CNN encoder forward with bf16 STORAGE and fp32 accumulation (BASELINE configs[4] names bf16; the
// reference has no reduced-precision behaviour of its own -- SURVEY 8a M5 -- so this is an opt-in variant,
// option "cnn_bf16", checked against the oracle with the same rounding: embeddings and filters rounded to bf16
// (round-to-nearest-even), products exact in fp32, fp32 accumulation, bias/ReLU/max-pool/projection in fp32).
//
// Same structure as cnn_fwd.hip (windows are contiguous runs of the [T][Ep] tile in LDS, no im2col), on
// v_mfma_f32_32x32x16_bf16: a lane's A operand is ONE 16-byte LDS read (8 bf16 = 8 consecutive k of its position
// row), its B operand one 16-byte global read; one MFMA covers 16 k (8 fp32 MFMAs' worth) in 32 cycles per SIMD --
// 16x the fp32 matrix rate, so the kernel lives on LDS bandwidth (4 KiB of window fragments per 4 MFMAs per wave)
// and the gather / epilogue, not on the matrix pipe.
#include "sse_kernels.h"

#define CB_THREADS 512
#define CB_SG 4  // sequences per wave work item (one filter fragment feeds 4 MFMAs)

typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct CnnBf16Args {
  const int32_t *ids;        // [B][T]
  const unsigned short *emb; // [V][Ep] bf16 bits, Ep % 8 == 0
  const unsigned short *Wc;  // per width: tiles of [32 filters][16 k] bf16 in fragment order, widths concatenated
  const float *bias;         // [576]
  float *featp;              // frag32(rows = b, red = feature): [ceil(B/32)][72][256]
  float *feat_rm;            // TRAIN: [B][576] row-major pooled features
  int32_t *pos;              // TRAIN: [B][576] arg-max positions (first maximum)
  int32_t *err;
  int32_t B, T, V, Ep, wbytes;
};

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {  // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

namespace {
__constant__ int q_fs[4] = {2, 3, 4, 5};
__constant__ int q_nt[4] = {8, 4, 4, 2};
__constant__ int q_foff[4] = {0, 256, 384, 512};
}  // namespace

// TRAIN: the running maximum is the 64-bit key (value bits << 32 | ~position) of cnn_fwd.hip, so the arg-max tape
// the backward pass needs comes out of the same atomicMax (equal values keep the first position).
template <int NB, bool TRAIN>
__global__ __launch_bounds__(CB_THREADS) void conv_pool_bf16_kernel(CnnBf16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];  // [NB][T][Ep] bf16 + 5*Ep pad | feat | counter
  const int tid = threadIdx.x, lane = tid & 63;
  const int T = a.T, Ep = a.Ep, E8 = Ep / 8;
  const int b0 = blockIdx.x * NB;
  int *feat = reinterpret_cast<int *>(xs + (size_t)NB * T * Ep + 5 * Ep);  // running max as int bits (values >= 0)
  unsigned long long *featk = reinterpret_cast<unsigned long long *>(feat);  // TRAIN: 64-bit keys (8-byte aligned: see cnn_bf16_lds_nb)
  int *s_next = feat + NB * 576 * (TRAIN ? 2 : 1);

  for (int i = tid; i < NB * T * E8; i += CB_THREADS) {  // 16 bytes = 8 bf16 per thread-step
    const int q = i % E8, tok = i / E8;
    const int b = b0 + tok / T, t = tok % T;
    int id = (b < a.B) ? a.ids[(size_t)b * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    reinterpret_cast<f32x4 *>(xs)[i] = *reinterpret_cast<const f32x4 *>(a.emb + (size_t)id * Ep + q * 8);
  }
  for (int i = tid; i < 5 * Ep; i += CB_THREADS) xs[(size_t)NB * T * Ep + i] = 0;
  for (int i = tid; i < NB * 576 * (TRAIN ? 2 : 1); i += CB_THREADS) feat[i] = 0;
  if (tid == 0) *s_next = 0;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.Wc), 0, a.wbytes, 0x00020000);
  const int voff = lane * 16;
  const int NSG = NB / CB_SG;
  const int PT = (T + 31) / 32;
  const int n_items = 18 * NSG * PT;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(s_next, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= n_items) break;
    const int sg = item % NSG, pt = (item / NSG) % PT;
    int tile = item / (NSG * PT), wi = 3, woff_tiles = 0;  // tile counted from the widest filter down
    while (tile >= q_nt[wi]) {
      tile -= q_nt[wi];
      --wi;
    }
    for (int j = 0; j < wi; ++j) woff_tiles += q_nt[j] * ((q_fs[j] * Ep + 15) / 16);
    const int fs = q_fs[wi], KG = (fs * Ep + 15) / 16, P = T - fs + 1;  // a last half group reads 8 bf16 past the window: weights 0
    const int wsoff = (woff_tiles + tile * KG) * 1024;
    const float bias = a.bias[q_foff[wi] + tile * 32 + (lane & 31)];
    // lane: position row (lane & 31) of the tile, k half (lane >> 5) -> 8 consecutive bf16
    const unsigned short *xa = xs + (size_t)(sg * CB_SG) * T * Ep + (size_t)(pt * 32 + (lane & 31)) * Ep + (lane >> 5) * 8;
    f32x16 acc[CB_SG];
#pragma unroll
    for (int s = 0; s < CB_SG; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
    auto wl = [&](int kg) -> bf16x8 {
      return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, wsoff + kg * 1024, 0));
    };
    auto al = [&](int s, int kg) -> bf16x8 {
      return *reinterpret_cast<const bf16x8 *>(xa + (size_t)s * T * Ep + kg * 16);
    };
    // two named operand sets: group kg+1's fragments are in flight under group kg's 4 MFMAs
    bf16x8 bx = wl(0), by, ax[CB_SG], ay[CB_SG];
#pragma unroll
    for (int s = 0; s < CB_SG; ++s) ax[s] = al(s, 0);
    int kg = 0;
    for (; kg + 1 < KG; kg += 2) {
      by = wl(kg + 1);
#pragma unroll
      for (int s = 0; s < CB_SG; ++s) ay[s] = al(s, kg + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CB_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[s], bx, acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int k2 = (kg + 2 < KG) ? kg + 2 : kg;
      bx = wl(k2);
#pragma unroll
      for (int s = 0; s < CB_SG; ++s) ax[s] = al(s, k2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CB_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[s], by, acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kg < KG) {
#pragma unroll
      for (int s = 0; s < CB_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[s], bx, acc[s], 0, 0, 0);
    }
    // bias + ReLU + max over this tile's valid positions (row = position, column = filter)
#pragma unroll
    for (int s = 0; s < CB_SG; ++s) {
      if constexpr (TRAIN) {
        unsigned long long key = 0;  // below every valid position's key
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = pt * 32 + mfma_row(r, lane);
          const float v = fmaxf(acc[s][r] + bias, 0.0f);
          const unsigned long long kv = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~p);
          if (p < P && kv > key) key = kv;
        }
        const unsigned long long other = __shfl_xor(key, 32);
        if (other > key) key = other;
        if (lane < 32) atomicMax(&featk[(sg * CB_SG + s) * 576 + q_foff[wi] + tile * 32 + lane], key);
      } else {
        float m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = pt * 32 + mfma_row(r, lane);
          const float v = fmaxf(acc[s][r] + bias, 0.0f);
          m = fmaxf(m, (p < P) ? v : 0.0f);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 32) atomicMax(&feat[(sg * CB_SG + s) * 576 + q_foff[wi] + tile * 32 + lane], __float_as_int(m));
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < NB * 576; i += CB_THREADS) {
    const int j = i % 576, b = b0 + i / 576;
    if (b < a.B) {
      float v;
      if constexpr (TRAIN) {
        const unsigned long long key = featk[i];
        v = __uint_as_float((unsigned)(key >> 32));
        a.feat_rm[(size_t)b * 576 + j] = v;
        a.pos[(size_t)b * 576 + j] = (int32_t)(~(unsigned)key);
      } else {
        v = __int_as_float(feat[i]);
      }
      a.featp[((size_t)(b >> 5) * 72 + (j >> 3)) * 256 + ((((j >> 2) & 1) * 32 + (b & 31)) << 2) + (j & 3)] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Projection tail of the bf16 variant on the bf16 matrix pipe (round 5): [B,576] . [576,S] + row l2-normalise with SPLIT
// operands -- x = hi + lo (bf16 each, x to 2^-17), a*b as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulation: the fp32
// projection's result to ~4e-6 relative at 1/5 of its matrix-pipe time (proj_norm_kernel: 288 v_mfma_f32_32x32x2_f32 of
// 64 cycles per 32 x 32 tile; here 108 v_mfma_f32_32x32x16_bf16 of 32).  It was 27 % of the bf16 encode for 5 % of its flops.
// A = the pooled features as conv_pool_bf16_kernel leaves them (frag32 fp32: k-group of 8 = two float4 per (row, k half));
// a lane (row, k octet) takes both float4 of ITS octet's group and splits them in registers (v_cvt_pk_bf16_f32).
// B = the projection matrix as split frag16 blocks [n tile][36 groups of 16][hi | lo][512] (pack_kn_x3_kernel).
struct ProjX3Args {
  const float *featp;          // frag32(rows = b, red = feature): [ceil(B/32)][72][256]
  const unsigned short *Mx3;   // [NTS][36][2][512]
  float *out;                  // [B][S]
  int32_t B, S, NTS, normalize;
};

__global__ __launch_bounds__(256) void proj_norm_x3_kernel(ProjX3Args a) {
  __shared__ __attribute__((aligned(16))) float red[32 * 4];
  const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
  const int mt = blockIdx.x;
  constexpr int PT = 4;   // up to Sp = 512
  constexpr int KG = 36;  // 576 features / 16
  const int row = lane & 31, oct = lane >> 5;
  // this lane's octet of group kg: frag32 block 2*kg + oct, float4 of (k half 0, row) and (k half 1, row)
  const float *ap = a.featp + (size_t)mt * 72 * 256 + (size_t)oct * 256 + row * 4;
  f32x16 pacc[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
  const sse_u32x4 *mp = reinterpret_cast<const sse_u32x4 *>(a.Mx3) + lane;
  // two named operand sets: group kg+1's fragments (two float4 of features, eight 16-byte filter fragments) are in flight under
  // group kg's twelve MFMAs -- without it every group waited out an L2 round trip (36 in a row: 60 us for 11 us of matrix work)
  auto lda = [&](int kg, f32x4 &x0, f32x4 &x1) {
    x0 = *reinterpret_cast<const f32x4 *>(ap + (size_t)kg * 512);
    x1 = *reinterpret_cast<const f32x4 *>(ap + (size_t)kg * 512 + 128);
  };
  auto ldb = [&](int kg, sse_u32x4 (&bh)[PT], sse_u32x4 (&bl)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int nt = wn + 4 * i;
      const sse_u32x4 *bp = mp + ((size_t)(nt < a.NTS ? nt : 0) * KG + kg) * 128;  // 2 blocks x 64 lanes of 16 bytes
      bh[i] = bp[0];
      bl[i] = bp[64];
    }
  };
  auto mma = [&](const f32x4 &x0, const f32x4 &x1, const sse_u32x4 (&bh)[PT], const sse_u32x4 (&bl)[PT]) {
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    sse_u32x4 ah, al;
    sse_split8(v, ah, al);
    const bf16x8 a_hi = __builtin_bit_cast(bf16x8, ah), a_lo = __builtin_bit_cast(bf16x8, al);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      if (wn + 4 * i < a.NTS) {
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bh[i]), b_lo = __builtin_bit_cast(bf16x8, bl[i]);
        pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, pacc[i], 0, 0, 0);
        pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, pacc[i], 0, 0, 0);
        pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, pacc[i], 0, 0, 0);
      }
    }
  };
  f32x4 ax0, ax1, ay0, ay1;
  sse_u32x4 bxh[PT], bxl[PT], byh[PT], byl[PT];
  lda(0, ax0, ax1);
  ldb(0, bxh, bxl);
  for (int kg = 0; kg < KG; kg += 2) {  // KG is even
    lda(kg + 1, ay0, ay1);
    ldb(kg + 1, byh, byl);
    __builtin_amdgcn_sched_barrier(0);
    mma(ax0, ax1, bxh, bxl);
    __builtin_amdgcn_sched_barrier(0);
    const int k2 = (kg + 2 < KG) ? kg + 2 : kg;
    lda(k2, ax0, ax1);
    ldb(k2, bxh, bxl);
    __builtin_amdgcn_sched_barrier(0);
    mma(ay0, ay1, byh, byl);
    __builtin_amdgcn_sched_barrier(0);
  }
  float ss[16], scale[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ss[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < PT; ++i) ss[r] += pacc[i][r] * pacc[i][r];  // (tiles past NTS hold zeros)
  }
  if (a.normalize) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = ss[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if ((lane & 31) == 0) red[mfma_row(r, lane) * 4 + wn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x4 p = *reinterpret_cast<const f32x4 *>(red + mfma_row(r, lane) * 4);
      scale[r] = 1.0f / sqrtf(fmaxf((p[0] + p[1]) + (p[2] + p[3]), 1e-12f));
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.0f;
  }
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + 4 * i, col = nt * 32 + (lane & 31);
    if (nt < a.NTS && col < a.S) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rw = mt * 32 + mfma_row(r, lane);
        if (rw < a.B) a.out[(size_t)rw * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

// M [576][S] (fp32 master) -> split frag16 blocks: (nt, kg) -> [hi 512 | lo 512]; lane l (column nt*32 + (l & 31), k octet l >> 5)
// owns k = kg*16 + 8*(l >> 5) + i
__global__ void pack_kn_x3_kernel(const float *__restrict__ Mv, int K, int N, int64_t total, unsigned short *__restrict__ out) {
  const int KG = K / 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int idx = (int)(i & 7), l = (int)((i >> 3) & 63);
    const int64_t blk = i >> 9;  // (nt * KG + kg)
    const int kg = (int)(blk % KG), nt = (int)(blk / KG);
    const int n = nt * 32 + (l & 31), k = kg * 16 + 8 * (l >> 5) + idx;
    const float x = (n < N && k < K) ? Mv[(size_t)k * N + n] : 0.0f;
    const unsigned short hi = f32_to_bf16(x);
    out[blk * 1024 + (size_t)l * 8 + idx] = hi;
    out[blk * 1024 + 512 + (size_t)l * 8 + idx] = f32_to_bf16(x - __uint_as_float((unsigned)hi << 16));
  }
}

size_t cnn_proj_x3_elems(int S) { return (size_t)((S + 31) / 32) * 36 * 1024; }

hipError_t launch_pack_cnn_proj_x3(const float *Mv, int S, unsigned short *Mx3, hipStream_t stream) {
  const int64_t total = (int64_t)((S + 31) / 32) * 36 * 512;
  hipLaunchKernelGGL(pack_kn_x3_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, Mv, 576, S, total, Mx3);
  return hipGetLastError();
}

hipError_t launch_cnn_proj_x3(const float *featp, const unsigned short *Mx3, float *out, int B, int S, int normalize, hipStream_t stream) {
  if (S > 512) return hipErrorInvalidValue;
  ProjX3Args p{featp, Mx3, out, B, S, (S + 31) / 32, normalize};
  hipLaunchKernelGGL(proj_norm_x3_kernel, dim3((B + 31) / 32), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// rows [R][C] fp32 -> [R][Cp] bf16 (zero padded columns)
__global__ void to_bf16_rows_kernel(const float *__restrict__ in, int64_t R, int C, int Cp, unsigned short *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * Cp; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    out[i] = (c < C) ? f32_to_bf16(in[(i / Cp) * C + c]) : (unsigned short)0;
  }
}

// conv filter W [fs][E][1][nf] (row-major [fs*E][nf]) -> per 32-filter tile and 16-k group one 1-KiB block in
// operand order: lane l (filter l&31, k half l>>5) owns 8 consecutive k' = d*Ep + e
__global__ void pack_conv_bf16_kernel(const float *__restrict__ W, int fs, int E, int Ep, int nf, int64_t total,
                                      unsigned short *__restrict__ out) {
  const int KG = (fs * Ep + 15) / 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    const int l = (int)((i >> 3) & 63);
    const int64_t blk = i >> 9;
    const int kg = (int)(blk % KG), nt = (int)(blk / KG);
    const int f = nt * 32 + (l & 31);
    const int kp = kg * 16 + (l >> 5) * 8 + j;
    const int d = kp / Ep, e = kp % Ep;
    out[i] = (f < nf && e < E && d < fs) ? f32_to_bf16(W[(size_t)(d * E + e) * nf + f]) : (unsigned short)0;
  }
}

size_t cnn_bf16_packed_weight_elems(int Ep) {
  static const int fs[4] = {2, 3, 4, 5}, nt[4] = {8, 4, 4, 2};
  size_t n = 0;
  for (int i = 0; i < 4; ++i) n += (size_t)nt[i] * ((fs[i] * Ep + 15) / 16) * 512;
  return n;
}

static size_t cnn_bf16_lds_nb(int T, int Ep, int nb, int train) {  // Ep % 8 == 0: the feat region starts 8-byte aligned
  return (size_t)(nb * T * Ep + 5 * Ep) * sizeof(unsigned short) + (size_t)nb * 576 * sizeof(int) * (train ? 2 : 1) + 16;
}

size_t cnn_bf16_lds_bytes(int T, int Ep, int train) {
  return cnn_bf16_lds_nb(T, Ep, cnn_bf16_lds_nb(T, Ep, 8, train) <= 160 * 1024 ? 8 : 4, train);
}

hipError_t launch_cnn_bf16_pack(const float *emb, int64_t V, int E, int Ep, unsigned short *emb_bf16, const float *const W[4],
                                unsigned short *Wc, hipStream_t stream) {
  static const int fs[4] = {2, 3, 4, 5}, nf[4] = {256, 128, 128, 64};
  const int64_t ne = V * Ep;
  hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((int)((ne + 255) / 256 < 8192 ? (ne + 255) / 256 : 8192)), dim3(256), 0, stream,
                     emb, V, E, Ep, emb_bf16);
  size_t off = 0;
  for (int i = 0; i < 4; ++i) {
    const int64_t total = (int64_t)(nf[i] / 32) * ((fs[i] * Ep + 15) / 16) * 512;
    hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, W[i], fs[i], E, Ep, nf[i],
                       total, Wc + off);
    off += (size_t)total;
  }
  return hipGetLastError();
}

// conv + pool in bf16 storage; the projection tail is the fp32 proj_norm_kernel of cnn_fwd.hip (launch_cnn_proj).
// feat_rm / pos non-null: training forward (row-major features + arg-max tape as launch_cnn_fwd's).
template <int NB, bool TRAIN>
static hipError_t launch_conv_pool_bf16(const CnnBf16Args &a, size_t lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_pool_bf16_kernel<NB, TRAIN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((conv_pool_bf16_kernel<NB, TRAIN>), dim3((a.B + NB - 1) / NB), dim3(CB_THREADS), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_cnn_fwd_bf16(const int32_t *ids, const unsigned short *emb_bf16, const unsigned short *Wc, const float *bias,
                               float *featp, int32_t *err, int B, int T, int V, int Ep, float *feat_rm, int32_t *pos,
                               hipStream_t stream) {
  const int train = (feat_rm != nullptr);
  const int nb = cnn_bf16_lds_nb(T, Ep, 8, train) <= 160 * 1024 ? 8 : 4;
  const size_t lds = cnn_bf16_lds_nb(T, Ep, nb, train);
  CnnBf16Args a{ids, emb_bf16, Wc, bias, featp, feat_rm, pos, err, B, T, V, Ep,
                (int32_t)(cnn_bf16_packed_weight_elems(Ep) * sizeof(unsigned short))};
  if (train) return nb == 8 ? launch_conv_pool_bf16<8, true>(a, lds, stream) : launch_conv_pool_bf16<4, true>(a, lds, stream);
  return nb == 8 ? launch_conv_pool_bf16<8, false>(a, lds, stream) : launch_conv_pool_bf16<4, false>(a, lds, stream);
}
