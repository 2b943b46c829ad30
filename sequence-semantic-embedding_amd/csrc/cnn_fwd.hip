// Text-CNN sequence encoder forward for gfx950 (network_mode 'source_only_cnn').
//
// Replaces sse_model.py:179-211: for filter widths (2,3,4,5) with (256,128,128,64)
// filters, a VALID convolution over [T,E] + bias + ReLU + max over the T-fs+1
// positions, concat -> [B,576], then `. src_M` and l2_normalize (:282).
//
// conv_pool_kernel: the convolution is a GEMM whose A operand needs no im2col --
// with the embedded sequence stored row-major [T][Ep] in LDS, the window of
// position p is the CONTIGUOUS run starting at p*Ep (k = d*Ep + e), so a lane's
// A fragment is one 16-byte LDS read.  M = positions (32 per MFMA tile), N =
// filters, K = fs*Ep; bias + ReLU + running max over positions are fused on the
// accumulators; [B,T,E] and the conv outputs never touch HBM.
// proj_norm_kernel: [B,576] . [576,S] + row l2-normalise.
#include "sse_kernels.h"

#define CNN_THREADS 512
// sequences per workgroup: 8, or 4 when 8 embedded sequences (+ the training keys) do not fit the 160 KB of LDS
// (e.g. the reference's default T = 80 with the arg-max tape)
#define CNN_SG 4   // sequences per wave work item (one B fragment feeds 4 MFMAs)

struct CnnArgs {
  const int32_t *ids;  // [B][T]
  const float *emb;    // [V][Ep]
  const float *Wc;     // per width: frag32(rows = filter, red = k' = d*Ep + e), widths concatenated
  const float *bias;   // [576]
  float *featp;        // frag32(rows = b, red = feature): [ceil(B/32)][72][256]
  int32_t *err;
  int32_t B, T, V, Ep, wbytes;
  float *feat_rm;      // TRAIN: [B][576] pooled features, row-major
  int32_t *pos;        // TRAIN: [B][576] first arg-max position of each pooled feature
};

__constant__ int c_fs[4] = {2, 3, 4, 5};
__constant__ int c_nt[4] = {8, 4, 4, 2};          // 32-filter tiles per width (256,128,128,64 filters)
__constant__ int c_foff[4] = {0, 256, 384, 512};  // feature offset of each width in the 576-vector

// TRAIN additionally records WHERE each maximum sits (the backward pass routes the gradient there):
// the running maximum becomes a 64-bit key (value bits << 32 | ~position), so equal values keep the
// first position, as numpy/TF arg-max do (all-PAD windows tie exactly).
template <bool TRAIN, int CNN_NB>
__global__ __launch_bounds__(CNN_THREADS) void conv_pool_kernel(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][T][Ep] + 5*Ep pad + work counter
  const int tid = threadIdx.x, lane = tid & 63;
  const int T = a.T, Ep = a.Ep, E4 = Ep / 4;
  const int b0 = blockIdx.x * CNN_NB;
  // all LDS in the dynamic region (16-B aligned base): xs | 5*Ep pad | feat[NB][576] | work counter
  int *feat = reinterpret_cast<int *>(xs + CNN_NB * T * Ep + 5 * Ep);  // running max as int bits (values >= 0)
  unsigned long long *featk = reinterpret_cast<unsigned long long *>(feat);  // TRAIN: 64-bit keys
  int *s_next = feat + CNN_NB * 576 * (TRAIN ? 2 : 1);

  // stage the embedded sequences (row-major, Ep-padded rows: 16-byte aligned windows)
  for (int i = tid; i < CNN_NB * T * E4; i += CNN_THREADS) {
    const int q = i % E4, tok = i / E4;
    const int b = b0 + tok / T, t = tok % T;
    int id = (b < a.B) ? a.ids[(size_t)b * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    reinterpret_cast<f32x4 *>(xs)[i] = *reinterpret_cast<const f32x4 *>(a.emb + (size_t)id * Ep + q * 4);
  }
  for (int i = tid; i < 5 * Ep; i += CNN_THREADS) xs[CNN_NB * T * Ep + i] = 0.0f;  // windows of the last rows read past the tile
  for (int i = tid; i < CNN_NB * 576 * (TRAIN ? 2 : 1); i += CNN_THREADS) feat[i] = 0;
  if (tid == 0) *s_next = 0;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Wc), 0, a.wbytes, 0x00020000);
  const int voff = lane * 16;
  // work items, most expensive first: (width 5..2) x (filter tile) x (position tile) x (sequence
  // group); waves pull them from an LDS counter; partial maxima meet in feat[] via atomicMax
  const int NSG = CNN_NB / CNN_SG;
  const int PT = (T + 31) / 32;  // position tiles (positions >= P are masked)
  const int n_items = 18 * NSG * PT;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(s_next, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= n_items) break;
    const int sg = item % NSG, pt = (item / NSG) % PT;
    int tile = item / (NSG * PT), wi = 3, woff_tiles = 0;  // tile counted from the widest filter down
    while (tile >= c_nt[wi]) {
      tile -= c_nt[wi];
      --wi;
    }
    for (int j = 0; j < wi; ++j) woff_tiles += c_nt[j] * ((c_fs[j] * Ep + 7) / 8);
    const int fs = c_fs[wi], KG = (fs * Ep + 7) / 8, P = T - fs + 1;  // a last half k-group reads 4 floats past the window: their weights are 0
    const int wsoff = (woff_tiles + tile * KG) * 1024;  // byte offset of this filter tile in the packed weights
    const float bias = a.bias[c_foff[wi] + tile * 32 + (lane & 31)];
    const float *xb = xs + (size_t)(sg * CNN_SG) * T * Ep + (lane & 31) * Ep + (lane >> 5) * 4;
    {
      f32x16 acc[CNN_SG];
#pragma unroll
      for (int s = 0; s < CNN_SG; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.0f;
      const float *xa = xb + (size_t)pt * 32 * Ep;
      // k-loop, hand software-pipelined (two named operand sets, no copies): filter fragment
      // of k-group kg+1 (global, via buffer descriptor) and the 4 window fragments (LDS) are in
      // flight while kg's 16 MFMAs issue
      auto wl = [&](int kg) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, wsoff + kg * 1024, 0));
      };
      f32x4 bx = wl(0), by, ax[CNN_SG], ay[CNN_SG];
#pragma unroll
      for (int s = 0; s < CNN_SG; ++s) ax[s] = *reinterpret_cast<const f32x4 *>(xa + (size_t)s * T * Ep);
      __builtin_amdgcn_s_setprio(1);
      int kg = 0;
      for (; kg + 1 < KG; kg += 2) {
        by = wl(kg + 1);
#pragma unroll
        for (int s = 0; s < CNN_SG; ++s) ay[s] = *reinterpret_cast<const f32x4 *>(xa + (size_t)s * T * Ep + (kg + 1) * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < CNN_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s][e], bx[e], acc[s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < KG) ? kg + 2 : kg;
        bx = wl(k2);
#pragma unroll
        for (int s = 0; s < CNN_SG; ++s) ax[s] = *reinterpret_cast<const f32x4 *>(xa + (size_t)s * T * Ep + k2 * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < CNN_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[s][e], by[e], acc[s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < KG) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < CNN_SG; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s][e], bx[e], acc[s], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      // bias + ReLU + max over this tile's valid positions (row = position, column = filter)
#pragma unroll
      for (int s = 0; s < CNN_SG; ++s) {
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
          if (lane < 32) atomicMax(&featk[(sg * CNN_SG + s) * 576 + c_foff[wi] + tile * 32 + lane], key);
        } else {
          float m = 0.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int p = pt * 32 + mfma_row(r, lane);
            const float v = fmaxf(acc[s][r] + bias, 0.0f);
            m = fmaxf(m, (p < P) ? v : 0.0f);
          }
          m = fmaxf(m, __shfl_xor(m, 32));
          if (lane < 32) atomicMax(&feat[(sg * CNN_SG + s) * 576 + c_foff[wi] + tile * 32 + lane], __float_as_int(m));
        }
      }
    }
  }
  __syncthreads();
  // features -> global, frag32(rows = b, red = feature)
  for (int i = tid; i < CNN_NB * 576; i += CNN_THREADS) {
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

// out[b][s] = sum_j feat[b][j] * M[j][s]  (+ row l2-normalise); one 32-row tile per workgroup of 4 waves
struct ProjArgs {
  const float *featp;  // frag32(rows = b, red = j), KG k-groups
  const float *Mp;     // frag32(rows = s, red = j)
  float *out;          // [B][S]
  int32_t B, S, KG, NTS, normalize;
};

__global__ __launch_bounds__(256) void proj_norm_kernel(ProjArgs a) {
  __shared__ __attribute__((aligned(16))) float red[32 * 4];
  const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
  const int mt = blockIdx.x;
  constexpr int PT = 4;  // up to Sp = 512
  const float *ap = a.featp + (size_t)mt * a.KG * 256 + lane * 4;
  f32x16 pacc[PT];
  float ss[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ss[r] = 0.0f;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + 4 * i;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
    if (nt < a.NTS) {
      const float *mp = a.Mp + (size_t)nt * a.KG * 256 + lane * 4;
      // two named operand sets: the fragments of k-groups kg+1 / kg+2 are in flight under kg's MFMAs
      f32x4 ax = *reinterpret_cast<const f32x4 *>(ap), bx = *reinterpret_cast<const f32x4 *>(mp), ay, by;
      int kg = 0;
      for (; kg + 1 < a.KG; kg += 2) {
        ay = *reinterpret_cast<const f32x4 *>(ap + (kg + 1) * 256);
        by = *reinterpret_cast<const f32x4 *>(mp + (kg + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < a.KG) ? kg + 2 : kg;
        ax = *reinterpret_cast<const f32x4 *>(ap + k2 * 256);
        bx = *reinterpret_cast<const f32x4 *>(mp + k2 * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], by[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < a.KG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ss[r] += pacc[i][r] * pacc[i][r];
    }
  }
  float scale[16];
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
        const int row = mt * 32 + mfma_row(r, lane);
        if (row < a.B) a.out[(size_t)row * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

// conv filter W [fs][E][1][nf] (row-major [fs*E][nf]) -> frag32(rows = filter, red = k' = d*Ep + e)
__global__ void pack_conv_kernel_k(const float *__restrict__ W, int fs, int E, int Ep, int nf, int64_t total4,
                                   f32x4 *__restrict__ out) {
  const int KG = (fs * Ep + 7) / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kg = (int)(blk % KG), nt = (int)(blk / KG);
    const int f = nt * 32 + (l & 31);
    f32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const int kp = kg * 8 + (l >> 5) * 4 + e4;
      const int d = kp / Ep, e = kp % Ep;
      if (f < nf && e < E && d < fs) v[e4] = W[(size_t)(d * E + e) * nf + f];
    }
    out[i] = v;
  }
}

static size_t cnn_lds_bytes_nb(int T, int Ep, int train, int nb) {
  return (size_t)(nb * T * Ep + 5 * Ep + nb * 576 * (train ? 2 : 1)) * sizeof(float) + 16;
}

static int cnn_pick_nb(int T, int Ep, int train) { return cnn_lds_bytes_nb(T, Ep, train, 8) <= 160 * 1024 ? 8 : 4; }

size_t cnn_lds_bytes(int T, int Ep, int train) { return cnn_lds_bytes_nb(T, Ep, train, cnn_pick_nb(T, Ep, train)); }

size_t cnn_packed_weight_floats(int Ep) {
  static const int fs[4] = {2, 3, 4, 5}, nt[4] = {8, 4, 4, 2};
  size_t n = 0;
  for (int i = 0; i < 4; ++i) n += (size_t)nt[i] * ((fs[i] * Ep + 7) / 8) * 256;
  return n;
}

hipError_t launch_pack_conv(const float *const W[4], int E, int Ep, float *out, hipStream_t stream) {
  static const int fs[4] = {2, 3, 4, 5}, nf[4] = {256, 128, 128, 64};
  size_t off = 0;
  for (int i = 0; i < 4; ++i) {
    const int KG = (fs[i] * Ep + 7) / 8;
    const int64_t total4 = (int64_t)(nf[i] / 32) * KG * 64;
    hipLaunchKernelGGL(pack_conv_kernel_k, dim3((int)((total4 + 255) / 256)), dim3(256), 0, stream, W[i], fs[i], E, Ep,
                       nf[i], total4, reinterpret_cast<f32x4 *>(out + off));
    off += (size_t)(nf[i] / 32) * KG * 256;
  }
  return hipGetLastError();
}

// projection tail alone (the bf16-storage convolution of cnn_fwd_bf16.hip feeds the same fp32 tail)
hipError_t launch_cnn_proj(const float *featp, const float *Mp, float *out, int B, int S, int normalize, hipStream_t stream) {
  ProjArgs p{featp, Mp, out, B, S, 72, (S + 31) / 32, normalize};
  hipLaunchKernelGGL(proj_norm_kernel, dim3((B + 31) / 32), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_cnn_fwd(const int32_t *ids, const float *emb, const float *Wc, const float *bias, const float *Mp,
                          float *featp, float *out, int32_t *err, int B, int T, int V, int Ep, int S, int normalize,
                          float *feat_rm, int32_t *pos, hipStream_t stream) {
  const bool train = feat_rm != nullptr;
  const int nb = cnn_pick_nb(T, Ep, train);
  const size_t lds = cnn_lds_bytes_nb(T, Ep, train, nb);
  const void *fn = train ? (nb == 8 ? reinterpret_cast<const void *>(conv_pool_kernel<true, 8>)
                                    : reinterpret_cast<const void *>(conv_pool_kernel<true, 4>))
                         : (nb == 8 ? reinterpret_cast<const void *>(conv_pool_kernel<false, 8>)
                                    : reinterpret_cast<const void *>(conv_pool_kernel<false, 4>));
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  CnnArgs a{ids, emb, Wc, bias, featp, err, B, T, V, Ep, (int32_t)(cnn_packed_weight_floats(Ep) * sizeof(float)),
            feat_rm, pos};
  const dim3 grid((B + nb - 1) / nb), block(CNN_THREADS);
  if (train && nb == 8) hipLaunchKernelGGL((conv_pool_kernel<true, 8>), grid, block, lds, stream, a);
  else if (train) hipLaunchKernelGGL((conv_pool_kernel<true, 4>), grid, block, lds, stream, a);
  else if (nb == 8) hipLaunchKernelGGL((conv_pool_kernel<false, 8>), grid, block, lds, stream, a);
  else hipLaunchKernelGGL((conv_pool_kernel<false, 4>), grid, block, lds, stream, a);
  ProjArgs p{featp, Mp, out, B, S, 72, (S + 31) / 32, normalize};
  hipLaunchKernelGGL(proj_norm_kernel, dim3((B + 31) / 32), dim3(256), 0, stream, p);
  return hipGetLastError();
}
