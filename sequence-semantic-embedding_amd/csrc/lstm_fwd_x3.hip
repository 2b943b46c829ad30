// LSTM sequence encoder forward on the bf16 matrix pipe with SPLIT operands ("bf16x3"), gfx950 -- OPT-IN (option
// "lstm_x3"): NOT the exact fp32 arithmetic of lstm_fwd.hip, but inside the encoder tolerance by two orders of magnitude.
//
// v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TF); v_mfma_f32_32x32x16_bf16 at 16x that.  Every fp32 operand is
// written as hi + lo with hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to 2^-18 relative), and a product a*b becomes the
// three bf16 MFMAs  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulated in fp32 -- the dropped a_lo*b_lo term is <= 2^-18 of
// the product.  3 MFMAs of 32 cycles per 16 k against 8 of 64 cycles: 5.3x less matrix-pipe time for a relative error
// of ~4e-6 per product (fp32: 6e-8), i.e. encodings within ~1e-5 of the fp32 path (north-star budget 1e-3; SURVEY 7
// "a bf16-input variant must be parity-qualified against the 1e-3 cosine budget": tests/test_gpu_encode.py).
//
// Structure = lstm_fwd.hip's inference configuration <2,2,1>: one 512-thread workgroup per 64 sequences, wave w owns
// hidden units [32w, 32w+32) for both 32-row tiles, weights are the MFMA's A operand (an accumulator lane owns ONE
// sequence and 16 units), gates in two passes (i,j then f,o), sigmoid(i)*tanh(j) parked in the lane's own slots of the
// other h buffer, the bias through a constant-1 embedding column.  What differs:
//  * operands are bf16 fragments (lane = row & 31, k octet = lane >> 5: 8 consecutive k = 16 bytes), each in a hi and
//    a lo copy: the kernel matrix is packed once per weight update (pack_lstm_x3_kernel), the embedding table split
//    once (split_emb_x3_kernel), h_t is split as it is produced;
//  * the k order of the h part is chosen so that a lane's 16 accumulator registers ARE two whole octets: registers
//    0..7 of lane (n, half) go to octet (group 2*ub, half), registers 8..15 to (group 2*ub + 1, half) -- h_t leaves the
//    wave as 16-byte conflict-free stores, no cross-lane exchange; the packed kernel rows follow the same permutation;
//  * the last step leaves h_T in fp32 (the frag32 layout of lstm_fwd.hip: the same four 16-byte slots per lane), and
//    the projection + l2-normalise tail is the exact fp32 code of lstm_fwd.hip.
// Hp = 256 (cell sizes 129..256: wave = unit block, both row tiles) and Hp = 128 (cell sizes <= 128: wave = unit block
// w & 3, row tile w >> 2).  Left-pad prefix skip as in lstm_fwd.hip, from a table this kernel records itself.
//
// TRAIN variant (option "train_fwd_x3": the forward of the LSTM train step): the operand roles of lstm_fwd.hip's training
// forward -- activations are the A operand, an accumulator lane owns ONE hidden unit and 16 rows -- because the gate tape
// is re-read lane-privately by the BPTT kernel in that layout.  The h part keeps the natural k order (a second packed copy
// of the kernel), h_t is scattered into the operand tiles as 2-byte pieces (4-way bank conflicts; the fp32 training
// forward pays 8-way on 4-byte pieces), the tapes are written as lstm_fwd.hip writes them (tape_a in the split form the
// dK GEMM on the bf16 pipe reads: the stored hi / lo pieces are copied, not re-split), no pad skip.
#include "sse_kernels.h"

#define X3_THREADS 512

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float x3_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float x3_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

__host__ __device__ static inline unsigned short x3_bf16(float f) {  // round to nearest even (finite inputs)
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ static inline float x3_f32(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

// 8 fp32 -> the hi and lo octets (16 bytes each)
__device__ __forceinline__ void x3_split8(const float (&v)[8], u32x4 &hi, u32x4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (hardware converter, see sse_kernels.h)
    unsigned h, l;
    sse_split2(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

// LDS (bytes): x [RT row tiles][KGX][hi|lo][1 KiB] (single-buffered) | h [2 bufs][RT row tiles][KGH = Hp/16][hi|lo][1 KiB] | red
size_t lstm_x3_lds_bytes(int KGX, int Hp, int RT) { return (size_t)RT * KGX * 2048 + (size_t)2 * RT * (Hp / 16) * 2048 + 1024; }

// UBN = unit blocks of 32 hidden units (Hp = 32 * UBN): 8 -> wave = unit block, MT = 2 row tiles; 4 -> MT = 1.
// RT = 32-row tiles per workgroup: 2, or 1 (UBN = 4 only): ONE tile and every unit block split by pass over two waves --
// wave ub computes the i,j pass and hands sigmoid(i)*tanh(j) over through the parking slots (same lane mapping in both
// waves), wave ub + 4 the f,o pass and owns the cell state: half the per-step latency on twice the workgroups, for batches
// that cannot fill the chip (the reference's default training shape: 128 pair rows).
template <int UBN, bool TRAIN, int RT>
__global__ __launch_bounds__(X3_THREADS) void lstm_fwd_x3_kernel(LstmX3Args a) {
  // (RT = 1 with UBN = 8: one tile, wave = unit block, no pass split -- for launches that cannot fill the chip with
  // 64-row workgroups; two such workgroups fit a CU)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  constexpr int KGH = 2 * UBN;  // h groups of 16 units
  constexpr int MT = (UBN == 8 && RT == 2) ? 2 : 1;
  constexpr int ROWS = 32 * RT, NXQ = X3_THREADS / ROWS;
  constexpr bool PSPLIT = (RT == 1 && UBN == 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = (UBN == 8) ? wave : (wave & 3);   // unit block of this wave
  const int mt0 = (UBN == 8 || PSPLIT) ? 0 : (wave >> 2);   // its first row tile
  const bool do_a = !PSPLIT || wave < 4, do_b = !PSPLIT || wave >= 4;
  const int KGX = a.KGX, KG = KGX + KGH, T = a.T;
  const int KGHe = min(KGH, (a.H + 15) / 16);  // h groups that can be non-zero
  unsigned char *xbase = smem3;
  unsigned char *hbase = smem3 + (size_t)RT * KGX * 2048;
  float *red = reinterpret_cast<float *>(hbase + (size_t)2 * RT * KGH * 2048);
  int *pass_flag = reinterpret_cast<int *>(red + 16);  /* LDS atomics, not volatile: see gs_flag_min4 in lstm_fwd_gs.hip */  // PSPLIT: [unit block] = steps whose i,j pass is parked
  if (tid < 4) pass_flag[tid] = 0;
  auto xptr = [&](int mt) -> unsigned char * { return xbase + (size_t)mt * KGX * 2048; };
  auto hptr = [&](int buf, int mt) -> unsigned char * { return hbase + (size_t)(buf * RT + mt) * KGH * 2048; };
  const int b0 = blockIdx.x * ROWS;

  // x gather: thread (row xr, octet xq + 8*i): one 16-byte piece of the hi table and one of the lo table
  const int xr = tid % ROWS, xq = tid / ROWS;
  const bool row_ok = (b0 + xr) < a.B;
  const int32_t *id_row = a.ids + (size_t)(row_ok ? (a.row_map ? a.row_map[b0 + xr] : b0 + xr) : 0) * T;
  const int EP = KGX * 16;  // columns of the split embedding table per copy
  auto fetch_id = [&](int t) -> int {
    int id = row_ok ? id_row[t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };
  auto x_store = [&](int oc, u32x4 hi, u32x4 lo) {
    unsigned char *dst = xptr(xr >> 5) + (size_t)(oc >> 1) * 2048 + (size_t)((oc & 1) * 32 + (xr & 31)) * 16;
    *reinterpret_cast<u32x4 *>(dst) = hi;
    *reinterpret_cast<u32x4 *>(dst + 1024) = lo;
  };
  // left-pad prefix skip: the state after p leading PAD (id 0) steps does not depend on the sequence, so the tile
  // starts at t0 = min over its rows of the leading-PAD count with (h, c) = pad_h / pad_c[t0] (recorded by an all-PAD
  // launch of THIS kernel: same arithmetic); rows beyond B count as all-PAD
  int t0 = 0;
  if (!TRAIN && a.pad_h != nullptr) {
    int lead = T;
    if (row_ok) {
      for (int t = xq; t < T; t += NXQ)
        if (id_row[t] != 0) {
          lead = t;
          break;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    if (lane == 0) red[wave] = __int_as_float(lead);
    __syncthreads();
    lead = T;
    for (int i = 0; i < 8; ++i) lead = min(lead, __float_as_int(red[i]));
    t0 = min(lead, T - 1);
    __syncthreads();
  }
  {
    const unsigned short *src = a.emb16 + (size_t)fetch_id(t0) * 2 * EP;
    for (int oc = xq; oc < 2 * KGX; oc += NXQ)
      x_store(oc, *reinterpret_cast<const u32x4 *>(src + oc * 8), *reinterpret_cast<const u32x4 *>(src + EP + oc * 8));
  }
  const int Hp = 32 * UBN;
  if (t0 > 0) {  // h_{t0-1} = pad_h[t0], the same for every row, split into the operand layout of buffer t0 & 1
    for (int i = tid; i < RT * KGH * 64; i += X3_THREADS) {
      const int sl = i & 63, kgh = (i >> 6) % KGH, mt = i / (64 * KGH);
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = a.pad_h[(size_t)t0 * Hp + (kgh >> 1) * 32 + mfma_row(8 * (kgh & 1) + e, sl)];
      u32x4 hi, lo;
      x3_split8(v8, hi, lo);
      unsigned char *dst = hptr(t0 & 1, mt) + (size_t)kgh * 2048 + sl * 16;
      *reinterpret_cast<u32x4 *>(dst) = hi;
      *reinterpret_cast<u32x4 *>(dst + 1024) = lo;
    }
  }

  f32x16 c[MT];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float c0 = (!TRAIN && t0 > 0) ? a.pad_c[(size_t)t0 * Hp + w * 32 + mfma_row(r, lane)] : 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) c[m][r] = c0;
  }
  __syncthreads();

  // weights: Wx3[unit block][k group][gate][hi|lo][1 KiB] through a buffer descriptor (voffset = 16 * lane)
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(a.Wx3), 0, UBN * KG * 8192, 0x00020000);
  const int wvoff = lane * 16;
  auto wl = [&](int soff) -> bf16x8 { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff, soff, 0)); };

  for (int t = t0; t < T; ++t) {
    const bool have_next = (t + 1) < T;
    u32x4 nhi = {0, 0, 0, 0}, nlo = {0, 0, 0, 0};
    int nid = 0;
    if (have_next) {
      nid = fetch_id(t + 1);
      if (xq < 2 * KGX) {
        const unsigned short *src = a.emb16 + (size_t)nid * 2 * EP;
        nhi = *reinterpret_cast<const u32x4 *>(src + xq * 8);
        nlo = *reinterpret_cast<const u32x4 *>(src + EP + xq * 8);
      }
    }
    const int cur = t & 1, nxt = (t + 1) & 1;
    const unsigned char *xa[MT], *ha[MT];
    unsigned char *hd[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xa[m] = xptr(mt0 + m) + lane * 16;
      ha[m] = hptr(cur, mt0 + m) + lane * 16;
      hd[m] = hptr(nxt, mt0 + m) + (size_t)w * 4096 + lane * 16;  // this lane's four 16-byte slots: groups 2w, 2w+1, hi | lo
    }
    const int kend = (t == 0) ? KGX : KGX + KGHe;  // h_{-1} = 0
    const int wsoff = w * KG * 8192;
    const int NT32 = a.NT32;
    if constexpr (TRAIN) {
      // A-tape for the dK GEMM on the bf16 pipe: [x_t | h_{t-1}] of both row tiles as split frag16 blocks, lane (k' & 31,
      // half) owning rows 16 j + 8 half .. + 7: the hi and lo pieces already sit in the operand tiles (lane = row there)
      const int KT = 2 + Hp / 32, nk = 64 + Hp;
      unsigned short *ta = reinterpret_cast<unsigned short *>(a.tape_a);
      for (int i = tid; i < nk * (ROWS / 8); i += X3_THREADS) {
        const int kp = i % nk, o = i / nk;  // o: rows 8*o .. 8*o+7 of the workgroup's rows
        const int mt = o >> 2, oc = o & 3;
        u32x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
        const bool xpart = kp < 64;
        const int k = xpart ? kp : kp - 64;
        if (!xpart || k < KGX * 16) {
          const unsigned char *src = (xpart ? xptr(mt) : hptr(cur, mt)) + (size_t)(k >> 4) * 2048 +
                                     (size_t)(((k >> 3) & 1) * 32 + oc * 8) * 16 + (k & 7) * 2;
          if (xpart || t > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned h0 = *reinterpret_cast<const unsigned short *>(src + (2 * e) * 16);
              const unsigned h1 = *reinterpret_cast<const unsigned short *>(src + (2 * e + 1) * 16);
              const unsigned l0 = *reinterpret_cast<const unsigned short *>(src + 1024 + (2 * e) * 16);
              const unsigned l1 = *reinterpret_cast<const unsigned short *>(src + 1024 + (2 * e + 1) * 16);
              hi[e] = h0 | (h1 << 16);
              lo[e] = l0 | (l1 << 16);
            }
          }
        }
        const size_t g16 = ((size_t)t * NT32 + blockIdx.x * RT + mt) * 2 + (oc >> 1);
        unsigned short *dst = ta + ((g16 * KT + (kp >> 5)) * 2) * 512 + ((oc & 1) * 32 + (kp & 31)) * 8;
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x4 *>(dst));  // tapes are streamed: keep the weights in L2
        __builtin_nontemporal_store(lo, reinterpret_cast<u32x4 *>(dst + 512));
      }
    }

    // one pass = two gates (byte offset g0 * 2048 inside a k group's 8 KiB), both row tiles, k groups [0, kend):
    // per group 4 weight fragments (L2) + 4 activation fragments (LDS) -> 12 MFMAs; two named operand sets
    auto a_frag = [&](int m, int kg, int hl) -> bf16x8 {
      const unsigned char *p = kg < KGX ? xa[m] + (size_t)kg * 2048 : ha[m] + (size_t)(kg - KGX) * 2048;
      return *reinterpret_cast<const bf16x8 *>(p + hl * 1024);
    };
    // weights are the A operand in inference (accumulator lane = sequence), the B operand in training (lane = unit)
    auto mm = [](const bf16x8 &wf, const bf16x8 &af, const f32x16 &c3) -> f32x16 {
      if constexpr (TRAIN) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf, c3, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, c3, 0, 0, 0);
    };
    auto gemm = [&](int g0, f32x16 (&acc)[MT][2]) {
      bf16x8 wp[2][2], wq[2][2], ap[MT][2], aq[MT][2];  // [gate][hi|lo], [row tile][hi|lo]
      const int base = wsoff + g0 * 2048;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) wp[g][hl] = wl(base + g * 2048 + hl * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) ap[m][hl] = a_frag(m, 0, hl);
      __builtin_amdgcn_s_setprio(1);
      int kg = 0;
      for (; kg + 1 < kend; kg += 2) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl) wq[g][hl] = wl(base + (kg + 1) * 8192 + g * 2048 + hl * 1024);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl) aq[m][hl] = a_frag(m, kg + 1, hl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][g] = mm(wp[g][1], ap[m][0], acc[m][g]);  // w_lo * a_hi
            acc[m][g] = mm(wp[g][0], ap[m][1], acc[m][g]);  // w_hi * a_lo
            acc[m][g] = mm(wp[g][0], ap[m][0], acc[m][g]);  // w_hi * a_hi
          }
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < kend) ? kg + 2 : kg;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl) wp[g][hl] = wl(base + k2 * 8192 + g * 2048 + hl * 1024);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl) ap[m][hl] = a_frag(m, k2, hl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][g] = mm(wq[g][1], aq[m][0], acc[m][g]);
            acc[m][g] = mm(wq[g][0], aq[m][1], acc[m][g]);
            acc[m][g] = mm(wq[g][0], aq[m][0], acc[m][g]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < kend) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][g] = mm(wp[g][1], ap[m][0], acc[m][g]);
            acc[m][g] = mm(wp[g][0], ap[m][1], acc[m][g]);
            acc[m][g] = mm(wp[g][0], ap[m][0], acc[m][g]);
          }
      }
      __builtin_amdgcn_s_setprio(0);
    };

    f32x16 g[MT][2];
    // ---- pass A: gates i, j -> sigmoid(i) * tanh(j), parked in this lane's slots of the other h buffer
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        g[m][0][r] = 0.0f;
        g[m][1][r] = 0.0f;
      }
    if (do_a && (TRAIN || w * 32 < a.H)) gemm(0, g);
    // gate tape (TRAIN), lstm_fwd.hip's layout: [t][tile32][unit block][q = si,tj,sf,so,c][reg / 4][lane][reg % 4]: a lane's
    // registers 4 q4 .. 4 q4 + 3 of a quantity are ONE 16-byte store here and one 16-byte load in the BPTT kernel
    float *tp[MT];
    if constexpr (TRAIN) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        tp[m] = a.tape_g + (((size_t)t * NT32 + blockIdx.x * RT + mt0 + m) * UBN + w) * 5 * 1024 + lane * 4;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        if (!do_a) break;
        f32x4 pij, si4, tj4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          si4[e] = x3_sigmoid(g[m][0][q4 * 4 + e]);
          tj4[e] = x3_tanh(g[m][1][q4 * 4 + e]);
          pij[e] = si4[e] * tj4[e];
        }
        if constexpr (TRAIN) {
          __builtin_nontemporal_store(si4, reinterpret_cast<f32x4 *>(tp[m] + q4 * 256));
          __builtin_nontemporal_store(tj4, reinterpret_cast<f32x4 *>(tp[m] + 1024 + q4 * 256));
        }
        *reinterpret_cast<f32x4 *>(hd[m] + q4 * 1024) = pij;  // lane-private round trip: any layout will do
      }
    // ---- pass B: gates f (+1 in the bias row), o -> c' = c * sigmoid(f) + pij ; h' = tanh(c') * sigmoid(o)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        g[m][0][r] = 0.0f;
        g[m][1][r] = 0.0f;
      }
    if (do_a && !do_b) {  // publish the parked products of this step (LDS operations of a wave complete in order)
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      if (lane == 0) __hip_atomic_store(pass_flag + w, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (do_b && (TRAIN || w * 32 < a.H)) gemm(2, g);
    if (do_b && !do_a) {  // the i,j pass of this unit block comes from the partner wave
      while (__hip_atomic_load(pass_flag + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (!do_b) break;
      float hv[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 pij = *reinterpret_cast<const f32x4 *>(hd[m] + q4 * 1024);
        f32x4 sf4, so4, cn4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = q4 * 4 + e;
          const float sf = x3_sigmoid(g[m][0][r]);
          const float so = x3_sigmoid(g[m][1][r]);
          const float cn = c[m][r] * sf + pij[e];
          c[m][r] = cn;
          hv[r] = x3_tanh(cn) * so;
          sf4[e] = sf;
          so4[e] = so;
          cn4[e] = cn;
          if (!TRAIN && a.rec_h != nullptr && blockIdx.x == 0 && mt0 + m == 0 && (lane & 31) == 0) {
            // sequence 0 of the launch: state after t+1 steps (builds the pad-prefix table)
            a.rec_h[(size_t)(t + 1) * Hp + w * 32 + mfma_row(r, lane)] = hv[r];
            a.rec_c[(size_t)(t + 1) * Hp + w * 32 + mfma_row(r, lane)] = cn;
          }
        }
        if constexpr (TRAIN) {
          __builtin_nontemporal_store(sf4, reinterpret_cast<f32x4 *>(tp[m] + 2048 + q4 * 256));
          __builtin_nontemporal_store(so4, reinterpret_cast<f32x4 *>(tp[m] + 3072 + q4 * 256));
          __builtin_nontemporal_store(cn4, reinterpret_cast<f32x4 *>(tp[m] + 4096 + q4 * 256));
        }
      }
      if (!TRAIN && w * 32 >= a.H) {  // a unit block made only of padding keeps h = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = 0.0f;
      }
      if constexpr (TRAIN) {
        // lane = unit 32 w + (lane & 31), register r = row mfma_row(r, lane): scatter into the (row-major-in-lanes) tiles
        const int unit = w * 32 + (lane & 31);
        unsigned char *tile = hptr(nxt, mt0 + m);
        if (have_next) {
          unsigned char *dst = tile + (size_t)(unit >> 4) * 2048 + (size_t)(((unit >> 3) & 1) * 32) * 16 + (unit & 7) * 2;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {  // rows mfma_row(r), mfma_row(r) + 1: one v_cvt_pk_bf16_f32 pair each for hi and lo
            unsigned hi, lo;
            sse_split2(hv[r], hv[r + 1], hi, lo);
            unsigned char *p = dst + mfma_row(r, lane) * 16;
            *reinterpret_cast<unsigned short *>(p) = (unsigned short)hi;
            *reinterpret_cast<unsigned short *>(p + 16) = (unsigned short)(hi >> 16);
            *reinterpret_cast<unsigned short *>(p + 1024) = (unsigned short)lo;
            *reinterpret_cast<unsigned short *>(p + 1024 + 16) = (unsigned short)(lo >> 16);
          }
        } else {
          float *hf = reinterpret_cast<float *>(tile) + (size_t)(unit >> 3) * 256 + ((((unit >> 2) & 1) * 32) << 2) + (unit & 3);
#pragma unroll
          for (int r = 0; r < 16; ++r) hf[mfma_row(r, lane) << 2] = hv[r];  // h_T in fp32, frag32 layout
        }
      } else if (have_next) {
        // registers 0..7 -> octet (group 2w, this lane's half), 8..15 -> (group 2w + 1, this lane's half); hi | lo
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = hv[8 * j + i];
          u32x4 hi, lo;
          x3_split8(v8, hi, lo);
          *reinterpret_cast<u32x4 *>(hd[m] + j * 2048) = hi;
          *reinterpret_cast<u32x4 *>(hd[m] + j * 2048 + 1024) = lo;
        }
      } else {
        // h_T in fp32, frag32 layout of lstm_fwd.hip (k-group 4w + q4 of the row tile: the same four slots)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<f32x4 *>(hd[m] + q4 * 1024) = f32x4{hv[q4 * 4], hv[q4 * 4 + 1], hv[q4 * 4 + 2], hv[q4 * 4 + 3]};
      }
    }
    __syncthreads();  // h_t complete; x_t and h_{t-1} no longer needed
    if (have_next) {
      if (xq < 2 * KGX) x_store(xq, nhi, nlo);
      for (int oc = xq + NXQ; oc < 2 * KGX; oc += NXQ) {
        const unsigned short *src = a.emb16 + (size_t)nid * 2 * EP;
        x_store(oc, *reinterpret_cast<const u32x4 *>(src + oc * 8), *reinterpret_cast<const u32x4 *>(src + EP + oc * 8));
      }
      __syncthreads();
    }
  }

  if constexpr (TRAIN) {
    // h_T, row-major [Bp][Hp], for dM = h_T^T . d(out)
    for (int i = tid; i < ROWS * Hp; i += X3_THREADS) {
      const int un = i % Hp, b = i / Hp;
      a.h_last[(size_t)(b0 + b) * Hp + un] =
          reinterpret_cast<const float *>(hptr(T & 1, b >> 5))[(size_t)(un >> 3) * 256 + ((((un >> 2) & 1) * 32 + (b & 31)) << 2) + (un & 3)];
    }
  }

  // ---- projection out = h_T . M (+ optional l2_normalize): the fp32 tail of lstm_fwd.hip, 4 waves per row tile
  constexpr int NWR = 8 / RT, PT = 16 / NWR, KGh32 = 4 * UBN;
  const int wn = wave % NWR, wm = wave / NWR;
  const float *hp = reinterpret_cast<const float *>(hptr(T & 1, wm)) + lane * 4;
  f32x16 pacc[PT];
  float *ssq = reinterpret_cast<float *>(hptr((T + 1) & 1, 0));  // [64][16]
  const int KGe = min(KGh32, (a.H + 7) / 8);
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + NWR * i;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
    if (nt < a.NTS) {
      const float *mp = a.Mp + (size_t)nt * KGh32 * 256 + lane * 4;
      f32x4 ax = *reinterpret_cast<const f32x4 *>(hp), bx = *reinterpret_cast<const f32x4 *>(mp), ay, by;
      int kg = 0;
      for (; kg + 1 < KGe; kg += 2) {
        ay = *reinterpret_cast<const f32x4 *>(hp + (kg + 1) * 256);
        by = *reinterpret_cast<const f32x4 *>(mp + (kg + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kg + 2 < KGe) ? kg + 2 : kg;
        ax = *reinterpret_cast<const f32x4 *>(hp + k2 * 256);
        bx = *reinterpret_cast<const f32x4 *>(mp + k2 * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[e], by[e], pacc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kg < KGe) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[e], bx[e], pacc[i], 0, 0, 0);
      }
      if (a.normalize) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = pacc[i][r] * pacc[i][r];
          v += __shfl_xor(v, 1);
          v += __shfl_xor(v, 2);
          v += __shfl_xor(v, 4);
          v += __shfl_xor(v, 8);
          v += __shfl_xor(v, 16);
          if ((lane & 31) == 0) ssq[(wm * 32 + mfma_row(r, lane)) * 16 + nt] = v;
        }
      }
    }
  }
  float scale[16];
  if (a.normalize) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *pr = ssq + (wm * 32 + mfma_row(r, lane)) * 16;
      float tot = 0.0f;
      for (int j = 0; j < a.NTS; ++j) tot += pr[j];
      scale[r] = 1.0f / sqrtf(fmaxf(tot, 1e-12f));
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scale[r] = 1.0f;
  }
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nt = wn + NWR * i;
    const int col = nt * 32 + (lane & 31);
    if (nt < a.NTS && col < a.S) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = b0 + wm * 32 + mfma_row(r, lane);
        if (row < a.B) a.out[(size_t)(a.row_map ? a.row_map[row] : row) * a.S + col] = pacc[i][r] * scale[r];
      }
    }
  }
}

// ---- packing --------------------------------------------------------------------------------------------------------
// Wx3[ub][kg][gate][hi|lo][lane][8]: lane (m = lane & 31 -> unit 32 ub + m, half = lane >> 5), element i -> reduction
// index k:  x part (kg < KGX): k = 16 kg + 8 half + i (k < E: kernel row k; k = E: bias (+1 for the forget gate));
// h part: group 2 ub' + j holds, at (half, i), hidden unit 32 ub' + mfma_row(8 j + i, half) -- the accumulator register
// order, see the file header.
__global__ void pack_lstm_x3_kernel(const float *__restrict__ K, const float *__restrict__ b, int E, int H, int KGX, int KG,
                                    int natural, int64_t total, unsigned short *__restrict__ out) {
  const int N4 = 4 * H;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t rest = idx >> 9;  // ((ub * KG + kg) * 4 + gate)  (hi and lo are written together)
    const int gate = (int)(rest & 3);
    rest >>= 2;
    const int kg = (int)(rest % KG), ub = (int)(rest / KG);
    const int m = lane & 31, half = lane >> 5, unit = ub * 32 + m;
    float wv = 0.0f;
    if (unit < H) {
      const int col = gate * H + unit;
      if (kg < KGX) {
        const int k = kg * 16 + half * 8 + i;
        if (k < E) wv = K[(size_t)k * N4 + col];
        else if (k == E) wv = b[col] + (gate == 2 ? 1.0f : 0.0f);
      } else {
        const int kgh = kg - KGX, r = 8 * (kgh & 1) + i;
        const int uk = natural ? kgh * 16 + half * 8 + i                        // training forward: unit order
                               : (kgh >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // inference: accumulator register order
        if (uk < H) wv = K[(size_t)(E + uk) * N4 + col];
      }
    }
    const unsigned short hi = x3_bf16(wv), lo = x3_bf16(wv - x3_f32(hi));
    const int64_t o = (((int64_t)(ub * KG + kg) * 4 + gate) * 2) * 512 + lane * 8 + i;
    out[o] = hi;
    out[o + 512] = lo;
  }
}

// emb16[id][hi|lo][EP]: columns < E the embedding, column E = 1.0 (carries the bias through the GEMM), rest 0
__global__ void split_emb_x3_kernel(const float *__restrict__ emb, int64_t V, int E, int EP, unsigned short *__restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < V * EP; idx += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(idx % EP);
    const int64_t id = idx / EP;
    const float v = col < E ? emb[id * E + col] : (col == E ? 1.0f : 0.0f);
    const unsigned short hi = x3_bf16(v), lo = x3_bf16(v - x3_f32(hi));
    out[id * 2 * EP + col] = hi;
    out[id * 2 * EP + EP + col] = lo;
  }
}

int lstm_x3_kgx(int E) { return (E + 1 + 15) / 16; }
size_t lstm_x3_weight_elems(int E, int Hp) { return (size_t)(Hp / 32) * (lstm_x3_kgx(E) + Hp / 16) * 4 * 2 * 512; }
size_t lstm_x3_emb_elems(int64_t V, int E) { return (size_t)V * 2 * lstm_x3_kgx(E) * 16; }

hipError_t launch_pack_lstm_x3(const float *K, const float *b, const float *emb, int64_t V, int E, int H, int Hp,
                               unsigned short *Wx3, unsigned short *emb16, int natural_k, hipStream_t stream) {
  const int KGX = lstm_x3_kgx(E), KG = KGX + Hp / 16;
  const int64_t total = (int64_t)(Hp / 32) * KG * 4 * 512;
  hipLaunchKernelGGL(pack_lstm_x3_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, K, b, E, H, KGX, KG, natural_k,
                     total, Wx3);
  if (emb16) {
    const int64_t ne = V * KGX * 16;
    hipLaunchKernelGGL(split_emb_x3_kernel, dim3((int)((ne + 255) / 256 < 8192 ? (ne + 255) / 256 : 8192)), dim3(256), 0, stream,
                       emb, V, E, KGX * 16, emb16);
  }
  return hipGetLastError();
}

template <int UBN, bool TRAIN, int RT>
static hipError_t launch_x3(const LstmX3Args &a, hipStream_t stream) {
  const size_t lds = lstm_x3_lds_bytes(a.KGX, 32 * UBN, RT);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_fwd_x3_kernel<UBN, TRAIN, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int blocks = TRAIN ? a.NT32 / RT : (a.B + 32 * RT - 1) / (32 * RT);
  hipLaunchKernelGGL((lstm_fwd_x3_kernel<UBN, TRAIN, RT>), dim3(blocks), dim3(X3_THREADS), lds, stream, a);
  return hipGetLastError();
}

// a.tape_g != nullptr: the training forward (tapes for NT32 32-row tiles, NT32 even; Wx3 packed with natural_k = 1)
hipError_t launch_lstm_fwd_x3(const LstmX3Args &a, hipStream_t stream) {
  if (a.H < 1 || a.H > 256 || a.B < 1 || a.KGX < 1 || a.KGX > 4) return hipErrorInvalidValue;
  if (a.tape_g != nullptr) {
    if (a.NT32 < 2 || (a.NT32 & 1) || !a.tape_a || !a.h_last) return hipErrorInvalidValue;
    if (a.H > 128)  // 32-row workgroups while they all fit one per CU (measured: 1024 rows 2.44 -> 2.09 ms; at 8192 rows,
                    // 384 tiles, the doubled weight traffic loses: 5.43 vs 5.67 ms)
      return (a.NT32 + a.tiles_elsewhere <= 256) ? launch_x3<8, true, 1>(a, stream) : launch_x3<8, true, 2>(a, stream);
    // cells <= 128: one-tile pass-split workgroups while the launch (and the other encoder's, running beside it) cannot
    // fill the chip with 64-row ones
    return (a.NT32 + a.tiles_elsewhere <= 256) ? launch_x3<4, true, 1>(a, stream) : launch_x3<4, true, 2>(a, stream);
  }
  return a.H > 128 ? launch_x3<8, false, 2>(a, stream) : launch_x3<4, false, 2>(a, stream);
}
