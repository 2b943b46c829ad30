// LSTM sequence encoder forward for a HANDFUL of sequences with the recurrent weights held on chip
// ("persistent RNN"): the single demo / web query of sse_demo.py:121-125 and webserver.py:144-147.
//
// lstm_small.hip runs one workgroup per 4 sequences and streams the whole [(E+H)][4H] kernel matrix (1.25 MB at
// H = 256) from L2 through ONE compute unit every step: ~16 us per step, 0.5 ms per query at T = 32, whatever else the
// other 255 CUs could do.  Here a CLUSTER of NWG = 16 (32) workgroups serves up to 4 sequences: workgroup p owns hidden
// units [p*UW, (p+1)*UW) -- all four gates of them -- and keeps its 4*UW columns of the matrix in LDS for the whole
// call (80 KB at H = 256), so a step reads no weights from memory at all.  What crosses workgroups per step is h_t:
// every workgroup publishes its UW units in a global exchange buffer and reads the full h_t back (1 KB per sequence).
// There is no barrier object: an exchange element is the 8-byte pair {value, tag} written with ONE 64-bit agent-scope
// atomic store, tag = (call epoch, step), and a reader simply re-reads an element until it carries the tag it waits
// for -- one store and one load round trip per step (a counter-based barrier with agent-scope fences cost 14 us per
// step, with atomics only 5 us; this costs ~2).  Two buffers alternate: whoever writes step t+2 into the buffer of
// step t has seen all of step t+1, which every workgroup produced after reading step t.
// The workgroups of a cluster are launched onto one XCD (grid = 8 x NWG, cluster = blockIdx % 8: workgroup ids are
// dealt round-robin to the 8 XCDs), so the exchange stays within one L2; correctness does not depend on that placement.
// All workgroups of a cluster must be resident at the same time: the launcher caps the grid at half the CUs, and a
// bounded spin turns a scheduling surprise into an error flag instead of a hang.
//
// Arithmetic: the fp32 fma chain of the matrix path in its k order (see lstm_small.hip), the same gate formulas,
// projection order and sum-of-squares tree: results are bit-identical to lstm_small.hip / lstm_fwd.hip, and the
// pad-prefix table of lstm_small.hip is this kernel's too.
#include <cstdlib>

#include "sse_kernels.h"
#include <atomic>

#define LP_RB 4      // sequences per cluster
#define LP_NT 256    // threads per workgroup: wave g computes gate g of (sequence, unit) = lane
#define LP_WAIT_TICKS 1000000  // 10 ms of the 100 MHz wall clock without the awaited word: give up (error bit 2)
#define LP_EPT 8     // exchange elements per thread per step: RB * H / NT (H <= 512)

__device__ __forceinline__ float lp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504089f * x)); }
__device__ __forceinline__ float lp_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008178f * x)); }

// exchange element: {float value, uint32 tag} as one 64-bit word, agent scope (the device-coherent level of the memory
// system; no cache write-back / invalidate fences are needed around single-copy-atomic 64-bit accesses)
__device__ __forceinline__ void lp_publish(unsigned long long *p, float v, unsigned int tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait gives up when its word has not come for 10 ms, or as soon as any other wait of the launch has given up (the error
// word is polled every 64 spins): a cluster that cannot become resident costs one time-out, not one per step and workgroup.
__device__ __forceinline__ float lp_await(const unsigned long long *p, unsigned int tag, int32_t *err) {
  unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  long long since = 0;
  while ((unsigned int)(w >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((++spins & 63) == 0) {
      if (since == 0) since = wall_clock64();
      if ((__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) != 0 || wall_clock64() - since >= LP_WAIT_TICKS) {
        atomicOr(err, 4);  // the producing workgroup never ran: report, do not hang
        break;
      }
    }
  }
  return __uint_as_float((unsigned int)w);
}

__global__ __launch_bounds__(LP_NT) void lstm_persist_kernel(LstmPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cluster = a.map_mode ? blockIdx.x / a.NWG : (blockIdx.x & 7);
  const int p = a.map_mode ? blockIdx.x % a.NWG : (blockIdx.x >> 3);
  const int NCL = (a.B + LP_RB - 1) / LP_RB;
  if (cluster >= NCL) return;  // launched only to keep a cluster's workgroups on one XCD
  const int E = a.E, H = a.H, T = a.T, S = a.S, N4 = 4 * H, NWG = a.NWG;
  const int UW = (H + NWG - 1) / NWG;                // hidden units per workgroup (<= 16)
  const int u0 = p * UW, nu = max(0, min(UW, H - u0));  // this workgroup's units [u0, u0 + nu)
  const int KX = (E + 8) & ~7, KH = (H + 7) & ~7;
  const int NG = ((KX + KH) / 8 + 3) / 4 * 4, KA = NG * 8;  // the k space of Waug (lstm_small.hip, ring depth 4)
  const int KAp = KA + 4;                                 // LDS row stride: 16-byte reads of 16 rows hit 64 distinct banks
  const int SW = (S + NWG - 1) / NWG;                     // projection columns per workgroup
  float *Wl = sm;                                         // [4][UW][KAp] this workgroup's columns, k in consumption order
  float *av = Wl + ((max(4 * UW * KAp, H * SW) + 3) & ~3);             // [RB][KAp] operand rows [x_t | 1 | 0.. | h_{t-1} | 0..]
  float *gs = av + ((max(LP_RB * KAp, LP_RB * S) + 3) & ~3);           // [4][64] gate pre-activations
  float *red = gs + 256;                                  // [RB][16] scratch
  const int b0 = cluster * LP_RB, nb = min(LP_RB, a.B - b0);
  unsigned long long *hx = a.hx + (size_t)cluster * 2 * LP_RB * H;  // [2][RB][H] {h, tag}
  unsigned long long *rawx = a.rawx + (size_t)cluster * LP_RB * S;  // [RB][S] {raw encoding, tag}
  const unsigned int epoch = a.epoch << 12;               // tag = epoch | step + 1 (T < 4095), unique per call
  // Where h_t is published.  A write-through (sc1) store is visible to every XCD but drops the line from the writer's L2,
  // so each of the 32 hand-offs of a query goes out to the memory side and back.  When all workgroups of the cluster sit
  // on ONE XCD (what blockIdx % 8 gives in practice; HIP promises nothing, so every workgroup publishes its XCC id here,
  // write-through, and the ids are compared once the weights are in LDS) a plain store leaves the line in the L2 the readers
  // share and their L1-bypassing loads are served from there.
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xF;
  const bool xcc_check = !a.write_through && LP_RB * H >= NWG;  // (tiny cells: fewer buffer slots than workgroups)

  // left-pad prefix skip, exactly as lstm_small.hip
  int t0 = 0;
  if (a.pad_h != nullptr) {
    int lead = T;
    if (wv < nb) {
      const int32_t *row = a.ids + (size_t)(b0 + wv) * T;
      for (int t = lane; t < T; t += 64)
        if (row[t] != 0) {
          lead = t;
          break;
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lead = min(lead, __shfl_xor(lead, o));
    }
    if (lane == 0) red[wv] = __int_as_float(lead);
    __syncthreads();
    lead = T;
    for (int r = 0; r < nb; ++r) lead = min(lead, __float_as_int(red[r]));
    t0 = min(lead, T - 1);
    __syncthreads();
  }
  // The ids go into slots 0..NWG-1 of the buffer with parity t0 & 1, tag "step 0" (h tags start at 1).  The first step
  // publishes tag t0 + 1 into the OTHER buffer; buffer t0 & 1 is first rewritten by step t0 + 1 (tag t0 + 2), which a
  // workgroup reaches only after it has read every peer's tag t0 + 1 -- and a peer publishes that after passing its own
  // id check.  (With the ids always in the even buffer a left-padded query with an odd PAD prefix overwrote them in its
  // very first step, before slower peers had compared them: an intermittent 10 ms give-up.)
  unsigned long long *xid = hx + (size_t)(t0 & 1) * LP_RB * H;
  if (tid == 0 && xcc_check) lp_publish(xid + p, __uint_as_float(xcc), epoch);
  auto fetch_id = [&](int b, int t) -> int {
    int id = (b < nb) ? a.ids[(size_t)(b0 + b) * T + t] : 0;
    if (id < 0 || id >= a.V) {
      atomicOr(a.err, 1);
      id = 0;
    }
    return id;
  };

  // ---- this workgroup's weight columns into LDS, once.  Row (g, u) holds column g*H + u0 + u of Waug with k in the
  // order the fma chain consumes it: within a k-group of 8, position 2e holds k = e and 2e + 1 holds k = 4 + e.
  // (several loads in flight per thread: one load -> store round trip per element would cost more than the whole encode)
  if ((UW & 3) == 0 && (H & 3) == 0) {
    // 16-byte loads: 4 consecutive units of one gate at one k
    const int Q = UW / 4, n4 = 4 * Q * KA;  // float4 pieces: (k, gate, unit quad)
    for (int i0 = 0; i0 < n4; i0 += 4 * LP_NT) {
      f32x4 w4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * LP_NT + tid;
        const int k = i / (4 * Q), r = i - k * (4 * Q), g = r / Q, q = r - g * Q;
        w4[j] = f32x4{0, 0, 0, 0};
        if (i < n4 && q * 4 < nu) w4[j] = *reinterpret_cast<const f32x4 *>(a.Waug + (size_t)k * N4 + g * H + u0 + q * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * LP_NT + tid;
        const int k = i / (4 * Q), r = i - k * (4 * Q), g = r / Q, q = r - g * Q;
        const int kk = k & 7, pos = (k & ~7) + ((kk & 3) << 1) + (kk >> 2);
        if (i < n4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) Wl[(size_t)(g * UW + q * 4 + e) * KAp + pos] = (q * 4 + e < nu) ? w4[j][e] : 0.0f;
        }
      }
    }
  } else {
    for (int i0 = 0; i0 < 4 * UW * KA; i0 += 8 * LP_NT) {
      float wv8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * LP_NT + tid;
        const int k = i / (4 * UW), r = i - k * (4 * UW);  // consecutive threads: consecutive columns of one k row
        const int g = r / UW, u = r - g * UW;
        wv8[j] = (i < 4 * UW * KA && u < nu) ? a.Waug[(size_t)k * N4 + g * H + u0 + u] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * LP_NT + tid;
        const int k = i / (4 * UW), r = i - k * (4 * UW);
        const int kk = k & 7, pos = (k & ~7) + ((kk & 3) << 1) + (kk >> 2);
        if (i < 4 * UW * KA) Wl[(size_t)r * KAp + pos] = wv8[j];
      }
    }
  }
  for (int i = tid; i < LP_RB * KAp; i += LP_NT) {
    const int b = i / KAp, k = i - b * KAp;
    float v = 0.0f;
    if (k < E) v = a.emb[(size_t)fetch_id(b, t0) * E + k];
    else if (k == E) v = 1.0f;
    else if (k >= KX && k - KX < H && t0 > 0) v = a.pad_h[(size_t)t0 * a.pad_stride + (k - KX)];
    av[i] = v;
  }
  // lane -> (sequence, unit of this workgroup); wave 0 owns the cell state
  const int lb = lane / UW, lu = lane - lb * UW;
  const bool lane_on = lb < LP_RB && lu < nu;
  float c = (wv == 0 && lane_on && t0 > 0) ? a.pad_c[(size_t)t0 * a.pad_stride + u0 + lu] : 0.0f;
  const float *wrow = Wl + (size_t)(wv * UW + (lane_on ? lu : 0)) * KAp;
  const float *vrow = av + (size_t)(lane_on ? lb : 0) * KAp;
  __syncthreads();

  // A workgroup that owns no hidden unit (H not a multiple of UW: e.g. H = 40, NWG = 16 leaves p = 14, 15 empty) publishes
  // nothing, so nobody ever waits for it -- and the two-buffer argument above ("whoever writes step t+2 has seen every
  // workgroup's step t+1") does not cover its READS: lagging behind, it would find tag t+2 where it expects t.  It has no
  // use for h_t before the projection, so it skips the steps and picks up h_T (the last write into its buffer) below.
  const bool bystander = nu == 0;
  // (a bystander publishes its XCC id like everybody -- it will READ h_T -- but does not wait for the others': the slots are
  // reused by step t0 + 1, which the others reach without it)
  bool wthrough = true;
  if (xcc_check && !bystander) {
    if (wv == 0) {
      bool same = true;
      for (int i = lane; i < NWG; i += 64) same = same && __float_as_uint(lp_await(xid + i, epoch, a.err)) == xcc;
      const bool all_same = __all(same);
      if (lane == 0) red[63] = all_same ? 0.0f : 1.0f;
    }
    __syncthreads();
    wthrough = red[63] != 0.0f;
  }
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * LP_RB * H * 8, 0x00020000);
  for (int t = bystander ? T : t0; t < T; ++t) {
    // ---- gate pre-activation of (gate wv, sequence lb, unit lu): the matrix path's fma chain
    float acc = 0.0f;
#pragma unroll 4
    for (int g = 0; g < NG; ++g) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(vrow + g * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(vrow + g * 8 + 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wrow + g * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wrow + g * 8 + 4);
      acc = __builtin_fmaf(lo[0], w0[0], acc);
      acc = __builtin_fmaf(hi[0], w0[1], acc);
      acc = __builtin_fmaf(lo[1], w0[2], acc);
      acc = __builtin_fmaf(hi[1], w0[3], acc);
      acc = __builtin_fmaf(lo[2], w1[0], acc);
      acc = __builtin_fmaf(hi[2], w1[1], acc);
      acc = __builtin_fmaf(lo[3], w1[2], acc);
      acc = __builtin_fmaf(hi[3], w1[3], acc);
    }
    gs[wv * 64 + lane] = acc;
    __syncthreads();  // gates complete; nobody reads the operand rows any more
    if (wv == 0) {
      if (lane_on && lb < nb) {
        const float si = lp_sigmoid(gs[lane]);
        const float tj = lp_tanh(gs[64 + lane]);
        const float sf = lp_sigmoid(gs[128 + lane]);
        const float so = lp_sigmoid(gs[192 + lane]);
        const float pij = __fmul_rn(si, tj);  // the matrix kernel parks this product (rounded) between its two passes
        c = __builtin_fmaf(c, sf, pij);
        const int slot = (((t + 1) & 1) * LP_RB + lb) * H + u0 + lu;
        const float hv = lp_tanh(c) * so;
        if (wthrough) {
          lp_publish(hx + slot, hv, epoch | (unsigned)(t + 1));
        } else {
          typedef unsigned int lp_u32x2 __attribute__((ext_vector_type(2)));
          __builtin_amdgcn_raw_buffer_store_b64(lp_u32x2{__float_as_uint(hv), epoch | (unsigned)(t + 1)}, hrs, slot * 8, 0, 0);
        }
      }
    } else if (t + 1 < T) {  // the other waves bring x_{t+1}
      for (int i = tid - 64; i < LP_RB * E; i += LP_NT - 64) {
        const int b = i / E, k = i - b * E;
        av[b * KAp + k] = a.emb[(size_t)fetch_id(b, t + 1) * E + k];
      }
    }
    // h_t of every workgroup of the cluster, each element as soon as it is there (the operand rows' h part is free:
    // the barrier above was passed by every wave after its last read)
    // (rows >= nb are not exchanged: whatever their operand rows hold never reaches an output.)  All of a thread's
    // elements are requested together, then re-read one by one until they carry this step's tag.
    {
      const unsigned long long *src = hx + (size_t)((t + 1) & 1) * LP_RB * H;
      const unsigned int tag = epoch | (unsigned)(t + 1);
      const int n_el = nb * H;
      unsigned long long w[LP_EPT];
#pragma unroll
      for (int j = 0; j < LP_EPT; ++j) {
        const int i = tid + j * LP_NT;
        w[j] = (i < n_el) ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
      }
#pragma unroll
      for (int j = 0; j < LP_EPT; ++j) {
        const int i = tid + j * LP_NT;
        if (i < n_el) {
          float v = __uint_as_float((unsigned int)w[j]);
          if ((unsigned int)(w[j] >> 32) != tag) v = lp_await(src + i, tag, a.err);
          const int b = i / H, unit = i - b * H;
          av[b * KAp + KX + unit] = v;
        }
      }
    }
    __syncthreads();
  }

  // ---- projection: this workgroup's columns [s0, s0 + ns) of out = h_T . M, in the matrix path's k order
  const int s0 = p * SW, ns = max(0, min(SW, S - s0));
  if (bystander) {
    if (ns == 0 && p != 0) return;
    const unsigned long long *src = hx + (size_t)(T & 1) * LP_RB * H;
    for (int i = tid; i < nb * H; i += LP_NT) {
      const int b = i / H, unit = i - b * H;
      av[b * KAp + KX + unit] = lp_await(src + i, epoch | (unsigned)T, a.err);
    }
    __syncthreads();
  }
  float *Ml = Wl;  // [H][SW]: the weight columns are no longer needed (region sized for both)
  for (int i = tid; i < H * SW; i += LP_NT) {
    const int k = i / SW, j = i - k * SW;
    Ml[i] = (j < ns) ? a.M[(size_t)k * S + s0 + j] : 0.0f;
  }
  __syncthreads();
  for (int i = tid; i < LP_RB * SW; i += LP_NT) {
    const int b = i / SW, j = i - b * SW;
    const float *hb = av + b * KAp + KX;
    float acc = 0.0f;
    for (int kb = 0; kb < KH; kb += 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = kb + e, k1 = kb + 4 + e;
        if (k0 < H) acc = __builtin_fmaf(hb[k0], Ml[k0 * SW + j], acc);
        if (k1 < H) acc = __builtin_fmaf(hb[k1], Ml[k1 * SW + j], acc);
      }
    }
    if (j < ns) lp_publish(rawx + (size_t)b * S + s0 + j, acc, epoch | (unsigned)(T + 1));
  }
  if (p != 0) return;

  // ---- workgroup 0 of the cluster: tf.nn.l2_normalize in the other kernels' summation order, and the result
  float *raw = av;  // [RB][S] (region sized for both)
  __syncthreads();  // the projection above read h_T from this region
  for (int i = tid; i < LP_RB * S; i += LP_NT) raw[i] = lp_await(rawx + i, epoch | (unsigned)(T + 1), a.err);
  __syncthreads();
  const int NTS = (S + 31) / 32;
  if (a.normalize) {
    for (int i = tid; i < LP_RB * NTS * 32; i += LP_NT) {  // 32-lane groups stay whole
      const int col = i & 31, tile = (i >> 5) % NTS, b = i / (32 * NTS);
      const int s = tile * 32 + col;
      float v = (s < S) ? raw[b * S + s] : 0.0f;
      v = v * v;
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if (col == 0) red[b * 16 + tile] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < LP_RB * S; i += LP_NT) {
    const int b = i / S, s = i - b * S;
    if (b >= nb) continue;
    float scale = 1.0f;
    if (a.normalize) {
      float tot = 0.0f;
      for (int j = 0; j < NTS; ++j) tot += red[b * 16 + j];
      scale = 1.0f / sqrtf(fmaxf(tot, 1e-12f));
    }
    a.out[(size_t)(b0 + b) * S + s] = raw[i] * scale;
  }
}

static size_t lp_lds_bytes(int E, int H, int S, int NWG) {
  const int KX = (E + 8) & ~7, KH = (H + 7) & ~7;
  const int NG = ((KX + KH) / 8 + 3) / 4 * 4, KAp = NG * 8 + 4;
  const int UW = (H + NWG - 1) / NWG, SW = (S + NWG - 1) / NWG;
  size_t wl = (size_t)4 * UW * KAp;
  if ((size_t)H * SW > wl) wl = (size_t)H * SW;
  size_t avn = (size_t)LP_RB * KAp;
  if ((size_t)LP_RB * S > avn) avn = (size_t)LP_RB * S;
  wl = (wl + 3) & ~(size_t)3;
  avn = (avn + 3) & ~(size_t)3;
  return (wl + avn + 256 + LP_RB * 16) * sizeof(float);
}

// workgroups per cluster for this shape: 16, or 32 when 16 columns-slices do not fit the LDS; 0 = shape not supported
int lstm_persist_nwg(int E, int H, int S) {
  if (H < 1 || H > 512 || S < 1 || S > 512) return 0;
  for (int nwg = 16; nwg <= 32; nwg *= 2) {
    const int UW = (H + nwg - 1) / nwg;
    if (UW * LP_RB <= 64 && lp_lds_bytes(E, H, S, nwg) <= 160 * 1024) return nwg;
  }
  return 0;
}

int lstm_persist_max_rows() { return 8 * LP_RB; }  // 8 clusters (one per XCD) of 4 sequences

size_t lstm_persist_hx_words(int H) { return (size_t)8 * 2 * LP_RB * H; }
size_t lstm_persist_raw_words(int S) { return (size_t)8 * LP_RB * S; }
int lstm_persist_max_steps() { return 4094; }

// a.epoch must differ from the epoch of every earlier launch that used the same exchange buffers (20 bits; the buffers
// start zeroed and epoch 0 is never used)
static std::atomic<long long> g_coop_refused{0};
void lstm_note_coop_refused() { g_coop_refused.fetch_add(1, std::memory_order_relaxed); }
long long lstm_coop_refused() { return g_coop_refused.load(std::memory_order_relaxed); }

hipError_t launch_lstm_persist(const LstmPersistArgs &a_in, hipStream_t stream) {
  LstmPersistArgs a = a_in;
  a.NWG = lstm_persist_nwg(a.E, a.H, a.S);
  if (a.NWG == 0 || a.B < 1 || a.B > lstm_persist_max_rows() || a.T > lstm_persist_max_steps() || a.epoch == 0 ||
      a.epoch >= (1u << 20))
    return hipErrorInvalidValue;
  const size_t lds = lp_lds_bytes(a.E, a.H, a.S, a.NWG);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // COOPERATIVE launch: the workgroups of a cluster hand h_t to each other every step, so all of them must be resident at
  // once; hipLaunchCooperativeKernel makes the runtime guarantee that (the grid fits the device by construction: at most
  // one workgroup per CU is asked for) instead of leaving it to the dispatcher's mood on a busy device.  The bounded spin
  // with its give-up flag stays as a belt.  (A runtime without cooperative launches falls back to the plain launch.)
  // Measured cost: +20 us per launch (single query 0.122 -> 0.142 ms); option lstm_cluster_coop = 0 takes the plain launch.
  static const bool no_coop = getenv("SSE_NO_COOP") != nullptr;  // measurement aid: plain launches
  if (!no_coop && !a.plain_launch) {
    void *args[] = {(void *)&a};
    hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(lstm_persist_kernel), dim3(8 * a.NWG), dim3(LP_NT), args, (unsigned)lds, stream);
    if (ce == hipSuccess) return hipGetLastError();
    (void)hipGetLastError();  // not supported / too large for this device: plain launch, and say so (counter lstm_coop_refused)
    lstm_note_coop_refused();
  }
  hipLaunchKernelGGL(lstm_persist_kernel, dim3(8 * a.NWG), dim3(LP_NT), lds, stream, a);
  return hipGetLastError();
}
