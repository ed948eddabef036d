// Fused multi-head attention for gfx950, bf16 in / fp32 accumulate, head_dim 64.
// Replaces speech_jobs/whisper_dist.py:147-171 (q·kᵀ, additive mask, softmax, probs·v, head
// merge) and its gradient without materialising the [B,H,Tq,Tk] score tensor.
//
// One structure serves forward, the dQ pass and the dK/dV pass:
//   - a wavefront OWNS 32 rows (queries in fwd/dQ, keys in dK/dV); the owner index sits on
//     the MFMA lane, its 64-wide vectors live in registers as B operands;
//   - the other ("streamed") index is walked in 64-row tiles that the workgroup (4 waves =
//     128 owners) stages ONCE, global -> LDS by DMA (global_load_lds_dwordx4), double-buffered
//     so the next tile is in flight under the current tile's MFMAs;
//   - ONE natural [row][64 d] LDS image per tile serves both products: row fragments by
//     ds_read_b128 for X[s][o] = sum_d T[s][d]·Own[o][d], and hardware-transposed fragments by
//     ds_read_b64_tr_b16 for Yᵀ[d][o] += sum_s T[s][d]·X[s][o].  The 16-byte chunk swizzle
//     phys = chunk ^ (((row >> 1) & 1) << 2 | ((row >> 2) & 3)) makes both read kinds
//     bank-conflict-free; it is applied to the DMA's per-lane source address and on the reads;
//   - X (scores / probabilities / dS) never leaves the accumulator registers: a 32x32 MFMA
//     result has its column on the lane and its rows in the registers, so it is directly the
//     B operand of the next MFMA that sums over its rows (k-order inside a 16-step permuted:
//     element j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3); the transposed reads fetch
//     the other operand in that same order);
//   - softmax statistics are per-lane scalars in fwd/dQ (query on the lane) and per-row
//     constants from LDS in dK/dV.
// The reference's decoder mask (W:416-418 + W:152-153) adds -1e9 in fp32 to keys j <= i;
// a fully masked row therefore softmaxes to exactly uniform, which is why (m, 1/l) are kept
// as two numbers instead of one log-sum-exp.
//
// Softmax arithmetic runs in the log2 domain: c2 = score_scale * log2(e) is folded into one
// fma per element, p = exp2(fma(s, c2, -m)) is a single v_exp_f32, and the stored statistic m
// is in log2 units (private to these three kernels).  The masked constant becomes -1e9*log2(e),
// which still absorbs every score in fp32, so the uniform-row behaviour is unchanged.  Tiles
// that need no masking (all of the encoder except its last key tile) take a branch-free path;
// the forward rescales its accumulators only when some row's maximum grew by more than 2^8
// (any m works as long as l and O were accumulated with the same one).
// The VALU budget matters as much as the MFMAs here: a 32x32x16 MFMA occupies the SIMD for
// 32 cycles, v_exp_f32 costs 8 issue cycles and every other vector op 4.
#include "tmi_common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace {

constexpr int HD = 64;
constexpr int TROWS = 64;             // streamed rows per tile
constexpr int IMG = TROWS * 128;      // bytes of one [64][64] bf16 image

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// ---- stored dropout mask (tmi_attn_desc.drop_mask): u32 words [B*H][NT = ceil(Tk/64)][2][TQP = ceil(Tq/128)*128].
// Word (bh, t, h, q) = the keep bits of the 32 scores that the forward lane (query q, lane half h) holds of key tile t:
// bit j (j = 0..15) = the even key of pair j, bit 16 + j = its odd key, pair j = 8 rbk + 2 g + jj <-> keys
// 64 t + 32 rbk + 8 g + 4 h + 2 jj + {0, 1} (rbk = 32-key half of the tile, g = accumulator quad, jj = pair of the quad).
// Inside a block of 64 query rows the words sit in the order the dK/dV kernel wants them as wave-wide select masks:
// row r at position mask_pos(r), so that positions 2e, 2e + 1 of each 32-row half are rows acc_row(e, 0), acc_row(e, 1) -
// one aligned 64-bit scalar load = the select mask of accumulator register e over the wave's 64 lanes.
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mask_pos(int r) {
  const int rr = r & 31;
  return (r & 32) + 2 * ((rr & 3) + 4 * (rr >> 3)) + ((rr >> 2) & 1);
}
// 0xffff in each half whose SIGNED 16-bit draw is >= T (kept), given tm1 = (T - 1) in both halves: (T - 1) - d saturates
// inside int16, so its sign is exact (v_pk_sub_i16 clamp + v_pk_ashrrev_i16)
__device__ __forceinline__ uint32_t keep_pair(uint32_t hw, uint32_t tm1) {
  const short2_t x = __builtin_elementwise_sub_sat(__builtin_bit_cast(short2_t, tm1), __builtin_bit_cast(short2_t, hw));
  return __builtin_bit_cast(uint32_t, x >> (short2_t){15, 15});
}
// all ones when bit BIT of w is set (v_bfe_i32 of a 1-bit field sign-extends; hipcc rewrites the C form into and + cmp + cndmask)
template <int BIT> __device__ __forceinline__ uint32_t bit_ones(uint32_t w) {
  uint32_t t;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(w), "n"(BIT));
  return t;
}
template <int... I, class F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
// Plain fmaxf: hipcc fuses the chain into v_max3_f32 by itself AND pads the MFMA -> VALU hazard in front of it.  Round 4
// bug fix: until then this was an inline-asm v_max3_f32, and hipcc neither models nor pads an asm statement
// (cdna_hip_programming.md 5.7 item 2): the first maxima read the score accumulators up to 12 wait states too early - stale
// registers on some waves of some launches, more of them the busier the CU (3 % of the encoder's attention outputs differed
// between two identical launches, by up to 20 % of their value; tools/attn_determinism.py).
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED2 = -1e9f * LOG2E;  // the reference's -1e9, in log2 units
constexpr float LAZY = 8.f;             // forward: rescale only when a maximum grew by > 2^8

// Make the compiler's own wait-count model see a register as already loaded.  Without this it
// believes the pre-loop global loads are still in flight (the explicit waits below are asm, opaque
// to it) and puts s_waitcnt vmcnt(0) at their first use INSIDE the tile loop, which drains the
// freshly issued LDS-DMA of the next tile every iteration.
#define PIN(v) asm volatile("" : "+v"(v))

// LDS-DMA from inline asm (see gemm_fast.hip: keeps hipcc from draining it before LDS reads)
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_wave_base);
  const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst_u)
               : "memory");
}

// stage rows [row0, row0+64) of a [T][.. 64 d ..] matrix (token stride st elements) into img
__device__ __forceinline__ void stage_img(char* img, const bf16_t* base, int64_t st, int row0, int T, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = 2 * wave + i;
    const int r = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ swz(r);
    int gr = row0 + r;
    gr = gr < T ? gr : T - 1;
    glds16(base + (int64_t)gr * st + c * 8, img + j * 1024);
  }
}

// Same staging with the per-lane part of the source address computed once: lp[i] points at the
// lane's 16 bytes of row r_i of tile 0; full tiles add a wave-uniform offset.
struct LaneSrc {
  const bf16_t* lp[2];
  const bf16_t* base;
  int64_t st;
  int T;
};
__device__ __forceinline__ LaneSrc lane_src(const bf16_t* base, int64_t st, int T, int wave, int lane) {
  LaneSrc L;
  L.base = base;
  L.st = st;
  L.T = T;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wave + i) + (lane >> 3);
    const int c = (lane & 7) ^ swz(r);
    L.lp[i] = base + (int64_t)r * st + c * 8;
  }
  return L;
}
__device__ __forceinline__ void stage_tile(char* img, const LaneSrc& L, int row0, int wave, int lane) {
  if (row0 + TROWS <= L.T) {
    const int64_t off = (int64_t)row0 * L.st;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(L.lp[i] + off, img + (2 * wave + i) * 1024);
  } else {
    stage_img(img, L.base, L.st, row0, L.T, wave, lane);
  }
}

// owner vectors: lane (c, h) holds Own[o][16kk + 8h .. +7], kk = 0..3
__device__ __forceinline__ void load_owner(bf16x8 (&f)[4], const bf16_t* base, int64_t st, int o, int T, int h) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (o < T) {
      f[kk] = *reinterpret_cast<const bf16x8*>(base + (int64_t)o * st + kk * 16 + h * 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[kk][j] = (bf16_t)0.f;
    }
  }
}

// X[s][o] = sum_d T[rb + s][d] * Own[o][d], s = 0..31
__device__ __forceinline__ f32x16 first_product(const char* img, int rb, const bf16x8 (&own)[4], int c, int h) {
  f32x16 x;
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = 0.f;
  const int row = rb + c;
  const int sw = swz(row);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(img + row * 128 + (((2 * kk + h) ^ sw) << 4));
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, own[kk], x, 0, 0, 0);
  }
  return x;
}

// Yt[blk][d][o] += sum_s T[rb + s][32 blk + d] * X[s][o]   (X as bf16 B operands: xb[sI][j] = X[8 sI + j][o])
__device__ __forceinline__ void second_product_b(const char* img, int rb, const bf16x8 (&xb)[2], f32x16 (&y)[2], int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int h = g >> 1, q = i >> 2, p = i & 3;
#pragma unroll
  for (int sI = 0; sI < 2; ++sI) {
    const bf16x8 b = xb[sI];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      bf16x4 part[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = rb + 16 * sI + 8 * half + 4 * h + q;
        const int col = 32 * blk + 16 * (g & 1) + 4 * p;
        const char* addr = img + row * 128 + ((((col >> 3) ^ swz(row))) << 4) + (col & 7) * 2;
        part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
      }
      const bf16x8 a = __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
      y[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y[blk], 0, 0, 0);
    }
  }
}
// the same with X still an fp32 accumulator (s = 0..31)
__device__ __forceinline__ void second_product(const char* img, int rb, const f32x16& x, f32x16 (&y)[2], int lane) {
  bf16x8 xb[2];
#pragma unroll
  for (int sI = 0; sI < 2; ++sI)
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[sI][j] = (bf16_t)x[8 * sI + j];
  second_product_b(img, rb, xb, y, lane);
}

// write Yt[blk][d][o] * scale to out[o][d] (bf16), owner o on the lane
__device__ __forceinline__ void store_owner(const f32x16 (&y)[2], bf16_t* base, int64_t st, int o, int T, int h,
                                            float scale) {
  if (o >= T) return;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (bf16_t)(y[blk][4 * g + i] * scale);
      *reinterpret_cast<bf16x4*>(base + (int64_t)o * st + 32 * blk + 8 * g + 4 * h) = v;
    }
  }
}


// the same owner-on-lane tile as fp32 into one row of 64 floats (key-split partials)
__device__ __forceinline__ void store_owner_f32(const f32x16 (&y)[2], float* row, int h) {
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = y[blk][4 * g + i];
      *reinterpret_cast<f32x4*>(row + 32 * blk + 8 * g + 4 * h) = v;
    }
  }
}

// Fold of the key-split partials.  One thread per (row, 4 columns): rows = B*H*Tq.
//   forward: m = max_s m_s; w_s = 2^(m_s - m); l = sum_s l_s w_s; o = (sum_s o_s w_s) * out_scale / l; stats = (m, 1/l)
//   dQ:      dq = (sum_s dq_s) * out_scale
template <bool FWD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part, int ks, int64_t BH, int Tq,
                                                           bf16_t* __restrict__ out, int64_t o_sb, int64_t o_st, int H,
                                                           float* __restrict__ stats, float out_scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = idx >> 4;
  const int ch = (int)(idx & 15);
  if (row >= BH * Tq) return;
  const int64_t bh = row / Tq;
  const int q = (int)(row - bh * Tq);
  const float* ml = part + BH * ks * (int64_t)Tq * HD;
  float m = -INFINITY;
  if constexpr (FWD) {
    for (int s_ = 0; s_ < ks; ++s_) m = fmaxf(m, ml[((bh * ks + s_) * Tq + q) * 2]);
  }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s_ = 0; s_ < ks; ++s_) {
    const int64_t r = (bh * ks + s_) * Tq + q;
    float w = 1.f;
    if constexpr (FWD) {
      w = ex2(ml[r * 2] - m);
      l += ml[r * 2 + 1] * w;
    }
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + r * HD + 4 * ch);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += v[i] * w;
  }
  float sc = out_scale;
  if constexpr (FWD) sc = out_scale / l;
  const int64_t b = bh / H, head = bh - b * H;
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(acc[i] * sc);
  *reinterpret_cast<bf16x4*>(out + b * o_sb + head * HD + (int64_t)q * o_st + 4 * ch) = o;
  if constexpr (FWD) {
    if (ch == 0) {
      stats[row * 2] = m;
      stats[row * 2 + 1] = 1.0f / l;
    }
  }
}

#define ZERO2(y)                          \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) y[i_][e_] = 0.f

// ------------------------------------------------------------------ forward
struct AttnP {
  tmi_attn_desc d;
  float dq_scale;
  float sscale;  // scores = (q . k) * sscale
  float c2;      // sscale * log2(e)
  // dropout on the probabilities (W:160): thr == 0 is off; see tmi_common.h for the generator.  The counter of
  // element (q, k) of stream b*H + head: row q, column k (pairs run along k).
  uint32_t drop_thr;
  float keep_scale;
  uint32_t seed_lo, seed_hi;
  uint32_t tm1;       // (thr - 32768 - 1) & 0xffff in both halves: keep_pair's threshold
  uint32_t* dmask;    // the stored keep bits (layout at mask_pos above)
  int NT, TQP;        // key tiles, padded query rows of dmask
  // XCD-aware launch (gx > 0): a 1-D grid whose workgroup L works on (batch, head) pair (L % 8) + 8 * (L / 8 / gx) and on
  // block (L / 8) % gx of it.  Workgroups are dealt round-robin over the 8 XCDs, so every workgroup of one (batch, head)
  // lands on one XCD and the K / V (or Q / dO) rows they all stream stay in that XCD's 4 MiB L2 (speed only: nothing
  // depends on the placement).  Needs B * H % 8 == 0.
  int gx;
  // key split (forward and dQ passes of a short query side against a long key side: cross-attention, Tq <= 128): the
  // workgroups of one (batch, head) are `ksplit` disjoint key ranges; each leaves its un-normalised partial in `part`
  // ([B*H][ksplit][Tq][64] fp32, then for the forward [B*H][ksplit][Tq][2] = (m, l)) and a combine kernel folds them.
  int ksplit;
  float* part;
  int nq;  // attn_bwd_small_kernel: blocks [0, nq) of a pair run the dQ pass
};

// ABL (diagnostics, TMI_ATTN_ABL): 1 = no softmax arithmetic (p = s), 2 = no second product, 3 = no staging after the
// prologue (every tile re-reads tile 0's images), 4 = no first product
template <bool DROP, int OCC, int ABL = 0>
__global__ __launch_bounds__(256, OCC) void attn_fwd_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K img | V img]
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  int bx_, head_;
  int64_t b_;
  if (P.gx > 0) {
    const int L = (int)blockIdx.x, r = L >> 3;
    const int bh = (r / P.gx) * 8 + (L & 7);
    bx_ = r % P.gx;
    head_ = bh % (int)P.d.H;
    b_ = bh / (int)P.d.H;
  } else {
    bx_ = (int)blockIdx.x;
    head_ = (int)blockIdx.y;
    b_ = (int64_t)blockIdx.z;
  }
  const int bx = bx_, head = head_;
  const int64_t b = b_;
  const int ks = P.ksplit, qt = bx / ks, sp = bx - qt * ks;
  const int q = qt * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bool causal = d.mask_mode == 1;
  const float c2 = P.c2;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;

  bf16x8 qf[4];
  load_owner(qf, qb, d.q_st, q, Tq, h);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) PIN(qf[kk]);
  f32x16 o[2];
  ZERO2(o);
  float m = -INFINITY, l = 0.f;
  const uint32_t tm1 = P.tm1;
  const uint32_t skey = tmi_stream_key(((uint64_t)P.seed_hi << 32) | P.seed_lo, (uint32_t)(b * d.H + head));
  const tmi_rowkey qrk = tmi_row_key(skey, (uint32_t)q);  // the mask row of this lane's query: one avalanche per kernel
  // this lane's column of the stored mask: word (b*H + head, tile, h, q); q < TQP for every lane of the grid
  uint32_t* mrow = nullptr;
  if constexpr (DROP) mrow = P.dmask + (((int64_t)(b * d.H + head) * P.NT) * 2 + h) * P.TQP + (q & ~63) + mask_pos(q & 63);

  const LaneSrc Ks = lane_src(kb, d.k_st, Tk, wave, lane);
  const LaneSrc Vs = lane_src(vb, d.v_st, Tk, wave, lane);
  const int ntiles_all = (Tk + TROWS - 1) / TROWS;
  const int per = (ntiles_all + ks - 1) / ks;
  const int t0 = sp * per, ntiles = min(ntiles_all, t0 + per);  // this workgroup's key tiles [t0, ntiles) (host: never empty)
  stage_tile(smem, Ks, t0 * TROWS, wave, lane);
  stage_tile(smem + IMG, Vs, t0 * TROWS, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  // one tile; EDGE = needs masking (causal, or the ragged last key tile)
  auto body = [&](int tile, auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    const char* Kimg = smem + cur * 2 * IMG;
    const char* Vimg = Kimg + IMG;
    if (ABL != 3 && tile + 1 < ntiles) {
      char* nx = smem + (cur ^ 1) * 2 * IMG;
      stage_tile(nx, Ks, (tile + 1) * TROWS, wave, lane);
      stage_tile(nx + IMG, Vs, (tile + 1) * TROWS, wave, lane);
    }
    f32x16 s[2];
    if constexpr (ABL == 4) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[0][e] = (float)(tile + e) * 0.01f; s[1][e] = (float)(tile - e) * 0.01f; }
      asm volatile("" : "+v"(s[0]), "+v"(s[1]));
    } else {
    s[0] = first_product(Kimg, 0, qf, c, h);   // s[key][q], keys 0..31 of the tile
    s[1] = first_product(Kimg, 32, qf, c, h);  // keys 32..63
    }
    const int key0 = tile * TROWS;
    float rs = 0.f;
    if constexpr (ABL == 1) {
      asm volatile("" : "+v"(s[0]), "+v"(s[1]));
      rs = 1.f;
    } else if constexpr (!edge) {
      // branch-free tile: maximum over the raw products (c2 > 0), lazy rescale
      float mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
      for (int e = 3; e < 15; e += 2) mx = max3(mx, s[0][e], s[0][e + 1]);
      mx = max3(mx, s[0][15], s[1][0]);
#pragma unroll
      for (int e = 1; e < 15; e += 2) mx = max3(mx, s[1][e], s[1][e + 1]);
      mx = fmaxf(mx, s[1][15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mc = mx * c2;
      if (__ballot(mc > m + LAZY) != 0) {
        const float mnew = fmaxf(m, mc);
        const float alpha = ex2(m - mnew);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
        m = mnew;
      }
      const float nm = -m;
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = ex2(fmaf(s[rbk][e], c2, nm));
          s[rbk][e] = pe;
          rs += pe;
        }
      }
    } else {
      float mx = -INFINITY;
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + 32 * rbk + acc_row(e, h);
          float x = s[rbk][e] * c2;
          if (causal && key <= q) x = x + MASKED2;
          if (key >= Tk) x = -INFINITY;
          s[rbk][e] = x;
          mx = fmaxf(mx, x);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m, mx);  // finite: every tile holds at least one key < Tk
      const float alpha = ex2(m - mnew);
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
      m = mnew;
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = ex2(s[rbk][e] - mnew);
          s[rbk][e] = pe;
          rs += pe;
        }
      }
    }
    l += rs;
    if constexpr (DROP) {
      // Dropout (W:160) on the bf16 operands of the second product: one generator evaluation per accumulator quad (4
      // consecutive keys), two draws tested per packed op, the 0xffff / 0 masks ANDed onto the bf16 pairs and gathered into
      // this lane's word of the stored mask (1 bit per score; the backward kernels read it instead of hashing).  The kept
      // probabilities are rescaled on the final store.
      bf16x8 pb[2][2];
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
        for (int sI = 0; sI < 2; ++sI)
#pragma unroll
          for (int j = 0; j < 8; ++j) pb[rbk][sI][j] = (bf16_t)s[rbk][8 * sI + j];
      const uint32_t x0 = qrk.a ^ (uint32_t)((key0 >> 2) + h);  // quad index of (rbk, g): (key0 >> 2) + h + 8 rbk + 2 g = base | const
      uint32_t w = 0;
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
        u32x4 u[2] = {__builtin_bit_cast(u32x4, pb[rbk][0]), __builtin_bit_cast(u32x4, pb[rbk][1])};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const tmi_quad hq = tmi_quad_hash_x(x0 ^ (uint32_t)(8 * rbk + 2 * g), qrk.b);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const uint32_t mk = keep_pair(jj ? hq.b : hq.a, tm1);
            u[g >> 1][2 * (g & 1) + jj] &= mk;
            // w |= mk & (0x00010001 << pair): one v_and_or_b32 (hipcc splits it into an AND per pair plus an OR3 tree)
            asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(w) : "v"(mk), "s"(0x00010001u << (8 * rbk + 2 * g + jj)));
          }
        }
        pb[rbk][0] = __builtin_bit_cast(bf16x8, u[0]);
        pb[rbk][1] = __builtin_bit_cast(bf16x8, u[1]);
      }
      mrow[(int64_t)tile * (2 * P.TQP)] = w;
      second_product_b(Vimg, 0, pb[0], o, lane);
      second_product_b(Vimg, 32, pb[1], o, lane);
    } else
    if constexpr (ABL == 2) {
      asm volatile("" :: "v"(s[0]), "v"(s[1]));
    } else {
    second_product(Vimg, 0, s[0], o, lane);
    second_product(Vimg, 32, s[1], o, lane);
    }
    // (with dropout the mask word is the youngest vector-memory operation of the tile: the seam waits for the DMA in front
    // of it, the store's acknowledgement may arrive during the next tile)
    if constexpr (DROP) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ABL != 3) cur ^= 1;
  };
  const int nfast = causal ? 0 : min(Tk / TROWS, ntiles);  // full, unmasked tiles first
  int tile = t0;
  for (; tile < nfast; ++tile) body(tile, std::false_type{});
  for (; tile < ntiles; ++tile) body(tile, std::true_type{});
  l += __shfl_xor(l, 32, 64);
  if (ks > 1) {  // partial of this key range: un-normalised o, (m, l); attn_combine_fwd_kernel finishes
    if (q < Tq) {
      const int64_t row = (((int64_t)b * d.H + head) * ks + sp) * Tq + q;
      store_owner_f32(o, P.part + row * HD, h);
      if (h == 0) {
        float* ml = P.part + (int64_t)d.B * d.H * ks * Tq * HD + row * 2;
        ml[0] = m;
        ml[1] = l;
      }
    }
    return;
  }
  const float inv = 1.0f / l;
  bf16_t* ob = reinterpret_cast<bf16_t*>(d.o) + b * d.o_sb + head * HD;
  store_owner(o, ob, d.o_st, q, Tq, h, DROP ? inv * P.keep_scale : inv);
  if (h == 0 && q < Tq) {
    float* st = d.stats + ((b * d.H + head) * Tq + q) * 2;
    st[0] = m;  // log2 units
    st[1] = inv;
  }
}

// ------------------------------------------------------------------ dQ pass (owner = query)
// block -> (block of the pair, head, batch): the XCD-aware 1-D grid of pick_grid, or the plain 3-D one
__device__ __forceinline__ void block_coords(const AttnP& P, int& bx, int& head, int64_t& b) {
  if (P.gx > 0) {
    const int L = (int)blockIdx.x, r = L >> 3;
    const int bh = (r / P.gx) * 8 + (L & 7);
    bx = r % P.gx;
    head = bh % (int)P.d.H;
    b = bh / (int)P.d.H;
  } else {
    bx = (int)blockIdx.x;
    head = (int)blockIdx.y;
    b = (int64_t)blockIdx.z;
  }
}

template <bool DROP>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnP& P, const int bx, const int head, const int64_t b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K img | V img]
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int ks = P.ksplit, qt = bx / ks, sp = bx - qt * ks;
  const int q = qt * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bool causal = d.mask_mode == 1;
  const float c2 = P.c2;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(d.o) + b * d.o_sb + head * HD;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(d.d_o) + b * d.do_sb + head * HD;

  bf16x8 qf[4], dof[4];
  float delta = 0.f;
  {
    bf16x8 of[4];
    load_owner(qf, qb, d.q_st, q, Tq, h);
    load_owner(dof, dob, d.do_st, q, Tq, h);
    load_owner(of, ob, d.o_st, q, Tq, h);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[kk][j] * (float)of[kk][j];
  }
  delta += __shfl_xor(delta, 32, 64);
  float m = 0.f, linv = 0.f;
  if (q < Tq) {
    const int64_t si = (b * d.H + head) * Tq + q;
    m = d.stats[si * 2];
    linv = d.stats[si * 2 + 1];
    if (h == 0) d.delta[si] = delta;
  }
  PIN(m);
  PIN(linv);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    PIN(qf[kk]);
    PIN(dof[kk]);
  }
  // p = exp2(x - m) / l = exp2(x - M), M = m + log2(l): one number is enough wherever no row is
  // fully masked (mask_mode 0); the masked mode keeps (m, 1/l) because there m = -1.44e9 absorbs log2(l)
  const float nM = lg2(linv) - m;
  f32x16 dq[2];
  ZERO2(dq);
  const uint32_t ks_bits = __float_as_uint(P.keep_scale);
  // the stored keep bits of this lane's scores: the word the forward lane of the same (query, lane half) wrote per key tile
  const uint32_t* mrow = nullptr;
  if constexpr (DROP) mrow = P.dmask + (((int64_t)(b * d.H + head) * P.NT) * 2 + h) * P.TQP + (q & ~63) + mask_pos(q & 63);

  const LaneSrc Ks = lane_src(kb, d.k_st, Tk, wave, lane);
  const LaneSrc Vs = lane_src(vb, d.v_st, Tk, wave, lane);
  const int ntiles_all = (Tk + TROWS - 1) / TROWS;
  const int per = (ntiles_all + ks - 1) / ks;
  const int t0 = sp * per, ntiles = min(ntiles_all, t0 + per);  // this workgroup's key tiles [t0, ntiles)
  uint32_t wcur = 0;
  if constexpr (DROP) wcur = mrow[(int64_t)t0 * (2 * P.TQP)];
  stage_tile(smem, Ks, t0 * TROWS, wave, lane);
  stage_tile(smem + IMG, Vs, t0 * TROWS, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PIN(wcur);
  int cur = 0;
  auto body = [&](int tile, auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    const char* Kimg = smem + cur * 2 * IMG;
    const char* Vimg = Kimg + IMG;
    uint32_t wnext = 0;
    if (tile + 1 < ntiles) {
      char* nx = smem + (cur ^ 1) * 2 * IMG;
      stage_tile(nx, Ks, (tile + 1) * TROWS, wave, lane);
      stage_tile(nx + IMG, Vs, (tile + 1) * TROWS, wave, lane);
      if constexpr (DROP) wnext = mrow[(int64_t)(tile + 1) * (2 * P.TQP)];  // under this tile's MFMAs, like the DMA
    }
    const int key0 = tile * TROWS;
    static_for(std::make_integer_sequence<int, 2>{}, [&](auto rbk_c) {
      constexpr int rbk = decltype(rbk_c)::value;
      f32x16 s = first_product(Kimg, 32 * rbk, qf, c, h);    // s[key][q]
      f32x16 dp = first_product(Vimg, 32 * rbk, dof, c, h);  // dp[key][q]
      // With dropout, d(p) = keep * keep_scale * d(dropped p): the factor is the keep bit of the score spread over a word
      // (v_bfe_i32) and ANDed onto keep_scale - two instructions per score, no generator.  Bit of accumulator register e:
      // pair 8 rbk + 2 (e >> 2) + ((e >> 1) & 1), + 16 for the odd key of the pair.
      static_for(std::make_integer_sequence<int, 16>{}, [&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        float kf = 1.f;
        if constexpr (DROP) kf = __uint_as_float(bit_ones<8 * rbk + 2 * (e >> 2) + ((e >> 1) & 1) + 16 * (e & 1)>(wcur) & ks_bits);
        if constexpr (!edge) {
          const float t = DROP ? fmaf(dp[e], kf, -delta) : dp[e] - delta;
          s[e] = ex2(fmaf(s[e], c2, nM)) * t;  // dS
        } else {
          const int key = key0 + 32 * rbk + acc_row(e, h);
          float x = s[e] * c2;
          if (causal && key <= q) x = x + MASKED2;
          const float pe = (key < Tk) ? ex2(x - m) * linv : 0.f;
          s[e] = pe * (DROP ? fmaf(dp[e], kf, -delta) : dp[e] - delta);
        }
      });
      second_product(Kimg, 32 * rbk, s, dq, lane);  // dQt[d][q] += sum_key K[key][d] dS[key][q]
    });
    PIN(wnext);  // the compiler's wait for this load belongs here, not after the next iteration's DMA issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wcur = wnext;
    cur ^= 1;
  };
  const int nfast = causal ? 0 : min(Tk / TROWS, ntiles);
  int tile = t0;
  for (; tile < nfast; ++tile) body(tile, std::false_type{});
  for (; tile < ntiles; ++tile) body(tile, std::true_type{});
  if (ks > 1) {  // partial dQ of this key range (unscaled); attn_combine_dq_kernel sums and scales
    if (q < Tq) store_owner_f32(dq, P.part + ((((int64_t)b * d.H + head) * ks + sp) * Tq + q) * HD, h);
    return;
  }
  bf16_t* dqb = reinterpret_cast<bf16_t*>(d.dq) + b * d.dq_sb + head * HD;
  store_owner(dq, dqb, d.dq_st, q, Tq, h, P.dq_scale * P.sscale);
}

template <bool DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dq_kernel(const AttnP P) {
  int bx, head;
  int64_t b;
  block_coords(P, bx, head, b);
  attn_bwd_dq_body<DROP>(P, bx, head, b);
}

// ------------------------------------------------------------------ dK/dV pass (owner = key)
constexpr int NCONST = 4;  // per streamed query row: m, 1/l, delta (/ keep_scale with dropout), -(m + log2 l)
// LOCAL_DELTA: delta = rowsum(dO . O) of the streamed query rows is
// computed here (in the dQ pass's summation order) instead of read from d.delta - the pass then depends on nothing the dQ
// pass writes, and the two can share one launch (attn_bwd_small_kernel).
template <bool DROP, bool LOCAL_DELTA>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnP& P, const int bx, const int head, const int64_t b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][Q img | dO img] then [2][NCONST][64] floats
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  // Which key a lane owns is free (its K / V rows are gathered and its dK / dV rows scattered row by row anyway).  With
  // dropout a wave owns the 32 keys of ONE word of the stored mask - key tile 2 blockIdx.x + (wave >> 1), forward lane half
  // wave & 1 - with lane c on bit c of the word (bit j / 16 + j = even / odd key of pair j, see mask_pos): the word of a
  // query row is then, bit for bit, the select mask of that row over the wave's lanes.
  int key;
  if constexpr (DROP) {
    const int j = c & 15;
    key = (2 * bx + (wave >> 1)) * 64 + 32 * (j >> 3) + 8 * ((j >> 1) & 3) + 4 * (wave & 1) + 2 * (j & 1) + (c >> 4);
  } else {
    key = bx * 128 + wave * 32 + c;
  }
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bool causal = d.mask_mode == 1;
  const float c2 = P.c2;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(d.d_o) + b * d.do_sb + head * HD;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(d.o) + b * d.o_sb + head * HD;
  const float* stats = d.stats + (b * d.H + head) * Tq * 2;
  const float* deltas = d.delta + (b * d.H + head) * Tq;
  float* rowc_base = reinterpret_cast<float*>(smem + 4 * IMG);

  bf16x8 kf[4], vf[4];
  load_owner(kf, kb, d.k_st, key, Tk, h);
  load_owner(vf, vb, d.v_st, key, Tk, h);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    PIN(kf[kk]);
    PIN(vf[kk]);
  }
  f32x16 dk[2], dv[2];
  ZERO2(dk);
  ZERO2(dv);
  // The wave's rows of the stored mask, as 64-bit scalars: pair (rbk, e) of query tile `tile` = words of rows
  // acc_row(e, 0) | acc_row(e, 1) of that 32-row half = the select mask of accumulator register e (scalar loads: the
  // address is wave-uniform, the data is read-only here)
  // acc_row(e, 1) of that 32-row half = the select mask of accumulator register e.  The 64 words of a tile are fetched by
  // four s_load_dwordx16 issued at the END of the previous tile, just before its vmcnt(0) + barrier (scalar loads share
  // lgkmcnt with the LDS reads and return out of order, so the first LDS wait after them drains them: the only place their
  // latency hides is a stretch without LDS waits, and the tile seam is one), and waited for right after the barrier; a
  // vector load of the tile after that one pulls its lines into L2 under a whole tile of work.  Inline asm because hipcc
  // issues a scalar load and waits for it in the same breath.
  typedef __attribute__((ext_vector_type(16))) uint32_t u32x16;
  const uint32_t* mwave = nullptr;
  if constexpr (DROP)
    mwave = (P.dmask + ((((int64_t)(b * d.H + head) * P.NT) + min(2 * bx + (wave >> 1), P.NT - 1)) * 2 + (wave & 1)) * P.TQP);  // (a wave past the last key tile owns no key: any row will do)
  const float inv_ks = DROP ? 1.0f / P.keep_scale : 1.0f;

  // per-tile row constants: thread t carries (which = t / 64, row = t % 64)
  // (the raw loads are combined only when they are written to LDS at the end of the iteration, so
  // that nothing waits on them while the next tile's DMA is in flight)
  const int cwhich = threadIdx.x >> 6;
  auto load_consts = [&](int row0, float& x0, float& x1) {
    const int qi = row0 + (threadIdx.x & 63);
    x0 = 0.f;
    x1 = 1.f;
    if (qi < Tq) {
      if (LOCAL_DELTA && cwhich == 2) {
        // (the dQ pass's order: lane half h sums chunks 16 kk + 8 h + j in (kk, j) order, then the halves are added)
        const bf16_t* orow = ob + (int64_t)qi * d.o_st;
        const bf16_t* drow = dob + (int64_t)qi * d.do_st;
        float part[2] = {0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + kk * 16 + hh * 8);
            const bf16x8 dv_ = *reinterpret_cast<const bf16x8*>(drow + kk * 16 + hh * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) part[hh] += (float)dv_[j] * (float)ov[j];
          }
        x0 = part[0] + part[1];
      } else {
        x0 = cwhich == 2 ? deltas[qi] : stats[qi * 2];
      }
      x1 = stats[qi * 2 + 1];
    }
  };
  // with dropout dS = keep_scale * (P_kept * dP - P * delta / keep_scale): the factor goes to the final store of dK
  auto make_const = [&](float x0, float x1) -> float {
    return cwhich == 1 ? x1 : (cwhich == 3 ? lg2(x1) - x0 : (cwhich == 2 ? x0 * inv_ks : x0));
  };

  const LaneSrc Qs = lane_src(qb, d.q_st, Tq, wave, lane);
  const LaneSrc Os = lane_src(dob, d.do_st, Tq, wave, lane);
  const int ntiles = (Tq + TROWS - 1) / TROWS;
  stage_tile(smem, Qs, 0, wave, lane);
  stage_tile(smem + IMG, Os, 0, wave, lane);
  {
    float x0, x1;
    load_consts(0, x0, x1);
    rowc_base[threadIdx.x] = make_const(x0, x1);
  }
  u32x16 mk0, mk1, mk2, mk3;  // select masks of the 32-row halves: (mk0, mk1) rows 0..31, (mk2, mk3) rows 32..63; 32 SGPRs live at a time
  // (the two warm-up dwords land in registers of their own that stay allocated until wait_masks0: a scalar load writes its
  // destination whenever it returns, so a destination the compiler believes dead - and hands to an address computation - would
  // be overwritten behind its back)
  uint32_t warm0 = 0, warm1 = 0;
  auto issue_masks0 = [&](int tile) {  // first half of a tile + one dword of each line of the second half (scalar-cache warm-up)
    const uint32_t* mt = mwave + tile * 64;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0"
                 : "=&s"(mk0), "=&s"(mk1), "=&s"(warm0), "=&s"(warm1) : "s"(mt) : "memory");
  };
  auto issue_masks1 = [&](int tile) {
    const uint32_t* mt = mwave + tile * 64;
    asm volatile("s_load_dwordx16 %0, %2, 0x80\n\ts_load_dwordx16 %1, %2, 0xc0" : "=&s"(mk2), "=&s"(mk3) : "s"(mt) : "memory");
  };
  auto wait_masks0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mk0), "+s"(mk1), "+s"(warm0), "+s"(warm1)::"memory"); };
  auto wait_masks1 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mk2), "+s"(mk3)::"memory"); };
  if constexpr (DROP) issue_masks0(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (DROP) wait_masks0();
  int cur = 0;
  auto body = [&](int tile, auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    const char* Qimg = smem + cur * 2 * IMG;
    const char* Oimg = Qimg + IMG;
    const float* rowc = rowc_base + cur * (NCONST * 64);
    float cn0 = 0.f, cn1 = 1.f;
    uint32_t warm = 0;
    const bool more = tile + 1 < ntiles;
    if (more) {
      char* nx = smem + (cur ^ 1) * 2 * IMG;
      stage_tile(nx, Qs, (tile + 1) * TROWS, wave, lane);
      stage_tile(nx + IMG, Os, (tile + 1) * TROWS, wave, lane);
      load_consts((tile + 1) * TROWS, cn0, cn1);
      if constexpr (DROP) {
        if (tile + 2 < ntiles) warm = mwave[(tile + 2) * 64 + lane];
      }
    }
    const int q0 = tile * TROWS;
    // select mask of accumulator register e of 32-row half rbk
    auto keep_of = [&](int rbk, int e) -> bool {
      const int i = 2 * ((8 * rbk + (e & 7)));  // dword index inside the 16-dword chunk 2 rbk + (e >> 3)
      const u32x16& v = (rbk == 0) ? ((e < 8) ? mk0 : mk1) : ((e < 8) ? mk2 : mk3);
      const int j = i & 15;
      return __builtin_amdgcn_inverse_ballot_w64(((uint64_t)v[j + 1] << 32) | v[j]);
    };
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      f32x16 s = first_product(Qimg, 32 * rbk, kf, c, h);   // s[q][key]
      f32x16 dp = first_product(Oimg, 32 * rbk, vf, c, h);  // dp[q][key]
      f32x16 ds;
      if constexpr (DROP) {
        if (rbk == 1) wait_masks1();
      }
      if constexpr (!edge) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = 32 * rbk + 8 * g + 4 * h;
          const f32x4 nM = *reinterpret_cast<const f32x4*>(rowc + 192 + r);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(rowc + 128 + r);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            const float pe = ex2(fmaf(s[e], c2, nM[i]));
            if constexpr (DROP) {  // dV sees the dropped probabilities; dS / keep_scale = P_kept * dP - P * (delta / keep_scale)
              s[e] = keep_of(rbk, e) ? pe : 0.f;  // v_cndmask on a scalar pair
              ds[e] = fmaf(s[e], dp[e], -(pe * dl[i]));
            } else {
              s[e] = pe;
              ds[e] = pe * (dp[e] - dl[i]);
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = 32 * rbk + acc_row(e, h);
          const int qi = q0 + r;
          float x = s[e] * c2;
          if (causal && key <= qi) x = x + MASKED2;
          const float pe = (qi < Tq) ? ex2(x - rowc[r]) * rowc[64 + r] : 0.f;
          if constexpr (DROP) {
            s[e] = keep_of(rbk, e) ? pe : 0.f;
            ds[e] = fmaf(s[e], dp[e], -(pe * rowc[128 + r]));
          } else {
            s[e] = pe;
            ds[e] = pe * (dp[e] - rowc[128 + r]);
          }
        }
      }
      if constexpr (DROP) {
        // second half's masks: issued once the first half's are spent (same registers), a scalar-cache hit by now, in flight
        // under the 16 MFMAs below and the next 8; waited for where that half's arithmetic starts
        if (rbk == 0) issue_masks1(tile);
      }
      second_product(Oimg, 32 * rbk, s, dv, lane);   // dVt[d][key] += sum_q dO[q][d] P[q][key]
      second_product(Qimg, 32 * rbk, ds, dk, lane);  // dKt[d][key] += sum_q Q[q][d] dS[q][key]
    }
    if (more) rowc_base[(cur ^ 1) * (NCONST * 64) + threadIdx.x] = make_const(cn0, cn1);
    PIN(cn0);  // the compiler's wait for these two loads belongs here, on every path, not after the
    PIN(cn1);  // next iteration's DMA issue
    PIN(warm);
    if constexpr (DROP) {
      if (more) issue_masks0(tile + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DROP) wait_masks0();
    cur ^= 1;
  };
  const int nfast = causal ? 0 : Tq / TROWS;
  int tile = 0;
  for (; tile < nfast; ++tile) body(tile, std::false_type{});
  for (; tile < ntiles; ++tile) body(tile, std::true_type{});
  bf16_t* dkb = reinterpret_cast<bf16_t*>(d.dk) + b * d.dk_sb + head * HD;
  bf16_t* dvb = reinterpret_cast<bf16_t*>(d.dv) + b * d.dv_sb + head * HD;
  store_owner(dk, dkb, d.dk_st, key, Tk, h, DROP ? P.sscale * P.keep_scale : P.sscale);
  store_owner(dv, dvb, d.dv_st, key, Tk, h, DROP ? P.keep_scale : 1.0f);
}

template <bool DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkv_kernel(const AttnP P) {
  int bx, head;
  int64_t b;
  block_coords(P, bx, head, b);
  attn_bwd_dkv_body<DROP, false>(P, bx, head, b);
}

// Both backward passes of a SMALL problem in one launch (the decoder's self-attention, Wav2Vec2's: at most two query blocks
// and two key blocks - each pass is a 10-13 us kernel that is mostly launch latency): blocks [0, P.nq) of a (batch, head) pair run
// the dQ pass, the rest the dK/dV pass with the row sums delta computed locally.  Plain 3-D grid.
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(const AttnP P) {
  const int bx = (int)blockIdx.x;
  if (bx < P.nq)
    attn_bwd_dq_body<DROP>(P, bx, (int)blockIdx.y, (int64_t)blockIdx.z);
  else
    attn_bwd_dkv_body<DROP, true>(P, bx - P.nq, (int)blockIdx.y, (int64_t)blockIdx.z);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool ok_mat(const void* p, int64_t sb, int64_t st) { return p && al16(p) && sb % 8 == 0 && st % 8 == 0; }

#ifndef TMI_ATTN_DROP_OCC
#define TMI_ATTN_DROP_OCC 2
#endif
#ifndef TMI_ATTN_DQ_DROP_OCC
#define TMI_ATTN_DQ_DROP_OCC 2
#endif
// workgroups per CU of the dropout variants (the forward's generator and mask packing need registers: 3 per CU spills, and
// measured 124.5 us against 122.9 at 2 per CU on the encoder layer; the dQ kernel spills at 3 per CU too)
constexpr int DROP_OCC = TMI_ATTN_DROP_OCC, DQ_DROP_OCC = TMI_ATTN_DQ_DROP_OCC;

int64_t mask_tiles(int64_t Tk) { return (Tk + TROWS - 1) / TROWS; }
int64_t mask_rows(int64_t Tq) { return (Tq + 127) / 128 * 128; }

void set_dropout(AttnP& P) {
  P.drop_thr = tmi_drop_thr(P.d.dropout_p);
  P.keep_scale = tmi_keep_scale(P.drop_thr);
  P.seed_lo = (uint32_t)P.d.dropout_seed;
  P.seed_hi = (uint32_t)(P.d.dropout_seed >> 32);
  P.tm1 = (((uint32_t)((int32_t)P.drop_thr - 32768 - 1)) & 0xffffu) * 0x10001u;
  P.dmask = reinterpret_cast<uint32_t*>(P.d.drop_mask);
  P.NT = (int)mask_tiles(P.d.Tk);
  P.TQP = (int)mask_rows(P.d.Tq);
}

// key split of the forward / dQ passes: a single query tile against >= 8 key tiles with no causal mask, when the caller
// gave a workspace (tmi_attn_workspace_bytes).  2..8 ranges, never an empty one.
int pick_ksplit(const tmi_attn_desc& d) {
  static const int off = [] { const char* e = getenv("TMI_ATTN_NO_KSPLIT"); return e ? atoi(e) : 0; }();
  const int64_t ntiles = (d.Tk + TROWS - 1) / TROWS;
  if (off || d.mask_mode != 0 || d.Tq > 128 || ntiles < 8 || !d.workspace) return 1;
  static const int maxks = [] { const char* e = getenv("TMI_ATTN_KSPLIT_MAX"); return e ? atoi(e) : 4; }();  // measured: 4 ranges 9.03, 8 ranges 9.10, one chain 9.17 ms/step
  int ks = maxks < 1 ? 1 : (maxks > 8 ? 8 : maxks);
  while (ks > 1) {
    const int64_t per = (ntiles + ks - 1) / ks;
    if ((ks - 1) * per < ntiles && d.B * d.H * ks <= 1024 &&
        (int64_t)d.B * d.H * ks * d.Tq * (HD + 2) * 4 <= d.workspace_bytes && (reinterpret_cast<uintptr_t>(d.workspace) & 15) == 0)
      break;
    --ks;
  }
  return ks;
}

// launch grid of `gx` blocks per (batch, head): XCD-aware 1-D form when the pairs divide over the 8 XCDs and there is more
// than one block per pair to share rows (TMI_ATTN_XCD=0 switches it off)
dim3 pick_grid(AttnP& P, unsigned gx) {
  static const int on = [] { const char* e = getenv("TMI_ATTN_XCD"); return e ? atoi(e) : 1; }();
  const int64_t bh = P.d.B * P.d.H;
  if (on && gx > 1 && bh % 8 == 0 && bh * gx < (1ll << 31)) {
    P.gx = (int)gx;
    return dim3((unsigned)(bh * gx), 1, 1);
  }
  P.gx = 0;
  return dim3(gx, (unsigned)P.d.H, (unsigned)P.d.B);
}

int check_common(const tmi_attn_desc& d) {
  if (d.B <= 0 || d.H <= 0 || d.Tq <= 0 || d.Tk <= 0 || d.B > 65535 || d.H > 65535 ||
      (d.mask_mode != 0 && d.mask_mode != 1) || !d.stats || d.score_scale < 0.f || !(d.dropout_p >= 0.f && tmi_drop_ok(d.dropout_p)) ||
      (d.dropout_p > 0.f && d.Tk > TMI_DROP_MAX_COLS))
    return 0;
  if (d.dropout_p > 0.f && tmi_drop_thr(d.dropout_p) > 0 &&
      (!d.drop_mask || !al16(d.drop_mask) || d.drop_mask_bytes < tmi_attn_dropmask_bytes(d.B, d.H, d.Tq, d.Tk)))
    return 0;
  return ok_mat(d.q, d.q_sb, d.q_st) && ok_mat(d.k, d.k_sb, d.k_st) && ok_mat(d.v, d.v_sb, d.v_st) &&
         ok_mat(d.o, d.o_sb, d.o_st);
}

}  // namespace

extern "C" int64_t tmi_attn_dropmask_bytes(int64_t B, int64_t H, int64_t Tq, int64_t Tk) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return 0;
  return B * H * mask_tiles(Tk) * 2 * mask_rows(Tq) * 4;
}

extern "C" int64_t tmi_attn_workspace_bytes(int64_t B, int64_t H, int64_t Tq) {
  return Tq <= 128 ? B * H * 8 * Tq * (HD + 2) * 4 : 0;
}

static int tmi_attn_fwd_impl(const tmi_attn_desc* dp, void* stream);
extern "C" int tmi_attn_fwd(const tmi_attn_desc* dp, void* stream) {
  if (tmi_plan_recording() && dp) {
    const tmi_attn_desc c_ = *dp;
    tmi_plan_push([c_, stream]() -> int {
      tmi_attn_desc e_ = c_;
      if (e_.dropout_p > 0.f) e_.dropout_seed += tmi_plan_seed_delta();
      return tmi_attn_fwd(&e_, stream);
    });
  }
  tmi_plan_enter();
  const int rc_ = tmi_attn_fwd_impl(dp, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_attn_fwd_impl(const tmi_attn_desc* dp, void* stream) {
  if (!dp || !check_common(*dp)) {
    tmi_set_error("tmi_attn_fwd: bad argument (16-byte aligned bf16 operands, strides multiple of 8; dropout_p > 0 needs drop_mask of tmi_attn_dropmask_bytes)");
    return TMI_ERR_INVALID;
  }
  AttnP P;
  P.d = *dp;
  P.dq_scale = 1.f;
  P.sscale = dp->score_scale != 0.f ? dp->score_scale : 1.f;
  P.c2 = P.sscale * LOG2E;
  set_dropout(P);
  P.ksplit = pick_ksplit(*dp);
  P.part = reinterpret_cast<float*>(dp->workspace);
  dim3 grid = pick_grid(P, (unsigned)((dp->Tq + 127) / 128 * P.ksplit));
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  // (workgroups per CU: 2 with dropout, 3 without - measured, tools/ab_attn.sh; the other occupancies (TMI_ATTN_FWD_OCC,
  // ..._DQ_OCC, ..._DKV_OCC: they spill) and the ablation kernels are built with `make EXPERIMENTS=1` only)
#ifdef TMI_ATTN_EXPERIMENTS
  static const int focc = [] { const char* e = getenv("TMI_ATTN_FWD_OCC"); return e ? atoi(e) : 0; }();
#else
  constexpr int focc = 0;
#endif
  if (P.drop_thr) {
#ifdef TMI_ATTN_EXPERIMENTS
    if (focc == 4) hipLaunchKernelGGL((attn_fwd_kernel<true, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else if (focc == 5) hipLaunchKernelGGL((attn_fwd_kernel<true, 5>), grid, dim3(256), 4 * IMG, hs, P);
    else if (focc == 3) hipLaunchKernelGGL((attn_fwd_kernel<true, 3>), grid, dim3(256), 4 * IMG, hs, P);
    else
#endif
    hipLaunchKernelGGL((attn_fwd_kernel<true, DROP_OCC>), grid, dim3(256), 4 * IMG, hs, P);
  } else if (focc == 4 || focc == 5) {
#ifdef TMI_ATTN_EXPERIMENTS
    if (focc == 4) hipLaunchKernelGGL((attn_fwd_kernel<false, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 5>), grid, dim3(256), 4 * IMG, hs, P);
#endif
  } else {
#ifdef TMI_ATTN_EXPERIMENTS  // ablation builds of the forward (diagnostics; not in the shipped library)
    static const int abl = [] { const char* e = getenv("TMI_ATTN_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 1>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 2) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 2>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 3) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 3>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 4) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else
#endif
    hipLaunchKernelGGL((attn_fwd_kernel<false, 3>), grid, dim3(256), 4 * IMG, hs, P);
  }
  if (P.ksplit > 1) {
    const int64_t rows = dp->B * dp->H * dp->Tq;
    hipLaunchKernelGGL((attn_combine_kernel<true>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, hs, P.part, P.ksplit,
                       dp->B * dp->H, (int)dp->Tq, reinterpret_cast<bf16_t*>(dp->o), dp->o_sb, dp->o_st, (int)dp->H, dp->stats,
                       P.drop_thr ? P.keep_scale : 1.f);
  }
  return tmi_check_launch("tmi_attn_fwd");
}

static int tmi_attn_bwd_impl(const tmi_attn_desc* dp, void* stream);
extern "C" int tmi_attn_bwd(const tmi_attn_desc* dp, void* stream) {
  if (tmi_plan_recording() && dp) {
    const tmi_attn_desc c_ = *dp;
    tmi_plan_push([c_, stream]() -> int {
      tmi_attn_desc e_ = c_;
      if (e_.dropout_p > 0.f) e_.dropout_seed += tmi_plan_seed_delta();
      return tmi_attn_bwd(&e_, stream);
    });
  }
  tmi_plan_enter();
  const int rc_ = tmi_attn_bwd_impl(dp, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_attn_bwd_impl(const tmi_attn_desc* dp, void* stream) {
  if (!dp || !check_common(*dp) || !ok_mat(dp->d_o, dp->do_sb, dp->do_st) || !ok_mat(dp->dq, dp->dq_sb, dp->dq_st) ||
      !ok_mat(dp->dk, dp->dk_sb, dp->dk_st) || !ok_mat(dp->dv, dp->dv_sb, dp->dv_st) || !dp->delta) {
    tmi_set_error("tmi_attn_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  AttnP P;
  P.d = *dp;
  P.dq_scale = dp->dq_scale;
  P.sscale = dp->score_scale != 0.f ? dp->score_scale : 1.f;
  P.c2 = P.sscale * LOG2E;
  set_dropout(P);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool do_dq = dp->bwd_passes != 2, do_dkv = dp->bwd_passes != 1;
  if (dp->bwd_passes < 0 || dp->bwd_passes > 3) {
    tmi_set_error("tmi_attn_bwd: bwd_passes must be 0 (both), 1 (dQ), 2 (dK/dV) or 3 (both)");
    return TMI_ERR_INVALID;
  }
  P.ksplit = pick_ksplit(*dp);
  P.part = reinterpret_cast<float*>(dp->workspace);
  P.nq = 0;
  // one or two query blocks against one or two key blocks (decoder self-attention, Wav2Vec2 at 2 s and 5 s clips): each pass
  // is a 10-13 us kernel, mostly launch latency, so both go out as ONE launch (TMI_ATTN_BWD_FUSE=0: two)
  static const int fuse = [] { const char* e = getenv("TMI_ATTN_BWD_FUSE"); return e ? atoi(e) : 1; }();
  if (fuse && do_dq && do_dkv && P.ksplit == 1 && dp->Tq <= 256 && dp->Tk <= 256) {
    P.gx = 0;
    P.nq = (int)((dp->Tq + 127) / 128);
    const dim3 g((unsigned)(P.nq + (dp->Tk + 127) / 128), (unsigned)dp->H, (unsigned)dp->B);
    const size_t lds = 4 * IMG + 2 * NCONST * 64 * sizeof(float);
    if (P.drop_thr) hipLaunchKernelGGL((attn_bwd_small_kernel<true>), g, dim3(256), lds, s, P);
    else hipLaunchKernelGGL((attn_bwd_small_kernel<false>), g, dim3(256), lds, s, P);
    return tmi_check_launch("tmi_attn_bwd(small)");
  }
  if (do_dq) {
  dim3 gq = pick_grid(P, (unsigned)((dp->Tq + 127) / 128 * P.ksplit));
#ifdef TMI_ATTN_EXPERIMENTS
  static const int qocc = [] { const char* e = getenv("TMI_ATTN_DQ_OCC"); return e ? atoi(e) : 0; }();
#else
  constexpr int qocc = 0;
#endif
  if (P.drop_thr) {
#ifdef TMI_ATTN_EXPERIMENTS
    if (qocc == 3) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 3>), gq, dim3(256), 4 * IMG, s, P);
    else if (qocc == 4) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 4>), gq, dim3(256), 4 * IMG, s, P);
    else if (qocc == 5) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 5>), gq, dim3(256), 4 * IMG, s, P);
    else
#endif
    hipLaunchKernelGGL((attn_bwd_dq_kernel<true, DQ_DROP_OCC>), gq, dim3(256), 4 * IMG, s, P);
  } else if (qocc == 4 || qocc == 5) {
#ifdef TMI_ATTN_EXPERIMENTS
    if (qocc == 4) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 4>), gq, dim3(256), 4 * IMG, s, P);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 5>), gq, dim3(256), 4 * IMG, s, P);
#endif
  } else
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 3>), gq, dim3(256), 4 * IMG, s, P);
  if (P.ksplit > 1) {
    const int64_t rows = dp->B * dp->H * dp->Tq;
    hipLaunchKernelGGL((attn_combine_kernel<false>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, P.part, P.ksplit,
                       dp->B * dp->H, (int)dp->Tq, reinterpret_cast<bf16_t*>(dp->dq), dp->dq_sb, dp->dq_st, (int)dp->H,
                       (float*)nullptr, P.dq_scale * P.sscale);
  }
  int rc = tmi_check_launch("tmi_attn_bwd(dq)");
  if (rc) return rc;
  }
  if (!do_dkv) return TMI_OK;
  P.ksplit = 1;
  dim3 gk = pick_grid(P, (unsigned)((dp->Tk + 127) / 128));
  const size_t klds = 4 * IMG + 2 * NCONST * 64 * sizeof(float);
#ifdef TMI_ATTN_EXPERIMENTS
  static const int kocc = [] { const char* e = getenv("TMI_ATTN_DKV_OCC"); return e ? atoi(e) : 0; }();
  if (P.drop_thr && kocc == 3) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 3>), gk, dim3(256), klds, s, P);
  else if (P.drop_thr && kocc == 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 4>), gk, dim3(256), klds, s, P);
  else if (!P.drop_thr && kocc == 3) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 3>), gk, dim3(256), klds, s, P);
  else if (!P.drop_thr && kocc == 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 4>), gk, dim3(256), klds, s, P);
  else
#endif
  if (P.drop_thr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 2>), gk, dim3(256), klds, s, P);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 2>), gk, dim3(256), klds, s, P);
  return tmi_check_launch("tmi_attn_bwd(dkv)");
}
