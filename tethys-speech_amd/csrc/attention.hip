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
#include <stdlib.h>

namespace {

constexpr int HD = 64;
constexpr int TROWS = 64;             // streamed rows per tile
constexpr int IMG = TROWS * 128;      // bytes of one [64][64] bf16 image

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

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

// Yt[blk][d][o] += sum_s T[rb + s][32 blk + d] * X[s][o]   (X = fp32 accumulator, s = 0..31)
__device__ __forceinline__ void second_product(const char* img, int rb, const f32x16& x, f32x16 (&y)[2], int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int h = g >> 1, q = i >> 2, p = i & 3;
#pragma unroll
  for (int sI = 0; sI < 2; ++sI) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (bf16_t)x[8 * sI + j];
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
  // key split (forward and dQ passes of a short query side against a long key side: cross-attention, Tq <= 128): the
  // workgroups of one (batch, head) are `ksplit` disjoint key ranges; each leaves its un-normalised partial in `part`
  // ([B*H][ksplit][Tq][64] fp32, then for the forward [B*H][ksplit][Tq][2] = (m, l)) and a combine kernel folds them.
  int ksplit;
  float* part;
  int dbg;  // TMI_ATTN_DBG (diagnostics)
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
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int ks = P.ksplit, qt = (int)blockIdx.x / ks, sp = (int)blockIdx.x - qt * ks;
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
  const uint32_t drop_thr = P.drop_thr;
  const uint32_t skey = tmi_stream_key(((uint64_t)P.seed_hi << 32) | P.seed_lo, (uint32_t)(b * d.H + head));
  const tmi_rowkey qrk = tmi_row_key(skey, (uint32_t)q);  // the mask row of this lane's query: one avalanche per kernel

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
    if constexpr (DROP) {  // zero the dropped probabilities; the kept ones are rescaled on the final store
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t hh = tmi_pair_hash(qrk, (uint32_t)((key0 + 32 * rbk + 8 * g + 4 * h) >> 1) + j);
            if ((hh & 0xffffu) < drop_thr) s[rbk][4 * g + 2 * j] = 0.f;
            if ((hh >> 16) < drop_thr) s[rbk][4 * g + 2 * j + 1] = 0.f;
          }
    }
    if constexpr (ABL == 2) {
      asm volatile("" :: "v"(s[0]), "v"(s[1]));
    } else {
    second_product(Vimg, 0, s[0], o, lane);
    second_product(Vimg, 32, s[1], o, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (P.dbg == 1) __builtin_amdgcn_s_sleep(20);
    if (P.dbg == 2) __syncthreads();
    if (P.dbg == 3) { __builtin_amdgcn_s_sleep(20); __syncthreads(); }
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
template <bool DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dq_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K img | V img]
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int ks = P.ksplit, qt = (int)blockIdx.x / ks, sp = (int)blockIdx.x - qt * ks;
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
  const uint32_t drop_thr = P.drop_thr;
  const float keep_scale = P.keep_scale;
  const uint32_t skey = tmi_stream_key(((uint64_t)P.seed_hi << 32) | P.seed_lo, (uint32_t)(b * d.H + head));
  const tmi_rowkey qrk = tmi_row_key(skey, (uint32_t)q);  // the mask row of this lane's query: one avalanche per kernel

  const LaneSrc Ks = lane_src(kb, d.k_st, Tk, wave, lane);
  const LaneSrc Vs = lane_src(vb, d.v_st, Tk, wave, lane);
  const int ntiles_all = (Tk + TROWS - 1) / TROWS;
  const int per = (ntiles_all + ks - 1) / ks;
  const int t0 = sp * per, ntiles = min(ntiles_all, t0 + per);  // this workgroup's key tiles [t0, ntiles)
  stage_tile(smem, Ks, t0 * TROWS, wave, lane);
  stage_tile(smem + IMG, Vs, t0 * TROWS, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  auto body = [&](int tile, auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    const char* Kimg = smem + cur * 2 * IMG;
    const char* Vimg = Kimg + IMG;
    if (tile + 1 < ntiles) {
      char* nx = smem + (cur ^ 1) * 2 * IMG;
      stage_tile(nx, Ks, (tile + 1) * TROWS, wave, lane);
      stage_tile(nx + IMG, Vs, (tile + 1) * TROWS, wave, lane);
    }
    const int key0 = tile * TROWS;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      f32x16 s = first_product(Kimg, 32 * rbk, qf, c, h);    // s[key][q]
      f32x16 dp = first_product(Vimg, 32 * rbk, dof, c, h);  // dp[key][q]
      // with dropout, d(p) = mask / keep * d(dropped p): one hash per pair of keys
      auto dpm = [&](int e, uint32_t hh) -> float {
        if constexpr (DROP) return ((e & 1) ? (hh >> 16) : (hh & 0xffffu)) < drop_thr ? 0.f : dp[e] * keep_scale;
        else return dp[e];
      };
      // d(p) - delta with the mask as a factor of an fma (select on the constant, not on the product)
      auto dpm_minus = [&](int e, uint32_t hh, float dl_) -> float {
        if constexpr (DROP) return fmaf(dp[e], ((e & 1) ? (hh >> 16) : (hh & 0xffffu)) < drop_thr ? 0.f : keep_scale, -dl_);
        else return dp[e] - dl_;
      };
      auto pair_draw = [&](int e) -> uint32_t {  // e even
        if constexpr (DROP) return tmi_pair_hash(qrk, (uint32_t)((key0 + 32 * rbk + 8 * (e >> 2) + 4 * h) >> 1) + ((e >> 1) & 1));
        else return 0u;
      };
      if constexpr (!edge) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const uint32_t hh = pair_draw(e);
          s[e] = ex2(fmaf(s[e], c2, nM)) * dpm_minus(e, hh, delta);  // dS
          s[e + 1] = ex2(fmaf(s[e + 1], c2, nM)) * dpm_minus(e + 1, hh, delta);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + 32 * rbk + acc_row(e, h);
          float x = s[e] * c2;
          if (causal && key <= q) x = x + MASKED2;
          const float pe = (key < Tk) ? ex2(x - m) * linv : 0.f;
          s[e] = pe * (dpm(e, pair_draw(e & ~1)) - delta);
        }
      }
      second_product(Kimg, 32 * rbk, s, dq, lane);  // dQt[d][q] += sum_key K[key][d] dS[key][q]
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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

// ------------------------------------------------------------------ dK/dV pass (owner = key)
constexpr int NCONST = 6;  // per streamed query row: m, 1/l, delta, -(m + log2 l), the two dropout row-key words (bits)
template <bool DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkv_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][Q img | dO img] then [2][NCONST][64] floats
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int key = blockIdx.x * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bool causal = d.mask_mode == 1;
  const float c2 = P.c2;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(d.d_o) + b * d.do_sb + head * HD;
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
  const uint32_t drop_thr = P.drop_thr;
  const float keep_scale = P.keep_scale;
  const uint32_t skey = tmi_stream_key(((uint64_t)P.seed_hi << 32) | P.seed_lo, (uint32_t)(b * d.H + head));
  const uint32_t khalf = (uint32_t)key >> 1, ksh = ((uint32_t)key & 1u) * 16u;
  const uint32_t lane_mask = 0xffffu << ksh, lane_thr = drop_thr << ksh;  // this key's half of a pair hash, in place
  auto keep_at = [&](float ra, float rb) -> bool {  // this lane's key against the query row whose mask-row key is (ra, rb)
    const uint32_t hh = tmi_pair_hash(tmi_rowkey{__float_as_uint(ra), __float_as_uint(rb)}, khalf);
    return ((hh >> ksh) & 0xffffu) >= drop_thr;
  };
  // row keys of the 64 streamed query rows: one thread each, next to the softmax constants in LDS
  auto put_row_keys = [&](float* dst, int row0) {  // threads 0..63
    const tmi_rowkey rk = tmi_row_key(skey, (uint32_t)(row0 + (int)threadIdx.x));
    dst[256 + threadIdx.x] = __uint_as_float(rk.a);
    dst[320 + threadIdx.x] = __uint_as_float(rk.b);
  };

  // per-tile row constants: thread t carries (which = t / 64, row = t % 64)
  // (the raw loads are combined only when they are written to LDS at the end of the iteration, so
  // that nothing waits on them while the next tile's DMA is in flight)
  const int cwhich = threadIdx.x >> 6;
  auto load_consts = [&](int row0, float& x0, float& x1) {
    const int qi = row0 + (threadIdx.x & 63);
    x0 = 0.f;
    x1 = 1.f;
    if (qi < Tq) {
      x0 = cwhich == 2 ? deltas[qi] : stats[qi * 2];
      x1 = stats[qi * 2 + 1];
    }
  };
  auto make_const = [&](float x0, float x1) -> float {
    return cwhich == 1 ? x1 : (cwhich == 3 ? lg2(x1) - x0 : x0);
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
    if (DROP && threadIdx.x < 64) put_row_keys(rowc_base, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  auto body = [&](int tile, auto edge_tag) {
    constexpr bool edge = decltype(edge_tag)::value;
    const char* Qimg = smem + cur * 2 * IMG;
    const char* Oimg = Qimg + IMG;
    const float* rowc = rowc_base + cur * (NCONST * 64);
    float cn0 = 0.f, cn1 = 1.f;
    const bool more = tile + 1 < ntiles;
    if (more) {
      char* nx = smem + (cur ^ 1) * 2 * IMG;
      stage_tile(nx, Qs, (tile + 1) * TROWS, wave, lane);
      stage_tile(nx + IMG, Os, (tile + 1) * TROWS, wave, lane);
      load_consts((tile + 1) * TROWS, cn0, cn1);
    }
    const int q0 = tile * TROWS;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      f32x16 s = first_product(Qimg, 32 * rbk, kf, c, h);   // s[q][key]
      f32x16 dp = first_product(Oimg, 32 * rbk, vf, c, h);  // dp[q][key]
      f32x16 ds;
      if constexpr (!edge) {
        // Dropout draws: the pair hash of (query row, key >> 1) is the SAME number in the two lanes of a key pair (the even
        // key takes its low half, the odd key the high half), so each lane computes it for half of its 16 query rows -
        // the even lane for accumulator registers 0..7, the odd lane for 8..15 - and the pair swaps through DPP
        // (quad_perm): 8 hashes + 16 cross-lane moves per lane instead of 16 hashes.
        uint32_t hx[DROP ? 8 : 1];
        if constexpr (DROP) {
          const int odd = lane & 1;
#pragma unroll
          for (int jg = 0; jg < 2; ++jg) {
            const int r = 32 * rbk + 8 * (jg + 2 * odd) + 4 * h;  // rows of registers 4 jg + 8 odd .. + 3
            const f32x4 ra4 = *reinterpret_cast<const f32x4*>(rowc + 256 + r);
            const f32x4 rb4 = *reinterpret_cast<const f32x4*>(rowc + 320 + r);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              hx[4 * jg + i] = tmi_pair_hash(tmi_rowkey{__float_as_uint(ra4[i]), __float_as_uint(rb4[i])}, khalf);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = 32 * rbk + 8 * g + 4 * h;
          const f32x4 nM = *reinterpret_cast<const f32x4*>(rowc + 192 + r);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(rowc + 128 + r);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            const float pe = ex2(fmaf(s[e], c2, nM[i]));
            if constexpr (DROP) {  // dV sees the dropped probabilities, dS the masked dP: ds = p * (mask/keep * dp - delta)
              // registers 0..7: the even lane's hash (quad_perm [0,0,2,2]); 8..15: the odd lane's ([1,1,3,3]).  The swap is
              // the DPP operand of the AND that cuts this lane's 16 bits out of the 32 (low half for the even key, high
              // half - left in place, compared with the threshold shifted likewise - for the odd one): one instruction
              // instead of move + bit-field extract.  REQUIRES a full EXEC mask: a DPP read from an inactive lane returns
              // stale data (bound_ctrl is off), and all four lanes of a quad are active here only because nothing above
              // this `!edge` body exits lane-divergently - keep it that way.  (s_nop: a DPP read needs two wait states after the VALU write of
              // its source, and the hazard recogniser does not look inside asm.)
              uint32_t cut;
              if (e < 8)
                asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "=v"(cut) : "v"(hx[e & 7]), "v"(lane_mask));
              else
                asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "=v"(cut) : "v"(hx[e & 7]), "v"(lane_mask));
              const bool keep = cut >= lane_thr;
              s[e] = keep ? pe : 0.f;
              ds[e] = pe * fmaf(dp[e], keep ? keep_scale : 0.f, -dl[i]);  // (select on the constant: one op fewer than on the product)
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
            const bool keep = keep_at(rowc[256 + r], rowc[320 + r]);
            s[e] = keep ? pe : 0.f;
            ds[e] = pe * ((keep ? dp[e] * keep_scale : 0.f) - rowc[128 + r]);
          } else {
            s[e] = pe;
            ds[e] = pe * (dp[e] - rowc[128 + r]);
          }
        }
      }
      second_product(Oimg, 32 * rbk, s, dv, lane);   // dVt[d][key] += sum_q dO[q][d] P[q][key]
      second_product(Qimg, 32 * rbk, ds, dk, lane);  // dKt[d][key] += sum_q Q[q][d] dS[q][key]
    }
    if (more) {
      rowc_base[(cur ^ 1) * (NCONST * 64) + threadIdx.x] = make_const(cn0, cn1);
      if (DROP && threadIdx.x < 64) put_row_keys(rowc_base + (cur ^ 1) * (NCONST * 64), (tile + 1) * TROWS);
    }
    PIN(cn0);  // the compiler's wait for these two loads belongs here, on every path, not after the
    PIN(cn1);  // next iteration's DMA issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  };
  const int nfast = causal ? 0 : Tq / TROWS;
  int tile = 0;
  for (; tile < nfast; ++tile) body(tile, std::false_type{});
  for (; tile < ntiles; ++tile) body(tile, std::true_type{});
  bf16_t* dkb = reinterpret_cast<bf16_t*>(d.dk) + b * d.dk_sb + head * HD;
  bf16_t* dvb = reinterpret_cast<bf16_t*>(d.dv) + b * d.dv_sb + head * HD;
  store_owner(dk, dkb, d.dk_st, key, Tk, h, P.sscale);
  store_owner(dv, dvb, d.dv_st, key, Tk, h, DROP ? P.keep_scale : 1.0f);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool ok_mat(const void* p, int64_t sb, int64_t st) { return p && al16(p) && sb % 8 == 0 && st % 8 == 0; }

#ifndef TMI_ATTN_DROP_OCC
#define TMI_ATTN_DROP_OCC 3
#endif
#ifndef TMI_ATTN_DQ_DROP_OCC
#define TMI_ATTN_DQ_DROP_OCC 2
#endif
// workgroups per CU of the dropout variants (the hash needs registers: the dQ kernel spills at 3 per CU)
constexpr int DROP_OCC = TMI_ATTN_DROP_OCC, DQ_DROP_OCC = TMI_ATTN_DQ_DROP_OCC;

void set_dropout(AttnP& P) {
  P.drop_thr = tmi_drop_thr(P.d.dropout_p);
  P.keep_scale = tmi_keep_scale(P.drop_thr);
  P.seed_lo = (uint32_t)P.d.dropout_seed;
  P.seed_hi = (uint32_t)(P.d.dropout_seed >> 32);
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

int check_common(const tmi_attn_desc& d) {
  if (d.B <= 0 || d.H <= 0 || d.Tq <= 0 || d.Tk <= 0 || d.B > 65535 || d.H > 65535 ||
      (d.mask_mode != 0 && d.mask_mode != 1) || !d.stats || d.score_scale < 0.f || !(d.dropout_p >= 0.f && tmi_drop_ok(d.dropout_p)) ||
      (d.dropout_p > 0.f && d.Tk > TMI_DROP_MAX_COLS))
    return 0;
  return ok_mat(d.q, d.q_sb, d.q_st) && ok_mat(d.k, d.k_sb, d.k_st) && ok_mat(d.v, d.v_sb, d.v_st) &&
         ok_mat(d.o, d.o_sb, d.o_st);
}

}  // namespace

extern "C" int64_t tmi_attn_workspace_bytes(int64_t B, int64_t H, int64_t Tq) {
  return Tq <= 128 ? B * H * 8 * Tq * (HD + 2) * 4 : 0;
}

extern "C" int tmi_attn_fwd(const tmi_attn_desc* dp, void* stream) {
  if (!dp || !check_common(*dp)) {
    tmi_set_error("tmi_attn_fwd: bad argument (16-byte aligned bf16 operands, strides multiple of 8)");
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
  static const int adbg = [] { const char* e = getenv("TMI_ATTN_DBG"); return e ? atoi(e) : 0; }();
  P.dbg = adbg;
  dim3 grid((unsigned)((dp->Tq + 127) / 128 * P.ksplit), (unsigned)dp->H, (unsigned)dp->B);
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  static const int focc = [] { const char* e = getenv("TMI_ATTN_FWD_OCC"); return e ? atoi(e) : 0; }();
  if (P.drop_thr) {
    if (focc == 4) hipLaunchKernelGGL((attn_fwd_kernel<true, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else if (focc == 5) hipLaunchKernelGGL((attn_fwd_kernel<true, 5>), grid, dim3(256), 4 * IMG, hs, P);
    else if (focc == 2) hipLaunchKernelGGL((attn_fwd_kernel<true, 2>), grid, dim3(256), 4 * IMG, hs, P);
    else hipLaunchKernelGGL((attn_fwd_kernel<true, DROP_OCC>), grid, dim3(256), 4 * IMG, hs, P);
  } else if (focc == 4 || focc == 5) {
    if (focc == 4) hipLaunchKernelGGL((attn_fwd_kernel<false, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 5>), grid, dim3(256), 4 * IMG, hs, P);
  } else {
    static const int abl = [] { const char* e = getenv("TMI_ATTN_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 1>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 2) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 2>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 3) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 3>), grid, dim3(256), 4 * IMG, hs, P);
    else if (abl == 4) hipLaunchKernelGGL((attn_fwd_kernel<false, 3, 4>), grid, dim3(256), 4 * IMG, hs, P);
    else {
      static const int ldsdbg = [] { const char* e = getenv("TMI_ATTN_LDS"); return e ? atoi(e) : 0; }();  // diagnostics: co-residency
      if (ldsdbg > 4 * IMG) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsdbg);
      hipLaunchKernelGGL((attn_fwd_kernel<false, 3>), grid, dim3(256), ldsdbg > 4 * IMG ? ldsdbg : 4 * IMG, hs, P);
    }
  }
  if (P.ksplit > 1) {
    const int64_t rows = dp->B * dp->H * dp->Tq;
    hipLaunchKernelGGL((attn_combine_kernel<true>), dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, hs, P.part, P.ksplit,
                       dp->B * dp->H, (int)dp->Tq, reinterpret_cast<bf16_t*>(dp->o), dp->o_sb, dp->o_st, (int)dp->H, dp->stats,
                       P.drop_thr ? P.keep_scale : 1.f);
  }
  return tmi_check_launch("tmi_attn_fwd");
}

extern "C" int tmi_attn_bwd(const tmi_attn_desc* dp, void* stream) {
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
  if (do_dq) {
  dim3 gq((unsigned)((dp->Tq + 127) / 128 * P.ksplit), (unsigned)dp->H, (unsigned)dp->B);
  static const int qocc = [] { const char* e = getenv("TMI_ATTN_DQ_OCC"); return e ? atoi(e) : 0; }();
  if (P.drop_thr) {
    if (qocc == 3) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 3>), gq, dim3(256), 4 * IMG, s, P);
    else if (qocc == 4) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 4>), gq, dim3(256), 4 * IMG, s, P);
    else if (qocc == 5) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 5>), gq, dim3(256), 4 * IMG, s, P);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<true, DQ_DROP_OCC>), gq, dim3(256), 4 * IMG, s, P);
  } else if (qocc == 4 || qocc == 5) {
    if (qocc == 4) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 4>), gq, dim3(256), 4 * IMG, s, P);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 5>), gq, dim3(256), 4 * IMG, s, P);
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
  dim3 gk((unsigned)((dp->Tk + 127) / 128), (unsigned)dp->H, (unsigned)dp->B);
  static const int kocc = [] { const char* e = getenv("TMI_ATTN_DKV_OCC"); return e ? atoi(e) : 0; }();
  const size_t klds = 4 * IMG + 2 * NCONST * 64 * sizeof(float);
  if (P.drop_thr) {
    if (kocc == 3) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 3>), gk, dim3(256), klds, s, P);
    else if (kocc == 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 4>), gk, dim3(256), klds, s, P);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, 2>), gk, dim3(256), klds, s, P);
  } else {
    if (kocc == 3) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 3>), gk, dim3(256), klds, s, P);
    else if (kocc == 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 4>), gk, dim3(256), klds, s, P);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, 2>), gk, dim3(256), klds, s, P);
  }
  return tmi_check_launch("tmi_attn_bwd(dkv)");
}
