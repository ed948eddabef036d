// Fast bf16 GEMM path of tmi_gemm (gfx950).  One kernel template covers the three layouts
// that carry the step's FLOPs; each operand is independently
//   KC  k-contiguous rows   (A[m][k] / Bt[n][k]):  LDS image [128 rows][8 x 16 B], fragments by
//       ds_read_b128, swizzle phys_chunk = chunk ^ ((row >> 1) & 7)
//   KS  k-strided           (A[k][m] / B[k][n]):   LDS image [64 k][16 x 16 B], fragments by
//       ds_read_b64_tr_b16 (hardware transpose of a 4(k) x 16(col) block), swizzle
//       phys_chunk = chunk ^ ((k & 3) << 2)
// forward X·W = (KC, KS) straight from the natural Keras [in,out] kernel; dgrad dY·Wᵀ = (KC, KC);
// wgrad Xᵀ·dY = (KS, KS).  Two tile configurations (MFMA 32x32x16 bf16, BK = 64): 256x256 per
// 512-thread workgroup, 8 waves as 2x4, each wave 128x64 (128 accumulator registers) — twice
// the MFMA work per staged byte and per LDS-DMA instruction issued — and 128x128 per 256-thread
// workgroup (waves 2x2 of 64x64, two workgroups per CU) for narrow / small problems.  Tiles are
// staged global -> LDS directly (global_load_lds_dwordx4), double-buffered: the next K-tile's
// DMA is in flight under the current tile's MFMAs, one barrier per K-tile.
// The LDS image is lane-linear as the DMA requires; the swizzle is applied to the per-lane
// SOURCE address and again on the read.
// Epilogue: accumulators go through LDS (fp32, per-wave 64x64) and leave as whole 16-byte
// row segments — bias/scale/accumulate/GELU/GELU'/residual are applied on 8-column chunks
// with 16-byte loads and stores (a 2-byte-per-lane store tail is store-issue-bound).
// Workgroup ids are remapped so that tiles sharing one A row-panel run on one XCD (L2).
#include "tmi_common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int FT_BYTES = 16384;  // one operand tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA issued from inline asm on purpose: hipcc would otherwise (a) treat the DMA as a
// pending LDS write that may alias every ds_read and wait vmcnt(0) before the first fragment
// read of the tile being computed, serialising the prefetch with the MFMAs.  The asm form is
// invisible to its wait-count pass; completion is waited for by hand (vmcnt(0) + barrier) before
// any wave reads the staged tile.  M0 (the DMA's LDS base) is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_wave_base);
  const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst_u)
               : "memory");
}

// Diagnostic switches of the kernels below (in-kernel stamps, "skip the epilogue / the stores" ablations, TMI_GEMM_DBG): the
// shipped library compiles them out - TMI_DBG(P) is the constant 0 and every `if (TMI_DBG(P) & ...)` folds away; a build with
// make EXPERIMENTS=1 (-DTMI_GEMM_EXPERIMENTS) reads the field (tools/p8_stamps.py, tools/gemm_epi_probe.py need that build).
#ifdef TMI_GEMM_EXPERIMENTS
#define TMI_DBG(P) ((P).dbg)
#else
#define TMI_DBG(P) 0
#endif

struct FastParams {
  tmi_gemm_desc d;
  int tiles_m, tiles_n, ktiles;
  int xm, xn;        // XCD partition of the tile grid: xm * xn == 8
  int ptm, ptn;      // tiles per XCD partition along m / n
  int walk_m;        // consecutive workgroups of an XCD walk along m (sharing a B panel) instead of along n (sharing an A panel)
  int wide;          // epilogue may use 16-byte accesses on C / aux / resid
  int64_t a_cols_rd; // KS operands: readable column count (multiple of 8)
  int64_t b_cols_rd;
  int dbg;           // TMI_GEMM_DBG (diagnostics only; read through TMI_DBG: a constant 0 unless built with -DTMI_GEMM_EXPERIMENTS)
  uint32_t drop_thr, drop_key;  // epilogue dropout (desc.dropout_p): threshold (0 = off) and stream key
  float drop_scale;
  int64_t split_c_stride;  // != 0: split s stores (no atomics) to C + s * split_c_stride (workspace slabs)
  int slots;               // persistent eight-phase kernel: tile slots (8 * ptm * ptn, padding included) dealt over gridDim.x workgroups
  int epi;                 // epilogue class of the lean interior path (EPI_*, epi_class()); 0 = the generic code only
  int xsplit;              // eight-phase kernel, split-K through slabs: the splits are dealt over XCD GROUPS (see launch_p8); 0 = blockIdx.y
};

// What an interior 32x64 piece has to do besides bias (+ q scale), decided once per launch on the host.  The generic
// epilogue tests every feature of tmi_gemm_desc per pass with all of its pointers live (it measured ~800 instructions
// and ~4.8 k cycles per piece with two waves per SIMD, whatever the features in use); a class is straight-line code.
constexpr int EPI_GENERIC = 0, EPI_SIMPLE = 1 /* bias, scale, GELU + saved pre-activation */, EPI_AUXIN = 2 /* x GELU'(aux_in) */,
              EPI_RESID = 3 /* bias, [saved pre-activation, GELU,] dropout, + residual */, EPI_ACC = 4 /* C += */;
inline int epi_class(const tmi_gemm_desc& d, bool wide) {
  static const int off = [] { const char* e = getenv("TMI_GEMM_LEAN_EPI"); return e && atoi(e) == 0; }();
  if (off || !wide || (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15 || d.bias_sb % 4))) return EPI_GENERIC;
  const bool drop = d.dropout_p > 0.f;
  if (d.accumulate) return (!d.aux_in && !d.resid && !d.act && !d.aux_out && !drop && d.scale_cols <= 0) ? EPI_ACC : EPI_GENERIC;
  if (d.aux_in) return (!d.resid && !d.act && !d.aux_out && !drop && d.scale_cols <= 0) ? EPI_AUXIN : EPI_GENERIC;
  if (d.resid) return d.scale_cols <= 0 ? EPI_RESID : EPI_GENERIC;  // (+ GELU and the saved pre-activation: the conv stem, W:311-342)
  return drop ? EPI_GENERIC : EPI_SIMPLE;
}

// bijective XCD-aware remap: consecutive new ids share an XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- KC staging: ROWS rows x 64 k (128 B per row); wave-instruction j covers rows 8j..8j+7
// EM(src, i, j): what to do with wave-instruction j (the i-th of this wave) whose lane reads 16 B
// at src: either issue the LDS-DMA to lds + j*1024 or load it into a register for a later
// ds_write_b128 to lds + j*1024 + lane*16 (the same lane-linear image).
template <int ROWS, int NW, typename EM>
__device__ __forceinline__ void stage_kc(const bf16_t* base, int64_t s_row, int64_t row0, int64_t nrows,
                                         int64_t k0, int wave, int lane, EM em) {
  constexpr int PER = ROWS / 8 / NW;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = PER * wave + i;
    const int r = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;
    em(base + grow * s_row + k0 + c * 8, i, j);
  }
}

// ---- KS staging: COLS/128 sub-images of [64 k-rows][128 cols] (256 B per row, 16 KiB each);
// wave-instruction j covers k-rows 4(j%16)..+3 of sub-image j/16
// (COLS == 64: one [64 k-rows][64 cols] image, 128 B per row; wave-instruction j covers k-rows
//  8j..8j+7, swizzle phys = chunk ^ (((k >> 1) & 1) << 2) — conflict-free for the transposed reads)
template <int COLS, int NW, typename EM>
__device__ __forceinline__ void stage_ks(const bf16_t* base, int64_t s_k, int64_t col0, int64_t ncols_rd,
                                         int64_t k0, int64_t kend, int wave, int lane, EM em) {
  if constexpr (COLS == 64) {
    constexpr int PER64 = 8 / NW;
#pragma unroll
    for (int i = 0; i < PER64; ++i) {
      const int j = PER64 * wave + i;
      const int kr = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ (((kr >> 1) & 1) << 2);
      int64_t gk = k0 + kr;
      gk = gk < kend ? gk : kend - 1;
      int64_t gc = col0 + c * 8;
      gc = gc + 8 <= ncols_rd ? gc : ncols_rd - 8;
      em(base + gk * s_k + gc, i, j);
    }
    return;
  }
  constexpr int PER = COLS / 8 / NW;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = PER * wave + i;
    const int sub = j >> 4, jj = j & 15;
    const int kr = 4 * jj + (lane >> 4);
    const int c = (lane & 15) ^ ((kr & 3) << 2);
    int64_t gk = k0 + kr;
    gk = gk < kend ? gk : kend - 1;
    int64_t gc = col0 + sub * 128 + c * 8;
    gc = gc + 8 <= ncols_rd ? gc : ncols_rd - 8;
    em(base + gk * s_k + gc, i, j);
  }
}

__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int row, int kk, int h) {
  const int off = ((2 * kk + h) ^ ((row >> 1) & 7)) * 16;
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + off);
}

template <bool W64>
__device__ __forceinline__ bf16x8 frag_ks(const char* tile, int col_base, int kk, int lane) {
  if constexpr (W64) {  // 128-byte-row image of a 64-column operand tile
    const int g = lane >> 4, i = lane & 15;
    const int h = g >> 1, q = i >> 2, p = i & 3;
    const int col = col_base + 16 * (g & 1) + 4 * p;
    bf16x4 part[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kr = 16 * kk + 8 * h + 4 * half + q;
      const int chunk = (col >> 3) ^ (((kr >> 1) & 1) << 2);
      const char* addr = tile + kr * 128 + chunk * 16 + (col & 7) * 2;
      part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
    }
    return __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
  }
  // 32x32x16 operand fragment for k-step kk from a [k][col] image: lane (r, h) gets
  // T[16kk + 8h + j][col_base + r], j = 0..7, as two transposed 4x16 block reads.
  const int g = lane >> 4, i = lane & 15;
  const int h = g >> 1;
  const int q = i >> 2, p = i & 3;
  const int colf = col_base + 16 * (g & 1) + 4 * p;  // first of this lane's 4 address columns
  tile += (colf >> 7) * 16384;                          // 128-column sub-image
  const int col = colf & 127;
  bf16x4 part[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int kr = 16 * kk + 8 * h + 4 * half + q;
    const int chunk = (col >> 3) ^ ((kr & 3) << 2);
    const char* addr = tile + kr * 256 + chunk * 16 + (col & 7) * 2;
    part[half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
  }
  return __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
}

// NI == 2 (the wave's output goes through wide_epilogue): the MFMA operands are passed SWAPPED, so an accumulator holds the
// block of C^T - lane = row (lane & 31), register r = column 8 (r >> 2) + 4 (lane >> 5) + (r & 3): four consecutive
// columns per register quad, which is what lets the epilogue stage it with 16-byte LDS writes.
template <bool A_KS, bool B_KS, int MI, int NI, bool A64 = false, bool B64 = false>
__device__ __forceinline__ void mma_tile(const char* As, const char* Bs, int row_base, int col_base, int lane,
                                         f32x16 (&acc)[MI][NI]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[MI], b[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (A_KS) a[i] = frag_ks<A64>(As, row_base + i * 32, kk, lane);
      else a[i] = frag_kc(As, row_base + i * 32 + r, kk, h);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if constexpr (B_KS) b[i] = frag_ks<B64>(Bs, col_base + i * 32, kk, lane);
      else b[i] = frag_kc(Bs, col_base + i * 32 + r, kk, h);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = NI == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

// ---- wide epilogue helpers
template <typename TC> struct Vec8;
template <> struct Vec8<float> {
  typedef float raw_t __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ raw_t load_raw(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  static __device__ __forceinline__ void widen(const raw_t& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = r[i];
  }
  static __device__ __forceinline__ raw_t pack(const float (&v)[8]) {
    raw_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = v[i];
    return r;
  }
  static __device__ __forceinline__ void store_raw(float* p, const raw_t& r) {
    *reinterpret_cast<f32x4*>(p) = f32x4{r[0], r[1], r[2], r[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{r[4], r[5], r[6], r[7]};
  }
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <> struct Vec8<bf16_t> {
  typedef bf16x8 raw_t;
  static __device__ __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void widen(const raw_t& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ raw_t pack(const float (&v)[8]) {
    raw_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    return r;
  }
  static __device__ __forceinline__ void store_raw(bf16_t* p, const raw_t& r) { *reinterpret_cast<bf16x8*>(p) = r; }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};

__device__ unsigned long long g_gemm_stamps[8];  // diagnostics (ABL == 8): cycles per loop phase, block 0 wave 0

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ unsigned long long g_epi_stamps[8];   // diagnostics (TMI_GEMM_DBG & 32): inside wide_epilogue, block 8 wave 0, first piece

// Staging image of one 32x64 fp32 piece: 32 rows of 256 B; the 16-byte chunk c (four columns) of row r lives at
// r * 256 + ((c ^ (r & 7)) << 4).  Conflict-free both ways: the accumulator layout writes one chunk per lane with eight
// consecutive rows per ds_write_b128 lane group (the XOR spreads them over the 32 banks), the row-major read-back
// (8 lanes x 2 chunks per row, 8 rows per instruction) hits every bank group of ds_read_b128 once.
__device__ __forceinline__ int epi_off(int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 4) | ((col & 3) << 2)); }

// The lean interior path: the piece lies wholly inside C and the launch is of class CLS.  Four passes of 8 rows; a lane
// owns 8 consecutive columns of one row per pass (16-byte accesses throughout).
template <typename TC, int CLS, int AHEAD>
__device__ __forceinline__ void lean_rows(const FastParams& P, const char* E, int64_t mw, int64_t nw, int64_t bz, int lane) {
  asm volatile("" : "+v"(lane));  // (opaque: nothing below is to be hoisted above the K loop of a persistent kernel)
  const tmi_gemm_desc& d = P.d;
  typedef typename Vec8<TC>::raw_t raw_t;
  const int chunk = lane & 7, rsub = lane >> 3;
  const int64_t n = nw + chunk * 8;
  const int64_t off0 = (mw + rsub) * d.ldc + n, step = 8 * d.ldc;
  TC* cp = reinterpret_cast<TC*>(d.C) + bz * d.c_sb + (int64_t)blockIdx.y * P.split_c_stride + off0;
  const char* Elo = E + rsub * 256 + (((2 * chunk) ^ rsub) << 4);
  const char* Ehi = E + rsub * 256 + (((2 * chunk + 1) ^ rsub) << 4);
  float bv[8];
  if (d.bias) {  // (16-byte aligned: epi_class)
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + bz * d.bias_sb + n), b1 = *reinterpret_cast<const f32x4*>(d.bias + bz * d.bias_sb + n + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { bv[i] = b0[i]; bv[4 + i] = b1[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = 0.f;
  }
  float sv[8];
  const bool scaled = CLS == EPI_SIMPLE && nw < d.scale_cols;  // (uniform: the wave's 64 columns against scale_cols)
  if (scaled) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sv[i] = (n + i < d.scale_cols) ? d.scale : 1.f;
  }
  const TC* xp = nullptr;  // the class's extra operand, walked like cp
  int64_t xstep = step;
  if constexpr (CLS == EPI_AUXIN) xp = reinterpret_cast<const TC*>(d.aux_in) + bz * d.c_sb + off0;
  if constexpr (CLS == EPI_RESID) { xp = reinterpret_cast<const TC*>(d.resid) + bz * d.r_sb + (mw + rsub) * d.r_ld + n; xstep = 8 * d.r_ld; }
  if constexpr (CLS == EPI_ACC) xp = cp;
  TC* ap = ((CLS == EPI_SIMPLE || CLS == EPI_RESID) && d.aux_out) ? reinterpret_cast<TC*>(d.aux_out) + bz * d.c_sb + off0 : nullptr;
#pragma unroll
  for (int pp = 0; pp < 4; pp += AHEAD) {
    raw_t x[AHEAD];
    f32x4 lo[AHEAD], hi[AHEAD];
    if constexpr (CLS != EPI_SIMPLE) {
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) x[j] = Vec8<TC>::load_raw(xp + (pp + j) * xstep);
    }
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
      lo[j] = *reinterpret_cast<const f32x4*>(Elo + (pp + j) * 2048);
      hi[j] = *reinterpret_cast<const f32x4*>(Ehi + (pp + j) * 2048);
    }
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
      float v[8], t[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = lo[j][i] + bv[i]; v[4 + i] = hi[j][i] + bv[4 + i]; }
      if constexpr (CLS == EPI_SIMPLE) {
        if (scaled) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] *= sv[i];
        }
        if (ap) Vec8<TC>::store(ap + (pp + j) * step, v);
        if (d.act == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = gelu_fwd_t<TC>(v[i]);
        }
      } else {
        if constexpr (CLS == EPI_RESID) {  // (the conv stem: x = GELU(u) + PE with u saved; uniform tests, off for the FFN / out-proj launches)
          if (ap) Vec8<TC>::store(ap + (pp + j) * step, v);
          if (d.act == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gelu_fwd_t<TC>(v[i]);
          }
        }
        Vec8<TC>::widen(x[j], t);
        if constexpr (CLS == EPI_AUXIN) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_t<TC>(t[i]);
        } else {
          if (CLS == EPI_RESID && P.drop_thr) tmi_drop8(v, mw + rsub + (pp + j) * 8, n, P.drop_key, P.drop_thr, P.drop_scale);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += t[i];
        }
      }
      // (plain stores: non-temporal ones make this launch 5-8 % faster alone - the C stream no longer evicts the operand
      // panels from L2 - and the STEP 0.15 ms slower: the consumer finds C neither in L2 nor in the Infinity Cache)
      Vec8<TC>::store(cp + (pp + j) * step, v);
    }
  }
}

// One 32x64 piece of a wave's output (accumulator blocks a0 | a1, C^T layout: see mma_tile) whose top-left element is
// C[mw][nw]; E is the wave's private 8 KiB staging image.  Round 4: 8 ds_write_b128 instead of 64 ds_write_b32 per piece
// (the old form cost ~4 k cycles per piece with eight waves at it: LDS stores run at 64 B/clk/CU), and the row passes
// taken AHEAD at a time, with their LDS reads and epilogue operand loads issued before the first use (AHEAD = 1 for the
// kernels that have no registers to spare: 128 accumulator registers, or a 128-register budget).
template <typename TC, int AHEAD = 2>
__device__ __forceinline__ void wide_epilogue(const FastParams& P, const f32x16& a0, const f32x16& a1, char* E,
                                              int64_t mw, int64_t nw, int64_t bz, int lane, bool atomic) {
  asm volatile("" : "+v"(lane));  // (opaque: see lean_rows)
  const tmi_gemm_desc& d = P.d;
  const bool st = (TMI_DBG(P) & 32) && blockIdx.x == 8 && threadIdx.x < 64;
  unsigned long long e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  if (st) e0 = stamp();
  {
    const int m = lane & 31, h = lane >> 5, sw = m & 7;
    char* Er = E + m * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<f32x4*>(Er + (((2 * q + h) ^ sw) << 4)) = f32x4{a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]};
      *reinterpret_cast<f32x4*>(Er + (((8 + 2 * q + h) ^ sw) << 4)) = f32x4{a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]};
    }
  }
  if (st) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); e1 = stamp(); }
  if (P.epi != EPI_GENERIC && !atomic && mw + 32 <= d.M && nw + 64 <= d.N && !(TMI_DBG(P) & 8)) {  // (uniform over the wave)
    if (P.epi == EPI_SIMPLE) lean_rows<TC, EPI_SIMPLE, AHEAD>(P, E, mw, nw, bz, lane);
    else if (P.epi == EPI_AUXIN) lean_rows<TC, EPI_AUXIN, AHEAD>(P, E, mw, nw, bz, lane);
    else if (P.epi == EPI_RESID) lean_rows<TC, EPI_RESID, AHEAD>(P, E, mw, nw, bz, lane);
    else lean_rows<TC, EPI_ACC, AHEAD>(P, E, mw, nw, bz, lane);
    if (st) {
      e3 = stamp();
      if (lane == 0 && g_epi_stamps[7] == 0) { g_epi_stamps[0] = e1 - e0; g_epi_stamps[1] = 0; g_epi_stamps[2] = e3 - e1; g_epi_stamps[7] = 1; }
    }
    return;
  }
  TC* C = reinterpret_cast<TC*>(d.C) + bz * d.c_sb + (int64_t)blockIdx.y * P.split_c_stride;
  if (atomic && P.split_c_stride == 0) {  // split-K: fp32 atomics, 256 contiguous bytes per wave-instruction
    if constexpr (sizeof(TC) == 4) {
      const int64_t n = nw + lane;
      if (n < d.N) {
        for (int row = 0; row < 32; ++row) {
          const int64_t m = mw + row;
          if (m >= d.M) break;
          atomicAdd(reinterpret_cast<float*>(C) + m * d.ldc + n, *reinterpret_cast<const float*>(E + epi_off(row, lane)));
        }
      }
    }
    return;
  }
  TC* aux_out = d.aux_out ? reinterpret_cast<TC*>(d.aux_out) + bz * d.c_sb : nullptr;
  const TC* aux_in = d.aux_in ? reinterpret_cast<const TC*>(d.aux_in) + bz * d.c_sb : nullptr;
  const TC* resid = d.resid ? reinterpret_cast<const TC*>(d.resid) + bz * d.r_sb : nullptr;
  const int chunk = lane & 7, rsub = lane >> 3;
  const int64_t n = nw + chunk * 8;
  if (n >= d.N) return;
  const bool full = P.wide && (n + 8 <= d.N);
  float bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bv[i] = (d.bias && n + i < d.N) ? d.bias[bz * d.bias_sb + n + i] : 0.f;
  // this lane's two chunks of row 8 p + rsub: (row & 7) == rsub for every pass, so the passes differ by an immediate
  const char* Elo = E + rsub * 256 + (((2 * chunk) ^ rsub) << 4);
  const char* Ehi = E + rsub * 256 + (((2 * chunk + 1) ^ rsub) << 4);
  if (st) e2 = stamp();
  if (full) {  // (edge pieces and launches outside the lean classes: one pass at a time, every feature tested)
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
      const int64_t m = mw + p * 8 + rsub;
      if (m >= d.M) continue;
      const int64_t idx = m * d.ldc + n;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Elo + p * 2048), hi = *reinterpret_cast<const f32x4*>(Ehi + p * 2048);
      float v[8], t[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = lo[i] + bv[i]; v[4 + i] = hi[i] + bv[4 + i]; }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n + i < d.scale_cols) v[i] *= d.scale;
      if (d.accumulate) {
        Vec8<TC>::load(C + idx, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
      }
      if (aux_out && !(TMI_DBG(P) & 8)) Vec8<TC>::store(aux_out + idx, v);
      if (d.act == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_fwd_t<TC>(v[i]);
      }
      if (aux_in) {
        Vec8<TC>::load(aux_in + idx, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_t<TC>(t[i]);
      }
      if (P.drop_thr) tmi_drop8(v, m, n, P.drop_key, P.drop_thr, P.drop_scale);
      if (resid) {
        Vec8<TC>::load(resid + m * d.r_ld + n, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
      }
      if (TMI_DBG(P) & 8) { asm volatile("" ::"v"(v[0]), "v"(v[3]), "v"(v[7])); continue; }  // diagnostics: everything but the store
      Vec8<TC>::store(C + idx, v);
    }
    if (st) {
      e3 = stamp();
      if (lane == 0 && g_epi_stamps[7] == 0) { g_epi_stamps[0] = e1 - e0; g_epi_stamps[1] = e2 - e1; g_epi_stamps[2] = e3 - e2; g_epi_stamps[7] = 1; }
    }
    return;
  }
  // ragged right edge / unaligned output: element by element
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const int64_t m = mw + p * 8 + rsub;
    if (m >= d.M) continue;
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Elo + p * 2048);
      const f32x4 b = *reinterpret_cast<const f32x4*>(Ehi + p * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    const int64_t idx = m * d.ldc + n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (n + i >= d.N) break;
      float x = v[i] + bv[i];
      if (n + i < d.scale_cols) x *= d.scale;
      if (d.accumulate) x += to_f32(C[idx + i]);
      if (aux_out) aux_out[idx + i] = from_f32<TC>(x);
      if (d.act == 1) x = gelu_fwd_t<TC>(x);
      if (aux_in) x *= gelu_grad_t<TC>(to_f32(aux_in[idx + i]));
      if (P.drop_thr) x = tmi_drop1(x, m, n + i, P.drop_key, P.drop_thr, P.drop_scale);
      if (resid) x += to_f32(resid[m * d.r_ld + n + i]);
      C[idx + i] = from_f32<TC>(x);
    }
  }
}

// The same epilogue for ONE 32x32 accumulator block through a 4 KiB staging image (configurations
// whose waves own 32x32 outputs): rows of 32 fp32, 16-byte chunks XOR-swizzled by (row & 7);
// 4 lanes x 8 columns per row, 16 rows per pass.
__device__ __forceinline__ int epi32_off(int row, int col) { return row * 128 + ((((col >> 2) ^ (row & 7)) << 4) | ((col & 3) << 2)); }

template <typename TC>
__device__ __forceinline__ void wide_epilogue32(const FastParams& P, const f32x16& a0, char* E, int64_t mw, int64_t nw,
                                                int64_t bz, int lane) {
  const tmi_gemm_desc& d = P.d;
  {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      *reinterpret_cast<float*>(E + epi32_off(row, c)) = a0[reg];
    }
  }
  TC* C = reinterpret_cast<TC*>(d.C) + bz * d.c_sb + (int64_t)blockIdx.y * P.split_c_stride;
  TC* aux_out = d.aux_out ? reinterpret_cast<TC*>(d.aux_out) + bz * d.c_sb : nullptr;
  const TC* aux_in = d.aux_in ? reinterpret_cast<const TC*>(d.aux_in) + bz * d.c_sb : nullptr;
  const TC* resid = d.resid ? reinterpret_cast<const TC*>(d.resid) + bz * d.r_sb : nullptr;
  const int chunk = lane & 3;
  const int64_t n = nw + chunk * 8;
  if (n >= d.N) return;
  const bool full = P.wide && (n + 8 <= d.N);
  float bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bv[i] = (d.bias && n + i < d.N) ? d.bias[bz * d.bias_sb + n + i] : 0.f;
#pragma unroll 1
  for (int p = 0; p < 2; ++p) {
    const int row = p * 16 + (lane >> 2);
    const int64_t m = mw + row;
    if (m >= d.M) continue;
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(E + epi32_off(row, chunk * 8));
      const f32x4 b = *reinterpret_cast<const f32x4*>(E + epi32_off(row, chunk * 8 + 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    const int64_t idx = m * d.ldc + n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] += bv[i];
      if (n + i < d.scale_cols) v[i] *= d.scale;
    }
    if (full) {
      float t[8];
      if (d.accumulate) {
        Vec8<TC>::load(C + idx, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
      }
      if (aux_out) Vec8<TC>::store(aux_out + idx, v);
      if (d.act == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_fwd_t<TC>(v[i]);
      }
      if (aux_in) {
        Vec8<TC>::load(aux_in + idx, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_t<TC>(t[i]);
      }
      if (P.drop_thr) tmi_drop8(v, m, n, P.drop_key, P.drop_thr, P.drop_scale);
      if (resid) {
        Vec8<TC>::load(resid + m * d.r_ld + n, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
      }
      Vec8<TC>::store(C + idx, v);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (n + i >= d.N) break;
        float x = v[i];
        if (d.accumulate) x += to_f32(C[idx + i]);
        if (aux_out) aux_out[idx + i] = from_f32<TC>(x);
        if (d.act == 1) x = gelu_fwd_t<TC>(x);
        if (aux_in) x *= gelu_grad_t<TC>(to_f32(aux_in[idx + i]));
        if (P.drop_thr) x = tmi_drop1(x, m, n + i, P.drop_key, P.drop_thr, P.drop_scale);
        if (resid) x += to_f32(resid[m * d.r_ld + n + i]);
        C[idx + i] = from_f32<TC>(x);
      }
    }
  }
}


template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// CFG 0: 128x128 tile, 4 waves (2x2) of 64x64, 2 stages (64 KiB, two workgroups per CU)
// CFG 1: 256x256 tile, 8 waves (2x4) of 128x64, 2 stages (128 KiB, one workgroup per CU): twice
//        the MFMA work per staged byte and per LDS-DMA instruction a wave has to issue
// SPEC (wave specialisation): the workgroup carries as many LOADER waves as consumer waves; the
//        loaders only issue the LDS-DMA of the next K-tile, the consumers only read fragments
//        and issue MFMAs, so no wave serialises ~100-cycle DMA issues with its MFMA stream
// CFG 2: CFG 0's tile with 4 loader + 4 consumer waves
// CFG 3: 256x128 tile, 4 consumer waves (2x2) of 128x64 + 4 loader waves (one of each per SIMD)
template <int CFG> struct Cfg;
template <> struct Cfg<0> { static constexpr int BM = 128, BN = 128, WM = 64, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
template <> struct Cfg<1> { static constexpr int BM = 256, BN = 256, WM = 128, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
template <> struct Cfg<2> { static constexpr int BM = 128, BN = 128, WM = 64, WN = 64, NSTAGE = 2; static constexpr bool SPEC = true; };
template <> struct Cfg<3> { static constexpr int BM = 256, BN = 128, WM = 128, WN = 64, NSTAGE = 2; static constexpr bool SPEC = true; };
// CFG 4 / 5: the same tiles with twice the waves (32x64 resp. 64x64 per wave): half the LDS-DMA
// instructions per wave per K-tile, so their ~100-cycle issue cost stops dominating each wave's loop
template <> struct Cfg<4> { static constexpr int BM = 128, BN = 128, WM = 32, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
template <> struct Cfg<5> { static constexpr int BM = 256, BN = 256, WM = 64, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
// CFG 6: 64x64 tile, 2 waves of 32x64 (32 KiB of LDS: several workgroups per CU).  For problems
// with few 128x128 tiles (decoder / Wav2Vec2 encoder, M = B*100): a CU stages operands at
// ~45 GB/s whatever the tile, so the time of such a GEMM is (BM + BN) * K * 2 B per workgroup
// over that rate — smaller tiles on more CUs finish sooner.
template <> struct Cfg<6> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
// CFG 7 / 8 / 9: tiles of CFG 4 / 5 / 6 with REGISTER staging (global_load_dwordx4 -> VGPR ->
// ds_write_b128 into the same image): an LDS-DMA wave-instruction occupies the CU's memory
// pipeline for ~55 cycles per KiB (measured with in-kernel stamps), which bounds a 128x128 tile
// at ~0.8 PF/s; plain 16-byte loads issue several times faster
template <> struct Cfg<7> { static constexpr int BM = 128, BN = 128, WM = 32, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
template <> struct Cfg<8> { static constexpr int BM = 256, BN = 256, WM = 64, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
template <> struct Cfg<9> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 64, NSTAGE = 2; static constexpr bool SPEC = false; };
// CFG 10: CFG 6 with a 4-stage ring and counted vmcnt.  Only when the grid is at most one workgroup
// per CU (64 KiB of LDS each: co-resident 2-stage workgroups hide latency better when there are
// enough of them — M800 N3072 K3072: 40 us on CFG 6, 61 us on CFG 10) and K is long (M800 N768 K3072
// with a k-strided B: 36 -> 30 us).
template <> struct Cfg<10> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 64, NSTAGE = 4; static constexpr bool SPEC = false; };
// CFG 11: the 64x64 tile on FOUR waves of 32x32 (all four SIMDs of the CU, half the fragment-read + MFMA
// chain per K-tile per wave): for grids of at most ~one workgroup per CU, where the K loop of CFG 6 is a
// latency chain on two SIMDs.  Never split (its epilogue has no atomic form).
template <> struct Cfg<11> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 32, NSTAGE = 2; static constexpr bool SPEC = false; };
// CFG 12: CFG 11 with the 4-stage ring of CFG 10
template <> struct Cfg<12> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 32, NSTAGE = 4; static constexpr bool SPEC = false; };
// CFG 13: CFG 12's tile with the K loop dealt over KG = 2 groups of four waves INSIDE the workgroup (group g multiplies
// K-tiles g, g + KG, ...; each group has its own 4-stage LDS-DMA ring, 128 KiB in all; the partial accumulators meet in
// LDS before the epilogue).  On grids of at most one workgroup per CU the K loop of CFG 12 is a dependent chain per K-tile
// and wave (barrier -> DMA issue -> fragment reads -> four dependent MFMAs: ~0.4 us per 64-deep K-tile whatever feeds LDS:
// an 8-stage DMA ring and a ring of registers filled by plain global loads with counted waits both measured level with the
// 4-stage DMA ring); a second wave per SIMD on OTHER K-tiles overlaps those chains (0.24 us per K-tile; a third group adds
// nothing: that is the CU's fill rate), and with the weights arriving from HBM, as they do in the step, the two rings'
// six K-tiles in flight matter more still (M 800, N 768, K 768, cold weights: 13.5 us on CFG 11, 9.4 on CFG 12, 8.9 here).
// K % 64 == 0 only (no tail zeroing).
template <> struct Cfg<13> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 32, NSTAGE = 4; static constexpr bool SPEC = false; };
// CFG 14: CFG 13's two K-groups on 2-stage rings (64 KiB: TWO workgroups per CU) - for 64x64 grids of more than one workgroup
// per CU, where CFG 13's 128 KiB would run a second, nearly empty round (Whisper-large's decoder: [800, 1280] = 260 tiles).
template <> struct Cfg<14> { static constexpr int BM = 64, BN = 64, WM = 32, WN = 32, NSTAGE = 2; static constexpr bool SPEC = false; };
template <int CFG> constexpr int kKG = (CFG == 13 || CFG == 14) ? 2 : 1;
template <int CFG> constexpr bool kRegStage = CFG >= 7 && CFG <= 9;

// Split-K through the workspace: every split runs the ordinary (non-atomic) epilogue into its own
// fp32 slab [nbatch][M][N] of the caller's workspace, and splitk_reduce_kernel, launched right
// behind on the same stream, sums the slabs into C.  Both passes are plain 16-byte streams at HBM
// rate; fp32 atomics reach ~0.5 TB/s on this part, and an in-kernel "last split gathers" scheme
// needs agent-scope coherence (whole-L2 write-backs) or sc1 accesses in a latency chain and pushed
// the 256x256 kernels into register spills.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab_elems,
                                                            float* __restrict__ C, int64_t M, int64_t N, int64_t ldc,
                                                            int64_t c_sb, int64_t nbatch, int accumulate, int vec_ok) {
  const int64_t per_batch = M * N;
  if (vec_ok) {  // N % 4 == 0, ldc % 4 == 0, 16-byte aligned bases
    const int64_t nv = nbatch * per_batch / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
      const int64_t e = i * 4, b = e / per_batch, r = e - b * per_batch, m = r / N, n = r - m * N;
      f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + e);
      for (int sp = 1; sp < nsplit; ++sp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + sp * slab_elems + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
      float* dst = C + b * c_sb + m * ldc + n;
      if (accumulate) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += o[j];
      }
      *reinterpret_cast<f32x4*>(dst) = acc;
    }
  } else {
    const int64_t ne = nbatch * per_batch;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < ne; e += (int64_t)gridDim.x * 256) {
      const int64_t b = e / per_batch, r = e - b * per_batch, m = r / N, n = r - m * N;
      float acc = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) acc += slabs[sp * slab_elems + e];
      float* dst = C + b * c_sb + m * ldc + n;
      *dst = accumulate ? *dst + acc : acc;
    }
  }
}

// ABL (diagnostics, compile-time so the production loop is untouched): 2 = no MFMA/fragment reads,
// 4 = no staging after the prologue
template <typename TC, bool A_KS, bool B_KS, int CFG, int ABL = 0>
__global__ __launch_bounds__((Cfg<CFG>::BM / Cfg<CFG>::WM) * (Cfg<CFG>::BN / Cfg<CFG>::WN) * (Cfg<CFG>::SPEC ? 128 : 64) * kKG<CFG>)
void gemm_fast_kernel(const FastParams P) {
  using K = Cfg<CFG>;
  constexpr int BM = K::BM, BN = K::BN, NSTAGE = K::NSTAGE;
  constexpr int WCOLS = BN / K::WN;           // waves along N
  constexpr int NW = (BM / K::WM) * WCOLS;
  constexpr int MI = K::WM / 32, NI = K::WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const tmi_gemm_desc& d = P.d;
  // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs, so
  // blockIdx.x % 8 labels the XCD.  The tile grid is cut into xm x xn rectangles, one per XCD,
  // sized so that an XCD's slice of B stays resident in its 4 MiB L2 while A panels stream.
  // (P.xsplit: split-K dealt over XCD groups instead of blockIdx.y - see launch_p8)
  const int xsh = P.xsplit == 2 ? 2 : P.xsplit == 4 ? 1 : 0;
  const int xsp = P.xsplit ? (int)(blockIdx.x & 7) >> xsh : 0;
  const int xcd = P.xsplit ? (int)(blockIdx.x & 7) & ((1 << xsh) - 1) : (int)(blockIdx.x & 7), lidx = blockIdx.x >> 3;
  int ltm, ltn;
  if (P.walk_m) { ltn = lidx / P.ptm; ltm = lidx - ltn * P.ptm; }
  else { ltm = lidx / P.ptn; ltn = lidx - ltm * P.ptn; }
  const int tm = (xcd / P.xn) * P.ptm + ltm, tn = (xcd % P.xn) * P.ptn + ltn;
  if (tm >= P.tiles_m || tn >= P.tiles_n) return;  // padding block of a ragged partition (whole workgroup)
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const int lane = threadIdx.x & 63;
  constexpr int KG = kKG<CFG>;                // K-groups of NW waves (CFG 13 / 14)
  static_assert(KG == 1 || !K::SPEC, "K-groups and loader waves do not combine");
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = K::SPEC && wave_all >= NW;       // wave-uniform role
  const bool consumer = !K::SPEC || wave_all < NW || KG > 1;
  const int kgroup = KG > 1 ? wave_all / NW : 0;
  const int wave = KG > 1 ? wave_all - kgroup * NW : (loader ? wave_all - NW : wave_all);  // index within its role / group
  const int wr = wave / WCOLS, wc = wave % WCOLS;
  constexpr int NTHREADS = NW * (K::SPEC ? 128 : 64);

  const int total_it = (int)d.kbatch * P.ktiles;
  const int nsplit = P.xsplit ? P.xsplit : (int)gridDim.y;
  const int per = (total_it + nsplit - 1) / nsplit;
  const int it0 = (P.xsplit ? xsp : (int)blockIdx.y) * per;
  const int nt = min(total_it, it0 + per) - it0;

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(d.A) + bz * d.a_sb;
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(d.B) + bz * d.b_sb;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;  // wave-instructions per wave per tile
  u32x4 ra[PA], rb[PB];                               // register staging only
  auto walk = [&](int it, auto emA, auto emB) {
    const int kb = it / P.ktiles, kt = it - kb * P.ktiles;
    if constexpr (A_KS) stage_ks<BM, NW>(Abase + kb * d.a_skb, d.a_sk, m0, P.a_cols_rd, (int64_t)kt * 64, d.K, wave, lane, emA);
    else stage_kc<BM, NW>(Abase + kb * d.a_skb, d.a_sm, m0, d.M, (int64_t)kt * 64, wave, lane, emA);
    if constexpr (B_KS) stage_ks<BN, NW>(Bbase + kb * d.b_skb, d.b_sk, n0, P.b_cols_rd, (int64_t)kt * 64, d.K, wave, lane, emB);
    else stage_kc<BN, NW>(Bbase + kb * d.b_skb, d.b_sn, n0, d.N, (int64_t)kt * 64, wave, lane, emB);
  };
  auto stage = [&](int it, int buf) {  // LDS-DMA
    char* As = smem + buf * STAGE;
    char* Bs = As + A_BYTES;
    walk(it, [&](const bf16_t* src, int, int j) { glds16(src, As + j * 1024); },
         [&](const bf16_t* src, int, int j) { glds16(src, Bs + j * 1024); });
  };
  auto fetch = [&](int it) {           // register staging, part 1: loads in flight
    walk(it, [&](const bf16_t* src, int i, int) { ra[i] = *reinterpret_cast<const u32x4*>(src); },
         [&](const bf16_t* src, int i, int) { rb[i] = *reinterpret_cast<const u32x4*>(src); });
  };
  auto commit = [&](int buf) {         // part 2: the same lane-linear image the DMA would write
    char* As = smem + buf * STAGE;
    char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(As + (PA * wave + i) * 1024 + lane * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(Bs + (PB * wave + i) * 1024 + lane * 16) = rb[i];
  };
  // K tail (K % 64 != 0, host allows it only in (KS, KS)): rows k >= K of a KS image were loaded
  // from a clamped row; zero them so they contribute nothing
  auto zero_tail = [&](int it, int buf) {
    if constexpr (A_KS && B_KS) {
      const int kt = it % P.ktiles;
      const int kvalid = (int)min((int64_t)64, d.K - (int64_t)kt * 64);
      if (kvalid < 64) {
        char* As = smem + buf * STAGE;
        if constexpr (BM == 64 && BN == 64) {  // two [64][128 B] images
          for (int idx = threadIdx.x; idx < 2 * (64 - kvalid) * 8; idx += NTHREADS) {
            const int img = idx / ((64 - kvalid) * 8), rem = idx % ((64 - kvalid) * 8);
            *reinterpret_cast<u32x4*>(As + img * 8192 + kvalid * 128 + rem * 16) = u32x4{0u, 0u, 0u, 0u};
          }
        } else {
          constexpr int NIMG = (BM + BN) / 128;  // consecutive 16 KiB [64][256 B] images: A sub-images, then B's
          for (int idx = threadIdx.x; idx < NIMG * (64 - kvalid) * 16; idx += NTHREADS) {
            const int img = idx / ((64 - kvalid) * 16), rem = idx % ((64 - kvalid) * 16);
            *reinterpret_cast<u32x4*>(As + img * 16384 + kvalid * 256 + rem * 16) = u32x4{0u, 0u, 0u, 0u};
          }
        }
        __syncthreads();
      }
    }
  };

  const bool stager = !K::SPEC || loader;
  if constexpr (ABL == 8) {  // stamped copy of the loop: where do an iteration's cycles go?
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
    if (nt > 0) {
      stage(it0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int buf = 0;
      for (int t = 0; t < nt; ++t) {
        const unsigned long long t0 = stamp();
        if (t + 1 < nt) stage(it0 + t + 1, buf ^ 1);
        const unsigned long long t1 = stamp();
        const char* As = smem + buf * STAGE;
        mma_tile<A_KS, B_KS, MI, NI, BM == 64, BN == 64>(As, As + A_BYTES, wr * K::WM, wc * K::WN, lane, acc);
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
        const unsigned long long t2 = stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = stamp();
        __syncthreads();
        const unsigned long long t4 = stamp();
        acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += 1;
        buf ^= 1;
      }
    }
    if (blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
      for (int i = 0; i < 5; ++i) g_gemm_stamps[i] = acc_t[i];
    wide_epilogue<TC>(P, acc[0][0], acc[0][1], smem + wave * 8192, m0 + wr * K::WM, n0 + wc * K::WN, bz, lane, false);
    return;
  }
  if constexpr (kRegStage<CFG>) {
    if (nt > 0) {
      fetch(it0);
      commit(0);
      __syncthreads();
      int buf = 0;
      for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) fetch(it0 + t + 1);   // global loads fly under this tile's MFMAs
        zero_tail(it0 + t, buf);
        const char* As = smem + buf * STAGE;
        mma_tile<A_KS, B_KS, MI, NI, BM == 64, BN == 64>(As, As + A_BYTES, wr * K::WM, wc * K::WN, lane, acc);
        if (t + 1 < nt) commit(buf ^ 1);      // nobody reads the other buffer during this iteration
        __syncthreads();
        buf ^= 1;
      }
    }
  } else if constexpr ((NSTAGE > 2 || kKG<CFG> > 1) && !K::SPEC && ABL == 0) {
    // Deep ring for latency-bound problems (one small workgroup per CU, nothing else to hide the
    // DMA round trip behind): tiles t+1 .. t+NSTAGE-1 are in flight while tile t is multiplied.
    // Counted wait: each wave issues LPT LDS-DMA instructions per tile, in tile order, so
    // "all but the newest (NSTAGE-2)*LPT" means tile t has landed.  Every slot of the schedule is
    // issued (past the last tile the last one is re-fetched into a slot nobody reads) to keep that
    // count exact.  One barrier per K-tile: it publishes tile t and retires the reads of tile t-1,
    // whose slot the stage issued right after it overwrites.
    // K-groups (KG > 1): group g walks tiles g, g + KG, ... in its own ring; every group makes the same number of trips
    // (and barrier calls), a group whose tile is past the end re-fetches the last tile and skips the MFMAs.
    if (nt > 0) {
      constexpr int LPT = PA + PB;
      static_assert((NSTAGE - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
      const int last = it0 + nt - 1;
      char* ring = smem + kgroup * (NSTAGE * STAGE);
      auto gstage = [&](int it, int buf) {
        char* As = ring + buf * STAGE;
        char* Bs = As + A_BYTES;
        walk(it, [&](const bf16_t* src, int, int j) { glds16(src, As + j * 1024); },
             [&](const bf16_t* src, int, int j) { glds16(src, Bs + j * 1024); });
      };
#pragma unroll
      for (int s_ = 0; s_ < NSTAGE - 1; ++s_) gstage(min(it0 + kgroup + KG * s_, last), s_);
      int slot = 0, fill = NSTAGE - 1;
      const int trips = (nt + KG - 1) / KG;
      for (int t = 0; t < trips; ++t) {
        wait_vmcnt<(NSTAGE - 2) * LPT>();
        __syncthreads();
        gstage(min(it0 + kgroup + KG * (t + NSTAGE - 1), last), fill);
        if constexpr (KG == 1) zero_tail(it0 + t, slot);
        const char* As = ring + slot * STAGE;
        if (KG == 1 || kgroup + KG * t < nt)
          mma_tile<A_KS, B_KS, MI, NI, BM == 64, BN == 64>(As, As + A_BYTES, wr * K::WM, wc * K::WN, lane, acc);
        slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        fill = fill + 1 == NSTAGE ? 0 : fill + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-fetched tail tiles must land before LDS is reused
      __syncthreads();
    }
    if constexpr (KG > 1) {
      // partial accumulators of groups 1 .. KG-1 meet group 0's in LDS: [group - 1][wave][register][lane] fp32, behind
      // the epilogue staging of group 0 (NW x 4 KiB)
      static_assert(MI == 1 && NI == 1, "K-groups are written for one 32x32 accumulator per wave");
      float* part = reinterpret_cast<float*>(smem + NW * 4096);
      if (kgroup > 0) {
        float* dst = part + ((kgroup - 1) * NW + wave) * 1024;
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[e * 64 + lane] = acc[0][0][e];
      }
      __syncthreads();
      if (kgroup > 0) return;
#pragma unroll
      for (int g = 1; g < KG; ++g) {
        const float* src = part + ((g - 1) * NW + wave) * 1024;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] += src[e * 64 + lane];
      }
    }
  } else
  if (nt > 0) {
    if (stager) stage(it0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      if (ABL != 4 && stager && t + 1 < nt) stage(it0 + t + 1, buf ^ 1);   // next tile's DMA flies under this tile's MFMAs
      zero_tail(it0 + t, buf);
      if (ABL != 2 && consumer) {
        const char* As = smem + buf * STAGE;
        mma_tile<A_KS, B_KS, MI, NI, BM == 64, BN == 64>(As, As + A_BYTES, wr * K::WM, wc * K::WN, lane, acc);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf ^= 1;
    }
  }
  if (!consumer) return;
  if (TMI_DBG(P) & 1) {
    if (acc[0][0][0] + acc[MI - 1][NI - 1][5] == 123.456f) reinterpret_cast<float*>(d.C)[0] = 0.f;  // keep acc live
    return;
  }
  if constexpr (NI == 1) {  // 32x32 per wave: the 32-column epilogue, 4 KiB of staging per wave
    char* E32 = smem + wave * 4096;
#pragma unroll
    for (int p = 0; p < MI; ++p)
      wide_epilogue32<TC>(P, acc[p][0], E32, m0 + wr * K::WM + p * 32, n0 + wc * K::WN, bz, lane);
    return;
  } else {
  static_assert(NI == 2, "epilogue pieces are 64 columns wide");
  const bool atomic = nsplit > 1;
  char* E = smem + wave * 8192;
  FastParams Q = P;
  if (P.xsplit && P.split_c_stride) {  // (the split's slab; the epilogue's own blockIdx.y term is zero in this mapping)
    Q.d.C = reinterpret_cast<TC*>(d.C) + (int64_t)xsp * P.split_c_stride;
    Q.split_c_stride = 0;
  }
#pragma unroll
  for (int p = 0; p < MI; ++p)
    wide_epilogue<TC, (MI * NI * 16 <= 32 ? 2 : 1)>(Q, acc[p][0], acc[p][1], E, m0 + wr * K::WM + p * 32, n0 + wc * K::WN, bz, lane,
                                                    atomic && P.split_c_stride == 0);
  }
}

// =====================================================================================
// 256x256 "eight-phase" kernel for (KC, KC) operands (dense dgrad: dX = dY . W^T).
//
// One workgroup per CU, 8 waves as 2 (wr) x 4 (wc), 128x64 outputs per wave = four 64x32
// quadrants Q(a, b).  A K-tile (64) is four phases, one quadrant each (Q00, Q01, Q11, Q10: the A
// fragments are reloaded once, B sub-block 0 is kept), and the loop body is two K-tiles = 8 phases
// on an even/odd pair of LDS buffers.  LDS holds, per buffer, four 16 KiB half-tiles grouped by what
// one phase reads: A^a = the a-th 64 rows of both wave rows, B^b = the b-th 32 columns of all four
// wave columns.  Every phase stages ONE half-tile (two LDS-DMA per lane) two or more phases after
// its last reader and three or more phases before its first, so the waits are counted
// (vmcnt(6) / vmcnt(8), never 0 in the loop) and the DMA stays in flight across the barriers:
//
//   phase  stages                 reads (ds_read_b128)        waits      MFMAs
//   1      O.B^1 <- tile 2i+1     E.A^0 (8), E.B^0 (4)                   Q00
//   2      O.A^1 <- 2i+1          E.B^1 (4)                   vmcnt(8)   Q01
//   3      E.B^0 <- 2i+2          E.A^1 (8)                              Q11
//   4      E.A^0 <- 2i+2          -                           vmcnt(6)   Q10
//   5      E.B^1 <- 2i+2          O.A^0 (8), O.B^0 (4)                   Q00
//   6      E.A^1 <- 2i+2          O.B^1 (4)                   vmcnt(8)   Q01
//   7      O.B^0 <- 2i+3          O.A^1 (8)                              Q11
//   8      O.A^0 <- 2i+3          -                           vmcnt(6)   Q10
//
// A phase is  [reads, stage, wait] barrier [MFMAs at raised priority] barrier.  The wr = 1 waves
// run one barrier behind the wr = 0 waves (they take one extra barrier on entry, the others one on
// exit), so one group's MFMA section always coincides with the other's read/stage section.  A wait
// in phase p covers reads from phase p + 1 on; restaging comes >= 2 phases after the last read: both
// margins absorb the one-barrier stagger.  Stages past the last K-tile re-read the last tile (keeps
// the counted waits exact); their data is never read.
constexpr int P8_HALF = 16384, P8_BUF = 4 * P8_HALF;

// LDS-DMA with a uniform 64-bit base and a per-lane 32-bit byte offset (the lane offsets are
// loop-invariant: 8 VGPRs address all four half-tile kinds)
__device__ __forceinline__ void glds16_so(const char* sbase, unsigned voff, char* lds_wave_base) {
  const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_wave_base);
  const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(dst_u)
               : "memory");
}

__device__ __forceinline__ void p8_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// BM = 256, or 192: the same loop on a 192 x 256 tile - wave rows of 96 = A^0 (64 rows) + A^1 (32 rows, one MFMA row block:
// the Q11 / Q10 phases issue 4 MFMAs instead of 8, A^1 is an 8 KiB half-tile staged with ONE LDS-DMA per lane, so the
// counted waits are vmcnt(7) / vmcnt(5)).  For outputs whose 256-row tiling leaves CUs idle or a round mostly empty
// (M = 12000, N = 768: 141 tiles on 256 CUs -> 189 tiles of 3/4 the work).  k-contiguous A only.
template <typename TC, bool A_KS, bool B_KS, int BM = 256, bool PERSIST = false, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(const FastParams P) {
  static_assert(BM == 256 || (BM == 192 && !A_KS), "192-row tiles: k-contiguous A only");
  constexpr int WR = BM / 2;          // rows per wave row
  constexpr int NMI = WR / 32;        // 32-row accumulator blocks per wave: 4 or 3
  constexpr int MI1 = NMI - 2;        // row blocks of the A^1 half: 2 or 1
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A^0 A^1 B^0 B^1][16 KiB] (+ 32 KiB of epilogue staging, PERSIST)
  const tmi_gemm_desc& d = P.d;
  // tile slot -> tile coordinates (XCD-aware, see gemm_fast_kernel); false for the padding slots of a ragged partition
  // this workgroup's split when the splits ride on the XCD index (P.xsplit; never with PERSIST): XCDs [s * 8 / S, (s + 1) * 8 / S)
  // take split s, so their L2s see ONE K range and all of its tiles' panels
  // (shifts, not divisions: an integer division by a runtime value goes through the vector unit and is no longer provably uniform)
  // S = 16 / 32 (deep reductions onto few tiles - the LM head's dgrad, K = 51904 onto 800 x 768): 2 / 4 K ranges per XCD, taken
  // by alternate workgroups of the XCD, every range with ALL the tiles - each operand byte enters exactly one L2, once
  const int xsh = P.xsplit == 2 ? 2 : P.xsplit == 4 ? 1 : 0;  // log2 of the XCDs per split
  const int xss = P.xsplit == 16 ? 1 : P.xsplit == 32 ? 2 : 0;  // log2 of the splits per XCD
  const int xsp = P.xsplit ? (((int)(blockIdx.x & 7) >> xsh) << xss) | ((int)(blockIdx.x >> 3) & ((1 << xss) - 1)) : 0;
  auto decode = [&](int slot, int64_t& m0_, int64_t& n0_) -> bool {
    int xcd = slot & 7;
    const int lidx = (slot >> 3) >> xss;
    if (P.xsplit) xcd &= (1 << xsh) - 1;
    int ltm, ltn;
    if (P.walk_m) { ltn = lidx / P.ptm; ltm = lidx - ltn * P.ptm; }
    else { ltm = lidx / P.ptn; ltn = lidx - ltm * P.ptn; }
    const int tm = (xcd / P.xn) * P.ptm + ltm, tn = (xcd % P.xn) * P.ptn + ltn;
    m0_ = (int64_t)tm * BM;
    n0_ = (int64_t)tn * 256;
    return tm < P.tiles_m && tn < P.tiles_n;
  };
  // PERSIST: workgroup b walks the slots b, b + gridDim.x, ... (gridDim.x % 8 == 0: it stays on its XCD and the workgroups
  // of an XCD that are resident together work on consecutive slots of its rectangle, i.e. share operand panels in L2)
  auto next_slot = [&](int slot, int64_t& m0_, int64_t& n0_) -> int {
    while (slot < P.slots && !decode(slot, m0_, n0_)) slot += (int)gridDim.x;
    return slot;
  };
  int64_t m0, n0;
  int slot = blockIdx.x;
  if constexpr (PERSIST) {
    slot = next_slot(slot, m0, n0);
    if (slot >= P.slots) return;
  } else {
    if (!decode(slot, m0, n0)) return;
  }
  const int64_t bz = blockIdx.z;
  const int lane = threadIdx.x & 63, lane_id = lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // this split's K-tiles [kt0, kt0 + nt)
  const int nsplit = P.xsplit ? P.xsplit : (int)gridDim.y;
  const int per = (P.ktiles + nsplit - 1) / nsplit;
  const int kt0 = (P.xsplit ? xsp : (int)blockIdx.y) * per;
  const int nt = min(P.ktiles, kt0 + per) - kt0;  // may be <= 0 for a trailing split
  // K tail (both operands k-strided, K % 8 == 0; host-checked): rows k >= klast of the last K-tile are
  // fetched from row klast - 1 (in bounds) and the A fragments covering them are zeroed
  const int klast = (int)(d.K - (int64_t)(P.ktiles - 1) * 64);
  const bool tail = klast < 64;
  const char* Abase = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.A) + bz * d.a_sb);
  const char* Bbase = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.B) + bz * d.b_sb);

  // ---- loop-invariant lane offsets of the staging loads: half-tile kind x wave-instruction.
  // k-contiguous operand: half-tile = [128 rows][64 k] (128 B rows, chunk ^= (row >> 1) & 7); A^a holds
  //   tile rows 128*w + 64*a + (0..63) at image rows 64*w + .., B^b tile columns 64*w + 32*b + (0..31)
  //   at image rows 32*w + ...
  // k-strided operand: half-tile = [64 k][128 cols] (256 B rows, stage_ks's swizzle), the same
  //   row/column groups laid along the image columns.
  unsigned offA[2][2], offB[2][2];
  int krs[2];  // k-row of this lane's two loads (k-strided images)
  auto set_offsets = [&](int64_t m0, int64_t n0) {
  // (an opaque copy of the lane id: the persistent kernel calls this inside its loop, and without it the compiler keeps every
  // lane-derived term below alive across the whole K loop - ~40 registers the loop needs for the deferred stores)
  int lane = lane_id;
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ir = 8 * (2 * wave + i) + (lane >> 3);  // image row (k-contiguous images)
    const int c = (lane & 7) ^ ((ir >> 1) & 7);
    const int kr = 4 * (2 * wave + i) + (lane >> 4);  // image row (k-strided images)
    const int cl = (lane & 15) ^ ((kr & 3) << 2);     // logical 8-column chunk this lane fetches
    krs[i] = kr;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if constexpr (!A_KS) {
        int64_t gr = m0 + (ir >> 6) * WR + 64 * a + (ir & 63);
        if (BM == 192 && a == 1) {  // A^1 is [2 wave rows][32 rows]: one load per lane, image row 8 * wave + lane / 8
          const int ir1 = 8 * wave + (lane >> 3);
          gr = m0 + (ir1 >> 5) * WR + 64 + (ir1 & 31);
          gr = gr < d.M ? gr : d.M - 1;
          offA[a][i] = (unsigned)((gr * d.a_sm + ((lane & 7) ^ ((ir1 >> 1) & 7)) * 8) * 2);
        } else {
          gr = gr < d.M ? gr : d.M - 1;
          offA[a][i] = (unsigned)((gr * d.a_sm + c * 8) * 2);
        }
      } else {
        int64_t gm = m0 + ((8 * cl) >> 6) * 128 + 64 * a + ((8 * cl) & 63);
        gm = gm + 8 <= P.a_cols_rd ? gm : P.a_cols_rd - 8;
        offA[a][i] = (unsigned)((kr * d.a_sk + gm) * 2);
      }
      if constexpr (!B_KS) {
        int64_t gc = n0 + (ir >> 5) * 64 + 32 * a + (ir & 31);
        gc = gc < d.N ? gc : d.N - 1;
        offB[a][i] = (unsigned)((gc * d.b_sn + c * 8) * 2);
      } else {
        int64_t gc = n0 + ((8 * cl) >> 5) * 64 + 32 * a + ((8 * cl) & 31);
        gc = gc + 8 <= P.b_cols_rd ? gc : P.b_cols_rd - 8;
        offB[a][i] = (unsigned)((kr * d.b_sk + gc) * 2);
      }
    }
  }
  };
  set_offsets(m0, n0);
  // PERSIST: the staging of a tile's last two K-tile steps already fetches the NEXT tile's first K-tiles (the ring never
  // drains between tiles); ``stage_next`` says the offsets above belong to the next tile
  bool has_next = false, stage_next = false;
  bool in_loop = false;  // (ABL == 4, diagnostics: the loop's stages are skipped, the prologue's are not)
  auto stage = [&](int buf, int kind, int t) {  // kind: 0 A^0, 1 A^1, 2 B^0, 3 B^1; t: tile within the split
    if (ABL == 4 && in_loop) return;
    if (PERSIST && stage_next) t -= nt;   // (the offsets were switched to the next tile: its K-tile 0 or 1; nt >= 2)
    else t = t < nt ? t : nt - 1;
    const int kt = kt0 + t;
    const char* src = kind < 2 ? Abase + (int64_t)kt * (A_KS ? 128 * d.a_sk : 128) : Bbase + (int64_t)kt * (B_KS ? 128 * d.b_sk : 128);
    char* dst = smem + buf * P8_BUF + kind * P8_HALF + (2 * wave) * 1024;
    unsigned o0 = kind == 0 ? offA[0][0] : kind == 1 ? offA[1][0] : kind == 2 ? offB[0][0] : offB[1][0];
    unsigned o1 = kind == 0 ? offA[0][1] : kind == 1 ? offA[1][1] : kind == 2 ? offB[0][1] : offB[1][1];
    if constexpr (A_KS && B_KS) {
      if (tail && kt == P.ktiles - 1) {  // keep the fetch inside the operand: clamp the k-row
        const int64_t sk2 = 2 * (kind < 2 ? d.a_sk : d.b_sk);
        if (krs[0] >= klast) o0 -= (unsigned)((krs[0] - (klast - 1)) * sk2);
        if (krs[1] >= klast) o1 -= (unsigned)((krs[1] - (klast - 1)) * sk2);
      }
    }
    if (BM == 192 && kind == 1) {
      glds16_so(src, o0, smem + buf * P8_BUF + kind * P8_HALF + wave * 1024);
      return;
    }
    glds16_so(src, o0, dst);
    glds16_so(src, o1, dst + 1024);
  };

  // ---- fragment addresses (k-contiguous images): the lane part is shared by A and B (same swizzle)
  const int r = lane & 31, h = lane >> 5;
  int xo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xo[kk] = r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4);
  const int arow = wr * 64 * 128, arow1 = wr * (MI1 * 32) * 128, brow = wc * 32 * 128;

  f32x16 acc[NMI][2];
#pragma unroll
  for (int i = 0; i < NMI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 ar[2][4], b0[4], b1[4];

  // LASTK (compile-time): this K-tile may be the ragged last one of a (KS, KS) launch.  Round 4: the zeroing of the A
  // fragments beyond K used to sit in every K-tile as 49 predicated selects between the fragment reads and the MFMAs (hipcc
  // if-converts the uniform test) - the weight gradients' K = 12000 is not a multiple of 64, so they all carried it; now
  // only the peeled last K-tile does.
  auto readA = [&](int bufoff, int a, int t, auto lastk_tag) {
    constexpr bool LASTK = decltype(lastk_tag)::value;
    if constexpr (ABL == 2) return;
    if constexpr (A_KS) {
      const char* img = smem + bufoff + a * P8_HALF;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ar[mi][kk] = frag_ks<false>(img, 64 * wr + 32 * mi, kk, lane);
      if constexpr (B_KS && LASTK) {
        if (tail && kt0 + t == P.ktiles - 1) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if (16 * kk + 8 * h >= klast) {
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int j = 0; j < 8; ++j) ar[mi][kk][j] = (bf16_t)0.f;
            }
        }
      }
    } else {
      const char* img = smem + bufoff + a * P8_HALF + (a == 0 ? arow : arow1);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (a == 1 && mi >= MI1) break;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ar[mi][kk] = *reinterpret_cast<const bf16x8*>(img + mi * 4096 + xo[kk]);
      }
    }
  };
  auto readB = [&](int bufoff, int b, bf16x8 (&br)[4]) {
    if constexpr (ABL == 2) return;
    if constexpr (B_KS) {
      const char* img = smem + bufoff + (2 + b) * P8_HALF;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) br[kk] = frag_ks<false>(img, 32 * wc, kk, lane);
    } else {
      const char* img = smem + bufoff + (2 + b) * P8_HALF + brow;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) br[kk] = *reinterpret_cast<const bf16x8*>(img + xo[kk]);
    }
  };
#define P8_MMA(A_, B_, BR_)                                                                               \
  do {                                                                                                    \
    if constexpr (ABL == 2) break;                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int mi = 0; mi < ((A_) == 0 ? 2 : MI1); ++mi) \
        acc[2 * (A_) + mi][B_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BR_[kk], ar[mi][kk], acc[2 * (A_) + mi][B_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                        \
  } while (0)

  // one K-tile = phases 1..4 of the table on buffer OWN_ (phases 5..8 are the same with the buffers swapped).
  // MID_: statement run between phase 2 and phase 3 (the persistent kernel switches its staging offsets to the next tile there)
#define P8_KTILE(OWN_, T_, MID_) P8_KTILE_L(OWN_, T_, MID_, std::false_type)
#define P8_KTILE_L(OWN_, T_, MID_, LASTK_)                                    \
  do {                                                                        \
    const int own = (OWN_), oth = own ^ 1;                                    \
    const int ownoff = own * P8_BUF;                                          \
    readB(ownoff, 0, b0);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                        \
    readA(ownoff, 0, (T_), LASTK_{});                                         \
    stage(oth, 3, (T_) + 1);                                                  \
    p8_barrier();                                                             \
    P8_MMA(0, 0, b0);                                                         \
    p8_barrier();                                                             \
                                                                              \
    readB(ownoff, 1, b1);                                                     \
    stage(oth, 1, (T_) + 1);                                                  \
    wait_vmcnt<(BM == 256 ? 8 : 7)>();                                        \
    p8_barrier();                                                             \
    P8_MMA(0, 1, b1);                                                         \
    p8_barrier();                                                             \
    MID_;                                                                     \
    readA(ownoff, 1, (T_), LASTK_{});                                         \
    stage(own, 2, (T_) + 2);                                                  \
    p8_barrier();                                                             \
    P8_MMA(1, 1, b1);                                                         \
    p8_barrier();                                                             \
                                                                              \
    stage(own, 0, (T_) + 2);                                                  \
    wait_vmcnt<(BM == 256 ? 6 : 5)>();                                        \
    p8_barrier();                                                             \
    P8_MMA(1, 0, b0);                                                         \
    p8_barrier();                                                             \
  } while (0)

  if constexpr (PERSIST) {
    // ---- persistent form (gridDim.y == gridDim.z == 1, nt == P.ktiles >= 2): the K-tile stream runs on across tiles.
    // During a tile's last two K-tile steps the stages fetch the next tile's K-tile 0 (whole) and the first-read half of
    // its K-tile 1 - exactly what the prologue below fetches for the first tile - so between tiles the ring never drains and
    // only the epilogue separates one tile's last MFMA from the next tile's first.  The epilogue's staging lives where no
    // DMA is in flight: the A^1 / B^1 half-tiles of the buffer the last K-tile was read from (waves 0-3) and the 32 KiB
    // above the ring (waves 4-7).
    stage(0, 2, 0);
    stage(0, 0, 0);
    stage(0, 3, 0);
    stage(0, 1, 0);
    stage(1, 2, 1);
    stage(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p8_barrier();
    if (wr == 1) p8_barrier();  // this group runs one barrier behind
    int g = 0;                  // K-tiles multiplied so far: the ring's parity
    // (Round 4, measured and removed: the finished rows of a tile held back in registers - two of a wave's three pieces fit -
    // and stored one per K-tile step under the NEXT tile's MFMAs, all waves at the head of the step or one wave per phase.
    // The launches got 8-10 us SLOWER (fc1 forward 81 -> 88, fc2 dgrad 76 -> 87): the CU's vector-memory pipe that takes a
    // store (~73 cycles per wave-instruction whatever its width: the 7 k-cycle epilogue of a 96 KiB tile IS 96 of them) is
    // the pipe the K loop's LDS-DMA keeps busy, so a store moved into the loop delays the staging it was meant to hide
    // under: profiles/r04_gemm_deferred_stores_ab.txt.)
    for (;;) {
      int64_t nm0 = 0, nn0 = 0;
      const int nslot = next_slot(slot + (int)gridDim.x, nm0, nn0);
      has_next = nslot < P.slots;
#pragma unroll
      for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      for (int t = 0; t < nt; ++t, ++g) {
        // (t == nt - 2, after phase 2: every later stage of this tile's loop belongs to the next tile)
        P8_KTILE(g & 1, t, if (has_next && t == nt - 2) { set_offsets(nm0, nn0); stage_next = true; });
      }
      stage_next = false;
      if (wr == 0) p8_barrier();  // rejoin the other group: nobody reads the ring any more
      {
        const int last = (g - 1) & 1;  // buffer of the last K-tile: its A^1 / B^1 halves are not being refilled
        char* E = wave < 4 ? smem + last * P8_BUF + (1 + 2 * (wave >> 1)) * P8_HALF + (wave & 1) * 8192
                           : smem + 2 * P8_BUF + (wave - 4) * 8192;
        const int64_t mw0 = m0 + wr * WR, nw0 = n0 + wc * 64;
        if (TMI_DBG(P) & 1) {
          if (acc[0][0][0] + acc[NMI - 1][1][5] == 123.456f) reinterpret_cast<float*>(d.C)[0] = 0.f;
        } else {
          // (written out: LLVM's pragma-unroll threshold declines four copies of the fp32 epilogue, and a rolled loop
          // indexes acc dynamically - 576 B of scratch per lane in the weight-gradient kernels)
          wide_epilogue<TC, 2>(P, acc[0][0], acc[0][1], E, mw0, nw0, bz, lane, false);
          wide_epilogue<TC, 2>(P, acc[1][0], acc[1][1], E, mw0 + 32, nw0, bz, lane, false);
          wide_epilogue<TC, 2>(P, acc[2][0], acc[2][1], E, mw0 + 64, nw0, bz, lane, false);
          if constexpr (NMI == 4) wide_epilogue<TC, 2>(P, acc[NMI - 1][0], acc[NMI - 1][1], E, mw0 + 96, nw0, bz, lane, false);
        }
      }
      if (!has_next) break;
      slot = nslot;
      m0 = nm0;
      n0 = nn0;
      p8_barrier();               // every wave is done with its staging image before the next stages land on it
      if (wr == 1) p8_barrier();  // stagger again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (unread) stages
    return;
  } else {
  const bool stamping = (TMI_DBG(P) & 16) && blockIdx.x == 8 && blockIdx.y == 0 && wave == 0;  // diagnostics: where a tile's cycles go
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
  if (stamping) ts[0] = stamp();
  if (nt > 0) {  // (uniform over the workgroup)
    // ---- prologue: tile 0 whole, the first-read half of tile 1
    stage(0, 2, 0);
    stage(0, 0, 0);
    stage(0, 3, 0);
    stage(0, 1, 0);
    stage(1, 2, 1);
    stage(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p8_barrier();
    if (wr == 1) p8_barrier();  // this group runs one barrier behind
    if (stamping) ts[1] = stamp();

    // one K-tile per iteration
    in_loop = true;
    if constexpr (A_KS && B_KS) {  // (the only layout with a K tail: its last K-tile is peeled)
      for (int t = 0; t < nt - 1; ++t) P8_KTILE(t & 1, t, (void)0);
      P8_KTILE_L((nt - 1) & 1, nt - 1, (void)0, std::true_type);
    } else {
      for (int t = 0; t < nt; ++t) P8_KTILE(t & 1, t, (void)0);
    }
    if (stamping) ts[2] = stamp();
    if (wr == 0) p8_barrier();  // rejoin the other group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (unread) stages must land before LDS is reused
    p8_barrier();
    if (stamping) ts[3] = stamp();
  }

  if (TMI_DBG(P) & 1) {  // diagnostics: no epilogue (keep the accumulators live)
    if (acc[0][0][0] + acc[NMI - 1][1][5] == 123.456f) reinterpret_cast<float*>(d.C)[0] = 0.f;
    return;
  }
  char* E = smem + wave * 8192;
  {
    const int64_t mw0 = m0 + wr * WR, nw0 = n0 + wc * 64;  // (written out: see the persistent form)
    FastParams Q = P;
    if (P.xsplit) {  // (the split's slab: the epilogue's own blockIdx.y term is zero in this mapping)
      Q.d.C = reinterpret_cast<TC*>(d.C) + (int64_t)xsp * P.split_c_stride;
      Q.split_c_stride = 0;
    }
    wide_epilogue<TC, 2>(Q, acc[0][0], acc[0][1], E, mw0, nw0, bz, lane, false);
    wide_epilogue<TC, 2>(Q, acc[1][0], acc[1][1], E, mw0 + 32, nw0, bz, lane, false);
    wide_epilogue<TC, 2>(Q, acc[2][0], acc[2][1], E, mw0 + 64, nw0, bz, lane, false);
    if constexpr (NMI == 4) wide_epilogue<TC, 2>(Q, acc[NMI - 1][0], acc[NMI - 1][1], E, mw0 + 96, nw0, bz, lane, false);
  }
  if (stamping) {
    ts[4] = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[5] = stamp();
    if (lane == 0)
      for (int i = 0; i < 5; ++i) g_gemm_stamps[i] = ts[i + 1] - ts[i];
  }
  }
#undef P8_KTILE
#undef P8_KTILE_L
#undef P8_MMA
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int64_t rup8(int64_t x) { return (x + 7) / 8 * 8; }

// split-K through workspace slabs: run `launch` on a copy of P whose C is the workspace, then reduce
template <typename F>
int launch_with_slabs(const FastParams& P, int splitk, hipStream_t stream, F&& launch) {
  const tmi_gemm_desc& d = P.d;
  FastParams Q = P;
  Q.d.C = d.workspace;
  Q.d.ldc = d.N;
  Q.d.c_sb = d.M * d.N;
  Q.d.accumulate = 0;
  Q.split_c_stride = d.nbatch * d.M * d.N;
  Q.drop_thr = 0;
  Q.wide = d.N % 4 == 0;
  Q.epi = epi_class(Q.d, Q.wide != 0);
  launch(Q);
  int rc = tmi_check_launch("tmi_gemm(split-K)");
  if (rc) return rc;
  float* C = reinterpret_cast<float*>(d.C);
  const int vec_ok = d.N % 4 == 0 && d.ldc % 4 == 0 && d.c_sb % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
  const int64_t n = d.nbatch * d.M * d.N;
  int64_t blocks = (n / 4 + 255) / 256;
  static const int64_t rcap = [] { const char* e = getenv("TMI_REDUCE_BLOCKS"); return e ? atoll(e) : 2048ll; }();
  if (blocks > rcap) blocks = rcap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     reinterpret_cast<const float*>(d.workspace), splitk, Q.split_c_stride, C, d.M, d.N, d.ldc, d.c_sb,
                     d.nbatch, d.accumulate, vec_ok);
  return tmi_check_launch("tmi_gemm(split-K reduce)");
}

// Order of an XCD's workgroups inside its rectangle of the tile grid.  Workgroups that are resident together should
// share the panel of the operand that does NOT fit the XCD's L2 (it is then fetched once while they run), and re-read
// the other operand from L2.  Walking along n (consecutive workgroups share an A row-panel) suits activations x
// weights (M = 12000: A is the stream, the weights stay in L2); the LM head (M = 800, N = 51904: 80 MB of weights,
// 1.2 MB of activations) and its weight gradient want the opposite, or every row-tile re-fetches each weight panel.
inline int walk_along_m(const tmi_gemm_desc& d, int xm, int xn) {
  static const int force = [] { const char* e = getenv("TMI_GEMM_WALK_M"); return e ? atoi(e) : -1; }();
  if (force >= 0) return force;
  const double a_share = (double)d.M * (double)d.K * (double)d.kbatch * 2.0 / xm;  // bytes of A one XCD touches
  const double b_share = (double)d.N * (double)d.K * (double)d.kbatch * 2.0 / xn;
  return b_share > 3.0 * 1048576.0 && a_share < b_share;
}

template <typename TC, bool A_KS, bool B_KS, int CFG>
int launch_cfg(const tmi_gemm_desc& d, hipStream_t stream) {
  using K = Cfg<CFG>;
  constexpr int NW = (K::BM / K::WM) * (K::BN / K::WN);
  constexpr int LDS_BYTES = kKG<CFG> * K::NSTAGE * (K::BM + K::BN) * 128;
  static_assert(LDS_BYTES >= NW * (K::WN == 32 ? 4096 : 8192), "epilogue staging must fit the ring");
  FastParams P;
  P.d = d;
  P.tiles_m = (int)((d.M + K::BM - 1) / K::BM);
  P.tiles_n = (int)((d.N + K::BN - 1) / K::BN);
  P.ktiles = (int)((d.K + 63) / 64);
  P.a_cols_rd = rup8(d.M);
  P.b_cols_rd = rup8(d.N);
  const int vecC = 16 / (int)sizeof(TC);
  P.wide = al16(d.C) && d.ldc % vecC == 0 && d.c_sb % vecC == 0 && (!d.aux_out || al16(d.aux_out)) &&
           (!d.aux_in || al16(d.aux_in)) && (!d.resid || (al16(d.resid) && d.r_ld % vecC == 0 && d.r_sb % vecC == 0));
  P.epi = epi_class(d, P.wide != 0);
  static const int dbg = [] { const char* e = getenv("TMI_GEMM_DBG"); return e ? atoi(e) : 0; }();
  P.dbg = dbg;
  P.split_c_stride = 0;
  P.xsplit = 0;
  P.slots = 0;
  P.drop_thr = tmi_drop_thr(d.dropout_p);
  P.drop_key = tmi_stream_key(d.dropout_seed, 0u);
  P.drop_scale = tmi_keep_scale(P.drop_thr);
  bool ws_split = false;
  auto kern = gemm_fast_kernel<TC, A_KS, B_KS, CFG>;
#ifdef TMI_GEMM_EXPERIMENTS
  if constexpr (sizeof(TC) == 2 && !A_KS && !B_KS && (CFG < 2 || CFG == 4 || CFG == 6)) {  // ablation builds exist for the bf16-out KC-A kernels only
    if ((dbg & 6) == 2) kern = gemm_fast_kernel<TC, A_KS, B_KS, CFG, 2>;
    if ((dbg & 6) == 4) kern = gemm_fast_kernel<TC, A_KS, B_KS, CFG, 4>;
    if (dbg == 8) kern = gemm_fast_kernel<TC, A_KS, B_KS, CFG, 8>;
    if (dbg & 14) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  }
#endif
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fast_kernel<TC, A_KS, B_KS, CFG>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) {
    tmi_set_error("tmi_gemm(fast): cannot raise the dynamic LDS limit");
    return TMI_ERR_LAUNCH;
  }
  // XCD partition: xn column groups x (8 / xn) row groups, one per XCD.  Aim for one XCD's share of
  // B (K x N/xn bf16) <= ~2 MiB, but never pay for it with padding: among the four splits take the one
  // with the fewest (padded) workgroups, nearest to the aimed one on ties.
  static const int force_xn = [] { const char* e = getenv("TMI_GEMM_XN"); return e ? atoi(e) : 0; }();
  int aim = 1;
  while (aim < 8 && (double)d.K * (double)d.kbatch * ((double)d.N / aim) * 2.0 > 2.0 * 1048576.0 && P.tiles_n >= 2 * aim) aim *= 2;
  int xn = aim;
  {
    int64_t best = -1;
    int best_dist = 0;
    for (int cand = 1; cand <= 8; cand *= 2) {
      const int64_t padded = (int64_t)((P.tiles_m + 8 / cand - 1) / (8 / cand)) * ((P.tiles_n + cand - 1) / cand);
      int dist = 0;
      for (int v = cand; v < aim; v *= 2) ++dist;
      for (int v = aim; v < cand; v *= 2) ++dist;
      if (best < 0 || padded < best || (padded == best && dist < best_dist)) {
        best = padded;
        best_dist = dist;
        xn = cand;
      }
    }
  }
  if (force_xn) xn = force_xn;
  P.xn = xn;
  P.xm = 8 / xn;
  P.ptm = (P.tiles_m + P.xm - 1) / P.xm;
  P.ptn = (P.tiles_n + P.xn - 1) / P.xn;
  P.walk_m = walk_along_m(d, P.xm, P.xn);
  int splitk = d.splitk > 1 ? d.splitk : 1;
  if (d.splitk == 0 && d.out_dtype == TMI_F32 && !d.bias && !d.accumulate && !d.act && !d.aux_out && !d.aux_in &&
      !d.resid && d.scale_cols <= 0) {
    // Auto split-K for weight-gradient shapes (fp32 atomics into a zeroed C).  Measured on MI355X
    // (tools/wgrad_bench.py): a split costs another pass of atomics over C at only ~0.5 TB/s, and
    // splits beyond one resident round (two workgroups per CU for the small tiles, one for the large)
    // do not shorten the K loop.  So: as many as fit one round, at most 5, at least 4 K-tiles each, and
    // the total atomic traffic bounded by what the K loop can amortise (160 KB per K-tile + 3 MB).
    const int64_t wgs = (int64_t)8 * P.ptm * P.ptn * d.nbatch;
    const int64_t its = (int64_t)d.kbatch * P.ktiles;
    const int64_t slots = (CFG == 1 || CFG == 3 || CFG == 5 || CFG == 8) ? 256 : 512;
    const int64_t out_bytes = d.M * d.N * d.nbatch * 4;
    int64_t want = slots / wgs;
    if (want > its / 4) want = its / 4;
    // with a workspace the reduction is plain stores + one streaming reduce pass; the fp32-atomic
    // fallback pays ~2 us per MB per split, so it is bounded by what the K loop amortises
    const int64_t slab_bytes = out_bytes;  // one fp32 [nbatch][M][N] slab per split
    static const int no_ws = [] { const char* e = getenv("TMI_GEMM_NO_WS"); return e ? atoi(e) : 0; }();
    static const int ws_cap = [] { const char* e = getenv("TMI_GEMM_WS_MAXSPLIT"); return e ? atoi(e) : 8; }();
    // (measured, tools/wgrad_bench.py: slabs + reduce launch match or beat atomics for long reductions,
    // lose to them below ~64 K-tiles where the extra launch shows)
    bool use_ws = !no_ws && sizeof(TC) == 4 && its >= 64 && d.workspace && (reinterpret_cast<uintptr_t>(d.workspace) & 15) == 0 &&
                  2 * slab_bytes <= d.workspace_bytes;
    if (use_ws) {
      if (want > ws_cap) want = ws_cap;
      if (want > d.workspace_bytes / slab_bytes) want = d.workspace_bytes / slab_bytes;
    } else {
      if (want > 5) want = 5;
      const int64_t budget = its * 160 * 1024 + 3 * 1048576;
      if (want > budget / out_bytes) want = budget / out_bytes;
      // tmi_set_deterministic: at most TWO atomic contributions per element (a + b = b + a: the sum of two does not depend
      // on the arrival order; three or more do - ADVICE r4)
      if (tmi_deterministic() && want > 2) want = 2;
    }
    static const int force_split = [] { const char* e = getenv("TMI_GEMM_SPLIT"); return e ? atoi(e) : 0; }();
    if (force_split > 0) want = force_split < its ? force_split : its;
    if (use_ws && want * slab_bytes > d.workspace_bytes) use_ws = false;
    if (use_ws && want > 1) ws_split = true;
    splitk = want < 1 ? 1 : (int)want;
  }
  if (K::WN == 32) {
    splitk = 1;
    ws_split = false;
  }
  static const int xsplit_on = [] { const char* e = getenv("TMI_GEMM_XSPLIT"); return e ? atoi(e) : 1; }();
  if (xsplit_on && d.nbatch == 1 && (splitk == 2 || splitk == 4 || splitk == 8)) {  // split-K over XCD groups: see launch_p8
    const int groups = 8 / splitk;
    int bxn = 1;
    int64_t bpad = -1;
    for (int cand = 1; cand <= groups; cand *= 2) {
      const int64_t padded = (int64_t)((P.tiles_m + groups / cand - 1) / (groups / cand)) * ((P.tiles_n + cand - 1) / cand);
      if (bpad < 0 || padded < bpad) { bpad = padded; bxn = cand; }
    }
    P.xsplit = splitk;
    P.xn = bxn;
    P.xm = groups / bxn;
    P.ptm = (P.tiles_m + P.xm - 1) / P.xm;
    P.ptn = (P.tiles_n + P.xn - 1) / P.xn;
    splitk = 1;  // (grid.y)
  }
  dim3 grid((unsigned)(8 * P.ptm * P.ptn), (unsigned)splitk, (unsigned)d.nbatch);
  if (ws_split) return launch_with_slabs(P, P.xsplit ? P.xsplit : splitk, stream, [&](const FastParams& Q) {
    hipLaunchKernelGGL(kern, grid, dim3(NW * (K::SPEC ? 128 : 64) * kKG<CFG>), LDS_BYTES, stream, Q);
  });
  hipLaunchKernelGGL(kern, grid, dim3(NW * (K::SPEC ? 128 : 64) * kKG<CFG>), LDS_BYTES, stream, P);
  return tmi_check_launch("tmi_gemm(fast)");
}

// eight-phase kernel: one K batch, operand spans addressable with 32-bit byte offsets; a K tail only
// when both operands are k-strided (then K % 8 == 0); split-K only through the workspace
inline bool p8_eligible(const tmi_gemm_desc& d, bool a_ks, bool b_ks) {
  const double a_span = a_ks ? ((double)(d.K - 1) * (double)d.a_sk + (double)d.M + 8.0) * 2.0
                             : ((double)(d.M - 1) * (double)d.a_sm + (double)d.K) * 2.0;
  const double b_span = b_ks ? ((double)(d.K - 1) * (double)d.b_sk + (double)d.N + 8.0) * 2.0
                             : ((double)(d.N - 1) * (double)d.b_sn + (double)d.K) * 2.0;
  const bool k_ok = d.K % 64 == 0 || (a_ks && b_ks && d.K % 8 == 0);
  return d.kbatch == 1 && k_ok && d.K >= 128 && d.a_sm >= 0 && d.b_sn >= 0 && d.a_sk >= 0 && d.b_sk >= 0 &&
         a_span < 4.0e9 && b_span < 4.0e9;
}

template <typename TC, bool A_KS, bool B_KS, int BM = 256>
int launch_p8(const tmi_gemm_desc& d, hipStream_t stream) {
  FastParams P;
  P.d = d;
  P.tiles_m = (int)((d.M + BM - 1) / BM);
  P.tiles_n = (int)((d.N + 255) / 256);
  P.ktiles = (int)((d.K + 63) / 64);
  P.a_cols_rd = rup8(d.M);
  P.b_cols_rd = rup8(d.N);
  const int vecC = 16 / (int)sizeof(TC);
  P.wide = al16(d.C) && d.ldc % vecC == 0 && d.c_sb % vecC == 0 && (!d.aux_out || al16(d.aux_out)) &&
           (!d.aux_in || al16(d.aux_in)) && (!d.resid || (al16(d.resid) && d.r_ld % vecC == 0 && d.r_sb % vecC == 0));
  P.epi = epi_class(d, P.wide != 0);
  static const int dbg8 = [] { const char* e = getenv("TMI_GEMM_DBG"); return e ? atoi(e) : 0; }();
  P.dbg = dbg8;
  P.split_c_stride = 0;
  P.xsplit = 0;
  P.drop_thr = tmi_drop_thr(d.dropout_p);
  P.drop_key = tmi_stream_key(d.dropout_seed, 0u);
  P.drop_scale = tmi_keep_scale(P.drop_thr);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<TC, A_KS, B_KS, BM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P8_BUF);
  if (attr != hipSuccess) {
    tmi_set_error("tmi_gemm(p8): cannot raise the dynamic LDS limit");
    return TMI_ERR_LAUNCH;
  }
  // XCD partition with the fewest padded workgroups (see launch_cfg)
  int aim = 1;
  while (aim < 8 && (double)d.K * ((double)d.N / aim) * 2.0 > 2.0 * 1048576.0 && P.tiles_n >= 2 * aim) aim *= 2;
  int xn = aim;
  int64_t best = -1;
  int best_dist = 0;
  for (int cand = 1; cand <= 8; cand *= 2) {
    const int64_t padded = (int64_t)((P.tiles_m + 8 / cand - 1) / (8 / cand)) * ((P.tiles_n + cand - 1) / cand);
    int dist = 0;
    for (int v = cand; v < aim; v *= 2) ++dist;
    for (int v = aim; v < cand; v *= 2) ++dist;
    if (best < 0 || padded < best || (padded == best && dist < best_dist)) {
      best = padded;
      best_dist = dist;
      xn = cand;
    }
  }
  P.xn = xn;
  P.xm = 8 / xn;
  P.ptm = (P.tiles_m + P.xm - 1) / P.xm;
  P.ptn = (P.tiles_n + P.xn - 1) / P.xn;
  P.walk_m = walk_along_m(d, P.xm, P.xn);
  // split-K (library-chosen only, fp32 output, workspace slabs only): fill the 256 CUs, >= 6 K-tiles per split
  int splitk = 1;
  if constexpr (sizeof(TC) == 4) {
    const int64_t slab_bytes = d.M * d.N * d.nbatch * 4;
    if (d.splitk == 0 && d.workspace && (reinterpret_cast<uintptr_t>(d.workspace) & 15) == 0 && !d.bias && !d.act &&
        !d.aux_out && !d.aux_in && !d.resid && d.scale_cols <= 0) {
      const int64_t wgs = (int64_t)8 * P.ptm * P.ptn * d.nbatch;
      // (cap 4, round 3: measured in the step - the weight gradients run beside the dgrad chain, where a split costs slab
      // traffic and CUs the other stream could use: 8.82 -> 8.74 ms/step against a cap of 8; 2 and 1 lose)
      static const int cap = [] { const char* e = getenv("TMI_GEMM_P8_MAXSPLIT"); return e ? atoi(e) : 4; }();
      int64_t want = 256 / wgs;
      if (want > P.ktiles / 6) want = P.ktiles / 6;
      if (want > cap) want = cap;
      // (round 4) a very deep reduction onto a handful of tiles (>= 32 K-tiles per split left): as many splits as fill the
      // chip, a power of two so that they can be dealt over the XCDs (below)
      static const int deep = [] { const char* e = getenv("TMI_GEMM_P8_DEEPSPLIT"); return e ? atoi(e) : 32; }();
      const int64_t tiles = (int64_t)P.tiles_m * P.tiles_n;
      if (d.nbatch == 1 && deep > cap && tiles <= 32) {
        int64_t w2 = 1;
        while (w2 * 2 <= deep && w2 * 2 * tiles <= 256 && P.ktiles / (w2 * 2) >= 32) w2 *= 2;
        if (w2 > want) want = w2;
      }
      if (want > d.workspace_bytes / slab_bytes) want = d.workspace_bytes / slab_bytes;
      if (want > 1) splitk = (int)want;
    }
  }
  P.slots = 8 * P.ptm * P.ptn;
  // More tiles than CUs (bf16-output forward / dgrad launches of two or three rounds): the persistent form - 256 workgroups
  // walk the slots, the K-tile ring runs on from one tile into the next (no prologue after the first tile) and the epilogue
  // is staged outside the ring.  TMI_GEMM_P8_PERSIST=0 restores one workgroup per tile.
  if constexpr (!(A_KS && B_KS)) {
    static const int persist = [] { const char* e = getenv("TMI_GEMM_P8_PERSIST"); return e ? atoi(e) : 1; }();
    if (persist && splitk == 1 && d.nbatch == 1 && P.slots > 256 && P.ktiles >= 2) {
      constexpr int LDS_P = 2 * P8_BUF + 32768;
      static const hipError_t attr_p = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<TC, A_KS, B_KS, BM, true>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
      if (attr_p == hipSuccess) {
        hipLaunchKernelGGL((gemm_p8_kernel<TC, A_KS, B_KS, BM, true>), dim3(256), dim3(512), LDS_P, stream, P);
        return tmi_check_launch("tmi_gemm(p8 persistent)");
      }
    }
  }
  // Split-K over XCD groups (round 4): with the split on blockIdx.y every XCD serves all S K-ranges, each with its own
  // panels; dealt over groups of 8 / S XCDs instead, an XCD's L2 holds ONE K-range and (tiles / (8 / S)) tiles that share its
  // row and column panels (fc weight gradient, S = 4: 15 -> 9 panel fetches from beyond L2 per 18 tiles).
  static const int xsplit_on = [] { const char* e = getenv("TMI_GEMM_XSPLIT"); return e ? atoi(e) : 1; }();
  if (xsplit_on && d.nbatch == 1 && (splitk == 16 || splitk == 32)) {  // splitk / 8 K ranges per XCD, each with all the tiles
    P.xsplit = splitk;
    P.xn = P.xm = 1;
    P.ptm = P.tiles_m;
    P.ptn = P.tiles_n;
    P.slots = 8 * P.ptm * P.ptn * (splitk / 8);
    dim3 gridx((unsigned)P.slots, 1u, 1u);
    return launch_with_slabs(P, splitk, stream, [&](const FastParams& Q) {
      hipLaunchKernelGGL((gemm_p8_kernel<TC, A_KS, B_KS, BM>), gridx, dim3(512), 2 * P8_BUF, stream, Q);
    });
  }
  if (xsplit_on && d.nbatch == 1 && (splitk == 2 || splitk == 4 || splitk == 8)) {
    const int groups = 8 / splitk;  // XCDs per split: 4, 2 or 1
    int bxn = 1;
    int64_t bpad = -1;
    for (int cand = 1; cand <= groups; cand *= 2) {
      const int64_t padded = (int64_t)((P.tiles_m + groups / cand - 1) / (groups / cand)) * ((P.tiles_n + cand - 1) / cand);
      if (bpad < 0 || padded < bpad) { bpad = padded; bxn = cand; }
    }
    P.xsplit = splitk;
    P.xn = bxn;
    P.xm = groups / bxn;
    P.ptm = (P.tiles_m + P.xm - 1) / P.xm;
    P.ptn = (P.tiles_n + P.xn - 1) / P.xn;
    P.slots = 8 * P.ptm * P.ptn;
    dim3 gridx((unsigned)(8 * P.ptm * P.ptn), 1u, 1u);
    return launch_with_slabs(P, splitk, stream, [&](const FastParams& Q) {
      hipLaunchKernelGGL((gemm_p8_kernel<TC, A_KS, B_KS, BM>), gridx, dim3(512), 2 * P8_BUF, stream, Q);
    });
  }
  dim3 grid((unsigned)(8 * P.ptm * P.ptn), (unsigned)splitk, (unsigned)d.nbatch);
#ifdef TMI_GEMM_EXPERIMENTS
  if constexpr (sizeof(TC) == 2 && !A_KS && !B_KS && BM == 192) {  // diagnostics: TMI_GEMM_DBG & 6 = 2 no MFMA / reads, 4 no staging in the loop
    if ((dbg8 & 6) && splitk == 1) {
      auto kern = (dbg8 & 6) == 2 ? gemm_p8_kernel<TC, A_KS, B_KS, BM, false, 2> : gemm_p8_kernel<TC, A_KS, B_KS, BM, false, 4>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P8_BUF);
      hipLaunchKernelGGL(kern, grid, dim3(512), 2 * P8_BUF, stream, P);
      return tmi_check_launch("tmi_gemm(p8 ablation)");
    }
  }
#endif
  if (splitk > 1) return launch_with_slabs(P, splitk, stream, [&](const FastParams& Q) {
    hipLaunchKernelGGL((gemm_p8_kernel<TC, A_KS, B_KS, BM>), grid, dim3(512), 2 * P8_BUF, stream, Q);
  });
  hipLaunchKernelGGL((gemm_p8_kernel<TC, A_KS, B_KS, BM>), grid, dim3(512), 2 * P8_BUF, stream, P);
  return tmi_check_launch("tmi_gemm(p8)");
}

template <typename TC, bool A_KS, bool B_KS>
int launch_fast(const tmi_gemm_desc& d, hipStream_t stream) {
  static const int force = [] { const char* e = getenv("TMI_GEMM_CFG"); return e ? atoi(e) : -1; }();
  // Measured on MI355X (tools/gemm_bench.py): the 256x256 tile wins for bf16-output GEMMs with a
  // long reduction (K >= 1536: its unoverlapped epilogue is amortised) or whose tile count fills
  // whole rounds of the 256 CUs; weight gradients (split-K, atomic epilogue) and small problems
  // run better on the 128x128 tile with two workgroups per CU.
  const int64_t big_tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256) * d.nbatch;
  const double round_eff = (double)big_tiles / (double)(((big_tiles + 255) / 256) * 256);
  const bool wgrad_like = d.splitk == 0 && d.out_dtype == TMI_F32;
  bool big = d.out_dtype == TMI_BF16 && d.M >= 2048 && d.N >= 512 && (d.K * d.kbatch >= 1536 || round_eff >= 0.8);
  if constexpr (!A_KS || B_KS) {  // (k-strided A with k-contiguous B is not instantiated)
    if (force == 10 && p8_eligible(d, A_KS, B_KS) && d.splitk <= 1) return launch_p8<TC, A_KS, B_KS>(d, stream);
  }
  if constexpr (!A_KS) {
    if (force == 14 && p8_eligible(d, false, B_KS) && d.splitk <= 1) return launch_p8<TC, false, B_KS, 192>(d, stream);
  }
  // too few 128x128 tiles to occupy the chip (and not a split-K weight gradient): 64x64 tiles
  const int64_t mid_tiles = ((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nbatch;
  const bool small = !wgrad_like && mid_tiles < 200 && d.M >= 64 && d.N >= 64;
  // (configurations 0-3 and 7-10 are round-1/2 experiments no rule picks: built only with -DTMI_GEMM_EXPERIMENTS - they were
  // 60 % of this file's compile time)
#ifdef TMI_GEMM_EXPERIMENTS
  if (force == 0) return launch_cfg<TC, A_KS, B_KS, 0>(d, stream);
  if (force == 1) return launch_cfg<TC, A_KS, B_KS, 1>(d, stream);
  if (force == 2) return launch_cfg<TC, A_KS, B_KS, 2>(d, stream);
  if (force == 3) return launch_cfg<TC, A_KS, B_KS, 3>(d, stream);
  if (force == 7) return launch_cfg<TC, A_KS, B_KS, 7>(d, stream);
  if (force == 8) return launch_cfg<TC, A_KS, B_KS, 8>(d, stream);
  if (force == 9) return launch_cfg<TC, A_KS, B_KS, 9>(d, stream);
  if (force == 11) return launch_cfg<TC, A_KS, B_KS, 10>(d, stream);
#endif
  if (force == 4) return launch_cfg<TC, A_KS, B_KS, 4>(d, stream);
  if (force == 5) return launch_cfg<TC, A_KS, B_KS, 5>(d, stream);
  if (force == 6) return launch_cfg<TC, A_KS, B_KS, 6>(d, stream);
  if (force == 12 && d.splitk <= 1) return launch_cfg<TC, A_KS, B_KS, 11>(d, stream);
  if (force == 13 && d.splitk <= 1) return launch_cfg<TC, A_KS, B_KS, 12>(d, stream);
  if (force == 15 && d.splitk <= 1 && d.K % 64 == 0) return launch_cfg<TC, A_KS, B_KS, 13>(d, stream);
  if (force == 16 && d.splitk <= 1 && d.K % 64 == 0) return launch_cfg<TC, A_KS, B_KS, 14>(d, stream);
  if (small) {
    // measured (tools/gemm_small.py, M = 800): four waves of 32x32 beat two of 32x64 wherever the grid is
    // at most one workgroup per CU (N 768, K 768: 12.3 -> 9.1 us) and tie elsewhere; the 4-stage ring adds
    // to that for long K on such grids (N 768, K 3072: 27.4 -> 21.4 us) and loses co-residency on larger ones
    const int64_t tiles64 = ((d.M + 63) / 64) * ((d.N + 63) / 64) * d.nbatch;
    if (d.splitk <= 1) {
      // at most one workgroup per CU: two K-groups of four waves overlap each other's per-K-tile chains
      // (tools/gemm_small2.py, M 800 N 768: K 3072 22.9 -> 17.4 us, K 768 9.2 -> 8.7 us; a third group adds nothing)
      static const int no_kg = [] { const char* e = getenv("TMI_GEMM_NO_KGROUPS"); return e ? atoi(e) : 0; }();
      if (!no_kg && tiles64 <= 256 && d.K % 64 == 0 && d.K * d.kbatch >= 512) return launch_cfg<TC, A_KS, B_KS, 13>(d, stream);
      // more than one workgroup per CU, up to two: the K-groups on 2-stage rings (Whisper-large's decoder, [800, 1280]:
      // K 1280 16.0 -> 14.8 us, K 5120 50.6 -> 41.1 us; tools/gemm_large_dec_probe.py.  TMI_GEMM_KG2=0: CFG 11)
      static const int kg2 = [] { const char* e = getenv("TMI_GEMM_KG2"); return e ? atoi(e) : 1024; }();
      if (!no_kg && kg2 > 0 && tiles64 > 256 && tiles64 <= 512 && d.K % 64 == 0 && d.K * d.kbatch >= kg2)
        return launch_cfg<TC, A_KS, B_KS, 14>(d, stream);
      if (tiles64 <= 256 && d.K * d.kbatch >= 1536) return launch_cfg<TC, A_KS, B_KS, 12>(d, stream);
      return launch_cfg<TC, A_KS, B_KS, 11>(d, stream);
    }
    return launch_cfg<TC, A_KS, B_KS, 6>(d, stream);
  }
  // fp32 results of a very deep reduction onto few tiles (the LM head's dgrad: [800, 768] from K = 51904, 134 us on the 128x128
  // kernel with 8 atomic splits): eight-phase kernel, 16 / 32 slab splits dealt two / four per XCD (launch_p8)
  if constexpr (!A_KS && sizeof(TC) == 4) {
    static const int deep_on = [] { const char* e = getenv("TMI_GEMM_P8_DEEP"); return e ? atoi(e) : 1; }();
    const int64_t t192 = ((d.M + 191) / 192) * ((d.N + 255) / 256), t256 = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    if (deep_on && force < 0 && wgrad_like && d.workspace && p8_eligible(d, false, B_KS) && d.nbatch == 1 && d.K >= 16384 &&
        d.M >= 512 && d.N >= 256 && t256 <= 32 && 16 * d.M * d.N * 4 <= d.workspace_bytes) {
      if (t192 <= 32 && (d.M + 191) / 192 * 192 < (d.M + 255) / 256 * 256) return launch_p8<TC, false, B_KS, 192>(d, stream);
      return launch_p8<TC, false, B_KS>(d, stream);
    }
  }
  // weight gradients with a large output and a long reduction: eight-phase kernel, split-K through
  // workspace slabs (-13 % against the 128x128 kernel with atomics; smaller outputs lose)
  if constexpr (A_KS && B_KS && sizeof(TC) == 4) {
    static const int no_p8w = [] { const char* e = getenv("TMI_GEMM_NO_P8"); return e ? atoi(e) : 0; }();
    // (round 4: the LM head's weight gradient - [768, 51904] from K = 800, 609 tiles - 140 -> 110 us: with two and a half
    // rounds of tiles the short reduction's prologue and epilogue overlap other tiles' loops)
    static const int wide_on = [] { const char* e = getenv("TMI_GEMM_P8_WIDE"); return e ? atoi(e) : 1; }();
    const bool wide_out = wide_on && big_tiles >= 512;
    if (!no_p8w && force < 0 && wgrad_like && d.workspace && p8_eligible(d, true, true) && d.M * d.N >= 768 * 2304 &&
        (d.K >= 4096 || (wide_out && d.K >= 512)) && d.nbatch == 1 &&
        (2 * d.M * d.N * 4 <= d.workspace_bytes || big_tiles >= 192))  // (>= 192 tiles: no split, no slabs)
      return launch_p8<TC, true, true>(d, stream);
  }
  // Eight-phase 256x256 kernel (measured, tools/gemm_p8_check.py, tools/gemm_rule_probe.py): +20 % on long
  // reductions, ahead or level from K = 1024 (d_model 1024 / 1280 layers); with a k-strided B and K = 768
  // its un-overlapped epilogue (one workgroup per CU) loses to the two co-resident 128x128 workgroups, so
  // those stay where they were.
  if constexpr (!A_KS) {
    static const int no_p8 = [] { const char* e = getenv("TMI_GEMM_NO_P8"); return e ? atoi(e) : 0; }();
    const bool light_epi = !d.aux_in && !d.aux_out && !d.act && d.N <= 1024;
    // (round 4: batched launches count their batches - the conv2 forward is 8 x [1500, 768] x K 2304 and ran 148 us on the
    // 2-stage 256x256 kernel)
    static const int p8_all = [] { const char* e = getenv("TMI_GEMM_P8_ALL"); return e ? atoi(e) : 0; }();  // (off: level or +0.04 ms in the step, profiles/r04_step_ab_p8_all.txt)
    // (round 4: the LM head's forward, [800, 51904] from K = 768 - 812 tiles walked by the persistent form: 127 -> 111 us)
    static const int wide_on = [] { const char* e = getenv("TMI_GEMM_P8_WIDE"); return e ? atoi(e) : 1; }();
    // (round 4: batched launches count their batches - the conv2 forward is 8 x [1500, 768] x K 2304; TMI_GEMM_P8_BATCHED=0: the 128x128 kernel)
    static const int p8_batched = [] { const char* e = getenv("TMI_GEMM_P8_BATCHED"); return e ? atoi(e) : 1; }();
    // (>= 160 of its 192-row tiles: Wav2Vec2's conv3, 8 x [1600, 512] x K 1536 = 144 tiles, runs 38 us on the 128x128 kernel and 46 here)
    const bool fills = ((d.M + 191) / 192) * ((d.N + 255) / 256) * d.nbatch >= 160;
    const bool m_ok = d.M >= 2048 || ((p8_all || (p8_batched && d.K >= 1024 && fills)) && d.M >= 1024 && d.M * d.nbatch >= 4096) ||
                      (wide_on && d.M >= 512 && big_tiles >= 512);
    if (!no_p8 && force < 0 && !wgrad_like && d.splitk <= 1 && p8_eligible(d, false, B_KS) && m_ok && d.N >= 256) {
      if (d.M < 1024) return launch_p8<TC, false, B_KS>(d, stream);  // (the LM head's forward: 256-row tiles measured ahead of 192)
      // 192-row tiles when they need fewer CU-rounds of work: cost = rounds of 256 workgroups x tile rows
      static const int no192 = [] { const char* e = getenv("TMI_GEMM_NO_P8_192"); return e ? atoi(e) : 0; }();
      // (round 4: on by default - with the lean epilogue the 192-row tile beats the two co-resident 128x128 workgroups in the step
      // as well: fc1 forward + GELU + saved pre-activation 115 -> 95 us, fc2 dgrad + GELU' 113 -> 97 us, step 8.22 -> 8.16 ms)
      static const int short192 = [] { const char* e = getenv("TMI_GEMM_P8_192_SHORTK"); return e ? atoi(e) : 1; }();
      const int64_t tn = (d.N + 255) / 256;
      const int64_t t256 = ((d.M + 255) / 256) * tn * d.nbatch, t192 = ((d.M + 191) / 192) * tn * d.nbatch;
      const int64_t c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
      // (the 192-row loop does ~10 % less per cycle - two of its four phases issue half the MFMAs - so it has to save more than that)
      const bool win192 = !no192 && c192 * 100 <= c256 * 85;
      // (round 4, TMI_GEMM_P8_ALL=1: short K too - with the lean epilogue the eight-phase kernel wins there as well: qkv forward
      // 12000 x 2304 x 768 65.8 us on the 2-stage 256x256 kernel in the step, 56 us here)
      const bool long_k = p8_all || d.K >= 1024 || (!B_KS && light_epi);
      // short K with heavy epilogues / k-strided weights (TMI_GEMM_P8_192_SHORTK=1, off): alone and with a plain epilogue the
      // 192-row tile wins there too (fc1 forward 12000 x 3072 x 768: 100.6 -> 80.3 us), but with the real GELU + aux epilogues
      // in the step the one-workgroup-per-CU kernel loses to two co-resident 128x128 workgroups: 9.06 -> 9.29 ms/step
      // (=2: only the forward fc1 shape - GELU + saved pre-activation, no aux_in - which runs alone on the chip: the forward
      // pass has no second stream beside it, so what counts there is the launch's own latency)
      const bool fwd_only = short192 == 2 && d.act == 1 && !d.aux_in;
      if (win192 && (long_k || short192 == 1 || fwd_only)) return launch_p8<TC, false, B_KS, 192>(d, stream);
      if (long_k) return launch_p8<TC, false, B_KS>(d, stream);
    }
  }
  return big ? launch_cfg<TC, A_KS, B_KS, 5>(d, stream) : launch_cfg<TC, A_KS, B_KS, 4>(d, stream);
}

template <bool A_KS, bool B_KS>
int launch_out(const tmi_gemm_desc& d, hipStream_t stream) {
  return d.out_dtype == TMI_F32 ? launch_fast<float, A_KS, B_KS>(d, stream) : launch_fast<bf16_t, A_KS, B_KS>(d, stream);
}


// =====================================================================================
// fp32 operands (the parity mode, --precision fp32: the reference's own arithmetic width).  Round 4: the same staging
// as the bf16 kernels instead of the generic kernel of gemm.hip (single LDS buffer, register staging, 4-way bank
// conflicts on its ds_read_b32 fragments, element-wise epilogue: 40-68 TF/s of the 157 TF/s exact-fp32 MFMA roof).
// 128 x 128 tile, 4 waves (2 x 2) of 64 x 64, K in slabs of 32 floats = the 128-byte rows of the bf16 images:
//   KC operand (k-contiguous rows): byte for byte the bf16 KC image and its DMA (the fp32 matrix is staged through its
//      bf16 view: 2 x the element strides); a 16-byte fragment read is 4 consecutive k of one row, and lane half h takes
//      chunk 2 j + h, so MFMA step (j, e) multiplies k = 8 j + 4 h + e - any pairing works as long as both operands use it;
//   KS operand (k-strided, contiguous columns): [32 k][64 columns] sub-images of 256-byte rows, filled lane-linearly by
//      the DMA (4 k-rows per wave-instruction); a fragment element is one ds_read_b32, 32 lanes on 32 consecutive columns
//      (conflict-free without a swizzle).
// v_mfma_f32_32x32x2_f32 takes 64 cycles: a slab is 64 MFMAs = 4096 cycles per wave against 32 KiB of staging and at most
// 48 LDS reads, so two co-resident workgroups (64 KiB of LDS each) keep the matrix pipe busy with the plain two-stage loop.
template <typename TC, bool A_KS, bool B_KS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const FastParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16 KiB | B 16 KiB]
  const tmi_gemm_desc& d = P.d;
  const int xcd = blockIdx.x & 7, lidx = blockIdx.x >> 3;
  int ltm, ltn;
  if (P.walk_m) { ltn = lidx / P.ptm; ltm = lidx - ltn * P.ptm; }
  else { ltm = lidx / P.ptn; ltn = lidx - ltm * P.ptn; }
  const int tm = (xcd / P.xn) * P.ptm + ltm, tn = (xcd % P.xn) * P.ptn + ltn;
  if (tm >= P.tiles_m || tn >= P.tiles_n) return;
  const int64_t m0 = (int64_t)tm * 128, n0 = (int64_t)tn * 128;
  // two batch levels (the parity mode's attention products: batch rows x heads): z = b2 * nbatch + b1
  const int64_t b2 = d.nbatch2 > 1 ? (int64_t)blockIdx.z / d.nbatch : 0, bz = (int64_t)blockIdx.z - b2 * d.nbatch;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int total_it = (int)d.kbatch * P.ktiles;
  const int nsplit = gridDim.y;
  const int per = (total_it + nsplit - 1) / nsplit;
  const int it0 = blockIdx.y * per;
  const int nt = min(total_it, it0 + per) - it0;
  const float* Abase = reinterpret_cast<const float*>(d.A) + bz * d.a_sb + b2 * d.a_sb2;
  const float* Bbase = reinterpret_cast<const float*>(d.B) + bz * d.b_sb + b2 * d.b_sb2;
  // K tail (K % 4 == 0, host-checked; 1500 keys = 46 slabs + 28): the last slab's fetches are clamped inside the operands
  // (finite, valid data) and the k >= kvalid part of the A image is zeroed before the MFMAs read it
  const int kvalid = (int)(d.K - (int64_t)(P.ktiles - 1) * 32);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // KS staging: 128 columns x 32 k = two [32 k][64 col] sub-images; wave-instruction j (16 per operand, 4 per wave) covers
  // k-rows 4 (j & 7) .. + 3 of sub-image j >> 3; a lane fetches 4 consecutive columns (clamped inside the operand)
  auto stage_ks32 = [&](const float* base, int64_t s_k, int64_t col0, int64_t ncols, int64_t k0, char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * wave + i;
      const int kr = 4 * (j & 7) + (lane >> 4);
      int64_t gc = col0 + (j >> 3) * 64 + (lane & 15) * 4;
      gc = gc + 4 <= ncols ? gc : ncols - 4;
      int64_t gk = k0 + kr;
      gk = gk < d.K ? gk : d.K - 1;
      glds16(base + gk * s_k + gc, img + j * 1024);
    }
  };
  // k-contiguous rows: the bf16 kernels' KC image byte for byte (stage_kc), with the K clamp
  auto stage_kc32 = [&](const float* base, int64_t s_row, int64_t row0, int64_t nrows, int64_t k0, char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * wave + i;
      const int rr = 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      int64_t grow = row0 + rr;
      grow = grow < nrows ? grow : nrows - 1;
      int64_t gk = k0 + 4 * c;
      gk = gk + 4 <= d.K ? gk : d.K - 4;
      glds16(base + grow * s_row + gk, img + j * 1024);
    }
  };
  auto stage = [&](int it, int buf) {
    const int kb = it / P.ktiles, kt = it - kb * P.ktiles;
    char* As = smem + buf * 32768;
    char* Bs = As + 16384;
    if constexpr (A_KS) stage_ks32(Abase + kb * d.a_skb, d.a_sk, m0, P.a_cols_rd, (int64_t)kt * 32, As);
    else stage_kc32(Abase + kb * d.a_skb, d.a_sm, m0, d.M, (int64_t)kt * 32, As);
    if constexpr (B_KS) stage_ks32(Bbase + kb * d.b_skb, d.b_sk, n0, P.b_cols_rd, (int64_t)kt * 32, Bs);
    else stage_kc32(Bbase + kb * d.b_skb, d.b_sn, n0, d.N, (int64_t)kt * 32, Bs);
  };
  auto zero_tail = [&](int it, int buf) {  // (uniform; the slab has landed and the barrier has been passed)
    if (kvalid < 32 && (it % P.ktiles) == P.ktiles - 1) {
      char* As = smem + buf * 32768;
      if constexpr (A_KS) {   // k-rows kvalid .. 31 of both [32 k][256 B] sub-images
        for (int idx = threadIdx.x; idx < 2 * (32 - kvalid) * 16; idx += 256) {
          const int img = idx / ((32 - kvalid) * 16), rem = idx % ((32 - kvalid) * 16);
          *reinterpret_cast<u32x4*>(As + img * 8192 + kvalid * 256 + rem * 16) = u32x4{0u, 0u, 0u, 0u};
        }
      } else {                // chunks kvalid / 4 .. 7 of every row, at their swizzled places
        const int nch = 8 - kvalid / 4;
        for (int idx = threadIdx.x; idx < 128 * nch; idx += 256) {
          const int row = idx / nch, c = kvalid / 4 + idx % nch;
          *reinterpret_cast<u32x4*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = u32x4{0u, 0u, 0u, 0u};
        }
      }
      __syncthreads();
    }
  };
  const int r = lane & 31, h = lane >> 5;
  auto mma = [&](const char* As, const char* Bs) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {   // k = 8 jj .. 8 jj + 7 of the slab: lane half h holds 8 jj + 4 h + e
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (A_KS) {
          const int col = wr * 64 + i * 32 + r;
          const char* p0 = As + (col >> 6) * 8192 + (col & 63) * 4 + (8 * jj + 4 * h) * 256;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[i][e] = *reinterpret_cast<const float*>(p0 + e * 256);
        } else {
          const bf16x8 t = frag_kc(As, wr * 64 + i * 32 + r, jj, h);
          a[i] = *reinterpret_cast<const f32x4*>(&t);
        }
        if constexpr (B_KS) {
          const int col = wc * 64 + i * 32 + r;
          const char* p0 = Bs + (col >> 6) * 8192 + (col & 63) * 4 + (8 * jj + 4 * h) * 256;
#pragma unroll
          for (int e = 0; e < 4; ++e) b[i][e] = *reinterpret_cast<const float*>(p0 + e * 256);
        } else {
          const bf16x8 t = frag_kc(Bs, wc * 64 + i * 32 + r, jj, h);
          b[i] = *reinterpret_cast<const f32x4*>(&t);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)   // (operands swapped: the accumulator is the C^T block wide_epilogue takes)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni][e], a[mi][e], acc[mi][ni], 0, 0, 0);
    }
  };
  if (nt > 0) {
    stage(it0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) stage(it0 + t + 1, buf ^ 1);   // the next slab's DMA flies under this slab's 64 MFMAs
      zero_tail(it0 + t, buf);
      const char* As = smem + buf * 32768;
      mma(As, As + 16384);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf ^= 1;
    }
  }
  const bool atomic = nsplit > 1;
  char* E = smem + wave * 8192;
  if (b2 == 0) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      wide_epilogue<TC, 2>(P, acc[p][0], acc[p][1], E, m0 + wr * 64 + p * 32, n0 + wc * 64, bz, lane, atomic);
  } else {  // (outer batches carry no bias / aux / residual terms: only C moves)
    FastParams Q = P;
    Q.d.C = reinterpret_cast<TC*>(d.C) + b2 * d.c_sb2;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      wide_epilogue<TC, 2>(Q, acc[p][0], acc[p][1], E, m0 + wr * 64 + p * 32, n0 + wc * 64, bz, lane, atomic);
  }
}

template <bool A_KS, bool B_KS>
int launch_f32(const tmi_gemm_desc& d, hipStream_t stream) {
  FastParams P;
  P.d = d;
  P.tiles_m = (int)((d.M + 127) / 128);
  P.tiles_n = (int)((d.N + 127) / 128);
  P.ktiles = (int)((d.K + 31) / 32);
  P.a_cols_rd = (d.M + 3) / 4 * 4;
  P.b_cols_rd = (d.N + 3) / 4 * 4;
  P.wide = al16(d.C) && d.ldc % 4 == 0 && d.c_sb % 4 == 0 && (!d.aux_out || al16(d.aux_out)) &&
           (!d.aux_in || al16(d.aux_in)) && (!d.resid || (al16(d.resid) && d.r_ld % 4 == 0 && d.r_sb % 4 == 0));
  P.epi = epi_class(d, P.wide != 0);
  static const int dbg = [] { const char* e = getenv("TMI_GEMM_DBG"); return e ? atoi(e) : 0; }();
  P.dbg = dbg;
  P.split_c_stride = 0;
  P.xsplit = 0;
  P.slots = 0;
  P.drop_thr = tmi_drop_thr(d.dropout_p);
  P.drop_key = tmi_stream_key(d.dropout_seed, 0u);
  P.drop_scale = tmi_keep_scale(P.drop_thr);
  auto kern = gemm_f32_kernel<float, A_KS, B_KS>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  if (attr != hipSuccess) {
    tmi_set_error("tmi_gemm(f32): cannot raise the dynamic LDS limit");
    return TMI_ERR_LAUNCH;
  }
  // XCD partition with the fewest padded workgroups (see launch_cfg); an fp32 operand share is twice the bf16 one
  int aim = 1;
  while (aim < 8 && (double)d.K * (double)d.kbatch * ((double)d.N / aim) * 4.0 > 2.0 * 1048576.0 && P.tiles_n >= 2 * aim) aim *= 2;
  int xn = aim;
  int64_t best = -1;
  int best_dist = 0;
  for (int cand = 1; cand <= 8; cand *= 2) {
    const int64_t padded = (int64_t)((P.tiles_m + 8 / cand - 1) / (8 / cand)) * ((P.tiles_n + cand - 1) / cand);
    int dist = 0;
    for (int v = cand; v < aim; v *= 2) ++dist;
    for (int v = aim; v < cand; v *= 2) ++dist;
    if (best < 0 || padded < best || (padded == best && dist < best_dist)) {
      best = padded;
      best_dist = dist;
      xn = cand;
    }
  }
  P.xn = xn;
  P.xm = 8 / xn;
  P.ptm = (P.tiles_m + P.xm - 1) / P.xm;
  P.ptn = (P.tiles_n + P.xn - 1) / P.xn;
  P.walk_m = walk_along_m(d, P.xm, P.xn);
  // split-K (library-chosen, weight-gradient shapes): through workspace slabs when there is a workspace, fp32 atomics else
  int splitk = d.splitk > 1 ? d.splitk : 1;
  bool ws_split = false;
  if (d.splitk == 0 && d.nbatch2 <= 1 && !d.bias && !d.accumulate && !d.act && !d.aux_out && !d.aux_in && !d.resid && d.scale_cols <= 0) {
    const int64_t wgs = (int64_t)8 * P.ptm * P.ptn * d.nbatch;
    const int64_t its = (int64_t)d.kbatch * P.ktiles;
    int64_t want = 512 / wgs;
    if (want > its / 8) want = its / 8;
    const int64_t slab_bytes = d.M * d.N * d.nbatch * 4;
    const bool use_ws = d.workspace && (reinterpret_cast<uintptr_t>(d.workspace) & 15) == 0 && 2 * slab_bytes <= d.workspace_bytes;
    if (use_ws) {
      if (want > 8) want = 8;
      if (want > d.workspace_bytes / slab_bytes) want = d.workspace_bytes / slab_bytes;
      ws_split = want > 1;
    } else if (want > 4) want = 4;
    if (!use_ws && tmi_deterministic() && want > 2) want = 2;  // (two atomic contributions per element commute: launch_cfg)
    splitk = want < 1 ? 1 : (int)want;
  }
  dim3 grid((unsigned)(8 * P.ptm * P.ptn), (unsigned)splitk, (unsigned)(d.nbatch * (d.nbatch2 > 1 ? d.nbatch2 : 1)));
  if (ws_split) return launch_with_slabs(P, splitk, stream, [&](const FastParams& Q) {
    hipLaunchKernelGGL(kern, grid, dim3(256), 65536, stream, Q);
  });
  hipLaunchKernelGGL(kern, grid, dim3(256), 65536, stream, P);
  return tmi_check_launch("tmi_gemm(f32)");
}

}  // namespace

// diagnostics: copies the ABL == 8 phase counters (host-synchronous)
extern "C" int tmi_debug_gemm_stamps(unsigned long long* out5) {
  static const int epi = [] { const char* e = getenv("TMI_GEMM_DBG"); return e ? (atoi(e) & 32) : 0; }();
  if (epi) {  // (TMI_GEMM_DBG & 32: the stamps inside wide_epilogue instead; reading them re-arms the probe)
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, got[8];
    if (hipMemcpyFromSymbol(got, HIP_SYMBOL(g_epi_stamps), sizeof(got)) != hipSuccess) return -2;
    for (int i = 0; i < 5; ++i) out5[i] = got[i];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_epi_stamps), z, sizeof(z)) == hipSuccess ? 0 : -2;
  }
  return hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_gemm_stamps), 5 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}

// Returns 1 and sets *rc if the fast path took the GEMM, 0 if the generic kernel must run.
int tmi_gemm_fast_try(const tmi_gemm_desc& d, hipStream_t stream, int* rc) {
  if (d.in_dtype == TMI_F32) {  // the parity mode's GEMMs: fp32 in, fp32 out, whole 32-float slabs, 16-byte aligned operands
    static const int off = [] { const char* e = getenv("TMI_GEMM_F32_FAST"); return e && atoi(e) == 0; }();
    if (off || d.out_dtype != TMI_F32 || d.K % 4 != 0 || d.K < 32 || !al16(d.A) || !al16(d.B) || d.a_sb % 4 || d.b_sb % 4 || d.a_skb % 4 ||
        d.b_skb % 4 || d.a_sb2 % 4 || d.b_sb2 % 4 || d.M < 32 || d.N < 32 || (d.K % 32 != 0 && d.kbatch > 1 && d.splitk != 1))
      return 0;
    const bool a_kc = d.a_sk == 1 && d.a_sm % 4 == 0, b_kc = d.b_sk == 1 && d.b_sn % 4 == 0;
    const bool a_ks = d.a_sm == 1 && d.a_sk % 4 == 0 && (d.M % 4 == 0 || d.a_sk >= (d.M + 3) / 4 * 4);
    const bool b_ks = d.b_sn == 1 && d.b_sk % 4 == 0 && (d.N % 4 == 0 || d.b_sk >= (d.N + 3) / 4 * 4);
    if (!(a_kc || a_ks) || !(b_kc || b_ks)) return 0;
    if (!a_kc && b_kc) return 0;  // (k-strided A with k-contiguous B: not on the step's path)
    if (!a_kc) *rc = launch_f32<true, true>(d, stream);
    else if (!b_kc) *rc = launch_f32<false, true>(d, stream);
    else *rc = launch_f32<false, false>(d, stream);
    return 1;
  }
  if (d.in_dtype != TMI_BF16) return 0;
  if (!al16(d.A) || !al16(d.B) || d.a_sb % 8 || d.b_sb % 8 || d.a_skb % 8 || d.b_skb % 8) return 0;
  // operand modes: KC = k-contiguous rows (16-byte aligned row starts), KS = k-strided with
  // contiguous columns whose 16-byte chunks are all readable (row stride >= round_up(cols, 8))
  const bool a_kc = d.a_sk == 1 && d.a_sm % 8 == 0;
  const bool b_kc = d.b_sk == 1 && d.b_sn % 8 == 0;
  // (overlapping rows — the Conv1D window trick — are fine when the column count itself is whole chunks)
  const bool a_ks = d.a_sm == 1 && d.a_sk % 8 == 0 && (d.M % 8 == 0 || d.a_sk >= rup8(d.M)) && d.M >= 8;
  const bool b_ks = d.b_sn == 1 && d.b_sk % 8 == 0 && (d.N % 8 == 0 || d.b_sk >= rup8(d.N)) && d.N >= 8;
  if (!(a_kc || a_ks) || !(b_kc || b_ks)) return 0;
  const bool A_KS = !a_kc, B_KS = !b_kc;
  // a KC image cannot mask a partial K tile (its DMA would read past the row): K tails only in (KS, KS)
  if (d.K % 64 != 0 && (!A_KS || !B_KS)) return 0;
  if (A_KS && B_KS) *rc = launch_out<true, true>(d, stream);
  else if (A_KS) *rc = launch_out<true, false>(d, stream);
  else if (B_KS) *rc = launch_out<false, true>(d, stream);
  else *rc = launch_out<false, false>(d, stream);
  return 1;
}
