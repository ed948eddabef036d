// tmi_gemm: strided, batched GEMM with fused epilogue on the gfx950 matrix cores.
// Replaces tf.keras.layers.Dense / Conv1D forward and the matmuls of their gradients
// (speech_jobs/whisper_dist.py:89-92,194-197,311-312,545 and tape.gradient at :833).
//
// Structure (v1, correctness-first): 128x128 output tile per 256-thread workgroup
// (4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 blocks), K staged through LDS in
// 128-byte slabs per row (64 bf16 / 32 fp32), register-staged so the next slab's global
// loads are in flight under the MFMAs.  Either operand may be k-contiguous (vector loads
// along k) or row-contiguous (vector loads along m/n, transposed while written to LDS), so
// forward (X·W), dgrad (dY·Wᵀ) and wgrad (Xᵀ·dY) all map onto this one kernel.
#include "tmi_common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;           // bytes of K per LDS row
constexpr int LDSROW = ROWB + 16;   // padded LDS row stride (bytes)
constexpr int TILE_BYTES = 128 * LDSROW;

enum { MODE_KVEC = 0, MODE_RVEC = 1, MODE_SCALAR = 2 };

struct GemmParams {
  tmi_gemm_desc d;
  int a_mode, b_mode;
  int tiles_m, tiles_n;
  int ktiles;  // per kbatch
};

// One operand tile = 128 rows x ROWB bytes of k.  `src(row, k)` = base[row*s_row + k*s_k].
template <typename T>
struct Stager {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = ROWB / sizeof(T);
  static constexpr int CH = 128 / VEC;     // row-chunks per k (RVEC)
  static constexpr int KSTEP = 256 / CH;   // k rows covered by one pass of 256 threads (RVEC)
  u32x4 r[4];

  __device__ __forceinline__ T ld1(const T* base, int64_t s_row, int64_t s_k, int64_t row,
                                   int64_t nrows, int64_t k, int64_t kend) const {
    return (row < nrows && k < kend) ? base[row * s_row + k * s_k] : from_f32<T>(0.f);
  }

  __device__ __forceinline__ void fetch(const T* base, int64_t s_row, int64_t s_k, int mode,
                                        int64_t row0, int64_t nrows, int64_t k0, int64_t kend) {
    const int t = threadIdx.x;
    if (mode == MODE_RVEC) {
      const int rc = t % CH;
      const int64_t row = row0 + (int64_t)rc * VEC;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + t / CH + KSTEP * i;
        if (k < kend && row + VEC <= nrows) {
          r[i] = *reinterpret_cast<const u32x4*>(base + k * s_k + row);
        } else {
          alignas(16) T tmp[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) tmp[j] = ld1(base, 1, s_k, row + j, nrows, k, kend);
          r[i] = *reinterpret_cast<u32x4*>(tmp);
        }
      }
    } else {
      const int c = t & 7;
      const int64_t k = k0 + (int64_t)c * VEC;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + (t >> 3) + 32 * i;
        if (mode == MODE_KVEC && row < nrows && k + VEC <= kend) {
          r[i] = *reinterpret_cast<const u32x4*>(base + row * s_row + k);
        } else {
          alignas(16) T tmp[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) tmp[j] = ld1(base, s_row, s_k, row, nrows, k + j, kend);
          r[i] = *reinterpret_cast<u32x4*>(tmp);
        }
      }
    }
  }

  __device__ __forceinline__ void commit(char* lds, int mode) const {
    const int t = threadIdx.x;
    if (mode == MODE_RVEC) {
      const int rc = t % CH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = t / CH + KSTEP * i;
        const T* e = reinterpret_cast<const T*>(&r[i]);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          *reinterpret_cast<T*>(lds + (rc * VEC + j) * LDSROW + kk * (int)sizeof(T)) = e[j];
      }
    } else {
      const int c = t & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (t >> 3) + 32 * i;
        *reinterpret_cast<u32x4*>(lds + row * LDSROW + c * 16) = r[i];
      }
    }
  }
};

template <typename T>
__device__ __forceinline__ void mma_tile(const char* As, const char* Bs, int wr, int wc, int lane,
                                         f32x16 (&acc)[2][2]);

template <>
__device__ __forceinline__ void mma_tile<bf16_t>(const char* As, const char* Bs, int wr, int wc,
                                                 int lane, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const bf16x8*>(As + (wr * 64 + i * 32 + r) * LDSROW + kk * 32 + h * 16);
      b[i] = *reinterpret_cast<const bf16x8*>(Bs + (wc * 64 + i * 32 + r) * LDSROW + kk * 32 + h * 16);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

template <>
__device__ __forceinline__ void mma_tile<float>(const char* As, const char* Bs, int wr, int wc,
                                                int lane, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    float a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const float*>(As + (wr * 64 + i * 32 + r) * LDSROW + (ks * 2 + h) * 4);
      b[i] = *reinterpret_cast<const float*>(Bs + (wc * 64 + i * 32 + r) * LDSROW + (ks * 2 + h) * 4);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

template <typename T, typename TC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + TILE_BYTES;
  const tmi_gemm_desc& d = P.d;
  constexpr int BK = ROWB / sizeof(T);

  const int tile = blockIdx.x;
  const int tm = tile / P.tiles_n, tn = tile % P.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // (kb, ktile) iteration range of this split
  const int total_it = (int)d.kbatch * P.ktiles;
  const int nsplit = gridDim.y;
  const int per = (total_it + nsplit - 1) / nsplit;
  const int it0 = blockIdx.y * per;
  const int it1 = min(total_it, it0 + per);

  // two batch levels: z = b2 * nbatch + b1 (nbatch2 <= 1: b2 = 0)
  const int64_t b2 = d.nbatch2 > 1 ? bz / d.nbatch : 0, b1 = bz - b2 * d.nbatch;
  const T* Abase = reinterpret_cast<const T*>(d.A) + b1 * d.a_sb + b2 * d.a_sb2;
  const T* Bbase = reinterpret_cast<const T*>(d.B) + b1 * d.b_sb + b2 * d.b_sb2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Stager<T> sa, sb;
  if (it0 < it1) {
    {
      const int kb = it0 / P.ktiles, kt = it0 % P.ktiles;
      sa.fetch(Abase + kb * d.a_skb, d.a_sm, d.a_sk, P.a_mode, m0, d.M, (int64_t)kt * BK, d.K);
      sb.fetch(Bbase + kb * d.b_skb, d.b_sn, d.b_sk, P.b_mode, n0, d.N, (int64_t)kt * BK, d.K);
    }
    sa.commit(As, P.a_mode);
    sb.commit(Bs, P.b_mode);
    __syncthreads();
    for (int it = it0; it < it1; ++it) {
      const bool more = (it + 1 < it1);
      if (more) {
        const int kb = (it + 1) / P.ktiles, kt = (it + 1) % P.ktiles;
        sa.fetch(Abase + kb * d.a_skb, d.a_sm, d.a_sk, P.a_mode, m0, d.M, (int64_t)kt * BK, d.K);
        sb.fetch(Bbase + kb * d.b_skb, d.b_sn, d.b_sk, P.b_mode, n0, d.N, (int64_t)kt * BK, d.K);
      }
      mma_tile<T>(As, Bs, wr, wc, lane, acc);
      __syncthreads();
      if (more) {
        sa.commit(As, P.a_mode);
        sb.commit(Bs, P.b_mode);
        __syncthreads();
      }
    }
  }

  if (b2 == 0) {
    gemm_epilogue<TC>(d, acc, m0, n0, b1, wr, wc, lane, gridDim.y > 1);
  } else {  // (outer batches carry no bias / aux / residual terms: only C moves)
    tmi_gemm_desc dd = d;
    dd.C = reinterpret_cast<TC*>(d.C) + b2 * d.c_sb2;
    gemm_epilogue<TC>(dd, acc, m0, n0, b1, wr, wc, lane, gridDim.y > 1);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int pick_mode(const void* p, int64_t s_row, int64_t s_k, int64_t sb, int64_t skb, int vec) {
  const bool strides_ok = (sb % vec == 0) && (skb % vec == 0) && al16(p);
  if (s_k == 1 && strides_ok && (s_row % vec == 0)) return MODE_KVEC;
  if (s_row == 1 && strides_ok && (s_k % vec == 0)) return MODE_RVEC;
  return MODE_SCALAR;
}

template <typename T, typename TC>
int launch(const tmi_gemm_desc& d, hipStream_t stream) {
  GemmParams P;
  P.d = d;
  constexpr int vec = 16 / sizeof(T);
  constexpr int BK = ROWB / sizeof(T);
  P.a_mode = pick_mode(d.A, d.a_sm, d.a_sk, d.a_sb, d.a_skb, vec);
  P.b_mode = pick_mode(d.B, d.b_sn, d.b_sk, d.b_sb, d.b_skb, vec);
  P.tiles_m = (int)((d.M + BM - 1) / BM);
  P.tiles_n = (int)((d.N + BN - 1) / BN);
  P.ktiles = (int)((d.K + BK - 1) / BK);
  int splitk = d.splitk > 1 ? d.splitk : 1;
  if (d.splitk == 0 && sizeof(TC) == 4 && !d.bias && !d.accumulate && !d.act && !d.aux_out && !d.aux_in && !d.resid &&
      d.scale_cols <= 0) {  // auto split-K for weight-gradient shapes, as on the fast path
    const int64_t tiles = (int64_t)P.tiles_m * P.tiles_n * d.nbatch * (d.nbatch2 > 1 ? d.nbatch2 : 1);
    const int64_t its = (int64_t)d.kbatch * P.ktiles;
    int64_t want = (512 + tiles - 1) / tiles;
    if (want > its / 4) want = its / 4;
    if (want > 64) want = 64;
    if (tmi_deterministic() && want > 2) want = 2;  // (two atomic contributions per element commute; more do not)
    splitk = want < 1 ? 1 : (int)want;
  }
  dim3 grid((unsigned)(P.tiles_m * P.tiles_n), (unsigned)splitk, (unsigned)(d.nbatch * (d.nbatch2 > 1 ? d.nbatch2 : 1)));
  hipLaunchKernelGGL((gemm_kernel<T, TC>), grid, dim3(256), 2 * TILE_BYTES, stream, P);
  return tmi_check_launch("tmi_gemm");
}

}  // namespace

int tmi_gemm_fast_try(const tmi_gemm_desc& d, hipStream_t stream, int* rc);  // gemm_fast.hip

static bool fast_disabled() {
  static const bool off = [] {
    const char* e = getenv("TMI_GEMM_GENERIC");
    return e && e[0] == '1';
  }();
  return off;
}

static int tmi_gemm_impl(const tmi_gemm_desc* dp, void* stream);
extern "C" int tmi_gemm(const tmi_gemm_desc* dp, void* stream) {
  if (tmi_plan_recording() && dp) {
    const tmi_gemm_desc c_ = *dp;
    tmi_plan_push([c_, stream]() -> int {
      tmi_gemm_desc e_ = c_;
      if (e_.dropout_p > 0.f) e_.dropout_seed += tmi_plan_seed_delta();
      return tmi_gemm(&e_, stream);
    });
  }
  tmi_plan_enter();
  const int rc_ = tmi_gemm_impl(dp, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_gemm_impl(const tmi_gemm_desc* dp, void* stream) {
  if (!dp) return TMI_ERR_INVALID;
  tmi_gemm_desc d = *dp;
  if (!d.A || !d.B || !d.C || d.M <= 0 || d.N <= 0 || d.K <= 0) {
    tmi_set_error("tmi_gemm: null operand or non-positive shape");
    return TMI_ERR_INVALID;
  }
  if (d.nbatch <= 0) d.nbatch = 1;
  if (d.kbatch <= 0) d.kbatch = 1;
  if (d.nbatch2 <= 0) d.nbatch2 = 1;
  if (d.nbatch * d.nbatch2 > 65535 || d.splitk > 65535) {
    tmi_set_error("tmi_gemm: nbatch/splitk exceed grid limits");
    return TMI_ERR_INVALID;
  }
  if (d.nbatch2 > 1 && (d.bias || d.act || d.aux_out || d.aux_in || d.resid || d.dropout_p > 0.f)) {
    tmi_set_error("tmi_gemm: an outer batch level (nbatch2) takes plain epilogues only");
    return TMI_ERR_INVALID;
  }
  if (d.splitk < 0) d.splitk = 1;
  if (d.splitk > 1) {
    if (d.out_dtype != TMI_F32 || d.bias || d.accumulate || d.act || d.aux_out || d.aux_in ||
        d.resid || d.scale_cols > 0) {
      tmi_set_error("tmi_gemm: splitk > 1 needs fp32 C and no epilogue terms");
      return TMI_ERR_INVALID;
    }
  }
  if (!(d.dropout_p >= 0.f && tmi_drop_ok(d.dropout_p)) || (d.dropout_p > 0.f && (d.nbatch != 1 || d.splitk != 1 || (d.N & 1) || d.N > TMI_DROP_MAX_COLS))) {
    tmi_set_error("tmi_gemm: epilogue dropout needs 0 <= p < 1, nbatch == 1, splitk == 1 and an even N <= 2^17");
    return TMI_ERR_INVALID;
  }
  if (d.act != 0 && d.act != 1) {
    tmi_set_error("tmi_gemm: unknown activation");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((d.nbatch2 == 1 || d.in_dtype == TMI_F32) && (d.in_dtype == TMI_BF16 || d.in_dtype == TMI_F32) &&
      (d.out_dtype == TMI_BF16 || d.out_dtype == TMI_F32) && !fast_disabled()) {
    int rc = 0;
    if (tmi_gemm_fast_try(d, s, &rc)) return rc;
  }
  if (d.in_dtype == TMI_F32 && d.out_dtype == TMI_F32) return launch<float, float>(d, s);
  if (d.in_dtype == TMI_BF16 && d.out_dtype == TMI_BF16) return launch<bf16_t, bf16_t>(d, s);
  if (d.in_dtype == TMI_BF16 && d.out_dtype == TMI_F32) return launch<bf16_t, float>(d, s);
  tmi_set_error("tmi_gemm: unsupported dtype pair");
  return TMI_ERR_UNSUPPORTED;
}
