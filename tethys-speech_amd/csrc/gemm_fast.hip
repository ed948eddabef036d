// Fast bf16 GEMM paths of tmi_gemm for the two layouts that carry the step's FLOPs:
//   NT  C[m][n] = sum_k A[m][k] * Bt[n][k]   both operands k-contiguous
//       (forward with the transposed weight shadow; dgrad with the natural Keras kernel)
//   TN  C[i][j] = sum_k A[k][i] * B[k][j]    both operands k-strided
//       (wgrad: Xᵀ·dY, the reduction index is the activation row)
// 128x128 output tile per 256-thread workgroup (waves 2x2, 64x64 each, MFMA 32x32x16 bf16),
// BK = 64.  Operand tiles are staged global -> LDS directly (global_load_lds_dwordx4, no
// VGPR round trip), double-buffered: the next tile's DMA is in flight under the current
// tile's MFMAs, one barrier per K-tile.  The LDS image is lane-linear as the DMA requires;
// the XOR swizzle that makes the fragment reads bank-conflict-free is applied to the per-lane
// SOURCE address and again on the read (same involution on both sides).
//   NT image: [128 rows][8 x 16 B chunks];  phys_chunk = chunk ^ ((row >> 1) & 7); fragments
//             by ds_read_b128.
//   TN image: [64 k-rows][16 x 16 B chunks]; phys_chunk = chunk ^ ((krow & 3) << 2); fragments
//             by ds_read_b64_tr_b16 (hardware transpose: a 4(k) x 16(m) block, column-major).
// Workgroup ids are remapped so that the tiles sharing one A row-panel run on one XCD (L2).
#include "tmi_common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int FT_BYTES = 16384;  // one operand tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

struct FastParams {
  tmi_gemm_desc d;
  int tiles_m, tiles_n, ktiles;
};

// bijective XCD-aware remap: consecutive new ids share an XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- NT staging: 128 rows x 64 k (128 B per row)
__device__ __forceinline__ void stage_nt(char* lds, const bf16_t* base, int64_t s_row, int64_t row0, int64_t nrows,
                                         int64_t k0, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * wave + i;
    const int r = 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;
    glds16(base + grow * s_row + k0 + c * 8, lds + j * 1024);
  }
}

__device__ __forceinline__ void mma_nt(const char* As, const char* Bs, int wr, int wc, int lane, f32x16 (&acc)[2][2]) {
  const int r = lane & 31, h = lane >> 5;
  const int sw = (r >> 1) & 7;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[2], b[2];
    const int off = ((2 * kk + h) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const bf16x8*>(As + (wr * 64 + i * 32 + r) * 128 + off);
      b[i] = *reinterpret_cast<const bf16x8*>(Bs + (wc * 64 + i * 32 + r) * 128 + off);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

// ---- TN staging: 64 k-rows x 128 cols (256 B per row)
__device__ __forceinline__ void stage_tn(char* lds, const bf16_t* base, int64_t s_k, int64_t col0, int64_t ncols,
                                         int64_t k0, int64_t kend, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * wave + i;
    const int kr = 4 * j + (lane >> 4);
    const int c = (lane & 15) ^ ((kr & 3) << 2);
    int64_t gk = k0 + kr;
    gk = gk < kend ? gk : kend - 1;
    int64_t gc = col0 + c * 8;
    gc = gc + 8 <= ncols ? gc : ncols - 8;
    glds16(base + gk * s_k + gc, lds + j * 1024);
  }
}

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int col_base, int kk, int lane) {
  // 32x32x16 operand fragment for k-step kk from a [k][col] image: lane (r, h) gets
  // T[16kk + 8h + j][col_base + r], j = 0..7, as two transposed 4x16 block reads.
  const int g = lane >> 4, i = lane & 15;
  const int h = g >> 1;
  const int q = i >> 2, p = i & 3;
  const int col = col_base + 16 * (g & 1) + 4 * p;  // first of this lane's 4 address columns
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

__device__ __forceinline__ void mma_tn(const char* As, const char* Bs, int wr, int wc, int lane, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = tr_frag(As, wr * 64 + i * 32, kk, lane);
      b[i] = tr_frag(Bs, wc * 64 + i * 32, kk, lane);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
}

template <typename TC, bool TN>
__global__ __launch_bounds__(256) void gemm_fast_kernel(const FastParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile]
  const tmi_gemm_desc& d = P.d;
  const int nwg = P.tiles_m * P.tiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / P.tiles_n, tn = tile % P.tiles_n;
  const int64_t m0 = (int64_t)tm * 128, n0 = (int64_t)tn * 128;
  const int64_t bz = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int total_it = (int)d.kbatch * P.ktiles;
  const int nsplit = gridDim.y;
  const int per = (total_it + nsplit - 1) / nsplit;
  const int it0 = blockIdx.y * per;
  const int it1 = min(total_it, it0 + per);

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(d.A) + bz * d.a_sb;
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(d.B) + bz * d.b_sb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto stage = [&](int it, int buf) {
    const int kb = it / P.ktiles, kt = it % P.ktiles;
    char* As = smem + buf * 2 * FT_BYTES;
    char* Bs = As + FT_BYTES;
    if constexpr (TN) {
      stage_tn(As, Abase + kb * d.a_skb, d.a_sk, m0, d.M, (int64_t)kt * 64, d.K, wave, lane);
      stage_tn(Bs, Bbase + kb * d.b_skb, d.b_sk, n0, d.N, (int64_t)kt * 64, d.K, wave, lane);
    } else {
      stage_nt(As, Abase + kb * d.a_skb, d.a_sm, m0, d.M, (int64_t)kt * 64, wave, lane);
      stage_nt(Bs, Bbase + kb * d.b_skb, d.b_sn, n0, d.N, (int64_t)kt * 64, wave, lane);
    }
  };
  // TN only: rows k >= K of the last tile were loaded from a clamped row; zero them
  auto zero_tail = [&](int it, int buf) {
    if constexpr (TN) {
      const int kt = it % P.ktiles;
      const int kvalid = (int)min((int64_t)64, d.K - (int64_t)kt * 64);
      if (kvalid < 64) {
        char* As = smem + buf * 2 * FT_BYTES;
        for (int idx = threadIdx.x; idx < (64 - kvalid) * 16; idx += 256) {
          const int off = kvalid * 256 + idx * 16;
          *reinterpret_cast<u32x4*>(As + off) = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(As + FT_BYTES + off) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
      }
    }
  };

  if (it0 < it1) {
    stage(it0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int it = it0; it < it1; ++it) {
      if (it + 1 < it1) stage(it + 1, cur ^ 1);
      zero_tail(it, cur);
      const char* As = smem + cur * 2 * FT_BYTES;
      const char* Bs = As + FT_BYTES;
      if constexpr (TN)
        mma_tn(As, Bs, wr, wc, lane, acc);
      else
        mma_nt(As, Bs, wr, wc, lane, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  gemm_epilogue<TC>(d, acc, m0, n0, bz, wr, wc, lane, nsplit > 1);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename TC, bool TN>
int launch_fast(const tmi_gemm_desc& d, hipStream_t stream) {
  FastParams P;
  P.d = d;
  P.tiles_m = (int)((d.M + 127) / 128);
  P.tiles_n = (int)((d.N + 127) / 128);
  P.ktiles = (int)((d.K + 63) / 64);
  const int splitk = d.splitk > 1 ? d.splitk : 1;
  dim3 grid((unsigned)(P.tiles_m * P.tiles_n), (unsigned)splitk, (unsigned)d.nbatch);
  hipLaunchKernelGGL((gemm_fast_kernel<TC, TN>), grid, dim3(256), 4 * FT_BYTES, stream, P);
  return tmi_check_launch("tmi_gemm(fast)");
}

}  // namespace

// Returns 1 and sets *rc if a fast path took the GEMM, 0 if the generic kernel must run.
int tmi_gemm_fast_try(const tmi_gemm_desc& d, hipStream_t stream, int* rc) {
  if (d.in_dtype != TMI_BF16) return 0;
  if (!al16(d.A) || !al16(d.B) || d.a_sb % 8 || d.b_sb % 8 || d.a_skb % 8 || d.b_skb % 8) return 0;
  const bool f32out = d.out_dtype == TMI_F32;
  // NT: both k-contiguous, K a multiple of the 64-wide tile, rows 16-byte aligned
  if (d.a_sk == 1 && d.b_sk == 1 && d.K % 64 == 0 && d.a_sm % 8 == 0 && d.b_sn % 8 == 0) {
    *rc = f32out ? launch_fast<float, false>(d, stream) : launch_fast<bf16_t, false>(d, stream);
    return 1;
  }
  // TN: both k-strided with contiguous columns; whole 16-byte column chunks
  if (d.a_sm == 1 && d.b_sn == 1 && d.a_sk % 8 == 0 && d.b_sk % 8 == 0 && d.M % 8 == 0 && d.N % 8 == 0 &&
      d.M >= 8 && d.N >= 8) {
    *rc = f32out ? launch_fast<float, true>(d, stream) : launch_fast<bf16_t, true>(d, stream);
    return 1;
  }
  return 0;
}
