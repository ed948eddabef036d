// LayerNorm forward / backward over the last axis (HBM-bound; one wavefront per row,
// 16-byte loads, wavefront-shuffle reductions).  Replaces
// tf.keras.layers.LayerNormalization(epsilon=1e-5) at speech_jobs/whisper_dist.py:214,216,
// 245,249,253,322,392 and its gradient.
// Algorithmic bytes: fwd 2*n*s (+8 B/row stats), bwd 3*n*s (+ dgamma/dbeta partials).
#include "tmi_common.h"
#include <type_traits>

namespace {

constexpr int LN_MAX_C = 2048;

// A lane holds NCH 16-byte chunks of its row: chunk j covers columns (j*64 + lane)*VEC .. +VEC.
// NCH is a template parameter so that a 768-wide row costs 2 chunks of registers, not the 4 (bf16)
// or 8 (fp32) a 2048-wide one needs: the kernels are HBM-bound and live on waves in flight.
template <typename T, int NCH_>
struct RowIO {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int NCH = NCH_;
  static constexpr int E = NCH * VEC;
  __device__ static __forceinline__ void load_raw(const T* row, int C, int lane, u32x4 (&raw)[NCH]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c0 = (j * 64 + lane) * VEC;
      if (c0 < C) raw[j] = *reinterpret_cast<const u32x4*>(row + c0);
      else raw[j] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  __device__ static __forceinline__ void unpack(const u32x4 (&raw)[NCH], float (&v)[E]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const T* e = reinterpret_cast<const T*>(&raw[j]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[j * VEC + i] = to_f32(e[i]);
    }
  }
  __device__ static __forceinline__ void store(T* row, int C, int lane, const float (&v)[E]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c0 = (j * 64 + lane) * VEC;
      if (c0 < C) {
        alignas(16) T e[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) e[i] = from_f32<T>(v[j * VEC + i]);
        *reinterpret_cast<u32x4*>(row + c0) = *reinterpret_cast<const u32x4*>(e);
      }
    }
  }
  __device__ static __forceinline__ void load_f32vec(const float* p, int C, int lane, float (&v)[E]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c0 = (j * 64 + lane) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[j * VEC + i] = (c0 < C) ? p[c0 + i] : 0.f;
    }
  }
};

// The flat dropout generator (tmi_keep: stream 0, row, column) applied to a lane's chunks of one row
template <typename IO>
__device__ __forceinline__ void drop_row(float (&v)[IO::E], uint32_t key, uint32_t row, int lane, uint32_t thr, float scale) {
  const tmi_rowkey rk = tmi_row_key(key, row);
#pragma unroll
  for (int j = 0; j < IO::NCH; ++j) {
    const uint32_t cp0 = (uint32_t)(((j * 64 + lane) * IO::VEC) >> 1);
#pragma unroll
    for (int i = 0; i < IO::VEC; i += 2) {
      const uint32_t hh = tmi_pair_hash(rk, cp0 + (i >> 1));
      const int e = j * IO::VEC + i;
      v[e] = (hh & 0xffffu) >= thr ? v[e] * scale : 0.f;
      v[e + 1] = (hh >> 16) >= thr ? v[e + 1] * scale : 0.f;
    }
  }
}

// Each wave walks rows blockIdx*4 + wave, + gridDim*4, ... with the next row's 16-byte loads
// issued before the current row is reduced.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int C, float eps, uint32_t drop_key, uint32_t drop_thr,
                                                     float drop_scale) {
  using IO = RowIO<T, NCH>;
  constexpr int E = IO::E;
  const int lane = threadIdx.x & 63;
  const int64_t step = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float g[E], b[E];
  IO::load_f32vec(gamma, C, lane, g);
  IO::load_f32vec(beta, C, lane, b);
  const float invC = 1.0f / (float)C;
  u32x4 nxt[NCH];
  IO::load_raw(x + row * C, C, lane, nxt);
  for (; row < rows; row += step) {
    float v[E];
    IO::unpack(nxt, v);
    if (row + step < rows) IO::load_raw(x + (row + step) * C, C, lane, nxt);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) s += v[i];
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool in = ((j * 64 + lane) * IO::VEC) < C;
#pragma unroll
      for (int i = 0; i < IO::VEC; ++i) {
        const float dlt = in ? v[j * IO::VEC + i] - mu : 0.f;
        q += dlt * dlt;
      }
    }
    const float var = wave_sum(q) * invC;
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = (v[i] - mu) * rs * g[i] + b[i];
    if (drop_thr) drop_row<IO>(v, drop_key, (uint32_t)row, lane, drop_thr, drop_scale);  // Dropout(LayerNorm(x)) in one pass (V:296, V:560, V:779)
    IO::store(y + row * C, C, lane, v);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

// dgamma/dbeta partials are folded over the block's 8 waves through LDS slabs (LDS float atomics
// measured 3x slower).  With a workspace (`part` != nullptr) every block then STORES its row of NRED * C sums to
// part[block][NRED][C] and ln_fold_kernel, launched right behind on the same stream, adds the rows in block order to the
// fp32 gradients: no atomics, bit-reproducible, and the cost does not depend on what runs beside it.  (Round 2 measured
// the atomic form at 0.316 ms/step alone and 0.748 ms/step with the weight-gradient stream saturating L2: every block
// adds to the same 2-3 * C addresses and same-address atomics serialise at the memory side.)  Without a workspace the
// atomic form remains: one global atomic per column per block, the column walk started at a different offset per block.
constexpr int LNB_WAVES = 8;

// out_k[c] += sum over b of part[b][k][c].  A workgroup owns 64 consecutive (k, c) entries; its 16 waves each take the
// rows b = w, w + 16, ... (at most 16 loads per lane at the default 256 rows, all in flight together: the first version
// of this fold, one thread per entry walking all rows, was a latency chain of 64 dependent round trips = 12 us per
// call), then wave 0 adds the 16 wave sums in wave order: a fixed association, so the result is a function of the
// partials alone.
constexpr int LNF_WAVES = 16;
__global__ __launch_bounds__(64 * LNF_WAVES) void ln_fold_kernel(const float* __restrict__ part, int nblocks, int nred, int C,
                                                                 float* __restrict__ o0, float* __restrict__ o1,
                                                                 float* __restrict__ o2) {
  __shared__ float red[LNF_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int64_t stride = (int64_t)nred * C;
  float acc = 0.f;
  if (i < nred * C) {
    const float* p = part + i;
#pragma unroll 8
    for (int b = w; b < nblocks; b += LNF_WAVES) acc += p[(int64_t)b * stride];
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && i < nred * C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < LNF_WAVES; ++j) t += red[j][lane];
    const int k = i / C, c = i - k * C;
    float* o = k == 0 ? o0 : k == 1 ? o1 : o2;
    o[c] += t;
  }
}

template <typename T, int NCH, bool EMIT = false>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C,
                                                     int accumulate_dx, float* __restrict__ colsum, T* __restrict__ masked,
                                                     uint32_t drop_key, uint32_t drop_thr, float drop_scale,
                                                     float* __restrict__ part, uint32_t dy_key, uint32_t dy_thr, float dy_scale) {
  // EMIT: the residual-stream gradient this kernel writes (dx) is the dy of the Dense layer below it, whose bias
  // gradient is its column sum - and, where that layer's output went through Dropout (W:205, V:396, V:431), the dy is
  // the MASKED dx.  Both come out of this pass: colsum[c] += sum_rows (masked ? mask*dx : dx), masked[row][c] = mask*dx,
  // instead of a dropout kernel and a column-sum kernel re-reading dx.
  using IO = RowIO<T, NCH>;
  constexpr int E = IO::E;
  constexpr int NRED = EMIT ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [8 waves][NRED][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[E], dg[E], db[E], dc[EMIT ? E : 1];
  IO::load_f32vec(gamma, C, lane, g);
#pragma unroll
  for (int i = 0; i < E; ++i) dg[i] = db[i] = 0.f;
  if constexpr (EMIT) {
#pragma unroll
    for (int i = 0; i < E; ++i) dc[i] = 0.f;
  }
  const float invC = 1.0f / (float)C;
  const int64_t step = (int64_t)gridDim.x * LNB_WAVES;
  int64_t row = (int64_t)blockIdx.x * LNB_WAVES + wave;
  u32x4 nx[NCH], ndy[NCH], nold[NCH];
  float nmu = 0.f, nrs = 0.f;
  if (row < rows) {
    IO::load_raw(x + row * C, C, lane, nx);
    IO::load_raw(dy + row * C, C, lane, ndy);
    if (accumulate_dx) IO::load_raw(dx + row * C, C, lane, nold);
    nmu = mean[row];
    nrs = rstd[row];
  }
  for (; row < rows; row += step) {
    // Rows of three chunks and more (C > 1024 in bf16: Whisper-large's 1280): the old dx stays in its raw registers until it
    // is added (its prefetch for the next row is issued after that, below) - unpacked at the top it is E more live floats,
    // and the emitting form spilled 33 registers to scratch (12000 x 1280: 50.6 us with the masked copy).
    constexpr bool LATE_OLD = NCH >= 3;
    float xv[E], dv[E], old[LATE_OLD ? 1 : E];
    IO::unpack(nx, xv);
    IO::unpack(ndy, dv);
    if constexpr (!LATE_OLD) {
      if (accumulate_dx) IO::unpack(nold, old);
    }
    // (the LayerNorm's output went through Dropout in the forward: its gradient is the masked, rescaled dy)
    if (dy_thr) drop_row<IO>(dv, dy_key, (uint32_t)row, lane, dy_thr, dy_scale);
    const float mu = nmu, rs = nrs;
    const int64_t nr = row + step;
    if (nr < rows) {
      IO::load_raw(x + nr * C, C, lane, nx);
      IO::load_raw(dy + nr * C, C, lane, ndy);
      if constexpr (!LATE_OLD) {
        if (accumulate_dx) IO::load_raw(dx + nr * C, C, lane, nold);
      }
      nmu = mean[nr];
      nrs = rstd[nr];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool in = ((j * 64 + lane) * IO::VEC) < C;
#pragma unroll
      for (int i = 0; i < IO::VEC; ++i) {
        const int e = j * IO::VEC + i;
        const float xh = in ? (xv[e] - mu) * rs : 0.f;
        const float gg = dv[e] * g[e];
        xv[e] = xh;
        s1 += gg;
        s2 += gg * xh;
        dg[e] += dv[e] * xh;
        db[e] += dv[e];
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const T* oe = reinterpret_cast<const T*>(&nold[j]);
#pragma unroll
      for (int i = 0; i < IO::VEC; ++i) {
        const int e = j * IO::VEC + i;
        float r = rs * (dv[e] * g[e] - s1 - xv[e] * s2);
        if (accumulate_dx) r += LATE_OLD ? to_f32(oe[i]) : old[e];
        dv[e] = r;
      }
    }
    IO::store(dx + row * C, C, lane, dv);
    if constexpr (LATE_OLD) {
      if (accumulate_dx && nr < rows) IO::load_raw(dx + nr * C, C, lane, nold);
    }
    if constexpr (EMIT) {
      if (masked && drop_thr) {
        const tmi_rowkey rk = tmi_row_key(drop_key, (uint32_t)row);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c0 = (j * 64 + lane) * IO::VEC;
          const uint32_t cp0 = (uint32_t)(c0 >> 1);
#pragma unroll
          for (int i = 0; i < IO::VEC; i += 2) {
            const uint32_t hh = tmi_pair_hash(rk, cp0 + (i >> 1));
            const int e = j * IO::VEC + i;
            dv[e] = (hh & 0xffffu) >= drop_thr ? dv[e] * drop_scale : 0.f;
            dv[e + 1] = (hh >> 16) >= drop_thr ? dv[e + 1] * drop_scale : 0.f;
          }
        }
      }
      // (drop_thr == 0 with a `masked` buffer: a plain second copy of dx - a snapshot for a weight gradient that is
      // launched after this buffer has been overwritten, tmi_layernorm_bwd_emit)
      if (masked) IO::store(masked + row * C, C, lane, dv);
#pragma unroll
      for (int e = 0; e < E; ++e) dc[e] += dv[e];
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c0 = (j * 64 + lane) * IO::VEC;
    if (c0 < C) {
#pragma unroll
      for (int i = 0; i < IO::VEC; ++i) {
        red[(wave * NRED + 0) * C + c0 + i] = dg[j * IO::VEC + i];
        red[(wave * NRED + 1) * C + c0 + i] = db[j * IO::VEC + i];
        if constexpr (EMIT) red[(wave * NRED + 2) * C + c0 + i] = dc[j * IO::VEC + i];
      }
    }
  }
  __syncthreads();
  if (part) {  // this block's row of the partial table: [NRED][C], plain coalesced stores
    float* prow = part + (int64_t)blockIdx.x * NRED * C;
    for (int i = threadIdx.x; i < NRED * C; i += 64 * LNB_WAVES) {
      const int k = i / C, c = i - k * C;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) a += red[(w * NRED + k) * C + c];
      prow[i] = a;
    }
    return;
  }
  const int start = (int)((blockIdx.x * 64u) % (unsigned)C);
  for (int i = threadIdx.x; i < C; i += 64 * LNB_WAVES) {
    int c = start + i;
    c = c >= C ? c - C : c;
    float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) {
      a += red[(w * NRED + 0) * C + c];
      b += red[(w * NRED + 1) * C + c];
      if constexpr (EMIT) cc += red[(w * NRED + 2) * C + c];
    }
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, b);
    if constexpr (EMIT) atomicAdd(colsum + c, cc);
  }
}

// chunk count for a row of C elements: smallest instantiated NCH that covers it
template <typename T>
inline int ln_nch(int64_t C) {
  const int vec = 16 / (int)sizeof(T);
  const int need = (int)((C + 64 * vec - 1) / (64 * vec));
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : 8;
}
template <typename T, typename F>
inline void ln_dispatch(int64_t C, F&& f) {
  switch (ln_nch<T>(C)) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    default: f(std::integral_constant<int, 8>{}); break;
  }
}

// Column sums (bias gradients).  A block owns 64 * VEC columns and a strided set of rows; its 8 waves
// keep four 16-byte row loads in flight each, fold through LDS and add to `out` with one atomic per
// column.  All blocks of a column group hit the same addresses and same-address atomics serialise in
// L2 (~20 ns each), so the grid is ~one block per CU rather than thousands of small ones.
constexpr int CS_WAVES = 8;
template <typename T>
__global__ __launch_bounds__(64 * CS_WAVES) void colsum_kernel(const T* __restrict__ dy, int64_t ld, float* __restrict__ out,
                                                               int64_t rows, int64_t N, int64_t dy_sb, int64_t out_sb) {
  constexpr int VEC = 16 / sizeof(T);
  dy += (int64_t)blockIdx.z * dy_sb;    // batch (tmi_colsum_batched): one matrix and one output vector per z
  out += (int64_t)blockIdx.z * out_sb;
  __shared__ float red[CS_WAVES][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (c0 < N) {
    const int64_t step = (int64_t)gridDim.y * CS_WAVES;
    int64_t row = (int64_t)blockIdx.y * CS_WAVES + wave;
    const T* p = dy + row * ld + c0;
    for (; row + 3 * step < rows; row += 4 * step, p += 4 * step * ld) {
      u32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4*>(p + u * step * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T* e = reinterpret_cast<const T*>(&raw[u]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += to_f32(e[i]);
      }
    }
    for (; row < rows; row += step, p += step * ld) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(p);
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += to_f32(e[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[wave][lane * VEC + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * VEC; c += 64 * CS_WAVES) {
    const int64_t n = (int64_t)blockIdx.x * 64 * VEC + c;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; ++w) a += red[w][c];
    if (n < N) atomicAdd(out + n, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u,
                                                       T* __restrict__ dx, int64_t nvec, int64_t dy_sb, int64_t u_sb,
                                                       int64_t dx_sb) {
  constexpr int VEC = 16 / sizeof(T);
  dy += (int64_t)blockIdx.y * dy_sb;  // batch (blockIdx.y) strides in elements, multiples of VEC
  u += (int64_t)blockIdx.y * u_sb;
  dx += (int64_t)blockIdx.y * dx_sb;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4 a = reinterpret_cast<const u32x4*>(dy)[i];
    const u32x4 b = reinterpret_cast<const u32x4*>(u)[i];
    const T* ea = reinterpret_cast<const T*>(&a);
    const T* eb = reinterpret_cast<const T*>(&b);
    alignas(16) T o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(to_f32(ea[j]) * gelu_grad_t<T>(to_f32(eb[j])));
    reinterpret_cast<u32x4*>(dx)[i] = *reinterpret_cast<const u32x4*>(o);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                         int64_t C, float eps, float dropout_p, uint64_t dropout_seed, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0 || C > LN_MAX_C || C % vec ||
      !al16(x) || !al16(y) || !(dropout_p >= 0.f && tmi_drop_ok(dropout_p)) || (dropout_p > 0.f && (C & 1))) {
    tmi_set_error("tmi_layernorm_fwd: bad argument (C must be a multiple of 16 bytes and <= 2048; 0 <= dropout_p < 1)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // whole rows per wave, at most ~1024 blocks of 4 waves (measured best on MI355X: 11.4 us for [12000, 768] bf16)
  static const int64_t cap_f = [] { const char* e = getenv("TMI_LN_FWD_BLOCKS"); return e ? atoll(e) : 1024ll; }();
  const int64_t rpw = (rows + 4 * cap_f - 1) / (4 * cap_f);
  dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
  const uint32_t thr = tmi_drop_thr(dropout_p);
  const uint32_t key = tmi_stream_key(dropout_seed, 0u);
  const float scale = tmi_keep_scale(thr);
  if (dtype == TMI_BF16) {
    ln_dispatch<bf16_t>(C, [&](auto nch) {
      hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, decltype(nch)::value>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta,
                         (bf16_t*)y, mean, rstd, rows, (int)C, eps, key, thr, scale);
    });
  } else if (dtype == TMI_F32) {
    ln_dispatch<float>(C, [&](auto nch) {
      hipLaunchKernelGGL((ln_fwd_kernel<float, decltype(nch)::value>), grid, dim3(256), 0, s, (const float*)x, gamma, beta,
                         (float*)y, mean, rstd, rows, (int)C, eps, key, thr, scale);
    });
  } else {
    return TMI_ERR_UNSUPPORTED;
  }
  return tmi_check_launch("tmi_layernorm_fwd");
}

extern "C" int tmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int64_t rows, int64_t C, float eps, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, C, eps, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, C, eps, 0.f, 0, dtype, stream);
  tmi_plan_leave();
  return rc_;
}

extern "C" int tmi_layernorm_dropout_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                         int64_t rows, int64_t C, float eps, float dropout_p, uint64_t dropout_seed, int32_t dtype,
                                         void* stream) {
  if (tmi_plan_recording())
    tmi_plan_push([=]() -> int { return tmi_layernorm_dropout_fwd(x, gamma, beta, y, mean, rstd, rows, C, eps, dropout_p, dropout_seed + tmi_plan_seed_delta(), dtype, stream); });
  tmi_plan_enter();
  const int rc_ = ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, C, eps, dropout_p, dropout_seed, dtype, stream);
  tmi_plan_leave();
  return rc_;
}

static int64_t ln_bwd_blocks(int64_t rows) {
  static const int64_t cap_b = [] { const char* e = getenv("TMI_LN_BWD_BLOCKS"); return e ? atoll(e) : 256ll; }();
  int64_t rpw = (rows + LNB_WAVES * cap_b - 1) / (LNB_WAVES * cap_b);
  if (rpw < 2) rpw = 2;  // amortise the per-block dgamma/dbeta fold
  return (rows + LNB_WAVES * rpw - 1) / (LNB_WAVES * rpw);
}

extern "C" int64_t tmi_layernorm_bwd_workspace_bytes(int64_t rows, int64_t C, int32_t emit) {
  if (rows <= 0 || C <= 0) return 0;
  return ln_bwd_blocks(rows) * (emit ? 3 : 2) * C * (int64_t)sizeof(float);
}

static int ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                         float* dgamma, float* dbeta, int64_t rows, int64_t C, int32_t accumulate_dx, float* colsum, void* masked,
                         float dropout_p, uint64_t dropout_seed, int32_t dtype, void* stream, const char* what,
                         float* workspace, int64_t workspace_bytes, float dy_dropout_p = 0.f, uint64_t dy_dropout_seed = 0) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0 || C > LN_MAX_C ||
      C % vec || !al16(x) || !al16(dy) || !al16(dx) || (masked && (!colsum || !al16(masked) || (C & 1))) ||
      !(dropout_p >= 0.f && tmi_drop_ok(dropout_p)) || !(dy_dropout_p >= 0.f && tmi_drop_ok(dy_dropout_p)) ||
      (dy_dropout_p > 0.f && (C & 1))) {
    tmi_set_error("tmi_layernorm_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t blocks = ln_bwd_blocks(rows);
  const bool emit = colsum != nullptr;
  if (workspace && (workspace_bytes < tmi_layernorm_bwd_workspace_bytes(rows, C, emit) || !al16(workspace))) {
    tmi_set_error("tmi_layernorm_bwd: workspace smaller than tmi_layernorm_bwd_workspace_bytes() or misaligned");
    return TMI_ERR_INVALID;
  }
  const size_t lds = (size_t)LNB_WAVES * (emit ? 3 : 2) * C * sizeof(float);  // <= 128 KiB (192 with emission) at C = 2048
  if (emit && lds > 160 * 1024) {
    tmi_set_error("tmi_layernorm_bwd_emit: C too wide for the three-way fold");
    return TMI_ERR_INVALID;
  }
  const uint32_t thr = masked ? tmi_drop_thr(dropout_p) : 0u;
  const uint32_t key = tmi_stream_key(dropout_seed, 0u);
  const float scale = tmi_keep_scale(thr);
  const uint32_t dy_thr = tmi_drop_thr(dy_dropout_p), dy_key = tmi_stream_key(dy_dropout_seed, 0u);
  const float dy_scale = tmi_keep_scale(dy_thr);
  void* mk = masked;  // p == 0: the copy is dx itself (a snapshot for a deferred reader)
  auto go = [&](auto tag_t, auto nch, auto em) {
    using T = decltype(tag_t);
    constexpr int N = decltype(nch)::value;
    constexpr bool EM = decltype(em)::value;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_kernel<T, N, EM>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LNB_WAVES * (EM ? 3 : 2) * LN_MAX_C * 4 > 160 * 1024 ? 160 * 1024 : LNB_WAVES * (EM ? 3 : 2) * LN_MAX_C * 4);
    (void)attr;
    hipLaunchKernelGGL((ln_bwd_kernel<T, N, EM>), dim3((unsigned)blocks), dim3(64 * LNB_WAVES), lds, s, (const T*)dy, (const T*)x,
                       gamma, mean, rstd, (T*)dx, dgamma, dbeta, rows, (int)C, accumulate_dx, colsum, (T*)mk, key, thr, scale,
                       workspace, dy_key, dy_thr, dy_scale);
  };
  if (dtype == TMI_BF16) {
    ln_dispatch<bf16_t>(C, [&](auto nch) {
      if (emit) go(bf16_t{}, nch, std::true_type{});
      else go(bf16_t{}, nch, std::false_type{});
    });
  } else if (dtype == TMI_F32) {
    ln_dispatch<float>(C, [&](auto nch) {
      if (emit) go(float{}, nch, std::true_type{});
      else go(float{}, nch, std::false_type{});
    });
  } else {
    return TMI_ERR_UNSUPPORTED;
  }
  if (workspace) {
    int rc = tmi_check_launch(what);
    if (rc) return rc;
    const int nred = emit ? 3 : 2;
    hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)((nred * C + 63) / 64)), dim3(64 * LNF_WAVES), 0, s, workspace, (int)blocks,
                       nred, (int)C, dgamma, dbeta, colsum);
  }
  return tmi_check_launch(what);
}

static int tmi_layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                 int32_t accumulate_dx, float* workspace, int64_t workspace_bytes, int32_t dtype, void* stream);
extern "C" int tmi_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                 int32_t accumulate_dx, float* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, workspace, workspace_bytes, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, workspace, workspace_bytes, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                 int32_t accumulate_dx, float* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
  return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, nullptr, nullptr, 0.f, 0, dtype, stream,
                       "tmi_layernorm_bwd", workspace, workspace_bytes);
}

extern "C" int tmi_layernorm_dropout_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                         void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C, int32_t accumulate_dx,
                                         float dropout_p, uint64_t dropout_seed, float* workspace, int64_t workspace_bytes,
                                         int32_t dtype, void* stream) {
  if (tmi_plan_recording())
    tmi_plan_push([=]() -> int { return tmi_layernorm_dropout_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, dropout_p, dropout_seed + tmi_plan_seed_delta(), workspace, workspace_bytes, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, nullptr, nullptr, 0.f, 0, dtype,
                                stream, "tmi_layernorm_dropout_bwd", workspace, workspace_bytes, dropout_p, dropout_seed);
  tmi_plan_leave();
  return rc_;
}

static int tmi_layernorm_bwd_emit_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                      int32_t accumulate_dx, float* colsum, void* masked, float dropout_p,
                                      uint64_t dropout_seed, float* workspace, int64_t workspace_bytes, int32_t dtype,
                                      void* stream);
extern "C" int tmi_layernorm_bwd_emit(const void* dy, const void* x, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                      int32_t accumulate_dx, float* colsum, void* masked, float dropout_p,
                                      uint64_t dropout_seed, float* workspace, int64_t workspace_bytes, int32_t dtype,
                                      void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_layernorm_bwd_emit(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, colsum, masked, dropout_p, dropout_seed + tmi_plan_seed_delta(), workspace, workspace_bytes, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_layernorm_bwd_emit_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, colsum, masked, dropout_p, dropout_seed, workspace, workspace_bytes, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_layernorm_bwd_emit_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                      int32_t accumulate_dx, float* colsum, void* masked, float dropout_p,
                                      uint64_t dropout_seed, float* workspace, int64_t workspace_bytes, int32_t dtype,
                                      void* stream) {
  if (!colsum) {
    tmi_set_error("tmi_layernorm_bwd_emit: colsum is required");
    return TMI_ERR_INVALID;
  }
  return ln_bwd_launch(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, accumulate_dx, colsum, masked, dropout_p, dropout_seed,
                       dtype, stream, "tmi_layernorm_bwd_emit", workspace, workspace_bytes);
}

static int tmi_colsum_batched_impl(const void* dy, int64_t ld, int64_t dy_sb, float* out, int64_t out_sb, int64_t rows, int64_t N,
                                  int64_t nbatch, int32_t dtype, void* stream);
extern "C" int tmi_colsum_batched(const void* dy, int64_t ld, int64_t dy_sb, float* out, int64_t out_sb, int64_t rows, int64_t N,
                                  int64_t nbatch, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_colsum_batched(dy, ld, dy_sb, out, out_sb, rows, N, nbatch, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_colsum_batched_impl(dy, ld, dy_sb, out, out_sb, rows, N, nbatch, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_colsum_batched_impl(const void* dy, int64_t ld, int64_t dy_sb, float* out, int64_t out_sb, int64_t rows, int64_t N,
                                  int64_t nbatch, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !out || rows <= 0 || N <= 0 || N % vec || ld % vec || !al16(dy) || nbatch <= 0 || nbatch > 65535 ||
      (nbatch > 1 && dy_sb % vec)) {
    tmi_set_error("tmi_colsum: bad argument (N, ld and the batch stride must be multiples of 16 bytes)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t xb = (N + 64 * vec - 1) / (64 * vec);
  static const int64_t cs_blocks = [] { const char* e = getenv("TMI_COLSUM_BLOCKS"); return e ? atoll(e) : 128ll; }();
  int64_t yb = (rows + CS_WAVES * 4 - 1) / (CS_WAVES * 4);  // at least four rows per wave
  const int64_t cap = (cs_blocks + xb * nbatch - 1) / (xb * nbatch);
  if (yb > cap) yb = cap;
  if (yb < 1 || tmi_deterministic()) yb = 1;  // (one workgroup per column group: a single contributor per column, no atomic order)
  dim3 grid((unsigned)xb, (unsigned)yb, (unsigned)nbatch);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(64 * CS_WAVES), 0, s, (const bf16_t*)dy, ld, out, rows, N, dy_sb, out_sb);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(64 * CS_WAVES), 0, s, (const float*)dy, ld, out, rows, N, dy_sb, out_sb);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_colsum");
}

static int tmi_colsum_impl(const void* dy, int64_t ld, float* out, int64_t rows, int64_t N, int32_t dtype,
                          void* stream);
extern "C" int tmi_colsum(const void* dy, int64_t ld, float* out, int64_t rows, int64_t N, int32_t dtype,
                          void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_colsum(dy, ld, out, rows, N, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_colsum_impl(dy, ld, out, rows, N, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_colsum_impl(const void* dy, int64_t ld, float* out, int64_t rows, int64_t N, int32_t dtype,
                          void* stream) {
  return tmi_colsum_batched(dy, ld, 0, out, 0, rows, N, 1, dtype, stream);
}

static int tmi_gelu_bwd_batched_impl(const void* dy, const void* u, void* dx, int64_t n, int64_t nbatch, int64_t dy_sb,
                                    int64_t u_sb, int64_t dx_sb, int32_t dtype, void* stream);
extern "C" int tmi_gelu_bwd_batched(const void* dy, const void* u, void* dx, int64_t n, int64_t nbatch, int64_t dy_sb,
                                    int64_t u_sb, int64_t dx_sb, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_gelu_bwd_batched(dy, u, dx, n, nbatch, dy_sb, u_sb, dx_sb, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_gelu_bwd_batched_impl(dy, u, dx, n, nbatch, dy_sb, u_sb, dx_sb, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_gelu_bwd_batched_impl(const void* dy, const void* u, void* dx, int64_t n, int64_t nbatch, int64_t dy_sb,
                                    int64_t u_sb, int64_t dx_sb, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !u || !dx || n <= 0 || nbatch <= 0 || nbatch > 65535 || n % vec || dy_sb % vec || u_sb % vec || dx_sb % vec ||
      !al16(dy) || !al16(u) || !al16(dx)) {
    tmi_set_error("tmi_gelu_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t nvec = n / vec;
  int64_t blocks = (nvec + 255) / 256;
  const int64_t cap = 4096 / nbatch > 0 ? 4096 / nbatch : 1;
  if (blocks > cap) blocks = cap;
  dim3 grid((unsigned)blocks, (unsigned)nbatch);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)dx, nvec,
                       dy_sb, u_sb, dx_sb);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)u, (float*)dx, nvec,
                       dy_sb, u_sb, dx_sb);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_gelu_bwd");
}

static int tmi_gelu_bwd_impl(const void* dy, const void* u, void* dx, int64_t n, int32_t dtype, void* stream);
extern "C" int tmi_gelu_bwd(const void* dy, const void* u, void* dx, int64_t n, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_gelu_bwd(dy, u, dx, n, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_gelu_bwd_impl(dy, u, dx, n, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_gelu_bwd_impl(const void* dy, const void* u, void* dx, int64_t n, int32_t dtype, void* stream) {
  return tmi_gelu_bwd_batched(dy, u, dx, n, 1, 0, 0, 0, dtype, stream);
}
