// LayerNorm forward / backward over the last axis (HBM-bound; one wavefront per row,
// 16-byte loads, wavefront-shuffle reductions).  Replaces
// tf.keras.layers.LayerNormalization(epsilon=1e-5) at speech_jobs/whisper_dist.py:214,216,
// 245,249,253,322,392 and its gradient.
// Algorithmic bytes: fwd 2*n*s (+8 B/row stats), bwd 3*n*s (+ dgamma/dbeta partials).
#include "tmi_common.h"

namespace {

constexpr int LN_E = 32;  // elements a lane can hold: C <= 64 * LN_E = 2048

template <typename T>
struct RowIO {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int NCH = LN_E / VEC;  // chunks per lane
  // chunk j of lane covers columns (j*64 + lane)*VEC .. +VEC
  __device__ static __forceinline__ void load(const T* row, int C, int lane, float (&v)[LN_E]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c0 = (j * 64 + lane) * VEC;
      if (c0 < C) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(row + c0);
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[j * VEC + i] = to_f32(e[i]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[j * VEC + i] = 0.f;
      }
    }
  }
  __device__ static __forceinline__ void store(T* row, int C, int lane, const float (&v)[LN_E]) {
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
  __device__ static __forceinline__ void load_f32vec(const float* p, int C, int lane, float (&v)[LN_E]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c0 = (j * 64 + lane) * VEC;
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[j * VEC + i] = (c0 < C) ? p[c0 + i] : 0.f;
    }
  }
};

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int C, float eps) {
  using IO = RowIO<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[LN_E], g[LN_E], b[LN_E];
  IO::load(x + row * C, C, lane, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_E; ++i) s += v[i];
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < IO::NCH; ++j) {
    const bool in = ((j * 64 + lane) * IO::VEC) < C;
#pragma unroll
    for (int i = 0; i < IO::VEC; ++i) {
      const float dlt = in ? v[j * IO::VEC + i] - mu : 0.f;
      q += dlt * dlt;
    }
  }
  const float var = wave_sum(q) / (float)C;
  const float rs = 1.0f / sqrtf(var + eps);
  IO::load_f32vec(gamma, C, lane, g);
  IO::load_f32vec(beta, C, lane, b);
#pragma unroll
  for (int i = 0; i < LN_E; ++i) v[i] = (v[i] - mu) * rs * g[i] + b[i];
  IO::store(y + row * C, C, lane, v);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// Each wave walks rows blockIdx*4+wave, +gridDim*4, ...; dgamma/dbeta partials are folded over
// the block's 4 waves in LDS and added to the fp32 gradient (zeroed by the caller) with one
// atomic per column per block: 256 contiguous bytes per wave-instruction.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C,
                                                     int accumulate_dx) {
  using IO = RowIO<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[LN_E], dg[LN_E], db[LN_E];
  IO::load_f32vec(gamma, C, lane, g);
#pragma unroll
  for (int i = 0; i < LN_E; ++i) dg[i] = db[i] = 0.f;
  const float invC = 1.0f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float xv[LN_E], dv[LN_E];
    IO::load(x + row * C, C, lane, xv);
    IO::load(dy + row * C, C, lane, dv);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < IO::NCH; ++j) {
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
    float old[LN_E];
    if (accumulate_dx) IO::load(dx + row * C, C, lane, old);
#pragma unroll
    for (int e = 0; e < LN_E; ++e) {
      float r = rs * (dv[e] * g[e] - s1 - xv[e] * s2);
      if (accumulate_dx) r += old[e];
      dv[e] = r;
    }
    IO::store(dx + row * C, C, lane, dv);
  }
#pragma unroll
  for (int j = 0; j < IO::NCH; ++j) {
    const int c0 = (j * 64 + lane) * IO::VEC;
    if (c0 < C) {
#pragma unroll
      for (int i = 0; i < IO::VEC; ++i) {
        red[(wave * 2 + 0) * C + c0 + i] = dg[j * IO::VEC + i];
        red[(wave * 2 + 1) * C + c0 + i] = db[j * IO::VEC + i];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * 2 + 0) * C + c];
      b += red[(w * 2 + 1) * C + c];
    }
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, b);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, int64_t ld, float* __restrict__ out,
                                                     int64_t rows, int64_t N) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[4][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (c0 < N) {
    for (int64_t row = (int64_t)blockIdx.y * 4 + wave; row < rows; row += (int64_t)gridDim.y * 4) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(dy + row * ld + c0);
      const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += to_f32(e[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[wave][lane * VEC + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * VEC; c += 256) {
    const int64_t n = (int64_t)blockIdx.x * 64 * VEC + c;
    if (n < N) atomicAdd(out + n, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u,
                                                       T* __restrict__ dx, int64_t nvec) {
  constexpr int VEC = 16 / sizeof(T);
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

extern "C" int tmi_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int64_t rows, int64_t C, float eps, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0 || C > 64 * LN_E || C % vec ||
      !al16(x) || !al16(y)) {
    tmi_set_error("tmi_layernorm_fwd: bad argument (C must be a multiple of 16 bytes and <= 2048)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y,
                       mean, rstd, rows, (int)C, eps);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean,
                       rstd, rows, (int)C, eps);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_layernorm_fwd");
}

extern "C" int tmi_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C,
                                 int32_t accumulate_dx, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0 || C > 64 * LN_E ||
      C % vec || !al16(x) || !al16(dy) || !al16(dx)) {
    tmi_set_error("tmi_layernorm_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = (rows + 7) / 8;  // two rows per wave: enough waves in flight to cover HBM latency
  if (blocks > 2048) blocks = 2048;
  const size_t lds = (size_t)8 * C * sizeof(float);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, s, (const bf16_t*)dy,
                       (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx, dgamma, dbeta, rows, (int)C, accumulate_dx);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, s, (const float*)dy,
                       (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, rows, (int)C, accumulate_dx);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_layernorm_bwd");
}

extern "C" int tmi_colsum(const void* dy, int64_t ld, float* out, int64_t rows, int64_t N, int32_t dtype,
                          void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !out || rows <= 0 || N <= 0 || N % vec || ld % vec || !al16(dy)) {
    tmi_set_error("tmi_colsum: bad argument (N and ld must be multiples of 16 bytes)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t xb = (N + 64 * vec - 1) / (64 * vec);
  int64_t yb = (rows + 31) / 32;  // 8 rows per wave
  const int64_t cap = (1024 + xb - 1) / xb;
  if (yb > cap) yb = cap;
  if (yb < 1) yb = 1;
  dim3 grid((unsigned)xb, (unsigned)yb);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, ld, out, rows, N);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, ld, out, rows, N);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_colsum");
}

extern "C" int tmi_gelu_bwd(const void* dy, const void* u, void* dx, int64_t n, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!dy || !u || !dx || n <= 0 || n % vec || !al16(dy) || !al16(u) || !al16(dx)) {
    tmi_set_error("tmi_gelu_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t nvec = n / vec;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dy,
                       (const bf16_t*)u, (bf16_t*)dx, nvec);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy,
                       (const float*)u, (float*)dx, nvec);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_gelu_bwd");
}
