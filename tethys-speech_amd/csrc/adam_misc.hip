// Fused multi-tensor Adam over a flat fp32 arena, bf16 weight shadows, input layout
// transform, sum of squares.  All pure-HBM kernels (16-byte accesses, grid-stride).
// Reference call sites: tf.keras.optimizers.Adam (speech_jobs/whisper_dist.py:901,
// speech_jobs/wav2vec2_dist.py:1271-1275) applied by optimizer.apply_gradients (W:834);
// tf.transpose + Conv1D "same" padding (W:329, W:311); tf.clip_by_global_norm (V:1243).
#include "tmi_common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float step_size, float vcorr_inv_sqrt, int eps_mode, float decay,
                                      float gscale) {
  g *= gscale;
  m = b1 * m + (1.0f - b1) * g;
  v = b2 * v + (1.0f - b2) * g * g;
  p *= decay;
  if (eps_mode == 0) {
    p -= step_size * m / (sqrtf(v) + eps);  // Keras-V2: eps beside sqrt(v), correction in step_size
  } else {
    p -= step_size * m / (sqrtf(v) * vcorr_inv_sqrt + eps);  // torch: eps beside sqrt(v_hat)
  }
}

// ZG: also write zeros over the gradients just consumed (the next step's backward accumulates into a clean arena
// without a separate fill pass: +4 B/param of stores here against 4 B/param of stores plus a launch there)
template <bool ZG>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float b1,
                                                   float b2, float eps, float step_size, float vcorr_inv_sqrt,
                                                   int eps_mode, float decay, float gscale,
                                                   bf16_t* __restrict__ mirror, const float* __restrict__ dev_scalars) {
  if (dev_scalars) {  // step-dependent scalars from device memory (a captured graph replays this launch)
    step_size = dev_scalars[0];
    vcorr_inv_sqrt = dev_scalars[1];
    decay = dev_scalars[2];
  }
  const int64_t nvec = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = pp[j], b = mm[j], c = vv[j];
      adam1(a, gg[j], b, c, b1, b2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale);
      pp[j] = a; mm[j] = b; vv[j] = c;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if constexpr (ZG) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mirror) {
      bf16x4 sh;
#pragma unroll
      for (int j = 0; j < 4; ++j) sh[j] = (bf16_t)pp[j];
      reinterpret_cast<bf16x4*>(mirror)[i] = sh;
    }
  }
  if (blockIdx.x == 0) {
    const int64_t i = nvec * 4 + threadIdx.x;
    if (i < n) {
      adam1(p[i], g[i], m[i], v[i], b1, b2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale);
      if constexpr (ZG) g[i] = 0.f;
      if (mirror) mirror[i] = (bf16_t)p[i];
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int64_t lds_, bf16_t* __restrict__ dst,
                                                        int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t total = rows * ldd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ldd, c = i % ldd;
    dst[i] = (c < cols) ? (bf16_t)src[r * lds_ + c] : (bf16_t)0.f;
  }
}

// 64x64 tile transpose through LDS: dst[c*ldd + r] = bf16(src[r*lds + c])
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, int64_t lds_,
                                                             bf16_t* __restrict__ dst, int64_t ldd, int64_t rows,
                                                             int64_t cols) {
  __shared__ float tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[c * ldd + r] = (bf16_t)tile[tx][i];
  }
}

// feats [B, C, T] fp32 -> out [B, Tp, C] (Tp = T + pl + pr), pad rows zero.
template <typename T>
__global__ __launch_bounds__(256) void feat_cl_kernel(const float* __restrict__ feats, T* __restrict__ out, int C,
                                                      int64_t Tin, int64_t Tp, int pl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);  // [64][C+1]
  const int64_t b = blockIdx.y;
  const int64_t tp0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t t = tp0 + tx - pl;
  for (int c = ty; c < C; c += 4)
    tile[tx * (C + 1) + c] = (t >= 0 && t < Tin) ? feats[(b * C + c) * Tin + t] : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * C; i += 256) {
    const int row = i / C, c = i % C;
    const int64_t tp = tp0 + row;
    if (tp < Tp) out[(b * Tp + tp) * C + c] = from_f32<T>(tile[row * (C + 1) + c]);
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = x[i];
    s += a * a;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// out[r, c] = (resid ? resid[r, c] : 0) + (keep(r, c) ? in[r, c] * scale : 0)      (Dropout, see tmi_common.h)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ in, int64_t ld_in, const T* __restrict__ resid,
                                                      int64_t ld_res, T* __restrict__ out, int64_t ld_out, int64_t rows,
                                                      int64_t cols, uint32_t key, uint32_t thr, float scale) {
  const int64_t pairs_per_row = cols >> 1;  // cols is even (host-checked): a counter pair never straddles rows
  const int64_t total = rows * pairs_per_row;
  for (int64_t pid = (int64_t)blockIdx.x * 256 + threadIdx.x; pid < total; pid += (int64_t)gridDim.x * 256) {
    const int64_t r = pid / pairs_per_row, c = (pid - r * pairs_per_row) * 2;
    const uint32_t h = tmi_pair_hash(tmi_row_key(key, (uint32_t)r), (uint32_t)(c >> 1));
    const bool k0 = (h & 0xffffu) >= thr, k1 = (h >> 16) >= thr;
    float a0 = k0 ? to_f32(in[r * ld_in + c]) * scale : 0.f;
    float a1 = k1 ? to_f32(in[r * ld_in + c + 1]) * scale : 0.f;
    if (resid) {
      a0 += to_f32(resid[r * ld_res + c]);
      a1 += to_f32(resid[r * ld_res + c + 1]);
    }
    out[r * ld_out + c] = from_f32<T>(a0);
    out[r * ld_out + c + 1] = from_f32<T>(a1);
  }
}

// The same, eight elements (four counter pairs) per thread per iteration with 16-byte accesses: cols % 8 == 0 and
// 16-byte aligned rows (host-checked).
template <typename T> struct Ld8;
template <> struct Ld8<bf16_t> {
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
template <> struct Ld8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};
template <typename T>
__global__ __launch_bounds__(256) void dropout_vec_kernel(const T* __restrict__ in, int64_t ld_in, const T* __restrict__ resid,
                                                          int64_t ld_res, T* __restrict__ out, int64_t ld_out, int64_t rows,
                                                          int64_t cols, uint32_t key, uint32_t thr, float scale) {
  const int64_t cpr = cols >> 3, total = rows * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / cpr, c = (idx - r * cpr) * 8;
    const tmi_rowkey rk = tmi_row_key(key, (uint32_t)r);
    const uint32_t cp0 = (uint32_t)(c >> 1);
    float v[8], t[8];
    Ld8<T>::load(in + r * ld_in + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = tmi_pair_hash(rk, cp0 + j);
      v[2 * j] = (h & 0xffffu) >= thr ? v[2 * j] * scale : 0.f;
      v[2 * j + 1] = (h >> 16) >= thr ? v[2 * j + 1] * scale : 0.f;
    }
    if (resid) {
      Ld8<T>::load(resid + r * ld_res + c, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += t[i];
    }
    Ld8<T>::store(out + r * ld_out + c, v);
  }
}

// Row-sparse form for embedding tables (tf.keras.layers.Embedding, W:382): a row that has never received a gradient
// has m = v = 0, and with g = 0 its Adam update is exactly zero (m, v stay 0; p -= step * 0 / (0 + eps)).  One wave per
// row: read g; if the row is all zero and was never active, touch nothing else (4 of the 28-32 B/param); otherwise
// update it and mark it active.  Bit-identical to the dense kernel on every row.
template <bool ZG>
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t nrows, int row_len,
                                                        unsigned char* __restrict__ active, float b1, float b2, float eps,
                                                        float step_size, float vcorr_inv_sqrt, int eps_mode, float decay,
                                                        float gscale, bf16_t* __restrict__ mirror) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int nvec = row_len >> 2;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t base = r * row_len;
    bool any = false;
    for (int i = lane; i < nvec; i += 64) {
      const f32x4 gg = reinterpret_cast<const f32x4*>(g + base)[i];
      any |= (gg[0] != 0.f) | (gg[1] != 0.f) | (gg[2] != 0.f) | (gg[3] != 0.f);
    }
    const bool live = __ballot(any) != 0ull;
    const bool was = active[r] != 0;
    if (!live && !was) continue;  // (wave-uniform)
    if (live && !was && lane == 0) active[r] = 1;
    for (int i = lane; i < nvec; i += 64) {
      f32x4 pp = reinterpret_cast<f32x4*>(p + base)[i];
      const f32x4 gg = reinterpret_cast<const f32x4*>(g + base)[i];
      f32x4 mm = reinterpret_cast<f32x4*>(m + base)[i];
      f32x4 vv = reinterpret_cast<f32x4*>(v + base)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = pp[j], b = mm[j], c = vv[j];
        adam1(a, gg[j], b, c, b1, b2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale);
        pp[j] = a; mm[j] = b; vv[j] = c;
      }
      reinterpret_cast<f32x4*>(p + base)[i] = pp;
      reinterpret_cast<f32x4*>(m + base)[i] = mm;
      reinterpret_cast<f32x4*>(v + base)[i] = vv;
      if constexpr (ZG) reinterpret_cast<f32x4*>(g + base)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (mirror) {
        bf16x4 sh;
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[j] = (bf16_t)pp[j];
        reinterpret_cast<bf16x4*>(mirror + base)[i] = sh;
      }
    }
  }
}

// Adam over the reference's variables with the gradient clipping of V:1243 (tf.clip_by_global_norm before the update)
// and V:1274 (Keras clipnorm: tf.clip_by_norm per variable) folded in as a per-variable factor of g, from ONE
// sum-of-squares pass over the raw gradients:
//   c_g = clip_global / max(sqrt(sum_s sumsq[s]), clip_global)                 (1 when clip_global == 0)
//   c_s = clip_each   / max(c_g * sqrt(sumsq[s]), clip_each)                   (1 when clip_each == 0)
//   g'  = g * gscale * c_g * c_s
// The arena is walked as a table of chunks (lo, hi, variable) of at most a few thousand elements, never crossing a
// variable boundary (built once per model on the host): workgroups stride over the table, so the launch is as
// uniform as the flat kernel's.  No clipped copy of the arena is written.
template <bool ZG>
__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, const int64_t* __restrict__ chunks,
                                                            int64_t nchunks,
                                                            const float* __restrict__ sumsq, int nseg, float clip_global,
                                                            float clip_each, float b1, float b2, float eps, float step_size,
                                                            float vcorr_inv_sqrt, int eps_mode, float decay, float gscale,
                                                            bf16_t* __restrict__ mirror) {
  __shared__ float red[4];
  float cg = 1.0f;
  if (clip_global > 0.f) {
    float t = 0.f;
    for (int i = threadIdx.x; i < nseg; i += 256) t += sumsq[i];
    t = block_sum_256(t, red);
    cg = clip_global / fmaxf(sqrtf(t), clip_global);
  }
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
  const int64_t a = chunks[3 * ch], b = chunks[3 * ch + 1];
  const int sgi = (int)chunks[3 * ch + 2];
  float cs = 1.0f;
  if (clip_each > 0.f) cs = clip_each / fmaxf(cg * sqrtf(sumsq[sgi]), clip_each);
  const float scale = gscale * cg * cs;
  const bool vec = (a & 3) == 0;
  const int64_t nv = vec ? (b - a) / 4 : 0;
  auto upd = [&](int64_t e, f32x4 pp, const f32x4 gg, f32x4 mm, f32x4 vv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = pp[j], y = mm[j], z = vv[j];
      adam1(x, gg[j], y, z, b1, b2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, scale);
      pp[j] = x; mm[j] = y; vv[j] = z;
    }
    *reinterpret_cast<f32x4*>(p + e) = pp;
    *reinterpret_cast<f32x4*>(m + e) = mm;
    *reinterpret_cast<f32x4*>(v + e) = vv;
    if constexpr (ZG) *reinterpret_cast<f32x4*>(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mirror) {
      bf16x4 sh;
#pragma unroll
      for (int j = 0; j < 4; ++j) sh[j] = (bf16_t)pp[j];
      *reinterpret_cast<bf16x4*>(mirror + e) = sh;
    }
  };
  int64_t i = threadIdx.x;
  for (; i + 256 < nv; i += 512) {  // two 4 x 16-byte load groups in flight per thread
    const int64_t e0 = a + 4 * i, e1 = e0 + 1024;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(p + e0), g0 = *reinterpret_cast<const f32x4*>(g + e0);
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(m + e0), v0 = *reinterpret_cast<const f32x4*>(v + e0);
    const f32x4 p1 = *reinterpret_cast<const f32x4*>(p + e1), g1 = *reinterpret_cast<const f32x4*>(g + e1);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(m + e1), v1 = *reinterpret_cast<const f32x4*>(v + e1);
    upd(e0, p0, g0, m0, v0);
    upd(e1, p1, g1, m1, v1);
  }
  for (; i < nv; i += 256) {
    const int64_t e = a + 4 * i;
    upd(e, *reinterpret_cast<const f32x4*>(p + e), *reinterpret_cast<const f32x4*>(g + e), *reinterpret_cast<const f32x4*>(m + e),
        *reinterpret_cast<const f32x4*>(v + e));
  }
  for (int64_t e = a + nv * 4 + threadIdx.x; e < b; e += 256) {
    adam1(p[e], g[e], m[e], v[e], b1, b2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, scale);
    if constexpr (ZG) g[e] = 0.f;
    if (mirror) mirror[e] = (bf16_t)p[e];
  }
  }
}

}  // namespace

static int tmi_adam_step_segments_impl(float* p, float* g, float* m, float* v, const int64_t* chunks, int64_t nchunks,
                                      const float* sumsq, int64_t nseg, float clip_global, float clip_each, float lr, float beta1, float beta2,
                                      float eps, int32_t step, int32_t eps_mode, float weight_decay, float gscale,
                                      void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream);
extern "C" int tmi_adam_step_segments(float* p, float* g, float* m, float* v, const int64_t* chunks, int64_t nchunks,
                                      const float* sumsq, int64_t nseg, float clip_global, float clip_each, float lr, float beta1, float beta2,
                                      float eps, int32_t step, int32_t eps_mode, float weight_decay, float gscale,
                                      void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_adam_step_segments(p, g, m, v, chunks, nchunks, sumsq, nseg, clip_global, clip_each, lr, beta1, beta2, eps, (int32_t)(step + tmi_plan_step_delta()), eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, max_blocks, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_adam_step_segments_impl(p, g, m, v, chunks, nchunks, sumsq, nseg, clip_global, clip_each, lr, beta1, beta2, eps, step, eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, max_blocks, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_adam_step_segments_impl(float* p, float* g, float* m, float* v, const int64_t* chunks, int64_t nchunks,
                                      const float* sumsq, int64_t nseg, float clip_global, float clip_each, float lr, float beta1, float beta2,
                                      float eps, int32_t step, int32_t eps_mode, float weight_decay, float gscale,
                                      void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream) {
  if (!p || !g || !m || !v || !chunks || nchunks <= 0 || nseg <= 0 || step <= 0 || max_blocks < 0 || (eps_mode != 0 && eps_mode != 1) ||
      ((clip_global > 0.f || clip_each > 0.f) && !sumsq) || clip_global < 0.f || clip_each < 0.f || !al16(p) || !al16(g) ||
      !al16(m) || !al16(v) || (bf16_mirror && (reinterpret_cast<uintptr_t>(bf16_mirror) & 7))) {
    tmi_set_error("tmi_adam_step_segments: bad argument");
    return TMI_ERR_INVALID;
  }
  const double c1 = 1.0 - pow((double)beta1, (double)step);
  const double c2 = 1.0 - pow((double)beta2, (double)step);
  float step_size, vcorr_inv_sqrt;
  if (eps_mode == 0) {
    step_size = (float)((double)lr * sqrt(c2) / c1);
    vcorr_inv_sqrt = 1.0f;
  } else {
    step_size = (float)((double)lr / c1);
    vcorr_inv_sqrt = (float)(1.0 / sqrt(c2));
  }
  const float decay = 1.0f - lr * weight_decay;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t nblocks = nchunks < 1024 ? nchunks : 1024;
  if (max_blocks > 0 && nblocks > max_blocks) nblocks = max_blocks;  // a throttled grid for a slice that runs beside other work
  dim3 grid((unsigned)nblocks);
  if (zero_grad)
    hipLaunchKernelGGL(adam_segments_kernel<true>, grid, dim3(256), 0, s, p, g, m, v, chunks, nchunks, sumsq, (int)nseg, clip_global,
                       clip_each, beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale, (bf16_t*)bf16_mirror);
  else
    hipLaunchKernelGGL(adam_segments_kernel<false>, grid, dim3(256), 0, s, p, g, m, v, chunks, nchunks, sumsq, (int)nseg, clip_global,
                       clip_each, beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale, (bf16_t*)bf16_mirror);
  return tmi_check_launch("tmi_adam_step_segments");
}

static int tmi_adam_step_rows_impl(float* p, float* g, float* m, float* v, int64_t nrows, int64_t row_len,
                                  unsigned char* active, float lr, float beta1, float beta2, float eps, int32_t step,
                                  int32_t eps_mode, float weight_decay, float gscale, void* bf16_mirror, int32_t zero_grad,
                                  void* stream);
extern "C" int tmi_adam_step_rows(float* p, float* g, float* m, float* v, int64_t nrows, int64_t row_len,
                                  unsigned char* active, float lr, float beta1, float beta2, float eps, int32_t step,
                                  int32_t eps_mode, float weight_decay, float gscale, void* bf16_mirror, int32_t zero_grad,
                                  void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_adam_step_rows(p, g, m, v, nrows, row_len, active, lr, beta1, beta2, eps, (int32_t)(step + tmi_plan_step_delta()), eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_adam_step_rows_impl(p, g, m, v, nrows, row_len, active, lr, beta1, beta2, eps, step, eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_adam_step_rows_impl(float* p, float* g, float* m, float* v, int64_t nrows, int64_t row_len,
                                  unsigned char* active, float lr, float beta1, float beta2, float eps, int32_t step,
                                  int32_t eps_mode, float weight_decay, float gscale, void* bf16_mirror, int32_t zero_grad,
                                  void* stream) {
  if (!p || !g || !m || !v || !active || nrows <= 0 || row_len <= 0 || row_len % 4 || step <= 0 ||
      (eps_mode != 0 && eps_mode != 1) || !al16(p) || !al16(g) || !al16(m) || !al16(v) ||
      (bf16_mirror && (reinterpret_cast<uintptr_t>(bf16_mirror) & 7)) || weight_decay != 0.f) {
    tmi_set_error("tmi_adam_step_rows: bad argument (16-byte aligned arenas, row_len % 4 == 0, no weight decay: a decayed row is never idle)");
    return TMI_ERR_INVALID;
  }
  const double c1 = 1.0 - pow((double)beta1, (double)step);
  const double c2 = 1.0 - pow((double)beta2, (double)step);
  float step_size, vcorr_inv_sqrt;
  if (eps_mode == 0) {
    step_size = (float)((double)lr * sqrt(c2) / c1);
    vcorr_inv_sqrt = 1.0f;
  } else {
    step_size = (float)((double)lr / c1);
    vcorr_inv_sqrt = (float)(1.0 / sqrt(c2));
  }
  int64_t blocks = (nrows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (zero_grad)
    hipLaunchKernelGGL(adam_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, nrows, (int)row_len, active,
                       beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, 1.0f, gscale, (bf16_t*)bf16_mirror);
  else
    hipLaunchKernelGGL(adam_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, nrows, (int)row_len, active,
                       beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, 1.0f, gscale, (bf16_t*)bf16_mirror);
  return tmi_check_launch("tmi_adam_step_rows");
}

static int tmi_adam_step_impl(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t step, int32_t eps_mode, float weight_decay,
                             float gscale, void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream);
extern "C" int tmi_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t step, int32_t eps_mode, float weight_decay,
                             float gscale, void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_adam_step(p, g, m, v, n, lr, beta1, beta2, eps, (int32_t)(step + tmi_plan_step_delta()), eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, max_blocks, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, step, eps_mode, weight_decay, gscale, bf16_mirror, zero_grad, max_blocks, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_adam_step_impl(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t step, int32_t eps_mode, float weight_decay,
                             float gscale, void* bf16_mirror, int32_t zero_grad, int32_t max_blocks, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0 || (eps_mode != 0 && eps_mode != 1) || !al16(p) || !al16(g) ||
      !al16(m) || !al16(v) || (bf16_mirror && (reinterpret_cast<uintptr_t>(bf16_mirror) & 7))) {
    tmi_set_error("tmi_adam_step: bad argument (arenas must be 16-byte aligned, step >= 1)");
    return TMI_ERR_INVALID;
  }
  // bias corrections in double on the host, as Keras computes lr_t from python-side scalars
  const double c1 = 1.0 - pow((double)beta1, (double)step);
  const double c2 = 1.0 - pow((double)beta2, (double)step);
  float step_size, vcorr_inv_sqrt;
  if (eps_mode == 0) {
    step_size = (float)((double)lr * sqrt(c2) / c1);
    vcorr_inv_sqrt = 1.0f;
  } else {
    step_size = (float)((double)lr / c1);
    vcorr_inv_sqrt = (float)(1.0 / sqrt(c2));
  }
  const float decay = 1.0f - lr * weight_decay;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  // two workgroups per CU stream the arena faster than thousands (measured: 0.83 vs 1.17 ms on 148 M parameters)
  static const int64_t cap = [] { const char* e = getenv("TMI_ADAM_BLOCKS"); return e ? atoll(e) : 512ll; }();
  if (blocks > cap) blocks = cap;
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;  // a slice updated beside other work: leave it the CUs
  if (zero_grad)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g,
                       m, v, n, beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale,
                       (bf16_t*)bf16_mirror, (const float*)nullptr);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g,
                       m, v, n, beta1, beta2, eps, step_size, vcorr_inv_sqrt, eps_mode, decay, gscale,
                       (bf16_t*)bf16_mirror, (const float*)nullptr);
  return tmi_check_launch("tmi_adam_step");
}

// host-side scalars of one Adam step, exactly as tmi_adam_step derives them: out3 = {step_size, vcorr_inv_sqrt, decay}
extern "C" int tmi_adam_scalars(float lr, float beta1, float beta2, int32_t step, int32_t eps_mode, float weight_decay,
                                float* out3) {
  if (!out3 || step <= 0 || (eps_mode != 0 && eps_mode != 1)) {
    tmi_set_error("tmi_adam_scalars: bad argument");
    return TMI_ERR_INVALID;
  }
  const double c1 = 1.0 - pow((double)beta1, (double)step);
  const double c2 = 1.0 - pow((double)beta2, (double)step);
  if (eps_mode == 0) {
    out3[0] = (float)((double)lr * sqrt(c2) / c1);
    out3[1] = 1.0f;
  } else {
    out3[0] = (float)((double)lr / c1);
    out3[1] = (float)(1.0 / sqrt(c2));
  }
  out3[2] = 1.0f - lr * weight_decay;
  return TMI_OK;
}

// the same update with the step-dependent scalars read from DEVICE memory (dev_scalars[3], filled from
// tmi_adam_scalars before each replay): the launch itself carries nothing that changes from step to
// step, so it can sit in a captured HIP graph
static int tmi_adam_step_dev_impl(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                                 float eps, const float* dev_scalars, int32_t eps_mode, float gscale, void* bf16_mirror,
                                 void* stream);
extern "C" int tmi_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                                 float eps, const float* dev_scalars, int32_t eps_mode, float gscale, void* bf16_mirror,
                                 void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_adam_step_dev(p, g, m, v, n, beta1, beta2, eps, dev_scalars, eps_mode, gscale, bf16_mirror, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_adam_step_dev_impl(p, g, m, v, n, beta1, beta2, eps, dev_scalars, eps_mode, gscale, bf16_mirror, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_adam_step_dev_impl(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                                 float eps, const float* dev_scalars, int32_t eps_mode, float gscale, void* bf16_mirror,
                                 void* stream) {
  if (!p || !g || !m || !v || !dev_scalars || n <= 0 || (eps_mode != 0 && eps_mode != 1) || !al16(p) || !al16(g) ||
      !al16(m) || !al16(v) || (bf16_mirror && (reinterpret_cast<uintptr_t>(bf16_mirror) & 7))) {
    tmi_set_error("tmi_adam_step_dev: bad argument");
    return TMI_ERR_INVALID;
  }
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p,
                     const_cast<float*>(g), m, v, n, beta1, beta2, eps, 0.f, 1.f, eps_mode, 1.f, gscale, (bf16_t*)bf16_mirror,
                     dev_scalars);
  return tmi_check_launch("tmi_adam_step_dev");
}

static int tmi_cast_bf16_impl(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                             void* stream);
extern "C" int tmi_cast_bf16(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                             void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_cast_bf16(src, lds_, dst, ldd, rows, cols, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_cast_bf16_impl(src, lds_, dst, ldd, rows, cols, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_cast_bf16_impl(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                             void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds_ < cols || ldd < cols) {
    tmi_set_error("tmi_cast_bf16: bad argument");
    return TMI_ERR_INVALID;
  }
  int64_t blocks = (rows * ldd + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, lds_, (bf16_t*)dst, ldd, rows, cols);
  return tmi_check_launch("tmi_cast_bf16");
}

// ---- gradient exchange staging (dist.py): fp32 arena slice <-> wire buffer
// pack: dst = bf16(src * scale); unpack: dst[i] = scale * sum_p src[p * part_stride + i] (src bf16 or fp32): the
// widening of a reduced bf16 bucket (nparts = 1) and the local sum of the N pieces a mesh reduce-scatter delivers.
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n,
                                                        float scale, int vec) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + 8 * i), b = *reinterpret_cast<const f32x4*>(src + 8 * i + 4);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (bf16_t)(a[j] * scale); o[4 + j] = (bf16_t)(b[j] * scale); }
      *reinterpret_cast<bf16x8*>(dst + 8 * i) = o;
    }
    for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (bf16_t)(src[i] * scale);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (bf16_t)(src[i] * scale);
  }
}

template <typename TS>
__global__ __launch_bounds__(256) void grad_unpack_kernel(const TS* __restrict__ src, int64_t nparts, int64_t part_stride,
                                                          float* __restrict__ dst, int64_t n, float scale, int vec) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t nv = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int64_t p = 0; p < nparts; ++p) {
        const TS* s = src + p * part_stride + 8 * i;
        if constexpr (sizeof(TS) == 2) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(s);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        } else {
          const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
        }
      }
      *reinterpret_cast<f32x4*>(dst + 8 * i) = f32x4{acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale};
      *reinterpret_cast<f32x4*>(dst + 8 * i + 4) = f32x4{acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale};
    }
    for (int64_t i = (nv << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      float a = 0.f;
      for (int64_t p = 0; p < nparts; ++p) a += (float)src[p * part_stride + i];
      dst[i] = a * scale;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      float a = 0.f;
      for (int64_t p = 0; p < nparts; ++p) a += (float)src[p * part_stride + i];
      dst[i] = a * scale;
    }
  }
}

static int tmi_grad_pack_impl(const float* src, void* dst, int64_t n, float scale, void* stream);
extern "C" int tmi_grad_pack(const float* src, void* dst, int64_t n, float scale, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_grad_pack(src, dst, n, scale, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_grad_pack_impl(src, dst, n, scale, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_grad_pack_impl(const float* src, void* dst, int64_t n, float scale, void* stream) {
  if (!src || !dst || n <= 0) {
    tmi_set_error("tmi_grad_pack: bad argument");
    return TMI_ERR_INVALID;
  }
  const int vec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? 1 : 0;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(grad_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     (bf16_t*)dst, n, scale, vec);
  return tmi_check_launch("tmi_grad_pack");
}

static int tmi_grad_unpack_impl(const void* src, int32_t src_dtype, int64_t nparts, int64_t part_stride, float* dst,
                               int64_t n, float scale, void* stream);
extern "C" int tmi_grad_unpack(const void* src, int32_t src_dtype, int64_t nparts, int64_t part_stride, float* dst,
                               int64_t n, float scale, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_grad_unpack(src, src_dtype, nparts, part_stride, dst, n, scale, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_grad_unpack_impl(src, src_dtype, nparts, part_stride, dst, n, scale, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_grad_unpack_impl(const void* src, int32_t src_dtype, int64_t nparts, int64_t part_stride, float* dst,
                               int64_t n, float scale, void* stream) {
  if (!src || !dst || n <= 0 || nparts <= 0 || (nparts > 1 && part_stride < n) || (src_dtype != TMI_F32 && src_dtype != TMI_BF16)) {
    tmi_set_error("tmi_grad_unpack: bad argument");
    return TMI_ERR_INVALID;
  }
  const int vec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                   (nparts == 1 || part_stride % 8 == 0)) ? 1 : 0;
  int64_t blocks = (n / 8 + 255) / 256;
  static const int64_t cap = [] { const char* e = getenv("TMI_GRAD_BLOCKS"); return e ? atoll(e) : 4096ll; }();
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (src_dtype == TMI_BF16)
    hipLaunchKernelGGL(grad_unpack_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)src, nparts, part_stride, dst, n, scale, vec);
  else
    hipLaunchKernelGGL(grad_unpack_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, nparts, part_stride, dst, n, scale, vec);
  return tmi_check_launch("tmi_grad_unpack");
}

static int tmi_transpose_cast_bf16_impl(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows,
                                       int64_t cols, void* stream);
extern "C" int tmi_transpose_cast_bf16(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows,
                                       int64_t cols, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_transpose_cast_bf16(src, lds_, dst, ldd, rows, cols, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_transpose_cast_bf16_impl(src, lds_, dst, ldd, rows, cols, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_transpose_cast_bf16_impl(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows,
                                       int64_t cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds_ < cols || ldd < rows || (cols + 63) / 64 > 65535 * 32) {
    tmi_set_error("tmi_transpose_cast_bf16: bad argument");
    return TMI_ERR_INVALID;
  }
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  if (grid.y > 65535) {
    tmi_set_error("tmi_transpose_cast_bf16: too many row tiles");
    return TMI_ERR_INVALID;
  }
  hipLaunchKernelGGL(transpose_cast_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, lds_,
                     (bf16_t*)dst, ldd, rows, cols);
  return tmi_check_launch("tmi_transpose_cast_bf16");
}

static int tmi_feat_to_channels_last_impl(const float* feats, void* out, int64_t B, int64_t C, int64_t T,
                                         int64_t pad_left, int64_t pad_right, int32_t dtype, void* stream);
extern "C" int tmi_feat_to_channels_last(const float* feats, void* out, int64_t B, int64_t C, int64_t T,
                                         int64_t pad_left, int64_t pad_right, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_feat_to_channels_last(feats, out, B, C, T, pad_left, pad_right, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_feat_to_channels_last_impl(feats, out, B, C, T, pad_left, pad_right, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_feat_to_channels_last_impl(const float* feats, void* out, int64_t B, int64_t C, int64_t T,
                                         int64_t pad_left, int64_t pad_right, int32_t dtype, void* stream) {
  if (!feats || !out || B <= 0 || C <= 0 || C > 256 || T <= 0 || pad_left < 0 || pad_right < 0 || B > 65535) {
    tmi_set_error("tmi_feat_to_channels_last: bad argument (C <= 256)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t Tp = T + pad_left + pad_right;
  dim3 grid((unsigned)((Tp + 63) / 64), (unsigned)B);
  const size_t lds = (size_t)64 * (C + 1) * sizeof(float);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(feat_cl_kernel<bf16_t>, grid, dim3(256), lds, s, feats, (bf16_t*)out, (int)C, T, Tp,
                       (int)pad_left);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(feat_cl_kernel<float>, grid, dim3(256), lds, s, feats, (float*)out, (int)C, T, Tp,
                       (int)pad_left);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_feat_to_channels_last");
}

static int tmi_sumsq_impl(const float* x, float* out, int64_t n, int32_t accumulate, void* stream);
extern "C" int tmi_sumsq(const float* x, float* out, int64_t n, int32_t accumulate, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_sumsq(x, out, n, accumulate, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_sumsq_impl(x, out, n, accumulate, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_sumsq_impl(const float* x, float* out, int64_t n, int32_t accumulate, void* stream) {
  if (!x || !out || n <= 0) {
    tmi_set_error("tmi_sumsq: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!accumulate && hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) return TMI_ERR_LAUNCH;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, out, n);
  return tmi_check_launch("tmi_sumsq");
}

// ---------------------------------------------------------------------------------------------
// Log-mel epilogue of the audio front end (speech_jobs/whisper_dist.py:752-764): the windowed DFT is
// a tmi_gemm (frames as overlapping rows, Hann folded into the cos | -sin matrix); this kernel turns
// one frame's spectrum [re(0..nb-1) | im(0..nb-1)] into power, applies the mel matrix [nb, n_mels] and
// writes log(mel + eps), either frame-major [F, n_mels] (the reference's layout) or channels-first
// [n_mels, F] (what the encoder consumes).  One wave per frame; HBM-bound (2*nb*4 B in, n_mels*4 B out).
namespace {
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ spec, int64_t lds_, const float* __restrict__ mel,
                                                     float* __restrict__ out, int64_t F, int nb, int n_mels, float eps,
                                                     int channels_first, int64_t out_ld) {
  extern __shared__ float pw[];  // [4 waves][nb]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f = (int64_t)blockIdx.x * 4 + wave;
  if (f >= F) return;  // (whole wave; no block barrier below)
  float* p = pw + wave * nb;
  const float* s = spec + f * lds_;
  for (int k = lane; k < nb; k += 64) {
    const float re = s[k], im = s[nb + k];
    p[k] = re * re + im * im;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes are visible to its own reads
  for (int m = lane; m < n_mels; m += 64) {
    float acc = 0.f;
    for (int k = 0; k < nb; ++k) acc = fmaf(p[k], mel[(int64_t)k * n_mels + m], acc);
    const float v = logf(acc + eps);
    if (channels_first) out[(int64_t)m * out_ld + f] = v;
    else out[f * out_ld + m] = v;
  }
}
}  // namespace

static int tmi_logmel_from_spectrum_impl(const float* spec, int64_t ld_spec, const float* mel, float* out, int64_t frames,
                                        int32_t n_bins, int32_t n_mels, float eps, int32_t channels_first, int64_t ld_out,
                                        void* stream);
extern "C" int tmi_logmel_from_spectrum(const float* spec, int64_t ld_spec, const float* mel, float* out, int64_t frames,
                                        int32_t n_bins, int32_t n_mels, float eps, int32_t channels_first, int64_t ld_out,
                                        void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_logmel_from_spectrum(spec, ld_spec, mel, out, frames, n_bins, n_mels, eps, channels_first, ld_out, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_logmel_from_spectrum_impl(spec, ld_spec, mel, out, frames, n_bins, n_mels, eps, channels_first, ld_out, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_logmel_from_spectrum_impl(const float* spec, int64_t ld_spec, const float* mel, float* out, int64_t frames,
                                        int32_t n_bins, int32_t n_mels, float eps, int32_t channels_first, int64_t ld_out,
                                        void* stream) {
  if (!spec || !mel || !out || frames <= 0 || n_bins <= 0 || n_mels <= 0 || ld_spec < 2 * (int64_t)n_bins ||
      ld_out < (channels_first ? frames : (int64_t)n_mels) || n_bins > 4096) {
    tmi_set_error("tmi_logmel_from_spectrum: bad argument");
    return TMI_ERR_INVALID;
  }
  const int64_t blocks = (frames + 3) / 4;
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * n_bins * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), spec, ld_spec, mel, out, frames, (int)n_bins, (int)n_mels, eps,
                     (int)channels_first, ld_out);
  return tmi_check_launch("tmi_logmel_from_spectrum");
}

// Dropout over a [rows, cols] tensor (cols even), optionally added to a residual: the forward of W:205 / W:342 /
// W:411 and, applied to the incoming gradient with the same seed, their backward.  In-place (out == in) is fine.
static int tmi_dropout_impl(const void* in, int64_t ld_in, const void* resid, int64_t ld_res, void* out, int64_t ld_out,
                           int64_t rows, int64_t cols, float p, uint64_t seed, int32_t dtype, void* stream);
extern "C" int tmi_dropout(const void* in, int64_t ld_in, const void* resid, int64_t ld_res, void* out, int64_t ld_out,
                           int64_t rows, int64_t cols, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_dropout(in, ld_in, resid, ld_res, out, ld_out, rows, cols, p, seed + tmi_plan_seed_delta(), dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_dropout_impl(in, ld_in, resid, ld_res, out, ld_out, rows, cols, p, seed, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_dropout_impl(const void* in, int64_t ld_in, const void* resid, int64_t ld_res, void* out, int64_t ld_out,
                           int64_t rows, int64_t cols, float p, uint64_t seed, int32_t dtype, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || (cols & 1) || cols > TMI_DROP_MAX_COLS || ld_in < cols || ld_out < cols || (resid && ld_res < cols) ||
      !(p >= 0.f && tmi_drop_ok(p))) {
    tmi_set_error("tmi_dropout: bad argument (cols must be even, 0 <= p < 1)");
    return TMI_ERR_INVALID;
  }
  const uint32_t thr = tmi_drop_thr(p);
  const float scale = tmi_keep_scale(thr);
  const uint32_t key = tmi_stream_key(seed, 0u);
  const int64_t pairs = rows * (cols >> 1);
  const unsigned blocks = (unsigned)((pairs + 255) / 256 < 2048 ? (pairs + 255) / 256 : 2048);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int esz = dtype == TMI_BF16 ? 2 : 4;
  auto al = [&](const void* p, int64_t ld) { return !p || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld * esz) % 16 == 0); };
  if (cols % 8 == 0 && al(in, ld_in) && al(resid, ld_res) && al(out, ld_out) && (dtype == TMI_BF16 || dtype == TMI_F32)) {
    const int64_t chunks = rows * (cols >> 3);
    const unsigned vb = (unsigned)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048);
    if (dtype == TMI_BF16)
      hipLaunchKernelGGL(dropout_vec_kernel<bf16_t>, dim3(vb), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(in), ld_in,
                         reinterpret_cast<const bf16_t*>(resid), ld_res, reinterpret_cast<bf16_t*>(out), ld_out, rows, cols, key, thr, scale);
    else
      hipLaunchKernelGGL(dropout_vec_kernel<float>, dim3(vb), dim3(256), 0, s, reinterpret_cast<const float*>(in), ld_in,
                         reinterpret_cast<const float*>(resid), ld_res, reinterpret_cast<float*>(out), ld_out, rows, cols, key, thr, scale);
    return tmi_check_launch("tmi_dropout");
  }
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(in), ld_in,
                       reinterpret_cast<const bf16_t*>(resid), ld_res, reinterpret_cast<bf16_t*>(out), ld_out, rows, cols, key, thr, scale);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(in), ld_in,
                       reinterpret_cast<const float*>(resid), ld_res, reinterpret_cast<float*>(out), ld_out, rows, cols, key, thr, scale);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_dropout");
}
