// Wav2Vec2-specific kernels (gfx950): GroupNorm+GELU of the conv feature encoder, layout
// packs that turn the grouped positional Conv1D into a batched window-GEMM, hard vector
// quantiser, contrastive cross-entropy, gradient clipping.  All HBM-bound.
// Reference call sites (speech_jobs/wav2vec2_dist.py, "V:"): GroupNormalization V:140-196 +
// gelu V:132-136 applied after every conv layer V:283-288; pos_conv_embed V:271-277,291;
// Wav2Vec2Quantizer.call V:581-667; _compute_contrastive_loss V:866-899;
// tf.clip_by_global_norm V:1243; Adam(clipnorm=1.0) V:1271-1275.
#include "tmi_common.h"

namespace {

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------- GroupNorm + GELU
// x [B][T][C] (batch stride xsb), contiguous channel groups of Cg = C / G.
// Thread tid of a 256-thread block always sees the same channel chunk (256*VEC % C == 0), hence
// one group: it keeps (a, b) partial sums for that group over the block's row range.
// mode 0 (forward stats):   a = sum x,            b = sum x^2
// mode 1 (backward sums):   a = sum dz*gamma,     b = sum dz*gamma*xhat,  dz = dy * gelu'(z)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int64_t xsb,
                                                         const T* __restrict__ dy, int64_t dysb,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ stats,
                                                         float* __restrict__ part, int Tn, int C, int G,
                                                         int rows_per_chunk, int nchunks) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float sa[256], sb[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int Cg = C / G;
  const int cpr = C / VEC;                       // vector chunks per row
  const int c0 = (threadIdx.x % cpr) * VEC;      // this thread's channels (fixed)
  const int g = c0 / Cg;
  const int r0 = chunk * rows_per_chunk, r1 = min(Tn, r0 + rows_per_chunk);
  const int rstep = 256 / cpr > 0 ? 256 / cpr : 1;
  float a = 0.f, bsum = 0.f;
  float gm[VEC], bt[VEC], mean = 0.f, rstd = 0.f;
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; }
    mean = stats[(b * G + g) * 2];
    rstd = stats[(b * G + g) * 2 + 1];
  }
  if (threadIdx.x < cpr * rstep) {
    auto one = [&](const u32x4& raw, const u32x4& rawd) {
      const T* e = reinterpret_cast<const T*>(&raw);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const float v = to_f32(e[i]); a += v; bsum += v * v; }
      } else {
        const T* ed = reinterpret_cast<const T*>(&rawd);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float xh = (to_f32(e[i]) - mean) * rstd;
          const float z = gm[i] * xh + bt[i];
          const float dz = to_f32(ed[i]) * gelu_grad_t<T>(z);
          a += dz * gm[i];
          bsum += dz * gm[i] * xh;
        }
      }
    };
    const T* xb = x + (int64_t)b * xsb + c0;
    const T* db = MODE == 1 ? dy + (int64_t)b * dysb + c0 : nullptr;
    int r = r0 + threadIdx.x / cpr;
    // four row loads in flight per thread (the chunk is a short stream: latency, not bandwidth, bounds it)
    for (; r + 3 * rstep < r1; r += 4 * rstep) {
      u32x4 raw[4], rawd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        raw[u] = *reinterpret_cast<const u32x4*>(xb + (int64_t)(r + u * rstep) * C);
        if (MODE == 1) rawd[u] = *reinterpret_cast<const u32x4*>(db + (int64_t)(r + u * rstep) * C);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(raw[u], rawd[u]);
    }
    for (; r < r1; r += rstep) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xb + (int64_t)r * C);
      u32x4 rawd = raw;
      if (MODE == 1) rawd = *reinterpret_cast<const u32x4*>(db + (int64_t)r * C);
      one(raw, rawd);
    }
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = bsum;
  __syncthreads();
  // fixed-order fold in two parallel levels: over the rstep row lanes of each vector column, then over
  // the Cg / VEC columns of each group
  if (threadIdx.x < cpr) {
    float ta = sa[threadIdx.x], tb = sb[threadIdx.x];
    for (int k = 1; k < rstep; ++k) { ta += sa[threadIdx.x + k * cpr]; tb += sb[threadIdx.x + k * cpr]; }
    sa[threadIdx.x] = ta;
    sb[threadIdx.x] = tb;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int per = Cg / VEC;  // vector columns per group
    float ta = 0.f, tb = 0.f;
    for (int t = 0; t < per; ++t) { ta += sa[threadIdx.x * per + t]; tb += sb[threadIdx.x * per + t]; }
    float* o = part + (((int64_t)b * nchunks + chunk) * G + threadIdx.x) * 2;
    o[0] = ta;
    o[1] = tb;
  }
}

// The fold of the chunk partials, done by the CONSUMER instead of a launch of its own (gn_finalize_kernel until round 4) (round 5: a 4.7 us kernel between two
// others costs its duration plus a kernel boundary, 14 times per Wav2Vec2 step): every workgroup of an apply kernel folds
// the chunk partials of ITS batch row - thread t takes group t % G and every (256 / G)-th chunk from t / G on, in double,
// then the first G threads add the slices in slice order (a fixed association: the result is a function of the partials
// alone) - and leaves (v0, v1) per group in `sh` ([2 * G] floats of LDS).  mode 0: (mean, rstd); mode 1: (mean(dxhat),
// mean(dxhat * xhat)).  The workgroups with blockIdx.x == 0 also store the pairs to `out` (the backward reads the
// statistics again).  `part` is L2-resident (written by the launch before) and 2 * nchunks * G floats per batch row.
__device__ __forceinline__ void gn_fold_block(const float* __restrict__ part, int b, int G, int nchunks, double count, float eps,
                                              int mode, float* __restrict__ sh, float* __restrict__ out) {
  __shared__ double shd[512];
  const int t = threadIdx.x;
  const int nsub = G >= 256 ? 1 : 256 / G;
  double a = 0.0, s_ = 0.0;
  if (t < nsub * G) {
    const int g = t % G, sub = t / G;
    for (int c = sub; c < nchunks; c += nsub) {
      const float* p = part + (((int64_t)b * nchunks + c) * G + g) * 2;
      a += p[0];
      s_ += p[1];
    }
  }
  shd[2 * t] = a;
  shd[2 * t + 1] = s_;
  __syncthreads();
  if (t < G) {
    double A = 0.0, S = 0.0;
    for (int sub = 0; sub < nsub; ++sub) {
      A += shd[2 * (sub * G + t)];
      S += shd[2 * (sub * G + t) + 1];
    }
    float v0, v1;
    if (mode == 0) {
      const double mean = A / count;
      double var = S / count - mean * mean;
      if (var < 0.0) var = 0.0;
      v0 = (float)mean;
      v1 = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      v0 = (float)(A / count);
      v1 = (float)(S / count);
    }
    sh[2 * t] = v0;
    sh[2 * t + 1] = v1;
    if (out && blockIdx.x == 0) {
      out[((int64_t)b * G + t) * 2] = v0;
      out[((int64_t)b * G + t) * 2 + 1] = v1;
    }
  }
  __syncthreads();
}

// forward apply: y = gelu(gamma * xhat + beta), written with its own batch stride (padded buffers)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const T* __restrict__ x, int64_t xsb,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ stats, T* __restrict__ y,
                                                           int64_t ysb, int Tn, int C, int G,
                                                           const float* __restrict__ part, int nchunks, double count, float eps) {
  constexpr int VEC = 16 / sizeof(T);
  const int b = blockIdx.y;
  const int Cg = C / G;
  __shared__ float shs[512];
  gn_fold_block(part, b, G, nchunks, count, eps, 0, shs, stats);  // (mean, rstd) of this batch row's groups
  const int64_t nvec = (int64_t)Tn * C / VEC;
  // 256 * VEC is a multiple of C (gn_check), so a thread keeps its channels across the grid-stride loop:
  // gamma / beta / statistics are loaded once
  const int c0 = (int)(((int64_t)threadIdx.x * VEC) % C);
  const int g = c0 / Cg;
  const float mean = shs[2 * g], rstd = shs[2 * g + 1];
  float gm[VEC], bt[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; }
  const T* xb = x + (int64_t)b * xsb;
  T* yb = y + (int64_t)b * ysb;
  const int64_t step = (int64_t)gridDim.x * 256;
  auto one = [&](const u32x4& raw, int64_t v) {
    const T* e = reinterpret_cast<const T*>(&raw);
    alignas(16) T o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = from_f32<T>(gelu_fwd_t<T>(gm[i] * ((to_f32(e[i]) - mean) * rstd) + bt[i]));
    *reinterpret_cast<u32x4*>(yb + v * VEC) = *reinterpret_cast<const u32x4*>(o);
  };
  int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; v + step < nvec; v += 2 * step) {
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(xb + v * VEC);
    const u32x4 r1 = *reinterpret_cast<const u32x4*>(xb + (v + step) * VEC);
    one(r0, v);
    one(r1, v + step);
  }
  if (v < nvec) one(*reinterpret_cast<const u32x4*>(xb + v * VEC), v);
}

// backward apply: dx = rstd * (dxhat - m1 - xhat * m2), dxhat = dz * gamma; dgamma += dz*xhat, dbeta += dz
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const T* __restrict__ x, int64_t xsb,
                                                           const T* __restrict__ dy, int64_t dysb,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ stats,
                                                           float* __restrict__ sums, T* __restrict__ dx,
                                                           int64_t dxsb, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int Tn, int C, int G,
                                                           int rows_per_chunk, const float* __restrict__ part, int nchunks,
                                                           double count) {
  constexpr int VEC = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [rstep][2][C]
  const int b = blockIdx.y, chunk = blockIdx.x;
  __shared__ float shs[512];
  gn_fold_block(part, b, G, nchunks, count, 0.f, 1, shs, sums);  // (mean(dxhat), mean(dxhat * xhat)) of this batch row's groups
  const int Cg = C / G;
  const int cpr = C / VEC;
  const int c0 = (threadIdx.x % cpr) * VEC;
  const int g = c0 / Cg;
  const int rstep = 256 / cpr > 0 ? 256 / cpr : 1;
  const int rl = threadIdx.x / cpr;
  const int r0 = chunk * rows_per_chunk, r1 = min(Tn, r0 + rows_per_chunk);
  float gm[VEC], bt[VEC], dg[VEC], db[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; dg[i] = db[i] = 0.f; }
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float m1 = shs[2 * g], m2 = shs[2 * g + 1];
  if (threadIdx.x < cpr * rstep) {
    for (int r = r0 + rl; r < r1; r += rstep) {
      const int64_t off = (int64_t)r * C + c0;
      const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (int64_t)b * xsb + off);
      const u32x4 rawd = *reinterpret_cast<const u32x4*>(dy + (int64_t)b * dysb + off);
      const T* e = reinterpret_cast<const T*>(&raw);
      const T* ed = reinterpret_cast<const T*>(&rawd);
      alignas(16) T o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xh = (to_f32(e[i]) - mean) * rstd;
        const float z = gm[i] * xh + bt[i];
        const float dz = to_f32(ed[i]) * gelu_grad_t<T>(z);
        dg[i] += dz * xh;
        db[i] += dz;
        o[i] = from_f32<T>(rstd * (dz * gm[i] - m1 - xh * m2));
      }
      *reinterpret_cast<u32x4*>(dx + (int64_t)b * dxsb + off) = *reinterpret_cast<const u32x4*>(o);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      red[(rl * 2 + 0) * C + c0 + i] = dg[i];
      red[(rl * 2 + 1) * C + c0 + i] = db[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, s = 0.f;
    for (int r = 0; r < rstep; ++r) { a += red[(r * 2 + 0) * C + c]; s += red[(r * 2 + 1) * C + c]; }
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, s);
  }
}

// ---------------------------------------------------------------- conv layer 0 as an FIR, fused with GroupNorm + GELU
// The first feature-encoder layer (V:283-288 with i = 0) is Conv1D(C, k = 10, stride 5, "same", no bias) on ONE input
// channel: 2*k flop per output element against 2-4 bytes written - a filter bank, not a GEMM.  Its output u0
// [B, T0, C] (52 MB at base size) is never materialised: the statistics pass, the apply pass and both backward passes
// recompute it from the raw audio (a few hundred samples per workgroup, staged in LDS) with the k x C taps in registers.
//   forward : stats (sum u, sum u^2 per (batch, group)) -> finalize -> y = gelu(gamma * xhat + beta)      [writes y only]
//   backward: sums (sum dxhat, sum dxhat*xhat)          -> finalize -> du on the fly; dW[j][c] += in[S t + j - pl] * du,
//             dgamma, dbeta                                                                  [reads dy twice, writes nothing big]
// Thread layout as the GroupNorm kernels above: a thread owns 8 consecutive channels (one group), C/8 threads cover a
// time step, 256/(C/8) time steps run in parallel, a workgroup walks a chunk of rows_per_chunk time steps.
template <int KW>
__device__ __forceinline__ void fir_load_taps(const float* __restrict__ w, int C, int c0, float (&tap)[KW][8]) {
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + (int64_t)j * C + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + (int64_t)j * C + c0 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { tap[j][i] = a[i]; tap[j][4 + i] = b[i]; }
  }
}
template <int KW>
__device__ __forceinline__ void fir8(const float* __restrict__ win, const float (&tap)[KW][8], float (&u)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = 0.f;
#pragma unroll
  for (int j = 0; j < KW; ++j) {
    const float x = win[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = fmaf(x, tap[j][i], u[i]);
  }
}
// stage audio samples [r0*S - pl, r0*S - pl + n) of batch row `a` (zeros outside [0, Tin)) into LDS
__device__ __forceinline__ void fir_stage(float* win, const float* __restrict__ a, int64_t first, int n, int Tin) {
  for (int i = threadIdx.x; i < n; i += 256) {
    const int64_t idx = first + i;
    win[i] = (idx >= 0 && idx < Tin) ? a[idx] : 0.f;
  }
}
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// MODE 0: a = sum u, b = sum u^2;  MODE 1: a = sum dz*gamma, b = sum dz*gamma*xhat (dz = dy * gelu'(z))
template <typename T, int KW, int S, int MODE>
__global__ __launch_bounds__(256) void fir_gn_partial_kernel(const float* __restrict__ audio, int64_t asb, int Tin, int pl,
                                                             const float* __restrict__ w, const T* __restrict__ dy,
                                                             int64_t dysb, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ stats,
                                                             float* __restrict__ part, int Tn, int C, int G,
                                                             int rows_per_chunk, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* win = reinterpret_cast<float*>(smem);  // [rows_per_chunk * S + KW]
  __shared__ float sa[256], sb[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int Cg = C / G, cpr = C / 8;
  const int c0 = (threadIdx.x % cpr) * 8, g = c0 / Cg;
  const int rstep = 256 / cpr;
  const int r0 = chunk * rows_per_chunk, r1 = min(Tn, r0 + rows_per_chunk);
  float tap[KW][8];
  fir_load_taps<KW>(w, C, c0, tap);
  fir_stage(win, audio + (int64_t)b * asb, (int64_t)r0 * S - pl, (r1 - r0) * S + KW, Tin);
  float gm[8], bt[8], mean = 0.f, rstd = 0.f;
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; }
    mean = stats[(b * G + g) * 2];
    rstd = stats[(b * G + g) * 2 + 1];
  }
  __syncthreads();
  float a = 0.f, bsum = 0.f;
  for (int r = r0 + threadIdx.x / cpr; r < r1; r += rstep) {
    float u[8];
    fir8<KW>(win + (r - r0) * S, tap, u);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a += u[i]; bsum += u[i] * u[i]; }
    } else {
      float d[8];
      load8<T>(dy + (int64_t)b * dysb + (int64_t)r * C + c0, d);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (u[i] - mean) * rstd;
        const float dz = d[i] * gelu_grad_t<T>(gm[i] * xh + bt[i]);
        a += dz * gm[i];
        bsum += dz * gm[i] * xh;
      }
    }
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = bsum;
  __syncthreads();
  if (threadIdx.x < cpr) {
    float ta = sa[threadIdx.x], tb = sb[threadIdx.x];
    for (int k = 1; k < rstep; ++k) { ta += sa[threadIdx.x + k * cpr]; tb += sb[threadIdx.x + k * cpr]; }
    sa[threadIdx.x] = ta;
    sb[threadIdx.x] = tb;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int per = Cg / 8;
    float ta = 0.f, tb = 0.f;
    for (int t = 0; t < per; ++t) { ta += sa[threadIdx.x * per + t]; tb += sb[threadIdx.x * per + t]; }
    float* o = part + (((int64_t)b * nchunks + chunk) * G + threadIdx.x) * 2;
    o[0] = ta;
    o[1] = tb;
  }
}

template <typename T, int KW, int S>
__global__ __launch_bounds__(256) void fir_gn_apply_fwd_kernel(const float* __restrict__ audio, int64_t asb, int Tin, int pl,
                                                               const float* __restrict__ w, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ stats,
                                                               T* __restrict__ y, int64_t ysb, int Tn, int C, int G,
                                                               int rows_per_chunk, const float* __restrict__ part, int nchunks,
                                                               double count, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* win = reinterpret_cast<float*>(smem);
  const int b = blockIdx.y, chunk = blockIdx.x;
  __shared__ float shs[512];
  gn_fold_block(part, b, G, nchunks, count, eps, 0, shs, stats);
  const int Cg = C / G, cpr = C / 8;
  const int c0 = (threadIdx.x % cpr) * 8, g = c0 / Cg;
  const int rstep = 256 / cpr;
  const int r0 = chunk * rows_per_chunk, r1 = min(Tn, r0 + rows_per_chunk);
  float tap[KW][8];
  fir_load_taps<KW>(w, C, c0, tap);
  fir_stage(win, audio + (int64_t)b * asb, (int64_t)r0 * S - pl, (r1 - r0) * S + KW, Tin);
  float gm[8], bt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; }
  const float mean = shs[2 * g], rstd = shs[2 * g + 1];
  __syncthreads();
  for (int r = r0 + threadIdx.x / cpr; r < r1; r += rstep) {
    float u[8];
    fir8<KW>(win + (r - r0) * S, tap, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = gelu_fwd_t<T>(gm[i] * ((u[i] - mean) * rstd) + bt[i]);
    store8<T>(y + (int64_t)b * ysb + (int64_t)r * C + c0, u);
  }
}

// backward apply: du = rstd * (dz*gamma - m1 - xhat*m2) stays in registers; per-workgroup partial sums of
// dW [KW][C], dgamma [C], dbeta [C] go to wpart[(b*nchunks + chunk)][(KW + 2) * C] (summed by fir_reduce_kernel)
template <typename T, int KW, int S>
__global__ __launch_bounds__(256) void fir_gn_apply_bwd_kernel(const float* __restrict__ audio, int64_t asb, int Tin, int pl,
                                                               const float* __restrict__ w, const T* __restrict__ dy,
                                                               int64_t dysb, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ stats,
                                                               float* __restrict__ sums, float* __restrict__ wpart,
                                                               int Tn, int C, int G, int rows_per_chunk, int nchunks,
                                                               const float* __restrict__ part, int npart, double count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* win = reinterpret_cast<float*>(smem);                       // [rows_per_chunk * S + KW]
  float* red = win + ((rows_per_chunk * S + KW + 3) & ~3);           // [rstep][C] fold buffer
  const int b = blockIdx.y, chunk = blockIdx.x;
  __shared__ float shs[512];
  gn_fold_block(part, b, G, npart, count, 0.f, 1, shs, sums);
  const int Cg = C / G, cpr = C / 8;
  const int c0 = (threadIdx.x % cpr) * 8, g = c0 / Cg;
  const int rstep = 256 / cpr, rl = threadIdx.x / cpr;
  const int r0 = chunk * rows_per_chunk, r1 = min(Tn, r0 + rows_per_chunk);
  float tap[KW][8], acc[KW + 2][8];
  fir_load_taps<KW>(w, C, c0, tap);
  fir_stage(win, audio + (int64_t)b * asb, (int64_t)r0 * S - pl, (r1 - r0) * S + KW, Tin);
  float gm[8], bt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { gm[i] = gamma[c0 + i]; bt[i] = beta[c0 + i]; }
#pragma unroll
  for (int j = 0; j < KW + 2; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float m1 = shs[2 * g], m2 = shs[2 * g + 1];
  __syncthreads();
  for (int r = r0 + rl; r < r1; r += rstep) {
    const float* wr = win + (r - r0) * S;
    float u[8], d[8];
    fir8<KW>(wr, tap, u);
    load8<T>(dy + (int64_t)b * dysb + (int64_t)r * C + c0, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (u[i] - mean) * rstd;
      const float dz = d[i] * gelu_grad_t<T>(gm[i] * xh + bt[i]);
      acc[KW][i] += dz * xh;   // dgamma
      acc[KW + 1][i] += dz;    // dbeta
      u[i] = rstd * (dz * gm[i] - m1 - xh * m2);  // du
    }
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const float x = wr[j];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(x, u[i], acc[j][i]);
    }
  }
  // fold the rstep row lanes, one quantity at a time, in a fixed order
  float* out = wpart + ((int64_t)b * nchunks + chunk) * (int64_t)(KW + 2) * C;
  for (int j = 0; j < KW + 2; ++j) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = 0.f;
#pragma unroll
      for (int jj = 0; jj < KW + 2; ++jj) v = jj == j ? acc[jj][i] : v;  // (static indexing keeps acc in registers)
      red[rl * C + c0 + i] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int k = 0; k < rstep; ++k) t += red[k * C + c];
      out[j * C + c] = t;
    }
  }
}

// dst[i] += sum over the partial rows of src[p][i]  (dW, dgamma, dbeta of the FIR layer): blockIdx.y takes every
// gridDim.y-th partial row, eight loads in flight per thread, one atomic per element per y-slice
__global__ __launch_bounds__(256) void fir_reduce_kernel(const float* __restrict__ src, int nparts, int64_t len_w,
                                                         int64_t C, float* __restrict__ dW, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta) {
  const int64_t len = len_w + 2 * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  float t = 0.f;
  int p = blockIdx.y;
  const int ps = gridDim.y;
  for (; p + 7 * ps < nparts; p += 8 * ps) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(p + u * ps) * len + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  for (; p < nparts; p += ps) t += src[(int64_t)p * len + i];
  float* dst = i < len_w ? dW + i : (i < len_w + C ? dgamma + (i - len_w) : dbeta + (i - len_w - C));
  atomicAdd(dst, t);
}

// ---------------------------------------------------------------- grouped pos-conv layout packs
// x [R][C] (rows r < R_valid come from the source, others are zero) -> xg [G][R][Cg]
template <typename T>
__global__ __launch_bounds__(256) void group_pack_kernel(const T* __restrict__ x, T* __restrict__ xg, int64_t R,
                                                         int C, int G, int64_t Tp, int64_t pl, int64_t Tn) {
  // destination row r of batch b = r / Tp is source row (b*Tn + r%Tp - pl) when 0 <= r%Tp - pl < Tn
  const int Cg = C / G;
  const int64_t total = R * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t g = i / (R * Cg), rem = i % (R * Cg);
    const int64_t r = rem / Cg, j = rem % Cg;
    const int64_t b = r / Tp, t = r % Tp - pl;
    T v = from_f32<T>(0.f);
    if (t >= 0 && t < Tn) v = x[(b * Tn + t) * C + g * Cg + j];
    xg[i] = v;
  }
}

// out[b*Tn + t][g*Cg + j] = yg[g][b*Tp + t + row_off][j] (+ bias[c]) (+ resid[..])
template <typename T>
__global__ __launch_bounds__(256) void group_unpack_kernel(const T* __restrict__ yg, const float* __restrict__ bias,
                                                           const T* __restrict__ resid, T* __restrict__ out, int64_t R,
                                                           int C, int G, int64_t Tp, int64_t row_off, int64_t Tn,
                                                           int64_t Bn) {
  const int Cg = C / G;
  const int64_t total = Bn * Tn * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / C;
    const int c = (int)(i % C);
    const int64_t b = row / Tn, t = row % Tn;
    const int g = c / Cg, j = c % Cg;
    float v = to_f32(yg[((int64_t)g * R + b * Tp + t + row_off) * Cg + j]);
    if (bias) v += bias[c];
    if (resid) v += to_f32(resid[i]);
    out[i] = from_f32<T>(v);
  }
}

// Keras grouped kernel w [k][Cg][C] fp32 -> forward form wf[g][(kk,i)][o] = w[kk][i][g*Cg+o] and
// backward form wb[g][(kk',o)][i] = w[k-1-kk'][i][g*Cg+o]
template <typename T>
__global__ __launch_bounds__(256) void posconv_pack_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                                           T* __restrict__ wb, int k, int Cg, int G) {
  const int C = Cg * G;
  const int64_t total = (int64_t)k * Cg * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int kk = (int)(i / ((int64_t)Cg * C));
    const int ii = (int)((i / C) % Cg);
    const int c = (int)(i % C);
    const int g = c / Cg, o = c % Cg;
    const T v = from_f32<T>(w[i]);
    wf[((int64_t)g * k * Cg + (int64_t)kk * Cg + ii) * Cg + o] = v;
    wb[((int64_t)g * k * Cg + (int64_t)(k - 1 - kk) * Cg + o) * Cg + ii] = v;
  }
}

// ---------------------------------------------------------------- hard vector quantiser
// h [rows][G*gd] (projected features), codebook [G][Nc][gd] fp32.  One wave per (row, group):
// squared distance to every code as sum((h-c)^2) in fp32 (the reference's own form, so ties
// and rounding match), argmin with first-index tie-break, quantised row = the code itself.
template <typename T>
__global__ __launch_bounds__(256) void vq_kernel(const T* __restrict__ h, const float* __restrict__ cb,
                                                 int32_t* __restrict__ idx_out, T* __restrict__ q, int64_t rows,
                                                 int G, int Nc, int gd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* hs = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * gd;
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * G) return;
  const int64_t row = item / G;
  const int g = (int)(item % G);
  for (int d = lane; d < gd; d += 64) hs[d] = to_f32(h[row * G * gd + g * gd + d]);
  __builtin_amdgcn_wave_barrier();
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < Nc; c += 64) {
    const float* cv = cb + ((int64_t)g * Nc + c) * gd;
    float s = 0.f;
    for (int d = 0; d < gd; ++d) {
      const float df = hs[d] - cv[d];
      s += df * df;
    }
    if (s < best) { best = s; bi = c; }  // c ascending per lane: first index kept on ties
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) idx_out[row * G + g] = bi;
  const float* cv = cb + ((int64_t)g * Nc + bi) * gd;
  for (int d = lane; d < gd; d += 64) q[row * G * gd + g * gd + d] = from_f32<T>(cv[d]);
}

// The quantiser with the code choice GIVEN (tmi_vq_assign): q[row][g] = codebook[g][idx[row][g]].  Teacher-forced runs
// (a recorded code sequence replayed: the bf16 golden-curve test feeds the fp64 oracle's choices so that the hard
// argmin's discontinuity does not turn rounding noise into a different trajectory).
template <typename T>
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ cb, const int32_t* __restrict__ idx,
                                                        T* __restrict__ q, int64_t rows, int G, int Nc, int gd) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * G) return;
  const int64_t row = item / G;
  const int g = (int)(item % G);
  int bi = idx[row * G + g];
  bi = bi < 0 ? 0 : (bi >= Nc ? Nc - 1 : bi);
  const float* cv = cb + ((int64_t)g * Nc + bi) * gd;
  for (int d = lane; d < gd; d += 64) q[row * G * gd + g * gd + d] = from_f32<T>(cv[d]);
}

// perplexity = mean_g exp(-sum_c p log(p + 1e-10)), p = clip(count / rows, 1e-10, 1)   (V:653-660)
__global__ __launch_bounds__(256) void vq_perplexity_kernel(const int32_t* __restrict__ idx, float* __restrict__ out,
                                                            int64_t rows, int G, int Nc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* cnt = reinterpret_cast<int*>(smem);  // [G*Nc]
  __shared__ float red[4];
  for (int i = threadIdx.x; i < G * Nc; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < rows * G; i += 256) atomicAdd(&cnt[(i % G) * Nc + idx[i]], 1);
  __syncthreads();
  float total = 0.f;
  for (int g = 0; g < G; ++g) {
    float s = 0.f;
    for (int c = threadIdx.x; c < Nc; c += 256) {
      float p = (float)cnt[g * Nc + c] / (float)rows;
      p = fminf(fmaxf(p, 1e-10f), 1.0f);
      s += p * logf(p + 1e-10f);
    }
    s = block_sum_256(s, red);
    total += expf(-s);
  }
  if (threadIdx.x == 0) out[0] = total / (float)G;
}

// dcodebook[g][idx][:] += dq[row][g*gd:(g+1)*gd]   (gradient of one_hot @ codebook, V:638)
template <typename T>
__global__ __launch_bounds__(256) void vq_bwd_kernel(const int32_t* __restrict__ idx, const T* __restrict__ dq,
                                                     float* __restrict__ dcb, int64_t rows, int G, int Nc, int gd) {
  const int64_t total = rows * G * gd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t rg = i / gd;
    const int d = (int)(i % gd);
    const int g = (int)(rg % G);
    atomicAdd(dcb + ((int64_t)g * Nc + idx[rg]) * gd + d, to_f32(dq[i]));
  }
}

// ---------------------------------------------------------------- contrastive cross-entropy
// S [B][T][T] fp32 = <h_t, q_t'> (all pairs).  Row (b,t): logits = [S[t][t], S[t][neg[b][n]]...] / temp,
// loss = logsumexp - logit0; S row is replaced by dloss/dS (scaled by grad_scale).
__global__ __launch_bounds__(128) void contrastive_kernel(float* __restrict__ S, const int32_t* __restrict__ neg,
                                                          int64_t neg_sb, int64_t neg_st,
                                                          float* __restrict__ row_loss, int Tn, int Nn,
                                                          float inv_temp, float grad_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = reinterpret_cast<float*>(smem);  // [Tn] gradient row
  float* lg = acc + Tn;                         // [Nn+1] logits
  __shared__ float red[2];
  const int64_t row = blockIdx.x;
  const int b = (int)(row / Tn), t = (int)(row % Tn);
  float* Sr = S + row * Tn;
  const int32_t* nb = neg + (int64_t)b * neg_sb + (int64_t)t * neg_st;
  for (int i = threadIdx.x; i < Tn; i += 128) acc[i] = 0.f;
  for (int i = threadIdx.x; i <= Nn; i += 128) lg[i] = (i == 0 ? Sr[t] : Sr[nb[i - 1]]) * inv_temp;
  __syncthreads();
  float mx = -INFINITY;
  for (int i = threadIdx.x; i <= Nn; i += 128) mx = fmaxf(mx, lg[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(red[0], red[1]);
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i <= Nn; i += 128) sum += expf(lg[i] - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = red[0] + red[1];
  if (threadIdx.x == 0) row_loss[row] = mx + logf(sum) - lg[0];
  const float inv = 1.0f / sum;
  // scatter d loss / d logits back onto the S row (indices may repeat or equal t): LDS atomics
  for (int i = threadIdx.x; i <= Nn; i += 128) {
    float gl = expf(lg[i] - mx) * inv;
    if (i == 0) gl -= 1.0f;
    atomicAdd(&acc[i == 0 ? t : nb[i - 1]], gl * inv_temp * grad_scale);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Tn; i += 128) Sr[i] = acc[i];
}

// ---------------------------------------------------------------- clipping
// per-segment sum of squares: seg_off[nseg+1] element offsets into g
__global__ __launch_bounds__(256) void segment_sumsq_kernel(const float* __restrict__ g,
                                                            const int64_t* __restrict__ seg_off,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  const int s = blockIdx.x;
  const int64_t lo = seg_off[s], hi = seg_off[s + 1];
  const int64_t n = hi - lo;
  const int64_t per = ((n + gridDim.y - 1) / gridDim.y + 3) / 4 * 4;  // whole float4s (segments start 16-B aligned)
  const int64_t a = lo + per * blockIdx.y, b = min(hi, a + per);
  float acc = 0.f;
  const bool vec = (lo & 3) == 0;
  const int64_t nv = vec && b > a ? (b - a) / 4 : 0;
  const f32x4* gv = reinterpret_cast<const f32x4*>(g + a);
  int64_t i = threadIdx.x;
  for (; i + 768 < nv; i += 1024) {  // four 16-byte loads in flight per thread
    const f32x4 v0 = gv[i], v1 = gv[i + 256], v2 = gv[i + 512], v3 = gv[i + 768];
    acc += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
    acc += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    acc += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
    acc += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
  }
  for (; i < nv; i += 256) {
    const f32x4 v = gv[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (int64_t j = a + nv * 4 + threadIdx.x; j < b; j += 256) {
    const float v = g[j];
    acc += v * v;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0 && a < b) atomicAdd(out + s, acc);
}

// The same sums over the chunk table of tmi_adam_step_segments ([nchunks][3] = lo, hi, segment; chunks of <= 8192 elements
// starting 16-byte aligned): a grid-stride walk over equal pieces of work instead of one (segment x slice) block each - the
// variables range from 512 to 2.4 M elements, and 16 slices of a bias vector are 16 idle workgroups.
__global__ __launch_bounds__(256) void segment_sumsq_chunks_kernel(const float* __restrict__ g, const int64_t* __restrict__ chunks,
                                                                   int64_t nchunks, float* __restrict__ out) {
  __shared__ float red[4];
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t lo = chunks[3 * c], hi = chunks[3 * c + 1];
    const int seg = (int)chunks[3 * c + 2];
    const int64_t nv = ((lo & 3) == 0) ? (hi - lo) / 4 : 0;
    const f32x4* gv = reinterpret_cast<const f32x4*>(g + lo);
    float acc = 0.f;
    int64_t i = threadIdx.x;
    for (; i + 768 < nv; i += 1024) {
      const f32x4 v0 = gv[i], v1 = gv[i + 256], v2 = gv[i + 512], v3 = gv[i + 768];
      acc += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
      acc += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
      acc += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
      acc += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
    }
    for (; i < nv; i += 256) {
      const f32x4 v = gv[i];
      acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int64_t j = lo + nv * 4 + threadIdx.x; j < hi; j += 256) {
      const float v = g[j];
      acc += v * v;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(out + seg, acc);
    __syncthreads();  // red[] is reused by the next chunk
  }
}

// g[i] *= clip / max(sqrt(sumsq[seg(i)] ), clip)     (tf.clip_by_norm per segment; one segment = global norm)
__global__ __launch_bounds__(256) void segment_clip_kernel(float* __restrict__ g, const int64_t* __restrict__ seg_off,
                                                           const float* __restrict__ sumsq, float clip) {
  const int s = blockIdx.x;
  const int64_t lo = seg_off[s], hi = seg_off[s + 1];
  const float scale = clip / fmaxf(sqrtf(sumsq[s]), clip);
  if (scale == 1.0f) return;
  const int64_t n = hi - lo;
  const int64_t per = ((n + gridDim.y - 1) / gridDim.y + 3) / 4 * 4;
  const int64_t a = lo + per * blockIdx.y, b = min(hi, a + per);
  const bool vec = (lo & 3) == 0;
  const int64_t nv = vec && b > a ? (b - a) / 4 : 0;
  f32x4* gv = reinterpret_cast<f32x4*>(g + a);
  int64_t i = threadIdx.x;
  for (; i + 768 < nv; i += 1024) {
    f32x4 v0 = gv[i], v1 = gv[i + 256], v2 = gv[i + 512], v3 = gv[i + 768];
    v0 *= scale; v1 *= scale; v2 *= scale; v3 *= scale;
    gv[i] = v0; gv[i + 256] = v1; gv[i + 512] = v2; gv[i + 768] = v3;
  }
  for (; i < nv; i += 256) {
    f32x4 v = gv[i];
    v *= scale;
    gv[i] = v;
  }
  for (int64_t j = a + nv * 4 + threadIdx.x; j < b; j += 256) g[j] *= scale;
}

// out[0] = (isnan(a) ? 0 : a + w * b) * scale     (V:1220-1231: loss assembly on the device)
__global__ void loss_combine_kernel(const float* a, const float* b, float w, float scale, float* out) {
  float v = a[0] + w * b[0];
  if (v != v) v = 0.f;
  out[0] = v * scale;
}

template <typename T> struct Launch {
  static constexpr int VEC = 16 / sizeof(T);
};

}  // namespace

#define DISPATCH_T(dtype, EXPR_BF16, EXPR_F32)       \
  if ((dtype) == TMI_BF16) { EXPR_BF16; }            \
  else if ((dtype) == TMI_F32) { EXPR_F32; }         \
  else return TMI_ERR_UNSUPPORTED;

extern "C" int64_t tmi_groupnorm_chunks(int64_t T) {
  // row chunks per sample.  At most 32: every chunk's workgroup ends with one fp32 atomic per channel into
  // dgamma / dbeta, and same-address atomics serialise (~20 ns each); B * 32 workgroups still fill the chip
  int64_t n = (T + 63) / 64;
  if (n > 32) n = 32;
  return n < 1 ? 1 : n;
}

static int gn_check(const void* x, int64_t B, int64_t T, int64_t C, int64_t G, int32_t dtype) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!x || B <= 0 || B > 65535 || T <= 0 || C <= 0 || G <= 0 || G > 256 || C % G || (C / G) % vec || C % vec ||
      (256 * vec) % C || !al16(x))
    return 0;
  return 1;
}

static int tmi_groupnorm_gelu_fwd_impl(const void* x, int64_t x_sb, const float* gamma, const float* beta, void* y,
                                      int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C,
                                      int64_t G, float eps, int32_t dtype, void* stream);
extern "C" int tmi_groupnorm_gelu_fwd(const void* x, int64_t x_sb, const float* gamma, const float* beta, void* y,
                                      int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C,
                                      int64_t G, float eps, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_groupnorm_gelu_fwd(x, x_sb, gamma, beta, y, y_sb, stats, part, B, T, C, G, eps, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_groupnorm_gelu_fwd_impl(x, x_sb, gamma, beta, y, y_sb, stats, part, B, T, C, G, eps, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_groupnorm_gelu_fwd_impl(const void* x, int64_t x_sb, const float* gamma, const float* beta, void* y,
                                      int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C,
                                      int64_t G, float eps, int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!gn_check(x, B, T, C, G, dtype) || !gamma || !beta || !y || !stats || !part || !al16(y) || x_sb % vec ||
      y_sb % vec) {
    tmi_set_error("tmi_groupnorm_gelu_fwd: bad argument (C/G and 2048/C (bf16) must be whole 16-byte chunks)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (int)tmi_groupnorm_chunks(T);
  const int rpc = (int)((T + nch - 1) / nch);
  dim3 gp((unsigned)nch, (unsigned)B);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((gn_partial_kernel<bf16_t, 0>), gp, dim3(256), 0, s, (const bf16_t*)x, x_sb, (const bf16_t*)nullptr, (int64_t)0, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch),
             hipLaunchKernelGGL((gn_partial_kernel<float, 0>), gp, dim3(256), 0, s, (const float*)x, x_sb, (const float*)nullptr, (int64_t)0, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch));
  // (the fold of the chunk partials happens inside the apply kernel: gn_fold_block)
  const double cnt = (double)T * (double)(C / G);
  int64_t blocks = (T * C / vec + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  dim3 ga((unsigned)blocks, (unsigned)B);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gn_apply_fwd_kernel<bf16_t>, ga, dim3(256), 0, s, (const bf16_t*)x, x_sb, gamma, beta, stats, (bf16_t*)y, y_sb, (int)T, (int)C, (int)G, part, nch, cnt, eps),
             hipLaunchKernelGGL(gn_apply_fwd_kernel<float>, ga, dim3(256), 0, s, (const float*)x, x_sb, gamma, beta, stats, (float*)y, y_sb, (int)T, (int)C, (int)G, part, nch, cnt, eps));
  return tmi_check_launch("tmi_groupnorm_gelu_fwd");
}

static int tmi_groupnorm_gelu_bwd_impl(const void* x, int64_t x_sb, const void* dy, int64_t dy_sb, const float* gamma,
                                      const float* beta, const float* stats, void* dx, int64_t dx_sb, float* dgamma,
                                      float* dbeta, float* part, float* sums, int64_t B, int64_t T, int64_t C, int64_t G,
                                      int32_t dtype, void* stream);
extern "C" int tmi_groupnorm_gelu_bwd(const void* x, int64_t x_sb, const void* dy, int64_t dy_sb, const float* gamma,
                                      const float* beta, const float* stats, void* dx, int64_t dx_sb, float* dgamma,
                                      float* dbeta, float* part, float* sums, int64_t B, int64_t T, int64_t C, int64_t G,
                                      int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_groupnorm_gelu_bwd(x, x_sb, dy, dy_sb, gamma, beta, stats, dx, dx_sb, dgamma, dbeta, part, sums, B, T, C, G, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_groupnorm_gelu_bwd_impl(x, x_sb, dy, dy_sb, gamma, beta, stats, dx, dx_sb, dgamma, dbeta, part, sums, B, T, C, G, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_groupnorm_gelu_bwd_impl(const void* x, int64_t x_sb, const void* dy, int64_t dy_sb, const float* gamma,
                                      const float* beta, const float* stats, void* dx, int64_t dx_sb, float* dgamma,
                                      float* dbeta, float* part, float* sums, int64_t B, int64_t T, int64_t C, int64_t G,
                                      int32_t dtype, void* stream) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!gn_check(x, B, T, C, G, dtype) || !dy || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || !part ||
      !sums || !al16(dy) || !al16(dx) || x_sb % vec || dy_sb % vec || dx_sb % vec) {
    tmi_set_error("tmi_groupnorm_gelu_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (int)tmi_groupnorm_chunks(T);
  const int rpc = (int)((T + nch - 1) / nch);
  dim3 gp((unsigned)nch, (unsigned)B);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((gn_partial_kernel<bf16_t, 1>), gp, dim3(256), 0, s, (const bf16_t*)x, x_sb, (const bf16_t*)dy, dy_sb, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch),
             hipLaunchKernelGGL((gn_partial_kernel<float, 1>), gp, dim3(256), 0, s, (const float*)x, x_sb, (const float*)dy, dy_sb, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch));
  const double cnt = (double)T * (double)(C / G);
  const int cpr = (int)C / vec;
  const int rstep = 256 / cpr > 0 ? 256 / cpr : 1;
  const size_t lds = (size_t)rstep * 2 * C * sizeof(float);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gn_apply_bwd_kernel<bf16_t>, gp, dim3(256), lds, s, (const bf16_t*)x, x_sb, (const bf16_t*)dy, dy_sb, gamma, beta, stats, sums, (bf16_t*)dx, dx_sb, dgamma, dbeta, (int)T, (int)C, (int)G, rpc, part, nch, cnt),
             hipLaunchKernelGGL(gn_apply_bwd_kernel<float>, gp, dim3(256), lds, s, (const float*)x, x_sb, (const float*)dy, dy_sb, gamma, beta, stats, sums, (float*)dx, dx_sb, dgamma, dbeta, (int)T, (int)C, (int)G, rpc, part, nch, cnt));
  return tmi_check_launch("tmi_groupnorm_gelu_bwd");
}

// ---- conv layer 0 as an FIR fused with GroupNorm + GELU (see the kernels above)
static int fir_check(const float* audio, const float* w, int64_t B, int64_t Tin, int64_t T, int64_t C, int64_t G, int64_t k,
                     int64_t stride) {
  return audio && w && B > 0 && B <= 65535 && Tin > 0 && T > 0 && C > 0 && C <= 2048 && C % 8 == 0 && 2048 % C == 0 && G > 0 &&
         G <= 256 && C % G == 0 && (C / G) % 8 == 0 && k == 10 && stride == 5 && al16(w);
}
// chunks per sample of the FIR kernels: ~100 time steps each (the kernels are VALU-bound: several waves per SIMD)
extern "C" int64_t tmi_fir_chunks(int64_t T) {
  int64_t n = (T + 99) / 100;
  if (n > 64) n = 64;
  return n < 1 ? 1 : n;
}
extern "C" int64_t tmi_groupnorm_chunks(int64_t T);
extern "C" int64_t tmi_fir_gn_workspace_floats(int64_t B, int64_t T, int64_t C) {
  // per-workgroup partials of dW (10 x C), dgamma, dbeta.  The backward apply pass that writes them runs on
  // tmi_groupnorm_chunks(T) chunks per sample (see tmi_fir_groupnorm_gelu_bwd), which exceeds tmi_fir_chunks(T) for short
  // clips (T < ~3200): size for the larger of the two
  const int64_t a = tmi_fir_chunks(T), b = tmi_groupnorm_chunks(T);
  return B * (a > b ? a : b) * 12 * C;
}

static int tmi_fir_groupnorm_gelu_fwd_impl(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const float* gamma, const float* beta, void* y,
                                          int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C, int64_t G,
                                          float eps, int32_t dtype, void* stream);
extern "C" int tmi_fir_groupnorm_gelu_fwd(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const float* gamma, const float* beta, void* y,
                                          int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C, int64_t G,
                                          float eps, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_fir_groupnorm_gelu_fwd(audio, a_sb, Tin, pad_left, w, k, stride, gamma, beta, y, y_sb, stats, part, B, T, C, G, eps, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_fir_groupnorm_gelu_fwd_impl(audio, a_sb, Tin, pad_left, w, k, stride, gamma, beta, y, y_sb, stats, part, B, T, C, G, eps, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_fir_groupnorm_gelu_fwd_impl(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const float* gamma, const float* beta, void* y,
                                          int64_t y_sb, float* stats, float* part, int64_t B, int64_t T, int64_t C, int64_t G,
                                          float eps, int32_t dtype, void* stream) {
  if (!fir_check(audio, w, B, Tin, T, C, G, k, stride) || !gamma || !beta || !y || !stats || !part || !al16(y) || y_sb % 8 ||
      pad_left < 0) {
    tmi_set_error("tmi_fir_groupnorm_gelu_fwd: bad argument (kernel 10, stride 5, C % 8 == 0, 2048 % C == 0, (C/G) % 8 == 0)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (int)tmi_fir_chunks(T);
  const int rpc = (int)((T + nch - 1) / nch);
  const size_t lds = (size_t)(rpc * 5 + 10) * sizeof(float);
  dim3 gp((unsigned)nch, (unsigned)B);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((fir_gn_partial_kernel<bf16_t, 10, 5, 0>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const bf16_t*)nullptr, (int64_t)0, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch),
             hipLaunchKernelGGL((fir_gn_partial_kernel<float, 10, 5, 0>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const float*)nullptr, (int64_t)0, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch));
  const double cnt = (double)T * (double)(C / G);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((fir_gn_apply_fwd_kernel<bf16_t, 10, 5>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, gamma, beta, stats, (bf16_t*)y, y_sb, (int)T, (int)C, (int)G, rpc, part, nch, cnt, eps),
             hipLaunchKernelGGL((fir_gn_apply_fwd_kernel<float, 10, 5>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, gamma, beta, stats, (float*)y, y_sb, (int)T, (int)C, (int)G, rpc, part, nch, cnt, eps));
  return tmi_check_launch("tmi_fir_groupnorm_gelu_fwd");
}

static int tmi_fir_groupnorm_gelu_bwd_impl(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const void* dy, int64_t dy_sb, const float* gamma,
                                          const float* beta, const float* stats, float* dW, float* dgamma, float* dbeta,
                                          float* part, float* sums, float* wpart, int64_t B, int64_t T, int64_t C, int64_t G,
                                          int32_t dtype, void* stream);
extern "C" int tmi_fir_groupnorm_gelu_bwd(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const void* dy, int64_t dy_sb, const float* gamma,
                                          const float* beta, const float* stats, float* dW, float* dgamma, float* dbeta,
                                          float* part, float* sums, float* wpart, int64_t B, int64_t T, int64_t C, int64_t G,
                                          int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_fir_groupnorm_gelu_bwd(audio, a_sb, Tin, pad_left, w, k, stride, dy, dy_sb, gamma, beta, stats, dW, dgamma, dbeta, part, sums, wpart, B, T, C, G, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_fir_groupnorm_gelu_bwd_impl(audio, a_sb, Tin, pad_left, w, k, stride, dy, dy_sb, gamma, beta, stats, dW, dgamma, dbeta, part, sums, wpart, B, T, C, G, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_fir_groupnorm_gelu_bwd_impl(const float* audio, int64_t a_sb, int64_t Tin, int64_t pad_left, const float* w,
                                          int64_t k, int64_t stride, const void* dy, int64_t dy_sb, const float* gamma,
                                          const float* beta, const float* stats, float* dW, float* dgamma, float* dbeta,
                                          float* part, float* sums, float* wpart, int64_t B, int64_t T, int64_t C, int64_t G,
                                          int32_t dtype, void* stream) {
  if (!fir_check(audio, w, B, Tin, T, C, G, k, stride) || !dy || !gamma || !beta || !stats || !dW || !dgamma || !dbeta ||
      !part || !sums || !wpart || !al16(dy) || dy_sb % 8 || pad_left < 0) {
    tmi_set_error("tmi_fir_groupnorm_gelu_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (int)tmi_fir_chunks(T);
  const int rpc = (int)((T + nch - 1) / nch);
  const int nwin = rpc * 5 + 10;
  const size_t lds = (size_t)nwin * sizeof(float);
  dim3 gp((unsigned)nch, (unsigned)B);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((fir_gn_partial_kernel<bf16_t, 10, 5, 1>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const bf16_t*)dy, dy_sb, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch),
             hipLaunchKernelGGL((fir_gn_partial_kernel<float, 10, 5, 1>), gp, dim3(256), lds, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const float*)dy, dy_sb, gamma, beta, stats, part, (int)T, (int)C, (int)G, rpc, nch));
  const double cnt = (double)T * (double)(C / G);
  // the apply pass ends with a 12-quantity fold per workgroup and holds 256 registers: coarser chunks (one workgroup
  // per CU at base size) measured faster than the finer ones the other passes use (71 vs 84 us)
  const int nchA = (int)tmi_groupnorm_chunks(T);
  const int rpcA = (int)((T + nchA - 1) / nchA);
  const int nwinA = rpcA * 5 + 10;
  dim3 gpA((unsigned)nchA, (unsigned)B);
  const int rstep = 256 / (int)(C / 8);
  const size_t lds2 = (size_t)(((nwinA + 3) & ~3) + rstep * C) * sizeof(float);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((fir_gn_apply_bwd_kernel<bf16_t, 10, 5>), gpA, dim3(256), lds2, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const bf16_t*)dy, dy_sb, gamma, beta, stats, sums, wpart, (int)T, (int)C, (int)G, rpcA, nchA, part, nch, cnt),
             hipLaunchKernelGGL((fir_gn_apply_bwd_kernel<float, 10, 5>), gpA, dim3(256), lds2, s, audio, a_sb, (int)Tin, (int)pad_left, w, (const float*)dy, dy_sb, gamma, beta, stats, sums, wpart, (int)T, (int)C, (int)G, rpcA, nchA, part, nch, cnt));
  const int64_t rb = (12 * C + 255) / 256;
  const int64_t ry = B * nchA >= 64 ? 8 : 1;
  hipLaunchKernelGGL(fir_reduce_kernel, dim3((unsigned)rb, (unsigned)ry), dim3(256), 0, s, wpart, (int)(B * nchA), (int64_t)10 * C, C, dW, dgamma, dbeta);
  return tmi_check_launch("tmi_fir_groupnorm_gelu_bwd");
}

static int tmi_group_pack_impl(const void* x, void* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t Tp,
                              int64_t pad_left, int32_t dtype, void* stream);
extern "C" int tmi_group_pack(const void* x, void* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t Tp,
                              int64_t pad_left, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_group_pack(x, xg, B, T, C, G, Tp, pad_left, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_group_pack_impl(x, xg, B, T, C, G, Tp, pad_left, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_group_pack_impl(const void* x, void* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t Tp,
                              int64_t pad_left, int32_t dtype, void* stream) {
  if (!x || !xg || B <= 0 || T <= 0 || C <= 0 || G <= 0 || C % G || Tp < T + pad_left || pad_left < 0) {
    tmi_set_error("tmi_group_pack: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t R = B * Tp;
  int64_t blocks = (R * C + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(group_pack_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)xg, R, (int)C, (int)G, Tp, pad_left, T),
             hipLaunchKernelGGL(group_pack_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (float*)xg, R, (int)C, (int)G, Tp, pad_left, T));
  return tmi_check_launch("tmi_group_pack");
}

static int tmi_group_unpack_impl(const void* yg, const float* bias, const void* resid, void* out, int64_t B, int64_t T,
                                int64_t C, int64_t G, int64_t Tp, int64_t row_off, int32_t dtype, void* stream);
extern "C" int tmi_group_unpack(const void* yg, const float* bias, const void* resid, void* out, int64_t B, int64_t T,
                                int64_t C, int64_t G, int64_t Tp, int64_t row_off, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_group_unpack(yg, bias, resid, out, B, T, C, G, Tp, row_off, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_group_unpack_impl(yg, bias, resid, out, B, T, C, G, Tp, row_off, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_group_unpack_impl(const void* yg, const float* bias, const void* resid, void* out, int64_t B, int64_t T,
                                int64_t C, int64_t G, int64_t Tp, int64_t row_off, int32_t dtype, void* stream) {
  if (!yg || !out || B <= 0 || T <= 0 || C <= 0 || G <= 0 || C % G || row_off < 0 || Tp < T + row_off) {
    tmi_set_error("tmi_group_unpack: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = (B * T * C + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(group_unpack_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)yg, bias, (const bf16_t*)resid, (bf16_t*)out, B * Tp, (int)C, (int)G, Tp, row_off, T, B),
             hipLaunchKernelGGL(group_unpack_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)yg, bias, (const float*)resid, (float*)out, B * Tp, (int)C, (int)G, Tp, row_off, T, B));
  return tmi_check_launch("tmi_group_unpack");
}

static int tmi_posconv_pack_weights_impl(const float* w, void* wf, void* wb, int64_t k, int64_t Cg, int64_t G,
                                        int32_t dtype, void* stream);
extern "C" int tmi_posconv_pack_weights(const float* w, void* wf, void* wb, int64_t k, int64_t Cg, int64_t G,
                                        int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_posconv_pack_weights(w, wf, wb, k, Cg, G, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_posconv_pack_weights_impl(w, wf, wb, k, Cg, G, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_posconv_pack_weights_impl(const float* w, void* wf, void* wb, int64_t k, int64_t Cg, int64_t G,
                                        int32_t dtype, void* stream) {
  if (!w || !wf || !wb || k <= 0 || Cg <= 0 || G <= 0) {
    tmi_set_error("tmi_posconv_pack_weights: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = (k * Cg * Cg * G + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(posconv_pack_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, w, (bf16_t*)wf, (bf16_t*)wb, (int)k, (int)Cg, (int)G),
             hipLaunchKernelGGL(posconv_pack_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, w, (float*)wf, (float*)wb, (int)k, (int)Cg, (int)G));
  return tmi_check_launch("tmi_posconv_pack_weights");
}

static int tmi_vq_nearest_impl(const void* h, const float* codebook, int32_t* idx, void* q, float* perplexity,
                              int64_t rows, int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream);
extern "C" int tmi_vq_nearest(const void* h, const float* codebook, int32_t* idx, void* q, float* perplexity,
                              int64_t rows, int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_vq_nearest(h, codebook, idx, q, perplexity, rows, G, Nc, gd, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_vq_nearest_impl(h, codebook, idx, q, perplexity, rows, G, Nc, gd, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_vq_nearest_impl(const void* h, const float* codebook, int32_t* idx, void* q, float* perplexity,
                              int64_t rows, int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream) {
  if (!h || !codebook || !idx || !q || !perplexity || rows <= 0 || G <= 0 || Nc <= 0 || gd <= 0 || gd > 1024 ||
      G * Nc > 8192) {
    tmi_set_error("tmi_vq_nearest: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows * G + 3) / 4));
  const size_t lds = (size_t)4 * gd * sizeof(float);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(vq_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)h, codebook, idx, (bf16_t*)q, rows, (int)G, (int)Nc, (int)gd),
             hipLaunchKernelGGL(vq_kernel<float>, grid, dim3(256), lds, s, (const float*)h, codebook, idx, (float*)q, rows, (int)G, (int)Nc, (int)gd));
  hipLaunchKernelGGL(vq_perplexity_kernel, dim3(1), dim3(256), (size_t)G * Nc * sizeof(int), s, idx, perplexity, rows,
                     (int)G, (int)Nc);
  return tmi_check_launch("tmi_vq_nearest");
}

static int tmi_vq_assign_impl(const float* codebook, const int32_t* idx, void* q, float* perplexity, int64_t rows,
                             int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream);
extern "C" int tmi_vq_assign(const float* codebook, const int32_t* idx, void* q, float* perplexity, int64_t rows,
                             int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_vq_assign(codebook, idx, q, perplexity, rows, G, Nc, gd, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_vq_assign_impl(codebook, idx, q, perplexity, rows, G, Nc, gd, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_vq_assign_impl(const float* codebook, const int32_t* idx, void* q, float* perplexity, int64_t rows,
                             int64_t G, int64_t Nc, int64_t gd, int32_t dtype, void* stream) {
  if (!codebook || !idx || !q || !perplexity || rows <= 0 || G <= 0 || Nc <= 0 || gd <= 0 || G * Nc > 8192) {
    tmi_set_error("tmi_vq_assign: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows * G + 3) / 4));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(vq_assign_kernel<bf16_t>, grid, dim3(256), 0, s, codebook, idx, (bf16_t*)q, rows, (int)G, (int)Nc, (int)gd),
             hipLaunchKernelGGL(vq_assign_kernel<float>, grid, dim3(256), 0, s, codebook, idx, (float*)q, rows, (int)G, (int)Nc, (int)gd));
  hipLaunchKernelGGL(vq_perplexity_kernel, dim3(1), dim3(256), (size_t)G * Nc * sizeof(int), s, idx, perplexity, rows,
                     (int)G, (int)Nc);
  return tmi_check_launch("tmi_vq_assign");
}

static int tmi_vq_bwd_impl(const int32_t* idx, const void* dq, float* dcodebook, int64_t rows, int64_t G, int64_t Nc,
                          int64_t gd, int32_t dtype, void* stream);
extern "C" int tmi_vq_bwd(const int32_t* idx, const void* dq, float* dcodebook, int64_t rows, int64_t G, int64_t Nc,
                          int64_t gd, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_vq_bwd(idx, dq, dcodebook, rows, G, Nc, gd, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_vq_bwd_impl(idx, dq, dcodebook, rows, G, Nc, gd, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_vq_bwd_impl(const int32_t* idx, const void* dq, float* dcodebook, int64_t rows, int64_t G, int64_t Nc,
                          int64_t gd, int32_t dtype, void* stream) {
  if (!idx || !dq || !dcodebook || rows <= 0 || G <= 0 || Nc <= 0 || gd <= 0) {
    tmi_set_error("tmi_vq_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = (rows * G * gd + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(vq_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, idx, (const bf16_t*)dq, dcodebook, rows, (int)G, (int)Nc, (int)gd),
             hipLaunchKernelGGL(vq_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, idx, (const float*)dq, dcodebook, rows, (int)G, (int)Nc, (int)gd));
  return tmi_check_launch("tmi_vq_bwd");
}

static int tmi_contrastive_fwd_bwd_impl(float* S, const int32_t* neg, int64_t neg_sb, int64_t neg_st, float* row_loss,
                                       int64_t B, int64_t T, int64_t Nn, float temperature, float grad_scale,
                                       void* stream);
extern "C" int tmi_contrastive_fwd_bwd(float* S, const int32_t* neg, int64_t neg_sb, int64_t neg_st, float* row_loss,
                                       int64_t B, int64_t T, int64_t Nn, float temperature, float grad_scale,
                                       void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_contrastive_fwd_bwd(S, neg, neg_sb, neg_st, row_loss, B, T, Nn, temperature, grad_scale, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_contrastive_fwd_bwd_impl(S, neg, neg_sb, neg_st, row_loss, B, T, Nn, temperature, grad_scale, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_contrastive_fwd_bwd_impl(float* S, const int32_t* neg, int64_t neg_sb, int64_t neg_st, float* row_loss,
                                       int64_t B, int64_t T, int64_t Nn, float temperature, float grad_scale,
                                       void* stream) {
  if (!S || !neg || !row_loss || B <= 0 || T <= 0 || Nn < 0 || T > 8192 || Nn > 4096 || temperature <= 0.f ||
      neg_sb < 0 || neg_st < 0) {
    tmi_set_error("tmi_contrastive_fwd_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  const size_t lds = (size_t)(T + Nn + 1) * sizeof(float);
  hipLaunchKernelGGL(contrastive_kernel, dim3((unsigned)(B * T)), dim3(128), lds, reinterpret_cast<hipStream_t>(stream),
                     S, neg, neg_sb, neg_st, row_loss, (int)T, (int)Nn, 1.0f / temperature, grad_scale);
  return tmi_check_launch("tmi_contrastive_fwd_bwd");
}

static int tmi_segment_sumsq_impl(const float* g, const int64_t* seg_off, float* out, int64_t nseg, void* stream);
extern "C" int tmi_segment_sumsq(const float* g, const int64_t* seg_off, float* out, int64_t nseg, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_segment_sumsq(g, seg_off, out, nseg, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_segment_sumsq_impl(g, seg_off, out, nseg, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_segment_sumsq_impl(const float* g, const int64_t* seg_off, float* out, int64_t nseg, void* stream) {
  if (!g || !seg_off || !out || nseg <= 0 || nseg > 65535) {
    tmi_set_error("tmi_segment_sumsq: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)nseg * sizeof(float), s) != hipSuccess) return TMI_ERR_LAUNCH;
  const unsigned split = nseg == 1 ? 512u : 16u;  // (every block ends with an atomic on out[s])
  hipLaunchKernelGGL(segment_sumsq_kernel, dim3((unsigned)nseg, split), dim3(256), 0, s, g, seg_off, out);
  return tmi_check_launch("tmi_segment_sumsq");
}

static int tmi_segment_sumsq_chunks_impl(const float* g, const int64_t* chunks, int64_t nchunks, float* out, int64_t nseg,
                                        void* stream);
extern "C" int tmi_segment_sumsq_chunks(const float* g, const int64_t* chunks, int64_t nchunks, float* out, int64_t nseg,
                                        void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_segment_sumsq_chunks(g, chunks, nchunks, out, nseg, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_segment_sumsq_chunks_impl(g, chunks, nchunks, out, nseg, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_segment_sumsq_chunks_impl(const float* g, const int64_t* chunks, int64_t nchunks, float* out, int64_t nseg,
                                        void* stream) {
  if (!g || !chunks || !out || nchunks <= 0 || nseg <= 0) {
    tmi_set_error("tmi_segment_sumsq_chunks: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out, 0, (size_t)nseg * sizeof(float), s) != hipSuccess) return TMI_ERR_LAUNCH;
  hipLaunchKernelGGL(segment_sumsq_chunks_kernel, dim3((unsigned)(nchunks < 2048 ? nchunks : 2048)), dim3(256), 0, s, g, chunks,
                     nchunks, out);
  return tmi_check_launch("tmi_segment_sumsq_chunks");
}

static int tmi_segment_clip_impl(float* g, const int64_t* seg_off, const float* sumsq, int64_t nseg, float clip,
                                void* stream);
extern "C" int tmi_segment_clip(float* g, const int64_t* seg_off, const float* sumsq, int64_t nseg, float clip,
                                void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_segment_clip(g, seg_off, sumsq, nseg, clip, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_segment_clip_impl(g, seg_off, sumsq, nseg, clip, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_segment_clip_impl(float* g, const int64_t* seg_off, const float* sumsq, int64_t nseg, float clip,
                                void* stream) {
  if (!g || !seg_off || !sumsq || nseg <= 0 || nseg > 65535 || clip <= 0.f) {
    tmi_set_error("tmi_segment_clip: bad argument");
    return TMI_ERR_INVALID;
  }
  const unsigned split = nseg == 1 ? 512u : 16u;
  hipLaunchKernelGGL(segment_clip_kernel, dim3((unsigned)nseg, split), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), g, seg_off, sumsq, clip);
  return tmi_check_launch("tmi_segment_clip");
}

static int tmi_loss_combine_impl(const float* a, const float* b, float w, float scale, float* out, void* stream);
extern "C" int tmi_loss_combine(const float* a, const float* b, float w, float scale, float* out, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_loss_combine(a, b, w, scale, out, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_loss_combine_impl(a, b, w, scale, out, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_loss_combine_impl(const float* a, const float* b, float w, float scale, float* out, void* stream) {
  if (!a || !b || !out) {
    tmi_set_error("tmi_loss_combine: bad argument");
    return TMI_ERR_INVALID;
  }
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), a, b, w, scale,
                     out);
  return tmi_check_launch("tmi_loss_combine");
}
