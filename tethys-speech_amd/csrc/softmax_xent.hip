// Row softmax (materialised-score fp32 attention path), shifted sparse cross-entropy with
// fused gradient, decoder embedding gather / gradient.  All HBM-bound row kernels.
// Reference call sites: speech_jobs/whisper_dist.py:147-167 (scores/mask/softmax),
// :585-600 (shifted SparseCategoricalCrossentropy + reduce_mean), :405-408 + :559-563
// (embedding of the right-shifted labels + positional encoding).
#include "tmi_common.h"

namespace {

constexpr int SM_E = 32;  // Tk <= 64*32 = 2048

__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ s, int64_t rows, int Tq, int Tk,
                                                          int mask_mode) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = s + row * Tk;
  const int qi = (int)(row % Tq);
  float v[SM_E];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SM_E; ++j) {
    const int c = j * 64 + lane;
    if (c < Tk) {
      float x = p[c];
      if (mask_mode == 1 && c <= qi) x = x + (-1e9f);  // fp32 add, as tf: score is absorbed
      v[j] = x;
      mx = fmaxf(mx, x);
    } else {
      v[j] = -INFINITY;
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < SM_E; ++j) {
    const int c = j * 64 + lane;
    const float e = (c < Tk) ? expf(v[j] - mx) : 0.f;
    v[j] = e;
    sum += e;
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int j = 0; j < SM_E; ++j) {
    const int c = j * 64 + lane;
    if (c < Tk) p[c] = v[j] * inv;
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                          int64_t rows, int Tk) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = p + row * Tk;
  float* dr = dp + row * Tk;
  float pv[SM_E], dv[SM_E];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < SM_E; ++j) {
    const int c = j * 64 + lane;
    pv[j] = (c < Tk) ? pr[c] : 0.f;
    dv[j] = (c < Tk) ? dr[c] : 0.f;
    dot += pv[j] * dv[j];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int j = 0; j < SM_E; ++j) {
    const int c = j * 64 + lane;
    if (c < Tk) dr[c] = pv[j] * (dv[j] - dot);
  }
}

// tmi_linear_xent: the operands of the LM head whose output the logits are.  bf16 logits lose the target logit's low bits
// (ulp 0.03 at |z| in [4, 8)), and the loss is lse - z_target: measured on the headline golden, that rounding alone is up to
// 6.5e-4 of the 7.7e-4 loss-curve error (profiles/r04_bf16_margin.txt).  With the operands at hand the row's workgroup
// recomputes z_target = x[row, :] . w[:, target] in fp32 (d multiply-adds) for the LOSS; the gradient is unchanged.
struct LmOperands {
  const bf16_t* x;   // [rows, d], row stride x_ld; nullptr = plain cross-entropy
  const bf16_t* w;   // element (k, n) at w[k * w_sk + n * w_sn]
  int64_t x_ld, w_sk, w_sn;
  int d;
};
__device__ __forceinline__ float lm_target_logit(const LmOperands& lm, int64_t row, int target, float* red) {
  float z = 0.f;
  for (int k = threadIdx.x; k < lm.d; k += 256) z = fmaf((float)lm.x[row * lm.x_ld + k], (float)lm.w[k * lm.w_sk + (int64_t)target * lm.w_sn], z);
  return block_sum_256(z, red);
}

// One 256-thread block per logits row.  Pass 1: online (max, sum); pass 2: write gradient.
template <typename T>
__global__ __launch_bounds__(256) void xent_kernel(T* __restrict__ logits, int64_t ld, const int32_t* __restrict__ labels,
                                                   float* __restrict__ row_loss, int S, int64_t V, float grad_scale, const LmOperands lm) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int b = (int)(row / S), t = (int)(row % S);
  T* lr = logits + row * ld;
  const int64_t nch = ld / VEC;  // ld is a multiple of VEC (host-checked)
  if (t >= S - 1) {  // unused row (W:586 drops the last position): zero gradient
    alignas(16) T z[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) z[i] = from_f32<T>(0.f);
    for (int64_t ch = threadIdx.x; ch < nch; ch += 256)
      reinterpret_cast<u32x4*>(lr)[ch] = *reinterpret_cast<const u32x4*>(z);
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  const int target = labels[(int64_t)b * S + t + 1];
  float mx = -INFINITY, sum = 0.f;
  for (int64_t ch = threadIdx.x; ch < nch; ch += 256) {
    const u32x4 raw = reinterpret_cast<const u32x4*>(lr)[ch];
    const T* e = reinterpret_cast<const T*>(&raw);
    float lm = -INFINITY;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (ch * VEC + i < V) lm = fmaxf(lm, to_f32(e[i]));
    if (lm > mx) {
      sum *= expf(mx - lm);  // exp(-inf) = 0 on first chunk
      mx = lm;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (ch * VEC + i < V) sum += expf(to_f32(e[i]) - mx);
  }
  const float gmx = block_max_256(mx, red);
  const float part = (mx == -INFINITY) ? 0.f : sum * expf(mx - gmx);
  const float gsum = block_sum_256(part, red);
  const float lse = gmx + logf(gsum);
  const float inv = 1.0f / gsum;
  float zt = 0.f;
  if (lm.x) zt = lm_target_logit(lm, row, target, red);  // (uniform over the block)
  if (threadIdx.x == 0) {
    // The loss's two terms must see the SAME target logit: with the fp32 recomputation z_t for the subtraction, the target's
    // term of the log-sum-exp is swapped for exp(z_t) as well - otherwise a row whose target dominates (lse ~ rounded
    // logit) keeps the rounding error of the stored logit, up to half a bf16 ulp, and can come out negative.
    // The gradient is untouched (`inv` is the sum over the stored logits).
    // Written as log1p(sum of the OTHER terms * exp(max - z_t)): non-negative by construction and free of the
    // max + log(sum) - z_t cancellation (three numbers of size |z| for a result that may be 1e-6).
    if (lm.x) {
      const float rest = fmaxf(gsum - expf(to_f32(lr[target]) - gmx), 0.f);
      const float dz = gmx - zt;
      row_loss[row] = dz < 80.f ? log1pf(rest * expf(dz)) : dz + logf(rest);
    } else {
      row_loss[row] = lse - to_f32(lr[target]);
    }
  }
  __syncthreads();  // the target logit is read before anyone overwrites it
  for (int64_t ch = threadIdx.x; ch < nch; ch += 256) {
    const u32x4 raw = reinterpret_cast<const u32x4*>(lr)[ch];
    const T* e = reinterpret_cast<const T*>(&raw);
    alignas(16) T o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int64_t c = ch * VEC + i;
      float g = 0.f;
      if (c < V) {
        g = expf(to_f32(e[i]) - gmx) * inv;
        if (c == target) g -= 1.0f;
        g *= grad_scale;
      }
      o[i] = from_f32<T>(g);
    }
    reinterpret_cast<u32x4*>(lr)[ch] = *reinterpret_cast<const u32x4*>(o);
  }
}

// bf16 rows short enough to be held on chip (ld <= 256 * 8 * XR elements: the 51904-column Whisper logits are 26 16-byte
// chunks per thread): ONE read of the row, base-2 exponentials (v_exp_f32; the output is bf16), one write.  The generic
// kernel above reads the row twice and spends most of its time in expf (two calls per element).
// Round 4: the row's chunks are split between registers (XREG per thread) and LDS (XLDS per thread, 36 KiB per workgroup).
// With all 26-28 chunks in registers the kernel needed 156 VGPRs = 3 workgroups per CU = 768 resident rows, and the step's
// 800 rows ran as a full round plus a 32-row tail (64 us); at <= 128 VGPRs and 36 KiB four workgroups fit: one round.
constexpr int XREG = 17, XLDS = 9, XR = XREG + XLDS;
__global__ __launch_bounds__(256, 4) void xent_rows_bf16_kernel(bf16_t* __restrict__ logits, int64_t ld, const int32_t* __restrict__ labels,
                                                                float* __restrict__ row_loss, int S, int64_t V, float grad_scale,
                                                                const LmOperands lm) {
  __shared__ float red[4];
  __shared__ u32x4 spill[XLDS * 256];  // chunk XREG + k of thread t at spill[k * 256 + t]
  const int64_t row = blockIdx.x;
  const int b = (int)(row / S), t = (int)(row % S);
  bf16_t* lr = logits + row * ld;
  const int nch = (int)(ld / 8);
  if (t >= S - 1) {
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
    for (int ch = threadIdx.x; ch < nch; ch += 256) reinterpret_cast<u32x4*>(lr)[ch] = z;
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  const int target = labels[(int64_t)b * S + t + 1];
  constexpr float L2E = 1.44269504088896340736f;
  u32x4 raw[XREG];
  float mx = -INFINITY;
  const int Vi = (int)V;
  // Loads first, unconditionally (a chunk past the row is fetched from the row's last chunk), then the padding columns
  // (>= V: the tail of one chunk and the chunks after it) are turned into -inf so that the passes below are branch-free.
  // The test is per chunk SLOT and wave-uniform - only the slots whose 256 chunks can reach V enter it - so no branch
  // sits between a load and the next (data-dependent tests there issued the row's 26 loads one at a time).
  auto load = [&](int ch) -> u32x4 { return reinterpret_cast<const u32x4*>(lr)[ch < nch ? ch : nch - 1]; };
  auto patch = [&](u32x4& r, int slot, int ch) {
    if ((slot * 256 + 256) * 8 > Vi) {
      const int nv = Vi - ch * 8;  // valid columns of this chunk (<= 0: none)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t keep = (2 * d < nv ? 0xffffu : 0u) | (2 * d + 1 < nv ? 0xffff0000u : 0u);
        r[d] = (r[d] & keep) | (0xff80ff80u & ~keep);  // bf16 -inf = 0xff80
      }
    }
  };
  auto max8 = [](const u32x4& r) -> float {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&r);
    float m8 = (float)e[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m8 = fmaxf(m8, (float)e[i]);
    return m8;
  };
  {  // (the LDS-bound chunks first and fenced off: hoisted above them, the XREG register loads would keep all 26 chunks live at once)
    u32x4 tmp[XLDS];
#pragma unroll
    for (int k = 0; k < XLDS; ++k) tmp[k] = load((XREG + k) * 256 + (int)threadIdx.x);
#pragma unroll
    for (int k = 0; k < XLDS; ++k) {
      patch(tmp[k], XREG + k, (XREG + k) * 256 + (int)threadIdx.x);
      spill[k * 256 + threadIdx.x] = tmp[k];  // (read back by the same thread only: no barrier needed)
      mx = fmaxf(mx, max8(tmp[k]));
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < XREG; ++j) raw[j] = load(j * 256 + (int)threadIdx.x);
#pragma unroll
  for (int j = 0; j < XREG; ++j) {
    patch(raw[j], j, j * 256 + (int)threadIdx.x);
    mx = fmaxf(mx, max8(raw[j]));
  }
  // (opaque between the passes: otherwise the bf16 -> fp32 unpacking is shared across them and 17 x 8 floats stay live)
  auto pin = [&]() {
#pragma unroll
    for (int j = 0; j < XREG; ++j) asm volatile("" : "+v"(raw[j]));
  };
  pin();
  const float gmx = block_max_256(mx, red);
  const float nb = -gmx * L2E;
  float sum = 0.f;
  auto sum8 = [&](const u32x4& r) {
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&r);
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += __builtin_amdgcn_exp2f(fmaf((float)e[i], L2E, nb));  // exp2(-inf) = 0 for the padding
  };
#pragma unroll
  for (int j = 0; j < XREG; ++j) {
    sum8(raw[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < XLDS; ++k) {
    if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);  // (three LDS reads in flight, not nine held beside raw[])
    sum8(spill[k * 256 + threadIdx.x]);
  }
  pin();
  const float gsum = block_sum_256(sum, red);
  const float inv = grad_scale / gsum;
  float zt = 0.f;
  if (lm.x) zt = lm_target_logit(lm, row, target, red);  // (uniform over the block)
  if (threadIdx.x == 0) {
    if (lm.x) {  // (the log-sum-exp with the target's term at the recomputed logit too: see xent_kernel)
      const float rest = fmaxf(gsum - __builtin_amdgcn_exp2f(fmaf((float)lr[target], L2E, nb)), 0.f);
      const float dz = gmx - zt;
      row_loss[row] = dz < 80.f ? log1pf(rest * expf(dz)) : dz + logf(rest);
    } else {
      row_loss[row] = gmx + logf(gsum) - (float)lr[target];
    }
  }
  __syncthreads();  // the target logit is read before anyone overwrites it
  const int tch = target >> 3, ti = target & 7;
  auto emit = [&](int ch, const u32x4& r) {
    if (ch < nch) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&r);
      float g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = __builtin_amdgcn_exp2f(fmaf((float)e[i], L2E, nb)) * inv;
      if (ch == tch) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i == ti) g[i] -= grad_scale;
      }
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (bf16_t)g[i];
      *reinterpret_cast<bf16x8*>(lr + (int64_t)ch * 8) = o;
    }
  };
#pragma unroll
  for (int j = 0; j < XREG; ++j) {
    emit(j * 256 + (int)threadIdx.x, raw[j]);
    __builtin_amdgcn_sched_barrier(0);  // (chunk by chunk: interleaved, the scheduler keeps several chunks' fp32 values live)
  }
#pragma unroll
  for (int k = 0; k < XLDS; ++k) {
    __builtin_amdgcn_sched_barrier(0);
    emit((XREG + k) * 256 + (int)threadIdx.x, spill[k * 256 + threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        int64_t n, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

__device__ __forceinline__ int dec_id(const int32_t* labels, int64_t r, int S, int start_id) {
  const int t = (int)(r % S);
  return t == 0 ? start_id : labels[r - 1];  // labels[b, t-1]
}

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int32_t* __restrict__ labels, const float* __restrict__ table,
                                                        const float* __restrict__ pe, T* __restrict__ out, int S,
                                                        int D, int start_id) {
  const int64_t r = blockIdx.x;
  const int t = (int)(r % S);
  const int id = dec_id(labels, r, S, start_id);
  const float* src = table + (int64_t)id * D;
  const float* pr = pe + (int64_t)t * D;
  for (int c = threadIdx.x; c < D; c += 256) out[r * D + c] = from_f32<T>(src[c] + pr[c]);
}

// Block r owns token id(r) iff no earlier row has the same id; it then sums dy over every
// row with that id, in row order (deterministic).  The matching rows are compacted into an
// LDS list first so the column sums run 8 independent loads deep (the pad token owns ~40 %
// of all positions: a serial chain over them was the whole kernel's duration).
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ labels, const T* __restrict__ dy,
                                                        float* __restrict__ dtable, int n, int S, int D, int start_id) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* list = reinterpret_cast<int*>(smem);  // [n]
  __shared__ int dup, cnt;
  const int r = blockIdx.x;
  const int id = dec_id(labels, r, S, start_id);
  if (threadIdx.x == 0) { dup = 0; cnt = 0; }
  __syncthreads();
  for (int q = threadIdx.x; q < r; q += 256)
    if (dec_id(labels, q, S, start_id) == id) dup = 1;
  __syncthreads();
  if (dup) return;
  if (threadIdx.x < 64) {  // one wave builds the ordered list with ballot prefix sums
    int base = 0;
    for (int q0 = r; q0 < n; q0 += 64) {
      const int q = q0 + threadIdx.x;
      const bool hit = q < n && dec_id(labels, q, S, start_id) == id;
      const unsigned long long m = __ballot(hit);
      if (hit) list[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = q;
      base += __popcll(m);
    }
    if (threadIdx.x == 0) cnt = base;
  }
  __syncthreads();
  const int k = cnt;
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= k; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = to_f32(dy[(int64_t)list[i + u] * D + c]);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < k; ++i) s += to_f32(dy[(int64_t)list[i] * D + c]);
    dtable[(int64_t)id * D + c] = s;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int tmi_softmax_fwd_impl(float* s, int64_t rows, int64_t Tq, int64_t Tk, int32_t mask_mode, void* stream);
extern "C" int tmi_softmax_fwd(float* s, int64_t rows, int64_t Tq, int64_t Tk, int32_t mask_mode, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_softmax_fwd(s, rows, Tq, Tk, mask_mode, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_softmax_fwd_impl(s, rows, Tq, Tk, mask_mode, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_softmax_fwd_impl(float* s, int64_t rows, int64_t Tq, int64_t Tk, int32_t mask_mode, void* stream) {
  if (!s || rows <= 0 || Tq <= 0 || Tk <= 0 || Tk > 64 * SM_E || (mask_mode != 0 && mask_mode != 1)) {
    tmi_set_error("tmi_softmax_fwd: bad argument (Tk <= 2048)");
    return TMI_ERR_INVALID;
  }
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), s, rows, (int)Tq, (int)Tk, mask_mode);
  return tmi_check_launch("tmi_softmax_fwd");
}

static int tmi_softmax_bwd_impl(const float* p, float* dp, int64_t rows, int64_t Tk, void* stream);
extern "C" int tmi_softmax_bwd(const float* p, float* dp, int64_t rows, int64_t Tk, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_softmax_bwd(p, dp, rows, Tk, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_softmax_bwd_impl(p, dp, rows, Tk, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_softmax_bwd_impl(const float* p, float* dp, int64_t rows, int64_t Tk, void* stream) {
  if (!p || !dp || rows <= 0 || Tk <= 0 || Tk > 64 * SM_E) {
    tmi_set_error("tmi_softmax_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p, dp, rows, (int)Tk);
  return tmi_check_launch("tmi_softmax_bwd");
}

static int xent_launch(void* logits, int64_t ld, const int32_t* labels, float* row_loss, int64_t B, int64_t S, int64_t V,
                       float grad_scale, int32_t dtype, const LmOperands& lm, void* stream, const char* what) {
  const int vec = dtype == TMI_BF16 ? 8 : 4;
  if (!logits || !labels || !row_loss || B <= 0 || S <= 1 || V <= 0 || ld < V || ld % vec || !al16(logits)) {
    tmi_set_error("tmi_xent_fwd_bwd / tmi_linear_xent: bad argument (ld must be a multiple of 16 bytes, >= V)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)(B * S));
  static const int gen = [] { const char* e = getenv("TMI_XENT_GENERIC"); return e ? atoi(e) : 0; }();
  if (dtype == TMI_BF16 && !gen && ld <= (int64_t)256 * 8 * XR)
    hipLaunchKernelGGL(xent_rows_bf16_kernel, grid, dim3(256), 0, s, (bf16_t*)logits, ld, labels, row_loss, (int)S, V, grad_scale, lm);
  else if (dtype == TMI_BF16)
    hipLaunchKernelGGL(xent_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)logits, ld, labels, row_loss, (int)S, V,
                       grad_scale, lm);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(xent_kernel<float>, grid, dim3(256), 0, s, (float*)logits, ld, labels, row_loss, (int)S, V,
                       grad_scale, LmOperands{});  // (fp32 logits carry their target logit exactly)
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch(what);
}

static int tmi_xent_fwd_bwd_impl(void* logits, int64_t ld, const int32_t* labels, float* row_loss, int64_t B,
                                int64_t S, int64_t V, float grad_scale, int32_t dtype, void* stream);
extern "C" int tmi_xent_fwd_bwd(void* logits, int64_t ld, const int32_t* labels, float* row_loss, int64_t B,
                                int64_t S, int64_t V, float grad_scale, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_xent_fwd_bwd(logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_xent_fwd_bwd_impl(logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_xent_fwd_bwd_impl(void* logits, int64_t ld, const int32_t* labels, float* row_loss, int64_t B,
                                int64_t S, int64_t V, float grad_scale, int32_t dtype, void* stream) {
  return xent_launch(logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, LmOperands{}, stream, "tmi_xent_fwd_bwd");
}

static int tmi_linear_xent_impl(const void* x, int64_t x_ld, const void* w, int64_t w_sk, int64_t w_sn, int64_t d, void* logits,
                               int64_t ld, const int32_t* labels, float* row_loss, int64_t B, int64_t S, int64_t V,
                               float grad_scale, int32_t dtype, void* stream);
extern "C" int tmi_linear_xent(const void* x, int64_t x_ld, const void* w, int64_t w_sk, int64_t w_sn, int64_t d, void* logits,
                               int64_t ld, const int32_t* labels, float* row_loss, int64_t B, int64_t S, int64_t V,
                               float grad_scale, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_linear_xent(x, x_ld, w, w_sk, w_sn, d, logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_linear_xent_impl(x, x_ld, w, w_sk, w_sn, d, logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_linear_xent_impl(const void* x, int64_t x_ld, const void* w, int64_t w_sk, int64_t w_sn, int64_t d, void* logits,
                               int64_t ld, const int32_t* labels, float* row_loss, int64_t B, int64_t S, int64_t V,
                               float grad_scale, int32_t dtype, void* stream) {
  if (!x || !w || d <= 0 || d > (1 << 20) || x_ld < d || w_sk == 0 || w_sn == 0) {
    tmi_set_error("tmi_linear_xent: bad LM-head operands");
    return TMI_ERR_INVALID;
  }
  LmOperands lm{};
  if (dtype == TMI_BF16) lm = LmOperands{(const bf16_t*)x, (const bf16_t*)w, x_ld, w_sk, w_sn, (int)d};
  return xent_launch(logits, ld, labels, row_loss, B, S, V, grad_scale, dtype, lm, stream, "tmi_linear_xent");
}

static int tmi_sum_scale_impl(const float* x, float* out, int64_t n, float scale, void* stream);
extern "C" int tmi_sum_scale(const float* x, float* out, int64_t n, float scale, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_sum_scale(x, out, n, scale, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_sum_scale_impl(x, out, n, scale, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_sum_scale_impl(const float* x, float* out, int64_t n, float scale, void* stream) {
  if (!x || !out || n <= 0) {
    tmi_set_error("tmi_sum_scale: bad argument");
    return TMI_ERR_INVALID;
  }
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, n,
                     scale);
  return tmi_check_launch("tmi_sum_scale");
}

static int tmi_embed_fwd_impl(const int32_t* labels, const float* table, const float* pe, void* out, int64_t B,
                             int64_t S, int64_t D, int32_t start_id, int32_t dtype, void* stream);
extern "C" int tmi_embed_fwd(const int32_t* labels, const float* table, const float* pe, void* out, int64_t B,
                             int64_t S, int64_t D, int32_t start_id, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_embed_fwd(labels, table, pe, out, B, S, D, start_id, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_embed_fwd_impl(labels, table, pe, out, B, S, D, start_id, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_embed_fwd_impl(const int32_t* labels, const float* table, const float* pe, void* out, int64_t B,
                             int64_t S, int64_t D, int32_t start_id, int32_t dtype, void* stream) {
  if (!labels || !table || !pe || !out || B <= 0 || S <= 0 || D <= 0) {
    tmi_set_error("tmi_embed_fwd: bad argument");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)(B * S));
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, labels, table, pe, (bf16_t*)out, (int)S,
                       (int)D, start_id);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, s, labels, table, pe, (float*)out, (int)S,
                       (int)D, start_id);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_embed_fwd");
}

static int tmi_embed_bwd_impl(const int32_t* labels, const void* dy, float* dtable, int64_t B, int64_t S,
                             int64_t D, int32_t start_id, int32_t dtype, void* stream);
extern "C" int tmi_embed_bwd(const int32_t* labels, const void* dy, float* dtable, int64_t B, int64_t S,
                             int64_t D, int32_t start_id, int32_t dtype, void* stream) {
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_embed_bwd(labels, dy, dtable, B, S, D, start_id, dtype, stream); });
  tmi_plan_enter();
  const int rc_ = tmi_embed_bwd_impl(labels, dy, dtable, B, S, D, start_id, dtype, stream);
  tmi_plan_leave();
  return rc_;
}
static int tmi_embed_bwd_impl(const int32_t* labels, const void* dy, float* dtable, int64_t B, int64_t S,
                             int64_t D, int32_t start_id, int32_t dtype, void* stream) {
  if (!labels || !dy || !dtable || B <= 0 || S <= 0 || D <= 0 || B * S > 16384) {
    tmi_set_error("tmi_embed_bwd: bad argument (B*S <= 16384)");
    return TMI_ERR_INVALID;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int n = (int)(B * S);
  const size_t lds = (size_t)n * sizeof(int);
  if (dtype == TMI_BF16)
    hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(n), dim3(256), lds, s, labels, (const bf16_t*)dy, dtable, n,
                       (int)S, (int)D, start_id);
  else if (dtype == TMI_F32)
    hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(n), dim3(256), lds, s, labels, (const float*)dy, dtable, n,
                       (int)S, (int)D, start_id);
  else
    return TMI_ERR_UNSUPPORTED;
  return tmi_check_launch("tmi_embed_bwd");
}
