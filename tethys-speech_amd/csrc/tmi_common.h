// Shared device helpers for the gfx950 kernels of libtethys_mi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include "../../include/tethys_mi.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define TMI_WAVE 64

// thread-local error slot, filled by launch helpers (host side)
void tmi_set_error(const char* msg);
int tmi_check_launch(const char* what);
int tmi_deterministic();  // runtime.hip: tmi_set_deterministic

// Launch plans (plan.hip).  Every launching entry point is a thin wrapper around its `_impl`: while this thread records a
// plan (and is not already inside another entry point) the wrapper appends a closure that calls the entry point again
// with the same arguments - copied by value, per-step ones (dropout seeds, the Adam step) offset by the replay's deltas.
bool tmi_plan_recording();
void tmi_plan_push(std::function<int()> fn);
void tmi_plan_enter();
void tmi_plan_leave();
uint64_t tmi_plan_seed_delta();
int64_t tmi_plan_step_delta();

template <typename T> struct tmi_type;
template <> struct tmi_type<float> { static constexpr int id = TMI_F32; };
template <> struct tmi_type<bf16_t> { static constexpr int id = TMI_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// exact-erf GELU (tf.keras.activations.gelu(approximate=False)) and its derivative
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16-output variants: Abramowitz-Stegun 7.1.26 erf (|abs err| <= 1.5e-7, far below a bf16
// ulp) — 1 rcp + 1 exp + 6 fma instead of erff's ~60 instructions, so the GELU epilogue of
// the FFN GEMMs stays under the store time.  cdf and pdf share the one exponential.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);  // v_rcp_f32 (1 ulp): __frcp_rn expands to the ~10-instruction IEEE division
  const float e = __expf(-z * z);  // = exp(-x^2/2)
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float erf_abs = 1.0f - p * t * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143267794f * e;
}
template <typename TC> __device__ __forceinline__ float gelu_fwd_t(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_fwd_t<bf16_t>(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return x * c;
}
template <typename TC> __device__ __forceinline__ float gelu_grad_t(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_grad_t<bf16_t>(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return c + x * p;
}

// ---- counter-based dropout (tf.keras.layers.Dropout sites W:160, W:205, W:342, W:411 in training).
// TF's stateful RNG stream cannot be reproduced; the keep decision here is a pure function of
// (seed, stream, row, column) of the tensor the Dropout layer sees, so forward and backward regenerate the same mask
// and nothing is stored.  Two levels, so that the per-element cost is five full-rate VALU instructions per TWO elements:
//   row key   (ra, rb) = (mix32(key ^ row * 0x9E3779B1), mix32((key + 0x632BE5AB) ^ row * 0x85EBCA6B))
//                        two full avalanches per ROW (once per lane in the attention kernels): every bit of seed,
//                        stream and row reaches both words
//   pair hash x = ra ^ (column >> 1);  h = mul24(x, 0x9E3779);  h ^= (h >> 15) ^ rb;  h = mul24(h, 0x85EBCB)
// (v_mul_u32_u24 is a full-rate VALU instruction, the 32-bit v_mul_lo_u32 is quarter rate; v_xor3_b32 takes the two
// xors).  The low / high 16 bits of h are the draws of the even / odd column of the pair: an element is dropped when its
// draw is < thr = round(p * 65536); kept values are scaled by 65536 / (65536 - thr), the exact inverse of the keep
// probability.  Columns < TMI_DROP_MAX_COLS = 2^17 (above that the multiplicative pair hash starts to correlate columns
// 2^17 apart; the entry points refuse wider masks), rows < 2^32.  The oracle restates the same integer arithmetic (oracle/dropout.py).
__host__ __device__ __forceinline__ uint32_t tmi_mix32(uint32_t x) {  // "lowbias32" finaliser
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t tmi_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return (uint32_t)((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu));
#endif
}
constexpr int64_t TMI_DROP_MAX_COLS = 1 << 17;
struct tmi_rowkey { uint32_t a, b; };
__host__ __device__ __forceinline__ tmi_rowkey tmi_row_key(uint32_t stream_key, uint32_t row) {
  return tmi_rowkey{tmi_mix32(stream_key ^ (row * 0x9E3779B1u)), tmi_mix32((stream_key + 0x632BE5ABu) ^ (row * 0x85EBCA6Bu))};
}
// 32 bits for column pair `cp` (= column >> 1) of the row with key `rk`
__host__ __device__ __forceinline__ uint32_t tmi_pair_hash(tmi_rowkey rk, uint32_t cp) {
  uint32_t h = tmi_mul24(rk.a ^ cp, 0x9E3779u);
#if defined(__HIP_DEVICE_COMPILE__)
  h = __builtin_amdgcn_bitop3_b32(h, h >> 15, rk.b, 0x96);  // the same three-way XOR as ONE v_bitop3_b32 (truth table 0x96); hipcc emits two v_xor
#else
  h = h ^ (h >> 15) ^ rk.b;
#endif
  return tmi_mul24(h, 0x85EBCBu);
}
__host__ __device__ __forceinline__ uint32_t tmi_drop_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
// rates the 16-bit threshold can express: thr <= 65535 (p < ~0.999992).  thr == 65536 would make the keep scale infinite and
// overflow the in-place form `thr << 16` of the dK/dV kernel (its odd keys would all be kept while fwd / dQ drop them)
__host__ __device__ __forceinline__ bool tmi_drop_ok(float p) { return p < 1.f && tmi_drop_thr(p) <= 65535u; }
__host__ __device__ __forceinline__ float tmi_keep_scale(uint32_t thr) { return 65536.0f / (float)(65536u - thr); }
// key of a stream: seed (64 bit) and a 32-bit stream id (batch*heads + head for attention; 0 for flat tensors)
__host__ __device__ __forceinline__ uint32_t tmi_stream_key(uint64_t seed, uint32_t stream_id) {
  return tmi_mix32((uint32_t)seed ^ tmi_mix32((uint32_t)(seed >> 32) + stream_id));
}
// keep decision of element (row, col) of a stream
__host__ __device__ __forceinline__ bool tmi_keep(uint32_t stream_key, uint32_t row, uint32_t col, uint32_t thr) {
  const uint32_t h = tmi_pair_hash(tmi_row_key(stream_key, row), col >> 1);
  const uint32_t r = (col & 1) ? (h >> 16) : (h & 0xffffu);
  return r >= thr;
}

// ---- attention-probability dropout (W:160): the generator of the flash kernels.  Only tmi_attn_fwd evaluates it (the
// backward kernels read the keep bits the forward stored, 1 bit per score), so it is shaped for that kernel's lane: one
// evaluation serves the FOUR consecutive keys 4cq .. 4cq+3 that a lane holds in one accumulator quad -
//   x = ra ^ cq;  h = mul24(x, 0x9E3779);  h ^= (h >> 15) ^ rb;  ha = mul24(h, 0x85EBCB);  hb = mul24(h, 0xC2B2AE)
// (six full-rate VALU instructions per four scores; (ra, rb) = tmi_row_key of (stream b*H + head, query row)).  Draws: key
// 4cq + {0, 1, 2, 3} = {low half of ha, high half of ha, low half of hb, high half of hb}, read as SIGNED 16-bit numbers;
// a probability is kept when its draw >= thr - 32768 (thr = round(p * 65536), the same rate and keep scale as the flat
// generator above).  The signed form is what lets the kernel test two draws with one v_pk_sub_i16 (saturating) + one
// v_pk_ashrrev_i16 and apply the result to a packed bf16 pair.  Restated in oracle/dropout.py (keep_attention).
struct tmi_quad { uint32_t a, b; };
__host__ __device__ __forceinline__ tmi_quad tmi_quad_hash_x(uint32_t x, uint32_t rb) {  // x = ra ^ cq
  uint32_t h = tmi_mul24(x, 0x9E3779u);
#if defined(__HIP_DEVICE_COMPILE__)
  h = __builtin_amdgcn_bitop3_b32(h, h >> 15, rb, 0x96);
#else
  h = h ^ (h >> 15) ^ rb;
#endif
  return tmi_quad{tmi_mul24(h, 0x85EBCBu), tmi_mul24(h, 0xC2B2AEu)};
}
__host__ __device__ __forceinline__ tmi_quad tmi_quad_hash(tmi_rowkey rk, uint32_t cq) { return tmi_quad_hash_x(rk.a ^ cq, rk.b); }
__host__ __device__ __forceinline__ bool tmi_keep_attn(uint32_t stream_key, uint32_t row, uint32_t col, uint32_t thr) {
  const tmi_quad hq = tmi_quad_hash(tmi_row_key(stream_key, row), col >> 2);
  const uint32_t w = (col & 2) ? hq.b : hq.a;
  const int32_t d = (int32_t)(int16_t)((col & 1) ? (w >> 16) : (w & 0xffffu));
  return d >= (int32_t)thr - 32768;
}

// dropout term of a GEMM epilogue (tmi_gemm_desc.dropout_p): 8 consecutive columns n .. n+7 (n even) of output row m
__device__ __forceinline__ void tmi_drop8(float (&v)[8], int64_t m, int64_t n, uint32_t key, uint32_t thr, float scale) {
  const tmi_rowkey rk = tmi_row_key(key, (uint32_t)m);
  const uint32_t cp0 = (uint32_t)(n >> 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t h = tmi_pair_hash(rk, cp0 + j);
    v[2 * j] = (h & 0xffffu) >= thr ? v[2 * j] * scale : 0.f;
    v[2 * j + 1] = (h >> 16) >= thr ? v[2 * j + 1] * scale : 0.f;
  }
}
__device__ __forceinline__ float tmi_drop1(float v, int64_t m, int64_t n, uint32_t key, uint32_t thr, float scale) {
  return tmi_keep(key, (uint32_t)m, (uint32_t)n, thr) ? v * scale : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
