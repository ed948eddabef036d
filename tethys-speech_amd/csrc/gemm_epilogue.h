// Shared GEMM epilogue: 2x2 MFMA 32x32 accumulator blocks per wave (64x64), waves 2x2.
#pragma once
#include "tmi_common.h"

template <typename TC>
__device__ __forceinline__ void gemm_epilogue(const tmi_gemm_desc& d, f32x16 (&acc)[2][2], int64_t m0, int64_t n0,
                                              int64_t bz, int wr, int wc, int lane, bool atomic) {
  const uint32_t drop_thr = tmi_drop_thr(d.dropout_p);
  const uint32_t drop_key = drop_thr ? tmi_stream_key(d.dropout_seed, 0u) : 0u;
  const float drop_scale = tmi_keep_scale(drop_thr);
  TC* C = reinterpret_cast<TC*>(d.C) + bz * d.c_sb;
  TC* aux_out = d.aux_out ? reinterpret_cast<TC*>(d.aux_out) + bz * d.c_sb : nullptr;
  const TC* aux_in = d.aux_in ? reinterpret_cast<const TC*>(d.aux_in) + bz * d.c_sb : nullptr;
  const TC* resid = d.resid ? reinterpret_cast<const TC*>(d.resid) + bz * d.r_sb : nullptr;
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t n = n0 + wc * 64 + ni * 32 + c;
      if (n >= d.N) continue;
      const float bv = d.bias ? d.bias[bz * d.bias_sb + n] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int64_t m = m0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (m >= d.M) continue;
        const int64_t idx = m * d.ldc + n;
        float v = acc[mi][ni][reg];
        if (atomic) {
          if constexpr (sizeof(TC) == 4) atomicAdd(reinterpret_cast<float*>(C) + idx, v);
          continue;
        }
        v += bv;
        if (n < d.scale_cols) v *= d.scale;
        if (d.accumulate) v += to_f32(C[idx]);
        if (aux_out) aux_out[idx] = from_f32<TC>(v);
        if (d.act == 1) v = gelu_fwd_t<TC>(v);
        if (aux_in) v *= gelu_grad_t<TC>(to_f32(aux_in[idx]));
        if (drop_thr) v = tmi_drop1(v, m, n, drop_key, drop_thr, drop_scale);
        if (resid) v += to_f32(resid[m * d.r_ld + n]);
        C[idx] = from_f32<TC>(v);
      }
    }
  }
}
