// Fused multi-head attention for gfx950, bf16 in / fp32 accumulate, head_dim 64.
// Replaces speech_jobs/whisper_dist.py:147-171 (q·kᵀ, additive mask, softmax, probs·v, head
// merge) and its gradient without materialising the [B,H,Tq,Tk] score tensor.
//
// One template serves forward, the dQ pass and the dK/dV pass:
//   - a wavefront OWNS 32 rows (queries in fwd/dQ, keys in dK/dV); the owner index sits on
//     the MFMA lane, its 64-wide vectors live in registers as B operands;
//   - the other ("streamed") index is walked in 32-row tiles staged through LDS once per
//     workgroup (4 waves = 128 owners share each tile), in natural [row][d] form for the
//     "first" products (X[s][o] = sum_d T[s][d]·Own[o][d]) and in transposed [d][row] form
//     for the "second" products (Yᵀ[d][o] += sum_s T[s][d]·X[s][o]);
//   - X (scores / probabilities / dS) never leaves the accumulator registers: a 32x32 MFMA
//     result has its column on the lane and its rows in the registers, so it is directly the
//     B operand of the next MFMA that sums over its rows (k-order inside a step permuted:
//     element j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3));
//   - softmax statistics are per-lane scalars in fwd/dQ (query on the lane) and per-row
//     constants from LDS in dK/dV.
// The reference's decoder mask (W:416-418 + W:152-153) adds -1e9 in fp32 to keys j <= i;
// a fully masked row therefore softmaxes to exactly uniform, which is why (m, 1/l) are kept
// as two numbers instead of one log-sum-exp.
#include "tmi_common.h"

namespace {

constexpr int HD = 64;
constexpr int TN_STRIDE = 144;  // natural tile: 32 rows x (128 B + 16 pad)
constexpr int TT_STRIDE = 80;   // transposed tile: 64 rows (d) x (64 B + 16 pad)
constexpr int TN_BYTES = 32 * TN_STRIDE;
constexpr int TT_BYTES = 64 * TT_STRIDE;

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ void tile_fetch(u32x4& reg, const bf16_t* base, int64_t st, int row0, int T) {
  const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;
  const int gr = row0 + row;
  if (gr < T) {
    reg = *reinterpret_cast<const u32x4*>(base + (int64_t)gr * st + ch * 8);
  } else {
    reg = u32x4{0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ void tile_commit_nat(char* Tn, const u32x4& reg) {
  const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;
  *reinterpret_cast<u32x4*>(Tn + row * TN_STRIDE + ch * 16) = reg;
}
__device__ __forceinline__ void tile_commit_tr(char* Tt, const u32x4& reg) {
  const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;
  const bf16_t* e = reinterpret_cast<const bf16_t*>(&reg);
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16_t*>(Tt + (ch * 8 + j) * TT_STRIDE + row * 2) = e[j];
}

// owner vectors: lane (c, h) holds Own[o0 + c][16kk + 8h .. +7], kk = 0..3
__device__ __forceinline__ void load_owner(bf16x8 (&f)[4], const bf16_t* base, int64_t st, int o, int T, int h) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (o < T) {
      f[kk] = *reinterpret_cast<const bf16x8*>(base + (int64_t)o * st + kk * 16 + h * 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[kk][j] = (bf16_t)0.f;
    }
  }
}

// X[s][o] = sum_d Tn[s][d] * Own[o][d]
__device__ __forceinline__ f32x16 first_product(const char* Tn, const bf16x8 (&own)[4], int c, int h) {
  f32x16 x;
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Tn + c * TN_STRIDE + kk * 32 + h * 16);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, own[kk], x, 0, 0, 0);
  }
  return x;
}

// Yt[blk][d][o] += sum_s Tt[32*blk + d][s] * X[s][o]   (X given as fp32 accumulator)
__device__ __forceinline__ void second_product(const char* Tt, const f32x16& x, f32x16 (&y)[2], int c, int h) {
#pragma unroll
  for (int sI = 0; sI < 2; ++sI) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (bf16_t)x[8 * sI + j];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const char* p = Tt + (32 * blk + c) * TT_STRIDE + (16 * sI + 4 * h) * 2;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 16);
      const bf16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      y[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y[blk], 0, 0, 0);
    }
  }
}

// write Yt[blk][d][o] * scale to out[o][d] (bf16), owner o on the lane
__device__ __forceinline__ void store_owner(const f32x16 (&y)[2], bf16_t* base, int64_t st, int o, int T, int h,
                                            float scale) {
  if (o >= T) return;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (bf16_t)(y[blk][4 * g + i] * scale);
      *reinterpret_cast<bf16x4*>(base + (int64_t)o * st + 32 * blk + 8 * g + 4 * h) = v;
    }
  }
}

struct AttnP {
  tmi_attn_desc d;
  float dq_scale;
};

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kn = smem;
  char* Vt = smem + TN_BYTES;
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;

  bf16x8 qf[4];
  load_owner(qf, qb, d.q_st, q, Tq, h);

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int ntiles = (Tk + 31) / 32;
  u32x4 kreg, vreg;
  tile_fetch(kreg, kb, d.k_st, 0, Tk);
  tile_fetch(vreg, vb, d.v_st, 0, Tk);
  tile_commit_nat(Kn, kreg);
  tile_commit_tr(Vt, vreg);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    if (more) {
      tile_fetch(kreg, kb, d.k_st, (tile + 1) * 32, Tk);
      tile_fetch(vreg, vb, d.v_st, (tile + 1) * 32, Tk);
    }
    f32x16 s = first_product(Kn, qf, c, h);  // s[key][q]
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = tile * 32 + acc_row(e, h);
      float x = s[e];
      if (d.mask_mode == 1 && key <= q) x = x + (-1e9f);
      if (key >= Tk) x = -INFINITY;
      s[e] = x;
      mx = fmaxf(mx, x);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);
    const float alpha = __expf(m - mnew);
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __expf(s[e] - mnew);
      s[e] = p;
      rs += p;
    }
    l = l * alpha + rs;
    m = mnew;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    second_product(Vt, s, o, c, h);
    __syncthreads();
    if (more) {
      tile_commit_nat(Kn, kreg);
      tile_commit_tr(Vt, vreg);
      __syncthreads();
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  bf16_t* ob = reinterpret_cast<bf16_t*>(d.o) + b * d.o_sb + head * HD;
  store_owner(o, ob, d.o_st, q, Tq, h, inv);
  if (h == 0 && q < Tq) {
    float* st = d.stats + ((b * d.H + head) * Tq + q) * 2;
    st[0] = m;
    st[1] = inv;
  }
}

// ------------------------------------------------------------------ dQ pass (owner = query)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kn = smem;
  char* Kt = smem + TN_BYTES;
  char* Vn = smem + TN_BYTES + TT_BYTES;
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;
  const bf16_t* ob = reinterpret_cast<const bf16_t*>(d.o) + b * d.o_sb + head * HD;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(d.d_o) + b * d.do_sb + head * HD;

  bf16x8 qf[4], dof[4], of[4];
  load_owner(qf, qb, d.q_st, q, Tq, h);
  load_owner(dof, dob, d.do_st, q, Tq, h);
  load_owner(of, ob, d.o_st, q, Tq, h);
  float delta = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)dof[kk][j] * (float)of[kk][j];
  delta += __shfl_xor(delta, 32, 64);
  float m = 0.f, linv = 0.f;
  if (q < Tq) {
    const int64_t si = (b * d.H + head) * Tq + q;
    m = d.stats[si * 2];
    linv = d.stats[si * 2 + 1];
    if (h == 0) d.delta[si] = delta;
  }

  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;

  const int ntiles = (Tk + 31) / 32;
  u32x4 kreg, vreg;
  tile_fetch(kreg, kb, d.k_st, 0, Tk);
  tile_fetch(vreg, vb, d.v_st, 0, Tk);
  tile_commit_nat(Kn, kreg);
  tile_commit_tr(Kt, kreg);
  tile_commit_nat(Vn, vreg);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    if (more) {
      tile_fetch(kreg, kb, d.k_st, (tile + 1) * 32, Tk);
      tile_fetch(vreg, vb, d.v_st, (tile + 1) * 32, Tk);
    }
    f32x16 s = first_product(Kn, qf, c, h);    // s[key][q]
    f32x16 dp = first_product(Vn, dof, c, h);  // dp[key][q]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = tile * 32 + acc_row(e, h);
      float x = s[e];
      if (d.mask_mode == 1 && key <= q) x = x + (-1e9f);
      const float p = (key < Tk) ? __expf(x - m) * linv : 0.f;
      s[e] = p * (dp[e] - delta);  // dS
    }
    second_product(Kt, s, dq, c, h);  // dQt[d][q] += sum_key K[key][d] dS[key][q]
    __syncthreads();
    if (more) {
      tile_commit_nat(Kn, kreg);
      tile_commit_tr(Kt, kreg);
      tile_commit_nat(Vn, vreg);
      __syncthreads();
    }
  }
  bf16_t* dqb = reinterpret_cast<bf16_t*>(d.dq) + b * d.dq_sb + head * HD;
  store_owner(dq, dqb, d.dq_st, q, Tq, h, P.dq_scale);
}

// ------------------------------------------------------------------ dK/dV pass (owner = key)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qn = smem;
  char* Qt = Qn + TN_BYTES;
  char* On = Qt + TT_BYTES;
  char* Ot = On + TN_BYTES;
  float* rowc = reinterpret_cast<float*>(Ot + TT_BYTES);  // [3][32]: m, linv, delta of the q tile
  const tmi_attn_desc& d = P.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int key = blockIdx.x * 128 + wave * 32 + c;
  const int Tq = (int)d.Tq, Tk = (int)d.Tk;
  const bf16_t* qb = reinterpret_cast<const bf16_t*>(d.q) + b * d.q_sb + head * HD;
  const bf16_t* kb = reinterpret_cast<const bf16_t*>(d.k) + b * d.k_sb + head * HD;
  const bf16_t* vb = reinterpret_cast<const bf16_t*>(d.v) + b * d.v_sb + head * HD;
  const bf16_t* dob = reinterpret_cast<const bf16_t*>(d.d_o) + b * d.do_sb + head * HD;
  const float* stats = d.stats + (b * d.H + head) * Tq * 2;
  const float* deltas = d.delta + (b * d.H + head) * Tq;

  bf16x8 kf[4], vf[4];
  load_owner(kf, kb, d.k_st, key, Tk, h);
  load_owner(vf, vb, d.v_st, key, Tk, h);

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[i][e] = dv[i][e] = 0.f;

  const int ntiles = (Tq + 31) / 32;
  u32x4 qreg, oreg;
  float creg = 0.f;
  auto fetch_consts = [&](int row0) {
    // threads 0..95: (which = t/32, r = t%32)
    const int t = threadIdx.x;
    if (t < 96) {
      const int which = t >> 5, r = t & 31, qi = row0 + r;
      float v = 0.f;
      if (qi < Tq) v = which == 0 ? stats[qi * 2] : (which == 1 ? stats[qi * 2 + 1] : deltas[qi]);
      creg = v;
    }
  };
  tile_fetch(qreg, qb, d.q_st, 0, Tq);
  tile_fetch(oreg, dob, d.do_st, 0, Tq);
  fetch_consts(0);
  tile_commit_nat(Qn, qreg);
  tile_commit_tr(Qt, qreg);
  tile_commit_nat(On, oreg);
  tile_commit_tr(Ot, oreg);
  if (threadIdx.x < 96) rowc[threadIdx.x] = creg;
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    if (more) {
      tile_fetch(qreg, qb, d.q_st, (tile + 1) * 32, Tq);
      tile_fetch(oreg, dob, d.do_st, (tile + 1) * 32, Tq);
      fetch_consts((tile + 1) * 32);
    }
    f32x16 s = first_product(Qn, kf, c, h);   // s[q][key]
    f32x16 dp = first_product(On, vf, c, h);  // dp[q][key]
    f32x16 ds;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = acc_row(e, h);
      const int qi = tile * 32 + r;
      float x = s[e];
      if (d.mask_mode == 1 && key <= qi) x = x + (-1e9f);
      const float p = (qi < Tq) ? __expf(x - rowc[r]) * rowc[32 + r] : 0.f;
      s[e] = p;
      ds[e] = p * (dp[e] - rowc[64 + r]);
    }
    second_product(Ot, s, dv, c, h);   // dVt[d][key] += sum_q dO[q][d] P[q][key]
    second_product(Qt, ds, dk, c, h);  // dKt[d][key] += sum_q Q[q][d] dS[q][key]
    __syncthreads();
    if (more) {
      tile_commit_nat(Qn, qreg);
      tile_commit_tr(Qt, qreg);
      tile_commit_nat(On, oreg);
      tile_commit_tr(Ot, oreg);
      if (threadIdx.x < 96) rowc[threadIdx.x] = creg;
      __syncthreads();
    }
  }
  bf16_t* dkb = reinterpret_cast<bf16_t*>(d.dk) + b * d.dk_sb + head * HD;
  bf16_t* dvb = reinterpret_cast<bf16_t*>(d.dv) + b * d.dv_sb + head * HD;
  store_owner(dk, dkb, d.dk_st, key, Tk, h, 1.0f);
  store_owner(dv, dvb, d.dv_st, key, Tk, h, 1.0f);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool ok_mat(const void* p, int64_t sb, int64_t st) { return p && al16(p) && sb % 8 == 0 && st % 8 == 0; }

int check_common(const tmi_attn_desc& d) {
  if (d.B <= 0 || d.H <= 0 || d.Tq <= 0 || d.Tk <= 0 || d.B > 65535 || d.H > 65535 ||
      (d.mask_mode != 0 && d.mask_mode != 1) || !d.stats)
    return 0;
  return ok_mat(d.q, d.q_sb, d.q_st) && ok_mat(d.k, d.k_sb, d.k_st) && ok_mat(d.v, d.v_sb, d.v_st) &&
         ok_mat(d.o, d.o_sb, d.o_st);
}

}  // namespace

extern "C" int tmi_attn_fwd(const tmi_attn_desc* dp, void* stream) {
  if (!dp || !check_common(*dp)) {
    tmi_set_error("tmi_attn_fwd: bad argument (16-byte aligned bf16 operands, strides multiple of 8)");
    return TMI_ERR_INVALID;
  }
  AttnP P;
  P.d = *dp;
  P.dq_scale = 1.f;
  dim3 grid((unsigned)((dp->Tq + 127) / 128), (unsigned)dp->H, (unsigned)dp->B);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), TN_BYTES + TT_BYTES, reinterpret_cast<hipStream_t>(stream), P);
  return tmi_check_launch("tmi_attn_fwd");
}

extern "C" int tmi_attn_bwd(const tmi_attn_desc* dp, void* stream) {
  if (!dp || !check_common(*dp) || !ok_mat(dp->d_o, dp->do_sb, dp->do_st) || !ok_mat(dp->dq, dp->dq_sb, dp->dq_st) ||
      !ok_mat(dp->dk, dp->dk_sb, dp->dk_st) || !ok_mat(dp->dv, dp->dv_sb, dp->dv_st) || !dp->delta) {
    tmi_set_error("tmi_attn_bwd: bad argument");
    return TMI_ERR_INVALID;
  }
  AttnP P;
  P.d = *dp;
  P.dq_scale = dp->dq_scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 gq((unsigned)((dp->Tq + 127) / 128), (unsigned)dp->H, (unsigned)dp->B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), 2 * TN_BYTES + TT_BYTES, s, P);
  int rc = tmi_check_launch("tmi_attn_bwd(dq)");
  if (rc) return rc;
  dim3 gk((unsigned)((dp->Tk + 127) / 128), (unsigned)dp->H, (unsigned)dp->B);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, gk, dim3(256), 2 * TN_BYTES + 2 * TT_BYTES + 96 * sizeof(float), s, P);
  return tmi_check_launch("tmi_attn_bwd(dkv)");
}
