// Multi-head self-attention for head dims other than 64 (16, 32, 48, 96, 128), forward and backward, S <= 256.
// The reference builds nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads) for ANY pair with
// embed_dim % num_heads == 0 (train_vit.py:76-111 -> torch activation.py MultiheadAttention); its documented
// configurations all have head_dim 64 and run csrc/attn.hip, this file serves the others (e.g. 256 / 8, 256 / 2).
// Same contract as rvit_attn_fwd / rvit_attn_bwd: packed q | k | v rows, the dropout site indexed
// ((b*H + h)*S + query)*S2 + key, lse saved by the forward.
//
// gfx950 structure: v_mfma_f32_16x16x16_bf16 throughout, because its accumulator layout (lane = column, four
// consecutive rows per lane) IS its B-operand layout: every product below whose contraction runs over the rows of a
// tile that was just computed (P^T, dS^T, P, dS) takes that tile straight from the accumulator registers.  The
// operands that are needed transposed (V^T, K^T, dO^T, Q^T) come from the row-major LDS images through the hardware
// transposing read ds_read_b64_tr_b16 (a 16-lane group reads a 4-row x 16-column block and each lane receives one
// column), so every tensor is staged once, in one orientation.
//   forward / backward-dQ: one workgroup per (batch, head); K and V resident in LDS, staged once; a wave owns 16 queries
//                per round (eight waves, or four when S <= 64) and keeps its whole score row in registers (no online
//                rescale).
//   backward-dK/dV: one workgroup per (batch, head); Q and dO resident in LDS together with lse and
//                delta = rowsum(dO * O); a wave owns 16 keys per round and sweeps the queries.  No atomics: every output
//                element is written by exactly one lane.
#include "common.h"
#include "rvit_internal.h"

namespace {

template <int DH>
struct DhGeom {
  static constexpr int DT = DH / 16;          // 16-column tiles of a head
  static constexpr int ROWB = DH * 2 + 16;    // bytes per image row (the 16 spare bytes spread the rows over the banks)
  static constexpr int CPR = DH / 8;          // 16-byte chunks per row
};

__device__ __forceinline__ f32x4 dh_mfma(bf16x4 a, bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 dh_row4(const char* p) { return *reinterpret_cast<const bf16x4*>(p); }
// lane (g = lane >> 4, j = lane & 15) receives image[row0 + 4g + 0..3][col0 + j]
template <int ROWB>
__device__ __forceinline__ bf16x4 dh_tr4(const char* image, int row0, int col0, int lane) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int j = lane & 15, g = lane >> 4;
  const char* p = image + (row0 + 4 * g + (j >> 2)) * ROWB + (col0 + (j & 3) * 4) * 2;
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
__device__ __forceinline__ bf16x4 dh_pack4(const f32x4& v) {
  bf16x4 r;
  r[0] = (bf16)v[0]; r[1] = (bf16)v[1]; r[2] = (bf16)v[2]; r[3] = (bf16)v[3];
  return r;
}
__device__ __forceinline__ bf16x4 dh_load4(const bf16* p, bool ok) {
  uint2 u = ok ? *reinterpret_cast<const uint2*>(p) : make_uint2(0, 0);
  return __builtin_bit_cast(bf16x4, u);
}
__device__ __forceinline__ void dh_store4(bf16* p, const f32x4& v, float mul) {
  uint2 u;
  u.x = pack_bf16x2(v[0] * mul, v[1] * mul);
  u.y = pack_bf16x2(v[2] * mul, v[3] * mul);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float dh_sum_groups(float v) {   // over the four lanes that share lane & 15
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float dh_max_groups(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}

// rows [0, rows_pad) of a [S][DH] slice (row pitch ld elements) -> row-major image; rows >= S are zero.  Four loads of
// a thread are in flight before the first LDS write (the loop is latency-bound otherwise: one workgroup per CU at
// head_dim 128)
template <int DH>
__device__ __forceinline__ void dh_stage(const bf16* src, long ld, int S, int rows_pad, char* image, int t, int nthreads) {
  constexpr int CPR = DhGeom<DH>::CPR, ROWB = DhGeom<DH>::ROWB;
  const int total = rows_pad * CPR;
  for (int id0 = t; id0 < total; id0 += 4 * nthreads) {
    uint4 u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = id0 + i * nthreads, row = id / CPR, c = id - row * CPR;
      u[i] = (id < total && row < S) ? *reinterpret_cast<const uint4*>(src + (long)row * ld + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = id0 + i * nthreads, row = id / CPR, c = id - row * CPR;
      if (id < total) *reinterpret_cast<uint4*>(image + row * ROWB + c * 16) = u[i];
    }
  }
}

template <int DH, int NC>
__global__ __launch_bounds__(512) void attn_dh_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                         float* __restrict__ lse, int S, int H, long ldq, long ldo,
                                                         float scale_log2e, float scale, DropKey dk) {
  constexpr int DT = DhGeom<DH>::DT, ROWB = DhGeom<DH>::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, g = lane >> 4;
  const int pair = blockIdx.x, nw = blockDim.x >> 6;
  const int b = pair / H, h = pair - b * H, E = H * DH;
  const int nch = (S + 15) >> 4, SP = nch * 16;
  char* const sK = smem;
  char* const sV = smem + SP * ROWB;
  const bf16* base = qkv + (long)b * S * ldq + h * DH;
  dh_stage<DH>(base + E, ldq, S, SP, sK, t, blockDim.x);
  dh_stage<DH>(base + 2 * E, ldq, S, SP, sV, t, blockDim.x);
  __syncthreads();

  for (int qb = wave; qb < nch; qb += nw) {   // wave-uniform; no barrier inside
  const int query = qb * 16 + j;
  const bool q_ok = query < S;
  const bf16* qrow = base + (long)(q_ok ? query : 0) * ldq;
  bf16x4 qf[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) qf[dt] = dh_load4(qrow + dt * 16 + 4 * g, q_ok);

  // S^T[key][query]: rows = keys 16c + 4g + r, column = this lane's query
  f32x4 sc[NC];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    sc[c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c < nch) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) a = dh_mfma(dh_row4(sK + (c * 16 + j) * ROWB + (dt * 16 + 4 * g) * 2), qf[dt], a);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[c][r] = (c * 16 + 4 * g + r < S) ? a[r] : -INFINITY;
        mx = fmaxf(mx, sc[c][r]);
      }
    }
  }
  mx = dh_max_groups(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pe = __builtin_amdgcn_exp2f((sc[c][r] - mx) * scale_log2e);
      sc[c][r] = pe;
      sum += pe;
    }
  sum = dh_sum_groups(sum);
  if (dk.thresh) {
    const uint32_t base2 = (uint32_t)(pair * S + query) * (uint32_t)((S + 1) >> 1);
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (c < nch) {
        float m[4];
        drop_keep4(dk, base2 + c * 8 + 2 * g, dk.scale, m);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[c][r] *= m[r];
      }
  }
  // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c)
    if (c < nch) {
      const bf16x4 pb = dh_pack4(sc[c]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = dh_mfma(dh_tr4<ROWB>(sV, c * 16, dt * 16, lane), pb, o[dt]);
    }
  if (q_ok) {
    const float inv = 1.0f / sum;
    bf16* orow = out + ((long)b * S + query) * ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dh_store4(orow + dt * 16 + 4 * g, o[dt], inv);
    if (lse && g == 0) lse[(long)pair * S + query] = mx * scale + logf(sum);
  }
  }
}

// dQ of 16 queries per wave: P recomputed from lse, dS^T = P^T o (dP^T o mask - delta), dQ^T = K^T dS^T
template <int DH, int NC>
__global__ __launch_bounds__(512) void attn_dh_bwd_q_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                           const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                           bf16* __restrict__ dqkv, int S, int H, long ldq, long ldo,
                                                           long lddo, long lddq, float scale_log2e, float scale, DropKey dk) {
  constexpr int DT = DhGeom<DH>::DT, ROWB = DhGeom<DH>::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, g = lane >> 4;
  const int pair = blockIdx.x, nw = blockDim.x >> 6;
  const int b = pair / H, h = pair - b * H, E = H * DH;
  const int nch = (S + 15) >> 4, SP = nch * 16;
  char* const sK = smem;
  char* const sV = smem + SP * ROWB;
  const bf16* base = qkv + (long)b * S * ldq + h * DH;
  dh_stage<DH>(base + E, ldq, S, SP, sK, t, blockDim.x);
  dh_stage<DH>(base + 2 * E, ldq, S, SP, sV, t, blockDim.x);
  __syncthreads();

  for (int qb = wave; qb < nch; qb += nw) {   // wave-uniform; no barrier inside
  const int query = qb * 16 + j;
  const bool q_ok = query < S;
  const long token = (long)b * S + (q_ok ? query : 0);
  bf16x4 qf[DT], df[DT];
  float delta = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    qf[dt] = dh_load4(qkv + token * ldq + h * DH + dt * 16 + 4 * g, q_ok);
    df[dt] = dh_load4(dO + token * lddo + h * DH + dt * 16 + 4 * g, q_ok);
    const bf16x4 of = dh_load4(O + token * ldo + h * DH + dt * 16 + 4 * g, q_ok);
#pragma unroll
    for (int r = 0; r < 4; ++r) delta += (float)df[dt][r] * (float)of[r];
  }
  delta = dh_sum_groups(delta);
  const float lse2 = (q_ok ? lse[(long)pair * S + query] : 0.f) * 1.44269504088896340736f;
  const uint32_t base2 = (uint32_t)(pair * S + query) * (uint32_t)((S + 1) >> 1);

  f32x4 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nch; ++c) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int off = (c * 16 + j) * ROWB + (dt * 16 + 4 * g) * 2;
      s = dh_mfma(dh_row4(sK + off), qf[dt], s);
      dp = dh_mfma(dh_row4(sV + off), df[dt], dp);
    }
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (dk.thresh) drop_keep4(dk, base2 + c * 8 + 2 * g, dk.scale, m);
    f32x4 ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (c * 16 + 4 * g + r < S) ? __builtin_amdgcn_exp2f(s[r] * scale_log2e - lse2) : 0.f;
      ds[r] = p * (dp[r] * m[r] - delta);
    }
    const bf16x4 dsb = dh_pack4(ds);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = dh_mfma(dh_tr4<ROWB>(sK, c * 16, dt * 16, lane), dsb, dq[dt]);
  }
  if (q_ok) {
    bf16* row = dqkv + token * lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dh_store4(row + dt * 16 + 4 * g, dq[dt], scale);
  }
  }
}

// dK and dV of 16 keys per wave: sweep over the queries with Q, dO, lse and delta resident in LDS
template <int DH, int NC>
__global__ __launch_bounds__(512) void attn_dh_bwd_kv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                            const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, int S, int H, long ldq, long ldo,
                                                            long lddo, long lddq, float scale_log2e, float scale, DropKey dk) {
  constexpr int DT = DhGeom<DH>::DT, ROWB = DhGeom<DH>::ROWB, CPR = DhGeom<DH>::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, g = lane >> 4;
  const int pair = blockIdx.x, nw = blockDim.x >> 6;
  const int b = pair / H, h = pair - b * H, E = H * DH;
  const int nch = (S + 15) >> 4, SP = nch * 16;
  char* const sQ = smem;
  char* const sD = smem + SP * ROWB;
  float* const sLse = reinterpret_cast<float*>(smem + 2 * SP * ROWB);   // lse * log2(e); +inf on the padding rows
  float* const sDel = sLse + SP;
  const bf16* base = qkv + (long)b * S * ldq + h * DH;
  const bf16* dbase = dO + (long)b * S * lddo + h * DH;
  dh_stage<DH>(base, ldq, S, SP, sQ, t, blockDim.x);
  dh_stage<DH>(dbase, lddo, S, SP, sD, t, blockDim.x);
  if (t < SP) {   // one query row per thread (S <= 256)
    float d = 0.f, l = INFINITY;
    if (t < S) {
      const bf16* orow = O + ((long)b * S + t) * ldo + h * DH;
      const bf16* drow = dbase + (long)t * lddo;
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        const bf16x8 ov = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(orow + c * 8));
        const bf16x8 dv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(drow + c * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)dv[e] * (float)ov[e];
      }
      l = lse[(long)pair * S + t] * 1.44269504088896340736f;
    }
    sDel[t] = d;
    sLse[t] = l;
  }
  __syncthreads();

  for (int kb = wave; kb < nch; kb += nw) {   // wave-uniform; no barrier inside
  const int key = kb * 16 + j;
  const bool k_ok = key < S;
  const long ktoken = (long)b * S + (k_ok ? key : 0);
  bf16x4 kf[DT], vf[DT];
  f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    kf[dt] = dh_load4(qkv + ktoken * ldq + E + h * DH + dt * 16 + 4 * g, k_ok);
    vf[dt] = dh_load4(qkv + ktoken * ldq + 2 * E + h * DH + dt * 16 + 4 * g, k_ok);
    dkacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const uint32_t S2h = (uint32_t)((S + 1) >> 1);

  for (int c = 0; c < nch; ++c) {
    // S[query][key] and dP[query][key]: rows = queries 16c + 4g + r, column = this lane's key
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int off = (c * 16 + j) * ROWB + (dt * 16 + 4 * g) * 2;
      s = dh_mfma(dh_row4(sQ + off), kf[dt], s);
      dp = dh_mfma(dh_row4(sD + off), vf[dt], dp);
    }
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + c * 16 + 4 * g);
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + c * 16 + 4 * g);
    f32x4 pd, ds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] * scale_log2e - l4[r]);   // 0 on the padding rows (lse = +inf)
      float m = 1.f;
      if (dk.thresh) m = drop_mult(dk, 2u * ((uint32_t)(pair * S + c * 16 + 4 * g + r) * S2h) + (uint32_t)key);
      pd[r] = p * m;
      ds[r] = p * (dp[r] * m - d4[r]);
    }
    const bf16x4 pb = dh_pack4(pd), dsb = dh_pack4(ds);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dvacc[dt] = dh_mfma(dh_tr4<ROWB>(sD, c * 16, dt * 16, lane), pb, dvacc[dt]);
      dkacc[dt] = dh_mfma(dh_tr4<ROWB>(sQ, c * 16, dt * 16, lane), dsb, dkacc[dt]);
    }
  }
  if (k_ok) {
    bf16* row = dqkv + ktoken * lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dh_store4(row + E + dt * 16 + 4 * g, dkacc[dt], scale);
      dh_store4(row + 2 * E + dt * 16 + 4 * g, dvacc[dt], 1.f);
    }
  }
  }
}

struct DhArgs {
  const bf16 *qkv, *O, *dO;
  const float* lse_in;
  bf16 *out, *dqkv;
  float* lse_out;
  int B, S, H;
  long ldq, ldo, lddo, lddq;
  float scale;
  DropKey dk;
  hipStream_t st;
};

template <int DH, int NC>
int dh_launch_fwd(const DhArgs& a) {
  const int nch = (a.S + 15) / 16, lds = 2 * nch * 16 * DhGeom<DH>::ROWB;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)attn_dh_fwd_kernel<DH, NC>}, NC * 32 * DhGeom<DH>::ROWB, "attn_dh_fwd")) return rc;
  const int threads = nch <= 4 ? 256 : 512;   // one wave per 16-query block and round
  hipLaunchKernelGGL((attn_dh_fwd_kernel<DH, NC>), dim3((unsigned)(a.B * a.H)), dim3(threads), lds, a.st, a.qkv, a.out,
                     a.lse_out, a.S, a.H, a.ldq, a.ldo, a.scale * 1.44269504088896340736f, a.scale, a.dk);
  RVIT_CHECK_LAUNCH("attn_dh_fwd");
  return RVIT_OK;
}

template <int DH, int NC>
int dh_launch_bwd(const DhArgs& a) {
  const int nch = (a.S + 15) / 16, lds = 2 * nch * 16 * DhGeom<DH>::ROWB + 2 * nch * 16 * 4;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)attn_dh_bwd_q_kernel<DH, NC>, (const void*)attn_dh_bwd_kv_kernel<DH, NC>},
                                NC * 32 * DhGeom<DH>::ROWB + NC * 128, "attn_dh_bwd"))
    return rc;
  const int threads = nch <= 4 ? 256 : 512;
  const float sl2 = a.scale * 1.44269504088896340736f;
  hipLaunchKernelGGL((attn_dh_bwd_q_kernel<DH, NC>), dim3((unsigned)(a.B * a.H)), dim3(threads), lds, a.st, a.qkv, a.O, a.dO,
                     a.lse_in, a.dqkv, a.S, a.H, a.ldq, a.ldo, a.lddo, a.lddq, sl2, a.scale, a.dk);
  RVIT_CHECK_LAUNCH("attn_dh_bwd_q");
  hipLaunchKernelGGL((attn_dh_bwd_kv_kernel<DH, NC>), dim3((unsigned)(a.B * a.H)), dim3(threads), lds, a.st, a.qkv, a.O, a.dO,
                     a.lse_in, a.dqkv, a.S, a.H, a.ldq, a.ldo, a.lddo, a.lddq, sl2, a.scale, a.dk);
  RVIT_CHECK_LAUNCH("attn_dh_bwd_kv");
  return RVIT_OK;
}

template <int DH>
int dh_pick(const DhArgs& a, bool bwd) {
  const int nch = (a.S + 15) / 16;
  if (nch <= 4) return bwd ? dh_launch_bwd<DH, 4>(a) : dh_launch_fwd<DH, 4>(a);
  if (nch <= 8) return bwd ? dh_launch_bwd<DH, 8>(a) : dh_launch_fwd<DH, 8>(a);
  if (nch <= 13) return bwd ? dh_launch_bwd<DH, 13>(a) : dh_launch_fwd<DH, 13>(a);
  return bwd ? dh_launch_bwd<DH, 16>(a) : dh_launch_fwd<DH, 16>(a);
}

int dh_dispatch(const DhArgs& a, int Dh, bool bwd, const char* what) {
  switch (Dh) {
    case 16: return dh_pick<16>(a, bwd);
    case 32: return dh_pick<32>(a, bwd);
    case 48: return dh_pick<48>(a, bwd);
    case 96: return dh_pick<96>(a, bwd);
    case 128: return dh_pick<128>(a, bwd);
  }
  rvit_set_error("%s: head_dim %d unsupported (16, 32, 48, 96, 128 here; 64 is rvit_attn_fwd / rvit_attn_bwd)", what, Dh);
  return RVIT_EUNSUPPORTED;
}

}  // namespace

int rvit_attn_dh_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, int Dh, long ldq, long ldo,
                          float scale, float drop_p, uint64_t drop_seed, hipStream_t st) {
  RVIT_CHECK_ARG(qkv && out, "attn_dh_fwd: null pointer");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attn_dh_fwd: drop_p must be in [0, 1)");
  RVIT_CHECK_ARG(drop_p == 0.f || (long)B * H * S * (S + (S & 1)) < 0xffffffffL, "attn_dh_fwd: dropout element index exceeds 32 bits");
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256 && Dh > 0, "attn_dh_fwd: unsupported shape B=%d H=%d S=%d Dh=%d (S<=256)", B, H, S, Dh);
  RVIT_CHECK_ARG(scale > 0.f, "attn_dh_fwd: scale must be positive");
  RVIT_CHECK_ARG(ldq >= 3L * H * Dh && (ldq & 7) == 0 && ldo >= (long)H * Dh && (ldo & 3) == 0,
                 "attn_dh_fwd: bad pitches ldq=%ld ldo=%ld", ldq, ldo);
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && ((uintptr_t)out & 7) == 0, "attn_dh_fwd: alignment");
  DhArgs a{};
  a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.lse_out = lse;
  a.B = B; a.S = S; a.H = H; a.ldq = ldq; a.ldo = ldo; a.scale = scale; a.dk = make_drop_key(drop_p, drop_seed); a.st = st;
  return dh_dispatch(a, Dh, false, "attn_dh_fwd");
}

int rvit_attn_dh_bwd_impl(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H,
                          int Dh, long ldq, long ldo, long lddo, long lddq, float scale, float drop_p, uint64_t drop_seed,
                          hipStream_t st) {
  RVIT_CHECK_ARG(qkv && O && dO && lse && dqkv, "attn_dh_bwd: null pointer");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attn_dh_bwd: drop_p must be in [0, 1)");
  RVIT_CHECK_ARG(drop_p == 0.f || (long)B * H * S * (S + (S & 1)) < 0xffffffffL, "attn_dh_bwd: dropout element index exceeds 32 bits");
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256 && Dh > 0, "attn_dh_bwd: unsupported shape B=%d H=%d S=%d Dh=%d (S<=256)", B, H, S, Dh);
  RVIT_CHECK_ARG(scale > 0.f, "attn_dh_bwd: scale must be positive");
  RVIT_CHECK_ARG(ldq >= 3L * H * Dh && lddq >= 3L * H * Dh && ldo >= (long)H * Dh && lddo >= (long)H * Dh &&
                     ((ldq | lddq | ldo | lddo) & 7) == 0,
                 "attn_dh_bwd: bad pitches");
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && rvit_aligned16(O) && rvit_aligned16(dO) && rvit_aligned16(dqkv), "attn_dh_bwd: alignment");
  DhArgs a{};
  a.qkv = (const bf16*)qkv; a.O = (const bf16*)O; a.dO = (const bf16*)dO; a.lse_in = lse; a.dqkv = (bf16*)dqkv;
  a.B = B; a.S = S; a.H = H; a.ldq = ldq; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.scale = scale;
  a.dk = make_drop_key(drop_p, drop_seed); a.st = st;
  return dh_dispatch(a, Dh, true, "attn_dh_bwd");
}
