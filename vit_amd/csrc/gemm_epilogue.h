// Epilogue of the GEMM / conv kernels for 8 consecutive output columns n .. n+7 of logical row m, kept apart from the
// main loop in gemm.hip: per-column scale / bias, gate (gelu' or relu mask of a saved pre-activation), bf16
// pre-activation copy, dropout before the residual or after the activation, residual (fp32 or bf16, optionally
// periodic in m), ReLU / GELU, output row remap, fp32 or bf16 store.  `v` holds the fp32 accumulators.
#pragma once
#include "common.h"
#include "rvit_internal.h"

__device__ __forceinline__ void rvit_gemm_epilogue_row8(const GemmArgs& p, float (&v)[8], long m, int n) {
  long orow = m;
  if (p.row_group > 0) {
    const long g = m / p.row_group;
    orow = g * (p.row_group + p.row_extra) + (m - g * p.row_group) + p.row_extra;
  }
  const long rrow = p.r_period > 0 ? (m % p.r_period) : orow;
  if (p.vec) {
    if (p.scale) {
      const float4 a = *reinterpret_cast<const float4*>(p.scale + n);
      const float4 b = *reinterpret_cast<const float4*>(p.scale + n + 4);
      v[0] *= a.x; v[1] *= a.y; v[2] *= a.z; v[3] *= a.w; v[4] *= b.x; v[5] *= b.y; v[6] *= b.z; v[7] *= b.w;
    }
    if (p.bias) {
      const float4 a = *reinterpret_cast<const float4*>(p.bias + n);
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (p.gate) {
      const uint4 gq = *reinterpret_cast<const uint4*>(p.G + orow * p.ldg + n);
      const float gv[8] = {bf16_lo(gq.x), bf16_hi(gq.x), bf16_lo(gq.y), bf16_hi(gq.y),
                           bf16_lo(gq.z), bf16_hi(gq.z), bf16_lo(gq.w), bf16_hi(gq.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] *= p.gate == RVIT_GATE_GELU_GRAD ? gelu_erf_grad(gv[e]) : (gv[e] > 0.f ? 1.f : 0.f);
    }
    if (p.C2) {
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(p.C2 + orow * p.ldc2 + n) = o;
    }
    if (p.drop_mode == RVIT_DROP_PRE_RESIDUAL) {
      const uint32_t di = (uint32_t)(orow * p.N + n);
      drop_mult8(p.drop, di, v);   // vec form: N % 8 == 0 and n % 8 == 0, so di is even
    }
    if (p.R) {
      if (p.r_f32) {
        const float* rp = (const float*)p.R + rrow * p.ldr + n;
        const float4 a = *reinterpret_cast<const float4*>(rp);
        const float4 b = *reinterpret_cast<const float4*>(rp + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      } else {
        const uint4 r = *reinterpret_cast<const uint4*>((const bf16*)p.R + rrow * p.ldr + n);
        v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
        v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
      }
    }
    if (p.act == RVIT_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == RVIT_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    } else if (p.act >= RVIT_ACT_HSWISH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rvit_apply_act(v[e], p.act);
    }
    if (p.drop_mode == RVIT_DROP_POST_ACT) {
      const uint32_t di = (uint32_t)(orow * p.N + n);
      drop_mult8(p.drop, di, v);   // vec form: N % 8 == 0 and n % 8 == 0, so di is even
    }
    if (p.c_f32) {
      float* cp = (float*)p.C + orow * p.ldc + n;
      *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>((bf16*)p.C + orow * p.ldc + n) = o;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (n + e >= p.N) break;
      float x = v[e];
      if (p.scale) x *= p.scale[n + e];
      if (p.bias) x += p.bias[n + e];
      if (p.gate) {
        const float gv = (float)p.G[orow * p.ldg + n + e];
        x *= p.gate == RVIT_GATE_GELU_GRAD ? gelu_erf_grad(gv) : (gv > 0.f ? 1.f : 0.f);
      }
      if (p.C2) p.C2[orow * p.ldc2 + n + e] = (bf16)x;
      if (p.drop_mode == RVIT_DROP_PRE_RESIDUAL) x = rvit_mul_rn(x, drop_mult(p.drop, (uint32_t)(orow * p.N + n + e)));
      if (p.R) x += p.r_f32 ? ((const float*)p.R)[rrow * p.ldr + n + e]
                            : (float)((const bf16*)p.R)[rrow * p.ldr + n + e];
      x = rvit_apply_act(x, p.act);
      if (p.drop_mode == RVIT_DROP_POST_ACT) x = rvit_mul_rn(x, drop_mult(p.drop, (uint32_t)(orow * p.N + n + e)));
      if (p.c_f32) ((float*)p.C)[orow * p.ldc + n + e] = x;
      else ((bf16*)p.C)[orow * p.ldc + n + e] = (bf16)x;
    }
  }
}

// The same epilogue split into its loads and the rest, for kernels that run it in several passes over one tile
// (gemm8p.hip).  vmcnt counts loads AND stores in issue order on gfx950: a pass that loads its row operands (residual,
// gate) or the per-column scale / bias right before using them waits for every store of the pass before it to reach L2
// (~1 us per pass: 8 of the 30 us a ViT-B fc1 tile took).  Split, the column operands are read once per tile, a pass's
// row operands are requested BEFORE the previous pass's stores are issued, and the stores are never waited for.
// vec form only (p.vec != 0); same operations in the same order as rvit_gemm_epilogue_row8: identical bits.
struct EpiCols { float sc[8], bi[8]; };
struct EpiRow { long orow; float4 ra, rb; uint4 g; };

__device__ __forceinline__ void rvit_epi_cols_load(const GemmArgs& p, int n, EpiCols& c) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { c.sc[e] = 1.f; c.bi[e] = 0.f; }
  if (p.scale) {
    const float4 a = *reinterpret_cast<const float4*>(p.scale + n);
    const float4 b = *reinterpret_cast<const float4*>(p.scale + n + 4);
    c.sc[0] = a.x; c.sc[1] = a.y; c.sc[2] = a.z; c.sc[3] = a.w; c.sc[4] = b.x; c.sc[5] = b.y; c.sc[6] = b.z; c.sc[7] = b.w;
  }
  if (p.bias) {
    const float4 a = *reinterpret_cast<const float4*>(p.bias + n);
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n + 4);
    c.bi[0] = a.x; c.bi[1] = a.y; c.bi[2] = a.z; c.bi[3] = a.w; c.bi[4] = b.x; c.bi[5] = b.y; c.bi[6] = b.z; c.bi[7] = b.w;
  }
}

// m must be a valid logical row (callers clamp rows past M and mask the store instead)
__device__ __forceinline__ void rvit_epi_row_load(const GemmArgs& p, long m, int n, EpiRow& r) {
  long orow = m;
  if (p.row_group > 0) {
    const long g = m / p.row_group;
    orow = g * (p.row_group + p.row_extra) + (m - g * p.row_group) + p.row_extra;
  }
  r.orow = orow;
  const long rrow = p.r_period > 0 ? (m % p.r_period) : orow;
  if (p.gate) r.g = *reinterpret_cast<const uint4*>(p.G + orow * p.ldg + n);
  if (p.R) {
    if (p.r_f32) {
      const float* rp = (const float*)p.R + rrow * p.ldr + n;
      r.ra = *reinterpret_cast<const float4*>(rp);
      r.rb = *reinterpret_cast<const float4*>(rp + 4);
    } else {
      const uint4 q = *reinterpret_cast<const uint4*>((const bf16*)p.R + rrow * p.ldr + n);
      r.ra = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
    }
  }
}

__device__ __forceinline__ void rvit_epi_row_apply(const GemmArgs& p, float (&v)[8], const EpiRow& r, const EpiCols& c,
                                                   int n, bool store) {
  const long orow = r.orow;
  if (p.scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= c.sc[e];
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += c.bi[e];
  }
  if (p.gate) {
    const uint4 gq = r.g;
    const float gv[8] = {bf16_lo(gq.x), bf16_hi(gq.x), bf16_lo(gq.y), bf16_hi(gq.y),
                         bf16_lo(gq.z), bf16_hi(gq.z), bf16_lo(gq.w), bf16_hi(gq.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] *= p.gate == RVIT_GATE_GELU_GRAD ? gelu_erf_grad(gv[e]) : (gv[e] > 0.f ? 1.f : 0.f);
  }
  if (p.C2 && store) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p.C2 + orow * p.ldc2 + n) = o;
  }
  if (p.drop_mode == RVIT_DROP_PRE_RESIDUAL) {
    const uint32_t di = (uint32_t)(orow * p.N + n);
    drop_mult8(p.drop, di, v);   // N % 8 == 0 and n % 8 == 0: di is even
  }
  if (p.R) {
    if (p.r_f32) {
      v[0] += r.ra.x; v[1] += r.ra.y; v[2] += r.ra.z; v[3] += r.ra.w;
      v[4] += r.rb.x; v[5] += r.rb.y; v[6] += r.rb.z; v[7] += r.rb.w;
    } else {
      const uint32_t q0 = __float_as_uint(r.ra.x), q1 = __float_as_uint(r.ra.y), q2 = __float_as_uint(r.ra.z),
                     q3 = __float_as_uint(r.ra.w);
      v[0] += bf16_lo(q0); v[1] += bf16_hi(q0); v[2] += bf16_lo(q1); v[3] += bf16_hi(q1);
      v[4] += bf16_lo(q2); v[5] += bf16_hi(q2); v[6] += bf16_lo(q3); v[7] += bf16_hi(q3);
    }
  }
  if (p.act == RVIT_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (p.act == RVIT_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
  } else if (p.act >= RVIT_ACT_HSWISH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rvit_apply_act(v[e], p.act);
  }
  if (p.drop_mode == RVIT_DROP_POST_ACT) {
    const uint32_t di = (uint32_t)(orow * p.N + n);
    drop_mult8(p.drop, di, v);   // N % 8 == 0 and n % 8 == 0: di is even
  }
  if (!store) return;
  if (p.c_f32) {
    float* cp = (float*)p.C + orow * p.ldc + n;
    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>((bf16*)p.C + orow * p.ldc + n) = o;
  }
}

// Batch-statistics hook shared by the conv epilogues: sum and sum of squares of eight stored bf16 values
__device__ __forceinline__ void rvit_stat_acc8(const uint4& o, float (&s1)[8], float (&s2)[8]) {
  const float x[8] = {bf16_lo(o.x), bf16_hi(o.x), bf16_lo(o.y), bf16_hi(o.y), bf16_lo(o.z), bf16_hi(o.z), bf16_lo(o.w), bf16_hi(o.w)};
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] += x[e]; s2[e] = fmaf(x[e], x[e], s2[e]); }
}

// x summed over the sixteen lanes of a DPP row (lanes 16 k .. 16 k + 15): four row rotations on the vector ALU, every lane
// gets a total (lane-dependent association, fixed from run to run; callers read one lane per row).
__device__ __forceinline__ float rvit_sum_row16(float x) {
  x += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  x += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
  x += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
  x += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
  return x;
}

// Does a conv launch with `slots` M tiles write the per-tile BatchNorm statistics (GemmArgs::stat_part)?  Only the plain
// conv of a conv -> BatchNorm pair: no scale / bias / residual / activation, forward direction, slots within the workspace.
static inline bool rvit_conv_stats_wanted(const GemmArgs& a, long slots) {
  return a.stat_part && a.stat_slots && !a.bnr_y && !a.scale && !a.bias && !a.R && a.act == RVIT_ACT_NONE && !a.transposed &&
         slots > 0 && slots <= a.stat_cap_slots;
}
// ... or the backward sums of the BatchNorm its output enters (GemmArgs::bnr_y)?  A residual is allowed (the gradient
// arriving over the other branch of a block).
static inline bool rvit_conv_bnr_wanted(const GemmArgs& a, long slots) {
  return a.stat_part && a.stat_slots && a.bnr_y && a.bnr_fsums && !a.scale && !a.bias && a.act == RVIT_ACT_NONE && slots > 0 &&
         slots <= a.stat_cap_slots;
}
// 0 none, 1 forward statistics, 2 BatchNorm backward sums
static inline int rvit_conv_stat_mode(const GemmArgs& a, long slots) {
  return rvit_conv_stats_wanted(a, slots) ? 1 : rvit_conv_bnr_wanted(a, slots) ? 2 : 0;
}

// BatchNorm backward sums of 8 channels of one row: d = the stored gradient where the ReLU bit is set
// (bn.hip bn_bwd_reduce_kernel's terms: s1 += d, s2 += d (y - mean); rstd is applied to the slot total)
__device__ __forceinline__ void rvit_bnr_acc8(const uint4& o, const uint4& y, unsigned mask, const float (&mean)[8], float (&s1)[8],
                                              float (&s2)[8]) {
  const float d[8] = {bf16_lo(o.x), bf16_hi(o.x), bf16_lo(o.y), bf16_hi(o.y), bf16_lo(o.z), bf16_hi(o.z), bf16_lo(o.w), bf16_hi(o.w)};
  const float v[8] = {bf16_lo(y.x), bf16_hi(y.x), bf16_lo(y.y), bf16_hi(y.y), bf16_lo(y.z), bf16_hi(y.z), bf16_lo(y.w), bf16_hi(y.w)};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float de = (mask >> e) & 1u ? d[e] : 0.f;
    s1[e] += de;
    s2[e] += de * (v[e] - mean[e]);
  }
}
// rstd of channel c from the forward sums (as bn.hip computes it)
__device__ __forceinline__ float rvit_bnr_rstd(const float* fsums, int C, int c, float inv_m, float eps) {
  const float mean = fsums[c] * inv_m;
  const float var = fmaxf(fsums[C + c] * inv_m - mean * mean, 0.f);
  return 1.0f / sqrtf(var + eps);
}

// x summed over the eight lanes l ^ {0, 8, 16, 24, 32, 40, 48, 56} (every lane gets the total; fixed association):
// DPP row_ror:8 inside the 16-lane row, v_permlane16_swap between rows 0 <-> 1 and 2 <-> 3, v_permlane32_swap between
// the wave halves.
__device__ __forceinline__ float rvit_sum_lanes_8_16_32(float x) {
  x += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
