// LayerNorm forward/backward over the embedding axis (torch transformer.py:774,947-950;
// train_vit.py:119,166).  HBM-bound: one wave per row, whole row held in registers,
// 16-byte loads, fp32 statistics, bf16 (GEMM operand) or fp32 output.
#include "common.h"
#include "rvit_internal.h"
#include "mx_common.h"

namespace {

constexpr int LN_MAX_IT = 8;  // E <= 2048

template <bool IN_F32>
__device__ __forceinline__ void load4(const void* base, long idx, float (&v)[4]) {
  if constexpr (IN_F32) {
    const float4 f = *reinterpret_cast<const float4*>((const float*)base + idx);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>((const bf16*)base + idx);
    v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
  }
}

template <bool IN_F32, bool OUT_F32, int IT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            long ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, long rows, int E, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float v[IT][4];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
      load4<IN_F32>(x, row * ldx + e, v[it]);
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    } else {
      v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
    }
  }
  const float inv_e = 1.0f / (float)E;
  const float mean = ln_mean(wave_sum(s), inv_e);
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[it][k] = __fsub_rn(v[it][k], mean);
        q = ln_sq(q, v[it][k]);
      }
    }
  }
  const float rstd = ln_rstd(wave_sum(q), inv_e, eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + e);
      const float4 b = *reinterpret_cast<const float4*>(beta + e);
      const float o0 = ln_affine(v[it][0], rstd, g.x, b.x);
      const float o1 = ln_affine(v[it][1], rstd, g.y, b.y);
      const float o2 = ln_affine(v[it][2], rstd, g.z, b.z);
      const float o3 = ln_affine(v[it][3], rstd, g.w, b.w);
      if constexpr (OUT_F32) {
        *reinterpret_cast<float4*>((float*)y + row * ldy + e) = make_float4(o0, o1, o2, o3);
      } else {
        uint2 o;
        o.x = pack_bf16x2(o0, o1);
        o.y = pack_bf16x2(o2, o3);
        *reinterpret_cast<uint2*>((bf16*)y + row * ldy + e) = o;
      }
    }
  }
}

// LayerNorm whose output is MXFP8 (the operand format of gemm_mxfp8): same statistics and affine as above, the result
// rounded to bf16 first (so the bytes equal layernorm_fwd -> quantize_mx, which this kernel replaces: one pass
// over the fp32 stream instead of a bf16 round trip through HBM), then quantised per 32-element block = 8 adjacent
// lanes.  E % 128 == 0; scales in the tiled layout of the consumer GEMM.
template <bool IN_F32, int IT>
__global__ __launch_bounds__(256) void layernorm_mx_fwd_kernel(const void* __restrict__ x, long ldx,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, uint8_t* __restrict__ q,
                                                               long ldq, uint8_t* __restrict__ sc, long rows, int E,
                                                               float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // wave-uniform
  const int lane = threadIdx.x & 63;
  float v[IT][4];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
      load4<IN_F32>(x, row * ldx + e, v[it]);
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    } else {
      v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
    }
  }
  const float inv_e = 1.0f / (float)E;
  const float mean = ln_mean(wave_sum(s), inv_e);
  float qq = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[it][k] = __fsub_rn(v[it][k], mean);
        qq = ln_sq(qq, v[it][k]);
      }
    }
  }
  const float rstd = ln_rstd(wave_sum(qq), inv_e, eps);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    const bool ok = e < E;   // E % 128 == 0: the 8 lanes of a block are all in or all out
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + e);
      const float4 b = *reinterpret_cast<const float4*>(beta + e);
      const uint32_t p0 = pack_bf16x2(ln_affine(v[it][0], rstd, g.x, b.x), ln_affine(v[it][1], rstd, g.y, b.y));
      const uint32_t p1 = pack_bf16x2(ln_affine(v[it][2], rstd, g.z, b.z), ln_affine(v[it][3], rstd, g.w, b.w));
      o[0] = bf16_lo(p0); o[1] = bf16_hi(p0); o[2] = bf16_lo(p1); o[3] = bf16_hi(p1);
    }
    float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const int sE = mx_scale_e8m0(amax);
    if (ok) {
      *reinterpret_cast<uint32_t*>(q + row * ldq + e) = mx_pack4(o[0], o[1], o[2], o[3], mx_inv_scale(sE));
      if ((lane & 7) == 0) sc[mx_scale_offset(row, e, E)] = (uint8_t)sE;
    }
  }
}

template <bool IN_F32, bool OUT_F32>
int launch_ln_fwd(const void* x, long ldx, const float* g, const float* b, void* y, long ldy, float* mean,
                  float* rstd, long rows, int E, float eps, hipStream_t st) {
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int it = (E + 255) / 256;
#define LN_CASE(N)                                                                                       \
  case N:                                                                                                \
    hipLaunchKernelGGL((layernorm_fwd_kernel<IN_F32, OUT_F32, N>), dim3(grid), dim3(256), 0, st, x, ldx, \
                       g, b, y, ldy, mean, rstd, rows, E, eps);                                          \
    break;
  switch (it) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    default: rvit_set_error("layernorm: E=%d unsupported (max %d)", E, LN_MAX_IT * 256); return RVIT_EUNSUPPORTED;
  }
#undef LN_CASE
  RVIT_CHECK_LAUNCH("layernorm_fwd");
  return RVIT_OK;
}

}  // namespace

int rvit_layernorm_mx_fwd_impl(const void* x, int x_f32, long ldx, const float* gamma, const float* beta, uint8_t* q,
                               long ldq, uint8_t* sc, long rows, int E, float eps, hipStream_t st) {
  RVIT_CHECK_ARG(x && q && sc && gamma && beta, "layernorm_mx: null pointer");
  RVIT_CHECK_ARG(E > 0 && (E & 127) == 0 && E <= LN_MAX_IT * 256, "layernorm_mx: E=%d must be a multiple of 128, at most %d", E,
                 LN_MAX_IT * 256);
  RVIT_CHECK_ARG((ldx & 3) == 0 && ldx >= E && (ldq & 3) == 0 && ldq >= E, "layernorm_mx: bad row pitch");
  RVIT_CHECK_ARG(rvit_aligned16(gamma) && rvit_aligned16(beta) && (((uintptr_t)x) & (x_f32 ? 15 : 7)) == 0 &&
                     (((uintptr_t)q) & 3) == 0,
                 "layernorm_mx: alignment");
  if (rows <= 0) return RVIT_OK;
  const unsigned grid = (unsigned)((rows + 3) / 4);
#define LNMX_CASE(N)                                                                                              \
  case N:                                                                                                         \
    if (x_f32)                                                                                                    \
      hipLaunchKernelGGL((layernorm_mx_fwd_kernel<true, N>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, beta, q, \
                         ldq, sc, rows, E, eps);                                                                  \
    else                                                                                                          \
      hipLaunchKernelGGL((layernorm_mx_fwd_kernel<false, N>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, beta,  \
                         q, ldq, sc, rows, E, eps);                                                               \
    break;
  switch ((E + 255) / 256) {
    LNMX_CASE(1) LNMX_CASE(2) LNMX_CASE(3) LNMX_CASE(4) LNMX_CASE(5) LNMX_CASE(6) LNMX_CASE(7) LNMX_CASE(8)
    default: return RVIT_EUNSUPPORTED;
  }
#undef LNMX_CASE
  RVIT_CHECK_LAUNCH("layernorm_mx_fwd");
  return RVIT_OK;
}

int rvit_layernorm_fwd_impl(const void* x, int x_f32, long ldx, const float* gamma, const float* beta, void* y,
                            int y_f32, long ldy, float* mean, float* rstd, long rows, int E, float eps,
                            hipStream_t st) {
  RVIT_CHECK_ARG(x && y && gamma && beta, "layernorm: null pointer");
  RVIT_CHECK_ARG(E > 0 && (E & 3) == 0, "layernorm: E=%d must be a positive multiple of 4", E);
  RVIT_CHECK_ARG((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= E && ldy >= E, "layernorm: bad row pitch");
  RVIT_CHECK_ARG(rvit_aligned16(gamma) && rvit_aligned16(beta), "layernorm: gamma/beta must be 16-byte aligned");
  RVIT_CHECK_ARG((((uintptr_t)x) & (x_f32 ? 15 : 7)) == 0 && (((uintptr_t)y) & (y_f32 ? 15 : 7)) == 0,
                 "layernorm: x/y alignment");
  if (rows <= 0) return RVIT_OK;
  if (x_f32) {
    if (y_f32) return launch_ln_fwd<true, true>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, E, eps, st);
    return launch_ln_fwd<true, false>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, E, eps, st);
  }
  if (y_f32) return launch_ln_fwd<false, true>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, E, eps, st);
  return launch_ln_fwd<false, false>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, E, eps, st);
}
