// MXFP8 (OCP microscaling: e4m3 values, one E8M0 scale per 32 elements) helpers shared by the kernels that emit it:
// the MX GEMM epilogue and the quantiser (gemm_mx.hip) and the MX-output LayerNorm (norm.hip).
#pragma once
#include "common.h"

// E8M0 scale of an MX block with absolute maximum `amax`: 2^(floor(log2 amax) - 8) (8 = emax of e4m3),
// biased by 127; amax == 0 -> scale 2^-127 (all elements quantise to 0).
__device__ __forceinline__ int mx_scale_e8m0(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xff);  // biased exponent of amax (0 for 0 / denormals)
  int s = e - 8;
  return s < 0 ? 0 : s;
}
// 4 floats -> 4 packed e4m3 bytes after multiplying by 2^-scale (exact) and saturating to +-448
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d, float inv_scale) {
  a = fminf(fmaxf(a * inv_scale, -448.f), 448.f);
  b = fminf(fmaxf(b * inv_scale, -448.f), 448.f);
  c = fminf(fmaxf(c * inv_scale, -448.f), 448.f);
  d = fminf(fmaxf(d * inv_scale, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
// byte offset of the scale of (row r, element k) in the tiled scale matrix of a [rows, K] operand
__device__ __forceinline__ long mx_scale_offset(long r, int k, int K) {
  const int kt = k >> 7, nk4 = ((K >> 7) + 3) >> 2;
  return ((((r >> 4) * nk4 + (kt >> 2)) * 64 + (r & 15) * 4 + ((k >> 5) & 3)) << 2) + (kt & 3);
}
__device__ __forceinline__ float mx_inv_scale(int s) { return __uint_as_float((uint32_t)(254 - s) << 23); }  // 2^(127-s)
