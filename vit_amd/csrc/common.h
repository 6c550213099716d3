// Shared device/host helpers for librvit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include "../../include/rvit.h"  // error / dtype / activation codes

// thread-local last-error text (api.hip owns the storage)
void rvit_set_error(const char* fmt, ...);

#define RVIT_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      rvit_set_error(__VA_ARGS__);           \
      return RVIT_EINVAL;                    \
    }                                        \
  } while (0)

#define RVIT_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      rvit_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return RVIT_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

static inline bool rvit_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

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

// exact-erf GELU (torch nn.GELU(approximate='none'), activation='gelu')
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx gelu_erf
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  union { bf16 h[2]; uint32_t u; } r;
  r.h[0] = (bf16)lo;
  r.h[1] = (bf16)hi;
  return r.u;
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
