// Shared device/host helpers for librvit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>
#include <mutex>

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

// Raising hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: a launcher keeps one of
// these (function-local static: initialised thread-safely) and calls ensure() before every launch.  The fast path is
// one hipGetDevice and one atomic load; the attribute is set under a mutex the first time a device (or a larger
// size) is seen, so the library stays re-entrant across threads and correct on every GPU of the process.
constexpr int RVIT_MAX_DEVICES = 64;
struct RvitLdsLimit {
  std::atomic<int> have[RVIT_MAX_DEVICES];
  std::mutex mu;
  RvitLdsLimit() { for (auto& h : have) h.store(0, std::memory_order_relaxed); }
  int ensure(std::initializer_list<const void*> kernels, int bytes, const char* what) {
    if (bytes <= 65536) return RVIT_OK;   // the default limit
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RVIT_MAX_DEVICES) {
      rvit_set_error("%s: no current device (index %d)", what, dev);
      return RVIT_ELAUNCH;
    }
    if (have[dev].load(std::memory_order_acquire) >= bytes) return RVIT_OK;
    std::lock_guard<std::mutex> lk(mu);
    if (have[dev].load(std::memory_order_relaxed) >= bytes) return RVIT_OK;
    for (const void* k : kernels) {
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e != hipSuccess) {
        rvit_set_error("%s: cannot raise the dynamic LDS limit to %d bytes on device %d: %s", what, bytes, dev,
                       hipGetErrorString(e));
        return RVIT_ELAUNCH;
      }
    }
    have[dev].store(bytes, std::memory_order_release);
    return RVIT_OK;
  }
};

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

// LayerNorm row arithmetic with explicit roundings (no floating-point contraction): every kernel that normalises a
// row (norm.hip, encoder.hip) produces the same bits, whatever the compiler would have fused in its context.
__device__ __forceinline__ float ln_mean(float sum, float inv_e) { return __fmul_rn(sum, inv_e); }
__device__ __forceinline__ float ln_sq(float q, float d) { return __fmaf_rn(d, d, q); }
__device__ __forceinline__ float ln_rstd(float sq_sum, float inv_e, float eps) {
  return 1.0f / sqrtf(__fadd_rn(__fmul_rn(sq_sum, inv_e), eps));
}
__device__ __forceinline__ float ln_affine(float d, float rstd, float g, float b) {
  return __fmaf_rn(__fmul_rn(d, rstd), g, b);
}

// erf-form GELU (torch nn.GELU(approximate='none'), activation='gelu').  The library erff costs ~45 vector-ALU
// instructions per element (two branches, both taken by a wave): in the fc1 epilogue that was 60-110 us of a 300 us
// GEMM.
// Forward: gelu(x) = x/2 (1 + erf(x / sqrt 2)) with erf(x / sqrt 2) = xc P(xc^2), xc = x clamped to +-3.5 sqrt 2
// (1 - erf(3.5) = 7e-7), P the degree-12 Chebyshev interpolant of erf(sqrt(u)/sqrt 2)/sqrt(u) on [0, 24.5] evaluated
// by Horner in t = 2u/24.5 - 1 (|coefficients| <= 0.29: no cancellation).  |error in Phi| <= 3e-7 in fp32 (outputs
// are bf16, 2^-9 relative).  No transcendental instruction: 17 multiply-adds that the compiler issues as packed
// v_pk_fma_f32 pairs, ~40 cycles per wave and element against ~95 for the rational form with v_exp_f32 + v_rcp_f32
// (7.1.26 of Abramowitz & Stegun, below), which in the fused encoder kernel was more vector-ALU time than the MFMA
// time of the two GEMMs around it.
__device__ __forceinline__ float gelu_erf(float x) {
  const float xc = fminf(fmaxf(x, -4.94974756f), 4.94974756f);
  const float t = fmaf(xc * xc, 0.0816326514f, -1.0f);
  float p = fmaf(t, 1.212488976e-03f, -2.922325628e-03f);
  p = fmaf(p, t, 2.721123397e-03f);
  p = fmaf(p, t, -4.854901228e-03f);
  p = fmaf(p, t, 1.249409094e-02f);
  p = fmaf(p, t, -2.146489173e-02f);
  p = fmaf(p, t, 3.163446859e-02f);
  p = fmaf(p, t, -4.571118578e-02f);
  p = fmaf(p, t, 6.254854053e-02f);
  p = fmaf(p, t, -8.105718344e-02f);
  p = fmaf(p, t, 1.037653387e-01f);
  p = fmaf(p, t, -1.419164836e-01f);
  p = fmaf(p, t, 2.855813503e-01f);
  const float hx = 0.5f * x;
  return fmaf(hx, xc * p, hx);
}
// Backward: d/dx gelu_erf = Phi(x) + x phi(x) = 1/2 + xc R(xc^2), the same construction as the forward: R is the
// degree-13 Chebyshev interpolant of (Phi(x) - 1/2 + x phi(x)) / x in u = x^2 on [0, 24.5], Horner in t = 2u/24.5 - 1
// (|coefficients| <= 0.15), xc = x clamped to +-3.5 sqrt 2 (the derivative is within 1e-5 of 1 / 0 beyond).  |error| <=
// 1e-6 in fp32 against the closed form (outputs are bf16).  14 multiply-adds that pack into v_pk_fma_f32 pairs and no
// transcendental: ~40 cycles per wave and element.  The previous form (7.1.26 of Abramowitz & Stegun: v_exp_f32 +
// v_rcp_f32 at a quarter of the vector rate + 15 other instructions, ~95 cycles) made the epilogue of the ViT-B
// linear2^T x gelu' product (128 outputs per lane) cost 13 us of vector-ALU time per tile against 17 us of K loop.
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float xc = fminf(fmaxf(x, -4.94974756f), 4.94974756f);
  const float t = fmaf(xc * xc, 0.0816326514f, -1.0f);
  float p = fmaf(t, -4.666657257e-03f, 1.088103004e-02f);
  p = fmaf(p, t, -7.566211462e-03f);
  p = fmaf(p, t, 1.130461103e-02f);
  p = fmaf(p, t, -3.476696418e-02f);
  p = fmaf(p, t, 5.481268282e-02f);
  p = fmaf(p, t, -6.535895903e-02f);
  p = fmaf(p, t, 7.775751320e-02f);
  p = fmaf(p, t, -8.585037863e-02f);
  p = fmaf(p, t, 8.280111830e-02f);
  p = fmaf(p, t, -7.392019825e-02f);
  p = fmaf(p, t, 6.823001217e-02f);
  p = fmaf(p, t, -7.630426704e-02f);
  p = fmaf(p, t, 1.436635906e-01f);
  return fmaf(xc, p, 0.5f);
}

// Activation codes of include/rvit.h on one fp32 value (the kernels that take `act` as a runtime argument)
__device__ __forceinline__ float rvit_apply_act(float x, int act) {
  if (act == RVIT_ACT_RELU) return fmaxf(x, 0.f);
  if (act == RVIT_ACT_GELU) return gelu_erf(x);
  if (act == RVIT_ACT_HSWISH) return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
  if (act == RVIT_ACT_HSIGMOID) return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
  return x;
}

// Counter-based dropout (nn.Dropout of train_vit.py:100,107,125 and the attention-probability
// dropout of torch functional.py:6569-6570).  Element `idx` of a dropout site keyed by (k0, k1) is
// kept iff 16 bits of an integer hash of (key, idx >> 1) reach the threshold: one 32-bit hash decides TWO
// neighbouring elements (its low half the even one, its high half the odd one).  Forward and backward
// regenerate the same decision from (key, idx), no mask is stored.  Cost matters: the hash runs once per
// output element in GEMM epilogues and on every attention probability (three times: forward and both backward
// passes), and v_mul_lo_u32 issues at a QUARTER of the vector rate.  History: two full lowbias32 rounds per ELEMENT
// (four multiplies) cost more vector-ALU time in a ViT-B linear1 epilogue than the tile's whole K loop took on the
// matrix pipe; three 32-bit multiplies per PAIR still made the hash 59 % of the vector-ALU cycles of the attention
// backward (rocprofv3 + ISA count: 48 v_mul_lo_u32 among 505 vector instructions per 32 x 32 block).  This form uses
// only full-rate instructions: two rounds, each a pair of 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24: the low
// 24 bits of the word through one constant, bits 8..31 through another, so every input bit enters) and an
// xor-shift; 12 instructions = 48 cycles per pair against 88.  Avalanche (flip one input bit, each output bit flips
// with probability 0.488-0.525 over 2e5 inputs), drop rate, lag correlations of consecutive indices and seed
// independence were checked against the previous hash before the switch (tests/test_dropout_gpu.py keeps the
// statistical checks).  The drop probability is quantised to 1/65536 (p = 0.1 -> 0.100006).
__device__ __forceinline__ uint32_t rvit_mix24(uint32_t x, uint32_t c_lo, uint32_t c_hi) {
  return __umul24(x, c_lo) + __umul24(x >> 8, c_hi);
}
// full-width finaliser (lowbias32) for the places that hash once per sample, not per element (preproc.hip)
__device__ __forceinline__ uint32_t rvit_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropKey {
  uint32_t k0, k1, thresh;  // thresh = round(p * 2^16); 0 disables
  float scale;              // 1 / (1 - p)
};
__host__ __device__ inline DropKey make_drop_key(float p, uint64_t seed) {
  DropKey k;
  // The two key words are the halves of splitmix64(seed), computed here on the host: the element hash adds k0 to the
  // element index and xors k1 in between its two rounds, so RELATED keys would give related masks (seed + 1 as k0 is
  // the same mask shifted by one pair; keys that differ only in k1 end in hashes a constant apart).  After this
  // finaliser any two distinct seeds - consecutive integers included - give unrelated (k0, k1).
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  k.k0 = (uint32_t)z;
  k.k1 = (uint32_t)(z >> 32);
  k.thresh = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
  k.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return k;
}
// a multiply that no later add is contracted with (hipcc inlines __fmul_rn as a plain, contractable `*`): left to the
// compiler, x * mult + residual becomes one fma in some kernels and not in others, and their outputs differ in the last bit
__device__ __forceinline__ float rvit_mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
// 32 decision bits for the element pair (2 * pair, 2 * pair + 1)
__device__ __forceinline__ uint32_t rvit_drop_hash(const DropKey& k, uint32_t pair) {
  uint32_t h = rvit_mix24(pair + k.k0, 0xB5297Bu, 0x68E31Du);
  h ^= h >> 15;
  h = rvit_mix24(h ^ k.k1, 0x9E3779u, 0x7FEB35u);
  return h ^ (h >> 16);
}
__device__ __forceinline__ float drop_mult(const DropKey& k, uint32_t idx) {
  const uint32_t h = rvit_drop_hash(k, idx >> 1);
  const uint32_t bits = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
  return bits >= k.thresh ? k.scale : 0.f;
}
// four consecutive elements starting at the EVEN index 2 * pair: two hashes; m[e] = `keep_value` or 0
__device__ __forceinline__ void drop_keep4(const DropKey& k, uint32_t pair, float keep_value, float (&m)[4]) {
  const uint32_t h0 = rvit_drop_hash(k, pair), h1 = rvit_drop_hash(k, pair + 1);
  m[0] = (h0 & 0xffffu) >= k.thresh ? keep_value : 0.f;
  m[1] = (h0 >> 16) >= k.thresh ? keep_value : 0.f;
  m[2] = (h1 & 0xffffu) >= k.thresh ? keep_value : 0.f;
  m[3] = (h1 >> 16) >= k.thresh ? keep_value : 0.f;
}
// eight consecutive elements starting at an EVEN index: four hashes
__device__ __forceinline__ void drop_mult8(const DropKey& k, uint32_t idx0, float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = rvit_drop_hash(k, (idx0 >> 1) + q);
    v[2 * q] = rvit_mul_rn(v[2 * q], (h & 0xffffu) >= k.thresh ? k.scale : 0.f);
    v[2 * q + 1] = rvit_mul_rn(v[2 * q + 1], (h >> 16) >= k.thresh ? k.scale : 0.f);
  }
}

// Two floats -> one dword of two bf16 (round to nearest even), `lo` in the low half.  Written as a VECTOR conversion:
// that is one v_cvt_pk_bf16_f32; two scalar casts stored into a union compile to two conversions + a shift + an or
// (four vector instructions per pair; found in round 3 in the ISA of the stem entry's epilogue).  Same rounding.
typedef __bf16 rvit_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rvit_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(rvit_f32x2{lo, hi}, rvit_bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
