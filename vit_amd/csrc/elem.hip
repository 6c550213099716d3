// HBM-bound layout / elementwise kernels around the stem and the token assembly.
#include "common.h"
#include "rvit_internal.h"

namespace {

// NCHW fp32 image batch -> zero-haloed NHWC4 bf16 (channel 3 = 0).  This is the layout the
// first stem conv (train_vit.py:87; resnet conv1 7x7/2, pad 3) gathers from: for a fixed kernel
// row the 7 taps x 3 channels of one output pixel are one contiguous 64-byte run (8 px x 4 ch).
// One thread converts 4 horizontally adjacent pixels.  Halo pixels are never written (the buffer
// is zeroed once by its owner).
__global__ __launch_bounds__(256) void input_pack_kernel(const float* __restrict__ x, bf16* __restrict__ out, int B,
                                                         int C, int H, int W, int Hp, int Wp, int pt, int pl) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int w4 = W >> 2;
  const long total = (long)B * H * w4;
  if (idx >= total) return;
  const int wq = (int)(idx % w4);
  const long bh = idx / w4;
  const int h = (int)(bh % H);
  const int b = (int)(bh / H);
  const long plane = (long)H * W;
  const float* src = x + ((long)b * C) * plane + (long)h * W + wq * 4;
  float4 c0 = *reinterpret_cast<const float4*>(src);
  float4 c1 = C > 1 ? *reinterpret_cast<const float4*>(src + plane) : make_float4(0, 0, 0, 0);
  float4 c2 = C > 2 ? *reinterpret_cast<const float4*>(src + 2 * plane) : make_float4(0, 0, 0, 0);
  uint2* dst = reinterpret_cast<uint2*>(out + (((long)b * Hp + h + pt) * Wp + wq * 4 + pl) * 4);
  dst[0] = make_uint2(pack_bf16x2(c0.x, c1.x), pack_bf16x2(c2.x, 0.f));
  dst[1] = make_uint2(pack_bf16x2(c0.y, c1.y), pack_bf16x2(c2.y, 0.f));
  dst[2] = make_uint2(pack_bf16x2(c0.z, c1.z), pack_bf16x2(c2.z, 0.f));
  dst[3] = make_uint2(pack_bf16x2(c0.w, c1.w), pack_bf16x2(c2.w, 0.f));
}

__device__ __forceinline__ uint32_t max_bf16x2(uint32_t a, uint32_t b) {
  const float lo = fmaxf(bf16_lo(a), bf16_lo(b));
  const float hi = fmaxf(bf16_hi(a), bf16_hi(b));
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}

// 3x3 / stride 2 / pad 1 max-pool on NHWC bf16 (resnet stem).  8 channels per thread.
// Optionally records the argmax tap (0..8) per element for the backward pass.
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                      uint8_t* __restrict__ argmax, int B, int H, int W, int C, int Ho,
                                                      int Wo) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = C >> 3;
  const long total = (long)B * Ho * Wo * c8;
  if (idx >= total) return;
  const int cq = (int)(idx % c8);
  long r = idx / c8;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int b = (int)(r / Ho);
  float best[8];
  int arg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; arg[k] = 0; }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const uint4 u = *reinterpret_cast<const uint4*>(x + (((long)b * H + hi) * W + wi) * C + cq * 8);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = bf16_lo(w[k]), hi2 = bf16_hi(w[k]);
        if (lo > best[2 * k]) { best[2 * k] = lo; arg[2 * k] = kh * 3 + kw; }
        if (hi2 > best[2 * k + 1]) { best[2 * k + 1] = hi2; arg[2 * k + 1] = kh * 3 + kw; }
      }
    }
  }
  uint4 o;
  o.x = (__float_as_uint(best[0]) >> 16) | (__float_as_uint(best[1]) & 0xffff0000u);
  o.y = (__float_as_uint(best[2]) >> 16) | (__float_as_uint(best[3]) & 0xffff0000u);
  o.z = (__float_as_uint(best[4]) >> 16) | (__float_as_uint(best[5]) & 0xffff0000u);
  o.w = (__float_as_uint(best[6]) >> 16) | (__float_as_uint(best[7]) & 0xffff0000u);
  const long obase = (((long)b * Ho + ho) * Wo + wo) * C + cq * 8;
  *reinterpret_cast<uint4*>(y + obase) = o;
  if (argmax) {
    uint2 a;
    a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
    *reinterpret_cast<uint2*>(argmax + obase) = a;
  }
}

// x[b, 0, :] = cls_token + pos_embed[0]   (train_vit.py:155-159), fp32 residual stream.
__global__ __launch_bounds__(256) void embed_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                        float* __restrict__ x, int B, int S, int E, DropKey dk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * E) return;
  const int e = (int)(idx % E);
  const long b = idx / E;
  float v = cls[e] + pos[e];
  if (dk.thresh) v *= drop_mult(dk, (uint32_t)(b * S * E + e));  // pos_drop, element index of the token matrix
  x[b * S * E + e] = v;
}

// out = in * mask / (1-p) with the mask of site (p, seed), element index = row * cols + col.
// Backward of the dropout sites whose forward ran inside a GEMM epilogue; in-place allowed.
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ in, T* __restrict__ out, long rows,
                                                            int cols, long ld_in, long ld_out, DropKey dk) {
  const int c4 = cols >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * c4) return;
  const long r = idx / c4;
  const int c = (int)(idx - r * c4) * 4;
  const uint32_t di = (uint32_t)(r * cols + c);
  float v[4];
  if constexpr (sizeof(T) == 4) {
    const float4 q = *reinterpret_cast<const float4*>(in + r * ld_in + c);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    const uint2 q = *reinterpret_cast<const uint2*>(in + r * ld_in + c);
    v[0] = bf16_lo(q.x); v[1] = bf16_hi(q.x); v[2] = bf16_lo(q.y); v[3] = bf16_hi(q.y);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= drop_mult(dk, di + e);
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(out + r * ld_out + c) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 w;
    w.x = pack_bf16x2(v[0], v[1]);
    w.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + r * ld_out + c) = w;
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, long n, DropKey dk) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = drop_mult(dk, (uint32_t)i) != 0.f ? 1 : 0;
}

}  // namespace

int rvit_input_pack_impl(const float* x, void* out, int B, int C, int H, int W, int Hp, int Wp, int pt, int pl,
                         hipStream_t st) {
  RVIT_CHECK_ARG(x && out, "input_pack: null pointer");
  RVIT_CHECK_ARG(B > 0 && C >= 1 && C <= 3 && (W & 3) == 0, "input_pack: bad shape B=%d C=%d W=%d", B, C, W);
  RVIT_CHECK_ARG(pt >= 0 && pl >= 0 && H + pt <= Hp && W + pl <= Wp, "input_pack: halo geometry");
  RVIT_CHECK_ARG(rvit_aligned16(x) && ((uintptr_t)out & 7) == 0, "input_pack: alignment");
  const long total = (long)B * H * (W >> 2);
  hipLaunchKernelGGL(input_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, (bf16*)out, B, C,
                     H, W, Hp, Wp, pt, pl);
  RVIT_CHECK_LAUNCH("input_pack");
  return RVIT_OK;
}

int rvit_maxpool_impl(const void* x, void* y, void* argmax, int B, int H, int W, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && y, "maxpool: null pointer");
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && (C & 7) == 0, "maxpool: bad shape");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(y), "maxpool: alignment");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16*)x,
                     (bf16*)y, (uint8_t*)argmax, B, H, W, C, Ho, Wo);
  RVIT_CHECK_LAUNCH("maxpool");
  return RVIT_OK;
}

int rvit_embed_cls_impl(const float* cls, const float* pos, float* x, int B, int S, int E, float drop_p,
                        uint64_t drop_seed, hipStream_t st) {
  RVIT_CHECK_ARG(cls && pos && x && B > 0 && S > 0 && E > 0, "embed_cls: bad arguments");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (long)B * S * E < 0xffffffffL, "embed_cls: bad dropout arguments");
  const long total = (long)B * E;
  hipLaunchKernelGGL(embed_cls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cls, pos, x, B, S, E,
                     make_drop_key(drop_p, drop_seed));
  RVIT_CHECK_LAUNCH("embed_cls");
  return RVIT_OK;
}

int rvit_dropout_apply_impl(const void* in, void* out, int f32, long rows, int cols, long ld_in, long ld_out,
                            float p, uint64_t seed, hipStream_t st) {
  RVIT_CHECK_ARG(in && out && rows >= 0 && cols > 0 && (cols & 3) == 0 && ld_in >= cols && ld_out >= cols,
                 "dropout_apply: bad shape");
  RVIT_CHECK_ARG((ld_in & 3) == 0 && (ld_out & 3) == 0 && ((uintptr_t)in & (f32 ? 15 : 7)) == 0 &&
                     ((uintptr_t)out & (f32 ? 15 : 7)) == 0,
                 "dropout_apply: alignment");
  RVIT_CHECK_ARG(p >= 0.f && p < 1.f && rows * cols < 0xffffffffL, "dropout_apply: bad p or too many elements");
  if (rows == 0) return RVIT_OK;
  const long total = rows * (cols >> 2);
  const DropKey dk = make_drop_key(p, seed);
  if (f32)
    hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)in, (float*)out, rows, cols, ld_in, ld_out, dk);
  else
    hipLaunchKernelGGL(dropout_apply_kernel<bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16*)in, (bf16*)out, rows, cols, ld_in, ld_out, dk);
  RVIT_CHECK_LAUNCH("dropout_apply");
  return RVIT_OK;
}

int rvit_dropout_mask_impl(uint8_t* out, long n, float p, uint64_t seed, hipStream_t st) {
  RVIT_CHECK_ARG(out && n >= 0 && n < 0xffffffffL && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
  if (n == 0) return RVIT_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n,
                     make_drop_key(p, seed));
  RVIT_CHECK_LAUNCH("dropout_mask");
  return RVIT_OK;
}
