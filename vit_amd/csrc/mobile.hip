// HBM-bound kernels of the MobileNetV3 stem (the reference's `backbone='mobilenetv3'` = timm mobilenetv3_small_100,
// features_only, train_vit.py:68-74,87,145-146; inference): depthwise k x k convolution with folded BatchNorm and
// activation, the squeeze-excite global average pool, and the per-(image, channel) gate multiply.  The pointwise
// convolutions and the two squeeze-excite projections are plain GEMMs (gemm.hip) over the NHWC maps.
// All maps are NHWC bf16 with C % 8 == 0; a thread owns 8 consecutive channels of one pixel (16-byte accesses).
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ __forceinline__ void unpack8f(const uint4 u, float (&v)[8]) {
  v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
  v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8f(const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

// out[b,ho,wo,c] = act(scale[c] * sum_{kh,kw} x[b, ho*s - pad + kh, wo*s - pad + kw, c] * w[kh*K + kw][c] + bias[c])
// w: [K*K][C] bf16 (tap-major, so the 8 weights of a thread's channels are one 16-byte load per tap).
template <int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                                     bf16* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo,
                                                     int stride, int pad, int act) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * Wo * cpr;
  if (i >= total) return;
  const int chunk = (int)(i % cpr);
  long pix = i / cpr;
  const int wo = (int)(pix % Wo);
  pix /= Wo;
  const int ho = (int)(pix % Ho);
  const int b = (int)(pix / Ho);
  const int c0 = chunk * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int hi0 = ho * stride - pad, wi0 = wo * stride - pad;
#pragma unroll
  for (int kh = 0; kh < K; ++kh) {
    const int hi = hi0 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int wi = wi0 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      float xv[8], wv[8];
      unpack8f(*reinterpret_cast<const uint4*>(x + (((long)b * H + hi) * W + wi) * C + c0), xv);
      unpack8f(*reinterpret_cast<const uint4*>(w + (long)(kh * K + kw) * C + c0), wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
    }
  }
  const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = rvit_apply_act(fmaf(acc[e], sc[e], bi[e]), act);
  *reinterpret_cast<uint4*>(out + (((long)b * Ho + ho) * Wo + wo) * C + c0) = pack8f(acc);
}

// out[b, c] = mean over the HW pixels of x[b, :, c]  (bf16 out: the A operand of the squeeze-excite reduce GEMM).
// One workgroup per image; thread -> (8-channel chunk, row lane); row lanes combined through LDS in lane order.
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int HW, int C) {
  extern __shared__ float ap_red[];  // [rows_per_pass][C]
  const int cpr = C >> 3;
  const int rpp = 256 / cpr;
  const int chunk = threadIdx.x % cpr, r = threadIdx.x / cpr;
  const bf16* base = x + (long)blockIdx.x * HW * C + chunk * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (r < rpp) {
    for (int p = r; p < HW; p += rpp) {
      float v[8];
      unpack8f(*reinterpret_cast<const uint4*>(base + (long)p * C), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ap_red[r * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  const float inv = 1.0f / (float)HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += ap_red[rr * C + c];
    out[(long)blockIdx.x * C + c] = (bf16)(t * inv);
  }
}

// x[b, p, c] *= gate[b, c]  (squeeze-excite), in place
__global__ __launch_bounds__(256) void channel_scale_kernel(bf16* __restrict__ x, const bf16* __restrict__ gate, long total,
                                                            int HW, int C) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i % cpr);
  const long pix = i / cpr;
  const long b = pix / HW;
  float v[8], g[8];
  bf16* p = x + pix * C + chunk * 8;
  unpack8f(*reinterpret_cast<const uint4*>(p), v);
  unpack8f(*reinterpret_cast<const uint4*>(gate + b * C + chunk * 8), g);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= g[e];
  *reinterpret_cast<uint4*>(p) = pack8f(v);
}

}  // namespace

int rvit_dwconv_impl(const void* x, const void* w, const float* scale, const float* bias, void* out, int B, int H, int W,
                     int C, int K, int stride, int pad, int act, hipStream_t st) {
  RVIT_CHECK_ARG(x && w && scale && bias && out, "dwconv: null pointer");
  RVIT_CHECK_ARG(B >= 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "dwconv: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0)", B, H, W, C);
  RVIT_CHECK_ARG((K == 3 || K == 5) && (stride == 1 || stride == 2) && pad >= 0 && pad < K, "dwconv: K in {3,5}, stride in {1,2}");
  RVIT_CHECK_ARG(act >= RVIT_ACT_NONE && act <= RVIT_ACT_HSIGMOID && act != RVIT_ACT_GELU, "dwconv: bad activation %d", act);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w) && rvit_aligned16(out) && rvit_aligned16(scale) && rvit_aligned16(bias),
                 "dwconv: operands must be 16-byte aligned");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  const long total = (long)B * Ho * Wo * (C >> 3);
  if (total <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "dwconv: too many elements");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (K == 3) hipLaunchKernelGGL(dwconv_kernel<3>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, scale, bias, (bf16*)out, B, H, W, C, Ho, Wo, stride, pad, act);
  else hipLaunchKernelGGL(dwconv_kernel<5>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, scale, bias, (bf16*)out, B, H, W, C, Ho, Wo, stride, pad, act);
  RVIT_CHECK_LAUNCH("dwconv");
  return RVIT_OK;
}

int rvit_avgpool_impl(const void* x, void* out, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0 && C <= 2048, "avgpool: bad arguments (C %% 8 == 0, C <= 2048)");
  RVIT_CHECK_ARG(rvit_aligned16(x), "avgpool: x must be 16-byte aligned");
  if (B == 0) return RVIT_OK;
  const int rpp = 256 / (C >> 3);
  RVIT_CHECK_ARG(rpp >= 1, "avgpool: C too large");
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)B), dim3(256), (size_t)rpp * C * sizeof(float), st, (const bf16*)x, (bf16*)out, HW, C);
  RVIT_CHECK_LAUNCH("avgpool");
  return RVIT_OK;
}

int rvit_channel_scale_impl(void* x, const void* gate, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && gate && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0, "channel_scale: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(gate), "channel_scale: operands must be 16-byte aligned");
  const long total = (long)B * HW * (C >> 3);
  if (total <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "channel_scale: too many elements");
  hipLaunchKernelGGL(channel_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (bf16*)x, (const bf16*)gate, total, HW, C);
  RVIT_CHECK_LAUNCH("channel_scale");
  return RVIT_OK;
}
