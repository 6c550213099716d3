// Training pass of the MobileNetV3 stem (the reference's `backbone='mobilenetv3'` in model.train(), train_vit.py:68-74,87,
// 145-146 forward and train_vit.py:939 backward): what the inference kernels of mobile.hip do not cover.
//   - train-mode BatchNorm2d on NHWC bf16 maps of ANY channel count C % 8 == 0 (bn.hip needs C/8 to be a power of two:
//     the resnet widths) with ReLU / hard-swish / no activation and an optional residual, forward and backward;
//   - depthwise-convolution input and weight gradients;
//   - the squeeze-excite gate multiply with a separate output and an additive per-(image, channel) term (its backward),
//     and the per-(image, channel) dot product of two maps (the gradient of the gate).
// All are HBM-bound: 16-byte accesses, a thread owns 8 consecutive channels.  Every reduction over pixels is ORDERED:
// a workgroup owns a contiguous run of rows, adds its threads' sums in a fixed order in LDS and stores one partial per
// workgroup to the caller's workspace; rvit_ordered_finalize (ordered.hip) adds the partials in workgroup order.  No
// floating-point atomics: two runs of a step give the same bits.
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
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ float mt_act(float u, int act) {
  const float r6 = fminf(fmaxf(u + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
  return act == RVIT_ACT_RELU ? fmaxf(u, 0.f) : act == RVIT_ACT_HSWISH ? u * r6 : u;
}
// d act(u) / du as torch differentiates them (hardswish_backward: 0 below -3, u/3 + 1/2 up to and including 3, 1 above)
__device__ __forceinline__ float mt_act_grad(float u, int act) {
  if (act == RVIT_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == RVIT_ACT_HSWISH) return u < -3.f ? 0.f : (u <= 3.f ? u * (1.0f / 3.0f) + 0.5f : 1.f);
  return 1.f;
}

constexpr int MT_THREADS = 256;

// Per-channel batch statistics of a thread's 8 channels from the forward sums (sum y, sum y^2), as bn.hip derives them.
struct BnConst { float mean[8], rstd[8]; };
__device__ __forceinline__ BnConst bn_const(const float* __restrict__ sums, int C, int c0, long M, float eps) {
  BnConst k;
  const float inv_m = 1.0f / (float)M;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    k.mean[e] = sums[c0 + e] * inv_m;
    const float var = fmaxf(sums[C + c0 + e] * inv_m - k.mean[e] * k.mean[e], 0.f);
    k.rstd[e] = 1.0f / sqrtf(var + eps);
  }
  return k;
}

// Two per-channel sums over the rows of one workgroup's range -> part[blockIdx.x][2C].  fn(m, a, b) adds row m's
// contribution of this thread's 8 channels.  Threads are (chunk, row lane); row lanes are added in lane order.
template <typename RowFn>
__device__ __forceinline__ void channel_reduce2(long M, int C, float* __restrict__ part, RowFn&& fn) {
  extern __shared__ float mt_red[];  // [rows per pass][2C]
  const int cpr = C >> 3, rpp = MT_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r = threadIdx.x / cpr;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  const long m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
  if (r < rpp) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
    for (long m = m0 + r; m < m1; m += rpp) fn(m, a, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mt_red[r * 2 * C + chunk * 8 + e] = a[e];
      mt_red[r * 2 * C + C + chunk * 8 + e] = b[e];
    }
  }
  __syncthreads();
  float* dst = part + (long)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += MT_THREADS) {
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += mt_red[rr * 2 * C + i];
    dst[i] = t;
  }
}

__global__ __launch_bounds__(MT_THREADS) void mb_bn_stats_kernel(const bf16* __restrict__ y, long M, int C, float* __restrict__ part) {
  const int c0 = (threadIdx.x % (C >> 3)) * 8;
  channel_reduce2(M, C, part, [&](long m, float (&a)[8], float (&b)[8]) {
    float v[8];
    unpack8f(*reinterpret_cast<const uint4*>(y + m * C + c0), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += v[e]; b[e] = fmaf(v[e], v[e], b[e]); }
  });
}

// z = act(gamma * (y - mean) * rstd + beta + res); workgroup 0 also updates the running statistics (momentum, unbiased
// variance, as nn.BatchNorm2d in train()).
__global__ __launch_bounds__(MT_THREADS) void mb_bn_apply_kernel(const bf16* __restrict__ y, const float* __restrict__ sums,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const bf16* __restrict__ res, bf16* __restrict__ z, long M, int C,
                                                                 float eps, int act, float momentum,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int cpr = C >> 3;
  const long total = M * cpr;
  if (blockIdx.x == 0 && running_mean) {
    const float inv_m = 1.0f / (float)M;
    for (int c = threadIdx.x; c < C; c += MT_THREADS) {
      const float mean = sums[c] * inv_m;
      const float var = fmaxf(sums[C + c] * inv_m - mean * mean, 0.f);
      const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  for (long i = (long)blockIdx.x * MT_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * MT_THREADS) {
    const int c0 = (int)(i % cpr) * 8;
    const BnConst k = bn_const(sums, C, c0, M, eps);
    float g[8], bt[8], v[8];
    load8f(gamma + c0, g);
    load8f(beta + c0, bt);
    unpack8f(*reinterpret_cast<const uint4*>(y + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf((v[e] - k.mean[e]) * k.rstd[e], g[e], bt[e]);
    if (res) {
      float rv[8];
      unpack8f(*reinterpret_cast<const uint4*>(res + i * 8), rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mt_act(v[e], act);
    *reinterpret_cast<uint4*>(z + i * 8) = pack8f(v);
  }
}

// g = dz * act'(u), u = gamma * xhat + beta (recomputed from the saved convolution output y: nothing but y and the
// sums is kept from the forward pass);  part[wg] = (sum g, sum g * xhat)
__global__ __launch_bounds__(MT_THREADS) void mb_bn_bwd_reduce_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ y,
                                                                      const float* __restrict__ fsums, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, long M, int C, float eps,
                                                                      int act, float* __restrict__ part) {
  const int c0 = (threadIdx.x % (C >> 3)) * 8;
  const BnConst k = bn_const(fsums, C, c0, M, eps);
  float g[8], bt[8];
  load8f(gamma + c0, g);
  load8f(beta + c0, bt);
  channel_reduce2(M, C, part, [&](long m, float (&a)[8], float (&b)[8]) {
    float d[8], v[8];
    unpack8f(*reinterpret_cast<const uint4*>(dz + m * C + c0), d);
    unpack8f(*reinterpret_cast<const uint4*>(y + m * C + c0), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - k.mean[e]) * k.rstd[e];
      const float gg = d[e] * mt_act_grad(fmaf(xh, g[e], bt[e]), act);
      a[e] += gg;
      b[e] = fmaf(gg, xh, b[e]);
    }
  });
}

// dy = gamma * rstd * (g - bs1/M - xhat * bs2/M); workgroup 0: dgamma += bs2, dbeta += bs1
__global__ __launch_bounds__(MT_THREADS) void mb_bn_bwd_apply_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ y,
                                                                     const float* __restrict__ fsums, const float* __restrict__ bsums,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     bf16* __restrict__ dy, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, long M, int C, float eps, int act) {
  const int cpr = C >> 3;
  const long total = M * cpr;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += MT_THREADS) {
      dgamma[c] += bsums[C + c];
      dbeta[c] += bsums[c];
    }
  }
  const float inv_m = 1.0f / (float)M;
  for (long i = (long)blockIdx.x * MT_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * MT_THREADS) {
    const int c0 = (int)(i % cpr) * 8;
    const BnConst k = bn_const(fsums, C, c0, M, eps);
    float g[8], bt[8], d[8], v[8], s1[8], s2[8];
    load8f(gamma + c0, g);
    load8f(beta + c0, bt);
    load8f(bsums + c0, s1);
    load8f(bsums + C + c0, s2);
    unpack8f(*reinterpret_cast<const uint4*>(dz + i * 8), d);
    unpack8f(*reinterpret_cast<const uint4*>(y + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - k.mean[e]) * k.rstd[e];
      const float gg = d[e] * mt_act_grad(fmaf(xh, g[e], bt[e]), act);
      d[e] = g[e] * k.rstd[e] * (gg - s1[e] * inv_m - xh * s2[e] * inv_m);
    }
    *reinterpret_cast<uint4*>(dy + i * 8) = pack8f(d);
  }
}

// dx[b,h,w,c] = sum_{kh,kw} dy[b, (h + pad - kh)/s, (w + pad - kw)/s, c] * w[kh*K + kw][c] over the taps whose quotient is
// exact and inside the output map.  Thread = (input pixel, 8 channels); clamped unconditional loads, zeroed by select.
template <int K>
__global__ __launch_bounds__(MT_THREADS) void dwconv_dgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ w,
                                                                  bf16* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                                  int Wo, int stride, int pad) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * MT_THREADS + threadIdx.x;
  if (i >= (long)B * H * W * cpr) return;
  const int c0 = (int)(i % cpr) * 8;
  long pix = i / cpr;
  const int wi = (int)(pix % W);
  pix /= W;
  const int hi = (int)(pix % H);
  const long b = pix / H;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int kh = 0; kh < K; ++kh) {
    const int th = hi + pad - kh;
    const int ho = stride == 2 ? th >> 1 : th;
    const bool hok = th >= 0 && ho < Ho && (stride == 1 || (th & 1) == 0);
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int tw = wi + pad - kw;
      const int wo = stride == 2 ? tw >> 1 : tw;
      const bool ok = hok && tw >= 0 && wo < Wo && (stride == 1 || (tw & 1) == 0);
      uint4 u = *reinterpret_cast<const uint4*>(dy + ((b * Ho + min(max(ho, 0), Ho - 1)) * Wo + min(max(wo, 0), Wo - 1)) * C + c0);
      if (!ok) u = zero;
      float dv[8], wv[8];
      unpack8f(u, dv);
      unpack8f(*reinterpret_cast<const uint4*>(w + (long)(kh * K + kw) * C + c0), wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(dv[e], wv[e], acc[e]);
    }
  }
  *reinterpret_cast<uint4*>(dx + i * 8) = pack8f(acc);
}

// dw[c][kh*K + kw] = sum over output pixels of dy[b,ho,wo,c] * x[b, ho*s - pad + kh, wo*s - pad + kw, c].
// grid (row splits, K): blockIdx.y is the kernel row kh, so a thread carries K x 8 sums; output pixels are the rows of
// the reduction.  part[blockIdx.x] is a whole [C][K*K] image of the gradient (each kh writes its own K columns).
template <int K>
__global__ __launch_bounds__(MT_THREADS) void dwconv_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                                  float* __restrict__ part, int B, int H, int W, int C, int Ho,
                                                                  int Wo, int stride, int pad) {
  extern __shared__ float mt_red[];  // [rows per pass][K][C]
  const int cpr = C >> 3, rpp = MT_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r = threadIdx.x / cpr, c0 = chunk * 8;
  const int kh = blockIdx.y;
  const long M = (long)B * Ho * Wo;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  const long m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  if (r < rpp) {
    float acc[K][8];
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[kw][e] = 0.f;
    for (long m = m0 + r; m < m1; m += rpp) {
      const int wo = (int)(m % Wo);
      const long t = m / Wo;
      const int ho = (int)(t % Ho);
      const long b = t / Ho;
      const int hi = ho * stride - pad + kh;
      const bool hok = (unsigned)hi < (unsigned)H;
      float dv[8];
      unpack8f(*reinterpret_cast<const uint4*>(dy + m * C + c0), dv);
      const bf16* row = x + (b * H + min(max(hi, 0), H - 1)) * (long)W * C + c0;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int wi = wo * stride - pad + kw;
        uint4 u = *reinterpret_cast<const uint4*>(row + (long)min(max(wi, 0), W - 1) * C);
        if (!(hok && (unsigned)wi < (unsigned)W)) u = zero;
        float xv[8];
        unpack8f(u, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[kw][e] = fmaf(dv[e], xv[e], acc[kw][e]);
      }
    }
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
      for (int e = 0; e < 8; ++e) mt_red[(r * K + kw) * C + c0 + e] = acc[kw][e];
  }
  __syncthreads();
  float* dst = part + (long)blockIdx.x * C * K * K;
  for (int i = threadIdx.x; i < K * C; i += MT_THREADS) {
    const int kw = i / C, c = i - kw * C;
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += mt_red[(rr * K + kw) * C + c];
    dst[(long)c * K * K + kh * K + kw] = t;
  }
}

// out[b,p,c] = x[b,p,c] * gate[b,c] + add[b,c]   (add may be null; out may alias x)
__global__ __launch_bounds__(MT_THREADS) void channel_affine_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gate,
                                                                    const bf16* __restrict__ add, bf16* __restrict__ out,
                                                                    long total, int HW, int C) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * MT_THREADS + threadIdx.x;
  if (i >= total) return;
  const int c0 = (int)(i % cpr) * 8;
  const long b = (i / cpr) / HW;
  float v[8], g[8];
  unpack8f(*reinterpret_cast<const uint4*>(x + i * 8), v);
  unpack8f(*reinterpret_cast<const uint4*>(gate + b * C + c0), g);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= g[e];
  if (add) {
    float a[8];
    unpack8f(*reinterpret_cast<const uint4*>(add + b * C + c0), a);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
  }
  *reinterpret_cast<uint4*>(out + i * 8) = pack8f(v);
}

// out[b,c] = (sum over the HW pixels of a[b,p,c] * x[b,p,c]) * hsig'(gate[b,c]) (gate null: no factor): the gradient of
// the squeeze-excite gate's pre-activation.  hard-sigmoid' = 1/6 strictly inside (0, 1) of its output, 0 at the ends.
// One workgroup per image, row lanes added in order.
__global__ __launch_bounds__(MT_THREADS) void channel_dot_kernel(const bf16* __restrict__ a, const bf16* __restrict__ x,
                                                                 const bf16* __restrict__ gate, bf16* __restrict__ out, int HW,
                                                                 int C) {
  extern __shared__ float mt_red[];  // [rows per pass][C]
  const int cpr = C >> 3, rpp = MT_THREADS / cpr;
  const int chunk = threadIdx.x % cpr, r = threadIdx.x / cpr;
  const long base = (long)blockIdx.x * HW * C + chunk * 8;
  if (r < rpp) {
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int p = r; p < HW; p += rpp) {
      float av[8], xv[8];
      unpack8f(*reinterpret_cast<const uint4*>(a + base + (long)p * C), av);
      unpack8f(*reinterpret_cast<const uint4*>(x + base + (long)p * C), xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = fmaf(av[e], xv[e], s[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) mt_red[r * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += MT_THREADS) {
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += mt_red[rr * C + c];
    if (gate) {
      const float g = (float)gate[(long)blockIdx.x * C + c];
      t = (g > 0.f && g < 1.f) ? t * (1.0f / 6.0f) : 0.f;
    }
    out[(long)blockIdx.x * C + c] = (bf16)t;
  }
}

bool mt_shape_ok(long M, int C) { return M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0; }

// workgroups of a row reduction: enough to fill the chip, never more than rows / 8 or than the workspace holds
long mt_splits(long M, long elems, size_t ws_bytes) {
  long g = 512;
  if (g > (M + 7) / 8) g = (M + 7) / 8;
  const long cap = (long)(ws_bytes / 4 / (size_t)elems);
  if (g > cap) g = cap;
  if (g < 1) g = cap >= 1 ? 1 : 0;
  return g;
}

}  // namespace

int rvit_mb_bn_stats_impl(const void* y, long M, int C, float* sums, void* ws, size_t ws_bytes, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && ws && mt_shape_ok(M, C) && rvit_aligned16(y) && rvit_aligned16(ws), "mb_bn_stats: bad arguments (M=%ld C=%d; a workspace is required)", M, C);
  const long g = mt_splits(M, 2L * C, ws_bytes);
  RVIT_CHECK_ARG(g >= 1, "mb_bn_stats: workspace too small");
  const int rpp = MT_THREADS / (C >> 3);
  hipLaunchKernelGGL(mb_bn_stats_kernel, dim3((unsigned)g), dim3(MT_THREADS), (size_t)rpp * 2 * C * sizeof(float), st,
                     (const bf16*)y, M, C, (float*)ws);
  RVIT_CHECK_LAUNCH("mb_bn_stats");
  return rvit_ordered_finalize((const float*)ws, (int)g, 2L * C, sums, 1, 2 * C, 2L * C, st);
}

int rvit_mb_bn_apply_impl(const void* y, const float* sums, const float* gamma, const float* beta, const void* res, void* z,
                          long M, int C, float eps, int act, float momentum, float* running_mean, float* running_var,
                          hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && gamma && beta && z && mt_shape_ok(M, C), "mb_bn_apply: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(y) && rvit_aligned16(z) && rvit_aligned16(res) && rvit_aligned16(gamma) && rvit_aligned16(beta),
                 "mb_bn_apply: alignment");
  RVIT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "mb_bn_apply: running stats come in pairs");
  RVIT_CHECK_ARG(act == RVIT_ACT_NONE || act == RVIT_ACT_RELU || act == RVIT_ACT_HSWISH, "mb_bn_apply: activation none, relu or hswish");
  const long total = M * (C >> 3);
  long blocks = (total + MT_THREADS - 1) / MT_THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mb_bn_apply_kernel, dim3((unsigned)blocks), dim3(MT_THREADS), 0, st, (const bf16*)y, sums, gamma, beta,
                     (const bf16*)res, (bf16*)z, M, C, eps, act, momentum, running_mean, running_var);
  RVIT_CHECK_LAUNCH("mb_bn_apply");
  return RVIT_OK;
}

int rvit_mb_bn_bwd_impl(const void* dz, const void* y, const float* fsums, float* bsums, const float* gamma, const float* beta,
                        void* dy, float* dgamma, float* dbeta, long M, int C, float eps, int act, void* ws, size_t ws_bytes,
                        hipStream_t st) {
  RVIT_CHECK_ARG(dz && y && fsums && bsums && gamma && beta && dy && dgamma && dbeta && ws && mt_shape_ok(M, C),
                 "mb_bn_bwd: bad arguments (a workspace is required)");
  RVIT_CHECK_ARG(rvit_aligned16(dz) && rvit_aligned16(y) && rvit_aligned16(dy) && rvit_aligned16(ws) && rvit_aligned16(gamma) &&
                 rvit_aligned16(beta) && rvit_aligned16(bsums), "mb_bn_bwd: alignment");
  RVIT_CHECK_ARG(act == RVIT_ACT_NONE || act == RVIT_ACT_RELU || act == RVIT_ACT_HSWISH, "mb_bn_bwd: activation none, relu or hswish");
  const long g = mt_splits(M, 2L * C, ws_bytes);
  RVIT_CHECK_ARG(g >= 1, "mb_bn_bwd: workspace too small");
  const int rpp = MT_THREADS / (C >> 3);
  hipLaunchKernelGGL(mb_bn_bwd_reduce_kernel, dim3((unsigned)g), dim3(MT_THREADS), (size_t)rpp * 2 * C * sizeof(float), st,
                     (const bf16*)dz, (const bf16*)y, fsums, gamma, beta, M, C, eps, act, (float*)ws);
  RVIT_CHECK_LAUNCH("mb_bn_bwd_reduce");
  if (const int rc = rvit_ordered_finalize((const float*)ws, (int)g, 2L * C, bsums, 1, 2 * C, 2L * C, st)) return rc;
  const long total = M * (C >> 3);
  long blocks = (total + MT_THREADS - 1) / MT_THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mb_bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(MT_THREADS), 0, st, (const bf16*)dz, (const bf16*)y,
                     fsums, bsums, gamma, beta, (bf16*)dy, dgamma, dbeta, M, C, eps, act);
  RVIT_CHECK_LAUNCH("mb_bn_bwd_apply");
  return RVIT_OK;
}

int rvit_dwconv_dgrad_impl(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                           hipStream_t st) {
  RVIT_CHECK_ARG(dy && w && dx && B >= 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "dwconv_dgrad: bad shape");
  RVIT_CHECK_ARG((K == 3 || K == 5) && (stride == 1 || stride == 2) && pad >= 0 && pad < K, "dwconv_dgrad: K in {3,5}, stride in {1,2}");
  RVIT_CHECK_ARG(rvit_aligned16(dy) && rvit_aligned16(w) && rvit_aligned16(dx), "dwconv_dgrad: alignment");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  const long total = (long)B * H * W * (C >> 3);
  if (total <= 0 || Ho <= 0 || Wo <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + MT_THREADS - 1) / MT_THREADS < 0x7fffffffL, "dwconv_dgrad: too many elements");
  const dim3 grid((unsigned)((total + MT_THREADS - 1) / MT_THREADS));
  if (K == 3) hipLaunchKernelGGL(dwconv_dgrad_kernel<3>, grid, dim3(MT_THREADS), 0, st, (const bf16*)dy, (const bf16*)w, (bf16*)dx, B, H, W, C, Ho, Wo, stride, pad);
  else hipLaunchKernelGGL(dwconv_dgrad_kernel<5>, grid, dim3(MT_THREADS), 0, st, (const bf16*)dy, (const bf16*)w, (bf16*)dx, B, H, W, C, Ho, Wo, stride, pad);
  RVIT_CHECK_LAUNCH("dwconv_dgrad");
  return RVIT_OK;
}

int rvit_dwconv_wgrad_impl(const void* dy, const void* x, float* dw, int B, int H, int W, int C, int K, int stride, int pad,
                           void* ws, size_t ws_bytes, hipStream_t st) {
  RVIT_CHECK_ARG(dy && x && dw && ws && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && C <= 2048, "dwconv_wgrad: bad shape (a workspace is required)");
  RVIT_CHECK_ARG((K == 3 || K == 5) && (stride == 1 || stride == 2) && pad >= 0 && pad < K, "dwconv_wgrad: K in {3,5}, stride in {1,2}");
  RVIT_CHECK_ARG(rvit_aligned16(dy) && rvit_aligned16(x) && rvit_aligned16(ws), "dwconv_wgrad: alignment");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  RVIT_CHECK_ARG(Ho > 0 && Wo > 0, "dwconv_wgrad: empty output");
  const long M = (long)B * Ho * Wo;
  const long g = mt_splits(M, (long)C * K * K, ws_bytes);
  RVIT_CHECK_ARG(g >= 1, "dwconv_wgrad: workspace too small");
  const int rpp = MT_THREADS / (C >> 3);
  const size_t lds = (size_t)rpp * K * C * sizeof(float);
  if (K == 3) hipLaunchKernelGGL(dwconv_wgrad_kernel<3>, dim3((unsigned)g, 3), dim3(MT_THREADS), lds, st, (const bf16*)dy, (const bf16*)x, (float*)ws, B, H, W, C, Ho, Wo, stride, pad);
  else hipLaunchKernelGGL(dwconv_wgrad_kernel<5>, dim3((unsigned)g, 5), dim3(MT_THREADS), lds, st, (const bf16*)dy, (const bf16*)x, (float*)ws, B, H, W, C, Ho, Wo, stride, pad);
  RVIT_CHECK_LAUNCH("dwconv_wgrad");
  return rvit_ordered_finalize((const float*)ws, (int)g, (long)C * K * K, dw, 1, C * K * K, (long)C * K * K, st);
}

int rvit_channel_affine_impl(const void* x, const void* gate, const void* add, void* out, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && gate && out && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0, "channel_affine: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(gate) && rvit_aligned16(add) && rvit_aligned16(out), "channel_affine: alignment");
  const long total = (long)B * HW * (C >> 3);
  if (total <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + MT_THREADS - 1) / MT_THREADS < 0x7fffffffL, "channel_affine: too many elements");
  hipLaunchKernelGGL(channel_affine_kernel, dim3((unsigned)((total + MT_THREADS - 1) / MT_THREADS)), dim3(MT_THREADS), 0, st,
                     (const bf16*)x, (const bf16*)gate, (const bf16*)add, (bf16*)out, total, HW, C);
  RVIT_CHECK_LAUNCH("channel_affine");
  return RVIT_OK;
}

int rvit_channel_dot_impl(const void* a, const void* x, const void* gate, void* out, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(a && x && out && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0 && C <= 2048, "channel_dot: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(a) && rvit_aligned16(x), "channel_dot: alignment");
  if (B == 0) return RVIT_OK;
  const int rpp = MT_THREADS / (C >> 3);
  hipLaunchKernelGGL(channel_dot_kernel, dim3((unsigned)B), dim3(MT_THREADS), (size_t)rpp * C * sizeof(float), st,
                     (const bf16*)a, (const bf16*)x, (const bf16*)gate, (bf16*)out, HW, C);
  RVIT_CHECK_LAUNCH("channel_dot");
  return RVIT_OK;
}
