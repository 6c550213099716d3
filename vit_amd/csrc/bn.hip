// Training-mode BatchNorm2d (+ residual add + ReLU) on NHWC bf16 maps, forward and backward, and
// the max-pool backward.  Replaces the BatchNorm2d / ReLU / residual-add / MaxPool2d modules of the
// timm resnet stem (train_vit.py:87,145-146) in train() mode and their autograd (train_vit.py:939).
// All kernels are HBM-bound: 16-byte accesses, 8 channels per thread, per-channel partial sums
// reduced in LDS and added with one global atomic per channel per workgroup.
//
//   forward : sums[c] = (sum y, sum y^2)                     (bn_stats)
//             z = relu( gamma*(y-mean)*rstd + beta + res )    (bn_apply; workgroup 0 also updates
//                                                              running_mean / running_var)
//   backward: dzm = dz * (z > 0);  bs[c] = (sum dzm, sum dzm*xhat)          (bn_bwd_reduce)
//             dy = gamma*rstd*(dzm - bs1/M - xhat*bs2/M); dgamma += bs2; dbeta += bs1  (bn_bwd_apply)
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
  v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
  v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

// grid-stride row partition shared by the reduction kernels
struct RowSpan { long m0, m1; int chunk, r, rows_per_pass; };
__device__ __forceinline__ RowSpan row_span(long M, int C) {
  RowSpan s;
  const int cpr = C >> 3;
  s.rows_per_pass = 256 / cpr;
  s.chunk = threadIdx.x % cpr;
  s.r = threadIdx.x / cpr;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  s.m0 = (long)blockIdx.x * per;
  s.m1 = s.m0 + per < M ? s.m0 + per : M;
  return s;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16* __restrict__ y, long M, int C,
                                                       float* __restrict__ sums) {
  extern __shared__ float red[];  // [2*C]
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
  __syncthreads();
  const RowSpan s = row_span(M, C);
  float a[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = 0.f; q[k] = 0.f; }
  if (s.r < s.rows_per_pass)
    for (long m = s.m0 + s.r; m < s.m1; m += s.rows_per_pass) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(y + m * C + s.chunk * 8), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += v[k]; q[k] += v[k] * v[k]; }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    atomicAdd(&red[s.chunk * 8 + k], a[k]);
    atomicAdd(&red[C + s.chunk * 8 + k], q[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(sums + i, red[i]);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ y, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const bf16* __restrict__ res, bf16* __restrict__ z, long M, int C,
                                                       float eps, int relu, float momentum,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var) {
  const int cpr = C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long m = idx / cpr;
  const int c0 = (int)(idx - m * cpr) * 8;
  if (m >= M) return;
  const float inv_m = 1.0f / (float)M;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float mean = sums[c0 + k] * inv_m;
    const float var = fmaxf(sums[C + c0 + k] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    sc[k] = gamma[c0 + k] * rstd;
    sh[k] = beta[c0 + k] - mean * sc[k];
    if (m == 0 && running_mean) {  // exactly one thread per channel: row 0
      const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      running_mean[c0 + k] = (1.f - momentum) * running_mean[c0 + k] + momentum * mean;
      running_var[c0 + k] = (1.f - momentum) * running_var[c0 + k] + momentum * unbiased;
    }
  }
  float v[8];
  unpack8(*reinterpret_cast<const uint4*>(y + m * C + c0), v);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
  if (res) {
    float r[8];
    unpack8(*reinterpret_cast<const uint4*>(res + m * C + c0), r);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += r[k];
  }
  if (relu) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
  }
  *reinterpret_cast<uint4*>(z + m * C + c0) = pack8(v);
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ z,
                                                            const bf16* __restrict__ y, const float* __restrict__ fsums,
                                                            long M, int C, float eps, int relu,
                                                            float* __restrict__ bsums) {
  extern __shared__ float red[];  // [2*C]
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
  __syncthreads();
  const RowSpan s = row_span(M, C);
  const float inv_m = 1.0f / (float)M;
  float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = s.chunk * 8 + k;
    mean[k] = fsums[c] * inv_m;
    const float var = fmaxf(fsums[C + c] * inv_m - mean[k] * mean[k], 0.f);
    rstd[k] = 1.0f / sqrtf(var + eps);
    s1[k] = 0.f; s2[k] = 0.f;
  }
  if (s.r < s.rows_per_pass)
    for (long m = s.m0 + s.r; m < s.m1; m += s.rows_per_pass) {
      const long o = m * C + s.chunk * 8;
      float d[8], yv[8];
      unpack8(*reinterpret_cast<const uint4*>(dz + o), d);
      unpack8(*reinterpret_cast<const uint4*>(y + o), yv);
      if (relu) {
        float zv[8];
        unpack8(*reinterpret_cast<const uint4*>(z + o), zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s1[k] += d[k];
        s2[k] += d[k] * (yv[k] - mean[k]) * rstd[k];
      }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    atomicAdd(&red[s.chunk * 8 + k], s1[k]);
    atomicAdd(&red[C + s.chunk * 8 + k], s2[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(bsums + i, red[i]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ z,
                                                           const bf16* __restrict__ y, const float* __restrict__ fsums,
                                                           const float* __restrict__ bsums,
                                                           const float* __restrict__ gamma, bf16* __restrict__ dy,
                                                           bf16* __restrict__ dzm_out, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long M, int C, float eps,
                                                           int relu) {
  const int cpr = C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long m = idx / cpr;
  const int c0 = (int)(idx - m * cpr) * 8;
  if (m >= M) return;
  const float inv_m = 1.0f / (float)M;
  const long o = m * C + c0;
  float d[8], yv[8];
  unpack8(*reinterpret_cast<const uint4*>(dz + o), d);
  unpack8(*reinterpret_cast<const uint4*>(y + o), yv);
  if (relu) {
    float zv[8];
    unpack8(*reinterpret_cast<const uint4*>(z + o), zv);
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
  }
  if (dzm_out) *reinterpret_cast<uint4*>(dzm_out + o) = pack8(d);
  float out[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const float mean = fsums[c] * inv_m;
    const float var = fmaxf(fsums[C + c] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float xhat = (yv[k] - mean) * rstd;
    const float b1 = bsums[c], b2 = bsums[C + c];
    out[k] = gamma[c] * rstd * (d[k] - b1 * inv_m - xhat * b2 * inv_m);
    if (m == 0) {  // one thread per channel
      dgamma[c] += b2;
      dbeta[c] += b1;
    }
  }
  *reinterpret_cast<uint4*>(dy + o) = pack8(out);
}

// dx[b,hi,wi,c] = sum over the (<= 4) 3x3/2 windows containing (hi,wi) whose recorded argmax tap
// is this pixel of dpool.  Gather form: no atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16* __restrict__ dpool,
                                                          const uint8_t* __restrict__ argmax, bf16* __restrict__ dx,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = C >> 3;
  const long total = (long)B * H * W * c8;
  if (idx >= total) return;
  const int cq = (int)(idx % c8);
  long r = idx / c8;
  const int wi = (int)(r % W);
  r /= W;
  const int hi = (int)(r % H);
  const int b = (int)(r / H);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hn = hi + 1 - kh;
    if (hn < 0 || (hn & 1) || (hn >> 1) >= Ho) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wn = wi + 1 - kw;
      if (wn < 0 || (wn & 1) || (wn >> 1) >= Wo) continue;
      const long o = ((((long)b * Ho + (hn >> 1)) * Wo + (wn >> 1)) * C) + cq * 8;
      const uint2 am = *reinterpret_cast<const uint2*>(argmax + o);
      float d[8];
      unpack8(*reinterpret_cast<const uint4*>(dpool + o), d);
      const int tap = kh * 3 + kw;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int a = (int)(((k < 4 ? am.x : am.y) >> (8 * (k & 3))) & 0xff);
        if (a == tap) acc[k] += d[k];
      }
    }
  }
  *reinterpret_cast<uint4*>(dx + (((long)b * H + hi) * W + wi) * C + cq * 8) = pack8(acc);
}

bool bn_shape_ok(long M, int C) { return M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0 && (256 % (C >> 3)) == 0; }

}  // namespace

int rvit_bn_stats_impl(const void* y, long M, int C, float* sums, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && bn_shape_ok(M, C) && rvit_aligned16(y), "bn_stats: bad arguments (M=%ld C=%d)", M, C);
  const int rows_per_pass = 256 / (C >> 3);
  long blocks = (M + rows_per_pass * 8 - 1) / (rows_per_pass * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), st, (const bf16*)y, M, C,
                     sums);
  RVIT_CHECK_LAUNCH("bn_stats");
  return RVIT_OK;
}

int rvit_bn_apply_impl(const void* y, const float* sums, const float* gamma, const float* beta, const void* res,
                       void* z, long M, int C, float eps, int relu, float momentum, float* running_mean,
                       float* running_var, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && gamma && beta && z && bn_shape_ok(M, C), "bn_apply: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(y) && rvit_aligned16(z) && (!res || rvit_aligned16(res)), "bn_apply: alignment");
  RVIT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_apply: running stats come in pairs");
  const long total = M * (C >> 3);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16*)y, sums,
                     gamma, beta, (const bf16*)res, (bf16*)z, M, C, eps, relu, momentum, running_mean, running_var);
  RVIT_CHECK_LAUNCH("bn_apply");
  return RVIT_OK;
}

int rvit_bn_bwd_reduce_impl(const void* dz, const void* z, const void* y, const float* fsums, long M, int C, float eps,
                            int relu, float* bsums, hipStream_t st) {
  RVIT_CHECK_ARG(dz && y && fsums && bsums && (!relu || z) && bn_shape_ok(M, C), "bn_bwd_reduce: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(dz) && rvit_aligned16(y) && (!z || rvit_aligned16(z)), "bn_bwd_reduce: alignment");
  const int rows_per_pass = 256 / (C >> 3);
  long blocks = (M + rows_per_pass * 8 - 1) / (rows_per_pass * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float), st,
                     (const bf16*)dz, (const bf16*)z, (const bf16*)y, fsums, M, C, eps, relu, bsums);
  RVIT_CHECK_LAUNCH("bn_bwd_reduce");
  return RVIT_OK;
}

int rvit_bn_bwd_apply_impl(const void* dz, const void* z, const void* y, const float* fsums, const float* bsums,
                           const float* gamma, void* dy, void* dzm_out, float* dgamma, float* dbeta, long M, int C,
                           float eps, int relu, hipStream_t st) {
  RVIT_CHECK_ARG(dz && y && fsums && bsums && gamma && dy && dgamma && dbeta && (!relu || z) && bn_shape_ok(M, C),
                 "bn_bwd_apply: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(dz) && rvit_aligned16(y) && rvit_aligned16(dy) && (!z || rvit_aligned16(z)) &&
                     (!dzm_out || rvit_aligned16(dzm_out)),
                 "bn_bwd_apply: alignment");
  const long total = M * (C >> 3);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16*)dz,
                     (const bf16*)z, (const bf16*)y, fsums, bsums, gamma, (bf16*)dy, (bf16*)dzm_out, dgamma, dbeta, M,
                     C, eps, relu);
  RVIT_CHECK_LAUNCH("bn_bwd_apply");
  return RVIT_OK;
}

int rvit_maxpool_bwd_impl(const void* dpool, const void* argmax, void* dx, int B, int H, int W, int C, hipStream_t st) {
  RVIT_CHECK_ARG(dpool && argmax && dx && B > 0 && H > 0 && W > 0 && (C & 7) == 0, "maxpool_bwd: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(dpool) && rvit_aligned16(dx) && (((uintptr_t)argmax) & 7) == 0, "maxpool_bwd: alignment");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * H * W * (C >> 3);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16*)dpool,
                     (const uint8_t*)argmax, (bf16*)dx, B, H, W, C, Ho, Wo);
  RVIT_CHECK_LAUNCH("maxpool_bwd");
  return RVIT_OK;
}
