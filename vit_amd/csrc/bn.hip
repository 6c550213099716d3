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
#include <stdlib.h>
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

// ReLU sign bits of 8 packed bf16 outputs (bit k = element k > 0) and their use on a gradient: the backward
// kernels then read one byte per 8 channels instead of the 16-byte forward output they only need the sign of.
__device__ __forceinline__ unsigned mask8(const uint4 o) {
  unsigned m = 0;
  const unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m |= ((short)(w[i] & 0xffffu) > 0 ? 1u : 0u) << (2 * i);
    m |= ((int)w[i] > 0xffff ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}
// dy = a * dzm + b * y + c with the two roundings written out (explicit fused multiply-adds): every kernel that applies
// it gives the same bits, whatever the compiler would have contracted in its context
__device__ __forceinline__ float bn_dy(float a, float d, float b, float y, float c) { return __fmaf_rn(a, d, __fmaf_rn(b, y, c)); }
__device__ __forceinline__ void apply_mask8(float (&d)[8], unsigned m) {
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = (m >> k) & 1u ? d[k] : 0.f;
}

// grid-stride row partition shared by the reduction kernels
struct RowSpan { long m0, m1; int chunk, r, rows_per_pass; };
__device__ __forceinline__ RowSpan row_span(long M, int C, int threads = 256) {
  RowSpan s;
  const int cpr = C >> 3;
  s.rows_per_pass = threads / cpr;
  s.chunk = threadIdx.x % cpr;
  s.r = threadIdx.x / cpr;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  s.m0 = (long)blockIdx.x * per;
  s.m1 = s.m0 + per < M ? s.m0 + per : M;
  return s;
}

// Every kernel below gives a thread one fixed 8-channel chunk and a strided set of rows of the
// workgroup's contiguous row range: the per-channel constants are derived once per thread, a pass of
// the workgroup touches 4 KB of contiguous memory, and UNROLL rows are in flight per thread.
// The two reduction kernels run 1024 threads per workgroup and at most one workgroup per CU: all
// per-channel totals live in a few cache lines of one L2 channel, whose atomic rate (a few per clock)
// would otherwise dominate.  Lanes of a wave that share a channel chunk combine by shuffles first.
constexpr int RED_THREADS = 1024;

__device__ __forceinline__ void red_commit(float* red, int C, int chunk, float (&a)[8], float (&b)[8]) {
  const int cpr = C >> 3, lane = threadIdx.x & 63;
  for (int off = 32; off >= cpr; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] += __shfl_xor(a[k], off, 64);
      b[k] += __shfl_xor(b[k], off, 64);
    }
  }
  if (cpr >= 64 || lane < cpr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      atomicAdd(&red[chunk * 8 + k], a[k]);
      atomicAdd(&red[C + chunk * 8 + k], b[k]);
    }
  }
}

// Ordered form of the same reduction (ws != null): no floating-point atomics anywhere, so the 2*C totals are the
// same bits on every run.  BatchNorm statistics feed every later activation: through bf16 storage and ReLU a last-bit
// difference in one mean flips signs of near-zero pre-activations and, on a random-weight stem at batch 8, moves
// gradients by ~20 % between two runs of the same step (tests/test_dp_gpu.py measured it).
//   1. lanes that share a channel chunk combine by the xor butterfly (fixed order);
//   2. every (wave | row group) stores its 2*C values to its own LDS slot, thread i adds the slots in slot order;
//   3. the workgroup stores its 2*C partials to ws[64 + blockIdx.x * 2C ..]; the workgroup that takes the LAST ticket
//      of the counter ws[0] adds the partials in workgroup order into `sums` and re-arms the counter.
// ws: (64 + gridDim.x * 2 * C) floats, ws[0] zero before the first call; sequential calls on one stream may share it.
__device__ __forceinline__ int det_slots(int C) { return (C >> 3) < 64 ? RED_THREADS / 64 : RED_THREADS / (C >> 3); }

// p[0] + p[stride] + ... (n terms) added strictly in order, with up to 32 loads in flight: the partials come from HBM /
// L2 at ~1 us a round trip and this runs in ONE workgroup at the tail of the kernel (eight in flight cost 8 round trips
// for the 64 partials of a 512-channel map; the order of the additions, and so every bit, is the same)
__device__ __forceinline__ float det_sum_run(const float* __restrict__ p, long stride, int n) {
  float t = 0.f;
  int k = 0;
  for (; k + 32 <= n; k += 32) {
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = __builtin_nontemporal_load(p + (long)(k + j) * stride);
#pragma unroll
    for (int j = 0; j < 32; ++j) t += v[j];
  }
  for (; k + 8 <= n; k += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + (long)(k + j) * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) t += v[j];
  }
  for (; k < n; ++k) t += __builtin_nontemporal_load(p + (long)k * stride);
  return t;
}

__device__ __forceinline__ void red_commit_det(float* red, int C, int chunk, float (&a)[8], float (&b)[8],
                                               float* __restrict__ ws, float* __restrict__ sums) {
  __shared__ unsigned ticket;
  const int cpr = C >> 3, lane = threadIdx.x & 63;
  for (int off = 32; off >= cpr; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] += __shfl_xor(a[k], off, 64);
      b[k] += __shfl_xor(b[k], off, 64);
    }
  }
  const int slot = cpr < 64 ? (int)(threadIdx.x >> 6) : (int)(threadIdx.x / cpr);
  if (cpr >= 64 || lane < cpr) {
    float* r = red + (long)slot * 2 * C;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      r[chunk * 8 + k] = a[k];
      r[C + chunk * 8 + k] = b[k];
    }
  }
  __syncthreads();
  const int nslots = det_slots(C), cols = 2 * C, nblk = (int)gridDim.x;
  float* part = ws + 64 + (long)blockIdx.x * cols;
  for (int i = threadIdx.x; i < cols; i += RED_THREADS) {
    float t = 0.f;
    for (int sl = 0; sl < nslots; ++sl) t += red[(long)sl * cols + i];
    part[i] = t;
  }
  __syncthreads();   // every thread's partial stores have been issued and waited for
  if (threadIdx.x == 0) {
    __threadfence();   // release (one L2 write-back per workgroup, not one per wave)
    ticket = atomicAdd(reinterpret_cast<unsigned*>(ws), 1u);
  }
  __syncthreads();
  if (ticket != (unsigned)nblk - 1u) return;
  __threadfence();   // acquire: the other workgroups' partials
  // The last workgroup adds the partials in a fixed association: G thread groups each add a contiguous run of
  // workgroups in order (det_sum_run: up to 32 loads in flight), then the G group
  // totals are added in group order.
  const float* all = ws + 64;
  if (cols <= RED_THREADS) {
    const int G = RED_THREADS / cols, g = (int)threadIdx.x / cols, i = (int)threadIdx.x - g * cols;
    const int per = (nblk + G - 1) / G;
    float t = 0.f;
    if (g < G) {
      const int blk = g * per;
      const int end = blk + per < nblk ? blk + per : nblk;
      if (end > blk) t = det_sum_run(all + (long)blk * cols + i, cols, end - blk);
    }
    __syncthreads();   // the slot values in `red` are dead
    if (g < G) red[g * cols + i] = t;
    __syncthreads();
    if (g == 0) {
      float tot = 0.f;
      for (int gg = 0; gg < G; ++gg) tot += red[gg * cols + i];
      sums[i] += tot;
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += RED_THREADS) sums[i] += det_sum_run(all + i, cols, nblk);
  }
  if (threadIdx.x == 0) *reinterpret_cast<unsigned*>(ws) = 0u;   // re-armed for the next call on this stream
}

__global__ __launch_bounds__(RED_THREADS) void bn_stats_kernel(const bf16* __restrict__ y, long M, int C,
                                                               float* __restrict__ sums, float* __restrict__ ws) {
  extern __shared__ float red[];  // [2*C]; ordered form: [slots][2*C]
  if (!ws) {
    for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) red[i] = 0.f;
    __syncthreads();
  }
  const RowSpan s = row_span(M, C, RED_THREADS);
  const long step = s.rows_per_pass;
  const bf16* base = y + s.chunk * 8;
  float a[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = 0.f; q[k] = 0.f; }
  long m = s.m0 + s.r;
  for (; m + 3 * step < s.m1; m += 4 * step) {
    uint4 u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const uint4*>(base + (m + j * step) * C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
      unpack8(u[j], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += v[k]; q[k] += v[k] * v[k]; }
    }
  }
  for (; m < s.m1; m += step) {
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(base + m * C), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] += v[k]; q[k] += v[k] * v[k]; }
  }
  if (ws) {
    red_commit_det(red, C, s.chunk, a, q, ws, sums);
    return;
  }
  red_commit(red, C, s.chunk, a, q);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) atomicAdd(sums + i, red[i]);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16* __restrict__ y, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const bf16* __restrict__ res, bf16* __restrict__ z, long M, int C,
                                                       float eps, int relu, float momentum,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var,
                                                       uint8_t* __restrict__ mask) {
  const RowSpan s = row_span(M, C);
  const int cpr = C >> 3;
  const long step = s.rows_per_pass;
  const int c0 = s.chunk * 8;
  const float inv_m = 1.0f / (float)M;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float mean = sums[c0 + k] * inv_m;
    const float var = fmaxf(sums[C + c0 + k] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    sc[k] = gamma[c0 + k] * rstd;
    sh[k] = beta[c0 + k] - mean * sc[k];
    if (blockIdx.x == 0 && s.r == 0 && running_mean) {  // exactly one thread per channel
      const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      running_mean[c0 + k] = (1.f - momentum) * running_mean[c0 + k] + momentum * mean;
      running_var[c0 + k] = (1.f - momentum) * running_var[c0 + k] + momentum * unbiased;
    }
  }
  const float lo = relu ? 0.f : -INFINITY;
  long m = s.m0 + s.r;
  for (; m + 3 * step < s.m1; m += 4 * step) {
    uint4 u[4], r4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long o = (m + j * step) * C + c0;
      u[j] = *reinterpret_cast<const uint4*>(y + o);
      if (res) r4[j] = *reinterpret_cast<const uint4*>(res + o);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
      unpack8(u[j], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
      if (res) {
        float r[8];
        unpack8(r4[j], r);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += r[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], lo);
      const uint4 o = pack8(v);
      *reinterpret_cast<uint4*>(z + (m + j * step) * C + c0) = o;
      if (mask) mask[(m + j * step) * cpr + s.chunk] = (uint8_t)mask8(o);
    }
  }
  for (; m < s.m1; m += step) {
    const long o = m * C + c0;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(y + o), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
    if (res) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(res + o), r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], lo);
    const uint4 ov = pack8(v);
    *reinterpret_cast<uint4*>(z + o) = ov;
    if (mask) mask[m * cpr + s.chunk] = (uint8_t)mask8(ov);
  }
}

// Stem entry in train(): BatchNorm (batch statistics) + ReLU + MaxPool2d(3, 2, 1) in one pass over the conv output:
// the normalised 112 x 112 map is never written.  One thread per (pooled pixel, 8 channels) reads the <= 9 window
// pixels (neighbouring threads share them through the caches), normalises, rounds to bf16 exactly as bn_apply does and
// keeps the first maximum in tap order (maxpool_kernel's rule); it also owns the 2 x 2 pixels (2ho..2ho+1, 2wo..2wo+1)
// of its window for the ReLU sign bits the backward pass reads.  Outputs: pooled map, argmax taps, sign bits.
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const bf16* __restrict__ y, const float* __restrict__ sums,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ pooled,
                                                            uint8_t* __restrict__ argmax, uint8_t* __restrict__ mask,
                                                            int B, int H, int W, int C, int Ho, int Wo, float eps,
                                                            float momentum, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var) {
  extern __shared__ float tab[];  // scale[C], shift[C]
  const long M = (long)B * H * W;
  const float inv_m = 1.0f / (float)M;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mean = sums[c] * inv_m;
    const float var = fmaxf(sums[C + c] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    tab[c] = sc;
    tab[C + c] = beta[c] - mean * sc;
    if (blockIdx.x == 0 && running_mean) {
      const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  __syncthreads();
  const int c8 = C >> 3;
  const long total = (long)B * Ho * Wo * c8;
  // grid-stride: a workgroup pays for its table (a dependent chain of loads before the first window) once, not once per
  // 32 pooled pixels
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
  const int cq = (int)(idx % c8);
  long r = idx / c8;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int b = (int)(r / Ho);
  float sc[8], sh[8], best[8];
  int arg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = tab[cq * 8 + k];
    sh[k] = tab[C + cq * 8 + k];
    best[k] = -INFINITY;
    arg[k] = 0;
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const long pix = ((long)b * H + hi) * W + wi;
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(y + pix * C + cq * 8), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
      const uint4 o = pack8(v);
      if (kh >= 1 && kw >= 1) mask[pix * c8 + cq] = (uint8_t)mask8(o);
      unpack8(o, v);  // the rounded values are what a max-pool over the stored map would compare
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (v[k] > best[k]) { best[k] = v[k]; arg[k] = kh * 3 + kw; }
    }
  }
  const long obase = (((long)b * Ho + ho) * Wo + wo) * C + cq * 8;
  *reinterpret_cast<uint4*>(pooled + obase) = pack8(best);
  uint2 a;
  a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
  a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
  *reinterpret_cast<uint2*>(argmax + obase) = a;
  }
}

// POOL variants of the two backward kernels (the stem's first BatchNorm, whose ReLU output feeds the 3x3 / stride-2
// max-pool): dz is not read from a map that rvit_maxpool3x3s2_bwd wrote but gathered here from the POOLED map's gradient
// and the recorded argmax taps - the 2 x 411 MB round trip of that map at B = 256 and the launch disappear.  The value
// is the one maxpool_bwd_kernel would have stored: the same fp32 sum over the (<= 4) windows in (kh, kw) order, rounded
// to bf16.  Branch-free: the candidates that do not exist read candidate A again with a tap that matches nothing.
struct PoolGeom {
  const uint8_t* argmax;
  int H, W, Ho, Wo;
  float r_hw, r_w;   // 1 / (H * W), 1 / W: row index -> (b, hi, wi) by float-reciprocal division + fix-up (M < 2^24)
};
__device__ __forceinline__ void pool_grad8(const bf16* __restrict__ dpool, const PoolGeom& g, long m, int C, int c0,
                                           float (&d)[8]) {
  const int hw = g.H * g.W, mi = (int)m;
  int b = (int)((float)mi * g.r_hw);
  int rem = mi - b * hw;
  if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
  int hi = (int)((float)rem * g.r_w);
  int wi = rem - hi * g.W;
  if (wi < 0) { --hi; wi += g.W; } else if (wi >= g.W) { ++hi; wi -= g.W; }
  // windows that contain row hi: ho = hi >> 1 with tap row 1 (even) or 2 (odd); odd rows also ho + 1 with tap row 0
  const int hoA = hi >> 1, woA = wi >> 1;
  const bool hB = (hi & 1) && hoA + 1 < g.Ho, wB = (wi & 1) && woA + 1 < g.Wo;
  const int ho_[2] = {hB ? hoA + 1 : hoA, hoA}, wo_[2] = {wB ? woA + 1 : woA, woA};
  const int kh_[2] = {hB ? 0 : 64, 1 + (hi & 1)}, kw_[2] = {wB ? 0 : 64, 1 + (wi & 1)};
  uint4 dv[4];
  uint2 am[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {   // (kh, kw) ascending: the order maxpool_bwd_kernel adds in
    const long o = ((((long)b * g.Ho + ho_[q >> 1]) * g.Wo + wo_[q & 1]) * C) + c0;
    dv[q] = *reinterpret_cast<const uint4*>(dpool + o);
    am[q] = *reinterpret_cast<const uint2*>(g.argmax + o);
  }
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tap = kh_[q >> 1] * 3 + kw_[q & 1];   // >= 64 for a candidate that does not exist: no byte matches
    float v[8];
    unpack8(dv[q], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = (int)(((k < 4 ? am[q].x : am[q].y) >> (8 * (k & 3))) & 0xff);
      if (a == tap) acc[k] += v[k];
    }
  }
  unpack8(pack8(acc), d);   // the bf16 rounding of the stored map
}

template <bool POOL>
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_reduce_kernel(const bf16* __restrict__ dz,
                                                                    const bf16* __restrict__ z,
                                                                    const bf16* __restrict__ y,
                                                                    const float* __restrict__ fsums, long M, int C,
                                                                    float eps, int relu, float* __restrict__ bsums,
                                                                    float* __restrict__ ws, const PoolGeom pg) {
  extern __shared__ float red[];  // [2*C]; ordered form: [slots][2*C]
  if (!ws) {
    for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) red[i] = 0.f;
    __syncthreads();
  }
  const RowSpan s = row_span(M, C, RED_THREADS);
  const long step = s.rows_per_pass;
  const int c0 = s.chunk * 8, cpr = C >> 3;
  const uint8_t* __restrict__ zmask = reinterpret_cast<const uint8_t*>(z);   // relu == 2: sign bits, not the map
  const float inv_m = 1.0f / (float)M;
  float mean[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mean[k] = fsums[c0 + k] * inv_m;
    s1[k] = 0.f; s2[k] = 0.f;
  }
  // s2 accumulates sum dzm*(y-mean); rstd is applied once at the end
  long m = s.m0 + s.r;
  for (; m + step < s.m1; m += 2 * step) {
    uint4 du[2], yu[2], zu[2];
    unsigned mu[2];
    float dq[2][8];   // POOL: the gathered gradient rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long o = (m + j * step) * C + c0;
      if constexpr (POOL) pool_grad8(dz, pg, m + j * step, C, c0, dq[j]);
      else du[j] = *reinterpret_cast<const uint4*>(dz + o);
      yu[j] = *reinterpret_cast<const uint4*>(y + o);
      if (relu == 1) zu[j] = *reinterpret_cast<const uint4*>(z + o);
      if (relu == 2) mu[j] = zmask[(m + j * step) * cpr + s.chunk];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float d[8], yv[8];
      if constexpr (POOL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = dq[j][k];
      } else {
        unpack8(du[j], d);
      }
      unpack8(yu[j], yv);
      if (relu == 1) {
        float zv[8];
        unpack8(zu[j], zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
      }
      if (relu == 2) apply_mask8(d, mu[j]);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1[k] += d[k]; s2[k] += d[k] * (yv[k] - mean[k]); }
    }
  }
  for (; m < s.m1; m += step) {
    const long o = m * C + c0;
    float d[8], yv[8];
    if constexpr (POOL) pool_grad8(dz, pg, m, C, c0, d);
    else unpack8(*reinterpret_cast<const uint4*>(dz + o), d);
    unpack8(*reinterpret_cast<const uint4*>(y + o), yv);
    if (relu == 1) {
      float zv[8];
      unpack8(*reinterpret_cast<const uint4*>(z + o), zv);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
    }
    if (relu == 2) apply_mask8(d, zmask[m * cpr + s.chunk]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] += d[k]; s2[k] += d[k] * (yv[k] - mean[k]); }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float var = fmaxf(fsums[C + c0 + k] * inv_m - mean[k] * mean[k], 0.f);
    s2[k] *= 1.0f / sqrtf(var + eps);
  }
  if (ws) {
    red_commit_det(red, C, s.chunk, s1, s2, ws, bsums);
    return;
  }
  red_commit(red, C, s.chunk, s1, s2);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) atomicAdd(bsums + i, red[i]);
}

template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ z,
                                                           const bf16* __restrict__ y, const float* __restrict__ fsums,
                                                           const float* __restrict__ bsums,
                                                           const float* __restrict__ gamma, bf16* __restrict__ dy,
                                                           bf16* __restrict__ dzm_out, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long M, int C, float eps,
                                                           int relu, const PoolGeom pg) {
  const RowSpan s = row_span(M, C);
  const long step = s.rows_per_pass;
  const int c0 = s.chunk * 8, cpr = C >> 3;
  const uint8_t* __restrict__ zmask = reinterpret_cast<const uint8_t*>(z);   // relu == 2: sign bits, not the map
  const float inv_m = 1.0f / (float)M;
  // dy = a*dzm + b*y + c  with a = gamma*rstd, b = -a*rstd*bs2/M, c = -a*bs1/M - b*mean
  float ca[8], cb[8], cc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const float mean = fsums[c] * inv_m;
    const float var = fmaxf(fsums[C + c] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float b1 = bsums[c], b2 = bsums[C + c];
    ca[k] = gamma[c] * rstd;
    cb[k] = -ca[k] * rstd * b2 * inv_m;
    cc[k] = -ca[k] * b1 * inv_m - cb[k] * mean;
    if (blockIdx.x == 0 && s.r == 0) {  // one thread per channel
      dgamma[c] += b2;
      dbeta[c] += b1;
    }
  }
  long m = s.m0 + s.r;
  for (; m + step < s.m1; m += 2 * step) {
    uint4 du[2], yu[2], zu[2];
    unsigned mu[2];
    float dq[2][8];   // POOL: the gathered gradient rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long o = (m + j * step) * C + c0;
      if constexpr (POOL) pool_grad8(dz, pg, m + j * step, C, c0, dq[j]);
      else du[j] = *reinterpret_cast<const uint4*>(dz + o);
      yu[j] = *reinterpret_cast<const uint4*>(y + o);
      if (relu == 1) zu[j] = *reinterpret_cast<const uint4*>(z + o);
      if (relu == 2) mu[j] = zmask[(m + j * step) * cpr + s.chunk];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long o = (m + j * step) * C + c0;
      float d[8], yv[8], out[8];
      if constexpr (POOL) {
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = dq[j][k];
      } else {
        unpack8(du[j], d);
      }
      unpack8(yu[j], yv);
      if (relu == 1) {
        float zv[8];
        unpack8(zu[j], zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
      }
      if (relu == 2) apply_mask8(d, mu[j]);
      if (dzm_out) *reinterpret_cast<uint4*>(dzm_out + o) = pack8(d);
#pragma unroll
      for (int k = 0; k < 8; ++k) out[k] = bn_dy(ca[k], d[k], cb[k], yv[k], cc[k]);
      *reinterpret_cast<uint4*>(dy + o) = pack8(out);
    }
  }
  for (; m < s.m1; m += step) {
    const long o = m * C + c0;
    float d[8], yv[8], out[8];
    if constexpr (POOL) pool_grad8(dz, pg, m, C, c0, d);
    else unpack8(*reinterpret_cast<const uint4*>(dz + o), d);
    unpack8(*reinterpret_cast<const uint4*>(y + o), yv);
    if (relu == 1) {
      float zv[8];
      unpack8(*reinterpret_cast<const uint4*>(z + o), zv);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = zv[k] > 0.f ? d[k] : 0.f;
    }
    if (relu == 2) apply_mask8(d, zmask[m * cpr + s.chunk]);
    if (dzm_out) *reinterpret_cast<uint4*>(dzm_out + o) = pack8(d);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = bn_dy(ca[k], d[k], cb[k], yv[k], cc[k]);
    *reinterpret_cast<uint4*>(dy + o) = pack8(out);
  }
}

// ---- the same two passes by 2 x 2 pixel blocks (H and W even: the stem's 112 x 112 map) -------------------------------------
// The four pixels (2i + di, 2j + dj) draw on the same four pooled positions (i + a, j + c): one thread fetches those four
// gradient chunks and tap records once and forms the four gathered gradients from them - 16 memory requests per four
// pixels where the per-pixel form issues 40 (it was bound by request rate: 238 / 285 us at B = 256 for 0.62 / 1.03 GB).
// Every pixel's value is the sum maxpool_bwd_kernel forms, in its (kh, kw) order, rounded to bf16: same bits per pixel.
struct PoolBlock { long pix[4]; };   // row index (pixel) of (2i, 2j), (2i, 2j+1), (2i+1, 2j), (2i+1, 2j+1)
__device__ __forceinline__ PoolBlock pool_block_grad(const bf16* __restrict__ dpool, const PoolGeom& g, long mb, int C, int c0,
                                                      float (&d)[4][8]) {
  const int hb = g.H >> 1, wb = g.W >> 1, bhw = hb * wb, mi = (int)mb;
  int b = (int)((float)mi * g.r_hw * 4.0f);
  int rem = mi - b * bhw;
  if (rem < 0) { --b; rem += bhw; } else if (rem >= bhw) { ++b; rem -= bhw; }
  int i = (int)((float)rem * g.r_w * 2.0f);
  int j = rem - i * wb;
  if (j < 0) { --i; j += wb; } else if (j >= wb) { ++i; j -= wb; }
  const bool hB = i + 1 < g.Ho, wB = j + 1 < g.Wo;
  const int i1 = hB ? i + 1 : i, j1 = wB ? j + 1 : j;
  const long prow0 = ((long)b * g.Ho + i) * g.Wo, prow1 = ((long)b * g.Ho + i1) * g.Wo;
  const long po[4] = {(prow0 + j) * C + c0, (prow0 + j1) * C + c0, (prow1 + j) * C + c0, (prow1 + j1) * C + c0};
  uint4 dv[4];
  uint2 am[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dv[q] = *reinterpret_cast<const uint4*>(dpool + po[q]);
    am[q] = *reinterpret_cast<const uint2*>(g.argmax + po[q]);
  }
  float v[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) unpack8(dv[q], v[q]);
  // (pooled position, tap) pairs of each pixel in (kh, kw) order; a pooled position that does not exist never matches
  const int t01 = wB ? 3 : 64, t10 = hB ? 1 : 64, t11 = (hB && wB) ? 0 : 64, t10b = hB ? 2 : 64, t01b = wB ? 6 : 64;
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = (int)(((k < 4 ? am[q].x : am[q].y) >> (8 * (k & 3))) & 0xff);
    float s;
    s = 0.f; if (a[0] == 4) s += v[0][k];
    acc[0][k] = s;
    s = 0.f; if (a[1] == t01) s += v[1][k]; if (a[0] == 5) s += v[0][k];
    acc[1][k] = s;
    s = 0.f; if (a[2] == t10) s += v[2][k]; if (a[0] == 7) s += v[0][k];
    acc[2][k] = s;
    s = 0.f; if (a[3] == t11) s += v[3][k]; if (a[2] == t10b) s += v[2][k]; if (a[1] == t01b) s += v[1][k]; if (a[0] == 8) s += v[0][k];
    acc[3][k] = s;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) unpack8(pack8(acc[q]), d[q]);   // the bf16 rounding of the stored map
  PoolBlock pb;
  const long r0 = ((long)b * g.H + 2 * i) * g.W + 2 * j;
  pb.pix[0] = r0; pb.pix[1] = r0 + 1; pb.pix[2] = r0 + g.W; pb.pix[3] = r0 + g.W + 1;
  return pb;
}

__global__ __launch_bounds__(RED_THREADS) void bn_bwd_reduce_pool2_kernel(const bf16* __restrict__ dpool,
                                                                          const uint8_t* __restrict__ zmask,
                                                                          const bf16* __restrict__ y,
                                                                          const float* __restrict__ fsums, long MB, int C,
                                                                          float eps, float* __restrict__ bsums,
                                                                          float* __restrict__ ws, const PoolGeom pg) {
  extern __shared__ float red[];
  if (!ws) {
    for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) red[i] = 0.f;
    __syncthreads();
  }
  const RowSpan s = row_span(MB, C, RED_THREADS);
  const int c0 = s.chunk * 8, cpr = C >> 3;
  const float inv_m = 0.25f / (float)MB;
  float mean[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mean[k] = fsums[c0 + k] * inv_m;
    s1[k] = 0.f; s2[k] = 0.f;
  }
  for (long m = s.m0 + s.r; m < s.m1; m += s.rows_per_pass) {
    float d[4][8];
    const PoolBlock pb = pool_block_grad(dpool, pg, m, C, c0, d);
    uint4 yu[4];
    unsigned mu[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yu[q] = *reinterpret_cast<const uint4*>(y + pb.pix[q] * C + c0);
      mu[q] = zmask[pb.pix[q] * cpr + s.chunk];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float yv[8];
      unpack8(yu[q], yv);
      apply_mask8(d[q], mu[q]);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1[k] += d[q][k]; s2[k] += d[q][k] * (yv[k] - mean[k]); }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float var = fmaxf(fsums[C + c0 + k] * inv_m - mean[k] * mean[k], 0.f);
    s2[k] *= 1.0f / sqrtf(var + eps);
  }
  if (ws) {
    red_commit_det(red, C, s.chunk, s1, s2, ws, bsums);
    return;
  }
  red_commit(red, C, s.chunk, s1, s2);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += RED_THREADS) atomicAdd(bsums + i, red[i]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_pool2_kernel(const bf16* __restrict__ dpool, const uint8_t* __restrict__ zmask,
                                                                 const bf16* __restrict__ y, const float* __restrict__ fsums,
                                                                 const float* __restrict__ bsums,
                                                                 const float* __restrict__ gamma, bf16* __restrict__ dy,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, long MB,
                                                                 int C, float eps, const PoolGeom pg) {
  const RowSpan s = row_span(MB, C);
  const int c0 = s.chunk * 8, cpr = C >> 3;
  const float inv_m = 0.25f / (float)MB;
  float ca[8], cb[8], cc[8];   // dy = a*dzm + b*y + c, as bn_bwd_apply_kernel
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const float mean = fsums[c] * inv_m;
    const float var = fmaxf(fsums[C + c] * inv_m - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float b1 = bsums[c], b2 = bsums[C + c];
    ca[k] = gamma[c] * rstd;
    cb[k] = -ca[k] * rstd * b2 * inv_m;
    cc[k] = -ca[k] * b1 * inv_m - cb[k] * mean;
    if (blockIdx.x == 0 && s.r == 0) {  // one thread per channel
      dgamma[c] += b2;
      dbeta[c] += b1;
    }
  }
  for (long m = s.m0 + s.r; m < s.m1; m += s.rows_per_pass) {
    float d[4][8];
    const PoolBlock pb = pool_block_grad(dpool, pg, m, C, c0, d);
    uint4 yu[4];
    unsigned mu[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yu[q] = *reinterpret_cast<const uint4*>(y + pb.pix[q] * C + c0);
      mu[q] = zmask[pb.pix[q] * cpr + s.chunk];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float yv[8], out[8];
      unpack8(yu[q], yv);
      apply_mask8(d[q], mu[q]);
#pragma unroll
      for (int k = 0; k < 8; ++k) out[k] = bn_dy(ca[k], d[q][k], cb[k], yv[k], cc[k]);
      *reinterpret_cast<uint4*>(dy + pb.pix[q] * C + c0) = pack8(out);
    }
  }
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

bool bn_shape_ok(long M, int C) { return M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0 && (256 % (C >> 3)) == 0; }  // C/8 a power of two <= 256

// workgroups for M rows when a thread should see about `rows_per_thread` rows; 8 workgroups per CU at most
static long bn_env_cap(const char* name) {
  const char* e = getenv(name);
  const long x = e ? atol(e) : 0;
  return x > 0 ? x : 0;
}
static int bn_rpt() { return 8; }
// Workgroup caps of the two elementwise passes.  On maps that fit the caches a few times over, ONE workgroup of 256
// threads per CU streaming a long contiguous run of rows beats 6-8 workgroups per CU on short runs (scripts/bench_bn.py,
// B = 256, us at 2048 / 512 / 256 workgroups: apply layer1 44.5 / 39.6 / 38.7, layer2 39.8 / 26.8 / 24.9, layer3 24.8 / 20.6 /
// 17.0, layer4 15.3 / 15.4 / 12.8; backward apply layer1 62.9 / 64.5 / 92.2, layer2 44.9 / 35.7 / 38.9, layer3 29.8 / 24.5 /
// 24.7, layer4 20.3 / 20.2 / 18.2): every thread pays a dependent chain for its per-channel constants before its first
// row, and fewer, longer streams keep DRAM pages open.  The 411 MB map of the stem entry wants all the workgroups it can
// get.  RVIT_BN_WG_APPLY / RVIT_BN_WG_BWD override the caps (A/B runs; the results do not depend on the grid).
static int bn_num_cu() {
  static const int v = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return v;
}
static long bn_apply_cap(long M, int C) {
  static const long env = bn_env_cap("RVIT_BN_WG_APPLY");
  if (env) return env;
  return M * C * 2 <= (128L << 20) ? bn_num_cu() : 2048;
}
static long bn_bwd_apply_cap(long M, int C) {
  static const long env = bn_env_cap("RVIT_BN_WG_BWD");
  if (env) return env;
  const long bytes = M * C * 2;
  return bytes <= (16L << 20) ? bn_num_cu() : bytes <= (64L << 20) ? 2L * bn_num_cu() : 2048;
}
long bn_blocks(long M, int C, int rows_per_thread) {
  const long per = (long)(256 / (C >> 3)) * rows_per_thread;
  long blocks = (M + per - 1) / per;
  return blocks > 2048 ? 2048 : blocks;
}

long red_blocks(long M, int C, bool ordered) {
  const long per = (long)(RED_THREADS / (C >> 3)) * 4;
  long blocks = (M + per - 1) / per;
  // ordered form: the last workgroup reads blocks * 2C partials on its own, so wide maps run fewer workgroups
  long cap = !ordered ? 256 : (C <= 64 ? 256 : (C <= 256 ? 128 : 64));   // (C = 128 at B = 256: 30.7 us at 128 workgroups, 33.2 at 256)
  {
    static const long env = bn_env_cap("RVIT_BN_WG_RED");
    if (env && (!ordered || env * 2 * C <= 262144)) cap = env;
  }
  return blocks > cap ? cap : blocks;
}
size_t red_lds(int C, bool ordered) {
  const int slots = (C >> 3) < 64 ? RED_THREADS / 64 : RED_THREADS / (C >> 3);
  const size_t fl = (size_t)(ordered ? slots : 1) * 2 * C;
  return (ordered && fl < RED_THREADS ? (size_t)RED_THREADS : fl) * sizeof(float);
}

}  // namespace

int rvit_bn_stats_impl(const void* y, long M, int C, float* sums, float* ws, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && bn_shape_ok(M, C) && rvit_aligned16(y), "bn_stats: bad arguments (M=%ld C=%d)", M, C);
  RVIT_CHECK_ARG(!ws || (((uintptr_t)ws) & 3) == 0, "bn_stats: workspace alignment");
  const long blocks = red_blocks(M, C, ws != nullptr);
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(RED_THREADS), red_lds(C, ws != nullptr), st,
                     (const bf16*)y, M, C, sums, ws);
  RVIT_CHECK_LAUNCH("bn_stats");
  return RVIT_OK;
}

int rvit_bn_apply_impl(const void* y, const float* sums, const float* gamma, const float* beta, const void* res,
                       void* z, long M, int C, float eps, int relu, float momentum, float* running_mean,
                       float* running_var, void* relu_mask, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && gamma && beta && z && bn_shape_ok(M, C), "bn_apply: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(y) && rvit_aligned16(z) && (!res || rvit_aligned16(res)), "bn_apply: alignment");
  RVIT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_apply: running stats come in pairs");
  RVIT_CHECK_ARG(!relu_mask || relu, "bn_apply: a ReLU mask needs relu");
  long blocks = bn_blocks(M, C, bn_rpt());
  if (const long cap = bn_apply_cap(M, C); blocks > cap) blocks = cap;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)y, sums,
                     gamma, beta, (const bf16*)res, (bf16*)z, M, C, eps, relu, momentum, running_mean, running_var,
                     (uint8_t*)relu_mask);
  RVIT_CHECK_LAUNCH("bn_apply");
  return RVIT_OK;
}

int rvit_bn_apply_pool_impl(const void* y, const float* sums, const float* gamma, const float* beta, void* pooled,
                            void* argmax, void* relu_mask, int B, int H, int W, int C, float eps, float momentum,
                            float* running_mean, float* running_var, hipStream_t st) {
  RVIT_CHECK_ARG(y && sums && gamma && beta && pooled && argmax && relu_mask && B > 0 && H > 0 && W > 0 &&
                     bn_shape_ok((long)B * H * W, C),
                 "bn_apply_pool: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(y) && rvit_aligned16(pooled) && (((uintptr_t)argmax) & 7) == 0, "bn_apply_pool: alignment");
  RVIT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_apply_pool: running stats come in pairs");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C >> 3);
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "bn_apply_pool: too many elements");
  long pool_blocks = (total + 255) / 256;
  {
    static const long env = bn_env_cap("RVIT_BN_WG_POOL");
    const long cap = env ? env : 16L * bn_num_cu();   // B = 256: 218 us at 25,088 workgroups, 210 at 4,096-8,192, 236 at 2,048
    if (pool_blocks > cap) pool_blocks = cap;
  }
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3((unsigned)pool_blocks), dim3(256), 2 * C * sizeof(float), st,
                     (const bf16*)y, sums, gamma, beta, (bf16*)pooled, (uint8_t*)argmax, (uint8_t*)relu_mask, B, H, W, C,
                     Ho, Wo, eps, momentum, running_mean, running_var);
  RVIT_CHECK_LAUNCH("bn_apply_pool");
  return RVIT_OK;
}

int rvit_bn_bwd_reduce_impl(const void* dz, const void* z, const void* y, const float* fsums, long M, int C, float eps,
                            int relu, float* bsums, float* ws, hipStream_t st) {
  RVIT_CHECK_ARG(dz && y && fsums && bsums && (!relu || z) && bn_shape_ok(M, C), "bn_bwd_reduce: bad arguments");
  RVIT_CHECK_ARG(relu >= 0 && relu <= 2, "bn_bwd_reduce: relu is 0, 1 (z = forward output) or 2 (z = sign bits)");
  RVIT_CHECK_ARG(rvit_aligned16(dz) && rvit_aligned16(y) && (relu != 1 || rvit_aligned16(z)), "bn_bwd_reduce: alignment");
  const long blocks = red_blocks(M, C, ws != nullptr);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3((unsigned)blocks), dim3(RED_THREADS), red_lds(C, ws != nullptr), st,
                     (const bf16*)dz, (const bf16*)z, (const bf16*)y, fsums, M, C, eps, relu, bsums, ws, PoolGeom{});
  RVIT_CHECK_LAUNCH("bn_bwd_reduce");
  return RVIT_OK;
}

int rvit_bn_bwd_apply_impl(const void* dz, const void* z, const void* y, const float* fsums, const float* bsums,
                           const float* gamma, void* dy, void* dzm_out, float* dgamma, float* dbeta, long M, int C,
                           float eps, int relu, hipStream_t st) {
  RVIT_CHECK_ARG(dz && y && fsums && bsums && gamma && dy && dgamma && dbeta && (!relu || z) && bn_shape_ok(M, C),
                 "bn_bwd_apply: bad arguments");
  RVIT_CHECK_ARG(relu >= 0 && relu <= 2, "bn_bwd_apply: relu is 0, 1 (z = forward output) or 2 (z = sign bits)");
  RVIT_CHECK_ARG(rvit_aligned16(dz) && rvit_aligned16(y) && rvit_aligned16(dy) && (relu != 1 || rvit_aligned16(z)) &&
                     (!dzm_out || rvit_aligned16(dzm_out)),
                 "bn_bwd_apply: alignment");
  long blocks = bn_blocks(M, C, bn_rpt());
  if (const long cap = bn_bwd_apply_cap(M, C); blocks > cap) blocks = cap;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)dz,
                     (const bf16*)z, (const bf16*)y, fsums, bsums, gamma, (bf16*)dy, (bf16*)dzm_out, dgamma, dbeta, M,
                     C, eps, relu, PoolGeom{});
  RVIT_CHECK_LAUNCH("bn_bwd_apply");
  return RVIT_OK;
}

// BatchNorm backward behind a 3x3 / stride-2 max-pool (the stem's bn1 -> relu -> maxpool, train_vit.py:145 through timm's
// resnet): rvit_maxpool3x3s2_bwd_nhwc + rvit_bn_bwd_reduce / rvit_bn_bwd_apply (relu = 2) without the un-pooled gradient map
// RVIT_POOL_BWD_PER_PIXEL=1: the per-pixel kernels also on even maps (A/B runs; same bits per pixel of dy given equal sums)
static bool pool_per_pixel() {
  static const bool v = [] { const char* e = getenv("RVIT_POOL_BWD_PER_PIXEL"); return e && e[0] == '1'; }();
  return v;
}
static int pool_geom(const void* argmax, int B, int H, int W, int C, PoolGeom* g, const char* what) {
  RVIT_CHECK_ARG(B > 0 && H > 1 && W > 1 && (long)B * H * W < (1L << 24), "%s: B*H*W must be below 2^24", what);
  RVIT_CHECK_ARG(argmax && (((uintptr_t)argmax) & 7) == 0, "%s: argmax alignment", what);
  g->argmax = (const uint8_t*)argmax;
  g->H = H; g->W = W; g->Ho = (H - 1) / 2 + 1; g->Wo = (W - 1) / 2 + 1;
  g->r_hw = 1.0f / (float)(H * W); g->r_w = 1.0f / (float)W;
  (void)C;
  return RVIT_OK;
}

int rvit_bn_bwd_reduce_pool_impl(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                                 const float* fsums, int B, int H, int W, int C, float eps, float* bsums, float* ws,
                                 hipStream_t st) {
  const long M = (long)B * H * W;
  RVIT_CHECK_ARG(dpool && relu_mask && y && fsums && bsums && bn_shape_ok(M, C), "bn_bwd_reduce_pool: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(dpool) && rvit_aligned16(y), "bn_bwd_reduce_pool: alignment");
  PoolGeom g;
  if (const int rc = pool_geom(argmax, B, H, W, C, &g, "bn_bwd_reduce_pool")) return rc;
  if (!(H & 1) && !(W & 1) && !pool_per_pixel()) {   // by 2 x 2 pixel blocks
    const long MB = M / 4, blocks = red_blocks(MB, C, ws != nullptr);
    hipLaunchKernelGGL(bn_bwd_reduce_pool2_kernel, dim3((unsigned)blocks), dim3(RED_THREADS), red_lds(C, ws != nullptr), st,
                       (const bf16*)dpool, (const uint8_t*)relu_mask, (const bf16*)y, fsums, MB, C, eps, bsums, ws, g);
    RVIT_CHECK_LAUNCH("bn_bwd_reduce_pool");
    return RVIT_OK;
  }
  const long blocks = red_blocks(M, C, ws != nullptr);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3((unsigned)blocks), dim3(RED_THREADS), red_lds(C, ws != nullptr), st,
                     (const bf16*)dpool, (const bf16*)relu_mask, (const bf16*)y, fsums, M, C, eps, 2, bsums, ws, g);
  RVIT_CHECK_LAUNCH("bn_bwd_reduce_pool");
  return RVIT_OK;
}

int rvit_bn_bwd_apply_pool_impl(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                                const float* fsums, const float* bsums, const float* gamma, void* dy, float* dgamma,
                                float* dbeta, int B, int H, int W, int C, float eps, hipStream_t st) {
  const long M = (long)B * H * W;
  RVIT_CHECK_ARG(dpool && relu_mask && y && fsums && bsums && gamma && dy && dgamma && dbeta && bn_shape_ok(M, C),
                 "bn_bwd_apply_pool: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(dpool) && rvit_aligned16(y) && rvit_aligned16(dy), "bn_bwd_apply_pool: alignment");
  PoolGeom g;
  if (const int rc = pool_geom(argmax, B, H, W, C, &g, "bn_bwd_apply_pool")) return rc;
  if (!(H & 1) && !(W & 1) && !pool_per_pixel()) {
    const long MB = M / 4;
    long blocks = bn_blocks(MB, C, 4);
    {
      static const long env = bn_env_cap("RVIT_BN_WG_POOLBWD");
      if (env && blocks > env) blocks = env;
    }
    hipLaunchKernelGGL(bn_bwd_apply_pool2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)dpool,
                       (const uint8_t*)relu_mask, (const bf16*)y, fsums, bsums, gamma, (bf16*)dy, dgamma, dbeta, MB, C, eps, g);
    RVIT_CHECK_LAUNCH("bn_bwd_apply_pool");
    return RVIT_OK;
  }
  const long blocks = bn_blocks(M, C, bn_rpt());
  hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)dpool,
                     (const bf16*)relu_mask, (const bf16*)y, fsums, bsums, gamma, (bf16*)dy, (bf16*)nullptr, dgamma, dbeta,
                     M, C, eps, 2, g);
  RVIT_CHECK_LAUNCH("bn_bwd_apply_pool");
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
