// HBM-bound backward / loss kernels of the training path (autograd of train_vit.py:141-171 and
// the loss of train_vit.py:894,937; run by `scaler.scale(loss).backward()`, train_vit.py:939).
#include "common.h"
#include "rvit_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm backward, fused with the residual-gradient accumulation of the pre-LN block:
//   dxs[row] (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// and an optional bf16 copy of the updated residual gradient (A operand of the next dgrad/wgrad).
// dgamma / dbeta are reduced per wave in registers over a grid-stride row loop and added with
// one atomic per column per wave.
constexpr int LNB_MAX_IT = 8;

template <int IT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dxs,
                                                            long lddx, int accumulate, bf16* __restrict__ dx_bf16,
                                                            long lddb, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, long rows, int E) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float gam[IT][4], dg[IT][4], db[IT][4];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    float4 g4 = make_float4(0, 0, 0, 0);
    if (e < E) g4 = *reinterpret_cast<const float4*>(gamma + e);
    gam[it][0] = g4.x; gam[it][1] = g4.y; gam[it][2] = g4.z; gam[it][3] = g4.w;
#pragma unroll
    for (int k = 0; k < 4; ++k) { dg[it][k] = 0.f; db[it][k] = 0.f; }
  }
  const float inv_e = 1.0f / (float)E;
  for (long row = wave_id; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    float xh[IT][4], g[IT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      if (e < E) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row * ldx + e);
        const uint2 dv = *reinterpret_cast<const uint2*>(dy + row * lddy + e);
        const float d[4] = {bf16_lo(dv.x), bf16_hi(dv.x), bf16_lo(dv.y), bf16_hi(dv.y)};
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[it][k] = (xs[k] - mu) * rs;
          g[it][k] = d[k] * gam[it][k];
          s1 += g[it][k];
          s2 += g[it][k] * xh[it][k];
          dg[it][k] += d[k] * xh[it][k];
          db[it][k] += d[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xh[it][k] = 0.f; g[it][k] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1) * inv_e, c2 = wave_sum(s2) * inv_e;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      if (e < E) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = rs * (g[it][k] - c1 - xh[it][k] * c2);
        float* dp = dxs + row * lddx + e;
        if (accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(dp);
          o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        }
        *reinterpret_cast<float4*>(dp) = make_float4(o[0], o[1], o[2], o[3]);
        if (dx_bf16) {
          uint2 w;
          w.x = pack_bf16x2(o[0], o[1]);
          w.y = pack_bf16x2(o[2], o[3]);
          *reinterpret_cast<uint2*>(dx_bf16 + row * lddb + e) = w;
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
    if (e < E) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(dgamma + e + k, dg[it][k]);
        atomicAdd(dbeta + e + k, db[it][k]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cross-entropy with label smoothing, mean over the batch (nn.CrossEntropyLoss(label_smoothing),
// train_vit.py:894,937), forward + gradient in one pass.  One thread per sample (C <= 64).
//   loss_i = (1-eps) * (-log p_y) + eps/C * sum_c (-log p_c);  dlogits = (p - target_dist) * gscale
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, long ldl,
                                                      const long* __restrict__ labels, float* __restrict__ loss_sum,
                                                      float* __restrict__ dlogits, long lddl,
                                                      bf16* __restrict__ dlogits_bf16, long lddb, int B, int C,
                                                      int C_pad, float eps, float gscale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float li = 0.f;
  if (i < B) {
    const float* z = logits + (long)i * ldl;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f, sz = 0.f;
    for (int c = 0; c < C; ++c) { se += expf(z[c] - mx); sz += z[c]; }
    const float lse = mx + logf(se);
    const int y = (int)labels[i];
    li = (1.f - eps) * (lse - z[y]) + eps * (lse - sz / (float)C);
    const float un = eps / (float)C;
    for (int c = 0; c < C_pad; ++c) {
      float d = 0.f;
      if (c < C) d = (expf(z[c] - lse) - un - (c == y ? 1.f - eps : 0.f)) * gscale;
      if (dlogits && c < C) dlogits[(long)i * lddl + c] = d;
      if (dlogits_bf16) dlogits_bf16[(long)i * lddb + c] = (bf16)d;
    }
  }
  li = wave_sum(li);
  if ((threadIdx.x & 63) == 0 && loss_sum) atomicAdd(loss_sum, li / (float)B);
}

// ---------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix into fp32 (bias gradients): out[n] += sum_m A[m,n].
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ a, long lda, float* __restrict__ out,
                                                     long M, int N, long row_group, long row_extra) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const long per = (M + gridDim.y - 1) / gridDim.y;
  const long m0 = (long)blockIdx.y * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;
  float s = 0.f;
  if (n < N)
    for (long m = m0 + ty; m < m1; m += 4) {
      long r = m;
      if (row_group > 0) {
        const long g = m / row_group;
        r = g * (row_group + row_extra) + (m - g * row_group) + row_extra;
      }
      s += (float)a[r * lda + n];
    }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(out + n, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// out[j] += sum_b x[b, j]  (pos_embed / cls_token gradients: sum of the token gradients over the batch)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, long ldb, float* __restrict__ out,
                                                        int B, long n) {
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= n) return;
  float4 s = make_float4(0, 0, 0, 0);
  for (int b = 0; b < B; ++b) {
    const float4 v = *reinterpret_cast<const float4*>(x + (long)b * ldb + j);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4 o = *reinterpret_cast<float4*>(out + j);
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *reinterpret_cast<float4*>(out + j) = o;
}

}  // namespace

int rvit_layernorm_bwd_impl(const void* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                            const float* gamma, float* dxs, long lddx, int accumulate, void* dx_bf16, long lddb,
                            float* dgamma, float* dbeta, long rows, int E, hipStream_t st) {
  RVIT_CHECK_ARG(dy && x && mean && rstd && gamma && dxs && dgamma && dbeta, "layernorm_bwd: null pointer");
  RVIT_CHECK_ARG(E > 0 && (E & 3) == 0 && E <= LNB_MAX_IT * 256, "layernorm_bwd: E=%d unsupported", E);
  RVIT_CHECK_ARG((lddy & 3) == 0 && (ldx & 3) == 0 && (lddx & 3) == 0 && (!dx_bf16 || (lddb & 3) == 0),
                 "layernorm_bwd: pitches must be multiples of 4");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(dxs) && rvit_aligned16(gamma) && ((uintptr_t)dy & 7) == 0 &&
                     ((uintptr_t)dx_bf16 & 7) == 0,
                 "layernorm_bwd: alignment");
  if (rows <= 0) return RVIT_OK;
  long blocks = (rows + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  const int it = (E + 255) / 256;
#define LNB_CASE(N)                                                                                             \
  case N:                                                                                                       \
    hipLaunchKernelGGL((layernorm_bwd_kernel<N>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)dy,    \
                       lddy, x, ldx, mean, rstd, gamma, dxs, lddx, accumulate, (bf16*)dx_bf16, lddb, dgamma,    \
                       dbeta, rows, E);                                                                         \
    break;
  switch (it) {
    LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5) LNB_CASE(6) LNB_CASE(7) LNB_CASE(8)
    default: rvit_set_error("layernorm_bwd: E=%d unsupported", E); return RVIT_EUNSUPPORTED;
  }
#undef LNB_CASE
  RVIT_CHECK_LAUNCH("layernorm_bwd");
  return RVIT_OK;
}

int rvit_ce_loss_impl(const float* logits, long ldl, const long* labels, float* loss_sum, float* dlogits, long lddl,
                      void* dlogits_bf16, long lddb, int B, int C, int C_pad, float eps, float gscale,
                      hipStream_t st) {
  RVIT_CHECK_ARG(logits && labels && B > 0 && C > 0 && C <= 4096 && C_pad >= C, "ce_loss: bad arguments");
  RVIT_CHECK_ARG(!dlogits_bf16 || lddb >= C_pad, "ce_loss: bf16 gradient pitch smaller than C_pad");
  hipLaunchKernelGGL(ce_loss_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, logits, ldl, labels,
                     loss_sum, dlogits, lddl, (bf16*)dlogits_bf16, lddb, B, C, C_pad, eps, gscale);
  RVIT_CHECK_LAUNCH("ce_loss");
  return RVIT_OK;
}

int rvit_colsum_impl(const void* a, long lda, float* out, long M, int N, long row_group, long row_extra,
                     hipStream_t st) {
  RVIT_CHECK_ARG(a && out && M >= 0 && N > 0 && lda >= N, "colsum: bad arguments");
  if (M == 0) return RVIT_OK;
  long ysplit = (M + 255) / 256;
  if (ysplit > 256) ysplit = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)ysplit), dim3(256), 0, st,
                     (const bf16*)a, lda, out, M, N, row_group, row_extra);
  RVIT_CHECK_LAUNCH("colsum");
  return RVIT_OK;
}

int rvit_batch_sum_impl(const float* x, long ldb, float* out, int B, long n, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && B > 0 && n > 0 && (n & 3) == 0 && (ldb & 3) == 0, "batch_sum: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(out), "batch_sum: alignment");
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, x, ldb, out, B, n);
  RVIT_CHECK_LAUNCH("batch_sum");
  return RVIT_OK;
}
