// HBM-bound backward / loss kernels of the training path (autograd of train_vit.py:141-171 and
// the loss of train_vit.py:894,937; run by `scaler.scale(loss).backward()`, train_vit.py:939).
#include "common.h"
#include "rvit_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm backward, fused with the residual-gradient accumulation of the pre-LN block:
//   dxs[row] (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// and an optional bf16 copy of the updated residual gradient (A operand of the next dgrad/wgrad), optionally with the
// dropout of the site that consumes it applied (the backward of dropout1 / dropout2, torch transformer.py:977,982: the
// gradient that reaches out_proj / linear2 is the residual gradient times that site's mask; element index row * E +
// col as rvit_dropout_apply, which this replaces: one pass of 155 MB read + 77 MB written per ViT-B site saved).
// dgamma / dbeta are reduced per wave in registers over a grid-stride row loop, then across the
// waves of the workgroup in LDS, and added with one global atomic per column per workgroup (at most
// 256 workgroups: same-address atomics serialise in L2, 2M of them cost more than the whole pass).
constexpr int LNB_MAX_IT = 8;

template <int IT, int THREADS>
__global__ __launch_bounds__(THREADS) void layernorm_bwd_kernel(const bf16* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dxs,
                                                            long lddx, int accumulate, bf16* __restrict__ dx_bf16,
                                                            long lddb, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, long rows, int E,
                                                            float* __restrict__ ws, DropKey dk) {
  extern __shared__ float lnb_red[];  // [2 * IT * 256]; ordered mode (ws): one such vector per wave
  constexpr int WAVES = THREADS / 64;
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * WAVES;
  if (!ws) {
    for (int i = threadIdx.x; i < 2 * IT * 256; i += THREADS) lnb_red[i] = 0.f;
    __syncthreads();
  }
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
          if (dk.thresh) {   // e is a multiple of 4 and E is even: two whole hash pairs
            float m[4];
            drop_keep4(dk, (uint32_t)((row * E + e) >> 1), dk.scale, m);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rvit_mul_rn((float)(bf16)o[k], m[k]);   // rounded first, as rvit_dropout_apply on the bf16 copy
          }
          uint2 w;
          w.x = pack_bf16x2(o[0], o[1]);
          w.y = pack_bf16x2(o[2], o[3]);
          *reinterpret_cast<uint2*>(dx_bf16 + row * lddb + e) = w;
        }
      }
    }
  }
  if (ws) {
    // ordered mode: every wave stores its sums to its own LDS vector, thread i adds the vectors in wave order and
    // stores the workgroup's partial to slot blockIdx.x of the workspace; rvit_ordered_finalize adds the slots in order
    constexpr int V = 2 * IT * 256;
    float* mine = lnb_red + (threadIdx.x >> 6) * V;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      *reinterpret_cast<float4*>(mine + e) = make_float4(dg[it][0], dg[it][1], dg[it][2], dg[it][3]);
      *reinterpret_cast<float4*>(mine + IT * 256 + e) = make_float4(db[it][0], db[it][1], db[it][2], db[it][3]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += THREADS) {
      float t = 0.f;
      for (int w = 0; w < WAVES; ++w) t += lnb_red[w * V + i];
      ws[(long)blockIdx.x * V + i] = t;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * 256 + lane * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(&lnb_red[e + k], dg[it][k]);
      atomicAdd(&lnb_red[IT * 256 + e + k], db[it][k]);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += THREADS) {
    atomicAdd(dgamma + e, lnb_red[e]);
    atomicAdd(dbeta + e, lnb_red[IT * 256 + e]);
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
  // ONE workgroup loops over the batch, so the loss is summed in a fixed order (lanes by butterfly, waves in order)
  __shared__ float ce_red[4];
  float li = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float* z = logits + (long)i * ldl;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f, sz = 0.f;
    for (int c = 0; c < C; ++c) { se += expf(z[c] - mx); sz += z[c]; }
    const float lse = mx + logf(se);
    const int y = (int)labels[i];
    li += (1.f - eps) * (lse - z[y]) + eps * (lse - sz / (float)C);
    const float un = eps / (float)C;
    for (int c = 0; c < C_pad; ++c) {
      float d = 0.f;
      if (c < C) d = (expf(z[c] - lse) - un - (c == y ? 1.f - eps : 0.f)) * gscale;
      if (dlogits && c < C) dlogits[(long)i * lddl + c] = d;
      if (dlogits_bf16) dlogits_bf16[(long)i * lddb + c] = (bf16)d;
    }
  }
  li = wave_sum(li);
  if ((threadIdx.x & 63) == 0) ce_red[threadIdx.x >> 6] = li;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) *loss_sum += (((ce_red[0] + ce_red[1]) + ce_red[2]) + ce_red[3]) / (float)B;
}

// ---------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix into fp32 (bias gradients): out[n] += sum_m A[m,n].
// A workgroup covers 256 columns (32 lanes x 16 B) x 8 row lanes, four rows in flight per thread;
// partial sums meet in LDS and leave with one atomic per column per workgroup.
__device__ __forceinline__ long colsum_row(long m, long row_group, long row_extra) {
  if (row_group <= 0) return m;
  const long g = m / row_group;
  return g * (row_group + row_extra) + (m - g * row_group) + row_extra;
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ a, long lda, float* __restrict__ out,
                                                     long M, int N, long row_group, long row_extra,
                                                     float* __restrict__ ws) {
  __shared__ float red[256];
  __shared__ float slots[8][256];   // ordered mode: one vector per row lane
  red[threadIdx.x] = 0.f;
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.x * 256 + tx * 8;
  const long per = (M + gridDim.y - 1) / gridDim.y;
  const long m0 = (long)blockIdx.y * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (n0 < N) {
    if (n0 + 8 <= N && (lda & 7) == 0) {
      long m = m0 + ty;
      for (; m + 24 < m1; m += 32) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = *reinterpret_cast<const uint4*>(a + colsum_row(m + 8 * u, row_group, row_extra) * lda + n0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s[0] += bf16_lo(v[u].x); s[1] += bf16_hi(v[u].x); s[2] += bf16_lo(v[u].y); s[3] += bf16_hi(v[u].y);
          s[4] += bf16_lo(v[u].z); s[5] += bf16_hi(v[u].z); s[6] += bf16_lo(v[u].w); s[7] += bf16_hi(v[u].w);
        }
      }
      for (; m < m1; m += 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(a + colsum_row(m, row_group, row_extra) * lda + n0);
        s[0] += bf16_lo(v.x); s[1] += bf16_hi(v.x); s[2] += bf16_lo(v.y); s[3] += bf16_hi(v.y);
        s[4] += bf16_lo(v.z); s[5] += bf16_hi(v.z); s[6] += bf16_lo(v.w); s[7] += bf16_hi(v.w);
      }
    } else {  // ragged tail columns or unaligned pitch
      for (long m = m0 + ty; m < m1; m += 8) {
        const bf16* row = a + colsum_row(m, row_group, row_extra) * lda;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (n0 + k < N) s[k] += (float)row[n0 + k];
      }
    }
    if (!ws) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&red[tx * 8 + k], s[k]);
    }
  }
  if (ws) {   // row lanes added in order, the workgroup's partial stored to slot blockIdx.y; finalised in slot order
#pragma unroll
    for (int k = 0; k < 8; ++k) slots[ty][tx * 8 + k] = s[k];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += slots[r][threadIdx.x];
    ws[((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = t;
    return;
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) atomicAdd(out + n, red[threadIdx.x]);
}

// out[j] += sum_b x[b, j]  (pos_embed / cls_token gradients: sum of the token gradients over the batch).  A workgroup
// owns 16 float4 columns; thread group g of 16 adds the rows g, g + 16, ... (eight loads in flight), the groups are
// combined through LDS in group order: one association on every run.  (One thread per column walking the whole batch
// was 62 us at B = 256 for 13 MB: a chain of 256 dependent round trips on 13 workgroups.)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, long ldb, float* __restrict__ out,
                                                        int B, long n) {
  __shared__ float4 red[16][16];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long j = ((long)blockIdx.x * 16 + c) * 4;
  float4 s = make_float4(0, 0, 0, 0);
  if (j < n) {
    for (int b0 = g; b0 < B; b0 += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 16 * u;
        v[u] = b < B ? *reinterpret_cast<const float4*>(x + (long)b * ldb + j) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && j < n) {
    float4 t = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) { t.x += red[k][c].x; t.y += red[k][c].y; t.z += red[k][c].z; t.w += red[k][c].w; }
    float4 o = *reinterpret_cast<float4*>(out + j);
    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    *reinterpret_cast<float4*>(out + j) = o;
  }
}

}  // namespace

int rvit_layernorm_bwd_impl(const void* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                            const float* gamma, float* dxs, long lddx, int accumulate, void* dx_bf16, long lddb,
                            float* dgamma, float* dbeta, long rows, int E, float* ws, size_t ws_bytes,
                            float drop_p, uint64_t drop_seed, hipStream_t st) {
  RVIT_CHECK_ARG(dy && x && mean && rstd && gamma && dxs && dgamma && dbeta, "layernorm_bwd: null pointer");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd: drop_p must be in [0, 1)");
  RVIT_CHECK_ARG(drop_p == 0.f || (dx_bf16 && rows * E < 0xffffffffL),
                 "layernorm_bwd: the dropout applies to the bf16 copy (required) and indexes its elements in 32 bits");
  const DropKey dk = make_drop_key(drop_p, drop_seed);
  RVIT_CHECK_ARG(E > 0 && (E & 3) == 0 && E <= LNB_MAX_IT * 256, "layernorm_bwd: E=%d unsupported", E);
  RVIT_CHECK_ARG((lddy & 3) == 0 && (ldx & 3) == 0 && (lddx & 3) == 0 && (!dx_bf16 || (lddb & 3) == 0),
                 "layernorm_bwd: pitches must be multiples of 4");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(dxs) && rvit_aligned16(gamma) && ((uintptr_t)dy & 7) == 0 &&
                     ((uintptr_t)dx_bf16 & 7) == 0,
                 "layernorm_bwd: alignment");
  if (rows <= 0) return RVIT_OK;
  const int it = (E + 255) / 256;
  // 16 waves per workgroup while the per-lane state fits 128 VGPRs, 4 beyond; one workgroup per CU at most
  const int threads = it <= 3 ? 1024 : 256;
  long blocks = (rows + threads / 64 - 1) / (threads / 64);
  if (blocks > 256) blocks = 256;
  const int V = 2 * it * 256;   // floats of one partial vector: dgamma at [0, E), dbeta at [it * 256, it * 256 + E)
  size_t lds = (size_t)V * sizeof(float);
  if (ws) {
    RVIT_CHECK_ARG(rvit_aligned16(ws) && ws_bytes >= (size_t)blocks * V * sizeof(float),
                   "layernorm_bwd: the ordered-reduction workspace needs %zu bytes", (size_t)blocks * V * sizeof(float));
    lds *= threads / 64;
  }
#define LNB_CASE(N, T)                                                                                          \
  case N: {                                                                                                     \
    static RvitLdsLimit lim;                                                                                    \
    if (const int rc = lim.ensure({(const void*)layernorm_bwd_kernel<N, T>}, (int)lds, "layernorm_bwd")) return rc; \
    hipLaunchKernelGGL((layernorm_bwd_kernel<N, T>), dim3((unsigned)blocks), dim3(T), lds,                      \
                       st, (const bf16*)dy, lddy, x, ldx, mean, rstd, gamma, dxs, lddx, accumulate,             \
                       (bf16*)dx_bf16, lddb, dgamma, dbeta, rows, E, ws, dk);                                   \
    break;                                                                                                      \
  }
  switch (it) {
    LNB_CASE(1, 1024) LNB_CASE(2, 1024) LNB_CASE(3, 1024) LNB_CASE(4, 256) LNB_CASE(5, 256) LNB_CASE(6, 256)
    LNB_CASE(7, 256) LNB_CASE(8, 256)
    default: rvit_set_error("layernorm_bwd: E=%d unsupported", E); return RVIT_EUNSUPPORTED;
  }
#undef LNB_CASE
  RVIT_CHECK_LAUNCH("layernorm_bwd");
  if (ws) {
    // weight and bias gradients of a LayerNorm lie back to back in a flat gradient buffer: then the two halves of a slot
    // are one vector and one launch adds both (same association per output, same bits)
    if (dbeta == dgamma + E && it * 256 == E) return rvit_ordered_finalize(ws, (int)blocks, V, dgamma, 1, 2 * E, 2 * E, st);
    if (const int rc = rvit_ordered_finalize(ws, (int)blocks, V, dgamma, 1, E, E, st)) return rc;
    return rvit_ordered_finalize(ws + it * 256, (int)blocks, V, dbeta, 1, E, E, st);
  }
  return RVIT_OK;
}

int rvit_ce_loss_impl(const float* logits, long ldl, const long* labels, float* loss_sum, float* dlogits, long lddl,
                      void* dlogits_bf16, long lddb, int B, int C, int C_pad, float eps, float gscale,
                      hipStream_t st) {
  RVIT_CHECK_ARG(logits && labels && B > 0 && C > 0 && C <= 4096 && C_pad >= C, "ce_loss: bad arguments");
  RVIT_CHECK_ARG(!dlogits_bf16 || lddb >= C_pad, "ce_loss: bf16 gradient pitch smaller than C_pad");
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, st, logits, ldl, labels,
                     loss_sum, dlogits, lddl, (bf16*)dlogits_bf16, lddb, B, C, C_pad, eps, gscale);
  RVIT_CHECK_LAUNCH("ce_loss");
  return RVIT_OK;
}

int rvit_colsum_impl(const void* a, long lda, float* out, long M, int N, long row_group, long row_extra,
                     float* ws, size_t ws_bytes, hipStream_t st) {
  RVIT_CHECK_ARG(a && out && M >= 0 && N > 0 && lda >= N, "colsum: bad arguments");
  if (M == 0) return RVIT_OK;
  RVIT_CHECK_ARG(((uintptr_t)a & 15) == 0 || (lda & 7) != 0, "colsum: a must be 16-byte aligned");
  const int xb = (N + 255) / 256;
  long ysplit = (M + 63) / 64;
  const long cap = (256 + xb - 1) / xb;  // few workgroups: the column totals share one L2 channel's atomic unit
  if (ysplit > cap) ysplit = cap;
  if (ws)
    RVIT_CHECK_ARG(rvit_aligned16(ws) && ws_bytes >= (size_t)xb * ysplit * 256 * sizeof(float),
                   "colsum: the ordered-reduction workspace needs %zu bytes", (size_t)xb * ysplit * 256 * sizeof(float));
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)xb, (unsigned)ysplit), dim3(256), 0, st,
                     (const bf16*)a, lda, out, M, N, row_group, row_extra, ws);
  RVIT_CHECK_LAUNCH("colsum");
  if (ws) return rvit_ordered_finalize(ws, (int)ysplit, (long)xb * 256, out, 1, N, N, st);
  return RVIT_OK;
}

int rvit_batch_sum_impl(const float* x, long ldb, float* out, int B, long n, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && B > 0 && n > 0 && (n & 3) == 0 && (ldb & 3) == 0, "batch_sum: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(out), "batch_sum: alignment");
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, x, ldb, out, B, n);
  RVIT_CHECK_LAUNCH("batch_sum");
  return RVIT_OK;
}
