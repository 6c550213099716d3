// Fused optimiser step on the flat parameter / gradient buffers of the training engine:
// global grad-norm clipping (torch.nn.utils.clip_grad_norm_, train_vit.py:942-943) + AdamW
// (torch.optim.AdamW(lr, weight_decay), train_vit.py:879-883,945) in two launches instead of the
// ~100 small multi-tensor launches of the stock path.  HBM-bound: 16 B read
// (p, g, m, v) + 12 B written (p, m, v) per parameter.
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ __forceinline__ unsigned short bf16_bits(float x) { return (unsigned short)(pack_bf16x2(x, 0.f) & 0xffffu); }

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out,
                                                    float* __restrict__ ws) {
  __shared__ float red[4];
  const long n4 = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = ((red[0] + red[1]) + red[2]) + red[3];
    if (ws) ws[blockIdx.x * 4] = t;   // ordered mode: one slot per workgroup, added in order by rvit_ordered_finalize
    else atomicAdd(out, t);
  }
}

// torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the bias-corrected sqrt):
//   g *= clip;  p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, const float* __restrict__ gnorm_sq,
                                                    float max_norm, bf16* __restrict__ mirror) {
  float clip = 1.f;
  if (gnorm_sq && max_norm > 0.f) {
    const float c = max_norm / (sqrtf(*gnorm_sq) + 1e-6f);  // clip_grad_norm_: coefficient clamped to 1
    clip = c < 1.f ? c : 1.f;
  }
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ge[4] = {gv.x * clip, gv.y * clip, gv.z * clip, gv.w * clip};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pe[k] *= 1.f - lr * wd;
      me[k] = b1 * me[k] + (1.f - b1) * ge[k];
      ve[k] = b2 * ve[k] + (1.f - b2) * ge[k] * ge[k];
      pe[k] -= lr * inv_bc1 * me[k] / (sqrtf(ve[k]) * inv_sqrt_bc2 + eps);
    }
    *reinterpret_cast<float4*>(p + i) = make_float4(pe[0], pe[1], pe[2], pe[3]);
    *reinterpret_cast<float4*>(m + i) = make_float4(me[0], me[1], me[2], me[3]);
    *reinterpret_cast<float4*>(v + i) = make_float4(ve[0], ve[1], ve[2], ve[3]);
    if (mirror) *reinterpret_cast<uint2*>(mirror + i) = make_uint2(pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]));
  } else {
    for (long j = i; j < n; ++j) {
      const float ge = g[j] * clip;
      float pe = p[j] * (1.f - lr * wd);
      const float me = b1 * m[j] + (1.f - b1) * ge, ve = b2 * v[j] + (1.f - b2) * ge * ge;
      pe -= lr * inv_bc1 * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
      p[j] = pe; m[j] = me; v[j] = ve;
      if (mirror) reinterpret_cast<unsigned short*>(mirror)[j] = bf16_bits(pe);
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<uint2*>(out + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  } else {
    for (long j = i; j < n; ++j) reinterpret_cast<unsigned short*>(out)[j] = bf16_bits(x[j]);
  }
}

// One launch transposes every matrix of a table: entry e = {src_off, dst_off, rows, cols, src_ld, dst_ld,
// tile_start, tiles_c} (int64 each); dst[c * dst_ld + r] = src[r * src_ld + c] for r < rows, c < cols.
// rows, cols, leading dimensions and offsets are multiples of 8 (16-byte accesses on both sides).
// A workgroup owns one 64x64 tile, found by binary search on tile_start; the tile goes through LDS
// with a 66-element row pitch so the column-wise reads of the second phase spread over the banks.
constexpr int TB_PITCH = 66;
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                const long* __restrict__ table, int n_entries) {
  __shared__ unsigned short tile[64 * TB_PITCH];
  int lo = 0, hi = n_entries - 1;
  const long t = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 8 + 6] <= t) lo = mid; else hi = mid - 1;
  }
  const long* e = table + lo * 8;
  const long src_off = e[0], dst_off = e[1];
  const int rows = (int)e[2], cols = (int)e[3], src_ld = (int)e[4], dst_ld = (int)e[5], tiles_c = (int)e[7];
  const int lt = (int)(t - e[6]);
  const int r0 = (lt / tiles_c) * 64, c0 = (lt % tiles_c) * 64;
  const unsigned short* s = reinterpret_cast<const unsigned short*>(src) + src_off;
  unsigned short* d = reinterpret_cast<unsigned short*>(dst) + dst_off;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * 256, r = idx >> 3, c = (idx & 7) * 8;
    if (r0 + r < rows && c0 + c < cols) {
      const uint4 u = *reinterpret_cast<const uint4*>(s + (long)(r0 + r) * src_ld + c0 + c);
      unsigned short* w = tile + r * TB_PITCH + c;   // 4-byte aligned: TB_PITCH and c are even
      reinterpret_cast<unsigned*>(w)[0] = u.x; reinterpret_cast<unsigned*>(w)[1] = u.y;
      reinterpret_cast<unsigned*>(w)[2] = u.z; reinterpret_cast<unsigned*>(w)[3] = u.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * 256, c = idx >> 3, r = (idx & 7) * 8;   // output row c, 8 source rows r..r+7
    if (c0 + c < cols && r0 + r < rows) {
      unsigned short h[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) h[k] = tile[(r + k) * TB_PITCH + c];
      uint4 u;
      u.x = h[0] | ((unsigned)h[1] << 16); u.y = h[2] | ((unsigned)h[3] << 16);
      u.z = h[4] | ((unsigned)h[5] << 16); u.w = h[6] | ((unsigned)h[7] << 16);
      *reinterpret_cast<uint4*>(d + (long)(c0 + c) * dst_ld + r0 + r) = u;
    }
  }
}

}  // namespace

int rvit_cast_bf16_impl(const float* x, bf16* out, long n, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && n >= 0 && rvit_aligned16(x) && ((uintptr_t)out & 7) == 0, "cast_bf16: bad arguments");
  if (n == 0) return RVIT_OK;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, x, out, n);
  RVIT_CHECK_LAUNCH("cast_bf16");
  return RVIT_OK;
}

int rvit_transpose_batched_impl(const bf16* src, bf16* dst, const long* table, int n_entries, long total_tiles,
                                hipStream_t st) {
  RVIT_CHECK_ARG(src && dst && table && n_entries > 0 && total_tiles > 0 && total_tiles < (1L << 31),
                 "transpose_batched: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(src) && rvit_aligned16(dst), "transpose_batched: alignment");
  // the table lives in device memory: its entries (multiples of 8, tile counts) are validated by the host
  // wrapper that builds it (vit_amd/trainer.py WeightMirror), which is the only producer
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, src, dst, table, n_entries);
  RVIT_CHECK_LAUNCH("transpose_batched");
  return RVIT_OK;
}

int rvit_sumsq_impl(const float* x, long n, float* out, float* ws, size_t ws_bytes, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && n >= 0 && rvit_aligned16(x), "sumsq: bad arguments");
  if (n == 0) return RVIT_OK;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  if (ws) RVIT_CHECK_ARG(ws_bytes >= (size_t)blocks * 16, "sumsq: the ordered-reduction workspace needs %zu bytes", (size_t)blocks * 16);
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, out, ws);
  RVIT_CHECK_LAUNCH("sumsq");
  if (ws) return rvit_ordered_finalize(ws, (int)blocks, 4, out, 1, 1, 1, st);
  return RVIT_OK;
}

int rvit_adamw_impl(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                    float wd, int step, const float* gnorm_sq, float max_norm, bf16* mirror, hipStream_t st) {
  RVIT_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adamw: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(p) && rvit_aligned16(g) && rvit_aligned16(m) && rvit_aligned16(v), "adamw: alignment");
  RVIT_CHECK_ARG(((uintptr_t)mirror & 7) == 0, "adamw: bf16 mirror must be 8-byte aligned");
  RVIT_CHECK_ARG(b1 >= 0.f && b1 < 1.f && b2 >= 0.f && b2 < 1.f && eps > 0.f, "adamw: bad hyper-parameters");
  if (n == 0) return RVIT_OK;
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2,
                     eps, wd, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), gnorm_sq, max_norm, mirror);
  RVIT_CHECK_LAUNCH("adamw");
  return RVIT_OK;
}
