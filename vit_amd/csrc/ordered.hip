// Ordered (bit-reproducible) completion of split reductions.  The reference asks its backend for deterministic
// kernels (train_vit.py:50, torch.backends.cudnn.deterministic = True).  Every reduction of the training path that is
// split over workgroups - weight gradients (split over the M range), bias / LayerNorm-affine column sums, the
// gradient norm - normally ends in fp32 atomics, whose arrival order differs from run to run.  With a caller-owned
// workspace the producers instead STORE one partial result per split ("slot") and this kernel adds the slots in
// ascending order:
//   out[r * ld + c] += sum_{s = 0 .. nslots-1} part[s * slot_stride + r * cols + c]
// HBM-bound: nslots * rows * cols * 4 bytes read, one read-modify-write of out.
#include "common.h"
#include "rvit_internal.h"

namespace {

template <bool VEC4>
__global__ __launch_bounds__(256) void ordered_finalize_kernel(const float* __restrict__ part, int nslots, long slot_stride,
                                                               float* __restrict__ out, long rows, int cols, long ld) {
  constexpr int W = VEC4 ? 4 : 1;
  const long per_row = cols / W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * per_row) return;
  const long r = i / per_row;
  const int c = (int)(i - r * per_row) * W;
  const float* src = part + r * cols + c;
  float* dst = out + r * ld + c;
  if constexpr (VEC4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 4 <= nslots; s += 4) {   // four loads in flight; the additions stay in slot order
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + j) * slot_stride));
        v[j] = make_float4(u[0], u[1], u[2], u[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { t.x += v[j].x; t.y += v[j].y; t.z += v[j].z; t.w += v[j].w; }
    }
    for (; s < nslots; ++s) {
      const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * slot_stride));
      const float4 v = make_float4(u[0], u[1], u[2], u[3]);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float4 o = *reinterpret_cast<float4*>(dst);
    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    *reinterpret_cast<float4*>(dst) = o;
  } else {
    float t = 0.f;
    for (int s = 0; s < nslots; ++s) t += src[(long)s * slot_stride];
    *dst += t;
  }
}

// Few outputs, many slots (the gradient norm: one output, 512 slots; LayerNorm affine gradients: 256-1024 outputs,
// 256 slots): G lanes per output, lane g adds slots g, g + G, ... in order, then the fixed xor butterfly over the G
// lanes: still one association on every run, without a slots-deep dependent chain in one thread.
template <int G>
__device__ __forceinline__ void wide_sum(const float* __restrict__ part, int nslots, long slot_stride, float* __restrict__ out,
                                         long n_out, int cols, long ld, long block) {
  const long o = (block * 256 + threadIdx.x) / G;
  const int g = threadIdx.x % G;
  const bool ok = o < n_out;
  const long r = ok ? o / cols : 0;
  const int c = ok ? (int)(o - r * cols) : 0;
  const float* src = part + r * cols + c;
  float t = 0.f;
  float prev = 0.f;   // the value the total is added to, requested with the slots instead of after them
  if (ok && g == 0) prev = out[r * ld + c];
  if (ok) {
    for (int s = g; s < nslots; s += 8 * G) {   // eight loads in flight (past the end: + 0); the additions stay in slot order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s + j * G < nslots ? __builtin_nontemporal_load(src + (long)(s + j * G) * slot_stride) : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += v[j];
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (ok && g == 0) out[r * ld + c] = prev + t;
}
template <int G>
__global__ __launch_bounds__(256) void ordered_finalize_wide_kernel(const float* __restrict__ part, int nslots, long slot_stride,
                                                                    float* __restrict__ out, long n_out, int cols, long ld) {
  wide_sum<G>(part, nslots, slot_stride, out, n_out, cols, ld, blockIdx.x);
}

// Mid-sized outputs (a 64 x 576 conv weight gradient over 256 slots: 36 workgroups of the kernel above, every thread a
// chain of 64 dependent rounds of loads: 21 us): a workgroup owns 256 / SG float4 columns, thread group g adds the slots
// g, g + SG, ... (eight loads in flight), the groups are combined through LDS in group order.  One association per run.
// Optional second job (the bias gradient that comes with a Linear's weight gradient: one vector of nb outputs over the same
// slots, partb / strideb / outb): the blocks past the first job's add it, 16 lanes per output - one launch for both.
template <int SG>
__global__ __launch_bounds__(256) void ordered_finalize_grouped_kernel(const float* __restrict__ part, int nslots, long slot_stride,
                                                                       float* __restrict__ out, long rows, int cols, long ld,
                                                                       unsigned blocks_a, const float* __restrict__ partb,
                                                                       long strideb, float* __restrict__ outb, int nb) {
  constexpr int CW = 256 / SG;
  __shared__ float4 red[SG][CW];
  if (blockIdx.x >= blocks_a) {   // block-uniform
    wide_sum<16>(partb, nslots, strideb, outb, nb, nb, nb, (long)blockIdx.x - blocks_a);
    return;
  }
  const int c = threadIdx.x % CW, g = threadIdx.x / CW;
  const long per_row = cols / 4;
  const long i = (long)blockIdx.x * CW + c;
  const bool ok = i < rows * per_row;
  const long r = ok ? i / per_row : 0;
  const int col = ok ? (int)(i - r * per_row) * 4 : 0;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);   // the values the totals are added to, requested with the slots
  if (g == 0 && ok) o = *reinterpret_cast<const float4*>(out + r * ld + col);
  if (ok) {
    const float* src = part + r * cols + col;
    int s = g;
    for (; s + 7 * SG < nslots; s += 8 * SG) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(s + j * SG) * slot_stride));
#pragma unroll
      for (int j = 0; j < 8; ++j) { t.x += v[j][0]; t.y += v[j][1]; t.z += v[j][2]; t.w += v[j][3]; }
    }
    for (; s < nslots; s += SG) {
      const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)s * slot_stride));
      t.x += u[0]; t.y += u[1]; t.z += u[2]; t.w += u[3];
    }
  }
  red[g][c] = t;
  __syncthreads();
  if (g == 0 && ok) {
#pragma unroll
    for (int k = 1; k < SG; ++k) { t.x += red[k][c].x; t.y += red[k][c].y; t.z += red[k][c].z; t.w += red[k][c].w; }
    float* dst = out + r * ld + col;
    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    *reinterpret_cast<float4*>(dst) = o;
  }
}

// First stage for very many slots (the per-tile statistics of the stem's first conv: 12,544 slots of 128 columns at
// B = 256): workgroup g adds the slots [g * per, (g + 1) * per) in slot order (eight loads in flight) and stores one
// slot of a G times smaller array; the second stage is the kernel above.  A fixed association on every run.
__global__ __launch_bounds__(128) void slot_group_sum_kernel(const float* __restrict__ part, int nslots, int cols, int per,
                                                             float* __restrict__ out) {
  const int c = blockIdx.y * 128 + threadIdx.x;
  if (c >= cols) return;
  int s = blockIdx.x * per;
  const int end = s + per < nslots ? s + per : nslots;
  float t = 0.f;
  for (; s + 8 <= end; s += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(part + (long)(s + j) * cols + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) t += v[j];
  }
  for (; s < end; ++s) t += __builtin_nontemporal_load(part + (long)s * cols + c);
  out[(long)blockIdx.x * cols + c] = t;
}

}  // namespace

// sums[0..2C) += the per-tile statistics slots a conv epilogue wrote (GemmArgs::stat_part), added in slot order.
// stage: scratch for the first stage when nslots > 512 (>= 256 * 2C floats), may be null otherwise.
int rvit_bn_stats_finalize_impl(const float* part, int nslots, int C, float* sums, float* stage, hipStream_t st) {
  RVIT_CHECK_ARG(part && sums && nslots > 0 && C > 0, "bn_stats_finalize: bad arguments");
  const int cols = 2 * C;
  if (nslots > 512) {
    RVIT_CHECK_ARG(stage != nullptr, "bn_stats_finalize: %d slots need the staging buffer", nslots);
    const int G = 256, per = (nslots + G - 1) / G, groups = (nslots + per - 1) / per;
    hipLaunchKernelGGL(slot_group_sum_kernel, dim3((unsigned)groups, (unsigned)((cols + 127) / 128)), dim3(128), 0, st, part,
                       nslots, cols, per, stage);
    RVIT_CHECK_LAUNCH("bn_stats_finalize");
    return rvit_ordered_finalize(stage, groups, cols, sums, 1, cols, cols, st);
  }
  return rvit_ordered_finalize(part, nslots, cols, sums, 1, cols, cols, st);
}

int rvit_ordered_finalize(const float* part, int nslots, long slot_stride, float* out, long rows, int cols, long ld,
                          hipStream_t st) {
  return rvit_ordered_finalize_pair(part, nslots, slot_stride, out, rows, cols, ld, nullptr, 0, nullptr, 0, st);
}

// ... with a second vector (partb: nslots slots strideb apart, nb outputs) added in the same launch where the first job
// runs the grouped kernel; two launches otherwise.  Which form runs depends on the shapes only: one association per
// output for a given reduction, on every run.
int rvit_ordered_finalize_pair(const float* part, int nslots, long slot_stride, float* out, long rows, int cols, long ld,
                               const float* partb, long strideb, float* outb, int nb, hipStream_t st) {
  RVIT_CHECK_ARG(part && out && nslots > 0 && rows > 0 && cols > 0 && ld >= cols, "ordered_finalize: bad arguments");
  RVIT_CHECK_ARG(!partb || (outb && nb > 0 && strideb >= nb), "ordered_finalize: bad second job");
  if (partb) {
    const bool vec2 = (cols & 3) == 0 && (ld & 3) == 0 && (slot_stride & 3) == 0 && rvit_aligned16(part) && rvit_aligned16(out);
    const long n2 = rows * (cols / 4);
    const long n_out2 = rows * cols;
    const bool wide_first = (n_out2 <= 16 && nslots >= 64) || (n_out2 <= 4096 && nslots >= 128) ||
                            (n_out2 <= 16384 && nslots >= 32 && !(vec2 && n_out2 > 4096));
    const bool grouped = vec2 && !wide_first && ((nslots >= 32 && n2 <= 16384) || (nslots >= 16 && n2 <= 65536));
    if (!grouped || nslots < 32) {   // the bias job keeps the kernel a call of its own would run
      if (const int rc = rvit_ordered_finalize_pair(part, nslots, slot_stride, out, rows, cols, ld, nullptr, 0, nullptr, 0, st)) return rc;
      return rvit_ordered_finalize_pair(partb, nslots, strideb, outb, 1, nb, nb, nullptr, 0, nullptr, 0, st);
    }
  }
  const long n_out = rows * cols;
  // few outputs, many slots (BatchNorm sums from 256-512 conv tiles, LayerNorm affine gradients from 256 workgroups): 64
  // lanes per output, so that a lane's run of slots is one round of loads
  if ((n_out <= 16 && nslots >= 64) || (n_out <= 4096 && nslots >= 128)) {
    hipLaunchKernelGGL(ordered_finalize_wide_kernel<64>, dim3((unsigned)((n_out * 64 + 255) / 256)), dim3(256), 0, st, part, nslots,
                       slot_stride, out, n_out, cols, ld);
    RVIT_CHECK_LAUNCH("ordered_finalize");
    return RVIT_OK;
  }
  const bool vec = (cols & 3) == 0 && (ld & 3) == 0 && (slot_stride & 3) == 0 && rvit_aligned16(part) && rvit_aligned16(out);
  // (above 4096 outputs the float4 grouped kernel below: the stem conv's 64 x 224 weight gradient over 512 slots took 35 us
  // on scalar loads)
  if (n_out <= 16384 && nslots >= 32 && !(vec && n_out > 4096)) {
    hipLaunchKernelGGL(ordered_finalize_wide_kernel<16>, dim3((unsigned)((n_out * 16 + 255) / 256)), dim3(256), 0, st, part, nslots,
                       slot_stride, out, n_out, cols, ld);
    RVIT_CHECK_LAUNCH("ordered_finalize");
    return RVIT_OK;
  }
  const long n = rows * (vec ? cols / 4 : cols);
  RVIT_CHECK_ARG((n + 255) / 256 < 0x7fffffffL, "ordered_finalize: too many elements");
  const unsigned blocks_b = partb ? (unsigned)(((long)nb * 16 + 255) / 256) : 0u;
  if (vec && nslots >= 32 && n <= 16384) {         // up to 1024 workgroups of 16 columns x 16 slot groups
    const unsigned blocks_a = (unsigned)((n + 15) / 16);
    hipLaunchKernelGGL(ordered_finalize_grouped_kernel<16>, dim3(blocks_a + blocks_b), dim3(256), 0, st, part, nslots, slot_stride,
                       out, rows, cols, ld, blocks_a, partb, strideb, outb, nb);
    RVIT_CHECK_LAUNCH("ordered_finalize");
    return RVIT_OK;
  }
  if (vec && nslots >= 16 && n <= 65536) {         // up to 1024 workgroups of 64 columns x 4 slot groups
    const unsigned blocks_a = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(ordered_finalize_grouped_kernel<4>, dim3(blocks_a + blocks_b), dim3(256), 0, st, part, nslots, slot_stride,
                       out, rows, cols, ld, blocks_a, partb, strideb, outb, nb);
    RVIT_CHECK_LAUNCH("ordered_finalize");
    return RVIT_OK;
  }
  const dim3 grid((unsigned)((n + 255) / 256));
  if (vec) hipLaunchKernelGGL(ordered_finalize_kernel<true>, grid, dim3(256), 0, st, part, nslots, slot_stride, out, rows, cols, ld);
  else hipLaunchKernelGGL(ordered_finalize_kernel<false>, grid, dim3(256), 0, st, part, nslots, slot_stride, out, rows, cols, ld);
  RVIT_CHECK_LAUNCH("ordered_finalize");
  return RVIT_OK;
}

// Split count of an ordered reduction over `units` work items: at most what the workspace holds (`sub` slots of
// `elems` floats per split), and such that no split is empty (every slot the finalisation reads has been written).
int rvit_ordered_splits(long units, long splits, int sub, long elems, size_t ws_bytes, const char* what, long* out) {
  const long cap = (long)(ws_bytes / 4 / (size_t)(sub * elems > 0 ? sub * elems : 1));
  if (cap < 1) {
    rvit_set_error("%s: the ordered-reduction workspace holds %zu bytes, one split needs %ld", what, ws_bytes, sub * elems * 4);
    return RVIT_EINVAL;
  }
  if (splits > cap) splits = cap;
  if (splits > units) splits = units;
  if (splits < 1) splits = 1;
  for (;;) {
    const long per = (units + splits - 1) / splits;
    const long need = (units + per - 1) / per;
    if (need == splits) break;
    splits = need;
  }
  *out = splits;
  return RVIT_OK;
}
