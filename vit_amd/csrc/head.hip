// Classifier tail in one launch: final LayerNorm of the class-token rows, Linear(E, E/2) + GELU, Linear(E/2, classes)
// (train_vit.py:127-131 `self.norm`, `self.head`; :166-185 `x = self.norm(x[:, 0]); return self.head(x)`; the Dropout
// between the two Linears is the identity in eval()).  The per-op sequence layernorm -> gemm(gelu) -> gemm is three
// launches of 3.5 + 15 + 18 us at B = 512 for 0.07 GFLOP: each is one dependent chain of L2 round trips on a handful of
// workgroups.  Here a workgroup owns 16 frames: their normalised rows and the hidden map live in LDS, every wave streams
// its own weight fragments straight from L2, two barriers.
//
// Numerics: the arithmetic of the per-op kernels in the same order - LayerNorm as norm.hip (same lane -> element map,
// same butterfly, explicit roundings of common.h), rounded to bf16; both products accumulated in fp32 over K ascending in
// 32-deep MFMA steps with the weights on the "A" side, as gemm.hip; bias, GELU (gelu_erf), bf16 for the hidden map; fp32
// logits = accumulator + bias.  Same bits as the three launches (tests/test_kernels_gpu.py::test_cls_head_fused).
#include "common.h"
#include "rvit_internal.h"

namespace {

constexpr int HD_ROWS = 16;   // frames per workgroup = one MFMA block of tokens

template <int IT>
__global__ __launch_bounds__(256) void cls_head_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      const bf16* __restrict__ w0, const float* __restrict__ b0,
                                                      const bf16* __restrict__ w3, const float* __restrict__ b3,
                                                      float* __restrict__ logits, long ldl, int B, int E, int HID, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pa = E * 2 + 16, ph = HID * 2 + 16;   // row pitches: 16 spare bytes spread the rows over the banks
  char* const sA = smem;                          // [16][E] bf16: LayerNorm output
  char* const sH = smem + HD_ROWS * pa;           // [16][HID] bf16: GELU map
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * HD_ROWS;

  // the first eight weight fragments of this wave's first hidden block are requested before the LayerNorm: they do not
  // depend on it, and their L2 round trip is what the tail of this kernel is made of
  const int fr = lane & 15, fq = lane >> 4;
  const int ksteps = E >> 5;
  bf16x8 wpre[8];
  {
    const bf16* wrow = w0 + (long)(wave * 16 + fr) * E + fq * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      wpre[i] = (wave * 16 < HID && i < ksteps) ? *reinterpret_cast<const bf16x8*>(wrow + i * 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // ---- LayerNorm: a wave per row, four rows per wave (layernorm_fwd_kernel<true, false, IT>'s arithmetic) ----
  for (int r = wave; r < HD_ROWS; r += 4) {
    const long row = row0 + r;
    if (row >= B) {   // wave-uniform: zero rows
      for (int e = lane * 4; e < E; e += 256) *reinterpret_cast<uint2*>(sA + r * pa + e * 2) = make_uint2(0u, 0u);
      continue;
    }
    float v[IT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      if (e < E) {
        const float4 f = *reinterpret_cast<const float4*>(x + row * ldx + e);
        v[it][0] = f.x; v[it][1] = f.y; v[it][2] = f.z; v[it][3] = f.w;
        s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
      } else {
        v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
      }
    }
    const float inv_e = 1.0f / (float)E;
    const float mean = ln_mean(wave_sum(s), inv_e);
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      if (e < E) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[it][k] = __fsub_rn(v[it][k], mean);
          q = ln_sq(q, v[it][k]);
        }
      }
    }
    const float rstd = ln_rstd(wave_sum(q), inv_e, eps);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = it * 256 + lane * 4;
      if (e < E) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + e);
        const float4 b = *reinterpret_cast<const float4*>(beta + e);
        uint2 o;
        o.x = pack_bf16x2(ln_affine(v[it][0], rstd, g.x, b.x), ln_affine(v[it][1], rstd, g.y, b.y));
        o.y = pack_bf16x2(ln_affine(v[it][2], rstd, g.z, b.z), ln_affine(v[it][3], rstd, g.w, b.w));
        *reinterpret_cast<uint2*>(sA + r * pa + e * 2) = o;
      }
    }
  }
  __syncthreads();

  // ---- Linear(E, HID) + bias + GELU: wave w owns hidden column blocks w, w + 4, ...; lane (fr, fq): weight row
  // n0 + fr and token fr on the operand side, hidden columns n0 + 4 fq .. + 3 of token fr on the result side.  Weight
  // fragments are requested eight K steps at a time, ahead of the MFMAs that use them ----
  for (int nb = wave; nb < (HID >> 4); nb += 4) {
    const int n0 = nb * 16;
    const bf16* wrow = w0 + (long)(n0 + fr) * E + fq * 8;
    const char* arow = sA + fr * pa + fq * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ksteps; k0 += 8) {
      bf16x8 wf[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (nb == wave && k0 == 0) wf[i] = wpre[i];
        else wf[i] = (k0 + i < ksteps) ? *reinterpret_cast<const bf16x8*>(wrow + (k0 + i) * 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (k0 + i < ksteps) {   // wave-uniform
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + (k0 + i) * 64);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af, acc, 0, 0, 0);
        }
      }
    }
    const float4 bb = *reinterpret_cast<const float4*>(b0 + n0 + 4 * fq);
    uint2 o;
    o.x = pack_bf16x2(gelu_erf(acc[0] + bb.x), gelu_erf(acc[1] + bb.y));
    o.y = pack_bf16x2(gelu_erf(acc[2] + bb.z), gelu_erf(acc[3] + bb.w));
    *reinterpret_cast<uint2*>(sH + fr * ph + (n0 + 4 * fq) * 2) = o;
  }
  // the class weights of this wave's first block, requested before the barrier
  const int hsteps = HID >> 5;
  bf16x8 w3pre[8];
  {
    const bool c_ok = wave * 16 + fr < C;
    const bf16* wrow = w3 + (long)(c_ok ? wave * 16 + fr : 0) * HID + fq * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      w3pre[i] = (c_ok && i < hsteps) ? *reinterpret_cast<const bf16x8*>(wrow + i * 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();

  // ---- Linear(HID, C) + bias -> fp32 logits; class rows past C are zero ----
  for (int cb = wave; cb < ((C + 15) >> 4); cb += 4) {
    const int c0 = cb * 16;
    const bool c_ok = c0 + fr < C;
    const bf16* wrow = w3 + (long)(c_ok ? c0 + fr : 0) * HID + fq * 8;
    const char* hrow = sH + fr * ph + fq * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < hsteps; k0 += 8) {
      bf16x8 wf[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (cb == wave && k0 == 0) wf[i] = w3pre[i];
        else wf[i] = (c_ok && k0 + i < hsteps) ? *reinterpret_cast<const bf16x8*>(wrow + (k0 + i) * 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (k0 + i < hsteps) {
          const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hrow + (k0 + i) * 64);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], hf, acc, 0, 0, 0);
        }
      }
    }
    const long row = row0 + fr;
    if (row < B) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + 4 * fq + r;
        if (c < C) logits[row * ldl + c] = acc[r] + b3[c];
      }
    }
  }
}

}  // namespace

bool rvit_cls_head_ok(int E, int HID, int C) {
  return E > 0 && (E & 31) == 0 && E <= 1024 && HID > 0 && (HID & 31) == 0 && HID <= 1024 && C > 0 && C <= 1024;
}

int rvit_cls_head_fwd_impl(const float* x, long ldx, const float* gamma, const float* beta, float eps, const void* w0,
                           const float* b0, const void* w3, const float* b3, float* logits, long ldl, int B, int E, int HID,
                           int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && gamma && beta && w0 && b0 && w3 && b3 && logits, "cls_head: null pointer");
  RVIT_CHECK_ARG(rvit_cls_head_ok(E, HID, C), "cls_head: E=%d HID=%d C=%d unsupported (multiples of 32 up to 1024)", E, HID, C);
  RVIT_CHECK_ARG(B >= 0 && ldx >= E && (ldx & 3) == 0 && ldl >= C, "cls_head: bad batch / pitches");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(gamma) && rvit_aligned16(beta) && rvit_aligned16(w0) && rvit_aligned16(b0) &&
                     rvit_aligned16(w3),
                 "cls_head: alignment");
  if (B == 0) return RVIT_OK;
  const int lds = HD_ROWS * (E * 2 + 16) + HD_ROWS * (HID * 2 + 16);   // <= 49.7 KB
  const unsigned grid = (unsigned)((B + HD_ROWS - 1) / HD_ROWS);
#define HD_CASE(N)                                                                                                    \
  case N:                                                                                                             \
    hipLaunchKernelGGL((cls_head_kernel<N>), dim3(grid), dim3(256), lds, st, x, ldx, gamma, beta, eps, (const bf16*)w0, \
                       b0, (const bf16*)w3, b3, logits, ldl, B, E, HID, C);                                           \
    break;
  switch ((E + 255) / 256) {
    HD_CASE(1) HD_CASE(2) HD_CASE(3) HD_CASE(4)
    default: return RVIT_EUNSUPPORTED;
  }
#undef HD_CASE
  RVIT_CHECK_LAUNCH("cls_head");
  return RVIT_OK;
}
