// bf16 MFMA GEMM with fused epilogue, and the implicit-GEMM convolution built on it.
//
//   C[M,N] = epi( A[M,K] . B[N,K]^T )          (both operands K-contiguous)
//
// One kernel template serves
//   * every Linear of the encoder/head (torch functional.py:5819-5833, 6632-6637;
//     transformer.py:980-982; train_vit.py:122-127),
//   * the 1x1 patch_embed conv (train_vit.py:93,149) incl. the token-major scatter,
//     pos_embed add (train_vit.py:152-159),
//   * every conv of the resnet18 stem (train_vit.py:87,145-146) as an implicit GEMM over an
//     NHWC activation: the A operand row for output pixel m and k-chunk (kh,kw,c0..) is a
//     contiguous channel run of the input pixel, gathered with a bounds predicate (zero pad).
//
// Structure (gfx950): 256 threads = 4 waves, BMxBN output tile, BK=64, register-staged
// global->LDS with two LDS buffers (loads of tile t+1 are in flight while tile t is consumed,
// LDS write after the MFMAs, one barrier per K tile), XOR-swizzled 128-B LDS rows so that every
// ds_read_b128 fragment read is bank-conflict free, v_mfma_f32_16x16x32_bf16 with the operands
// swapped (weights as the MFMA "A" side) so that each lane ends up with 4 consecutive output
// columns of one row -> 8/16-byte vector epilogue accesses.
#include "common.h"
#include "rvit_internal.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K tile = 128 B per LDS row

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool GATHER>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = (BM + BN) * 128;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int ntn = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x - tile_m * ntn;
  const long bm0 = (long)tile_m * BM;
  const int bn0 = tile_n * BN;

  const int lrow = t >> 3;  // 0..31
  const int lc = t & 7;     // 16-byte chunk within the 128-byte K tile row

  // ---- per-thread A row descriptors -------------------------------------------------------
  const bf16* a_ptr[A_IT];
  bool a_ok[A_IT];
  int hi0[A_IT], wi0[A_IT];
  int pix0[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const long grow = bm0 + lrow + 32 * i;
    a_ok[i] = grow < p.M;
    if constexpr (GATHER) {
      const long g = a_ok[i] ? grow : 0;
      const int hw = p.Ho * p.Wo;
      const int b = (int)(g / hw);
      const int rem = (int)(g - (long)b * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      hi0[i] = ho * p.stride - p.pad;
      wi0[i] = wo * p.stride - p.pad;
      pix0[i] = (b * p.H + hi0[i]) * p.W + wi0[i];
      a_ptr[i] = p.A;
    } else {
      a_ptr[i] = p.A + (a_ok[i] ? grow : 0) * p.lda;
      hi0[i] = wi0[i] = pix0[i] = 0;
    }
  }
  const bf16* b_ptr[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = bn0 + lrow + 32 * i;
    b_ok[i] = n < p.N;
    b_ptr[i] = p.B + (long)(b_ok[i] ? n : 0) * p.ldb;
  }

  uint4 ra[A_IT], rb[B_IT];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto stage_load = [&](int kt) {
    const int k = kt * BK + lc * 8;
    const bool k_ok = k < p.K;
    if constexpr (GATHER) {
      const int seg = k / p.seg_len;
      const int off = k - seg * p.seg_len;
      const int kh = seg / p.KW;
      const int kw = seg - kh * p.KW;
      const int dpix = kh * p.W + kw;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int hi = hi0[i] + kh, wi = wi0[i] + kw;
        const bool ok = a_ok[i] && k_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const bf16* src = p.A + (long)(pix0[i] + dpix) * p.Cpix + off;
        ra[i] = ok ? *reinterpret_cast<const uint4*>(src) : zero4;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        ra[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const uint4*>(a_ptr[i] + k) : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      rb[i] = (b_ok[i] && k_ok) ? *reinterpret_cast<const uint4*>(b_ptr[i] + k) : zero4;
  };
  auto stage_write = [&](int buf) {
    char* sA = smem + buf * TILE_BYTES;
    char* sB = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(sA + lds_off(lrow + 32 * i, lc)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(sB + lds_off(lrow + 32 * i, lc)) = rb[i];
  };

  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  stage_load(0);
  stage_write(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage_load(kt + 1);
    const char* sA = smem + buf * TILE_BYTES;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MR], bfr[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wm * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < NR; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sB + lds_off(wn * WN + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stage_write(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds rows n = nbase + 4*fq + {0..3} of column m = mbase + fr ----------
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const long m = bm0 + wm * WM + i * 16 + fr;
    if (m >= p.M) continue;
    long orow = m;
    if (p.row_group > 0) {
      const long g = m / p.row_group;
      orow = g * (p.row_group + p.row_extra) + (m - g * p.row_group) + p.row_extra;
    }
    const long rrow = p.r_period > 0 ? (m % p.r_period) : orow;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int n = bn0 + wn * WN + j * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.vec) {
        if (p.scale) {
          const float4 s = *reinterpret_cast<const float4*>(p.scale + n);
          v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
        }
        if (p.bias) {
          const float4 s = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
        }
        if (p.C2) {
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(p.C2 + orow * p.ldc2 + n) = o;
        }
        if (p.R) {
          if (p.r_f32) {
            const float4 r = *reinterpret_cast<const float4*>((const float*)p.R + rrow * p.ldr + n);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
          } else {
            const uint2 r = *reinterpret_cast<const uint2*>((const bf16*)p.R + rrow * p.ldr + n);
            v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
          }
        }
        if (p.act == RVIT_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == RVIT_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (p.c_f32) {
          *reinterpret_cast<float4*>((float*)p.C + orow * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>((bf16*)p.C + orow * p.ldc + n) = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.N) break;
          float x = v[e];
          if (p.scale) x *= p.scale[n + e];
          if (p.bias) x += p.bias[n + e];
          if (p.C2) p.C2[orow * p.ldc2 + n + e] = (bf16)x;
          if (p.R) x += p.r_f32 ? ((const float*)p.R)[rrow * p.ldr + n + e]
                                : (float)((const bf16*)p.R)[rrow * p.ldr + n + e];
          if (p.act == RVIT_ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == RVIT_ACT_GELU) x = gelu_erf(x);
          if (p.c_f32) ((float*)p.C)[orow * p.ldc + n + e] = x;
          else ((bf16*)p.C)[orow * p.ldc + n + e] = (bf16)x;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  const long tiles = ((a.M + BM - 1) / BM) * (long)((a.N + BN - 1) / BN);
  if (tiles <= 0) return RVIT_OK;
  if (tiles > 0x7fffffffL) {
    rvit_set_error("gemm: too many tiles (%ld)", tiles);
    return RVIT_EINVAL;
  }
  const size_t lds = 2 * (BM + BN) * 128;
  if (a.mode == 1)
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, true>), dim3((unsigned)tiles), dim3(256), lds, stream, a);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, false>), dim3((unsigned)tiles), dim3(256), lds, stream, a);
  RVIT_CHECK_LAUNCH("gemm_bf16");
  return RVIT_OK;
}

}  // namespace

int rvit_gemm_dispatch(const GemmArgs& a, hipStream_t stream) {
  RVIT_CHECK_ARG(a.A && a.B && a.C, "gemm: null operand");
  RVIT_CHECK_ARG(a.M >= 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%ld N=%d K=%d", a.M, a.N, a.K);
  RVIT_CHECK_ARG((a.K & 7) == 0, "gemm: K=%d must be a multiple of 8", a.K);
  RVIT_CHECK_ARG(rvit_aligned16(a.A) && rvit_aligned16(a.B), "gemm: A/B must be 16-byte aligned");
  RVIT_CHECK_ARG((a.ldb & 7) == 0, "gemm: ldb=%ld must be a multiple of 8", a.ldb);
  if (a.mode == 0) {
    RVIT_CHECK_ARG((a.lda & 7) == 0 && a.lda >= a.K, "gemm: lda=%ld invalid for K=%d", a.lda, a.K);
  } else {
    RVIT_CHECK_ARG((a.Cpix & 3) == 0 && (a.seg_len & 7) == 0 && a.seg_len > 0 && a.KW > 0,
                   "conv: bad gather geometry Cpix=%d seg_len=%d KW=%d", a.Cpix, a.seg_len, a.KW);
    RVIT_CHECK_ARG(((long)a.Cpix * 2) % 8 == 0, "conv: pixel pitch must keep 16-byte chunk alignment");
  }
  if (a.M == 0) return RVIT_OK;
  const int forced = a.tile;
  const long t128 = ((a.M + 127) / 128) * (long)((a.N + 127) / 128);
  int cfg;
  if (forced) cfg = forced;
  else if (a.N > 64 && t128 >= 384) cfg = 1;
  else if (((a.M + 127) / 128) * (long)((a.N + 63) / 64) >= 256) cfg = 2;
  else cfg = 3;
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2>(a, stream);
    case 2: return launch_cfg<128, 64, 2, 2>(a, stream);
    case 3: return launch_cfg<64, 64, 2, 2>(a, stream);
    default: rvit_set_error("gemm: unknown tile config %d", cfg); return RVIT_EINVAL;
  }
}
