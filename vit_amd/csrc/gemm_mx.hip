// MX-fp8 GEMM:  C[M,N] = act( (A_q . B_q^T) + bias ) + R,  A_q / B_q = OCP e4m3 with one E8M0 scale
// per 32 consecutive K elements (OCP Microscaling "MXFP8"), fp32 accumulate, on the block-scaled
// matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 MFMA rate on gfx950).
//
// BASELINE.json configs[4]: ViT-L/16-scale inference with fp8 MX weights + activations for the
// encoder GEMMs (QKV in-proj, out-proj, linear1, linear2: torch functional.py:5819-5833,6632-6637,
// transformer.py:980-982); softmax / LayerNorm / residual stream stay bf16 / fp32.
//
// Same skeleton as gemm.hip: 256x64 output tile, 8 waves of 32x64, operand tiles DMA'd global->LDS
// (global_load_lds_dwordx4) into a 2-stage ring, one raw s_barrier per K tile, XOR-swizzled 128-byte
// rows.  A K tile is 128 BYTES per row = 128 fp8 = ONE 16x16x128 instruction deep (the bf16 kernel's
// 64 elements): identical LDS / DMA traffic per tile at twice the FLOPs.
// Operand layout of the 16x16x128 instruction (measured with one-hot operands, scripts/debug_mx.py):
// lane (r = lane&15, g = lane>>4) holds k = 16g..16g+15 in VGPRs 0-3 and k = 64+16g..64+16g+15 in
// VGPRs 4-7 of row r (two ds_read_b128: chunks g and 4+g of the 128-byte row); the E8M0 scale of MX
// block b (k = 32b..32b+31) of row r is taken from byte 0 of the scale VGPR of lane (r, g = b).
// Scale matrices are therefore stored TILED, [ceil(rows/16)][ceil(K/512)][16 rows][4 blocks][4 K tiles]
// bytes: a lane's scales for four consecutive K tiles are one dword, the 64 lanes of a fragment read 256
// contiguous bytes, fetched straight from global once per four K tiles and one group ahead.  (Row-major
// [rows, K/32] scales cost 16 cache lines per fragment load and made the kernel 2.8x slower; one byte
// load per fragment per K tile was still 1.35x slower than this.)
// Epilogue as gemm.hip (LDS restaging, 16-byte global accesses): bias, exact-erf GELU, residual,
// f32 / bf16 output, and optionally an MX-fp8 copy of the result (values + E8M0 scales) so the next
// GEMM consumes it without a separate quantisation pass.
#include "common.h"
#include "rvit_internal.h"
#include "mx_common.h"

namespace {

__device__ uint4 g_zero_page_mx[4];

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

constexpr int MX_BM = 256, MX_BN = 64, MX_WAVES = 8, MX_WM = 32;
constexpr int MX_TILE_BYTES = (MX_BM + MX_BN) * 128;  // 40 KB per stage

__global__ __launch_bounds__(64 * MX_WAVES, 4) void gemm_mxfp8_kernel(const MxGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RPP = 64;                       // tile rows covered by one DMA pass of the workgroup
  constexpr int A_IT = MX_BM / RPP, B_IT = MX_BN / RPP;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  const int ntn = (p.N + MX_BN - 1) / MX_BN;
  int tid_lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tid_lin / ntn, tile_n = tid_lin - tile_m * ntn;
  const long bm0 = (long)tile_m * MX_BM;
  const int bn0 = tile_n * MX_BN;

  const int lrow = t >> 3;
  const int lc = (t & 7) ^ ((t >> 4) & 7);      // source chunk that lands at LDS chunk slot t&7
  const uint8_t* zero_src = reinterpret_cast<const uint8_t*>(g_zero_page_mx);
  const uint8_t* a_row[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const long grow = bm0 + lrow + RPP * i;
    a_row[i] = grow < p.M ? p.A + grow * p.lda + lc * 16 : nullptr;
  }
  const uint8_t* b_row;
  {
    const int n = bn0 + lrow;
    b_row = n < p.N ? p.B + (long)n * p.ldb + lc * 16 : nullptr;
  }
  char* const wave_dst = smem + wave * 1024;
  int k_issue = 0;

#define ISSUE_TILE(stage)                                                                     \
  do {                                                                                        \
    char* dstA = wave_dst + (stage) * MX_TILE_BYTES;                                          \
    char* dstB = dstA + MX_BM * 128;                                                          \
    const bool k_ok = k_issue < p.K;                                                          \
    _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                          \
      glds16((k_ok && a_row[i]) ? (const void*)(a_row[i] + k_issue) : (const void*)zero_src, dstA + i * (RPP * 128)); \
    _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                          \
      glds16((k_ok && b_row) ? (const void*)(b_row + k_issue) : (const void*)zero_src, dstB + i * (RPP * 128)); \
    k_issue += 128;                                                                           \
  } while (0)

  // ---- fragment / scale geometry -----------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int nk = p.K >> 7, nk4 = (nk + 3) >> 2;
  // tiled scales: dword ((row>>4) * nk4 + kt/4) * 64 + (row&15)*4 + block, byte kt%4
  const uint32_t* sa_blk[2];  // 16-row blocks past the padded row count are clamped: their data rows are zero pages
  const long mblk_last = (p.M + 15) / 16 - 1, nblk_last = (p.N + 15) / 16 - 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long mb = (bm0 + wave * MX_WM + i * 16) >> 4;
    if (mb > mblk_last) mb = mblk_last;
    sa_blk[i] = reinterpret_cast<const uint32_t*>(p.SA) + mb * nk4 * 64 + fr * 4 + fq;
  }
  const uint32_t* sb_blk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    long nb = (bn0 + j * 16) >> 4;
    if (nb > nblk_last) nb = nblk_last;
    sb_blk[j] = reinterpret_cast<const uint32_t*>(p.SB) + nb * nk4 * 64 + fr * 4 + fq;
  }
  uint32_t sa_w[2], sb_w[4], sa_n[2], sb_n[4];  // this lane's E8M0 scales: current group of 4 K tiles, next group
#define LOAD_SCALES(g_, sa_, sb_)                                                             \
  do {                                                                                        \
    const int go_ = (g_) * 64;                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) sa_[i] = sa_blk[i][go_];                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) sb_[j] = sb_blk[j][go_];                    \
  } while (0)

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  ISSUE_TILE(0);
  LOAD_SCALES(0, sa_n, sb_n);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt and its scales landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if ((kt & 3) == 0) {                         // a new group of 4 K tiles: its scale dwords landed with this wait
#pragma unroll
      for (int i = 0; i < 2; ++i) sa_w[i] = sa_n[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) sb_w[j] = sb_n[j];
    }
    const int sh = (kt & 3) * 8;
    int sa_s[2], sb_s[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) sa_s[i] = (int)((sa_w[i] >> sh) & 0xff);
#pragma unroll
    for (int j = 0; j < 4; ++j) sb_s[j] = (int)((sb_w[j] >> sh) & 0xff);
    ISSUE_TILE(stage ^ 1);                       // past K: zero pages
    if ((kt & 3) == 0 && kt + 4 < nk) LOAD_SCALES((kt >> 2) + 1, sa_n, sb_n);
    const char* sA = smem + stage * MX_TILE_BYTES;
    const char* sB = sA + MX_BM * 128;
    i32x8 af[2], bfr[4];
    const int c0 = (fq ^ sw) << 4, c1 = ((4 + fq) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* r = sA + (wave * MX_WM + i * 16 + fr) * 128;
      const uint4 lo = *reinterpret_cast<const uint4*>(r + c0), hi = *reinterpret_cast<const uint4*>(r + c1);
      af[i] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* r = sB + (j * 16 + fr) * 128;
      const uint4 lo = *reinterpret_cast<const uint4*>(r + c0), hi = *reinterpret_cast<const uint4*>(r + c1);
      bfr[j] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, sb_s[j], 0,
                                                                     sa_s[i]);
    stage ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_TILE
#undef LOAD_SCALES

  // ---- epilogue --------------------------------------------------------------------------------------
  __builtin_amdgcn_s_barrier();
  constexpr int EP_PITCH = 64 * 4 + 16;
  char* ep = smem + wave * (MX_WM * EP_PITCH);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ep + (i * 16 + fr) * EP_PITCH + (j * 16 + fq * 4) * 4) = acc[i][j];
  __syncthreads();
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int n = bn0 + ec;
  const bool n_ok = n < p.N;  // N % 8 == 0: a lane's 8 columns are all in or all out
#pragma unroll
  for (int pass = 0; pass < MX_WM / 8; ++pass) {
    const int row = pass * 8 + er;
    const long m = bm0 + wave * MX_WM + row;
    const bool ok = m < p.M && n_ok;
    float v[8];
    {
      const float4 lo = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4);
      const float4 hi = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4 + 16);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    if (ok) {
      if (p.bias) {
        const float4 a = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
      if (p.act == RVIT_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      } else if (p.act == RVIT_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.R) {
        if (p.r_f32) {
          const float* rp = (const float*)p.R + m * p.ldr + n;
          const float4 a = *reinterpret_cast<const float4*>(rp);
          const float4 b = *reinterpret_cast<const float4*>(rp + 4);
          v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        } else {
          const uint4 r = *reinterpret_cast<const uint4*>((const bf16*)p.R + m * p.ldr + n);
          v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
          v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
        }
      }
      if (p.C) {
        if (p.c_f32) {
          float* cp = (float*)p.C + m * p.ldc + n;
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          uint4 o;
          o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
          o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<uint4*>((bf16*)p.C + m * p.ldc + n) = o;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (p.Cq) {  // MX-fp8 copy: a 32-column block = 4 adjacent lanes (every lane takes part in the shuffles)
      float amax = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      const int s = mx_scale_e8m0(amax);
      const float inv = mx_inv_scale(s);
      if (ok) {
        uint2 q;
        q.x = mx_pack4(v[0], v[1], v[2], v[3], inv);
        q.y = mx_pack4(v[4], v[5], v[6], v[7], inv);
        *reinterpret_cast<uint2*>(p.Cq + m * p.ldcq + n) = q;
        if ((lane & 3) == 0)  // tiled scale layout of the consumer GEMM (its K = this N)
          p.SCq[mx_scale_offset(m, n, p.N)] = (uint8_t)s;
      }
    }
  }
}

// rows x K (bf16 or f32) -> e4m3 values + E8M0 scales per 32 elements (tiled layout, see the header); one
// lane = 8 consecutive elements
template <typename T>
__global__ __launch_bounds__(256) void quantize_mx_kernel(const T* __restrict__ x, long ldx, uint8_t* __restrict__ q,
                                                          long ldq, uint8_t* __restrict__ sc, long rows, int K) {
  const int k8 = K >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = idx < rows * k8;                 // K % 32 == 0: a block's 4 lanes are all in or all out
  const long r = ok ? idx / k8 : 0;
  const int c = ok ? (int)(idx - r * k8) * 8 : 0;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (ok) {
    if constexpr (sizeof(T) == 4) {
      const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c);
      const float4 b = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const uint4 u = *reinterpret_cast<const uint4*>(x + r * ldx + c);
      v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
      v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const int s = mx_scale_e8m0(amax);
  const float inv = mx_inv_scale(s);
  if (ok) {
    uint2 o;
    o.x = mx_pack4(v[0], v[1], v[2], v[3], inv);
    o.y = mx_pack4(v[4], v[5], v[6], v[7], inv);
    *reinterpret_cast<uint2*>(q + r * ldq + c) = o;
    if ((threadIdx.x & 3) == 0) sc[mx_scale_offset(r, c, K)] = (uint8_t)s;
  }
}

}  // namespace

int rvit_gemm_mxfp8_dispatch(const MxGemmArgs& a, hipStream_t st) {
  RVIT_CHECK_ARG(a.A && a.B && a.SA && a.SB && (a.C || a.Cq), "gemm_mxfp8: null operand");
  RVIT_CHECK_ARG(a.M >= 0 && a.N > 0 && a.K > 0 && (a.K & 127) == 0 && (a.N & 7) == 0,
                 "gemm_mxfp8: K=%d must be a multiple of 128, N=%d of 8", a.K, a.N);
  RVIT_CHECK_ARG(a.lda >= a.K && a.ldb >= a.K && (a.lda & 15) == 0 && (a.ldb & 15) == 0 && rvit_aligned16(a.A) &&
                     rvit_aligned16(a.B),
                 "gemm_mxfp8: A/B must be 16-byte aligned with pitches that are multiples of 16");

  RVIT_CHECK_ARG(a.act >= 0 && a.act <= 2, "gemm_mxfp8: bad act");
  if (a.C) RVIT_CHECK_ARG(a.ldc >= a.N && (a.ldc & (a.c_f32 ? 3 : 7)) == 0 && rvit_aligned16(a.C), "gemm_mxfp8: bad C");
  if (a.bias) RVIT_CHECK_ARG(rvit_aligned16(a.bias), "gemm_mxfp8: bias alignment");
  if (a.R) RVIT_CHECK_ARG(a.ldr >= a.N && (a.ldr & (a.r_f32 ? 3 : 7)) == 0 && rvit_aligned16(a.R), "gemm_mxfp8: bad R");
  if (a.Cq)
    RVIT_CHECK_ARG(a.SCq && (a.N & 127) == 0 && a.ldcq >= a.N && (a.ldcq & 7) == 0 && ((uintptr_t)a.Cq & 7) == 0,
                   "gemm_mxfp8: MX output needs N %% 128 == 0, an 8-byte aligned value matrix and a scale matrix");
  if (a.M == 0) return RVIT_OK;
  // large products with one of the encoder's epilogue forms: the eight-wave 256x256 kernel (gemm_mx8p.hip), same bits
  if (const int mode = rvit_gemm_mx8p_mode(a); mode >= 0) return rvit_gemm_mx8p(a, mode, st);
  const long tiles = ((a.M + MX_BM - 1) / MX_BM) * (long)((a.N + MX_BN - 1) / MX_BN);
  RVIT_CHECK_ARG(tiles < 0x7fffffffL, "gemm_mxfp8: too many tiles");
  constexpr int lds = 2 * MX_TILE_BYTES;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)gemm_mxfp8_kernel}, lds, "gemm_mxfp8")) return rc;
  hipLaunchKernelGGL(gemm_mxfp8_kernel, dim3((unsigned)tiles), dim3(64 * MX_WAVES), lds, st, a);
  RVIT_CHECK_LAUNCH("gemm_mxfp8");
  return RVIT_OK;
}

int rvit_quantize_mx_impl(const void* x, int x_f32, long ldx, uint8_t* q, long ldq, uint8_t* sc, long rows, int K,
                          hipStream_t st) {
  RVIT_CHECK_ARG(x && q && sc && rows >= 0 && K > 0 && (K & 127) == 0, "quantize_mx: K must be a multiple of 128");
  RVIT_CHECK_ARG(ldx >= K && ldq >= K && (ldx & 7) == 0 && (ldq & 7) == 0 && rvit_aligned16(x) && ((uintptr_t)q & 7) == 0,
                 "quantize_mx: pitches / alignment");
  if (rows == 0) return RVIT_OK;
  const long total = rows * (K >> 3);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (x_f32)
    hipLaunchKernelGGL(quantize_mx_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, ldx, q, ldq, sc, rows, K);
  else
    hipLaunchKernelGGL(quantize_mx_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, ldx, q, ldq, sc, rows, K);
  RVIT_CHECK_LAUNCH("quantize_mx");
  return RVIT_OK;
}
