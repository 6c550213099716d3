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
// Structure (gfx950): 2/4/8 waves, each owning a 32x64 slice of a {64,128,256}x64 output tile (large products: sixteen
// waves of 64x64 on a 256x256 tile), BK=64.  Operand tiles go
// global -> LDS directly (global_load_lds_dwordx4, no VGPR staging) into a 2-stage ring (every instantiation
// below has STAGES = 2; a 3-deep ring was measured in round 1 and cost more in occupancy than it hid): tile
// kt+1 is issued right after the barrier that frees its stage while tile kt is consumed, and the wave waits with a
// COUNTED vmcnt so that the newer tile stays in flight across the single raw s_barrier per K tile.  LDS rows are 128 B; the 16-byte
// chunk index is XOR-swizzled with (row>>1)&7 — applied on the per-lane SOURCE address (the DMA
// destination is lane-linear) and again on the ds_read_b128 fragment reads, which makes every
// fragment read bank-conflict free.  v_mfma_f32_16x16x32_bf16 with the operands swapped (weights
// as the MFMA "A" side): each lane ends up with 4 consecutive output columns of one row; the
// epilogue (scale, bias, residual, ReLU/GELU, optional pre-activation copy, row remap) restages
// the tile through LDS so that all its global accesses are whole 128-byte row segments.  Out-of-image / out-of-range chunks are sourced from a 16-byte zero page.
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
#include "common.h"
#include "rvit_internal.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K tile = 128 B per LDS row

__device__ uint4 g_zero_page[4];  // never written; source of padded / out-of-range chunks

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool GATHER>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(const GemmArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int RPP = NT / 8;  // tile rows covered by one global_load_lds pass of the workgroup
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int LOADS = A_IT + B_IT;  // global_load_lds instructions per wave per K tile
  static_assert(BM % RPP == 0 && BN % RPP == 0 && STAGES >= 2, "tile / thread-count mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = (BM + BN) * 128;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous id range ----
  const int ntn = (p.N + BN - 1) / BN;
  int tid_lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tid_lin / ntn, tile_n = tid_lin - tile_m * ntn;
  const long bm0 = (long)tile_m * BM;
  const int bn0 = tile_n * BN;

  const int lrow = t >> 3;                      // row within an RPP-row pass
  const int lc = (t & 7) ^ ((t >> 4) & 7);      // source chunk that lands at LDS chunk slot t&7
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page);

  // ---- per-thread row descriptors ---------------------------------------------------------------
  const bf16* a_row[A_IT];  // plain: row base pointer; gather: unused
  int a_pix[A_IT];          // gather: pixel index of tap (0,0); -1 marks an out-of-range row
  int a_hw[A_IT];           // gather: (hi0 << 16) | (wi0 & 0xffff)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const long grow = bm0 + lrow + RPP * i;
    const bool ok = grow < p.M;
    if constexpr (GATHER) {
      // float-reciprocal floor division + fix-up (exact for the ranges the dispatcher admits): the 64-bit `/`
      // sequence cost ~150 vector-ALU instructions per row, four rows per thread, before the first MFMA
      const int g = ok ? (int)grow : 0;
      const int hw = p.Ho * p.Wo;
      int b = (int)((float)g * (1.0f / (float)hw));
      int rem = g - b * hw;
      if (rem < 0) { --b; rem += hw; }
      if (rem >= hw) { ++b; rem -= hw; }
      int ho = (int)((float)rem * (1.0f / (float)p.Wo));
      int wo = rem - ho * p.Wo;
      if (wo < 0) { --ho; wo += p.Wo; }
      if (wo >= p.Wo) { ++ho; wo -= p.Wo; }
      int hi0, wi0;
      if (p.transposed) {  // rows are pixels of dX; taps walk dY at (ho + pad - kh) / stride
        hi0 = ok ? ho + p.pad : -20000;
        wi0 = wo + p.pad;
        a_pix[i] = b * p.H * p.W;
      } else {
        hi0 = ok ? ho * p.stride - p.pad : -20000;
        wi0 = wo * p.stride - p.pad;
        a_pix[i] = (b * p.H + hi0) * p.W + wi0;
      }
      a_hw[i] = (hi0 << 16) | (wi0 & 0xffff);
      a_row[i] = p.A;
    } else {
      long arow = grow;
      if (p.a_row_group > 0) {
        const long gq = grow / p.a_row_group;
        arow = gq * (p.a_row_group + p.a_row_extra) + (grow - gq * p.a_row_group) + p.a_row_extra;
      }
      a_row[i] = ok ? p.A + arow * p.lda : nullptr;
      a_pix[i] = 0;
      a_hw[i] = 0;
    }
  }
  const bf16* b_row[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = bn0 + lrow + RPP * i;
    b_row[i] = n < p.N ? p.B + (long)n * p.ldb : nullptr;
  }

  // running K position of the NEXT tile to issue (per thread: its own chunk)
  int k_issue = lc * 8;
  int g_kh = 0, g_kw = 0, g_off = 0;
  if constexpr (GATHER) {
    const int seg = k_issue / p.seg_len;
    g_off = k_issue - seg * p.seg_len;
    g_kh = seg / p.KW;
    g_kw = seg - g_kh * p.KW;
  }

  char* const wave_dst = smem + wave * 1024;  // + stage*TILE_BYTES + i*RPP*128 (+ BM*128 for B)

#define ISSUE_TILE(stage)                                                                        \
  do {                                                                                           \
    char* dstA = wave_dst + (stage) * TILE_BYTES;                                                \
    char* dstB = dstA + BM * 128;                                                                \
    const bool k_ok = k_issue < p.K;                                                             \
    if constexpr (GATHER) {                                                                      \
      const int dpix = g_kh * p.W + g_kw;                                                        \
      if (p.transposed) {                                                                        \
        const int sh = p.stride >> 1; /* stride 1 or 2 */                                        \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                       \
          const int hn = (a_hw[i] >> 16) - g_kh, wn_ = (int)(short)(a_hw[i] & 0xffff) - g_kw;    \
          const int hi = hn >> sh, wi = wn_ >> sh;                                               \
          const bool ok = k_ok && hn >= 0 && wn_ >= 0 && ((hn | wn_) & (p.stride - 1)) == 0 &&   \
                          hi < p.H && wi < p.W;                                                  \
          const bf16* src = ok ? p.A + (long)(a_pix[i] + hi * p.W + wi) * p.Cpix + g_off : zero_src; \
          glds16(src, dstA + i * (RPP * 128));                                                   \
        }                                                                                        \
      } else {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                       \
          const int hi = (a_hw[i] >> 16) + g_kh, wi = (int)(short)(a_hw[i] & 0xffff) + g_kw;     \
          const bool ok = k_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;  \
          const bf16* src = ok ? p.A + (long)(a_pix[i] + dpix) * p.Cpix + g_off : zero_src;      \
          glds16(src, dstA + i * (RPP * 128));                                                   \
        }                                                                                        \
      }                                                                                          \
    } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                         \
        const bf16* src = (k_ok && a_row[i]) ? a_row[i] + k_issue : zero_src;                    \
        glds16(src, dstA + i * (RPP * 128));                                                            \
      }                                                                                          \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                           \
      const bf16* src = (k_ok && b_row[i]) ? b_row[i] + k_issue : zero_src;                      \
      glds16(src, dstB + i * (RPP * 128));                                                              \
    }                                                                                            \
    k_issue += BK;                                                                               \
    if constexpr (GATHER) {                                                                      \
      g_off += BK;                                                                               \
      while (g_off >= p.seg_len) {                                                               \
        g_off -= p.seg_len;                                                                      \
        if (++g_kw == p.KW) { g_kw = 0; ++g_kh; }                                                \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  // prologue: STAGES-1 tiles in flight (a tile past K is all zero pages: harmless, keeps the
  // counted waits fixed)
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) ISSUE_TILE(s);

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;  // (row>>1)&7 for every fragment row of this lane
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed (this wave's share): leave only the younger tiles' loads in flight
    wait_vmcnt<LOADS*(STAGES - 2)>();
    __builtin_amdgcn_s_barrier();  // every wave's share landed; every wave finished tile kt-1
    {
      int st2 = stage + STAGES - 1;
      if (st2 >= STAGES) st2 -= STAGES;
      ISSUE_TILE(st2);  // overwrites the buffer of tile kt-1
    }
    const char* sA = smem + stage * TILE_BYTES;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MR], bfr[NR];
      const int coff = ((ks * 4 + fq) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < MR; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sA + (wm * WM + i * 16 + fr) * 128 + coff);
#pragma unroll
      for (int j = 0; j < NR; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sB + (wn * WN + j * 16 + fr) * 128 + coff);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (++stage == STAGES) stage = 0;
  }
  wait_vmcnt<0>();  // drain the zero-page tiles issued past K before the LDS is released
#undef ISSUE_TILE

  // ---- epilogue -----------------------------------------------------------------------------
  // The accumulators (lane: 4 consecutive columns of one row) are staged as fp32 through a
  // wave-private LDS region and read back row-major, 8 consecutive columns per lane, so every
  // global access of the epilogue (C, preact copy, residual) is a full 128-B row segment per
  // 8 lanes instead of 8-byte pieces strided over 16 rows.
  __builtin_amdgcn_s_barrier();  // every wave is done reading the operand ring
  constexpr int EP_PITCH = WN * 4 + 16;
  static_assert(WN == 64, "epilogue assumes 64-column wave tiles");
  // rows staged per pass: the whole wave tile when that fits the ring, else 16 at a time (the 256x256 tile)
  constexpr int EP_ROWS = (WAVES_M * WAVES_N * WM * EP_PITCH <= STAGES * TILE_BYTES) ? WM : 16;
  static_assert(WAVES_M * WAVES_N * EP_ROWS * EP_PITCH <= STAGES * TILE_BYTES, "epilogue staging exceeds the ring");
  char* ep = smem + wave * (EP_ROWS * EP_PITCH);
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int n = bn0 + wn * WN + ec;
  float st1[8], st2[8];   // statistics hook (p.stat_part): this lane's 8 columns over the rows it stores
#pragma unroll
  for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
#pragma unroll
  for (int i0 = 0; i0 < MR; i0 += EP_ROWS / 16) {
    if (i0 > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < EP_ROWS / 16; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
        *reinterpret_cast<f32x4*>(ep + (i * 16 + fr) * EP_PITCH + (j * 16 + fq * 4) * 4) = acc[i0 + i][j];
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < EP_ROWS / 8; ++pass) {
      const int row = pass * 8 + er;
      const long m = bm0 + wm * WM + i0 * 16 + row;
      if (m >= p.M || n >= p.N) continue;
      float v[8];
      {
        const float4 lo = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4);
        const float4 hi = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4 + 16);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
      rvit_gemm_epilogue_row8(p, v, m, n);   // leaves the final fp32 values in v
      if (p.stat_part) {
        uint4 o;   // as stored
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        rvit_stat_acc8(o, st1, st2);
      }
    }
  }
  // ---- per-tile batch statistics (train-mode BatchNorm after a conv): lanes that share a column chunk combine by the
  // xor butterfly, the waves of a column block through LDS in wave order, one slot per M tile: fixed association,
  // no atomics (rvit_bn_stats_finalize adds the slots in order)
  if (p.stat_part) {
    // lanes l, l ^ 8, l ^ 16, l ^ 32 on the vector ALU (row rotate, lane-row swaps): __shfl_xor would be 48 LDS
    // crossbar round trips per tile
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      st1[e] = rvit_sum_lanes_8_16_32(st1[e]);
      st2[e] = rvit_sum_lanes_8_16_32(st2[e]);
    }
    __syncthreads();   // every wave has read back its staging rows
    float* sred = reinterpret_cast<float*>(smem);   // [wave][sum 64 | sumsq 64]
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sred[wave * 128 + lane * 8 + e] = st1[e];
        sred[wave * 128 + 64 + lane * 8 + e] = st2[e];
      }
    }
    __syncthreads();
    for (int idx = t; idx < WAVES_N * 128; idx += WAVES_M * WAVES_N * 64) {
      const int wn_ = idx >> 7, q = idx & 127;
      float tot = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WAVES_M; ++w_) tot += sred[(w_ * WAVES_N + wn_) * 128 + q];
      const int col = bn0 + wn_ * WN + (q & 63);
      if (col < p.N) p.stat_part[(long)tile_m * 2 * p.N + (q >> 6) * p.N + col] = tot;
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  const long tiles = ((a.M + BM - 1) / BM) * (long)((a.N + BN - 1) / BN);
  if (tiles <= 0) return RVIT_OK;
  if (tiles > 0x7fffffffL) {
    rvit_set_error("gemm: too many tiles (%ld)", tiles);
    return RVIT_EINVAL;
  }
  constexpr int lds = STAGES * (BM + BN) * 128;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, true>,
                                 (const void*)gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, false>}, lds, "gemm"))
    return rc;
  GemmArgs b = a;
  if (b.stat_part) {   // one statistics slot per M tile (this kernel's epilogue writes them)
    const long slots = (a.M + BM - 1) / BM;
    // (the BatchNorm backward sums, bnr_y, exist in the direct 3x3 kernels only: the caller runs rvit_bn_bwd_reduce)
    if (!a.bnr_y && a.vec && !a.c_f32 && a.row_group == 0 && slots <= a.stat_cap_slots && a.stat_slots) *a.stat_slots = (int)slots;
    else b.stat_part = nullptr;
  }
  if (a.mode == 1)
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, true>), dim3((unsigned)tiles), dim3(64 * WAVES_M * WAVES_N), lds, stream, b);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, false>), dim3((unsigned)tiles), dim3(64 * WAVES_M * WAVES_N), lds, stream, b);
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
    RVIT_CHECK_ARG(a.H < 20000 && a.W < 20000, "conv: image too large for packed coordinates");
    RVIT_CHECK_ARG(a.Ho > 0 && a.Wo > 0 && a.M < 0x7fffffffL && a.M / ((long)a.Ho * a.Wo) < (1L << 22),
                   "conv: M=%ld outside the kernel's 32-bit row decomposition", a.M);
    RVIT_CHECK_ARG(!a.transposed || a.stride == 1 || a.stride == 2, "conv dgrad: stride must be 1 or 2");
  }
  if (a.drop_mode) {
    RVIT_CHECK_ARG(a.drop_mode == RVIT_DROP_PRE_RESIDUAL || a.drop_mode == RVIT_DROP_POST_ACT, "gemm: bad drop_mode");
    const long out_rows = a.row_group > 0 ? (a.M / a.row_group + 1) * (long)(a.row_group + a.row_extra) : a.M;
    RVIT_CHECK_ARG(out_rows * a.N < 0xffffffffL, "gemm: dropout element index exceeds 32 bits");
  }
  if (a.gate) {
    RVIT_CHECK_ARG(a.G && a.ldg >= a.N, "gemm: gate operand missing");
  }
  if (a.stat_slots) *a.stat_slots = 0;   // set by the kernels that implement the statistics epilogue
  if (a.M == 0) return RVIT_OK;
  // 3x3 / stride 1 / pad 1 convs (and their input gradients): LDS-resident patch kernel (conv3x3.hip),
  // at every batch size: the kernel choice must not depend on M or a sample's logits would depend on
  // the batch it travels in (fp32 summation order differs between the two kernels);
  // tile 10 forces it (11 / 12 / 13 / 14 / 15: its 64- / 128-channel / 4-wave / weights-resident / 256-channel variant), any other non-zero tile forces the gather kernel (tests compare the two)
  if (a.tile == 0 || (a.tile >= 10 && a.tile <= 19)) {
    const bool ok = rvit_conv3x3_direct_ok(a);
    if (ok) return rvit_conv3x3_direct(a, a.tile >= 10 ? a.tile - 10 : 0, stream);
    if (a.tile >= 10) {
      rvit_set_error("conv: tile 10 (direct 3x3) does not support this geometry");
      return RVIT_EUNSUPPORTED;
    }
  }
  const int forced = a.tile;
  // Measured on MI355X (profiles/r01_tile_sweep.txt): the 256x64 tile with 8 waves and two LDS
  // stages (80 KB -> 2 workgroups = 16 waves per CU) wins on every stem / encoder shape; smaller
  // tiles only when the problem cannot fill the chip.
  const long n64 = (a.N + 63) / 64;
  // 256x256 tile, sixteen waves of 64x64 in ONE workgroup per CU (128 KB of LDS): 2.5x fewer L2 -> LDS bytes per MAC
  // than 256x64 and 0.5 instead of 0.75 fragment reads per MFMA; same K order, so the results are bit-identical.
  // 1.2 vs 0.9 PFLOP/s at 4096^3, 825 vs 583 TFLOP/s on the ViT-B qkv shape (profiles/r01_gemm_256x256.txt); it needs
  // enough tiles for whole rounds over the 256 CUs and, with few rounds, a long K loop to pay for the un-overlapped
  // prologue / epilogue of a workgroup that has the CU to itself.
  const long t256 = ((a.M + 255) / 256) * (long)((a.N + 255) / 256);
  const bool big_plain = a.mode == 0 && a.K >= 512 && t256 >= 512 && (t256 >= 1024 || a.K >= 2048);
  // implicit-GEMM convs (the stride-2 3x3 convs of the stem): the gathered A tile is re-fetched once per 64 output
  // columns, so the wide tile pays off earlier: B=512 layer3.0.conv1 97.5 -> 82 us, layer4.0.conv1 93 -> 66 us; at 98
  // tiles (layer4, B=256) the chip is too empty and 256x64 wins again (profiles/r02_stride2_tiles.txt).  Same bits.
  const bool big_conv = a.mode == 1 && !a.transposed && (a.N & 255) == 0 && a.K >= 1024 && t256 >= 192;
  const bool big = big_plain || big_conv;
  // the other implicit-GEMM convs with N a multiple of 128 (layer2's stride-2 3x3 conv, the 1x1 downsample convs): a
  // 128x128 tile on eight waves, two workgroups per CU, fetches the gathered A tile once per 128 output columns:
  // B=512 layer2.0.conv1 147 -> 118 us, the three downsample convs 47 / 31 / 23 -> 45 / 29 / 21 us; 256x128 on sixteen
  // waves 135 us (profiles/r03_session2_notes.txt #7).  Same K order, same bits.
  const bool mid_conv = a.mode == 1 && !a.transposed && !big_conv && (a.N & 127) == 0 &&
                        ((a.M + 127) / 128) * (long)(a.N / 128) >= 192;
  int cfg;
  if (forced) cfg = forced;
  else if (a.mode == 0 && a.K >= 512 && t256 >= 512 && rvit_gemm8p_ok(a)) cfg = 5;   // eight waves, four-phase K tile (gemm8p.hip): 0.92-1.13 vs 0.59-0.90 PFLOP/s on the ViT-B Linears, same bits
  else if (big) cfg = 4;
  else if (mid_conv) cfg = 7;
  // plain products below the eight-wave kernel's range with N a multiple of 128 (the E = 256 encoder's Linears at a
  // training batch: 12,800 x {256, 768, 1024} x {256, 768, 1024}): 128x128 on eight waves, two workgroups per CU,
  // 5-10 % ahead of the 64-column tiles (scripts/bench_gemm_small.py: 17.1 -> 15.4, 23.6 -> 21.8, 15.9 -> 14.5 us)
  else if (a.mode == 0 && (a.N & 127) == 0 && ((a.M + 127) / 128) * (long)(a.N / 128) >= 192) cfg = 7;
  else if (((a.M + 255) / 256) * n64 >= 384) cfg = 1;
  else if (((a.M + 127) / 128) * n64 >= 256) cfg = 2;
  else cfg = 3;
  switch (cfg) {
    case 1: return launch_cfg<256, 64, 8, 1, 2>(a, stream);
    case 2: return launch_cfg<128, 64, 4, 1, 2>(a, stream);
    case 3: return launch_cfg<64, 64, 2, 1, 2>(a, stream);
    case 4: return launch_cfg<256, 256, 4, 4, 2>(a, stream);
    case 5: return rvit_gemm8p(a, stream);
    case 6: return launch_cfg<256, 128, 8, 2, 2>(a, stream);
    case 7: return launch_cfg<128, 128, 4, 2, 2>(a, stream);
    // (the same tile with a ring of 3 / 4 stages, one workgroup per CU: 20-31 us against 15.6-22.7 on the N = 768 / 1024
    // shapes, equal elsewhere: profiles/r03_session2_notes.txt)
    // 64x64 wave tiles (<256,64,4>, <128,64,2>, <512,64,8>: fewer LDS reads per MFMA) measured 20-35 % slower:
    // this loop is latency-bound and wants the 16 waves per CU (profiles/r01_tile_sweep_wave64.txt)
    default: rvit_set_error("gemm: unknown tile config %d", cfg); return RVIT_EINVAL;
  }
}
