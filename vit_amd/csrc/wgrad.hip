// Weight-gradient GEMM ("TN"): dW[N,K] += sum_m dY[m,n] * X[m,k], bf16 operands, fp32 atomics.
//
// Backward of every nn.Linear / conv weight on the path (autograd of train_vit.py:141-171 run by
// `scaler.scale(loss).backward()`, train_vit.py:939).  The contraction runs over the ROW index of
// both row-major operands, so fragments are columns of the staged tiles: they are produced by the
// gfx950 transposing LDS read (ds_read_b64_tr_b16), no transposed copies of activations exist in
// HBM.  X may be gathered implicit-GEMM style from an NHWC activation (conv weight gradients).
//
// Structure: 256 threads, 128(n) x 128(k) output tile, 64-row M tiles DMA'd global->LDS
// (global_load_lds_dwordx4) into a 2-stage ring; 256-byte LDS rows whose 32-byte segments are
// XOR-swizzled (on the DMA source address and on the read) so that every transposing read is
// bank-conflict free; the M range is split over gridDim.y and partial tiles are added with
// global_atomic_add_f32 after restaging through LDS (256 contiguous bytes per wave instruction,
// the full-rate atomic shape).  Accumulating into dW matches autograd's `.grad +=` semantics.
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ uint4 g_zero_page_w[4];
// -DRVIT_WG_STAMPS (diagnosis builds only, scripts/wgrad_stamps.py): cycles wave 0 of a workgroup spends in the counted
// wait, in the barrier and in the rest of the M loop of the sixteen-wave kernel
#ifdef RVIT_WG_STAMPS
__device__ unsigned long long g_wg_stamps[4096 * 4];
#define WG_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define WG_ACC(dst, a, b) dst += (b) - (a)
#else
#define WG_T(v) do {} while (0)
#define WG_ACC(dst, a, b) do {} while (0)
#endif

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, r);
}
// Transposing fragment read issued as inline assembly, for loops that keep LDS-DMA in flight across their reads.  A DS
// read the compiler can see makes its wait-count pass wait for EVERY LDS-DMA in flight (it cannot tell the ring stage
// being filled from the one being read): in the sixteen-wave M loop that was an s_waitcnt vmcnt(0) between the request
// of tile t+1 and the first fragment read of tile t - the whole L2 -> LDS latency exposed once per tile (rocprofv3: 53 %
// of the wave cycles parked at a wait; s_memtime: wave 0 waited 2,050 of a tile's 4,922 cycles at the barrier for
// waves still stuck there).  The workgroup barrier at the top of an iteration is what orders these reads after the DMA
// that filled THEIR stage; tr_wait16 retires them (lgkmcnt) before the MFMAs that consume them.
__device__ __forceinline__ void tr_read_asm(s16x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
__device__ __forceinline__ void tr_read_asm_1k(s16x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(dst) : "v"(lds_addr) : "memory");
}
__device__ __forceinline__ void tr_wait16(s16x4 (&y)[8], s16x4 (&x)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(x[0]),
                 "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
               :
               : "memory");
}
// exact floor division for 0 <= m < 2^31, 0 < d < 2^15 via float reciprocal + one fix-up
__device__ __forceinline__ int fdiv(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  int r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; }
  return q;
}

constexpr int TM = 64;             // M rows per staged tile
constexpr int TILE_B = TM * 256;   // bytes of one operand tile (64 rows x 128 bf16)

// NARROW (N <= 64, the 64-channel stem layers): only the first 64 columns of the dY tile are real, so
// instead of idling two waves on zero rows the wave pairs split the 64-row M tile in halves (32 rows
// each) over the same 64(n) x 128(k) output; the atomics of the epilogue add the two halves.
// W8 (not NARROW): eight waves; waves 0-3 contract rows 0-31 of every 64-row M tile, waves 4-7 rows 32-63, each group
// over the full 128 x 128 output (the epilogue's atomics add the two halves).  Twice the waves per SIMD (4 with two
// workgroups per CU), half the LDS-DMA instructions and half the MFMAs per wave between two barriers.
// W16 (plain operands only): sixteen waves on a 256 x 256 output tile, ONE workgroup per CU.  The staged M tile is four
// 64 x 128 sub-tiles (dY left/right, X left/right) in the layout above, one DMA pass each; every wave owns a 64 x 64
// quadrant of one (dY sub-tile, X sub-tile) pair and contracts all 64 rows.  Half the L2 -> LDS bytes per MAC of the
// 128 x 128 tile (the bound of this kernel on the ViT-B/L Linears).
template <bool GATHER, bool NARROW, bool W8, bool W16 = false>
__global__ __launch_bounds__(W16 ? 1024 : W8 ? 512 : 256, (W8 || W16) ? 4 : 2) void wgrad_bf16_kernel(const WgradArgs p, int ntiles, int nsplits) {
  static_assert(!(NARROW && W8) && !(W16 && (NARROW || W8 || GATHER)), "variant combination");
  constexpr int NPASS = W16 ? 1 : W8 ? 2 : 4;  // DMA passes per operand (sub-)tile
  constexpr int TN = W16 ? 256 : 128;          // output tile edge
  constexpr int NSUB = W16 ? 2 : 1;            // 128-column sub-tiles per operand
  constexpr int STAGE_B = 2 * NSUB * TILE_B;   // bytes of one ring stage
  constexpr int PASS_ROWS = TM / NPASS;        // rows per pass (one 16-byte chunk slot per thread)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (dY tile + X tile)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wq = W8 ? (wave & 3) : wave;
  // wave's 64x64 quadrant of the output tile: wn / wk index 64-column blocks of dY / X
  const int wn = NARROW ? 0 : W16 ? wq >> 2 : wq >> 1, wk = W16 ? (wq & 3) : (wq & 1);

  // workgroup -> (output tile, M split), tiles fastest.  (Pinning a split's tiles to one XCD so its L2
  // serves their re-reads was measured and is 5-15 % SLOWER on the stem shapes: profiles/r01_wgrad_splits.txt.)
  int bid = blockIdx.x;
  if constexpr (W16) {
    // XCD-aware id (as gemm.hip): the hardware deals workgroups to the 8 XCDs round-robin; remapped, the ~32 workgroups
    // that share an XCD's L2 are consecutive tiles of (mostly) ONE M split, which read the same 64 rows of their common
    // dY / X column blocks at the same time
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_id = bid % ntiles;
  const int split = bid / ntiles;
  const int ntk = (p.K + TN - 1) / TN;
  const int tile_n = tile_id / ntk, tile_k = tile_id - tile_n * ntk;
  const int n0 = tile_n * TN, k0 = tile_k * TN;

  // M range of this split, in 64-row tiles
  const long mtiles = (p.M + TM - 1) / TM;
  const long per = (mtiles + nsplits - 1) / nsplits;
  const long mt_begin = (long)split * per;
  long mt_end = mt_begin + per;
  if (mt_end > mtiles) mt_end = mtiles;
  const int nmt = (int)(mt_end - mt_begin);
  if (nmt <= 0) return;

  // DMA geometry: thread -> (row within a 16-row pass, destination chunk slot); source chunk is
  // the slot XOR the row swizzle
  const int lrow = t >> 4;
  const int f_w = (lrow & 3) | (((lrow >> 3) & 1) << 2);
  const int lc = (t & 15) ^ (f_w << 1);
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page_w);
  const int ncol = n0 + lc * 8;  // this thread's dY column chunk
  const int kcol = k0 + lc * 8;  // this thread's X column chunk
  const bool n_ok = ncol < p.N;
  const bool k_ok = kcol < p.K;
  int g_kh = 0, g_kw = 0, g_off = 0;
  float inv_hw = 0.f, inv_wo = 0.f;
  if constexpr (GATHER) {
    const int seg = kcol / p.seg_len;
    g_off = kcol - seg * p.seg_len;
    g_kh = seg / p.KW;
    g_kw = seg - g_kh * p.KW;
    inv_hw = 1.0f / (float)(p.Ho * p.Wo);
    inv_wo = 1.0f / (float)p.Wo;
  }
  char* const wave_dst = smem + wave * 1024;

#define ISSUE_MT(stage, mt)                                                                    \
  do {                                                                                         \
    char* dY_dst = wave_dst + (stage) * STAGE_B;                                               \
    char* dX_dst = dY_dst + NSUB * TILE_B;                                                     \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                     \
      const long m = (mt) * TM + ps * PASS_ROWS + lrow;                                        \
      const bool m_ok = m < p.M && (mt) < mt_end;                                              \
      long yr = m;                                                                             \
      if (!W16 && p.y_row_group > 0) {                                                         \
        const long gy = m / p.y_row_group;                                                     \
        yr = gy * (p.y_row_group + p.y_row_extra) + (m - gy * p.y_row_group) + p.y_row_extra;  \
      }                                                                                        \
      const bf16* ysrc = (m_ok && n_ok) ? p.dY + yr * p.ldy + ncol : zero_src;                 \
      glds16(ysrc, dY_dst + ps * (PASS_ROWS * 256));                                           \
      if constexpr (W16) {                                                                     \
        const bf16* ysrc2 = (m_ok && ncol + 128 < p.N) ? p.dY + yr * p.ldy + ncol + 128 : zero_src; \
        glds16(ysrc2, dY_dst + TILE_B);                                                        \
      }                                                                                        \
      const bf16* xsrc = zero_src;                                                             \
      if constexpr (GATHER) {                                                                  \
        if (m_ok && k_ok) {                                                                    \
          const int mi = (int)m;                                                               \
          const int b = fdiv(mi, p.Ho * p.Wo, inv_hw);                                         \
          const int rem = mi - b * p.Ho * p.Wo;                                                \
          const int ho = fdiv(rem, p.Wo, inv_wo);                                              \
          const int wo = rem - ho * p.Wo;                                                      \
          const int hi = ho * p.stride - p.pad + g_kh, wi = wo * p.stride - p.pad + g_kw;      \
          if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)                    \
            xsrc = p.X + ((long)(b * p.H + hi) * p.W + wi) * p.Cpix + g_off;                   \
        }                                                                                      \
      } else {                                                                                 \
        if (m_ok && k_ok) {                                                                    \
          long xr = m;                                                                         \
          if (!W16 && p.x_row_group > 0) {                                                     \
            const long g = m / p.x_row_group;                                                  \
            xr = g * (p.x_row_group + p.x_row_extra) + (m - g * p.x_row_group) + p.x_row_extra; \
          }                                                                                    \
          xsrc = p.X + xr * p.ldx + kcol;                                                      \
        }                                                                                      \
      }                                                                                        \
      glds16(xsrc, dX_dst + ps * (PASS_ROWS * 256));                                           \
      if constexpr (W16) {                                                                     \
        const bf16* xsrc2 = (xsrc != zero_src && kcol + 128 < p.K) ? xsrc + 128 : zero_src;    \
        glds16(xsrc2, dX_dst + TILE_B);                                                        \
      }                                                                                        \
    }                                                                                          \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Bias gradient (column sums of dY) on the matrix pipe: the waves that own the first K tile's left half also
  // multiply each dY fragment i with a B operand that is 1 in column i and 0 elsewhere, all into ONE accumulator:
  // bsum[row 4g+r][col i] = sum_m dY[m][16i + 4g + r].  Four MFMAs per 16, in a loop bound by the L2->LDS fill.
  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
  bool with_bias = false;
  if constexpr (!GATHER && !NARROW) with_bias = p.dbias != nullptr && tile_k == 0 && wk == 0;

  // transposing-read geometry of this lane
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int f_r = q | ((g & 1) << 2);
  const int rd_row = 8 * g + q;                                   // + ks*32 (+4 for the 2nd read)
  const int rd_chunk_lo = pp >> 1;                                // + column-block chunk base
  const int rd_byte = (pp & 1) * 8;

  if constexpr (W16) {
    // Schedule of a 64-row tile t (two 32-row steps, fragments R0 / R1, MFMA groups M0 / M1), one barrier per tile:
    //     wait R0(t) | M0(t) | R1(t), wait | wait DMA(t+1) | barrier | request DMA(t+2) | M1(t) | R0(t+1)
    // The barrier sits between the arrival of R1 and M1, where this wave's M0 (and the other waves' MFMAs) are still
    // executing, and the first reads of the next tile are issued behind M1 instead of behind the barrier: in the
    // {barrier, R0, M0, R1, M1} order every wave of a SIMD waited out the LDS latency of R0 at the same time with an
    // empty matrix pipe.  At the barrier every wave has all fragments of tile t in registers (its stage may be
    // overwritten: DMA(t+2)) and has seen its share of DMA(t+1) land (the next tile may be read).
    typedef __attribute__((address_space(3))) const char lds_cchar;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cchar*)smem;
    s16x4 yr[8], xr[8];   // [2 i] rows 0-3 of an 8-row group, [2 i + 1] rows 4-7
    auto read_step = [&](int stage, int ks) {
      const uint32_t aY = lds0 + stage * STAGE_B + (wn >> 1) * TILE_B;
      const uint32_t aX = lds0 + stage * STAGE_B + NSUB * TILE_B + (wk >> 1) * TILE_B;
      int fsw = f_r << 1;
      asm volatile("" : "+v"(fsw));  // swizzled chunk offsets rebuilt per step: hoisted they spill at 128 VGPRs
      const uint32_t rbase = (uint32_t)((ks * 32 + rd_row) * 256 + rd_byte);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ch = (uint32_t)((((wn & 1) * 8 + i * 2 + rd_chunk_lo) ^ fsw) << 4);
        tr_read_asm(yr[2 * i], aY + rbase + ch);
        tr_read_asm_1k(yr[2 * i + 1], aY + rbase + ch);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t ch = (uint32_t)((((wk & 1) * 8 + j * 2 + rd_chunk_lo) ^ fsw) << 4);
        tr_read_asm(xr[2 * j], aX + rbase + ch);
        tr_read_asm_1k(xr[2 * j + 1], aX + rbase + ch);
      }
    };
    auto mma_step = [&]() {
      bf16x8 yf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = __builtin_bit_cast(bf16x4, yr[2 * i]), hi = __builtin_bit_cast(bf16x4, yr[2 * i + 1]);
        yf[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x4 xl = __builtin_bit_cast(bf16x4, xr[2 * i]), xh = __builtin_bit_cast(bf16x4, xr[2 * i + 1]);
        xf[i] = bf16x8{xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
      if (with_bias) {
        int col = lane & 15;
        asm volatile("" : "+v"(col));  // rebuilt per step: four hoisted selector tuples would not fit 128 VGPRs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const short one = col == i ? (short)0x3f80 : (short)0;  // bf16 1.0
          const s16x8 sel = s16x8{one, one, one, one, one, one, one, one};
          bsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], __builtin_bit_cast(bf16x8, sel), bsum, 0, 0, 0);
        }
      }
    };
    ISSUE_MT(0, mt_begin);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    ISSUE_MT(1, mt_begin + 1);
    read_step(0, 0);
#ifdef RVIT_WG_STAMPS
    unsigned long long tw_vm = 0, tw_bar = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0; it < nmt; ++it) {
      const int stage = it & 1;
      tr_wait16(yr, xr);
      mma_step();               // M0(it)
      read_step(stage, 1);      // R1(it)
      tr_wait16(yr, xr);
      WG_T(t0_);
      wait_vmcnt<0>();          // this thread's share of tile it + 1 has landed
      WG_T(t1_);
      __builtin_amdgcn_s_barrier();
      WG_T(t2_);
      WG_ACC(tw_vm, t0_, t1_);
      WG_ACC(tw_bar, t1_, t2_);
      ISSUE_MT(stage, mt_begin + it + 2);   // into the stage of tile it; past mt_end: zero pages (keeps the protocol uniform)
      mma_step();               // M1(it)
      read_step(stage ^ 1, 0);  // R0(it + 1); after the last tile: a stage of zero pages, never multiplied
    }
    tr_wait16(yr, xr);
#ifdef RVIT_WG_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
      g_wg_stamps[blockIdx.x * 4 + 0] = tw_vm;
      g_wg_stamps[blockIdx.x * 4 + 1] = tw_bar;
      g_wg_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - t_begin;
      g_wg_stamps[blockIdx.x * 4 + 3] = (unsigned long long)nmt;
    }
#endif
  } else {
  typedef __attribute__((address_space(3))) const char lds_cchar_g;
  const uint32_t lds0g = (uint32_t)(uintptr_t)(lds_cchar_g*)smem;
  ISSUE_MT(0, mt_begin);
  int stage = 0;
#ifdef RVIT_WG_STAMPS
  unsigned long long tw_vm = 0, tw_bar = 0, tw_all = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  for (int it = 0; it < nmt; ++it) {
    WG_T(t0_);
    wait_vmcnt<0>();
    WG_T(t1_);
    __builtin_amdgcn_s_barrier();
    WG_T(t2_);
    WG_ACC(tw_vm, t0_, t1_);
    WG_ACC(tw_bar, t1_, t2_);
    ISSUE_MT(stage ^ 1, mt_begin + it + 1);  // past mt_end: zero pages (keeps the protocol uniform)
    // (this branch: the 128 x 128 variants; fragment reads as inline assembly for the reason given at tr_read_asm)
    const uint32_t aY = lds0g + stage * STAGE_B;
    const uint32_t aX = lds0g + stage * STAGE_B + NSUB * TILE_B;
    const int fsw = f_r << 1;
#pragma unroll
    for (int ks0 = 0; ks0 < ((NARROW || W8) ? 1 : 2); ++ks0) {
      const int ks = NARROW ? (wave >> 1) : W8 ? (wave >> 2) : ks0;
      s16x4 yr[8], xr[8];
      const uint32_t rbase = (uint32_t)((ks * 32 + rd_row) * 256 + rd_byte);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ch = (uint32_t)(((wn * 8 + i * 2 + rd_chunk_lo) ^ fsw) << 4);
        tr_read_asm(yr[2 * i], aY + rbase + ch);
        tr_read_asm_1k(yr[2 * i + 1], aY + rbase + ch);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t ch = (uint32_t)(((wk * 8 + j * 2 + rd_chunk_lo) ^ fsw) << 4);
        tr_read_asm(xr[2 * j], aX + rbase + ch);
        tr_read_asm_1k(xr[2 * j + 1], aX + rbase + ch);
      }
      tr_wait16(yr, xr);
      bf16x8 yf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = __builtin_bit_cast(bf16x4, yr[2 * i]), hi = __builtin_bit_cast(bf16x4, yr[2 * i + 1]);
        yf[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x4 xl = __builtin_bit_cast(bf16x4, xr[2 * i]), xh = __builtin_bit_cast(bf16x4, xr[2 * i + 1]);
        xf[i] = bf16x8{xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
      if constexpr (!GATHER && !NARROW) {
        if (with_bias) {
          int col = lane & 15;
          asm volatile("" : "+v"(col));  // rebuilt per step: four hoisted selector tuples would not fit 128 VGPRs
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const short one = col == i ? (short)0x3f80 : (short)0;  // bf16 1.0
            const s16x8 sel = s16x8{one, one, one, one, one, one, one, one};
            bsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], __builtin_bit_cast(bf16x8, sel), bsum, 0, 0, 0);
          }
        }
      }
    }
    stage ^= 1;
  }
#ifdef RVIT_WG_STAMPS
  tw_all = __builtin_amdgcn_s_memtime() - t_begin;
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    g_wg_stamps[blockIdx.x * 4 + 0] = tw_vm;
    g_wg_stamps[blockIdx.x * 4 + 1] = tw_bar;
    g_wg_stamps[blockIdx.x * 4 + 2] = tw_all;
    g_wg_stamps[blockIdx.x * 4 + 3] = (unsigned long long)nmt;
  }
#endif
  }
  wait_vmcnt<0>();
#undef ISSUE_MT
  __builtin_amdgcn_s_barrier();

  // ---- restage through LDS and add whole 256-byte rows atomically ---------------------------------
  // lane holds D[row n = 16i + 4g + r][col k = 16j + (lane&15)]
  // ordered mode (p.part): this split's tile is STORED to its own slot of the workspace, rvit_ordered_finalize adds
  // the slots in order afterwards; the wave groups that share an output tile (W8, NARROW) own a sub-slot each
  constexpr int SUB = (NARROW || W8) ? 2 : 1;
  const long slot = (long)split * SUB + (NARROW ? (wave >> 1) : W8 ? (wave >> 2) : 0);
  if constexpr (!GATHER && !NARROW) {
    if (with_bias && (lane & 15) < 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + 16 * (lane & 15) + 4 * g + r;
        if (n < p.N_store) {
          if (p.part_bias) p.part_bias[slot * p.N_store + n] = bsum[r];
          else atomicAdd(p.dbias + n, bsum[r]);
        }
      }
    }
  }
  float* const pslot = p.part ? p.part + slot * p.part_stride : nullptr;
  constexpr int EP = 64 * 4 + 16;  // fp32 row pitch of the wave-private staging area
  constexpr int EROWS = W16 ? 16 : 32;  // rows staged per pass (sixteen waves x 32 rows would not fit the ring)
  char* ep = smem + wave * (EROWS * EP);
  const int fr = lane & 15;
#pragma unroll
  for (int part = 0; part < 64 / EROWS; ++part) {
#pragma unroll
    for (int ii = 0; ii < EROWS / 16; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float*>(ep + (ii * 16 + 4 * g + r) * EP + (j * 16 + fr) * 4) = acc[part * (EROWS / 16) + ii][j][r];
    __syncthreads();
    const int k = k0 + wk * 64 + lane;
#pragma unroll 4
    for (int row = 0; row < EROWS; ++row) {
      const int n = n0 + wn * 64 + part * EROWS + row;
      if (n < p.N_store && k < p.K) {
        const float v = *reinterpret_cast<const float*>(ep + row * EP + lane * 4);
        if (pslot) pslot[(long)n * p.K + k] = v;
        else atomicAdd(p.dW + (long)n * p.ldw + k, v);
      }
    }
    __syncthreads();
  }
}

}  // namespace

#ifdef RVIT_WG_STAMPS
extern "C" int rvit_debug_wgrad_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg_stamps), (size_t)n * 8);
}
#endif

// Ordered mode: carve the caller's workspace into `sub` slots per split for the weight tile partials (+ the bias sums),
// limit the split count to what fits and make every split non-empty.
static int wgrad_ordered_setup(WgradArgs& o, long units, int sub, long* splits) {
  const long elems = (long)o.N_store * o.K + (o.dbias ? o.N_store : 0);
  RVIT_CHECK_ARG(rvit_aligned16(o.ws), "wgrad: the ordered-reduction workspace must be 16-byte aligned");
  if (const int rc = rvit_ordered_splits(units, *splits, sub, (elems + 3) & ~3L, o.ws_bytes, "wgrad", splits)) return rc;
  o.part_sub = sub;
  o.part = o.ws;
  o.part_stride = (((long)o.N_store * o.K) + 3) & ~3L;
  o.part_bias = o.dbias ? o.ws + *splits * sub * o.part_stride : nullptr;
  return RVIT_OK;
}
static int wgrad_ordered_finish(const WgradArgs& o, int splits, hipStream_t st) {
  if (!o.part) return RVIT_OK;
  const int nslots = splits * o.part_sub;
  // the bias sums ride in the weight gradient's launch where that runs the grouped kernel
  if (o.part_bias)
    return rvit_ordered_finalize_pair(o.part, nslots, o.part_stride, o.dW, o.N_store, o.K, o.ldw, o.part_bias, o.N_store, o.dbias,
                                      o.N_store, st);
  return rvit_ordered_finalize(o.part, nslots, o.part_stride, o.dW, o.N_store, o.K, o.ldw, st);
}

int rvit_wgrad_dispatch(const WgradArgs& a, hipStream_t st) {
  RVIT_CHECK_ARG(a.dY && a.X && a.dW, "wgrad: null pointer");
  RVIT_CHECK_ARG(a.M >= 0 && a.N > 0 && a.K > 0 && a.N_store > 0 && a.N_store <= a.N, "wgrad: bad shape");
  RVIT_CHECK_ARG((a.N & 7) == 0 && (a.K & 7) == 0 && (a.ldy & 7) == 0 && a.ldy >= a.N && a.ldw >= a.K,
                 "wgrad: N, K, ldy must be multiples of 8 (N=%d K=%d ldy=%ld)", a.N, a.K, a.ldy);
  RVIT_CHECK_ARG(rvit_aligned16(a.dY) && rvit_aligned16(a.X), "wgrad: operands must be 16-byte aligned");
  if (a.mode == 0) {
    RVIT_CHECK_ARG((a.ldx & 7) == 0 && a.ldx >= a.K, "wgrad: bad ldx");
  } else {
    RVIT_CHECK_ARG((a.Cpix & 3) == 0 && (a.seg_len & 7) == 0 && a.seg_len > 0 && a.KW > 0 && a.Ho > 0 && a.Wo > 0,
                   "wgrad: bad gather geometry");
    RVIT_CHECK_ARG(a.M < 0x7fffffffL && a.Ho * a.Wo < 32768 * 4 && a.Wo < 32768, "wgrad: gather index range");
  }
  RVIT_CHECK_ARG(!a.dbias || (a.mode == 0 && a.N > 64), "wgrad: the fused bias gradient needs a plain Linear with N > 64");
  if (a.M == 0) return RVIT_OK;
  // 3x3 / stride 1 convs: LDS-resident patch kernel (wgrad3x3.hip).  splits: 0 = automatic choice, -2 forces it,
  // -1 or an explicit split count keep the implicit-GEMM kernel below (tests compare the two)
  if (a.splits == -2 || (a.splits == 0 && rvit_conv3x3_wgrad_direct_ok(a) && rvit_conv3x3_wgrad_direct_pays(a))) {
    WgradArgs d = a;
    d.splits = 0;
    return rvit_conv3x3_wgrad_direct(d, st);
  }
  const long mtiles = (a.M + TM - 1) / TM;
  // Sixteen waves on 256 x 256 tiles, one workgroup per CU: for plain Linears with a long M range (ViT-B/L), where
  // the 128 x 128 kernel is bound by its L2 -> LDS fill.  a.splits == -3 forces it (tests), -4 forbids it.
  if (a.mode == 0 && a.N > 64 && a.splits != -4 && a.splits != -1 && a.splits <= 0 && a.y_row_group == 0 && a.x_row_group == 0) {
    const long tiles16 = (long)((a.N + 255) / 256) * ((a.K + 255) / 256);
    // one workgroup owns a CU, so whole rounds matter: as many M splits as fill ONE round of the chip (fewest atomics;
    // measured against 2-4 rounds on the ViT-B shapes: profiles/r01_wgrad_w16.txt)
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        rvit_set_error("wgrad: cannot query the device");
        return RVIT_ELAUNCH;
      }
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    long sp = n_cu / tiles16;
    if (sp > (mtiles + 3) / 4) sp = (mtiles + 3) / 4;
    if (sp < 1) sp = 1;
    const bool pays = a.N >= 256 && a.K >= 256 && (mtiles + sp - 1) / sp >= 12;
    if (a.splits == -3 || pays) {
      WgradArgs o = a;
      if (a.ws) {
        if (const int rc = wgrad_ordered_setup(o, mtiles, 1, &sp)) return rc;
      }
      RVIT_CHECK_ARG(tiles16 * sp < 0x7fffffffL, "wgrad: too many workgroups");
      const size_t lds16 = 2 * 4 * TILE_B;  // 128 KB
      static RvitLdsLimit lim16;
      if (const int rc = lim16.ensure({(const void*)wgrad_bf16_kernel<false, false, false, true>}, (int)lds16, "wgrad(w16)"))
        return rc;
      hipLaunchKernelGGL((wgrad_bf16_kernel<false, false, false, true>), dim3((unsigned)(tiles16 * sp)), dim3(1024), lds16, st, o,
                         (int)tiles16, (int)sp);
      RVIT_CHECK_LAUNCH("wgrad_bf16");
      return wgrad_ordered_finish(o, (int)sp, st);
    }
  }
  const long tiles = (long)((a.N + 127) / 128) * ((a.K + 127) / 128);
  // ~1024 workgroups (two fit a CU at 64 KB LDS each; two rounds measured best on the stem shapes,
  // profiles/r01_wgrad_splits.txt).  Every split adds N*K atomics, so a split must also own a few M
  // tiles to pay for its epilogue (the encoder Linears: M = B*S is short).
  // (gather form: AT MOST 1024, whole rounds of two workgroups per CU - 144 tiles x 8 splits = 1152 workgroups ran a
  // quarter round behind 144 x 7: 136 against 123 us on layer4's 3x3 convs at B = 256)
  const long target = a.mode == 1 ? 1024 : 512;
  long splits = a.splits > 0 ? a.splits : a.mode == 1 ? (target >= tiles ? target / tiles : 1) : (target + tiles - 1) / tiles;
  if (a.splits <= 0 && splits > (mtiles + 3) / 4) splits = (mtiles + 3) / 4;
  if (splits > mtiles) splits = mtiles;
  if (splits < 1) splits = 1;
  const bool narrow = a.N <= 64;
  WgradArgs o = a;
  if (a.ws) {
    // the variant (and with it the sub-slot count) depends on the split count: settle the splits for two sub-slots
    // when the first guess selects a two-group variant, then keep that variant
    const bool two = narrow || (a.mode == 0 && (mtiles + splits - 1) / splits >= 32);
    if (const int rc = wgrad_ordered_setup(o, mtiles, two ? 2 : 1, &splits)) return rc;
  }
  RVIT_CHECK_ARG(tiles * splits < 0x7fffffffL, "wgrad: too many workgroups");
  const size_t lds = 2 * 2 * TILE_B;  // 64 KB
  dim3 grid((unsigned)(tiles * splits));
  // Eight waves pay when a workgroup owns a long M range (ViT-B/L Linears: 20.2 vs 21.9 ms of weight gradients per
  // ViT-B step); with few M tiles per split the doubled epilogue atomics cost more than the main loop gains
  // (the E=256 encoder: 46 -> 71 us), and the gather variants gain nothing.
  const bool w8 = a.mode == 0 && (mtiles + splits - 1) / splits >= 32 && (!a.ws || o.part_sub == 2);
  if (!narrow && w8) {
    // eight waves: the wave-private epilogue staging (8 x 32 rows x 272 B) is what sizes the LDS
    const size_t lds8 = 8 * 32 * (64 * 4 + 16);
    static RvitLdsLimit lim8;
    if (const int rc = lim8.ensure({(const void*)wgrad_bf16_kernel<true, false, true>,
                                    (const void*)wgrad_bf16_kernel<false, false, true>}, (int)lds8, "wgrad(w8)"))
      return rc;
    if (a.mode == 1) hipLaunchKernelGGL((wgrad_bf16_kernel<true, false, true>), grid, dim3(512), lds8, st, o, (int)tiles, (int)splits);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<false, false, true>), grid, dim3(512), lds8, st, o, (int)tiles, (int)splits);
  } else if (a.mode == 1) {
    if (narrow) hipLaunchKernelGGL((wgrad_bf16_kernel<true, true, false>), grid, dim3(256), lds, st, o, (int)tiles, (int)splits);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<true, false, false>), grid, dim3(256), lds, st, o, (int)tiles, (int)splits);
  } else {
    if (narrow) hipLaunchKernelGGL((wgrad_bf16_kernel<false, true, false>), grid, dim3(256), lds, st, o, (int)tiles, (int)splits);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<false, false, false>), grid, dim3(256), lds, st, o, (int)tiles, (int)splits);
  }
  RVIT_CHECK_LAUNCH("wgrad_bf16");
  // a variant with one wave group per tile leaves the second sub-slot of a two-sub-slot layout unwritten: it is not read
  return wgrad_ordered_finish(o, (int)splits, st);
}
