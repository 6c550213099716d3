// Direct 3x3 / stride 1 / pad 1 NHWC convolution on the eight-wave staggered skeleton of gemm8p.hip (round 3).
//
// Replaces conv3x3.hip's generic kernel on the 3x3 stride-1 convs of resnet layers 2-4 (`self.cnn_backbone(x)[-1]`,
// train_vit.py:87,145-146) and, with `flip`, their input gradients, wherever its tiles fill whole rounds of the CUs
// (rvit_conv3x3_p8_pays).  Same products in the same order (64-channel slice -> tap -> two 32-deep MFMA steps), so the
// results are BIT-IDENTICAL to conv3x3.hip and to the implicit-GEMM kernel.
//
// Why.  conv3x3.hip runs two 8-wave workgroups per CU, one workgroup barrier per tap (16 MFMAs per wave between
// barriers, 0.75 fragment reads per MFMA) and stops at 0.33-0.36 of the bf16 MFMA peak: its waves run their LDS-read
// and MFMA phases in lockstep (profiles/r02_conv3x3_notes.txt).  Here ONE workgroup owns the CU: a wave owns a tall tile
// (7 x 4 fragments = 112 pixels x 64 channels, 56 MFMAs per tap fed by 22 fragment reads = 0.39 per MFMA), and waves
// 4-7 run one barrier behind waves 0-3 (gemm8p.hip's stagger), so on every SIMD one wave multiplies while its partner
// reads and issues DMA.
//
// What differs from the GEMM: the pixel operand is not staged per K tile.  The zero-padded input rows the tile's
// pixels touch ("padded row" space of conv3x3.hip: one shared zero row between images) are RESIDENT in LDS per
// 64-channel slice, double-buffered; the nine taps read them at row offsets kh*(W+2)+kw.  The next slice streams in
// one 1 KB piece per wave per tap (taps 0-7) beside the weight tile (TN x 64, two halves), the only per-tap DMA.
// A tap is two phases (gemm8p.hip's four phases of one quadrant each measured 7-10 % slower here: profiles/
// r03_conv3x3_p8_notes.txt):
//   phase 1: read both weight halves (8 reads, retired before the barrier) and pixel fragments 0-3 (8)      32 MFMAs
//   phase 2: read pixel fragments 4-6 (6); one patch piece of the next slice; stage the weight tile of tap t+2;
//            wait vmcnt(1 + 2 BL)                                                                            24 MFMAs
// (BL = DMA instructions per wave and weight half: 2 for 256 output channels per workgroup, 1 for 128.)  Hazards.  RAW:
// weight tile t+2 is staged in phase 2 of tap t, retired by the counted wait of tap t+1 (which leaves that tap's piece
// and tile t+3 in flight) and read in tap t+2; both wave groups pass a barrier between their wait and the other
// group's first read.  WAR: a tile's reads are retired before the first barrier of its phase 1 (lgkmcnt(8) behind
// reads issued weights-first), for the trailing group that is the barrier the leading group passes before it
// re-stages the slot in phase 2.  The patch buffer of slice c+1 was last read in tap 8 of slice c-1 and is complete
// after tap 8 of slice c (pieces are issued in taps 0-7, tap 8's counted wait retires the last one).  Every tap issues
// the same number of DMA instructions (a piece that does not exist goes from the zero page to a dummy LDS slot), so
// the counted wait is a constant.
//
// Tiles are `tile_len` CONSECUTIVE output pixels x TN channels; wave row r owns pixels [r*ppw, (r+1)*ppw) of the tile
// (ppw <= 112).  The host picks tile_len so that the tile count is a whole number of rounds over the CUs (B=512: half
// an image / one image / four images per tile on the 28- / 14- / 7-wide maps: exactly 4 / 2 / 1 rounds of 256
// workgroups); the result does not depend on that choice.
#include <stdlib.h>

#include "common.h"
#include "rvit_internal.h"
#include "gemm_epilogue.h"

namespace {

__device__ uint4 g_zero_page_c8[4];

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void c8_glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}
// exact floor division through the float reciprocal + one fix-up (ranges checked by the host; conv3x3.hip c3_fdiv)
__device__ __forceinline__ int c8_fdiv(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  const int r = m - q * d;
  if (r < 0) --q;
  if (r >= d) ++q;
  return q;
}

#define C8_SB() __builtin_amdgcn_sched_barrier(0)
#define C8_BAR() __builtin_amdgcn_s_barrier()

constexpr int C8_MF = 7;        // pixel fragments per wave
constexpr int C8_DUMMY = 8192;  // one dummy DMA slot per wave

template <int TN, int STATS = 0>
__global__ __launch_bounds__(512) void conv3x3_p8_kernel(const GemmArgs p, const int patch_bytes, const int tile_len,
                                                          const int ppw) {
  constexpr int WC = TN / 64;            // wave columns (64 channels each)
  constexpr int HALF = (TN / 2) * 128;   // bytes of one weight half tile (TN/2 rows x 64 k)
  constexpr int BL = TN / 128;           // DMA instructions per wave per half: 2 (TN 256) / 1 (TN 128)
  static_assert(TN == 256 || TN == 128, "conv3x3_p8: 256 or 128 channels per workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dummy = smem + 2 * patch_bytes;
  char* const ring = dummy + C8_DUMMY;   // [parity][half] of HALF bytes

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int grp = wave >> 2;             // waves 4-7 run one barrier behind (they share SIMDs with waves 0-3)
  const int H = p.H, W = p.W, Cin = p.Cpix, PW = W + 2, HP = H + 1;
  const int flip = p.transposed;

  // XCD-aware tile id: blocks b and b+8 share an XCD; each XCD owns a contiguous run of tiles (halo rows from its L2)
  const int ntn = p.N / TN;
  int tid_lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tid_lin / ntn, tile_n = tid_lin - tile_m * ntn;
  const int m0 = tile_m * tile_len;
  const int n0 = tile_n * TN;
  const int hw = H * W;
  int m_last = m0 + tile_len - 1;
  if (m_last >= (int)p.M) m_last = (int)p.M - 1;

  const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)W, inv_pw = 1.0f / (float)PW, inv_hp = 1.0f / (float)HP;
  int R_first, R_last;
  {
    const int b0 = c8_fdiv(m0, hw, inv_hw), r0 = m0 - b0 * hw;
    const int b1 = c8_fdiv(m_last, hw, inv_hw), r1 = m_last - b1 * hw;
    R_first = b0 * HP + c8_fdiv(r0, W, inv_w) + 1;
    R_last = b1 * HP + c8_fdiv(r1, W, inv_w) + 1;
  }
  const int R0 = R_first - 1;                    // first padded row held by the patch
  const int P = (R_last - R_first + 3) * PW;     // patch pixels (the host guarantees P * 128 <= patch_bytes)
  const int npass = (P + 7) >> 3;                // 1 KB pieces
  const int nB = c8_fdiv((int)p.M, hw, inv_hw);  // images

  // ---- patch pieces: piece `pass` (wave-uniform) = patch pixels pass*8 .. +7, lane -> pixel lane>>3, 16-byte slot lane&7
  const int prow_l = lane >> 3;
  const int pchunk = ((lane & 7) ^ (prow_l & 6)) * 8;   // source chunk that lands in this slot (row & 6 swizzle)
  const bf16* const zero_src = reinterpret_cast<const bf16*>(g_zero_page_c8);
  auto issue_piece = [&](int pass, int buf, int c0) {
    const bf16* src = zero_src;
    char* dst = dummy + wave * 1024;
    if (pass < npass) {
      dst = smem + buf * patch_bytes + pass * 1024;
      const int row = pass * 8 + prow_l;
      if (row < P) {
        const int pr = c8_fdiv(row, PW, inv_pw), pc = row - pr * PW;
        const int Rp = R0 + pr;
        const int b = c8_fdiv(Rp, HP, inv_hp), hh = Rp - b * HP - 1, ww = pc - 1;
        if (b < nB && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
          src = p.A + ((long)((b * H + hh) * W + ww) * Cin + c0 + pchunk);
      }
    }
    c8_glds16(src, dst);
  };

  // ---- weight half tiles: thread (wave, lane) fills half-tile rows j*64 + wave*8 + (lane >> 3), chunk slot lane & 7
  const int srow = wave * 8 + (lane >> 3);
  const int lc = (lane & 7) ^ ((srow >> 1) & 7);
  const bf16* pbw[2][BL];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      const int n = n0 + (j * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + (lane >> 3);
      pbw[h][j] = p.B + (long)n * p.ldb + lc * 8;
    }
  char* const wave_dst = ring + wave * 1024;
#define C8_STAGE_B(h, d, koff)                                                           \
  do {                                                                                   \
    char* dst_ = wave_dst + (d) * (2 * HALF) + (h) * HALF;                               \
    _Pragma("unroll") for (int j_ = 0; j_ < BL; ++j_) c8_glds16(pbw[h][j_] + (koff), dst_ + j_ * 8192); \
  } while (0)

  // ---- fragment geometry ----
  const int fr = lane & 15, fq = lane >> 4;
  const int mw0 = m0 + wr * ppw;                       // first pixel of this wave row
  int mw_end = mw0 + ppw;                              // one past its last live pixel
  if (mw_end > m_last + 1) mw_end = m_last + 1;
  int pb[C8_MF];
#pragma unroll
  for (int i = 0; i < C8_MF; ++i) {
    int m = mw0 + i * 16 + fr;
    if (m >= mw_end) m = m0;   // any in-window pixel: the result is discarded by the epilogue
    const int b = c8_fdiv(m, hw, inv_hw), r = m - b * hw;
    const int h = c8_fdiv(r, W, inv_w), w = r - h * W;
    pb[i] = (b * HP + h + 1 - R_first) * PW + w;
  }
  const int sw_b = (fr >> 1) & 7;
  const int bc0 = (fq ^ sw_b) << 4;                    // k-step 0 chunk of a weight row; k-step 1 = byte offset ^ 64
  const int bx1 = (bc0 ^ 64) - bc0;
  const char* const b_rd = ring + (wc * 32 + fr) * 128 + bc0;

  bf16x8 af[4][2], bq[2][2][2];
  f32x4 acc[C8_MF][4];
#pragma unroll
  for (int i = 0; i < C8_MF; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define C8_READ_B(nh, boff)                                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                 \
    bq[nh][j_][0] = *reinterpret_cast<const bf16x8*>(b_rd + (boff) + (nh) * HALF + j_ * 2048);       \
    bq[nh][j_][1] = *reinterpret_cast<const bf16x8*>(b_rd + (boff) + (nh) * HALF + j_ * 2048 + bx1); \
  }
  // pixel fragments i0 .. i0+n-1 of the wave at tap row offset `roff` out of patch buffer `patch`
#define C8_READ_A(i0, n, patch, roff)                                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                               \
    const int row_ = pb[(i0) + i_] + (roff);                                                         \
    const int ch_ = (fq ^ (row_ & 6)) << 4;          /* k-step 0 chunk; k-step 1 = byte offset ^ 64 */ \
    const char* a_ = (patch) + row_ * 128;                                                           \
    af[i_][0] = *reinterpret_cast<const bf16x8*>(a_ + ch_);                                          \
    af[i_][1] = *reinterpret_cast<const bf16x8*>(a_ + (ch_ ^ 64));                                   \
  }
#define C8_MMA(i0, n, nh)                                                                            \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_)                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                 \
      acc[(i0) + i_][(nh) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                       \
          bq[nh][j_][ks_], af[i_][ks_], acc[(i0) + i_][(nh) * 2 + j_], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)

  const int nsl = Cin >> 6;            // 64-channel slices
  const int ntap = nsl * 9;            // K tiles
  // element offset of K tile u in a weight row: tap (flipped for the input gradient) * Cin + slice * 64
  auto koff_of = [&](int u) {
    if (u > ntap - 1) u = ntap - 1;    // past the end: re-load the last tile (nobody reads it; keeps the counts fixed)
    const int c = u / 9, tp = u - c * 9;
    return (flip ? 8 - tp : tp) * Cin + (c << 6);
  };

  // ---- prologue: slice 0 of the patch, weight tiles 0 and 1 ----
  for (int pass = wave; pass < npass; pass += 8) issue_piece(pass, 0, 0);
  C8_STAGE_B(0, 0, koff_of(0));
  C8_STAGE_B(1, 0, koff_of(0));
  C8_STAGE_B(0, 1, koff_of(1));
  C8_STAGE_B(1, 1, koff_of(1));
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BL) : "memory");
  C8_SB();
  C8_BAR();
  C8_SB();
  if (grp == 1) C8_BAR();   // waves 4-7 run one barrier behind
  C8_SB();

  int u = 0;                // K tile
  for (int c = 0; c < nsl; ++c) {
    const char* const patch = smem + (c & 1) * patch_bytes;
    const bool has_next = c + 1 < nsl;
    const int c_next = (c + 1) << 6;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++u) {
      const int roff = (tap / 3) * PW + (tap % 3);
      const int boff = (u & 1) * (2 * HALF);
      const int k2 = koff_of(u + 2);
      /* phase 1: both weight halves (retired before the barrier) and pixel fragments 0-3; 32 MFMAs */
      C8_READ_B(0, boff);
      C8_READ_B(1, boff);
      C8_SB();
      C8_READ_A(0, 4, patch, roff);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      C8_SB();
      C8_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      C8_SB();
      C8_MMA(0, 4, 0);
      C8_MMA(0, 4, 1);
      C8_SB();
      C8_BAR();
      C8_SB();
      /* phase 2: pixel fragments 4-6; the DMA of this tap; 24 MFMAs */
      C8_READ_A(4, 3, patch, roff);
      issue_piece((has_next && tap < 8) ? tap * 8 + wave : 0x7fffffff, (c & 1) ^ 1, c_next);
      C8_STAGE_B(0, u & 1, k2);
      C8_STAGE_B(1, u & 1, k2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + 2 * BL) : "memory");
      C8_SB();
      C8_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      C8_SB();
      C8_MMA(4, 3, 1);
      C8_MMA(4, 3, 0);
      C8_SB();
      C8_BAR();
      C8_SB();
    }
  }
  if (grp == 0) C8_BAR();   // rejoin
  C8_SB();
#undef C8_STAGE_B
#undef C8_READ_A
#undef C8_READ_B
#undef C8_MMA

  // ---- epilogue, in registers (as gemm8p.hip): v_permlane16_swap between lane rows of channel blocks 2pr and 2pr+1
  // leaves every lane 8 consecutive channels of pixel row fr: a store instruction writes 16 pixels x 64 bytes.
  // Folded-BatchNorm scale / bias, residual (bf16, same layout as the output), ReLU.
  const int co_lane = (fq & 1) * 16 + (fq >> 1) * 8;
  float sc[2][8], bi[2][8];
  int nl[2];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    nl[pr] = n0 + wc * 64 + pr * 32 + co_lane;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[pr][e] = 1.f; bi[pr][e] = 0.f; }
    if (p.scale) {
      const float4 a = *reinterpret_cast<const float4*>(p.scale + nl[pr]), b = *reinterpret_cast<const float4*>(p.scale + nl[pr] + 4);
      sc[pr][0] = a.x; sc[pr][1] = a.y; sc[pr][2] = a.z; sc[pr][3] = a.w; sc[pr][4] = b.x; sc[pr][5] = b.y; sc[pr][6] = b.z; sc[pr][7] = b.w;
    }
    if (p.bias) {
      const float4 a = *reinterpret_cast<const float4*>(p.bias + nl[pr]), b = *reinterpret_cast<const float4*>(p.bias + nl[pr] + 4);
      bi[pr][0] = a.x; bi[pr][1] = a.y; bi[pr][2] = a.z; bi[pr][3] = a.w; bi[pr][4] = b.x; bi[pr][5] = b.y; bi[pr][6] = b.z; bi[pr][7] = b.w;
    }
  }
  const bool has_r = p.R != nullptr;
  const int relu = p.act == RVIT_ACT_RELU;
  // every residual row of the wave is requested up front (the fragment registers are dead: 14 x 16 bytes per lane fit):
  // one exposed memory latency per tile instead of one per fragment row
  float st1[2][8], st2[2][8];   // STATS: this lane's 16 channels over the pixels it stores
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int e = 0; e < 8; ++e) { st1[pr][e] = 0.f; st2[pr][e] = 0.f; }
  // STATS 2 (GemmArgs::bnr_y): the norm's input rows and ReLU bits travel one fragment row ahead of their use, with the
  // residual rows (all rows up front as below would not fit beside them)
  float bmean[2][8];
  uint4 nr[2], ny[2];
  unsigned nm[2];
  auto bnr_load = [&](int i0) {
    int mr = mw0 + i0 * 16 + fr;
    if (mr > m_last) mr = m_last;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (has_r) nr[pr] = *reinterpret_cast<const uint4*>((const bf16*)p.R + (long)mr * p.ldr + nl[pr]);
      ny[pr] = *reinterpret_cast<const uint4*>(p.bnr_y + (long)mr * p.N + nl[pr]);
      nm[pr] = p.bnr_mask ? p.bnr_mask[(long)mr * (p.N >> 3) + (nl[pr] >> 3)] : 0xffu;
    }
  };
  if constexpr (STATS == 2) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) bmean[pr][e] = p.bnr_fsums[nl[pr] + e] * p.bnr_inv_m;
    bnr_load(0);
  }
  uint4 rr[C8_MF][2];
  if (has_r && STATS != 2) {
#pragma unroll
    for (int i0 = 0; i0 < C8_MF; ++i0) {
      int mr = mw0 + i0 * 16 + fr;
      if (mr > m_last) mr = m_last;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) rr[i0][pr] = *reinterpret_cast<const uint4*>((const bf16*)p.R + (long)mr * p.ldr + nl[pr]);
    }
  }
#pragma unroll
  for (int i0 = 0; i0 < C8_MF; ++i0) {
    const int m = mw0 + i0 * 16 + fr;
    uint4 cr[2], cy[2];
    unsigned cm[2];
    if constexpr (STATS == 2) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) { cr[pr] = nr[pr]; cy[pr] = ny[pr]; cm[pr] = nm[pr]; }
      if (i0 + 1 < C8_MF) bnr_load(i0 + 1);
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i0][2 * pr][e]),
                                                          __float_as_uint(acc[i0][2 * pr + 1][e]), false, false);
        v[e] = __uint_as_float(sw2[0]);
        v[4 + e] = __uint_as_float(sw2[1]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[pr][e] + bi[pr][e];
      if (has_r) {
        const uint4 r = STATS == 2 ? cr[pr] : rr[i0][pr];
        v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
        v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (m < mw_end) {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>((bf16*)p.C + (long)m * p.ldc + nl[pr]) = o;
        if constexpr (STATS == 1) rvit_stat_acc8(o, st1[pr], st2[pr]);
        if constexpr (STATS == 2) rvit_bnr_acc8(o, cy[pr], cm[pr], bmean[pr], st1[pr], st2[pr]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the (unused) tiles past the end must land before the LDS is released
  if constexpr (STATS != 0) {
    // per-tile batch statistics (the conv of a conv -> BatchNorm pair in a training step, GemmArgs::stat_part): the
    // sixteen pixel lanes of a lane row on the vector ALU, then the wave rows of a column block through LDS in wave
    // order; one slot per M tile, fixed association, no atomics
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st1[pr][e] = rvit_sum_row16(st1[pr][e]);
        st2[pr][e] = rvit_sum_row16(st2[pr][e]);
      }
    __syncthreads();   // every wave is past its last fragment read: the patch buffers are free
    float* sred = reinterpret_cast<float*>(smem);   // [wave][sum 64 | sumsq 64], channel = column inside the wave's 64
    if (fr == 0) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sred[wave * 128 + pr * 32 + co_lane + e] = st1[pr][e];
          sred[wave * 128 + 64 + pr * 32 + co_lane + e] = st2[pr][e];
        }
    }
    __syncthreads();
    constexpr int WR = 8 / WC;
    for (int idx = t; idx < WC * 128; idx += 512) {
      const int wc_ = idx >> 7, q = idx & 127;
      float tot = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WR; ++w_) tot += sred[(w_ * WC + wc_) * 128 + q];
      const int col = n0 + wc_ * 64 + (q & 63);
      if (STATS == 2 && q >= 64) tot *= rvit_bnr_rstd(p.bnr_fsums, p.N, col, p.bnr_inv_m, p.bnr_eps);
      p.stat_part[(long)tile_m * 2 * p.N + (q >> 6) * p.N + col] = tot;
    }
  }
}

struct C8Plan {
  int tn, tile_len, ppw, patch_bytes, lds;
  long tiles;
  float eff;   // useful share of the MFMA work the launch issues: live pixel rows x filled rounds
};

// patch rows (padded-row space) the tile of pixels [m0, m0 + len) needs: R_last - R_first + 3, as the kernel computes
static int c8_rows(long m0, int len, long M, int H, int W) {
  long ml = m0 + len - 1;
  if (ml > M - 1) ml = M - 1;
  const long hw = (long)H * W, HP = H + 1;
  const long rf = (m0 / hw) * HP + (m0 % hw) / W + 1, rl = (ml / hw) * HP + (ml % hw) / W + 1;
  return (int)(rl - rf + 3);
}

static bool c8_plan_compute(const GemmArgs& a, int num_cu, C8Plan* out);
// the plan of the last geometry seen by this thread (a training / inference step repeats a handful of geometries)
static bool c8_plan(const GemmArgs& a, int num_cu, C8Plan* out) {
  struct Key { long M; int N, H, W, C, cu; };
  static thread_local Key keys[8];
  static thread_local C8Plan plans[8];
  static thread_local int ok[8], used = 0, next = 0;
  for (int i = 0; i < used; ++i)
    if (keys[i].M == a.M && keys[i].N == a.N && keys[i].H == a.H && keys[i].W == a.W && keys[i].C == a.Cpix && keys[i].cu == num_cu) {
      *out = plans[i];
      return ok[i] != 0;
    }
  C8Plan pl{};
  const bool good = c8_plan_compute(a, num_cu, &pl);
  const int slot = used < 8 ? used++ : (next++ & 7);
  keys[slot] = Key{a.M, a.N, a.H, a.W, a.Cpix, num_cu};
  plans[slot] = pl;
  ok[slot] = good;
  *out = pl;
  return good;
}

static bool c8_plan_compute(const GemmArgs& a, int num_cu, C8Plan* out) {
  const int tn = (a.N & 255) == 0 ? 256 : 128;
  if ((a.N & 127) != 0) return false;
  const int wrows = 8 / (tn / 64);
  const int max_len = wrows * 16 * C8_MF;
  const long ntn = a.N / tn;
  const int half = (tn / 2) * 128;
  // tile_len: the largest length <= max_len for which the tiles are a whole number of rounds over the CUs (r rounds),
  // else max_len; the patch of every tile must fit two buffers beside the weight ring and stream in within 8 taps
  for (int r = 1; r <= 256; ++r) {
    const long want = ((long)r * num_cu + ntn - 1) / ntn;            // m tiles for r rounds
    long len = (a.M + want - 1) / want;
    if (len > max_len) continue;
    if (len < 16 * wrows) len = 16 * wrows;
    const long mt = (a.M + len - 1) / len;
    int rows = 0;
    const long hw = (long)a.H * a.W;
    const long probe = mt < 2 * hw ? mt : 2 * hw;                    // the alignment pattern repeats within H*W tiles
    for (long i = 0; i < probe; ++i) {
      const int rr = c8_rows(i * len, (int)len, a.M, a.H, a.W);
      if (rr > rows) rows = rr;
    }
    const int rr_last = c8_rows((mt - 1) * len, (int)len, a.M, a.H, a.W);
    if (rr_last > rows) rows = rr_last;
    const int P = rows * (a.W + 2);
    const int pieces = (P + 7) / 8;
    const int lds = 2 * pieces * 1024 + C8_DUMMY + 4 * half;
    if (pieces > 64 || lds > 160 * 1024) continue;
    out->tn = tn; out->tile_len = (int)len; out->ppw = (int)((len + wrows - 1) / wrows);
    out->patch_bytes = pieces * 1024; out->lds = lds; out->tiles = mt * ntn;
    const long rounds = (out->tiles + num_cu - 1) / num_cu;
    out->eff = (float)((double)a.M / ((double)mt * wrows * 16 * C8_MF) * ((double)out->tiles / ((double)rounds * num_cu)));
    return true;
  }
  return false;
}

static int c8_num_cu() {
  static std::atomic<int> cached{0};
  int v = cached.load(std::memory_order_relaxed);
  if (v > 0) return v;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cached.store(n, std::memory_order_relaxed);
  return n;
}

}  // namespace

// Same admission as conv3x3.hip's generic kernel plus: at least 128 output channels, >= 128 input channels (two or
// more slices: the first slice's patch load is the only one nothing overlaps), a plan that fits the LDS.
bool rvit_conv3x3_p8_ok(const GemmArgs& a) {
  if (!rvit_conv3x3_direct_ok(a)) return false;
  if ((a.N & 127) != 0 || a.Cpix < 128 || a.M >= (1L << 30)) return false;
  C8Plan pl;
  return c8_plan(a, c8_num_cu(), &pl);
}

// Does it pay?  Every workgroup multiplies 7 fragments per wave whatever its tile holds and owns its CU: the kernel wins
// (B=512: layers 2 / 3 / 4 123 / 105 / 100 us against 133 / 118 / 106) when the tiles are nearly full and the rounds
// nearly whole, and loses on small or odd batches.  Safe to decide by batch size: the results are bit-identical.
bool rvit_conv3x3_p8_pays(const GemmArgs& a) {
  C8Plan pl;
  if (!rvit_conv3x3_direct_ok(a) || (a.N & 127) != 0 || a.Cpix < 128 || a.M >= (1L << 30)) return false;
  return c8_plan(a, c8_num_cu(), &pl) && pl.eff >= 0.82f;
}

int rvit_conv3x3_p8(const GemmArgs& a, hipStream_t st) {
  C8Plan pl;
  RVIT_CHECK_ARG(rvit_conv3x3_direct_ok(a) && (a.N & 127) == 0 && a.Cpix >= 128 && a.M < (1L << 30) && c8_plan(a, c8_num_cu(), &pl),
                 "conv3x3: the eight-wave four-phase kernel does not support this geometry");
  if (pl.tiles <= 0) return RVIT_OK;
  RVIT_CHECK_ARG(pl.tiles < 0x7fffffffL, "conv3x3: too many tiles");
  const long mtiles = pl.tiles / (a.N / pl.tn);
  const int stats = rvit_conv_stat_mode(a, mtiles);
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv3x3_p8_kernel<256, 0>, (const void*)conv3x3_p8_kernel<128, 0>,
                                 (const void*)conv3x3_p8_kernel<256, 1>, (const void*)conv3x3_p8_kernel<128, 1>,
                                 (const void*)conv3x3_p8_kernel<256, 2>, (const void*)conv3x3_p8_kernel<128, 2>},
                                160 * 1024, "conv3x3_p8"))
    return rc;
#define C8_LAUNCH(TN_, S_) \
  hipLaunchKernelGGL((conv3x3_p8_kernel<TN_, S_>), dim3((unsigned)pl.tiles), dim3(512), pl.lds, st, a, pl.patch_bytes, pl.tile_len, pl.ppw)
  if (pl.tn == 256) {
    if (stats == 1) C8_LAUNCH(256, 1); else if (stats == 2) C8_LAUNCH(256, 2); else C8_LAUNCH(256, 0);
  } else {
    if (stats == 1) C8_LAUNCH(128, 1); else if (stats == 2) C8_LAUNCH(128, 2); else C8_LAUNCH(128, 0);
  }
#undef C8_LAUNCH
  RVIT_CHECK_LAUNCH("conv3x3_p8");
  if (stats) *a.stat_slots = (int)mtiles;
  return RVIT_OK;
}
