// bf16 MFMA GEMM for the large Linears (ViT-B / ViT-L scale: every nn.Linear of the encoder, torch functional.py:5819-5833,
// transformer.py:980-982, and their input gradients): 256x256 output tile, eight waves in two groups that run half a phase
// apart ("ping-pong"), the K loop cut into four phases per 64-deep K tile.
//
//   C[M,N] = epi( A[M,K] . B[N,K]^T )      same operands, epilogue and accumulation order as gemm.hip (bit-identical)
//
// Why another kernel: gemm.hip's 256x256 tile keeps ONE 64 KB stage in flight while the other is multiplied, behind one
// workgroup barrier per K tile; on the ViT-B shapes it reaches 0.8-0.94 PFLOP/s.  Here
//   * a wave owns 128x64 of the tile (8 x 4 accumulator fragments = 128 registers), so one K tile is 64 MFMAs per wave
//     fed by 24 fragment reads (0.375 reads per MFMA against 0.5 / 0.75);
//   * the operand tile is staged as four 16 KB HALF tiles (A rows / B columns interleaved so that each half tile is read
//     by all waves in exactly one phase, below), one half tile per phase, and a counted vmcnt keeps three of them in
//     flight across the barriers: the DMA queue never drains inside the loop;
//   * waves 4-7 run one barrier behind waves 0-3: while one wave of a SIMD multiplies (16 MFMAs = one 64x32 quadrant of
//     its tile over the whole K tile), its partner reads fragments and issues DMA.
//
// Half tiles (128 rows x 128 B each; d = K-tile parity):
//   A half h: tile rows  wr*128 + h*64 + [0,64)  for wr = 0,1   -> half-tile row wr*64 + r
//   B half h: tile cols  wc*64  + h*32 + [0,32)  for wc = 0..3  -> half-tile row wc*32 + r
// so wave (wr, wc) reads rows [wr*64, +64) of an A half and rows [wc*32, +32) of a B half, and the quadrant order
// (A0,B0) (A0,B1) (A1,B1) (A1,B0) needs ONE new operand block per phase:
//   phase 1: read B0 (4 x ds_read_b128, retired before the barrier), A0 (8)   stage A1 of tile t+1   multiply (A0,B0)
//   phase 2: read B1 (4)                                                      stage B0 of tile t+2   multiply (A0,B1)
//   phase 3: read A1 (8)                                                      stage A0 of tile t+2   multiply (A1,B1)
//   phase 4: -                                    wait vmcnt(6)               stage B1 of tile t+2   multiply (A1,B0)
// Hazards.  RAW: a half tile is read one phase or more after the counted wait that retired it (phase 4's wait covers all
// of tile t+1; both groups pass a barrier between their wait and the other group's first read).  WAR: a half tile is
// re-staged two phases after its last read, or one phase after when those reads were retired before the reading phase's
// first barrier (B0).  LDS rows are 128 B with the 16-byte chunk index XORed by (row >> 1) & 7 on the DMA source address
// and on the fragment reads (conflict-free ds_read_b128, as in gemm.hip).
#include "common.h"
#include "rvit_internal.h"
#include "gemm_epilogue.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

constexpr int P8_HALF = 128 * 128;    // bytes of one half tile
constexpr int P8_BUF = 4 * P8_HALF;   // A0 | A1 | B0 | B1 of one K-tile parity
constexpr int P8_LDS = 2 * P8_BUF;    // 128 KB
constexpr int P8_HA0 = 0, P8_HB0 = 2;   // half-tile slots: A0, A1, B0, B1

__device__ __forceinline__ void p8_glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

typedef unsigned p8_u32x4 __attribute__((ext_vector_type(4)));
// nt (wave-uniform): nontemporal store.  The output of a wide Linear is far bigger than the L2 and its next reader is
// another kernel; kept out of the L2, the stores delay the next workgroup's first operand loads less (ViT-B in_proj /
// linear1, ViT-L linear1: +4-5 %).  Outputs that fit the 256 MB memory-side cache lose 4 % instead: the host decides.
__device__ __forceinline__ void p8_store16(void* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, int nt) {
  const p8_u32x4 v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<p8_u32x4*>(dst));
  else *reinterpret_cast<p8_u32x4*>(dst) = v;
}

#define P8_SB() __builtin_amdgcn_sched_barrier(0)
#define P8_BAR() __builtin_amdgcn_s_barrier()

// Epilogue forms.  MODE < 0: anything (runtime flags).  MODE >= 0: a "lean" form, a bit mask of P8F_* fixed at compile
// time, for operands that allow 16-byte accesses and use no per-column scale, gate, dropout, row remap or periodic
// residual: the forms of the encoder's Linears in inference and of the plain input-gradient products.  The generic
// epilogue issues ~170 vector instructions per 16-row fragment row and wave whatever the flags say (measured: 1,350
// cycles per fragment row, 5.4 us per tile before a single byte is stored); a lean one issues what its form needs.
constexpr int P8F_BIAS = 1, P8F_GELU = 2, P8F_C2 = 4, P8F_R32 = 8, P8F_CF32 = 16, P8F_DROP_PRE = 32, P8F_DROP_POST = 64,
              P8F_GGELU = 128;   // dropout before the residual / after the activation; gate = gelu'(saved pre-activation)

#ifdef P8_STAMPS
#define P8_STAMP(i) do { if (threadIdx.x == 0) { stamps[blockIdx.x * 16 + (i)] = wall_clock64(); stamps[blockIdx.x * 16 + 8 + (i)] = __builtin_readcyclecounter(); } } while (0)
template <int MODE>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs p, const int nt_store, unsigned long long* stamps) {
  P8_STAMP(0);
  if (threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stamps[blockIdx.x * 16 + 4] = ((unsigned long long)xcc << 32) | hwid;
  }
#else
#define P8_STAMP(i)
template <int MODE>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs p, const int nt_store) {
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware tile id (blocks b and b+8 share an XCD): each XCD owns a contiguous run of tiles; bijective for any grid
  const int ntn = (p.N + 255) >> 8;
  int tid_lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    tid_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tid_lin / ntn, tile_n = tid_lin - tile_m * ntn;
  const long bm0 = (long)tile_m * 256;
  const int bn0 = tile_n * 256;

  // ---- DMA sources: thread (wave, lane) fills half-tile rows j*64 + wave*8 + (lane >> 3), j = 0,1, chunk slot lane & 7
  const int srow = wave * 8 + (lane >> 3);
  const int lc = (lane & 7) ^ ((srow >> 1) & 7);   // source chunk that lands in this slot
  const bf16* pa[2][2];
  const bf16* pb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      long gm = bm0 + j * 128 + h * 64 + srow;           // rows past M re-read the last row: never stored
      if (gm > p.M - 1) gm = p.M - 1;
      pa[h][j] = p.A + gm * p.lda + lc * 8;
      int gn = bn0 + j * 128 + (wave >> 2) * 64 + h * 32 + (wave & 3) * 8 + (lane >> 3);
      if (gn > p.N - 1) gn = p.N - 1;
      pb[h][j] = p.B + (long)gn * p.ldb + lc * 8;
    }
  char* const wave_dst = smem + wave * 1024;

#define P8_STAGE_A(h, d, koff)                                                           \
  do {                                                                                   \
    char* dst_ = wave_dst + (d) * P8_BUF + (P8_HA0 + (h)) * P8_HALF;                     \
    p8_glds16(pa[h][0] + (koff), dst_);                                                  \
    p8_glds16(pa[h][1] + (koff), dst_ + 8192);                                           \
  } while (0)
#define P8_STAGE_B(h, d, koff)                                                           \
  do {                                                                                   \
    char* dst_ = wave_dst + (d) * P8_BUF + (P8_HB0 + (h)) * P8_HALF;                     \
    p8_glds16(pb[h][0] + (koff), dst_);                                                  \
    p8_glds16(pb[h][1] + (koff), dst_ + 8192);                                           \
  } while (0)

  // ---- fragment read addresses ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int c0 = (fq ^ sw) << 4;               // k-step 0 chunk; k-step 1 is chunk ^ 4 = byte offset ^ 64
  const int x1 = (c0 ^ 64) - c0;
  const char* const a_rd = smem + (wr * 64 + fr) * 128 + c0;
  const char* const b_rd = smem + 2 * P8_HALF + (wc * 32 + fr) * 128 + c0;

  bf16x8 af[4][2], bq[2][2][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define P8_READ_A(mh, d)                                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                           \
    af[i_][0] = *reinterpret_cast<const bf16x8*>(a_rd + (d) * P8_BUF + (mh) * P8_HALF + i_ * 2048);            \
    af[i_][1] = *reinterpret_cast<const bf16x8*>((a_rd + (d) * P8_BUF + (mh) * P8_HALF + i_ * 2048) + x1); \
  }
#define P8_READ_B(nh, d)                                                                                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                           \
    bq[nh][j_][0] = *reinterpret_cast<const bf16x8*>(b_rd + (d) * P8_BUF + (nh) * P8_HALF + j_ * 2048);        \
    bq[nh][j_][1] = *reinterpret_cast<const bf16x8*>((b_rd + (d) * P8_BUF + (nh) * P8_HALF + j_ * 2048) + x1); \
  }
#define P8_MMA(mh, nh)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                           \
      acc[(mh) * 4 + i_][(nh) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                             \
          bq[nh][j_][ks_], af[i_][ks_], acc[(mh) * 4 + i_][(nh) * 2 + j_], 0, 0, 0);                           \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)

  const int nk = p.K >> 6;
  const int klast = (nk - 1) * 64;
  // one K tile out of buffer d; k1 / k2 = element offsets of tiles t+1 / t+2 (clamped to the last tile: a re-load of
  // data nobody reads keeps the counted waits fixed)
#define P8_TILE(d, k1, k2)                                                                                     \
  do {                                                                                                         \
    /* phase 1 */                                                                                              \
    P8_READ_B(0, d);                                                                                           \
    P8_SB();                                                                                                   \
    P8_READ_A(0, d);                                                                                           \
    P8_STAGE_A(1, (d) ^ 1, k1);                                                                             \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                         \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    P8_SB();                                                                                                   \
    P8_MMA(0, 0);                                                                                              \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    P8_SB();                                                                                                   \
    /* phase 2 */                                                                                              \
    P8_READ_B(1, d);                                                                                           \
    P8_STAGE_B(0, d, k2);                                                                                   \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    P8_SB();                                                                                                   \
    P8_MMA(0, 1);                                                                                              \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    P8_SB();                                                                                                   \
    /* phase 3 */                                                                                              \
    P8_READ_A(1, d);                                                                                           \
    P8_STAGE_A(0, d, k2);                                                                                   \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    P8_SB();                                                                                                   \
    P8_MMA(1, 1);                                                                                              \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    P8_SB();                                                                                                   \
    /* phase 4 */                                                                                              \
    P8_STAGE_B(1, d, k2);                                                                                   \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                           \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    P8_SB();                                                                                                   \
    P8_MMA(1, 0);                                                                                              \
    P8_SB();                                                                                                   \
    P8_BAR();                                                                                                  \
    P8_SB();                                                                                                   \
  } while (0)

  // ---- prologue: all of tile 0, three half tiles of tile 1 ----
  {
    const int k1 = nk > 1 ? 64 : 0;
    P8_STAGE_B(0, 0, 0);
    P8_STAGE_A(0, 0, 0);
    P8_STAGE_B(1, 0, 0);
    P8_STAGE_A(1, 0, 0);
    P8_STAGE_B(0, 1, k1);
    P8_STAGE_A(0, 1, k1);
    P8_STAGE_B(1, 1, k1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    P8_SB();
    P8_BAR();
    P8_SB();
  }
  P8_STAMP(1);
  if (wr == 1) P8_BAR();   // waves 4-7 run one barrier behind
  P8_SB();

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    const int ka = min((kt + 1) * 64, klast), kb = min((kt + 2) * 64, klast), kc = min((kt + 3) * 64, klast);
    P8_TILE(0, ka, kb);
    P8_TILE(1, kb, kc);
  }
  if (kt < nk) {
    P8_TILE(0, klast, klast);
  }
  if (wr == 0) P8_BAR();   // rejoin
  P8_SB();
  P8_STAMP(2);
#undef P8_TILE
#undef P8_MMA
#undef P8_READ_A
#undef P8_READ_B
#undef P8_STAGE_A
#undef P8_STAGE_B

  // ---- epilogue ----
  // Vector form (N, leading dimensions and pointers allow 16-byte accesses: every Linear of the models): in registers.
  // A lane holds 4 consecutive columns of row fr for each of the four 16-column blocks of a fragment row;
  // v_permlane16_swap between lane rows (fq 0<->1, 2<->3) of blocks 2pr and 2pr+1 leaves every lane with 8 consecutive
  // columns, so a store instruction writes 16 rows x 64 contiguous bytes.  No LDS round trip, no barrier, and no wait
  // that covers a store: the column operands are read once per tile, the row operands (residual, gate) of fragment row
  // i+1 are requested before the stores of fragment row i are issued (vmcnt counts loads and stores in issue order; a
  // load waited for right after the previous row's stores costs their ~1 us round trip each time: measured 9.3 us of
  // epilogue per tile in the first version of this kernel, 30 us per ViT-B fc1 tile in all).
  // A runtime loop over the eight fragment rows (unrolled, the epilogue body would be emitted sixteen times); the row
  // is picked by wave-uniform compares so that `acc` is only ever indexed by constants (registers).
  const int co_lane = (fq & 1) * 16 + (fq >> 1) * 8;
  const long m0 = bm0 + wr * 128 + fr;
  const long mlast = p.M - 1;
  if constexpr (MODE >= 0) {
    // lean form: fully unrolled, `acc` indexed by constants; the residual of fragment row i+1 is requested before the
    // stores of fragment row i
    bool col_ok[2];
    int nl[2];
    float bi[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int n = bn0 + wc * 64 + pr * 32 + co_lane;
      col_ok[pr] = n < p.N;
      nl[pr] = col_ok[pr] ? n : 0;
      if constexpr ((MODE & P8F_BIAS) != 0) {
        const float4 x = *reinterpret_cast<const float4*>(p.bias + nl[pr]);
        const float4 y = *reinterpret_cast<const float4*>(p.bias + nl[pr] + 4);
        bi[pr][0] = x.x; bi[pr][1] = x.y; bi[pr][2] = x.z; bi[pr][3] = x.w;
        bi[pr][4] = y.x; bi[pr][5] = y.y; bi[pr][6] = y.z; bi[pr][7] = y.w;
      }
    }
    P8_STAMP(5);
    float4 rn[2][2];
    uint4 gn[2];
    if constexpr ((MODE & (P8F_R32 | P8F_GGELU)) != 0) {
      const long mr = m0 < mlast ? m0 : mlast;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        if constexpr ((MODE & P8F_R32) != 0) {
          const float* rp = (const float*)p.R + mr * p.ldr + nl[pr];
          rn[pr][0] = *reinterpret_cast<const float4*>(rp);
          rn[pr][1] = *reinterpret_cast<const float4*>(rp + 4);
        }
        if constexpr ((MODE & P8F_GGELU) != 0) gn[pr] = *reinterpret_cast<const uint4*>(p.G + mr * p.ldg + nl[pr]);
      }
    }
#pragma unroll
    for (int i0 = 0; i0 < 8; ++i0) {
      if (i0 == 1) P8_STAMP(6);
      const long m = m0 + i0 * 16;
      float4 rc[2][2];
      uint4 gc[2];
      if constexpr ((MODE & (P8F_R32 | P8F_GGELU)) != 0) {
        const long m1 = m0 + (i0 < 7 ? i0 + 1 : 7) * 16;   // rows past the matrix re-read its last row: never used
        const long mr = m1 < mlast ? m1 : mlast;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          if constexpr ((MODE & P8F_R32) != 0) {
            rc[pr][0] = rn[pr][0];
            rc[pr][1] = rn[pr][1];
            if (i0 < 7) {
              const float* rp = (const float*)p.R + mr * p.ldr + nl[pr];
              rn[pr][0] = *reinterpret_cast<const float4*>(rp);
              rn[pr][1] = *reinterpret_cast<const float4*>(rp + 4);
            }
          }
          if constexpr ((MODE & P8F_GGELU) != 0) {
            gc[pr] = gn[pr];
            if (i0 < 7) gn[pr] = *reinterpret_cast<const uint4*>(p.G + mr * p.ldg + nl[pr]);
          }
        }
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
        const bool ok = col_ok[pr] && m < p.M;
        if constexpr ((MODE & P8F_BIAS) != 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bi[pr][e];
        }
        if constexpr ((MODE & P8F_GGELU) != 0) {
          const uint4 gq = gc[pr];
          const float gv[8] = {bf16_lo(gq.x), bf16_hi(gq.x), bf16_lo(gq.y), bf16_hi(gq.y),
                               bf16_lo(gq.z), bf16_hi(gq.z), bf16_lo(gq.w), bf16_hi(gq.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(gv[e]);
        }
        if constexpr ((MODE & P8F_C2) != 0) {
          if (ok) {
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(p.C2 + m * p.ldc2 + nl[pr]) = o;
          }
        }
        if constexpr ((MODE & P8F_DROP_PRE) != 0) drop_mult8(p.drop, (uint32_t)(m * p.N + nl[pr]), v);
        if constexpr ((MODE & P8F_R32) != 0) {
          v[0] += rc[pr][0].x; v[1] += rc[pr][0].y; v[2] += rc[pr][0].z; v[3] += rc[pr][0].w;
          v[4] += rc[pr][1].x; v[5] += rc[pr][1].y; v[6] += rc[pr][1].z; v[7] += rc[pr][1].w;
        }
        if constexpr ((MODE & P8F_GELU) != 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if constexpr ((MODE & P8F_DROP_POST) != 0) drop_mult8(p.drop, (uint32_t)(m * p.N + nl[pr]), v);
        if (ok) {
          // nontemporal stores (P8_NT): the output of a large Linear is far bigger than the L2 and its next reader is
          // another kernel; kept out of the L2, the stores no longer delay the next workgroup's first operand loads
          if constexpr ((MODE & P8F_CF32) != 0) {
            float* cp = (float*)p.C + m * p.ldc + nl[pr];
            p8_store16(cp, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]), nt_store);
            p8_store16(cp + 4, __float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]), nt_store);
          } else {
            p8_store16((bf16*)p.C + m * p.ldc + nl[pr], pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                              pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]), nt_store);
          }
        }
      }
    }
    P8_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the (unused) tiles past K must land before the LDS is released
  } else if (p.vec) {
    // generic vector form: a runtime loop over the eight fragment rows (unrolled, the epilogue body would be emitted
    // sixteen times); the row is picked by wave-uniform compares so that `acc` is only ever indexed by constants
    // (one 32-column half of the wave's tile at a time: both halves' operands at once do not fit the register file)
#pragma unroll 1
    for (int pr = 0; pr < 2; ++pr) {
      const int n = bn0 + wc * 64 + pr * 32 + co_lane;
      const bool col_ok = n < p.N;
      const int nl = col_ok ? n : 0;
      EpiCols cols;
      rvit_epi_cols_load(p, nl, cols);
      EpiRow nxt;
      rvit_epi_row_load(p, m0 < mlast ? m0 : mlast, nl, nxt);
#pragma unroll 1
      for (int i0 = 0; i0 < 8; ++i0) {
        f32x4 cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = acc[0][j];
#pragma unroll
        for (int k = 1; k < 8; ++k)
          if (i0 == k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = acc[k][j];
          }
        const long m = m0 + i0 * 16;
        const long m1 = m0 + (i0 < 7 ? i0 + 1 : 7) * 16;   // rows past the matrix re-read its last row: never used
        const EpiRow rowv = nxt;
        rvit_epi_row_load(p, m1 < mlast ? m1 : mlast, nl, nxt);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // odd lane rows of the even block <-> even lane rows of the odd block (pr is wave-uniform)
          const float x = pr ? cur[2][e] : cur[0][e], y = pr ? cur[3][e] : cur[1][e];
          const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
          v[e] = __uint_as_float(sw2[0]);
          v[4 + e] = __uint_as_float(sw2[1]);
        }
        rvit_epi_row_apply(p, v, rowv, cols, nl, col_ok && m < p.M);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the (unused) tiles past K must land before the LDS is released
  } else {
    // scalar form (odd N or unaligned operands): accumulators through a wave-private LDS region, 16 rows per pass
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int EP_PITCH = 64 * 4 + 16;
    char* ep = smem + wave * (16 * EP_PITCH);
    const int er = lane >> 3, ec = (lane & 7) * 8;
    const int n = bn0 + wc * 64 + ec;
#pragma unroll 1
    for (int i0 = 0; i0 < 8; ++i0) {
      f32x4 cur[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = acc[0][j];
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if (i0 == k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) cur[j] = acc[k][j];
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(ep + fr * EP_PITCH + (j * 16 + fq * 4) * 4) = cur[j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 8 + er;
        const long m = bm0 + wr * 128 + i0 * 16 + row;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        {
          const float4 lo = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4);
          const float4 hi = *reinterpret_cast<const float4*>(ep + row * EP_PITCH + ec * 4 + 16);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        rvit_gemm_epilogue_row8(p, v, m, n);
      }
    }
  }
  P8_STAMP(3);
}

}  // namespace

bool rvit_gemm8p_ok(const GemmArgs& a) {
  return a.mode == 0 && a.a_row_group == 0 && (a.K & 63) == 0 && a.K >= 128 && a.M >= 1 && a.N >= 1;
}

int rvit_gemm8p(const GemmArgs& a, hipStream_t stream) {
  if (!rvit_gemm8p_ok(a)) {
    rvit_set_error("gemm: the 8-phase 256x256 kernel needs a plain A operand and K %% 64 == 0, K >= 128 (K=%d)", a.K);
    return RVIT_EUNSUPPORTED;
  }
  const long tiles = ((a.M + 255) / 256) * (long)((a.N + 255) / 256);
  if (tiles <= 0) return RVIT_OK;
  if (tiles > 0x7fffffffL) {
    rvit_set_error("gemm: too many tiles (%ld)", tiles);
    return RVIT_EINVAL;
  }
  const int nt_store = (double)a.M * a.N * (a.c_f32 ? 4 : 2) > 200e6;
  // lean epilogue forms (see the kernel): everything a form does not name must be absent
  int mode = -1;
  if (a.vec && !a.scale && (!a.gate || a.gate == RVIT_GATE_GELU_GRAD) && a.row_group == 0 && a.r_period == 0 &&
      (a.act == RVIT_ACT_NONE || a.act == RVIT_ACT_GELU) && (!a.R || a.r_f32)) {
    mode = (a.bias ? P8F_BIAS : 0) | (a.act == RVIT_ACT_GELU ? P8F_GELU : 0) | (a.C2 ? P8F_C2 : 0) | (a.R ? P8F_R32 : 0) |
           (a.c_f32 ? P8F_CF32 : 0) | (a.drop_mode == RVIT_DROP_PRE_RESIDUAL ? P8F_DROP_PRE : 0) |
           (a.drop_mode == RVIT_DROP_POST_ACT ? P8F_DROP_POST : 0) | (a.gate ? P8F_GGELU : 0);
  }
#ifdef P8_STAMPS
  // developer build: per-workgroup wall-clock / cycle stamps (start, prologue done, K loop done, end, inside the
  // epilogue) + hardware id, dumped to the file RVIT_G8_STAMPS names after a stream synchronisation
  static unsigned long long* d_stamps = nullptr;
  if (!d_stamps) (void)hipMalloc(&d_stamps, 16 * 8 * 32768);
  (void)hipMemsetAsync(d_stamps, 0, 16 * 8 * 32768, stream);
#define P8_LAUNCH(M_) hipLaunchKernelGGL(gemm8p_kernel<M_>, dim3((unsigned)tiles), dim3(512), P8_LDS, stream, a, nt_store, d_stamps)
#else
#define P8_LAUNCH(M_) hipLaunchKernelGGL(gemm8p_kernel<M_>, dim3((unsigned)tiles), dim3(512), P8_LDS, stream, a, nt_store)
#endif
#define P8_CASE(M_)                                                                          \
  case M_: {                                                                                 \
    static RvitLdsLimit lim;                                                                 \
    if (const int rc = lim.ensure({(const void*)gemm8p_kernel<M_>}, P8_LDS, "gemm8p")) return rc; \
    P8_LAUNCH(M_);                                                                           \
    break;                                                                                   \
  }
  switch (mode) {
    P8_CASE(0)                                        // plain product (input gradients), bf16 out
    P8_CASE(P8F_CF32)                                 // plain product, f32 out
    P8_CASE(P8F_BIAS)                                 // in_proj
    P8_CASE(P8F_BIAS | P8F_GELU)                      // linear1 (inference)
    P8_CASE(P8F_BIAS | P8F_GELU | P8F_C2)             // linear1 (training, no dropout)
    P8_CASE(P8F_BIAS | P8F_R32 | P8F_CF32)            // out_proj / linear2: + fp32 residual stream
    P8_CASE(P8F_BIAS | P8F_R32 | P8F_CF32 | P8F_DROP_PRE)        // ... training, dropout1 / dropout2
    P8_CASE(P8F_BIAS | P8F_GELU | P8F_C2 | P8F_DROP_POST)        // linear1, training with dropout
    P8_CASE(P8F_GGELU)                                           // input gradient through GELU (linear2^T)
    P8_CASE(P8F_GGELU | P8F_DROP_POST)                           // ... with the dropout of the hidden map
    default: {
      static RvitLdsLimit lim;
      if (const int rc = lim.ensure({(const void*)gemm8p_kernel<-1>}, P8_LDS, "gemm8p")) return rc;
      P8_LAUNCH(-1);
    }
  }
#undef P8_CASE
#undef P8_LAUNCH
#ifdef P8_STAMPS
  if (const char* path = getenv("RVIT_G8_STAMPS")) {
    (void)hipStreamSynchronize(stream);
    static unsigned long long h[16 * 32768];
    const long nt = tiles < 32768 ? tiles : 32768;
    (void)hipMemcpy(h, d_stamps, nt * 128, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(path, "wb")) { fwrite(h, 128, nt, f); fclose(f); }
  }
#endif
  RVIT_CHECK_LAUNCH("gemm8p");
  return RVIT_OK;
}
