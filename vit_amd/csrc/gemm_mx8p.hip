// MX-fp8 GEMM for the large Linears of the fp8 inference mode (BASELINE.json configs[4]: ViT-L/16-scale inference with MX
// fp8 weights + activations; torch functional.py:5819-5833, transformer.py:980-982): the eight-wave 256x256 kernel of
// gemm8p.hip on the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4.
//
//   C[M,N] = act( dequant(A_q) . dequant(B_q)^T + bias ) + R      (arguments and arithmetic of gemm_mx.hip: same K order,
//                                                                   one instruction per 128-deep K tile: identical bits)
//
// A K tile is 128 fp8 = 128 BYTES per row: the LDS image, the DMA stream, the half tiles, the four phases per K tile, the
// counted vmcnt and the two wave groups one barrier apart are gemm8p.hip's, byte for byte (see its header); a fragment is
// both 16-byte halves of a 128-byte row (the two ds_read_b128 that feed the bf16 kernel's two k-steps), and a phase issues
// 8 instructions of 32 cycles instead of 16 of 16: the same matrix-pipe time per K tile at twice the FLOPs.
// E8M0 scales (tiled [rows/16][K/512][16][4][4] bytes, gemm_mx.hip): the 16 + 16 scale blocks (256 B each) that a
// workgroup needs for FOUR K tiles are one 16-byte LDS-DMA per thread, issued one group ahead into a double buffer
// behind the operand ring, in phase 1 (before that phase's operand stage, so the six newest DMA at phase 4's wait are
// still the operand stages); a wave reads its 8 + 4 scale dwords from LDS at each K tile.  No ordinary global load is
// left in the K loop (beside LDS-DMA hipcc waits vmcnt(0) for any of those: cdna_hip_programming.md section 5).
// Epilogue in registers (v_permlane16_swap: 8 consecutive columns per lane), forms fixed at compile time; the MX copy of
// the output takes the block maximum over the four lanes that hold a 32-column block (lane rows fq = 0..3).
#include <stdlib.h>
#include "common.h"
#include "rvit_internal.h"
#include "mx_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int X8_HALF = 128 * 128;      // bytes of one half tile
constexpr int X8_BUF = 4 * X8_HALF;     // A0 | A1 | B0 | B1 of one K-tile parity
constexpr int X8_RING = 2 * X8_BUF;     // 128 KB
constexpr int X8_SCALES = 8192;         // 16 A blocks + 16 B blocks of 256 B: the scales of four K tiles
constexpr int X8_BIAS = X8_RING + 2 * X8_SCALES;   // the tile's 256 bias values
constexpr int X8_LDS = X8_BIAS + 1024;
constexpr int X8_HA0 = 0, X8_HB0 = 2;

// epilogue forms (bit masks fixed at compile time)
constexpr int X8F_GELU = 1, X8F_R32 = 2, X8F_CF32 = 4, X8F_CQ = 8, X8F_C = 16;

__device__ __forceinline__ void x8_glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

#define X8_SB() __builtin_amdgcn_sched_barrier(0)
#define X8_BAR() __builtin_amdgcn_s_barrier()

template <int MODE>
__global__ __launch_bounds__(512) void gemm_mx8p_kernel(const MxGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;

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

  // ---- DMA sources (operands): as gemm8p.hip, 16-byte chunks of 128-byte rows ----
  const int srow = wave * 8 + (lane >> 3);
  const int lc = (lane & 7) ^ ((srow >> 1) & 7);
  const uint8_t* pa[2][2];
  const uint8_t* pb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      long gm = bm0 + j * 128 + h * 64 + srow;           // rows past M re-read the last row: never stored
      if (gm > p.M - 1) gm = p.M - 1;
      pa[h][j] = p.A + gm * p.lda + lc * 16;
      int gn = bn0 + j * 128 + (wave >> 2) * 64 + h * 32 + (wave & 3) * 8 + (lane >> 3);
      if (gn > p.N - 1) gn = p.N - 1;
      pb[h][j] = p.B + (long)gn * p.ldb + lc * 16;
    }
  char* const wave_dst = smem + wave * 1024;
  // ---- DMA source (scales): waves 0-3 fetch the tile's 16 A blocks, waves 4-7 its 16 B blocks; a wave instruction is
  // four 256-byte blocks, lane = (block in the wave, 16-byte piece).  Blocks past the matrix are clamped to its last one.
  const int nk = p.K >> 7, nk4 = (nk + 3) >> 2;
  const uint8_t* ps;
  {
    const int blk = (wave & 3) * 4 + (lane >> 4);
    if (wave < 4) {
      long mb = (bm0 >> 4) + blk;
      const long last = (p.M + 15) / 16 - 1;
      if (mb > last) mb = last;
      ps = p.SA + mb * nk4 * 256 + (lane & 15) * 16;
    } else {
      long nb = (bn0 >> 4) + blk;
      const long last = (p.N + 15) / 16 - 1;
      if (nb > last) nb = last;
      ps = p.SB + nb * nk4 * 256 + (lane & 15) * 16;
    }
  }
#define X8_STAGE_SCALES(g_) x8_glds16(ps + (long)(g_) * 256, smem + X8_RING + ((g_) & 1) * X8_SCALES + wave * 1024)

#define X8_STAGE_A(h, d, koff)                                                           \
  do {                                                                                   \
    char* dst_ = wave_dst + (d) * X8_BUF + (X8_HA0 + (h)) * X8_HALF;                     \
    x8_glds16(pa[h][0] + (koff), dst_);                                                  \
    x8_glds16(pa[h][1] + (koff), dst_ + 8192);                                           \
  } while (0)
#define X8_STAGE_B(h, d, koff)                                                           \
  do {                                                                                   \
    char* dst_ = wave_dst + (d) * X8_BUF + (X8_HB0 + (h)) * X8_HALF;                     \
    x8_glds16(pb[h][0] + (koff), dst_);                                                  \
    x8_glds16(pb[h][1] + (koff), dst_ + 8192);                                           \
  } while (0)

  // ---- fragment read addresses ----
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int c0 = (fq ^ sw) << 4;               // k = 16 fq .. 16 fq + 15; k + 64 is chunk ^ 4 = byte offset ^ 64
  // LDS base addresses as opaque registers (buffer 0 / 1 x low / high half of the row x A / B, and the scale bytes):
  // every read below is base + a 16-bit immediate (left alone, the compiler materialises one address register per read)
  typedef __attribute__((address_space(3))) const char lds_cchar;
  lds_cchar* a_rd[2][2];
  lds_cchar* b_rd[2][2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      a_rd[d][hf] = (lds_cchar*)smem + d * X8_BUF + (wr * 64 + fr) * 128 + (c0 ^ (hf * 64));
      b_rd[d][hf] = (lds_cchar*)smem + d * X8_BUF + 2 * X8_HALF + (wc * 32 + fr) * 128 + (c0 ^ (hf * 64));
      asm volatile("" : "+v"(a_rd[d][hf]));
      asm volatile("" : "+v"(b_rd[d][hf]));
    }
  // this lane's scale dword of block b is [b][fr][fq]: byte (kt & 3) of it belongs to K tile kt.  The wave's A blocks
  // are wr*8 .. +7, its B blocks 16 + wc*4 .. +3.
  lds_cchar* sa_rd = (lds_cchar*)smem + X8_RING + (wr * 8) * 256 + (fr * 4 + fq) * 4;
  lds_cchar* sb_rd = (lds_cchar*)smem + X8_RING + (16 + wc * 4) * 256 + (fr * 4 + fq) * 4;
  asm volatile("" : "+v"(sa_rd));
  asm volatile("" : "+v"(sb_rd));

  i32x8 af[4], bq[2][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int sa_s[4], sb_s[4];   // E8M0 bytes: the A blocks of the row half in work, the wave's four B blocks

  typedef unsigned x8_u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const x8_u32x4 lds_cuint4;
  typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
#define X8_FRAG(dst, plo, phi)                                                                                 \
  do {                                                                                                         \
    const x8_u32x4 lo_ = *reinterpret_cast<lds_cuint4*>(plo), hi_ = *reinterpret_cast<lds_cuint4*>(phi);       \
    dst = i32x8{(int)lo_.x, (int)lo_.y, (int)lo_.z, (int)lo_.w, (int)hi_.x, (int)hi_.y, (int)hi_.z, (int)hi_.w}; \
  } while (0)
#define X8_READ_A(mh, d)                                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                             \
    X8_FRAG(af[i_], a_rd[d][0] + (mh) * X8_HALF + i_ * 2048, a_rd[d][1] + (mh) * X8_HALF + i_ * 2048)
#define X8_READ_B(nh, d)                                                                                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                             \
    X8_FRAG(bq[nh][j_], b_rd[d][0] + (nh) * X8_HALF + j_ * 2048, b_rd[d][1] + (nh) * X8_HALF + j_ * 2048)
  // scale bytes of K tile kt_ (group buffer (kt_ >> 2) & 1, byte kt_ & 3 of the lane's dword)
#define X8_READ_SA(mh, kt_)                                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                             \
    sa_s[i_] = *reinterpret_cast<lds_cu8*>(sa_rd + sc_off + ((mh) * 4 + i_) * 256)
#define X8_READ_SB(kt_)                                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) sb_s[j_] = *reinterpret_cast<lds_cu8*>(sb_rd + sc_off + j_ * 256)
#define X8_MMA(mh, nh)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                           \
      acc[(mh) * 4 + i_][(nh) * 2 + j_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                    \
          bq[nh][j_], af[i_], acc[(mh) * 4 + i_][(nh) * 2 + j_], 0, 0, 0, sb_s[(nh) * 2 + j_], 0, sa_s[i_]); \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)

  const int klast = (nk - 1) * 128;
  // one K tile (index kt_) out of buffer d; k1 / k2 = byte offsets of K tiles kt_+1 / kt_+2 (clamped to the last one)
#define X8_TILE(d, kt_, k1, k2)                                                                                \
  do {                                                                                                         \
    /* phase 1 */                                                                                              \
    const int sc_off = (((kt_) >> 2) & 1) * X8_SCALES + ((kt_) & 3);                                           \
    X8_READ_SB(kt_);                                                                                           \
    X8_READ_SA(0, kt_);                                                                                        \
    X8_READ_B(0, d);                                                                                           \
    X8_SB();                                                                                                   \
    X8_READ_A(0, d);                                                                                           \
    if (((kt_) & 3) == 0 && (kt_) + 4 < nk) X8_STAGE_SCALES(((kt_) >> 2) + 1);                                 \
    X8_STAGE_A(1, (d) ^ 1, k1);                                                                                \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                         \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    X8_SB();                                                                                                   \
    X8_MMA(0, 0);                                                                                              \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    X8_SB();                                                                                                   \
    /* phase 2 */                                                                                              \
    X8_READ_B(1, d);                                                                                           \
    X8_STAGE_B(0, d, k2);                                                                                      \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    X8_SB();                                                                                                   \
    X8_MMA(0, 1);                                                                                              \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    X8_SB();                                                                                                   \
    /* phase 3 */                                                                                              \
    X8_READ_SA(1, kt_);                                                                                        \
    X8_READ_A(1, d);                                                                                           \
    X8_STAGE_A(0, d, k2);                                                                                      \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    X8_SB();                                                                                                   \
    X8_MMA(1, 1);                                                                                              \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    X8_SB();                                                                                                   \
    /* phase 4 */                                                                                              \
    X8_STAGE_B(1, d, k2);                                                                                      \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                           \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    X8_SB();                                                                                                   \
    X8_MMA(1, 0);                                                                                              \
    X8_SB();                                                                                                   \
    X8_BAR();                                                                                                  \
    X8_SB();                                                                                                   \
  } while (0)

  // ---- prologue: the scales of K tiles 0-3, all of K tile 0, three half tiles of K tile 1 ----
  if (wave == 0) {   // the tile's 256 bias values: one DMA, 16 bytes per lane (gemm8p.hip)
    int nb = bn0 + lane * 4;
    if (nb > p.N - 4) nb = p.N - 4;
    x8_glds16(p.bias + nb, smem + X8_BIAS);
  }
  {
    const int k1 = nk > 1 ? 128 : 0;
    X8_STAGE_SCALES(0);
    X8_STAGE_B(0, 0, 0);
    X8_STAGE_A(0, 0, 0);
    X8_STAGE_B(1, 0, 0);
    X8_STAGE_A(1, 0, 0);
    X8_STAGE_B(0, 1, k1);
    X8_STAGE_A(0, 1, k1);
    X8_STAGE_B(1, 1, k1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    X8_SB();
    X8_BAR();
    X8_SB();
  }
  if (wr == 1) X8_BAR();   // waves 4-7 run one barrier behind
  X8_SB();

  // phase 1's scale reads are 8 more LDS operations in front of the B reads: its lgkmcnt(8) still retires the B reads
  // (LDS operations return in order), which is all the re-staging of B0 one phase later needs
  for (int kt = 0; kt < nk; kt += 2) {   // nk is even (dispatcher)
    const int ka = min((kt + 1) * 128, klast), kb = min((kt + 2) * 128, klast), kc = min((kt + 3) * 128, klast);
    X8_TILE(0, kt, ka, kb);
    X8_TILE(1, kt + 1, kb, kc);
  }
  if (wr == 0) X8_BAR();   // rejoin
  X8_SB();
#undef X8_TILE
#undef X8_MMA
#undef X8_READ_A
#undef X8_READ_B
#undef X8_FRAG
#undef X8_READ_SA
#undef X8_READ_SB
#undef X8_STAGE_A
#undef X8_STAGE_B
#undef X8_STAGE_SCALES

  // ---- epilogue (registers; gemm8p.hip) ----
  const int co_lane = (fq & 1) * 16 + (fq >> 1) * 8;
  const long m0 = bm0 + wr * 128 + fr;
  const long mlast = p.M - 1;
  bool col_ok[2];
  int nl[2];
  float bi[2][8];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const int n = bn0 + wc * 64 + pr * 32 + co_lane;
    col_ok[pr] = n < p.N;
    nl[pr] = col_ok[pr] ? n : 0;
    const int nloc = wc * 64 + pr * 32 + co_lane;   // staged by the prologue, behind its wait and barrier
    const float4 x = *reinterpret_cast<const float4*>(smem + X8_BIAS + nloc * 4);
    const float4 y = *reinterpret_cast<const float4*>(smem + X8_BIAS + nloc * 4 + 16);
    bi[pr][0] = x.x; bi[pr][1] = x.y; bi[pr][2] = x.z; bi[pr][3] = x.w;
    bi[pr][4] = y.x; bi[pr][5] = y.y; bi[pr][6] = y.z; bi[pr][7] = y.w;
  }
  float4 rn[2][2];
  if constexpr ((MODE & X8F_R32) != 0) {
    const long mr = m0 < mlast ? m0 : mlast;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float* rp = (const float*)p.R + mr * p.ldr + nl[pr];
      rn[pr][0] = *reinterpret_cast<const float4*>(rp);
      rn[pr][1] = *reinterpret_cast<const float4*>(rp + 4);
    }
  }
#pragma unroll
  for (int i0 = 0; i0 < 8; ++i0) {
    const long m = m0 + i0 * 16;
    float4 rc[2][2];
    if constexpr ((MODE & X8F_R32) != 0) {
      const long m1 = m0 + (i0 < 7 ? i0 + 1 : 7) * 16;   // rows past the matrix re-read its last row: never used
      const long mr = m1 < mlast ? m1 : mlast;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        rc[pr][0] = rn[pr][0];
        rc[pr][1] = rn[pr][1];
        if (i0 < 7) {
          const float* rp = (const float*)p.R + mr * p.ldr + nl[pr];
          rn[pr][0] = *reinterpret_cast<const float4*>(rp);
          rn[pr][1] = *reinterpret_cast<const float4*>(rp + 4);
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
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bi[pr][e];
      if constexpr ((MODE & X8F_GELU) != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      }
      if constexpr ((MODE & X8F_R32) != 0) {
        v[0] += rc[pr][0].x; v[1] += rc[pr][0].y; v[2] += rc[pr][0].z; v[3] += rc[pr][0].w;
        v[4] += rc[pr][1].x; v[5] += rc[pr][1].y; v[6] += rc[pr][1].z; v[7] += rc[pr][1].w;
      }
      if constexpr ((MODE & X8F_C) != 0) {
        if (ok) {
          if constexpr ((MODE & X8F_CF32) != 0) {
            float* cp = (float*)p.C + m * p.ldc + nl[pr];
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>((bf16*)p.C + m * p.ldc + nl[pr]) = o;
          }
        }
      }
      if constexpr ((MODE & X8F_CQ) != 0) {
        // MX copy: the 32-column block wc*64 + pr*32 .. +31 of row m is held by the four lanes (fr, fq = 0..3)
        if (!ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int s = mx_scale_e8m0(amax);
        const float inv = mx_inv_scale(s);
        if (ok) {
          uint2 q;
          q.x = mx_pack4(v[0], v[1], v[2], v[3], inv);
          q.y = mx_pack4(v[4], v[5], v[6], v[7], inv);
          *reinterpret_cast<uint2*>(p.Cq + m * p.ldcq + nl[pr]) = q;
          if (fq == 0) p.SCq[mx_scale_offset(m, nl[pr], p.N)] = (uint8_t)s;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of the (unused) K tiles past the end must land before the LDS is released
}

}  // namespace

// -1: not a case of this kernel (the caller runs gemm_mx.hip's), else the epilogue form
int rvit_gemm_mx8p_mode(const MxGemmArgs& a) {
  if ((a.K & 255) != 0 || !a.bias || a.act == RVIT_ACT_RELU) return -1;   // an even number of K tiles, bias always
  const long t256 = ((a.M + 255) / 256) * (long)((a.N + 255) / 256);
  static const long min_tiles = [] { const char* e = getenv("RVIT_MX8P_MIN_TILES"); return e ? atol(e) : 512L; }();   // A/B runs
  if (t256 < min_tiles) return -1;
  const int gelu = a.act == RVIT_ACT_GELU;
  if (a.C && !a.c_f32 && !a.R && !a.Cq && !gelu) return X8F_C;                                  // in_proj: bf16 out
  if (a.C && a.c_f32 && a.R && a.r_f32 && !a.Cq && !gelu) return X8F_C | X8F_CF32 | X8F_R32;    // out_proj / linear2
  if (!a.C && a.Cq && !a.R && gelu) return X8F_CQ | X8F_GELU;                                   // linear1: MX out
  return -1;
}

int rvit_gemm_mx8p(const MxGemmArgs& a, int mode, hipStream_t st) {
  const long tiles = ((a.M + 255) / 256) * (long)((a.N + 255) / 256);
  if (tiles <= 0) return RVIT_OK;
  RVIT_CHECK_ARG(tiles < 0x7fffffffL, "gemm_mxfp8: too many tiles");
#define X8_CASE(M_)                                                                                  \
  case M_: {                                                                                         \
    static RvitLdsLimit lim;                                                                         \
    if (const int rc = lim.ensure({(const void*)gemm_mx8p_kernel<M_>}, X8_LDS, "gemm_mx8p")) return rc; \
    hipLaunchKernelGGL(gemm_mx8p_kernel<M_>, dim3((unsigned)tiles), dim3(512), X8_LDS, st, a);       \
    break;                                                                                           \
  }
  switch (mode) {
    X8_CASE(X8F_C)
    X8_CASE(X8F_C | X8F_CF32 | X8F_R32)
    X8_CASE(X8F_CQ | X8F_GELU)
    default:
      rvit_set_error("gemm_mxfp8: no eight-wave kernel for epilogue form %d", mode);
      return RVIT_EINVAL;
  }
#undef X8_CASE
  RVIT_CHECK_LAUNCH("gemm_mx8p");
  return RVIT_OK;
}
