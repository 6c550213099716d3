// Stem entry of the resnet feature extractor, fused for inference:
//   conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BatchNorm + ReLU + max-pool 3x3 / stride 2 / pad 1
// (timm resnet18 `conv1`, `bn1`, `act1`, `maxpool` under `self.cnn_backbone(x)[-1]`,
// train_vit.py:87,145-146).  As two kernels the 112x112x64 map costs 1.6 MB of HBM writes plus
// 1.6 MB of reads per sample and is the largest single item of the inference step
// (profiles/r01_rocprof_summary.txt: conv1 662 us + max-pool 237 us of 4.2 ms at B = 512).  Fused,
// the map never leaves the CU: HBM sees the packed image (427 KB/sample) and the pooled map
// (401 KB/sample).
//
// A workgroup produces TP = 2 pooled rows of one image.  They need conv rows 2*py0-1 .. 2*py0+3
// (5 rows, one shared with the neighbouring band: 25 % recompute of a conv that is 6 % of the
// network) which need 15 rows of the zero-haloed NHWC4 image (rvit_input_pack_nhwc4: [B,230,232,4],
// 8 B per pixel) = ONE contiguous 27.8 KB block, DMA'd straight into LDS together with the 29 KB of
// weights.  In NHWC4 the 7 taps x 3 channels of one kernel row are one contiguous 64-byte run
// (8 px x 4 ch), so an MFMA B-fragment of output pixel c, kernel row kh is the 16 bytes at
//   patch[(2*r + kh) * 232 + 2*c] + 16*fq
// read in place: no im2col, no swizzle (16 consecutive pixels -> 16 consecutive 16-byte slots).
// Weights: [64, 232] bf16, k = kh*32 + px*4 + ch (7 x 32 used, row pitch 464 B keeps the
// ds_read_b128 of 16 rows conflict-free).  v_mfma_f32_16x16x32_bf16, K = 7 steps of 32.
// Conv outputs go through scale/bias/ReLU into an LDS map (144 B per pixel, conflict-free 8-byte
// writes) that aliases the dead patch and weights; the pool phase reads 3x3 windows from it and
// writes whole 128-byte pixels.  80.6 KB LDS -> two workgroups per CU overlap load / MFMA / pool.
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ uint4 g_zero_page_c1[4];
#ifdef RVIT_C1_STAMPS   // diagnosis build only: s_memtime per workgroup phase (scripts/c1_stamps.py)
__device__ unsigned long long g_c1_stamps[16384 * 8];
#define C1_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_c1_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// rolling-band kernel: per-phase cycle sums over the bands of a workgroup (wave 0)
#define C1R_DECL unsigned long long c1r_t = __builtin_amdgcn_s_memtime(), c1r_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define C1R_LAP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); c1r_acc[k] += n_ - c1r_t; c1r_t = n_; } while (0)
#define C1R_FLUSH do { if (threadIdx.x == 0 && blockIdx.x < 16384) for (int k_ = 0; k_ < 8; ++k_) g_c1_stamps[blockIdx.x * 8 + k_] = c1r_acc[k_]; } while (0)
#else
#define C1_STAMP(k) do {} while (0)
#define C1R_DECL do {} while (0)
#define C1R_LAP(k) do {} while (0)
#define C1R_FLUSH do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

constexpr int HP = 230, WP = 232;          // packed image rows / cols (halo 3 top/left, 3 / 5 bottom/right)
constexpr int WC = 112;                    // conv output is 112 x 112
constexpr int HQ = 56, WQ = 56;            // pooled output
constexpr int TP = 2;                      // pooled rows per workgroup
constexpr int NCR = 2 * TP + 1;            // conv rows per workgroup
constexpr int ROW_B = WP * 8;              // 1856 bytes per packed row
constexpr int W_PITCH = 464;               // weight row pitch in LDS and in the packed tensor (232 bf16)
constexpr int W_BYTES = 64 * W_PITCH;      // 29696 = 29 KB
constexpr int PATCH_BYTES = 28 * 1024;     // 15 rows = 27840 B, rounded to whole 1 KB DMA instructions
constexpr int CO_PITCH = 144;              // conv-out LDS pitch per pixel
constexpr int CO_BYTES = NCR * WC * CO_PITCH;  // 80640
constexpr int NBLK = NCR * (WC / 16);      // 35 blocks of 16 pixels
constexpr int BLK_PER_WAVE = (NBLK + 7) / 8;   // 5

// F32IN: the patch is built in LDS straight from the caller's fp32 NCHW image [B,3,224,224] (the model's input
// contract, train_vit.py:141-146) instead of being DMA'd from the packed bf16 NHWC4 copy: the separate pack pass
// (rvit_input_pack_nhwc4: 308 MB read + 218 MB written + 92 us per 512 frames) disappears.  A thread converts quads
// of four pixels (three 16-byte loads, one per channel plane) to four (c0, c1, c2, 0) bf16 pixels; halo rows and
// columns come from zero-filling the patch first.  The bf16 values equal the pack kernel's, so both forms give
// identical results.
template <bool F32IN>
__global__ __launch_bounds__(512, 4) void conv1_pool_kernel(const void* __restrict__ img_any, const bf16* __restrict__ w,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ bias, bf16* __restrict__ pooled,
                                                            int relu) {
  const bf16* __restrict__ img = static_cast<const bf16*>(img_any);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C1_STAMP(0);
  char* const sW = smem;
  char* const sP = smem + W_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int band = blockIdx.x % (HQ / TP), b = blockIdx.x / (HQ / TP);
  const int py0 = band * TP;
  const int cr0 = 2 * py0 - 1;                    // first conv row of the band (-1 for the top band: skipped)
  const int prow0 = cr0 < 0 ? 0 : 2 * cr0;        // first packed row held by the patch
  const int prow1 = 2 * (cr0 + NCR - 1) + 6;      // last packed row needed (<= 228)
  const int pbytes = (prow1 - prow0 + 1) * ROW_B;
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page_c1);

  // ---- stage weights (29 x 1 KB) and the image patch (<= 28 x 1 KB), both contiguous copies ---------
  for (int i = wave; i < W_BYTES / 1024; i += 8) glds16(reinterpret_cast<const char*>(w) + i * 1024 + lane * 16, sW + i * 1024);
  if constexpr (!F32IN) {
    const char* src = reinterpret_cast<const char*>(img) + ((long)b * HP + prow0) * ROW_B;
    for (int i = wave; i < PATCH_BYTES / 1024; i += 8) {
      const int off = i * 1024 + lane * 16;
      glds16(off < pbytes ? (const void*)(src + off) : (const void*)zero_src, sP + i * 1024);
    }
  } else {
    (void)pbytes; (void)zero_src;
    const float* __restrict__ x = static_cast<const float*>(img_any) + (long)b * 3 * 224 * 224;
    const int nrows = prow1 - prow0 + 1;                 // <= 15 padded rows; padded row pr = image row pr - 3
    // issue the loads first (2 units of 4 pixels x 3 planes per thread), zero the patch while they fly
    constexpr int UNITS = 15 * 56, UPT = (UNITS + 511) / 512;   // 2
    float4 v[UPT][3];
    int dst[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int id = t + u * 512;
      const int r = id / 56, j = id - r * 56;
      const int ir = prow0 + r - 3;
      dst[u] = -1;
      if (r < nrows && (unsigned)ir < 224u) {
        const float* src = x + (long)ir * 224 + 4 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[u][c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
        dst[u] = r * ROW_B + (4 * j + 3) * 8;             // padded column 4j + 3, 8 bytes per pixel
      }
    }
    for (int i = t; i < PATCH_BYTES / 16; i += 512) *reinterpret_cast<uint4*>(sP + i * 16) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (dst[u] < 0) continue;
      const float p0[4] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w};
      const float p1[4] = {v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
      const float p2[4] = {v[u][2].x, v[u][2].y, v[u][2].z, v[u][2].w};
      uint2 px[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        px[e].x = pack_bf16x2(p0[e], p1[e]);
        px[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + dst[u];                               // 8-byte aligned (offset 24 mod 32)
      *reinterpret_cast<uint2*>(d) = px[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{px[1].x, px[1].y, px[2].x, px[2].y};
      *reinterpret_cast<uint2*>(d + 24) = px[3];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  C1_STAMP(1);

  // ---- MFMA: each wave owns up to 5 blocks of 16 pixels x 64 channels ----------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[BLK_PER_WAVE][4];
  int a_off[BLK_PER_WAVE];
  bool live[BLK_PER_WAVE];
#pragma unroll
  for (int q = 0; q < BLK_PER_WAVE; ++q) {
    const int blk = wave + 8 * q;
    const int lr = blk / 7, cb = blk - lr * 7;
    live[q] = blk < NBLK && cr0 + lr >= 0;          // wave-uniform
    a_off[q] = ((2 * (cr0 + lr) - prow0) * WP + 2 * (cb * 16 + fr)) * 8 + fq * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll 1
  for (int kh = 0; kh < 7; ++kh) {
    bf16x8 bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bfr[j] = *reinterpret_cast<const bf16x8*>(sW + (j * 16 + fr) * W_PITCH + kh * 64 + fq * 16);
#pragma unroll
    for (int q = 0; q < BLK_PER_WAVE; ++q) {
      if (!live[q]) continue;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(sP + a_off[q] + kh * ROW_B);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af, acc[q][j], 0, 0, 0);
    }
  }
  C1_STAMP(2);
  __syncthreads();  // patch and weights are dead: the conv-out map aliases them
  C1_STAMP(3);

  // ---- scale / bias / ReLU -> bf16 conv-out map in LDS ([NCR][112] pixels, 144 B pitch) -----------------
  // lane holds channels 16j + 4fq + (0..3) of pixel fr of each of its blocks
  float sc[4][4], bi[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 s4 = *reinterpret_cast<const float4*>(scale + 16 * j + 4 * fq);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * j + 4 * fq);
    sc[j][0] = s4.x; sc[j][1] = s4.y; sc[j][2] = s4.z; sc[j][3] = s4.w;
    bi[j][0] = b4.x; bi[j][1] = b4.y; bi[j][2] = b4.z; bi[j][3] = b4.w;
  }
#pragma unroll
  for (int q = 0; q < BLK_PER_WAVE; ++q) {
    if (!live[q]) continue;
    const int blk = wave + 8 * q;
    char* dst = smem + (blk * 16 + fr) * CO_PITCH + fq * 8;   // blk*16 = lr*112 + cb*16
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[q][j][r] * sc[j][r] + bi[j][r];   // the ReLU is applied by the pool phase
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst + j * 32) = o;
    }
  }
  __syncthreads();
  C1_STAMP(4);

  // ---- 3x3 / 2 max-pool out of LDS; one 16-byte channel chunk of one pooled pixel per thread ------------
  // With the ReLU (the model's case) the pool runs on the bf16 BITS: relu and max commute with the (monotone,
  // sign-preserving) rounding to bf16, and non-negative bf16 values order like their bit patterns read as int16, while
  // every negative value (sign bit set) reads as a negative int16: max_i16 over the window and 0 IS relu(max(window)).
  // Two channels per v_pk_max_i16 instead of two unpacks + two v_max_f32: the pool phase was vector-ALU-bound
  // (profiles/r02_stem_entry_notes.txt).  A window whose values are all <= 0 gives +0.
  if (relu) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    for (int id = t; id < TP * WQ * 8; id += 512) {
      const int c8 = id & 7, pp = id >> 3;
      const int pyl = pp / WQ, px = pp - pyl * WQ;
      s16x2 best[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) best[k] = s16x2{0, 0};
#pragma unroll
      for (int dr = 0; dr < 3; ++dr) {
        const int lr = 2 * pyl + dr;
        if (cr0 + lr < 0) continue;
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          const int cc = 2 * px - 1 + dc;
          if (cc < 0) continue;
          const uint4 u = *reinterpret_cast<const uint4*>(smem + (lr * WC + cc) * CO_PITCH + c8 * 16);
          const unsigned uw[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s16x2 v;
            __builtin_memcpy(&v, &uw[k], 4);
            best[k] = __builtin_elementwise_max(best[k], v);
          }
        }
      }
      uint4 o;
      __builtin_memcpy(&o, best, 16);
      *reinterpret_cast<uint4*>(pooled + (((long)b * HQ + py0 + pyl) * WQ + px) * 64 + c8 * 8) = o;
    }
    C1_STAMP(5);
    return;
  }
  for (int id = t; id < TP * WQ * 8; id += 512) {
    const int c8 = id & 7, pp = id >> 3;
    const int pyl = pp / WQ, px = pp - pyl * WQ;
    float best[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) best[k] = -INFINITY;
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) {
      const int lr = 2 * pyl + dr;            // conv row cr0 + lr = 2*py - 1 + dr
      if (cr0 + lr < 0) continue;
#pragma unroll
      for (int dc = 0; dc < 3; ++dc) {
        const int cc = 2 * px - 1 + dc;
        if (cc < 0) continue;
        const uint4 u = *reinterpret_cast<const uint4*>(smem + (lr * WC + cc) * CO_PITCH + c8 * 16);
        best[0] = fmaxf(best[0], bf16_lo(u.x)); best[1] = fmaxf(best[1], bf16_hi(u.x));
        best[2] = fmaxf(best[2], bf16_lo(u.y)); best[3] = fmaxf(best[3], bf16_hi(u.y));
        best[4] = fmaxf(best[4], bf16_lo(u.z)); best[5] = fmaxf(best[5], bf16_hi(u.z));
        best[6] = fmaxf(best[6], bf16_lo(u.w)); best[7] = fmaxf(best[7], bf16_hi(u.w));
      }
    }
    uint4 o;  // the values are bf16 already: packing is exact
    o.x = pack_bf16x2(best[0], best[1]); o.y = pack_bf16x2(best[2], best[3]);
    o.z = pack_bf16x2(best[4], best[5]); o.w = pack_bf16x2(best[6], best[7]);
    *reinterpret_cast<uint4*>(pooled + (((long)b * HQ + py0 + pyl) * WQ + px) * 64 + c8 * 8) = o;
  }
  C1_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------------
// Rolling-band form (round 3; fp32 NCHW input + ReLU: what eval-mode inference runs).  The band kernel above fetches
// 1.84x its input (the 15-row patches of neighbouring bands overlap by 7 rows: profiles/r02_pmc_traffic.json), recomputes
// every fifth conv row, runs load -> MFMA -> LDS map -> pool strictly in sequence inside a workgroup and needs 80 KB of
// LDS for the conv-out map.  Here a workgroup walks DOWN an image (or a segment of it) two pooled rows at a time:
//   * the packed bf16 rows live in a 24-row LDS ring; a band needs 13 rows (conv rows c0 .. c0+3), 8 of them new.  The
//     new rows of band k+1 are requested (fp32, three planes) before the MFMAs of band k and written into the ring
//     after its stores, so their HBM latency hides behind the band; one workgroup barrier per band.  Every input row
//     is fetched once per segment;
//   * wave w owns conv columns 14w-1 .. 14w+14 of every row (16 columns = one MFMA block; 15 of them are the windows
//     of pooled pixels 7w .. 7w+6; 14 % more columns than the map has instead of 25 % more rows): the vertical
//     maximum is a register maximum, the horizontal one two DPP row shifts, and the conv row shared with the next band
//     (row c0+3 = 2*py+3) is carried in registers as packed bf16 -> no conv row is computed twice, no LDS map;
//   * maxima run on the bf16 BITS (v_pk_max_i16 with a floor of 0 = ReLU, as the band kernel's pool phase), pooled
//     pixels leave as 16-byte stores after a v_permlane16_swap between lane rows (gemm8p.hip's epilogue idiom).
// Same MFMAs in the same kh order, the same fused multiply-add of scale / bias and the same rounding: bit-identical
// to the band kernel (tests/test_kernels_gpu.py::test_stem_conv1_pool_fused).  74.75 KB LDS, 128 registers: two
// workgroups per CU whose phases interleave.
constexpr int RING_ROWS = 24;
constexpr int RING_BYTES = RING_ROWS * ROW_B;            // 44,544
constexpr int ROLL_SB_OFF = W_BYTES + RING_BYTES;        // scale[64], bias[64] as floats
constexpr int ROLL_LDS = ROLL_SB_OFF + 512;              // 74,752
constexpr int BANDS = HQ / 2;                            // 28 bands of two pooled rows per image

typedef short c1_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t c1_pkmax(uint32_t a, uint32_t b) {
  c1_s16x2 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  x = __builtin_elementwise_max(x, y);
  uint32_t r;
  __builtin_memcpy(&r, &x, 4);
  return r;
}

// NR conv rows x 16 columns x 64 channels: slot0 = ring slot of the first row's kh = 0 input row (wave-uniform).
// One fragment set and a runtime loop over the kernel rows: the four waves of a SIMD hide each other's LDS round trips.
// Measured against it on one box (scripts/bench_conv1.py, B = 512, 250 us): two fragment sets with the next kernel
// row requested before the MFMAs of this one (needs two passes of two rows to fit 128 registers, weights read twice)
// 263 us; the pixel fragment of row q requested for kernel row kh + 1 right after its last use 255 us; kernel rows
// unrolled: the compiler hoists every read and spills.  Raised wave priority around the MFMAs (the matrix pipe is the
// scarce unit, the other waves are doing vector work): -1..2 %, kept.
template <int NR>
__device__ __forceinline__ void c1_rows_mfma(const char* sWl, const char* sPl, int slot0, f32x4 (&acc)[NR][4]) {
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);   // keeps the epilogue's work from being hoisted over the loop (it spills then)
  __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
  for (int kh = 0; kh < 7; ++kh) {
    bf16x8 bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sWl + j * 16 * W_PITCH + kh * 64);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      int s = slot0 + 2 * q + kh;
      s = s >= RING_ROWS ? s - RING_ROWS : s;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(sPl + s * ROW_B);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af, acc[q][j], 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

// scale / bias (one fused multiply-add, as the band kernel) -> packed bf16: pk[q][2j + d] = channels 16j + 4fq + 2d, + 1
// of conv row q (the scale / bias quad of a channel block is read once for all rows)
template <int NR>
__device__ __forceinline__ void c1_affine_pack(const f32x4 (&acc)[NR][4], const float* sSB, int fq, uint32_t (&pk)[NR][8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 s4 = *reinterpret_cast<const float4*>(sSB + 16 * j + 4 * fq);
    const float4 b4 = *reinterpret_cast<const float4*>(sSB + 64 + 16 * j + 4 * fq);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      pk[q][2 * j] = pack_bf16x2(fmaf(acc[q][j][0], s4.x, b4.x), fmaf(acc[q][j][1], s4.y, b4.y));
      pk[q][2 * j + 1] = pack_bf16x2(fmaf(acc[q][j][2], s4.z, b4.z), fmaf(acc[q][j][3], s4.w, b4.w));
    }
  }
}

// one pooled row out of three packed conv rows: vertical maximum, horizontal maximum over lanes fr-1, fr, fr+1, floor 0
// (ReLU); the pooled pixel 7w + i sits in lane fr = 2i + 1.  `edge`: this lane is conv column -1 (outside the map).
__device__ __forceinline__ void c1_pool_store(const uint32_t (&a)[8], const uint32_t (&b)[8], const uint32_t (&c)[8],
                                              bool edge_wave, bool edge, bool store_ok, bf16* dst) {
  uint32_t h[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t m = c1_pkmax(c1_pkmax(a[k], b[k]), c[k]);
    if (edge_wave) m = edge ? 0u : m;   // wave-uniform: only wave 0 holds column -1
    const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
    h[k] = c1_pkmax(c1_pkmax(c1_pkmax(m, l), r), 0u);
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    // lane rows fq 0 <-> 1, 2 <-> 3 of channel blocks 2pr and 2pr + 1: every lane ends with 8 consecutive channels
    const auto s0 = __builtin_amdgcn_permlane16_swap(h[4 * pr + 0], h[4 * pr + 2], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(h[4 * pr + 1], h[4 * pr + 3], false, false);
    uint4 o;
    o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
    if (store_ok) *reinterpret_cast<uint4*>(dst + 32 * pr) = o;
  }
}

__global__ __launch_bounds__(512, 4) void conv1_pool_roll_kernel(const float* __restrict__ xin, const bf16* __restrict__ w,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ bias, bf16* __restrict__ pooled,
                                                                 int wpi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sW = smem;
  char* const sP = smem + W_BYTES;
  float* const sSB = reinterpret_cast<float*>(smem + ROLL_SB_OFF);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / wpi, seg = blockIdx.x - b * wpi;
  const int nb = BANDS / wpi;
  const int band0 = seg * nb;
  const float* __restrict__ x = xin + (long)b * 3 * 224 * 224;

  // ---- weights by LDS-DMA, scale / bias, zero ring ------------------------------------------------------------------
  for (int i = wave; i < W_BYTES / 1024; i += 8) glds16(reinterpret_cast<const char*>(w) + i * 1024 + lane * 16, sW + i * 1024);
  if (t < 64) sSB[t] = scale[t];
  else if (t < 128) sSB[t] = bias[t - 64];
  // first window: packed rows p_first .. 2 c0 + 12 (the two extra rows above serve the carried conv row c0 - 1)
  int c0 = 4 * band0;
  const int p_first = c0 > 0 ? 2 * c0 - 2 : 0;
  {
    const int nrows = 2 * c0 + 12 - p_first + 1;                 // 13 or 15
    constexpr int UPT = 2;                                       // 15 * 56 = 840 units of 4 pixels x 3 planes
    float4 v[UPT][3];
    int dst[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int id = t + u * 512;
      const int r = id / 56, j = id - r * 56;
      const int p = p_first + r, ir = p - 3;
      dst[u] = -1;
      if (r < nrows && (unsigned)ir < 224u) {
        const float* src = x + (long)ir * 224 + 4 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[u][c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
        dst[u] = (p % RING_ROWS) * ROW_B + (4 * j + 3) * 8;
      }
    }
    for (int i = t; i < RING_BYTES / 16; i += 512) *reinterpret_cast<uint4*>(sP + i * 16) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (dst[u] < 0) continue;
      const float p0[4] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w};
      const float p1[4] = {v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
      const float p2[4] = {v[u][2].x, v[u][2].y, v[u][2].z, v[u][2].w};
      uint2 px[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        px[e].x = pack_bf16x2(p0[e], p1[e]);
        px[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + dst[u];
      *reinterpret_cast<uint2*>(d) = px[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{px[1].x, px[1].y, px[2].x, px[2].y};
      *reinterpret_cast<uint2*>(d + 24) = px[3];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane operand bases: weights rows j*16 + fr, chunk fq of a kernel row; pixels: conv column 14w - 1 + fr reads the
  // 16 bytes at packed column 2 (14w - 1 + fr) + 2 fq (for w = fr = fq = 0 that is 16 bytes below the ring row: finite
  // weight bytes that only reach the masked column -1)
  const char* const sWl = sW + fr * W_PITCH + fq * 16;
  const char* const sPl = sP + 16 * (14 * wave - 1 + fr + fq);
  const bool edge = wave == 0 && fr == 0;
  const bool store_ok = (fr & 1) && fr < 14;
  const int px = 7 * wave + (fr >> 1);
  const int ch_off = (fq & 1) * 16 + (fq >> 1) * 8;

  // the band-invariant part of the prefetch unit of this thread: row r of the 8 new rows, quad j
  const int pf_r = t / 56, pf_j = t - pf_r * 56;
  const bool pf_thread = t < 8 * 56;
  const int pf_col = (4 * pf_j + 3) * 8;

  uint32_t T[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) T[k] = 0u;      // above the map: excluded = the ReLU floor
  if (c0 > 0) {
    f32x4 a1[1][4];
    uint32_t t1[1][8];
    c1_rows_mfma<1>(sWl, sPl, __builtin_amdgcn_readfirstlane((2 * c0 - 2) % RING_ROWS), a1);
    c1_affine_pack<1>(a1, sSB, fq, t1);
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = t1[0][k];
  }
  C1R_DECL;

  // Waves 4-7 run their epilogue one band late: an iteration is {MFMA(k), pack(k), pool(k)} for waves 0-3 and
  // {pool(k-1), MFMA(k), pack(k)} for waves 4-7 (they carry the band's packed rows over the barrier), so on every SIMD (wave w and w + 4 share one) one wave multiplies while
  // the other does the vector work of its pooling; without this all eight waves queued for the matrix pipe together
  // and then for the vector ALU together (stamps: 14.0k cycles per band against 7.2k of MFMA issue per SIMD).
  const bool late = wave >= 4;
  uint32_t r[4][8];                           // the band's four conv rows, packed bf16 (what the late group carries)
  auto pool = [&](int c_band) {
    bf16* dst = pooled + (((long)b * HQ + (c_band >> 1)) * WQ + px) * 64 + ch_off;
    c1_pool_store(T, r[0], r[1], wave == 0, edge, store_ok, dst);
    c1_pool_store(r[1], r[2], r[3], wave == 0, edge, store_ok, dst + WQ * 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = r[3][k];
  };
#pragma unroll 1
  for (int bi = 0; bi < nb; ++bi, c0 += 4) {
    const bool has_next = bi + 1 < nb;
    // ---- request the 8 new rows of the next band -------------------------------------------------------------------
    float4 nv[3];
    nv[0] = nv[1] = nv[2] = float4{0.f, 0.f, 0.f, 0.f};
    const int np = 2 * c0 + 13 + pf_r;                      // packed row; image row np - 3 (>= 10)
    if (has_next && pf_thread && np - 3 < 224) {
      const float* src = x + (long)(np - 3) * 224 + 4 * pf_j;
#pragma unroll
      for (int c = 0; c < 3; ++c) nv[c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
    }
    C1R_LAP(0);
    if (late && bi > 0) pool(c0 - 4);
    C1R_LAP(1);
    // ---- conv rows c0 .. c0 + 3 ---------------------------------------------------------------------------------------
    {
      f32x4 acc[4][4];
      c1_rows_mfma<4>(sWl, sPl, __builtin_amdgcn_readfirstlane((2 * c0) % RING_ROWS), acc);
      c1_affine_pack<4>(acc, sSB, fq, r);
    }
    C1R_LAP(2);
    if (!late) pool(c0);
    C1R_LAP(3);
    // ---- the next band's rows into the ring (rows below the image: zeros) ------------------------------------------
    if (has_next && pf_thread) {
      const float p0[4] = {nv[0].x, nv[0].y, nv[0].z, nv[0].w};
      const float p1[4] = {nv[1].x, nv[1].y, nv[1].z, nv[1].w};
      const float p2[4] = {nv[2].x, nv[2].y, nv[2].z, nv[2].w};
      uint2 q4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q4[e].x = pack_bf16x2(p0[e], p1[e]);
        q4[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + (np % RING_ROWS) * ROW_B + pf_col;
      *reinterpret_cast<uint2*>(d) = q4[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{q4[1].x, q4[1].y, q4[2].x, q4[2].y};
      *reinterpret_cast<uint2*>(d + 24) = q4[3];
    }
    C1R_LAP(4);
    __syncthreads();
    C1R_LAP(5);
  }
  if (late) pool(c0 - 4);
  C1R_FLUSH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Training form of the rolling band (round 3): the raw conv rows go to HBM as bf16 [B,112,112,64] (train-mode BatchNorm
// needs the batch statistics of the whole map before it can normalise a pixel) together with the per-channel sum and sum
// of squares of the STORED values, one slot of 128 floats per workgroup (rvit_bn_stats_finalize adds the slots in order).
// Replaces the implicit-GEMM launch of this conv (K = 224: four K steps per 256-pixel tile, 12,544 tiles at B = 256, all
// prologue and epilogue: 289 us) and its 12,544 statistics slots.  Same ring, same fragments and the same kh order as the
// kernel above; a wave owns conv columns 14w - 1 .. 14w + 14 and stores the fourteen in the middle.  No stagger of the
// two wave groups here, and a band's four rows in two passes of two (the 32 statistics accumulators take the registers
// the carried rows / the other two rows' accumulators would need at four waves per SIMD).
__global__ __launch_bounds__(512, 4) void conv1_rows_roll_kernel(const float* __restrict__ xin, const bf16* __restrict__ w,
                                                                 bf16* __restrict__ y, float* __restrict__ stat_part,
                                                                 int wpi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sW = smem;
  char* const sP = smem + W_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / wpi, seg = blockIdx.x - b * wpi;
  const int nb = BANDS / wpi;
  const int band0 = seg * nb;
  const float* __restrict__ x = xin + (long)b * 3 * 224 * 224;

  // weights [64][224] -> LDS rows of W_PITCH bytes (the pitch that keeps the fragment reads conflict-free)
  for (int i = t; i < 64 * 28; i += 512) {
    const int r = i / 28, c = i - r * 28;
    *reinterpret_cast<uint4*>(sW + r * W_PITCH + c * 16) = *reinterpret_cast<const uint4*>(w + r * 224 + c * 8);
  }
  int c0 = 4 * band0;
  {
    const int p_first = 2 * c0;
    constexpr int UPT = 2;                                       // 13 * 56 = 728 units of 4 pixels x 3 planes
    float4 v[UPT][3];
    int dst[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int id = t + u * 512;
      const int r = id / 56, j = id - r * 56;
      const int p = p_first + r, ir = p - 3;
      dst[u] = -1;
      if (r < 13 && (unsigned)ir < 224u) {
        const float* src = x + (long)ir * 224 + 4 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[u][c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
        dst[u] = (p % RING_ROWS) * ROW_B + (4 * j + 3) * 8;
      }
    }
    for (int i = t; i < RING_BYTES / 16; i += 512) *reinterpret_cast<uint4*>(sP + i * 16) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (dst[u] < 0) continue;
      const float p0[4] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w};
      const float p1[4] = {v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
      const float p2[4] = {v[u][2].x, v[u][2].y, v[u][2].z, v[u][2].w};
      uint2 px[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        px[e].x = pack_bf16x2(p0[e], p1[e]);
        px[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + dst[u];
      *reinterpret_cast<uint2*>(d) = px[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{px[1].x, px[1].y, px[2].x, px[2].y};
      *reinterpret_cast<uint2*>(d + 24) = px[3];
    }
  }
  __syncthreads();

  const char* const sWl = sW + fr * W_PITCH + fq * 16;
  const char* const sPl = sP + 16 * (14 * wave - 1 + fr + fq);
  const bool store_ok = fr >= 1 && fr <= 14;
  const int col = 14 * wave + fr - 1;
  const int ch_off = (fq & 1) * 16 + (fq >> 1) * 8;
  const int pf_r = t / 56, pf_j = t - pf_r * 56;
  const bool pf_thread = t < 8 * 56;
  const int pf_col = (4 * pf_j + 3) * 8;

  float s1[16], s2[16];   // channels 16 j + 4 fq + 2 d + e at index 4 j + 2 d + e
#pragma unroll
  for (int k = 0; k < 16; ++k) { s1[k] = 0.f; s2[k] = 0.f; }

#pragma unroll 1
  for (int bi = 0; bi < nb; ++bi, c0 += 4) {
    const bool has_next = bi + 1 < nb;
    float4 nv[3];
    nv[0] = nv[1] = nv[2] = float4{0.f, 0.f, 0.f, 0.f};
    const int np = 2 * c0 + 13 + pf_r;
    if (has_next && pf_thread && np - 3 < 224) {
      const float* src = x + (long)(np - 3) * 224 + 4 * pf_j;
#pragma unroll
      for (int c = 0; c < 3; ++c) nv[c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {   // two passes of two rows: 32 accumulator registers beside the 32 of the statistics
      f32x4 acc[2][4];
      c1_rows_mfma<2>(sWl, sPl, __builtin_amdgcn_readfirstlane((2 * c0 + 4 * half) % RING_ROWS), acc);
      bf16* dst = y + (((long)b * WC + c0 + 2 * half) * WC + col) * 64 + ch_off;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pk[2 * j] = pack_bf16x2(acc[q][j][0], acc[q][j][1]);
          pk[2 * j + 1] = pack_bf16x2(acc[q][j][2], acc[q][j][3]);
        }
        if (store_ok) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float lo = bf16_lo(pk[k]), hi = bf16_hi(pk[k]);
            s1[2 * k] += lo; s2[2 * k] = fmaf(lo, lo, s2[2 * k]);
            s1[2 * k + 1] += hi; s2[2 * k + 1] = fmaf(hi, hi, s2[2 * k + 1]);
          }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {   // lane rows fq 0 <-> 1, 2 <-> 3: eight consecutive channels per lane
          const auto w0 = __builtin_amdgcn_permlane16_swap(pk[4 * pr + 0], pk[4 * pr + 2], false, false);
          const auto w1 = __builtin_amdgcn_permlane16_swap(pk[4 * pr + 1], pk[4 * pr + 3], false, false);
          uint4 o;
          o.x = w0[0]; o.y = w1[0]; o.z = w0[1]; o.w = w1[1];
          if (store_ok) *reinterpret_cast<uint4*>(dst + (long)q * WC * 64 + 32 * pr) = o;
        }
      }
    }
    if (has_next && pf_thread) {
      const float p0[4] = {nv[0].x, nv[0].y, nv[0].z, nv[0].w};
      const float p1[4] = {nv[1].x, nv[1].y, nv[1].z, nv[1].w};
      const float p2[4] = {nv[2].x, nv[2].y, nv[2].z, nv[2].w};
      uint2 q4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q4[e].x = pack_bf16x2(p0[e], p1[e]);
        q4[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + (np % RING_ROWS) * ROW_B + pf_col;
      *reinterpret_cast<uint2*>(d) = q4[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{q4[1].x, q4[1].y, q4[2].x, q4[2].y};
      *reinterpret_cast<uint2*>(d + 24) = q4[3];
    }
    __syncthreads();
  }

  // ---- statistics: the sixteen pixel lanes of a lane row, then the eight waves in wave order, one slot per workgroup
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      s1[k] += __shfl_xor(s1[k], off, 64);
      s2[k] += __shfl_xor(s2[k], off, 64);
    }
  float* const sRed = reinterpret_cast<float*>(sP);   // [wave][sum 64 | sumsq 64]; the ring is dead (barrier above)
  if (fr == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ch = 16 * (k >> 2) + 4 * fq + (k & 3);
      sRed[wave * 128 + ch] = s1[k];
      sRed[wave * 128 + 64 + ch] = s2[k];
    }
  }
  __syncthreads();
  if (t < 128) {
    float tot = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < 8; ++w_) tot += sRed[w_ * 128 + t];
    stat_part[(long)blockIdx.x * 128 + t] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same conv on the rolling band (round 3):
//   dW[co][kh][px][ch] = sum over images, conv rows r, conv columns c of dY[r][c][co] * packed[2r + kh][2c + px][ch]
// As an implicit GEMM (wgrad.hip, N = 64, K = 224) this ran two K tiles per M split that each streamed all of dY, one
// 64-row tile per barrier: 330 us at B = 256 for 0.56 GB of operands.  Here a workgroup walks down an image as the
// forward kernel does (same ring of packed rows built from the fp32 input, every input row fetched once), one conv row
// per step: the row of dY (112 pixels x 64 channels) arrives by LDS-DMA, double-buffered, and wave kh contracts it with
// the packed row 2r + kh over the pixels: both operands through the transposing LDS read (the contraction index is the
// pixel: dY rows are 128 bytes apart, the windows of consecutive conv columns 16 bytes apart in the packed row - the
// read takes one address per lane, so overlapping rows cost nothing).  Wave kh keeps dW[0..63][kh][0..31] in 32
// registers for the whole walk; one partial slot of 64 x 224 floats per workgroup, added in slot order by
// rvit_ordered_finalize.  dY rows are padded to 128 pixels with zeros (four 32-pixel MFMA steps).
__device__ __forceinline__ int c1_swz(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }   // on 16-B chunks
__device__ __forceinline__ bf16x4 c1_tr_read(const char* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p)));
}
constexpr int WG_YROW = 128 * 128;                       // bytes of one dY row image (128 pixels x 64 channels)
constexpr int WG_YOFF = (RING_BYTES + 1023) & ~1023;     // the dY images start on a 1 KB boundary (LDS-DMA pieces)
constexpr int WG_LDS = WG_YOFF + 2 * WG_YROW;            // 77,824: two workgroups per CU
// LDS-DMA as inline assembly: hipcc's wait-count pass puts `s_waitcnt vmcnt(0)` in front of the first transposing LDS
// read it sees while a global_load_lds IT issued is outstanding, which here would expose the latency of the next row's
// request in every row.  The kernel's own waits + barriers order the accesses.  (m0 has no other user in this kernel.)
__device__ __forceinline__ void c1_glds16_asm(const void* src, char* lds_dst_wave_base) {
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)lds_dst_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory");
}

__global__ __launch_bounds__(512, 4) void conv1_wgrad_roll_kernel(const float* __restrict__ xin, const bf16* __restrict__ dy,
                                                                  float* __restrict__ part, int wpi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sP = smem;
  char* const sY = smem + WG_YOFF;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x / wpi, seg = blockIdx.x - b * wpi;
  const int nb = BANDS / wpi;
  const int band0 = seg * nb;
  const float* __restrict__ x = xin + (long)b * 3 * 224 * 224;
  const bf16* __restrict__ dyb = dy + (long)b * WC * WC * 64;
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page_c1);

  // dY row `r` -> buffer `buf`: 14 pieces of 1 KB (8 pixels each); piece p: lane -> pixel 8p + (lane >> 3), 16-byte chunk
  // slot lane & 7 holding source chunk slot ^ swz(pixel) (conflict-free transposing reads, as wgrad3x3.hip); waves 0-6 two
  // pieces each.  Rows past the segment: zero pages (keeps the wait counts fixed).
  auto issue_dy = [&](int r, int buf, bool live) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;
      if (piece < 14) {
        const int pix = piece * 8 + (lane >> 3);
        const bf16* src = live ? dyb + ((long)r * WC + pix) * 64 + (((lane & 7) ^ c1_swz(pix)) << 3) : zero_src;
        c1_glds16_asm(src, sY + buf * WG_YROW + piece * 1024);
      }
    }
  };

  int c0 = 4 * band0;
  const int r_end = c0 + 4 * nb;
  // ---- zero the ring and the dY images (their pixels 112-127 stay zero), first window of packed rows ----
  {
    const int p_first = 2 * c0;
    constexpr int UPT = 2;
    float4 v[UPT][3];
    int dst[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int id = t + u * 512;
      const int r = id / 56, j = id - r * 56;
      const int p = p_first + r, ir = p - 3;
      dst[u] = -1;
      if (r < 13 && (unsigned)ir < 224u) {
        const float* src = x + (long)ir * 224 + 4 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[u][c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
        dst[u] = (p % RING_ROWS) * ROW_B + (4 * j + 3) * 8;
      }
    }
    for (int i = t; i < WG_LDS / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (dst[u] < 0) continue;
      const float p0[4] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w};
      const float p1[4] = {v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
      const float p2[4] = {v[u][2].x, v[u][2].y, v[u][2].z, v[u][2].w};
      uint2 px[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        px[e].x = pack_bf16x2(p0[e], p1[e]);
        px[e].y = pack_bf16x2(p2[e], 0.f);
      }
      char* d = sP + dst[u];
      *reinterpret_cast<uint2*>(d) = px[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{px[1].x, px[1].y, px[2].x, px[2].y};
      *reinterpret_cast<uint2*>(d + 24) = px[3];
    }
  }
  issue_dy(c0, 0, true);

  // transposing-read geometry: within a 16-lane group lane (q, pp) supplies row q, columns 4 pp .. 4 pp + 3 of a 4 x 16
  // block and receives rows 0..3 of column lane & 15; lane group g takes pixels 8 g .. 8 g + 7 of a 32-pixel step
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int pix0 = 8 * g + q;                                   // + 32 ks (+ 4 for the second read)
  const int y_off = pix0 * 128 + (pp >> 1) * 16 + (pp & 1) * 8;  // + segment (co block ^ swizzle) << 5
  const int y_sw = c1_swz(pix0) >> 1;                           // the same for pix0 + 4 and + 32 ks
  const int x_off = pix0 * 16 + pp * 8;                          // + 32 col block, + 512 ks, + 64 for the second read
  const int pf_r = t / 56, pf_j = t - pf_r * 56;
  const int pf_col = (4 * pf_j + 3) * 8;

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 nv[3];
  int np = 0;
#pragma unroll 1
  for (int r = c0; r < r_end; ++r) {
    const int ri = r - c0;
    const bool band_start = (ri & 3) == 0, band_end = (ri & 3) == 3;
    const bool has_next_band = r + (4 - (ri & 3)) < r_end;
    // this row's dY pieces have landed (every wave's, after the barrier); the other buffer is free again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (band_start && has_next_band && wave < 7) {
      // request the 8 new packed rows of the next band (fp32, three planes) ahead of the pieces: the loop-top wait of the
      // next row covers both.  Rows below the image re-read row 0 and are zeroed when they are packed (band end).
      np = 2 * r + 13 + pf_r;
      const float* src = x + (long)(np - 3 < 224 ? np - 3 : 0) * 224 + 4 * pf_j;
#pragma unroll
      for (int c = 0; c < 3; ++c) nv[c] = *reinterpret_cast<const float4*>(src + (long)c * 224 * 224);
    }
    issue_dy(r + 1, (ri + 1) & 1, r + 1 < r_end);
    if (wave < 7) {
      const char* const xrow = sP + ((2 * r + wave) % RING_ROWS) * ROW_B + x_off;
      const char* const yrow = sY + (ri & 1) * WG_YROW + y_off;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 xf[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const bf16x4 lo = c1_tr_read(xrow + ks * 512 + jb * 32);
          const bf16x4 hi = c1_tr_read(xrow + ks * 512 + jb * 32 + 64);
          xf[jb] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const char* yp = yrow + ks * 4096 + ((i ^ y_sw) << 5);
          const bf16x4 lo = c1_tr_read(yp);
          const bf16x4 hi = c1_tr_read(yp + 512);
          const bf16x8 yf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) acc[i][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[jb], acc[i][jb], 0, 0, 0);
        }
      }
    }
    if (band_end && has_next_band && wave < 7) {   // the next band's rows into the ring (not read by this band)
      const float p0[4] = {nv[0].x, nv[0].y, nv[0].z, nv[0].w};
      const float p1[4] = {nv[1].x, nv[1].y, nv[1].z, nv[1].w};
      const float p2[4] = {nv[2].x, nv[2].y, nv[2].z, nv[2].w};
      const bool in_img = np - 3 < 224;
      uint2 q4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q4[e].x = in_img ? pack_bf16x2(p0[e], p1[e]) : 0u;
        q4[e].y = in_img ? pack_bf16x2(p2[e], 0.f) : 0u;
      }
      char* d = sP + (np % RING_ROWS) * ROW_B + pf_col;
      *reinterpret_cast<uint2*>(d) = q4[0];
      *reinterpret_cast<uint4*>(d + 8) = uint4{q4[1].x, q4[1].y, q4[2].x, q4[2].y};
      *reinterpret_cast<uint2*>(d + 24) = q4[3];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-page pieces requested past the last row

  // ---- one slot per workgroup: lane holds dW[co = 16 i + 4 g + e][kh = wave][16 jb + (lane & 15)] ----
  if (wave < 7) {
    float* dst = part + (long)blockIdx.x * (64 * 224) + wave * 32 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(16 * i + 4 * g + e) * 224 + 16 * jb] = acc[i][jb][e];
  }
}

}  // namespace

int rvit_stem_conv1_pool_impl(const void* img, const void* w, const float* scale, const float* bias, void* pooled,
                              int B, int relu, int img_f32, hipStream_t st) {
  RVIT_CHECK_ARG(img && w && scale && bias && pooled && B > 0, "stem_conv1_pool: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(img) && rvit_aligned16(w) && rvit_aligned16(scale) && rvit_aligned16(bias) &&
                     rvit_aligned16(pooled),
                 "stem_conv1_pool: operands must be 16-byte aligned");
  RVIT_CHECK_ARG((long)B * (HQ / TP) < 0x7fffffffL, "stem_conv1_pool: batch too large");
  static_assert(CO_BYTES >= W_BYTES + PATCH_BYTES, "conv-out map must cover the staging area it aliases");
  // fp32 input + ReLU (eval-mode inference): the rolling-band kernel.  Workgroups per image: the fewest of the divisors
  // of 28 bands that still give two workgroups to every CU (B = 512: one per image, every input row fetched once);
  // same bits whatever the split, so the choice by batch size keeps batch invariance.  RVIT_CONV1_ROLL=0: A/B runs.
  static const bool roll = [] { const char* e = getenv("RVIT_CONV1_ROLL"); return !e || atoi(e) != 0; }();
  if (img_f32 && relu && roll) {
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        rvit_set_error("stem_conv1_pool: cannot query the device");
        return RVIT_ELAUNCH;
      }
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    static const int divisors[6] = {1, 2, 4, 7, 14, 28};
    int wpi = 28;
    for (int d : divisors)
      if ((long)B * d >= 2L * n_cu) { wpi = d; break; }
    static const int wpi_env = [] { const char* e = getenv("RVIT_CONV1_WPI"); return e ? atoi(e) : 0; }();   // experiments
    if (wpi_env > 0 && 28 % wpi_env == 0) wpi = wpi_env;
    static RvitLdsLimit lim_roll;
    if (const int rc = lim_roll.ensure({(const void*)conv1_pool_roll_kernel}, ROLL_LDS, "stem_conv1_pool")) return rc;
    hipLaunchKernelGGL(conv1_pool_roll_kernel, dim3((unsigned)(B * wpi)), dim3(512), ROLL_LDS, st, (const float*)img,
                       (const bf16*)w, scale, bias, (bf16*)pooled, wpi);
    RVIT_CHECK_LAUNCH("stem_conv1_pool");
    return RVIT_OK;
  }
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv1_pool_kernel<false>, (const void*)conv1_pool_kernel<true>}, CO_BYTES,
                                "stem_conv1_pool"))
    return rc;
  if (img_f32)
    hipLaunchKernelGGL(conv1_pool_kernel<true>, dim3((unsigned)(B * (HQ / TP))), dim3(512), CO_BYTES, st, img,
                       (const bf16*)w, scale, bias, (bf16*)pooled, relu);
  else
    hipLaunchKernelGGL(conv1_pool_kernel<false>, dim3((unsigned)(B * (HQ / TP))), dim3(512), CO_BYTES, st, img,
                       (const bf16*)w, scale, bias, (bf16*)pooled, relu);
  RVIT_CHECK_LAUNCH("stem_conv1_pool");
  return RVIT_OK;
}

#ifdef RVIT_C1_STAMPS
extern "C" int rvit_debug_c1_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c1_stamps), (size_t)n * 8);
}
#endif

int rvit_stem_conv1_rows_impl(const float* x, const void* w, void* y, float* stat_ws, long stat_ws_bytes, int* stat_slots,
                              int B, hipStream_t st) {
  RVIT_CHECK_ARG(x && w && y && stat_ws && stat_slots && B > 0, "stem_conv1_rows: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w) && rvit_aligned16(y) && rvit_aligned16(stat_ws),
                 "stem_conv1_rows: operands must be 16-byte aligned");
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      n_cu <= 0)
    n_cu = 256;
  static const int divisors[6] = {1, 2, 4, 7, 14, 28};
  int wpi = 28;
  for (int d : divisors)
    if ((long)B * d >= 2L * n_cu) { wpi = d; break; }
  RVIT_CHECK_ARG((long)B * wpi * 128 * 4 <= stat_ws_bytes, "stem_conv1_rows: the statistics workspace holds %ld bytes, %ld needed",
                 stat_ws_bytes, (long)B * wpi * 512);
  RVIT_CHECK_ARG((long)B * wpi < 0x7fffffffL, "stem_conv1_rows: batch too large");
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv1_rows_roll_kernel}, ROLL_LDS, "stem_conv1_rows")) return rc;
  hipLaunchKernelGGL(conv1_rows_roll_kernel, dim3((unsigned)(B * wpi)), dim3(512), ROLL_LDS, st, x, (const bf16*)w, (bf16*)y,
                     stat_ws, wpi);
  RVIT_CHECK_LAUNCH("stem_conv1_rows");
  *stat_slots = B * wpi;
  return RVIT_OK;
}

// dW[64][224] += the weight gradient of the stem's 7x7 conv; ws: one slot of 64 x 224 floats per workgroup
int rvit_stem_conv1_wgrad_impl(const float* x, const void* dy, float* dw, float* ws, long ws_bytes, int B, hipStream_t st) {
  RVIT_CHECK_ARG(x && dy && dw && ws && B > 0, "stem_conv1_wgrad: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(dy) && rvit_aligned16(dw) && rvit_aligned16(ws),
                 "stem_conv1_wgrad: operands must be 16-byte aligned");
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      n_cu <= 0)
    n_cu = 256;
  static const int divisors[6] = {1, 2, 4, 7, 14, 28};
  int wpi = 28;
  for (int d : divisors)
    if ((long)B * d >= 2L * n_cu) { wpi = d; break; }
  // fewer workgroups per image when the workspace is short of slots
  while (wpi > 1 && (long)B * wpi * 64 * 224 * 4 > ws_bytes) {
    int next = 1;
    for (int d : divisors)
      if (d < wpi) next = d;
    wpi = next;
  }
  RVIT_CHECK_ARG((long)B * wpi * 64 * 224 * 4 <= ws_bytes, "stem_conv1_wgrad: the workspace holds %ld bytes, %ld needed", ws_bytes,
                 (long)B * wpi * 64 * 224 * 4);
  RVIT_CHECK_ARG((long)B * wpi < 0x7fffffffL, "stem_conv1_wgrad: batch too large");
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv1_wgrad_roll_kernel}, WG_LDS, "stem_conv1_wgrad")) return rc;
  hipLaunchKernelGGL(conv1_wgrad_roll_kernel, dim3((unsigned)(B * wpi)), dim3(512), WG_LDS, st, x, (const bf16*)dy, ws, wpi);
  RVIT_CHECK_LAUNCH("stem_conv1_wgrad");
  return rvit_ordered_finalize(ws, B * wpi, 64L * 224, dw, 64, 224, 224, st);
}
