// 3x3 / stride 1 / pad 1 NHWC convolution for 64 -> 64 channels on 56-wide maps with the WEIGHTS RESIDENT in LDS.
//
// The four 3x3 convs of resnet18's layer1 (`self.cnn_backbone(x)[-1]`, train_vit.py:87,145-146) and, with `flip`,
// their input gradients.  These are the only convs of the stem whose whole filter bank fits a CU: 9 taps x 64 x 64
// bf16 = 72 KB.  The generic direct kernel (conv3x3.hip) spends ~13 us per 224-pixel tile on them (patch load,
// then nine taps each behind a weight-tile DMA and a workgroup barrier) against 1.7 us of MFMA work.  Here
// one persistent workgroup per CU loads the filter bank once, double-buffers the input patch of the NEXT tile
// (LDS = 72 KB + 2 x 44 KB = exactly 160 KB) and runs the 9 x 2 MFMA steps of a tile with no barrier inside: one
// workgroup barrier per tile, waves drift freely in between so LDS reads, MFMAs and the epilogue's global traffic
// of different waves overlap.
//
// Geometry: a tile is 4 whole image rows (224 output pixels, H % 4 == 0); its patch is padded rows h0-1 .. h0+4,
// padded columns -1 .. 56 (6 x 58 pixels, 128 B each, 16-byte chunks XOR-swizzled with (row>>1)&7 as in
// conv3x3.hip); out-of-image pixels are DMA'd from a zero page.  7 waves, each 32 pixels x 64 output channels
// (weights on the MFMA A side: 0.75 ds_read_b128 per MFMA).  The accumulation order (tap-major, two k steps) is
// the generic kernel's, so both produce the same bits.
//
// Epilogue in registers (no LDS left to restage through): after scale/bias a lane holds 4 consecutive channels for
// each of four 16-channel groups; `v_permlane16_swap` between lane rows exchanges halves so that every lane owns
// 8 consecutive channels (16 bytes) and a store instruction writes 64-byte runs; the residual is read in the
// same layout.
#include "common.h"
#include "rvit_internal.h"
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

__device__ uint4 g_zero_page_l1[4];

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}

constexpr int L1_W = 56, L1_PW = 58, L1_ROWS = 4, L1_TILE = L1_ROWS * L1_W;
constexpr int L1_CWAVES = 7;                                   // compute waves (32 pixels x 64 channels each)
constexpr int L1_WAVES = L1_CWAVES + 1;                        // + one producer wave
constexpr int L1_P = (L1_ROWS + 2) * L1_PW;                    // 348 patch pixels
constexpr int L1_PASSES = (L1_P + 7) / 8;                      // 44 wave-level DMA instructions of 1 KB
constexpr int L1_PATCH_BYTES = L1_PASSES * 1024;               // 45056
constexpr int L1_WPASSES = 9 * 8;                              // filter bank: 9 taps x 64 rows x 128 B
constexpr int L1_WBYTES = L1_WPASSES * 1024;                   // 73728
constexpr int L1_LDS = L1_WBYTES + 2 * L1_PATCH_BYTES;         // 163840 = 160 KB
static_assert(L1_TILE == 32 * L1_CWAVES, "one 32-pixel slab per compute wave");
static_assert(L1_LDS <= 160 * 1024, "LDS budget");

// compile-time loop over the 18 MFMA steps (the step index selects constant LDS offsets and the register buffer)
template <int S, class F>
__device__ __forceinline__ void l1_steps(F&& f) {
  if constexpr (S < 18) {
    f(std::integral_constant<int, S>{});
    l1_steps<S + 1>(f);
  }
}

// Patch DMA descriptor of patch pixel `row` for 16-byte slot `slot` of its LDS row: (element offset relative to the
// tile's first pixel, a multiple of 8) | class; class 0 = halo row above, 1 = inside, 2 = halo row below,
// 3 = always zero (column padding / past the patch).  The slot holds chunk slot ^ (row & 6): with chunk bits [2:1]
// XORed by row bits [2:1], any 16 consecutive rows are read conflict-free by ds_read_b128 (whose 16-lane groups
// pair rows {0-3,12-15} at chunk c with rows {4-11} at chunk c^1), whatever the first row's alignment.
__device__ __forceinline__ int l1_desc(int row, int slot) {
  if (row >= L1_P) return 3;
  const int pr = row / L1_PW, pc = row - pr * L1_PW;
  if (pc < 1 || pc > L1_W) return 3;
  const int chunk = slot ^ (row & 6);
  return (((pr - 1) * L1_W + (pc - 1)) * 64 + chunk * 8) | (pr == 0 ? 0 : (pr == L1_ROWS + 1 ? 2 : 1));
}

template <bool HAS_R, bool RELU, bool AFFINE, int STATS = 0>
__global__ __launch_bounds__(64 * L1_WAVES, 2) void conv3x3_c64_kernel(const GemmArgs p, int ntiles, int tiles_per_img) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wts = smem;
  char* const patch0 = smem + L1_WBYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int flip = p.transposed;
  const int G = gridDim.x;
  // Workgroup b runs on XCD b % 8: give each XCD a run of consecutive tiles per round, so that the two halo rows a
  // tile shares with its vertical neighbour are fetched by CUs behind the same L2 at about the same time.
  const int vb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page_l1);

#define L1_DMA(desc_, tt_, pass_, buf_)                                                                  \
  do {                                                                                                   \
    const int img_ = (tt_) / tiles_per_img, ti_ = (tt_) - img_ * tiles_per_img;                          \
    const int c_ = (desc_) & 3;                                                                          \
    const bool ok_ = c_ == 1 || (c_ == 0 && ti_ > 0) || (c_ == 2 && ti_ < tiles_per_img - 1);            \
    glds16(ok_ ? p.A + (long)(tt_) * (L1_TILE * 64) + ((desc_) & ~7) : zero_src,                         \
           patch0 + (buf_) * L1_PATCH_BYTES + (pass_) * 1024);                                           \
  } while (0)

  // ---- prologue, all 8 waves: filter bank (tap slot s <- source tap flip ? 8 - s : s; row = output channel) and
  //      the first tile's patch
  {
    const int row_l = lane >> 3, slot = lane & 7;
#pragma unroll 1
    for (int pw = wave; pw < L1_WPASSES; pw += L1_WAVES) {
      const int tap = pw >> 3, row = (pw & 7) * 8 + row_l;
      const int chunk = slot ^ ((row >> 1) & 7);
      const int stap = flip ? 8 - tap : tap;
      glds16(p.B + (long)row * p.ldb + stap * 64 + chunk * 8, wts + pw * 1024);
    }
    const int tt0 = vb;
    if (tt0 < ntiles) {
#pragma unroll 1
      for (int pass = wave; pass < L1_PASSES; pass += L1_WAVES) {
        const int d = l1_desc(pass * 8 + row_l, slot);
        L1_DMA(d, tt0, pass, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the prologue

  if (wave == L1_CWAVES) {
    // ---- producer wave: streams the NEXT tile's patch into the other buffer while the seven compute waves work.
    // An LDS-DMA instruction holds its wave for ~100 cycles; issued by the compute waves (all at once, behind the
    // per-tile barrier) those cycles were lost to the MFMA pipe.
    int desc[L1_PASSES];
#pragma unroll
    for (int pass = 0; pass < L1_PASSES; ++pass) desc[pass] = l1_desc(pass * 8 + (lane >> 3), lane & 7);
    int buf = 0;
#pragma unroll 1
    for (int tt = vb; tt < ntiles; tt += G, buf ^= 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the patch of tile tt landed ...
      __builtin_amdgcn_s_barrier();                      // ... and every compute wave finished reading the other buffer
      asm volatile("" ::: "memory");
      if (tt + G < ntiles) {
#pragma unroll
        for (int pass = 0; pass < L1_PASSES; ++pass) L1_DMA(desc[pass], tt + G, pass, buf ^ 1);
      }
    }
    return;
  }

  // ---- compute waves: fragment geometry (tile independent) ------------------------------------------------
  int pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = wave * 32 + i * 16 + fr;   // pixel of the tile
    const int h = q / L1_W, w = q - h * L1_W;
    pb[i] = h * L1_PW + w;                   // + kh * PW + kw = patch pixel of tap (kh, kw)
  }
  const int sw_b = (fr >> 1) & 7;
  const char* const wfrag = wts + fr * 128;

  // per-lane epilogue constants: channels j*16 + fq*4 + e
  float sc[4][4], bi[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = j * 16 + fq * 4;
    float4 a = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AFFINE) {
      if (p.scale) a = *reinterpret_cast<const float4*>(p.scale + n);
      if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n);
    }
    sc[j][0] = a.x; sc[j][1] = a.y; sc[j][2] = a.z; sc[j][3] = a.w;
    bi[j][0] = b.x; bi[j][1] = b.y; bi[j][2] = b.z; bi[j][3] = b.w;
  }
  // after the row swap a lane owns channels pair*32 + (fq&1)*16 + (fq>>1)*8 .. +7 of pixel fr
  const int co_lane = (fq & 1) * 16 + (fq >> 1) * 8;

  float st1[2][8], st2[2][8];   // STATS: this lane's 16 channels over every pixel it stores, all tiles of the workgroup
  float bmean[2][8];            // STATS 2 (GemmArgs::bnr_y): the BatchNorm backward sums instead
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      st1[pr][e] = 0.f; st2[pr][e] = 0.f;
      bmean[pr][e] = STATS == 2 ? p.bnr_fsums[pr * 32 + co_lane + e] * p.bnr_inv_m : 0.f;
    }

  int buf = 0;
#pragma unroll 1
  for (int tt = vb; tt < ntiles; tt += G, buf ^= 1) {
    // the producer arrives here only after this tile's patch landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const long mbase = (long)tt * L1_TILE + wave * 32 + fr;
    uint4 res[2][2];
    if constexpr (HAS_R) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          res[i][pr] = *reinterpret_cast<const uint4*>((const bf16*)p.R + (mbase + i * 16) * 64 + pr * 32 + co_lane);
    }
    uint4 by[2][2];
    unsigned bm[2][2];
    if constexpr (STATS == 2) {   // the norm's input rows and ReLU bits of this tile, requested before its MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          by[i][pr] = *reinterpret_cast<const uint4*>(p.bnr_y + (mbase + i * 16) * 64 + pr * 32 + co_lane);
          bm[i][pr] = p.bnr_mask ? p.bnr_mask[(mbase + i * 16) * 8 + ((pr * 32 + co_lane) >> 3)] : 0xffu;
        }
    }

    const char* patch = patch0 + buf * L1_PATCH_BYTES;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 18 steps (9 taps x 2 k halves), fragments double-buffered in registers
    bf16x8 af[2][2], bw[2][4];
#define L1_LOAD_FRAGS(s_, d_)                                                                              \
  do {                                                                                                     \
    constexpr int tap_ = (s_) >> 1, ks_ = (s_) & 1, roff_ = (tap_ / 3) * L1_PW + (tap_ % 3);               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
      const int row = pb[i] + roff_;                                                                       \
      af[d_][i] = *reinterpret_cast<const bf16x8*>(patch + row * 128 + (((ks_ * 4 + fq) ^ (row & 6)) << 4)); \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
      bw[d_][j] = *reinterpret_cast<const bf16x8*>(wfrag + tap_ * 8192 + j * 2048 + (((ks_ * 4 + fq) ^ sw_b) << 4)); \
  } while (0)
    L1_LOAD_FRAGS(0, 0);
    l1_steps<0>([&](auto S) {
      constexpr int s = decltype(S)::value;
      // The compiler drains lgkmcnt to 0 before the first MFMA of a step (with LDS-DMA in the kernel it never emits
      // a counted wait), so the next step's six fragment reads go out right AFTER that MFMA: they have the other
      // seven MFMAs to land and the wait at the top of the next step finds them done.
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[s & 1][0], af[s & 1][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s + 1 < 18) L1_LOAD_FRAGS(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i + j > 0)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
#undef L1_LOAD_FRAGS

    // ---- epilogue ----------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][2 * pr][e], y = acc[i][2 * pr + 1][e];
          if constexpr (AFFINE) {
            x = x * sc[2 * pr][e] + bi[2 * pr][e];
            y = y * sc[2 * pr + 1][e] + bi[2 * pr + 1][e];
          }
          // odd lane rows of x <-> even lane rows of y
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
        }
        if constexpr (HAS_R) {
          const uint4 r = res[i][pr];
          v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
          v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
        }
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>((bf16*)p.C + (mbase + i * 16) * 64 + pr * 32 + co_lane) = o;
        if constexpr (STATS == 1) rvit_stat_acc8(o, st1[pr], st2[pr]);
        if constexpr (STATS == 2) rvit_bnr_acc8(o, by[i][pr], bm[i][pr], bmean[pr], st1[pr], st2[pr]);
      }
    }
  }
  if constexpr (STATS != 0) {
    // batch statistics of everything this workgroup stored (GemmArgs::stat_part; the conv of a conv -> BatchNorm pair in
    // a training step): the sixteen pixel lanes of a lane row on the vector ALU, the seven compute waves through LDS in
    // wave order, one slot per workgroup.  The producer wave has left: the barriers below count the compute waves only.
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st1[pr][e] = rvit_sum_row16(st1[pr][e]);
        st2[pr][e] = rvit_sum_row16(st2[pr][e]);
      }
    __builtin_amdgcn_s_barrier();   // every compute wave is past its last fragment read
    float* sred = reinterpret_cast<float*>(patch0);   // [wave][sum 64 | sumsq 64]
    if (fr == 0) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sred[wave * 128 + pr * 32 + co_lane + e] = st1[pr][e];
          sred[wave * 128 + 64 + pr * 32 + co_lane + e] = st2[pr][e];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t < 128) {
      float tot = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < L1_CWAVES; ++w_) tot += sred[w_ * 128 + t];
      if (STATS == 2 && t >= 64) tot *= rvit_bnr_rstd(p.bnr_fsums, 64, t - 64, p.bnr_inv_m, p.bnr_eps);
      p.stat_part[(long)blockIdx.x * 128 + t] = tot;
    }
  }
#undef L1_DMA
}

}  // namespace

bool rvit_conv3x3_c64_ok(const GemmArgs& a) {
  // on top of rvit_conv3x3_direct_ok (3x3 / stride 1 / pad 1, dense bf16 NHWC, M a whole number of images)
  // residual + scale/bias without ReLU (not a combination the model issues) does not fit the register budget
  if (a.R && a.act != RVIT_ACT_RELU && (a.scale || a.bias)) return false;
  return a.Cpix == 64 && a.N == 64 && a.W == L1_W && a.H >= L1_ROWS && a.H % L1_ROWS == 0 && a.ldb == 9 * 64 &&
         a.ldc == 64 && (!a.R || a.ldr == 64) && a.M % L1_TILE == 0 && a.M / L1_TILE < 0x7fffffffL;
}

template <bool HAS_R, bool RELU, bool AFFINE>
static int c64_launch(const GemmArgs& a, int n_cu, hipStream_t st) {
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv3x3_c64_kernel<HAS_R, RELU, AFFINE>}, L1_LDS, "conv3x3_c64")) return rc;
  const long ntiles = a.M / L1_TILE;
  const int grid = (int)(ntiles < n_cu ? ntiles : n_cu);
  if constexpr (!RELU && !AFFINE) {
    const int smode = rvit_conv_stat_mode(a, grid);   // one slot per workgroup
    if (smode != 0) {
      static RvitLdsLimit lim_s;
      if (const int rc = lim_s.ensure({(const void*)conv3x3_c64_kernel<HAS_R, false, false, 1>,
                                       (const void*)conv3x3_c64_kernel<HAS_R, false, false, 2>}, L1_LDS, "conv3x3_c64"))
        return rc;
      if (smode == 1)
        hipLaunchKernelGGL((conv3x3_c64_kernel<HAS_R, false, false, 1>), dim3(grid), dim3(64 * L1_WAVES), L1_LDS, st, a, (int)ntiles, a.H / L1_ROWS);
      else
        hipLaunchKernelGGL((conv3x3_c64_kernel<HAS_R, false, false, 2>), dim3(grid), dim3(64 * L1_WAVES), L1_LDS, st, a, (int)ntiles, a.H / L1_ROWS);
      RVIT_CHECK_LAUNCH("conv3x3_c64");
      *a.stat_slots = grid;
      return RVIT_OK;
    }
  }
  hipLaunchKernelGGL((conv3x3_c64_kernel<HAS_R, RELU, AFFINE>), dim3(grid), dim3(64 * L1_WAVES), L1_LDS, st, a, (int)ntiles, a.H / L1_ROWS);
  RVIT_CHECK_LAUNCH("conv3x3_c64");
  return RVIT_OK;
}

int rvit_conv3x3_c64(const GemmArgs& a, hipStream_t st) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      rvit_set_error("conv3x3_c64: cannot query the device");
      return RVIT_ELAUNCH;
    }
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (a.M / L1_TILE <= 0) return RVIT_OK;
  const bool relu = a.act == RVIT_ACT_RELU;
  const bool affine = a.scale || a.bias;
#define C64_CASE(r_, u_, f_) \
  if ((a.R != nullptr) == r_ && relu == u_ && affine == f_) return c64_launch<r_, u_, f_>(a, n_cu, st);
  C64_CASE(false, false, false) C64_CASE(false, false, true) C64_CASE(false, true, false) C64_CASE(false, true, true)
  C64_CASE(true, false, false) C64_CASE(true, true, false) C64_CASE(true, true, true)
#undef C64_CASE
  return RVIT_EUNSUPPORTED;
}
