// Direct weight gradient of the 3x3 / stride 1 / pad 1 convolutions with the input patch RESIDENT in LDS:
//   dW[co][kh][kw][ci] += sum_m dY[m][co] * X[pixel(m) + (kh-1, kw-1)][ci]
// (autograd of the 13 stride-1 3x3 convs of the resnet stem, train_vit.py:87,939).  The implicit-GEMM kernel
// (wgrad.hip) re-gathers the activation once per tap: 32 KB of L2 -> LDS fill per M MAC, and that fill, not the
// MFMA pipe, bounds it (profiles/r01_conv3x3_pingpong_experiment.txt).  Here a workgroup owns one (64 output
// channels x 64 input channels) pair and ALL NINE taps: per tile of <= 128 output pixels it stages the dY rows
// (16 KB) and the padded input rows they touch (<= 40 KB) once -- 9.6 KB per M MAC -- and NINE waves, one per
// tap, contract over the pixels with the hardware-transposing LDS read (ds_read_b64_tr_b16) by shifting the
// patch row, 64 x 64 fp32 accumulators each, which persist across the workgroup's whole pixel range.  Tiles are
// staged through a 3-deep ring (the DMAs of tiles t+1 and t+2 run under the MFMAs of tile t: one tile of MFMAs is
// shorter than the memory latency); the M range is split over workgroups and the
// 9 x 64 x 64 partial sums are added with fp32 atomics in whole 256-byte rows.
//
// LDS rows are 128 B (64 bf16); the 32-byte segment index is XOR-ed with f(row) = bit1(row) | bit3(row) << 1 on
// the DMA source and on the read, so the 16 rows x 32 B a wave's transposing read touches cover every bank
// exactly twice (the minimum for 512 B).  Geometry (padded-row space, tiles of whole image rows on wide maps) as
// in conv3x3.hip.
#include <climits>
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ uint4 g_zero_page_w3[4];

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }  // on 16-B chunks

constexpr int WD_TL = 128;                       // pixel rows of a dY tile (4 MFMA K steps of 32)
constexpr int WD_PMAX = 256;                     // patch pixels per buffer (bound; the launch sizes the buffers)
constexpr int WD_WAVES = 9;
constexpr int WD_PASSES = (WD_PMAX / 8 + WD_TL / 8 + WD_WAVES - 1) / WD_WAVES;  // 1 KB DMA passes per wave per tile
constexpr int WD_STAGES = 3;                     // tiles in the ring: two DMAs in flight under one tile of MFMAs

struct W3Args {
  const bf16* dY;   // [M, Cout]
  const bf16* X;    // [B, H, W, Cin]
  float* dW;        // [Cout, 9 * Cin]
  long M;
  int H, W, Cin, Cout;
  int tile_len;     // output pixels per tile (<= 128)
  int ntiles;       // ceil(M / tile_len)
  int nsplits;      // workgroups per (co, ci) pair
  int patch_bytes;  // bytes of the patch part of one ring stage (multiple of 1024); the dY tile follows it
  float* part;      // ordered mode: split s stores its [Cout][9 Cin] partial at part + s * part_stride (no atomics)
  long part_stride;
};

// ALIGNED: tiles are whole image rows (tile_len = rows * W, rows | H).  Then every tile has the same patch shape and
// the same pixel -> patch-row map: the per-lane DMA source offsets and fragment row addresses are computed ONCE;
// per tile only the tile's base pointer and the image-border row test remain (the first version recomputed both
// with integer divisions per tile and was bound by that address arithmetic on the vector ALU: ablation in
// profiles/r01_wgrad3x3_notes.txt).  The general form (tiles of 128 consecutive pixels on maps narrower than 14)
// keeps the per-tile computation; those layers are small.
template <bool ALIGNED>
__global__ __launch_bounds__(64 * WD_WAVES, 1) void conv3x3_wgrad_direct_kernel(const W3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;   // wave = tap
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout, PW = W + 2, HP = H + 2, hw = H * W;
  const int ncb = Cin >> 6;
  const int pair = blockIdx.x / p.nsplits, split = blockIdx.x - pair * p.nsplits;
  const int co0 = (pair / ncb) * 64, ci0 = (pair % ncb) * 64;
  const int per = (p.ntiles + p.nsplits - 1) / p.nsplits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < p.ntiles ? t_begin + per : p.ntiles;
  if (t_begin >= t_end) return;
  const int nB = (int)(p.M / hw);
  const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)W, inv_pw = 1.0f / (float)PW, inv_hp = 1.0f / (float)HP;
  auto fdiv = [](int m, int d, float inv_d) -> int {   // exact floor division for the ranges checked by the host
    int q = (int)((float)m * inv_d);
    int r = m - q * d;
    if (r < 0) --q;
    if (r >= d) ++q;
    return q;
  };
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_page_w3);
  const int prow_l = lane >> 3, slot = lane & 7;
  const int buf_bytes = p.patch_bytes + WD_TL * 128;
  char* const dump = smem + WD_STAGES * buf_bytes;
  const int tile_rows = p.tile_len / W;                       // ALIGNED only
  const int P_al = (tile_rows + 2) * PW, np_al = (P_al + 7) >> 3;

  // ---- per-lane DMA descriptors of the ALIGNED form: for pass it, the LDS offset inside a stage (-1: dump), the
  // source offset relative to the tile's first pixel (elements), the patch row (border test) or -1 for dY rows,
  // INT_MIN offset = always zero-filled
  int d_dst[WD_PASSES], d_off[WD_PASSES];   // d_dst: stage offset (multiple of 1024) | (patch row + 1) in the low bits
  if constexpr (ALIGNED) {
#pragma unroll
    for (int it = 0; it < WD_PASSES; ++it) {
      const int pass = wave + it * WD_WAVES;
      d_dst[it] = -1; d_off[it] = INT_MIN;
      if (pass < np_al) {
        const int row = pass * 8 + prow_l;
        d_dst[it] = pass * 1024;
        if (row < P_al) {
          const int pr = row / PW, pc = row - pr * PW;
          if (pc >= 1 && pc <= W) {
            d_off[it] = ((pr - 1) * W + pc - 1) * Cin + ci0 + ((slot ^ swz(row)) << 3);
            d_dst[it] |= pr + 1;
          }
        }
      } else if (pass < np_al + WD_TL / 8) {
        const int ypass = pass - np_al, row = ypass * 8 + prow_l;
        d_dst[it] = p.patch_bytes + ypass * 1024;
        if (row < p.tile_len) d_off[it] = row * Cout + co0 + ((slot ^ swz(row)) << 3);
      }
    }
  }

  // Every wave issues exactly WD_PASSES DMA instructions per tile (passes beyond the tile's data, and tiles beyond
  // the workgroup's range, land in a 1 KB dump area) so that `s_waitcnt vmcnt(WD_PASSES)` means "all but the
  // newest tile have landed".
  auto issue_tile = [&](int tile, int buf) {
    const bool live = tile < t_end;
    char* base = smem + buf * buf_bytes;
    const long m0 = (long)tile * p.tile_len;
    if constexpr (ALIGNED) {
      const int b0 = (int)(m0 / hw), h0 = ((int)m0 - b0 * hw) / W;     // wave-uniform
      const bf16* xb = p.X + m0 * Cin;
      const bf16* yb = p.dY + m0 * Cout;
#pragma unroll
      for (int it = 0; it < WD_PASSES; ++it) {
        const bf16* src = zero_src;
        const int pr1 = d_dst[it] & 1023;     // 0: a dY row (or nothing)
        if (live && d_off[it] != INT_MIN) {
          if (pr1 == 0) src = yb + d_off[it];
          else if ((unsigned)(h0 + pr1 - 2) < (unsigned)H) src = xb + d_off[it];
        }
        glds16(src, (live && d_dst[it] >= 0) ? base + (d_dst[it] & ~1023) : dump);
      }
    } else {
      long m_last = m0 + p.tile_len - 1;
      if (m_last >= p.M) m_last = p.M - 1;
      int R0 = 0, P = 0;
      if (live) {
        const int b0 = fdiv((int)m0, hw, inv_hw), r0 = (int)m0 - b0 * hw;
        const int b1 = fdiv((int)m_last, hw, inv_hw), r1 = (int)m_last - b1 * hw;
        const int R_first = b0 * HP + fdiv(r0, W, inv_w) + 1, R_last = b1 * HP + fdiv(r1, W, inv_w) + 1;
        R0 = R_first - 1;
        P = (R_last - R_first + 3) * PW;
      }
      const int np = (P + 7) >> 3;                 // patch passes; then WD_TL / 8 dY passes
#pragma unroll 1
      for (int it = 0; it < WD_PASSES; ++it) {
        const int pass = wave + it * WD_WAVES;
        const bf16* src = zero_src;
        char* dst = dump;
        if (live && pass < np) {
          const int row = pass * 8 + prow_l;
          if (row < P) {
            const int pr = fdiv(row, PW, inv_pw), pc = row - pr * PW;
            const int Rp = R0 + pr;
            const int b = fdiv(Rp, HP, inv_hp), hh = Rp - b * HP - 1, ww = pc - 1;
            if (b < nB && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
              src = p.X + ((long)(b * H + hh) * W + ww) * Cin + ci0 + ((slot ^ swz(row)) << 3);
          }
          dst = base + pass * 1024;
        } else if (live && pass < np + WD_TL / 8) {
          const int ypass = pass - np;
          const int row = ypass * 8 + prow_l;
          const long m = m0 + row;
          if (row < p.tile_len && m <= m_last) src = p.dY + m * Cout + co0 + ((slot ^ swz(row)) << 3);
          dst = base + p.patch_bytes + ypass * 1024;
        }
        glds16(src, dst);
      }
    }
  };

  // transposing-read geometry: within a 16-lane group lane (q, pp) supplies row q, columns 4 pp .. 4 pp + 3 of a
  // 4 x 16 block and receives rows 0..3 of column (lane & 15)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int rd_lo = (pp >> 1) * 16 + (pp & 1) * 8;      // byte offset inside the 32-byte segment of a 16-column block
  const int roff = (wave / 3) * PW + (wave % 3);
  const int fr = lane & 15;
  // dY fragment rows ks*32 + 8g + q (+4): the swizzle term is the same for all of them
  const int yrow0 = 8 * g + q;
  const int s2y = swz(yrow0) >> 1;
  const int ybase = yrow0 * 128 + rd_lo;
  // patch fragment rows of this lane's pixels (ks*32 + 8g + q [+4]) + this wave's tap offset: byte address of the row
  // (+ rd_lo) and its 2-bit segment swizzle, 8 of each; tile-independent when ALIGNED
  int xrow_b[WD_TL / 32][2], xs2p = 0;      // xs2p: 2 bits per (ks, half)
  if constexpr (ALIGNED) {
#pragma unroll
    for (int ks = 0; ks < WD_TL / 32; ++ks)
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        int pl = ks * 32 + 8 * g + q + 4 * hlf;
        if (pl >= p.tile_len) pl = 0;            // dY row is zero there: any in-window patch row
        const int hl = pl / W, w = pl - hl * W;
        const int prow = hl * PW + w + roff;
        xrow_b[ks][hlf] = prow * 128 + rd_lo;
        xs2p |= (swz(prow) >> 1) << (4 * ks + 2 * hlf);
      }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_tile(t_begin, 0);
  issue_tile(t_begin + 1, 1);
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WD_PASSES) : "memory");   // this tile's loads (the older of the two in flight)
    __builtin_amdgcn_s_barrier();   // ... of every wave; and everyone has left the stage that tile + 2 overwrites
    issue_tile(tile + 2, buf == 0 ? 2 : buf - 1);
    const char* sP = smem + buf * buf_bytes;
    const char* sY = sP + p.patch_bytes;
    if constexpr (!ALIGNED) {   // per-tile pixel -> patch-row map
      xs2p = 0;
      const long m0 = (long)tile * p.tile_len;
      long m_last = m0 + p.tile_len - 1;
      if (m_last >= p.M) m_last = p.M - 1;
      const int b0 = fdiv((int)m0, hw, inv_hw), r0 = (int)m0 - b0 * hw;
      const int R_first = b0 * HP + fdiv(r0, W, inv_w) + 1;
#pragma unroll
      for (int ks = 0; ks < WD_TL / 32; ++ks)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          long m = m0 + ks * 32 + 8 * g + q + 4 * hlf;
          if (m > m_last) m = m0;
          const int mi = (int)m;
          const int b = fdiv(mi, hw, inv_hw), r = mi - b * hw;
          const int h = fdiv(r, W, inv_w), w = r - h * W;
          const int prow = (b * HP + h + 1 - R_first) * PW + w + roff;
          xrow_b[ks][hlf] = prow * 128 + rd_lo;
          xs2p |= (swz(prow) >> 1) << (4 * ks + 2 * hlf);
        }
    }
    // keep the (cheap) segment arithmetic inside the loop: hoisted, its 48 addresses would not fit the 168
    // registers that three waves per SIMD leave
    int s2x = xs2p, s2yv = s2y;
    asm volatile("" : "+v"(s2x), "+v"(s2yv));
#pragma unroll
    for (int ks = 0; ks < WD_TL / 32; ++ks) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 lo = tr_read(sP + xrow_b[ks][0] + ((j ^ ((s2x >> (4 * ks)) & 3)) << 5));
        const bf16x4 hi = tr_read(sP + xrow_b[ks][1] + ((j ^ ((s2x >> (4 * ks + 2)) & 3)) << 5));
        xf[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // one dY fragment at a time: 64 accumulator + 16 + 4 fragment registers
        const char* yp = sY + ybase + ks * 4096 + ((i ^ s2yv) << 5);
        const bf16x4 lo = tr_read(yp);
        const bf16x4 hi = tr_read(yp + 512);
        const bf16x8 yf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf[j], acc[i][j], 0, 0, 0);
      }
    }
    buf = buf == WD_STAGES - 1 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump-area loads of the two tiles past the range
  __builtin_amdgcn_s_barrier();   // all fragment reads done: the buffers become epilogue staging

  // ---- epilogue: lane holds D[co = 16 i + 4 g + r][ci = 16 j + (lane & 15)]; restage 16 rows at a time through a
  // wave-private LDS area and add whole 256-byte rows (64 ci of one co and tap) atomically
  constexpr int EP = 64 * 4 + 16;
  char* ep = smem + wave * (16 * EP);
  const long ldw = 9L * Cin;
  float* const dst = p.part ? p.part + (long)split * p.part_stride : p.dW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(ep + (4 * g + r) * EP + (j * 16 + fr) * 4) = acc[i][j][r];
    __syncthreads();
#pragma unroll 4
    for (int row = 0; row < 16; ++row) {
      const float v = *reinterpret_cast<const float*>(ep + row * EP + lane * 4);
      float* q = dst + (long)(co0 + i * 16 + row) * ldw + wave * Cin + ci0 + lane;
      if (p.part) *q = v;
      else atomicAdd(q, v);
    }
    __syncthreads();
  }
}

}  // namespace

// Tile length: whole image rows (<= 128 pixels) on wide maps so that a tile never straddles images, else 128
// consecutive pixels; the padded-row window it touches must fit WD_PMAX patch pixels.
static int w3_tile_len(int H, int W) {
  if (W >= 14) {
    for (int r = WD_TL / W; r >= 1; --r)
      if (H % r == 0) return r * W;
  }
  return WD_TL;
}
static int w3_patch_pixels(int H, int W, int tl) {
  if (W >= 14 && tl % W == 0) return (tl / W + 2) * (W + 2);
  const int rows = (tl - 1 + W - 1) / W + 2 * ((tl - 1) / (H * W) + 1) + 3;
  return rows * (W + 2);
}

bool rvit_conv3x3_wgrad_direct_ok(const WgradArgs& a) {
  if (a.mode != 1 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.seg_len != a.Cpix || a.K != 9 * a.Cpix) return false;
  if ((a.Cpix & 63) || (a.N & 63) || a.N_store != a.N || a.ldy != a.N || a.ldw != a.K || a.y_row_group) return false;
  if (a.H != a.Ho || a.W != a.Wo || a.W < 2 || a.M % ((long)a.H * a.W) != 0 || a.M >= 0x7fffffffL) return false;
  if ((long)(a.M / ((long)a.H * a.W)) * (a.H + 2) * (a.W + 2) >= (1L << 24)) return false;   // float-reciprocal division range
  const int tl = w3_tile_len(a.H, a.W);
  return w3_patch_pixels(a.H, a.W, tl) <= WD_PMAX;
}

// Does the direct kernel pay against the implicit-GEMM one?  Not on maps narrower than 14 pixels with >= 64 channel pairs
// (layer4 of the stem: 512 -> 512 on 7 x 7): its general form recomputes the pixel -> patch map per tile and every pair
// re-reads its slices; scripts/bench_wgrad3x3.py, B = 64 / 128 / 256 / 512: 57 / 83 / 137 / 247 us direct against
// 57 / 77 / 123 / 215 for the gather kernel on whole rounds of workgroups.  (Same sums in another order.)
bool rvit_conv3x3_wgrad_direct_pays(const WgradArgs& a) {
  return !(a.W < 14 && (long)(a.N / 64) * (a.Cpix / 64) >= 64);
}

int rvit_conv3x3_wgrad_direct(const WgradArgs& a, hipStream_t st) {
  RVIT_CHECK_ARG(rvit_conv3x3_wgrad_direct_ok(a), "conv3x3 wgrad (direct): unsupported geometry");
  W3Args p;
  p.dY = a.dY; p.X = a.X; p.dW = a.dW; p.M = a.M;
  p.H = a.H; p.W = a.W; p.Cin = a.Cpix; p.Cout = a.N;
  p.tile_len = w3_tile_len(a.H, a.W);
  p.ntiles = (int)((a.M + p.tile_len - 1) / p.tile_len);
  const int pairs = (a.N / 64) * (a.Cpix / 64);
  // one workgroup per CU (three stages of 34-48 KB): ~256 workgroups in total, but a split must own a few tiles to pay
  // for its 147 KB of atomics
  int splits = a.splits > 0 ? a.splits : (256 + pairs - 1) / pairs;
  if (splits > (p.ntiles + 3) / 4) splits = (p.ntiles + 3) / 4;
  if (splits < 1) splits = 1;
  p.part = nullptr;
  p.part_stride = (long)a.N * a.K;
  if (a.ws) {
    RVIT_CHECK_ARG(rvit_aligned16(a.ws), "conv3x3 wgrad: the ordered-reduction workspace must be 16-byte aligned");
    long sp = splits;
    if (const int rc = rvit_ordered_splits(p.ntiles, sp, 1, p.part_stride, a.ws_bytes, "conv3x3 wgrad", &sp)) return rc;
    splits = (int)sp;
    p.part = a.ws;
  }
  p.nsplits = splits;
  p.patch_bytes = ((w3_patch_pixels(a.H, a.W, p.tile_len) + 7) / 8) * 1024;
  const int lds = WD_STAGES * (p.patch_bytes + WD_TL * 128) + 1024;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)conv3x3_wgrad_direct_kernel<true>, (const void*)conv3x3_wgrad_direct_kernel<false>},
                                lds, "conv3x3 wgrad"))
    return rc;
  const bool aligned = a.W >= 14 && p.tile_len % a.W == 0 && a.H % (p.tile_len / a.W) == 0;
  if (aligned) hipLaunchKernelGGL(conv3x3_wgrad_direct_kernel<true>, dim3((unsigned)(pairs * splits)), dim3(64 * WD_WAVES), lds, st, p);
  else hipLaunchKernelGGL(conv3x3_wgrad_direct_kernel<false>, dim3((unsigned)(pairs * splits)), dim3(64 * WD_WAVES), lds, st, p);
  RVIT_CHECK_LAUNCH("conv3x3_wgrad_direct");
  if (p.part) return rvit_ordered_finalize(p.part, splits, p.part_stride, a.dW, a.N, a.K, a.ldw, st);
  return RVIT_OK;
}
