// Pre-LN transformer encoder layer for inference at embedding width 256, as TWO kernels that keep a tile of
// tokens on the CU from the LayerNorm to the residual add (torch transformer.py:945-950:
// x = x + sa(norm1(x)); x = x + ff(norm2(x)); functional.py:5819-5833, 6569-6637):
//
//   enc_mlp_kernel   x += linear2(gelu(linear1(LayerNorm(x))))         (transformer.py:980-982)
//   enc_attn_kernel  x += out_proj(softmax(q k^T / sqrt(64)) v), q|k|v = in_proj(LayerNorm(x))
//
// Why: at E = 256 the per-op path (layernorm -> gemm -> [attention] -> gemm) is bound by what it moves, not by
// MFMA: every K = 256 GEMM re-reads its A tile from L2 once per 64 output columns and the 4E hidden map makes a
// bf16 round trip through HBM (per layer at 25,600 tokens: 104 MB hidden + 39 MB qkv + 26 MB LayerNorm outputs
// + 13 MB attention output against 26 MB for the fp32 residual stream itself).  Here a workgroup of 8 waves owns
// up to 112 tokens (mlp) or two whole images (attention; S <= 64 tokens each): the LayerNorm output is built in
// LDS as the MFMA operand image, the weights (1.5 MB per layer: L2-resident) stream from L2 straight into
// registers - each wave owns its own output columns, so a weight fragment has no second reader in the workgroup
// and an LDS copy would buy nothing - and q/k/v, the attention output and the GELU map live only in LDS.  HBM sees
// the residual stream once in, once out.
//
// Numerics: the arithmetic is that of the per-op kernels in the same order (LayerNorm as norm.hip; every product
// accumulated in fp32 over K ascending in 32-wide MFMA steps; bias, then residual; bf16 rounding of the LayerNorm
// output, q/k/v, attention output and GELU map exactly where the per-op path stores bf16), so both paths give
// the same bits (tests/test_encoder_gpu.py) and a frame's logits do not depend on which path its batch took.
//
// MFMA operand order as in gemm.hip: weights on the "A" side, tokens on the "B" side of
// v_mfma_f32_16x16x32_bf16, so a lane ends up with 4 consecutive output columns of one token: one 8-byte LDS
// write per tile for the bf16 maps, one float4 per tile for the residual stream.
#include "common.h"
#include "rvit_internal.h"

namespace {

constexpr int EN_E = 256;        // embedding width these kernels are built for
constexpr int EN_THREADS = 512;  // 8 waves: one workgroup per CU, two waves per SIMD
constexpr int EN_HC = 256;       // hidden columns per step of the mlp kernel

// Compile-time ablation for timing attribution (scripts/ablate_encoder.sh builds one library per mask; the shipped
// build has ENC_ABL == 0 and every test below is a constant false): 1 no GELU, 2 weight fragments loaded once,
// 4 no MFMA, 8 no LDS fragment reads, 16 no barrier in the loop, 32 no loop at all (LayerNorm + residual only).  Results are wrong under any of them.
#ifndef ENC_ABL
#define ENC_ABL 0
#endif
constexpr bool enc_abl(int bit) { return (ENC_ABL & bit) != 0; }

// LDS image of a [rows][256] bf16 MFMA operand: 512-byte rows, the low four bits of the 16-byte chunk index
// XOR-ed with row & 15.  A fragment read (ds_read_b128: lane -> row lane & 15, chunk 4 ks + (lane >> 4)) is served in
// 16-lane groups {rows 0-3, 12-15 at chunk c; rows 4-11 at chunk c ^ 1}: sixteen different chunks of a 256-byte
// bank line, conflict-free.
__device__ __forceinline__ int a_off(int row, int chunk) {
  return row * 512 + (((chunk & 16) | ((chunk ^ row) & 15)) << 4);
}

// Nothing is scheduled across this point: the software pipelines below (loads several steps ahead of their use)
// survive only if the compiler keeps the order they are written in; left alone it sinks every load next to its use.
#define ENC_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ bf16x8 ld_frag(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// LayerNorm of rows [row0, row0 + nrows) of x (fp32, 256 wide) into the operand image sA (ROWS rows, the rest zero):
// a wave per row, the row in registers, statistics as layernorm_fwd_kernel (norm.hip: the same xor butterfly in the
// same order, so the bf16 result is the same).  The loads are unconditional (rows past the end re-read row 0 and are
// zeroed afterwards) and the butterflies of all rows of a wave advance stage by stage, so that one wave has PER loads
// and then PER cross-lane exchanges in flight instead of one.
// S_img > 0: row r of the tile is token r % 64 of image r / 64 (tokens >= S_img: zero rows).
template <int ROWS>
__device__ __forceinline__ void ln_to_lds(const float* __restrict__ x, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float eps, long row0, int nrows, char* sA,
                                          int wave, int lane, int S_img) {
  constexpr int PER = ROWS / 8;
  float4 v[PER];
  bool ok[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = wave + 8 * i;
    int src = r;
    ok[i] = r < nrows;
    if (S_img > 0) {
      const int im = r >> 6, tk = r & 63;
      src = im * S_img + tk;
      ok[i] = tk < S_img && src < nrows;
    }
    v[i] = *reinterpret_cast<const float4*>(x + (row0 + (ok[i] ? src : 0)) * EN_E + lane * 4);
  }
  ENC_PIN();
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 b = *reinterpret_cast<const float4*>(beta + lane * 4);
  float s[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) s[i] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) tmp[i] = __shfl_xor(s[i], o, 64);
    ENC_PIN();
#pragma unroll
    for (int i = 0; i < PER; ++i) s[i] += tmp[i];
    ENC_PIN();
  }
  float q[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float mean = ln_mean(s[i], 1.0f / (float)EN_E);
    v[i].x = __fsub_rn(v[i].x, mean); v[i].y = __fsub_rn(v[i].y, mean);
    v[i].z = __fsub_rn(v[i].z, mean); v[i].w = __fsub_rn(v[i].w, mean);
    q[i] = ln_sq(ln_sq(ln_sq(ln_sq(0.f, v[i].x), v[i].y), v[i].z), v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) tmp[i] = __shfl_xor(q[i], o, 64);
    ENC_PIN();
#pragma unroll
    for (int i = 0; i < PER; ++i) q[i] += tmp[i];
    ENC_PIN();
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = wave + 8 * i;
    const float rstd = ln_rstd(q[i], 1.0f / (float)EN_E, eps);
    uint2 o;
    o.x = pack_bf16x2(ln_affine(v[i].x, rstd, g.x, b.x), ln_affine(v[i].y, rstd, g.y, b.y));
    o.y = pack_bf16x2(ln_affine(v[i].z, rstd, g.z, b.z), ln_affine(v[i].w, rstd, g.w, b.w));
    if (!ok[i]) o = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(sA + a_off(r, lane >> 1) + (lane & 1) * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// x += linear2(gelu(linear1(LayerNorm(x)))), in place on the fp32 residual stream.  A workgroup owns `rows_per_wg`
// <= 16 MT consecutive tokens; its LayerNorm output sits in LDS for the whole kernel.  Per step of 256 hidden columns:
//   phase 1  wave w computes hidden columns 32 w .. + 31 of every token (K = 256: 8 MFMA steps of 2 MT tiles),
//            adds the bias, applies GELU and writes the bf16 map into the second LDS image;
//   phase 2  wave w accumulates output columns 32 w .. + 31 of every token over these 256 hidden columns.
// Each token fragment read from LDS feeds two MFMAs (LDS time is half the MFMA time) and the fragments of step
// ks + 1 are read while step ks multiplies.  The weights are one linear stream per wave (16 steps of two fragments
// per 256 hidden columns) kept three to four steps ahead in a register ring of four steps, across the phase and
// step boundaries.  The ring is refilled TWO steps at a time: a fragment load takes 64 bytes of each of 16 weight rows,
// the next step's takes the other half of the same 128-byte lines, and issued back to back the second one hits in
// the vector L1; a step apart (8 waves x 8 KB of lines in flight against 32 KB of L1) it fetched every line twice.
template <int MT>
__device__ __forceinline__ void enc_mlp_body(float* __restrict__ x, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, const bf16* __restrict__ w1,
                                             const float* __restrict__ b1, const bf16* __restrict__ w2,
                                             const float* __restrict__ b2, const long row0, const int nrows, int hidden) {
  constexpr int ROWS = MT * 16;
  constexpr int RING = 4;
  // the dynamic LDS is named here, not passed in: through a generic pointer parameter the non-inlined calls of the
  // stack kernel turned every LDS access into a flat one (2.3x slower)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sH = smem + ROWS * 512;

  // The thread index passes through an empty asm: inside the stack kernel's layer loop nothing derived from it can be
  // hoisted out of the loop and kept alive across the OTHER half (that cost 157 spilled registers); each half keeps
  // the register allocation it has as a kernel of its own.
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // weight stream of this wave: step s < 8 of chunk j: linear1 rows j*256 + 32 wave + 16 nt + fr, K offset 32 s;
  // step s >= 8: linear2 rows 32 wave + 16 nt + fr, K offset j*256 + 32 (s - 8)
  const bf16* const w1p = w1 + (long)(wave * 32 + fr) * EN_E + fq * 8;
  const bf16* const w2p = w2 + (long)(wave * 32 + fr) * hidden + fq * 8;
  const long w2nt = 16L * hidden;
  bf16x8 wr[RING][2];
#define ENC_W_LOAD(slot, jj, ss)                                                                        \
  do {                                                                                                  \
    if ((ss) < 8) {                                                                                     \
      wr[slot][0] = ld_frag(w1p + (long)(jj) * (EN_HC * EN_E) + (ss) * 32);                             \
      wr[slot][1] = ld_frag(w1p + (long)(jj) * (EN_HC * EN_E) + 16 * EN_E + (ss) * 32);                 \
    } else {                                                                                            \
      wr[slot][0] = ld_frag(w2p + (jj) * EN_HC + ((ss) - 8) * 32);                                      \
      wr[slot][1] = ld_frag(w2p + w2nt + (jj) * EN_HC + ((ss) - 8) * 32);                               \
    }                                                                                                   \
  } while (0)
  // the first steps go out before the LayerNorm: they do not depend on it
#pragma unroll
  for (int s = 0; s < RING; ++s) ENC_W_LOAD(s, 0, s);

  ln_to_lds<ROWS>(x, gamma, beta, eps, row0, nrows, sA, wave, lane, 0);
  __syncthreads();

  f32x4 accY[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) accY[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nch = enc_abl(32) ? 0 : hidden / EN_HC;
  for (int j = 0; j < nch; ++j) {
    const int jn = j + 1 < nch ? j + 1 : 0;  // the ring runs past the last step into step 0 again: four unused loads
    // ---- phase 1 --------------------------------------------------------------------------------------
    f32x4 acc1[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc1[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = lds_frag(sA + a_off(i * 16 + fr, fq));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[(s + 1) & 1][i] = enc_abl(8) ? wr[(s + i) & 3][i & 1] : lds_frag(sA + a_off(i * 16 + fr, (s + 1) * 4 + fq));
      }
      ENC_PIN();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (enc_abl(4)) acc1[i][nt][0] += (float)af[s & 1][i][nt] * (float)wr[s & 3][nt][i & 7];
          else acc1[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s & 3][nt], af[s & 1][i], acc1[i][nt], 0, 0, 0);
        }
      ENC_PIN();
      if (!enc_abl(2) && (s & 1)) {
        ENC_W_LOAD((s - 1) & 3, j, s - 1 + RING);
        ENC_W_LOAD(s & 3, j, s + RING);
      }
      ENC_PIN();
    }
    // ---- bias, GELU, bf16 map -> sH ---------------------------------------------------------------
    if (!enc_abl(16)) __syncthreads();  // every wave has finished phase 2 of the previous step (reads of sH)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float4 bb = *reinterpret_cast<const float4*>(b1 + j * EN_HC + wave * 32 + nt * 16 + fq * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        uint2 o;
        if (enc_abl(1)) {
          o.x = pack_bf16x2(acc1[i][nt][0] + bb.x, acc1[i][nt][1] + bb.y);
          o.y = pack_bf16x2(acc1[i][nt][2] + bb.z, acc1[i][nt][3] + bb.w);
        } else {
          o.x = pack_bf16x2(gelu_erf(acc1[i][nt][0] + bb.x), gelu_erf(acc1[i][nt][1] + bb.y));
          o.y = pack_bf16x2(gelu_erf(acc1[i][nt][2] + bb.z), gelu_erf(acc1[i][nt][3] + bb.w));
        }
        *reinterpret_cast<uint2*>(sH + a_off(i * 16 + fr, wave * 4 + nt * 2 + (fq >> 1)) + (fq & 1) * 8) = o;
      }
    }
    if (!enc_abl(16)) __syncthreads();
    // ---- phase 2 --------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = lds_frag(sH + a_off(i * 16 + fr, fq));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[(s + 1) & 1][i] = enc_abl(8) ? wr[(s + i) & 3][i & 1] : lds_frag(sH + a_off(i * 16 + fr, (s + 1) * 4 + fq));
      }
      ENC_PIN();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (enc_abl(4)) accY[i][nt][0] += (float)af[s & 1][i][nt] * (float)wr[s & 3][nt][i & 7];
          else accY[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s & 3][nt], af[s & 1][i], accY[i][nt], 0, 0, 0);
        }
      ENC_PIN();
      if (!enc_abl(2) && (s & 1)) {
        if (s + 8 + RING < 16) {
          ENC_W_LOAD((s - 1) & 3, j, s - 1 + 8 + RING);
          ENC_W_LOAD(s & 3, j, s + 8 + RING);
        } else {
          ENC_W_LOAD((s - 1) & 3, jn, s - 1 + 8 + RING - 16);
          ENC_W_LOAD(s & 3, jn, s + 8 + RING - 16);
        }
      }
      ENC_PIN();
    }
  }
#undef ENC_W_LOAD

  // ---- bias, residual, store: all residual loads first (clamped rows), then the stores ---------------------------
  // v_permlane16_swap between lane rows of the wave's two 16-column blocks (fq 0<->1, 2<->3, as gemm8p.hip's epilogue)
  // leaves every lane 8 consecutive columns of token row fr: the four lanes of a row read and write its WHOLE 128-byte
  // line (round 2 wrote two 64-byte halves per line from two instructions: 43 % more write requests than bytes
  // needed, profiles/r02_pmc_traffic.json).  Same sums in the same order: the bits do not change.
  {
    const int col = wave * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(b2 + col), b1v = *reinterpret_cast<const float4*>(b2 + col + 4);
    float4 xr[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int r = i * 16 + fr;
      const float* xp = x + (row0 + (r < nrows ? r : 0)) * EN_E + col;
      xr[i][0] = *reinterpret_cast<const float4*>(xp);
      xr[i][1] = *reinterpret_cast<const float4*>(xp + 4);
    }
    ENC_PIN();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int r = i * 16 + fr;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(accY[i][0][e]), __float_as_uint(accY[i][1][e]), false, false);
        v[e] = __uint_as_float(sw2[0]);
        v[4 + e] = __uint_as_float(sw2[1]);
      }
      float4 o0, o1;
      o0.x = (v[0] + b0.x) + xr[i][0].x; o0.y = (v[1] + b0.y) + xr[i][0].y;
      o0.z = (v[2] + b0.z) + xr[i][0].z; o0.w = (v[3] + b0.w) + xr[i][0].w;
      o1.x = (v[4] + b1v.x) + xr[i][1].x; o1.y = (v[5] + b1v.y) + xr[i][1].y;
      o1.z = (v[6] + b1v.z) + xr[i][1].z; o1.w = (v[7] + b1v.w) + xr[i][1].w;
      if (r < nrows) {
        float* xp = x + (row0 + r) * EN_E + col;
        *reinterpret_cast<float4*>(xp) = o0;
        *reinterpret_cast<float4*>(xp + 4) = o1;
      }
    }
  }
}

template <int MT>
__global__ __launch_bounds__(EN_THREADS) void enc_mlp_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             const bf16* __restrict__ w1, const float* __restrict__ b1,
                                                             const bf16* __restrict__ w2, const float* __restrict__ b2,
                                                             long T, int rows_per_wg, int hidden) {
  const long row0 = (long)blockIdx.x * rows_per_wg;
  const long left = T - row0;
  const int nrows = left < rows_per_wg ? (int)left : rows_per_wg;
  enc_mlp_body<MT>(x, gamma, beta, eps, w1, b1, w2, b2, row0, nrows, hidden);
}

template <int MT>
int launch_mlp(float* x, const float* gamma, const float* beta, float eps, const bf16* w1, const float* b1,
               const bf16* w2, const float* b2, long T, int rows_per_wg, int hidden, hipStream_t st) {
  constexpr int ROWS = MT * 16;
  constexpr int LDS = 2 * ROWS * 512;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)enc_mlp_kernel<MT>}, LDS, "encoder_mlp")) return rc;
  const long grid = (T + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL((enc_mlp_kernel<MT>), dim3((unsigned)grid), dim3(EN_THREADS), LDS, st, x, gamma, beta, eps, w1, b1,
                     w2, b2, T, rows_per_wg, hidden);
  RVIT_CHECK_LAUNCH("encoder_mlp");
  return RVIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// x += out_proj(attention(in_proj(LayerNorm(x)))), in place on the fp32 residual stream, head_dim 64, H = 4, S <= 64.
// A workgroup owns TWO images: image i is rows 64 i .. 64 i + S - 1 of a 128-row tile (rows past S are zero rows of the
// LayerNorm image; their results are never stored).  Wave w = (wm, wn) = (w & 1, w >> 1).  Per head:
//   phase 1  q|k|v of this head for image wm: wave (wm, wn) owns q/k columns 32 (wn & 1) .. + 31 of q (wn < 2) or k
//            and v columns 16 wn .. + 15 (12 tiles, K = 256).  q and k go to LDS token-major (128-byte rows, the
//            swizzle of gemm.hip); v is computed with the MFMA operands swapped, so a lane holds 4 consecutive
//            TOKENS of one v column and v lands in LDS transposed (V^T[d][token], 272-byte rows) by 8-byte writes;
//   phase 2  wave (wm, wn) is queries 16 wn .. + 15 of image wm: S^T = K Q^T (keys on the accumulator rows), softmax
//            over the lane's 16 keys and the three lanes that hold the other keys of its query (2 shuffles), P^T
//            fed back from the accumulator registers as the "B" operand of O^T = V^T P^T: the 8 keys a lane holds
//            per MFMA step are {16 t + 4 g + e} of two key tiles t, and the V^T fragment is read as the same two
//            runs of 4 keys, so no LDS round trip and no transposing read; O (bf16, normalised) -> LDS token-major;
//   phase 3  wave (wm, wn) accumulates output columns 64 wn .. + 63 of image wm over this head's 64 columns of O.
// Two barriers per head.  q|k|v and out_proj products are accumulated exactly as rvit_gemm_bf16 does (same bits);
// the attention core uses 16x16x32 MFMAs where attn.hip uses 32x32x16, so scores and P V sums round differently
// in the last fp32 bit (tests/test_encoder_gpu.py bounds the difference).
__device__ __forceinline__ int k_off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
constexpr int EN_VT_PITCH = 272;  // bytes per V^T row: 128 tokens + 16: rows 4 banks apart, 8-byte accesses conflict-free
constexpr int EN_ATTN_LDS = 128 * 512 + 3 * 128 * 128 + 64 * EN_VT_PITCH;

// Attention core shared by enc_attn_kernel and attn16_kernel (same source, same bits): one wave = 16 queries of one
// (image, head).  sQ / sK: token-major 128-byte rows (k_off128), this image's rows start at `row0`, the wave's query
// row is `rowq`; vt: V^T of this image, row d at vt + d * VT_PITCH, key t at byte 2 t.  S^T = K Q^T with keys on the
// accumulator rows, softmax over the lane's 16 keys and the three lanes that hold the other keys of its query, the
// probabilities fed back from the accumulator registers as the "B" operand of O^T = V^T P^T (the 8 keys a lane holds
// per MFMA step are two runs of 4, read from V^T as the same two 8-byte runs).  w[dt]: 4 consecutive output columns
// 16 dt + 4 fq .. of the wave's query row fr, normalised, bf16.
template <int VT_PITCH>
__device__ __forceinline__ void attn16_core(const char* sQ, const char* sK, const char* vt, int row0, int rowq, int fr, int fq,
                                            int S, float scale_log2e, uint2 (&w)[4]) {
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = lds_frag(sQ + k_off128(rowq, ks * 4 + fq));
  f32x4 sc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 kf = lds_frag(sK + k_off128(row0 + kt * 16 + fr, ks * 4 + fq));
      sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sc[kt], 0, 0, 0);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (kt * 16 + fq * 4 + e >= S) sc[kt][e] = -INFINITY;
      mx = fmaxf(mx, sc[kt][e]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pe = __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(sc[kt][e], mx), scale_log2e));
      sc[kt][e] = pe;
      sum = __fadd_rn(sum, pe);
    }
  sum = __fadd_rn(sum, __shfl_xor(sum, 16, 64));
  sum = __fadd_rn(sum, __shfl_xor(sum, 32, 64));
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pb[e] = (bf16)sc[2 * kk][e];
      pb[4 + e] = (bf16)sc[2 * kk + 1][e];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const char* vrow = vt + (dt * 16 + fr) * VT_PITCH + fq * 8;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + (2 * kk) * 32);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + (2 * kk + 1) * 32);
      bf16x8 vf;
      vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
      vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb, o[dt], 0, 0, 0);
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    w[dt].x = pack_bf16x2(__fmul_rn(o[dt][0], inv), __fmul_rn(o[dt][1], inv));
    w[dt].y = pack_bf16x2(__fmul_rn(o[dt][2], inv), __fmul_rn(o[dt][3], inv));
  }
}

// The attention of the per-op path for S <= 64 (small batches, where the fused kernel's one workgroup per two images
// cannot fill the chip): one workgroup of four waves per (image, head), q / k / v read from the in_proj output
// [T, 3E], the core above, output bf16 [T, E].  Same bits as enc_attn_kernel's phase 2, so the engine may pick
// either path by batch size without a frame's logits depending on the batch it travels in.
constexpr int A16_VT_PITCH = 144;  // 64 keys + 16 bytes: rows 36 banks apart, 8-byte accesses conflict-free
__global__ __launch_bounds__(256) void attn16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int S, int H,
                                                     long ldq, long ldo, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char sQ[64 * 128];
  __shared__ __attribute__((aligned(16))) char sK[64 * 128];
  __shared__ __attribute__((aligned(16))) char sVt[64 * A16_VT_PITCH];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int E = H * 64;
  const bf16* base = qkv + (long)b * S * ldq + h * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = t + it * 256, row = id >> 3, c = id & 7;
    uint4 q4 = make_uint4(0, 0, 0, 0), k4 = q4, v4 = q4;
    if (row < S) {
      const bf16* p = base + (long)row * ldq + c * 8;
      q4 = *reinterpret_cast<const uint4*>(p);
      k4 = *reinterpret_cast<const uint4*>(p + E);
      v4 = *reinterpret_cast<const uint4*>(p + 2 * E);
    }
    *reinterpret_cast<uint4*>(sQ + k_off128(row, c)) = q4;
    *reinterpret_cast<uint4*>(sK + k_off128(row, c)) = k4;
    const unsigned short* ve = reinterpret_cast<const unsigned short*>(&v4);
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<unsigned short*>(sVt + (c * 8 + e) * A16_VT_PITCH + row * 2) = ve[e];
  }
  __syncthreads();
  if (wave * 16 >= S) return;
  const int rowq = wave * 16 + fr;
  uint2 w[4];
  attn16_core<A16_VT_PITCH>(sQ, sK, sVt, 0, rowq, fr, fq, S, scale_log2e, w);
  if (rowq < S) {
    bf16* orow = out + ((long)b * S + rowq) * ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<uint2*>(orow + dt * 16 + fq * 4) = w[dt];
  }
}

__device__ __forceinline__ void enc_attn_body(float* __restrict__ x, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, float eps, const bf16* __restrict__ wqkv,
                                              const float* __restrict__ bqkv, const bf16* __restrict__ wout,
                                              const float* __restrict__ bout, const int img0, int B, int S, float scale_log2e) {
  constexpr int H = EN_E / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;
  char* const sQ = smem + 128 * 512;
  char* const sK = sQ + 128 * 128;
  char* const sO = sK + 128 * 128;
  char* const sVt = sO + 128 * 128;

  int t = threadIdx.x;   // opaque, as in enc_mlp_body
  asm volatile("" : "+v"(t));
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int n_img = B - img0 < 2 ? B - img0 : 2;
  const long row0 = (long)img0 * S;
  const int nrows = n_img * S;

  // in_proj rows of this wave: slots 0, 1 = q (wn < 2) or k columns 32 (wn & 1) + 16 slot + fr; slot 2 = v column 16 wn + fr
  const bf16* const wqk = wqkv + ((long)((wn >> 1) * EN_E + (wn & 1) * 32 + fr)) * EN_E + fq * 8;
  const bf16* const wv = wqkv + ((long)(2 * EN_E + wn * 16 + fr)) * EN_E + fq * 8;
  bf16x8 ring[4][3];
#define ENC_QKV_LOAD(slot, hh, ss)                                             \
  do {                                                                         \
    ring[slot][0] = ld_frag(wqk + (long)(hh) * (64 * EN_E) + (ss) * 32);       \
    ring[slot][1] = ld_frag(wqk + (long)(hh) * (64 * EN_E) + 16 * EN_E + (ss) * 32); \
    ring[slot][2] = ld_frag(wv + (long)(hh) * (64 * EN_E) + (ss) * 32);        \
  } while (0)
#pragma unroll
  for (int s = 0; s < 4; ++s) ENC_QKV_LOAD(s, 0, s);

  ln_to_lds<128>(x, gamma, beta, eps, row0, nrows, sA, wave, lane, S);
  __syncthreads();

  f32x4 accY[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) accY[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int h = 0; h < H; ++h) {
    // ---- phase 1: q | k | v of head h ------------------------------------------------------------------
    f32x4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = lds_frag(sA + a_off(wm * 64 + i * 16 + fr, fq));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[(s + 1) & 1][i] = lds_frag(sA + a_off(wm * 64 + i * 16 + fr, (s + 1) * 4 + fq));
      }
      ENC_PIN();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s & 3][0], af[s & 1][i], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[s & 3][1], af[s & 1][i], acc[i][1], 0, 0, 0);
        acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s & 1][i], ring[s & 3][2], acc[i][2], 0, 0, 0);  // v: tokens on rows
      }
      ENC_PIN();
      if ((s & 1) && s + 4 < 8) {
        ENC_QKV_LOAD((s - 1) & 3, h, s - 1 + 4);
        ENC_QKV_LOAD(s & 3, h, s + 4);
      }
      ENC_PIN();
    }
    {
      char* const dst = wn < 2 ? sQ : sK;
      const float* const bqk = bqkv + (wn >> 1) * EN_E + h * 64 + (wn & 1) * 32 + fq * 4;
      const float bv = bqkv[2 * EN_E + h * 64 + wn * 16 + fr];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 bb = *reinterpret_cast<const float4*>(bqk + c * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint2 o;
          o.x = pack_bf16x2(acc[i][c][0] + bb.x, acc[i][c][1] + bb.y);
          o.y = pack_bf16x2(acc[i][c][2] + bb.z, acc[i][c][3] + bb.w);
          *reinterpret_cast<uint2*>(dst + k_off128(wm * 64 + i * 16 + fr, ((wn & 1) * 2 + c) * 2 + (fq >> 1)) + (fq & 1) * 8) = o;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 o;
        o.x = pack_bf16x2(acc[i][2][0] + bv, acc[i][2][1] + bv);
        o.y = pack_bf16x2(acc[i][2][2] + bv, acc[i][2][3] + bv);
        *reinterpret_cast<uint2*>(sVt + (wn * 16 + fr) * EN_VT_PITCH + (wm * 64 + i * 16 + fq * 4) * 2) = o;
      }
    }
    __syncthreads();

    // weights of phase 3 and of the next head's first steps go out before the attention core runs
    bf16x8 w3[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        w3[ks][nt] = ld_frag(wout + (long)(wn * 64 + nt * 16 + fr) * EN_E + h * 64 + ks * 32 + fq * 8);
    {
      const int hn = h + 1 < H ? h + 1 : 0;  // past the last head: four unused loads of head 0
#pragma unroll
      for (int s = 0; s < 4; ++s) ENC_QKV_LOAD(s, hn, s);
    }
    ENC_PIN();

    // ---- phase 2: queries 16 wn .. + 15 of image wm ------------------------------------------------------------
    {
      const int rowq = wm * 64 + wn * 16 + fr;
      uint2 w[4];
      attn16_core<EN_VT_PITCH>(sQ, sK, sVt + wm * 128, wm * 64, rowq, fr, fq, S, scale_log2e, w);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(sO + k_off128(rowq, dt * 2 + (fq >> 1)) + (fq & 1) * 8) = w[dt];
    }
    __syncthreads();

    // ---- phase 3: out_proj over this head's 64 columns ---------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 of[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) of[i] = lds_frag(sO + k_off128(wm * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          accY[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[ks][nt], of[i], accY[i][nt], 0, 0, 0);
    }
  }
#undef ENC_QKV_LOAD

  // ---- bias, residual, store (whole 128-byte lines per token row: lane swap as in enc_mlp_kernel) ---------------
  const bool img_ok = wm < n_img;
#pragma unroll
  for (int np = 0; np < 2; ++np) {
    const int col = wn * 64 + np * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(bout + col), b1v = *reinterpret_cast<const float4*>(bout + col + 4);
    float4 xr[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tk = i * 16 + fr;
      const bool ok = img_ok && tk < S;
      const float* xp = x + (row0 + (ok ? wm * S + tk : 0)) * EN_E + col;
      xr[i][0] = *reinterpret_cast<const float4*>(xp);
      xr[i][1] = *reinterpret_cast<const float4*>(xp + 4);
    }
    ENC_PIN();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tk = i * 16 + fr;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(accY[i][2 * np][e]), __float_as_uint(accY[i][2 * np + 1][e]), false, false);
        v[e] = __uint_as_float(sw2[0]);
        v[4 + e] = __uint_as_float(sw2[1]);
      }
      float4 o0, o1;
      o0.x = (v[0] + b0.x) + xr[i][0].x; o0.y = (v[1] + b0.y) + xr[i][0].y;
      o0.z = (v[2] + b0.z) + xr[i][0].z; o0.w = (v[3] + b0.w) + xr[i][0].w;
      o1.x = (v[4] + b1v.x) + xr[i][1].x; o1.y = (v[5] + b1v.y) + xr[i][1].y;
      o1.z = (v[6] + b1v.z) + xr[i][1].z; o1.w = (v[7] + b1v.w) + xr[i][1].w;
      if (img_ok && tk < S) {
        float* xp = x + (row0 + wm * S + tk) * EN_E + col;
        *reinterpret_cast<float4*>(xp) = o0;
        *reinterpret_cast<float4*>(xp + 4) = o1;
      }
    }
  }
}

__global__ __launch_bounds__(EN_THREADS) void enc_attn_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              const bf16* __restrict__ wqkv, const float* __restrict__ bqkv,
                                                              const bf16* __restrict__ wout, const float* __restrict__ bout,
                                                              int B, int S, float scale_log2e) {
  enc_attn_body(x, gamma, beta, eps, wqkv, bqkv, wout, bout, (int)blockIdx.x * 2, B, S, scale_log2e);
}

// ---------------------------------------------------------------------------------------------------------------
// The WHOLE encoder stack in one launch (round 3).  Both halves of a layer give a workgroup whole images (attention
// mixes tokens of one image only), and at 2 S <= 112 the mlp half takes the same 2 S consecutive token rows: the
// workgroup that owns images 2b, 2b + 1 needs nothing from any other workgroup from the first LayerNorm to the last
// residual add.  So the eight launches of a four-layer encoder become one: between the halves the fp32 residual rows
// go to memory and come straight back from the L2 of the CU's XCD (written microseconds earlier) instead of being
// drained to HBM behind a kernel boundary and fetched again, and the seven drain / launch gaps disappear.  The
// arithmetic is the two bodies above, unchanged: same bits as the two-kernel path and as the per-op path.
// Between the halves: the stores of the finishing half must be visible to every wave of the workgroup: release +
// barrier + acquire at WORKGROUP scope (the waves of a workgroup share their CU's vector L1, which is write-through, so
// this is a wait for the stores plus the barrier; agent-scope fences write back and invalidate the XCD's L2 at every
// boundary and made the launch 2.3x slower than the eight kernels it replaces).
constexpr int EN_MAX_LAYERS = 16;
struct EncLayerPtrs {
  const float *g1, *be1; const bf16* wqkv; const float* bqkv; const bf16* wout; const float* bout;
  const float *g2, *be2; const bf16* w1; const float* b1; const bf16* w2; const float* b2;
};
struct EncStackPtrs { EncLayerPtrs L[EN_MAX_LAYERS]; };

template <int MT>
__global__ __launch_bounds__(EN_THREADS) void enc_stack_kernel(float* __restrict__ x, const EncStackPtrs P, int n_layers,
                                                               float eps, int B, int S, int hidden, float scale_log2e) {
  const int img0 = (int)blockIdx.x * 2;
  const int n_img = B - img0 < 2 ? B - img0 : 2;
  const long row0 = (long)img0 * S;
  const int nrows = n_img * S;
#pragma unroll 1
  for (int l = 0; l < n_layers; ++l) {
    const EncLayerPtrs& q = P.L[l];
    enc_attn_body(x, q.g1, q.be1, eps, q.wqkv, q.bqkv, q.wout, q.bout, img0, B, S, scale_log2e);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    enc_mlp_body<MT>(x, q.g2, q.be2, eps, q.w1, q.b1, q.w2, q.b2, row0, nrows, hidden);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

template <int MT>
int launch_stack(float* x, const EncStackPtrs& P, int n_layers, float eps, int B, int S, int hidden, hipStream_t st) {
  constexpr int LDS = EN_ATTN_LDS > 2 * MT * 16 * 512 ? EN_ATTN_LDS : 2 * MT * 16 * 512;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)enc_stack_kernel<MT>}, LDS, "encoder_stack")) return rc;
  hipLaunchKernelGGL((enc_stack_kernel<MT>), dim3((unsigned)((B + 1) / 2)), dim3(EN_THREADS), LDS, st, x, P, n_layers, eps, B, S,
                     hidden, 0.125f * 1.44269504088896340736f);
  RVIT_CHECK_LAUNCH("encoder_stack");
  return RVIT_OK;
}

}  // namespace

// layer_ptrs: HOST array of 12 device pointers per layer, in the order of EncLayerPtrs (norm1 weight, bias, in_proj
// weight, bias, out_proj weight, bias, norm2 weight, bias, linear1 weight, bias, linear2 weight, bias)
int rvit_encoder_stack_impl(float* x, const void* const* layer_ptrs, int n_layers, float eps, int B, int S, int E, int H,
                            int hidden, hipStream_t st) {
  RVIT_CHECK_ARG(x && layer_ptrs, "encoder_stack: null operand");
  RVIT_CHECK_ARG(n_layers >= 0 && n_layers <= EN_MAX_LAYERS, "encoder_stack: %d layers (at most %d)", n_layers, EN_MAX_LAYERS);
  RVIT_CHECK_ARG(E == EN_E && H * 64 == E, "encoder_stack: built for embedding width %d with %d heads of 64, got E=%d H=%d",
                 EN_E, EN_E / 64, E, H);
  RVIT_CHECK_ARG(S >= 1 && 2 * S <= 112, "encoder_stack: sequence length %d outside 1..56 (two images per workgroup, at most 112 rows)", S);
  RVIT_CHECK_ARG(hidden > 0 && hidden % EN_HC == 0, "encoder_stack: hidden width %d must be a multiple of %d", hidden, EN_HC);
  RVIT_CHECK_ARG(B >= 0 && B < (1 << 30), "encoder_stack: bad batch %d", B);
  RVIT_CHECK_ARG(rvit_aligned16(x), "encoder_stack: x must be 16-byte aligned");
  EncStackPtrs P{};
  for (int l = 0; l < n_layers; ++l) {
    const void* const* q = layer_ptrs + 12 * l;
    for (int k = 0; k < 12; ++k) RVIT_CHECK_ARG(q[k] && rvit_aligned16(q[k]), "encoder_stack: layer %d operand %d is null or not 16-byte aligned", l, k);
    P.L[l] = EncLayerPtrs{(const float*)q[0], (const float*)q[1], (const bf16*)q[2], (const float*)q[3], (const bf16*)q[4],
                          (const float*)q[5], (const float*)q[6], (const float*)q[7], (const bf16*)q[8], (const float*)q[9],
                          (const bf16*)q[10], (const float*)q[11]};
  }
  if (B == 0 || n_layers == 0) return RVIT_OK;
  switch ((2 * S + 15) / 16) {
    case 1: return launch_stack<1>(x, P, n_layers, eps, B, S, hidden, st);
    case 2: return launch_stack<2>(x, P, n_layers, eps, B, S, hidden, st);
    case 3: return launch_stack<3>(x, P, n_layers, eps, B, S, hidden, st);
    case 4: return launch_stack<4>(x, P, n_layers, eps, B, S, hidden, st);
    case 5: return launch_stack<5>(x, P, n_layers, eps, B, S, hidden, st);
    case 6: return launch_stack<6>(x, P, n_layers, eps, B, S, hidden, st);
    default: return launch_stack<7>(x, P, n_layers, eps, B, S, hidden, st);
  }
}

int rvit_encoder_mlp_impl(float* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                          const void* w2, const float* b2, long T, int E, int hidden, int rows_per_wg, hipStream_t st) {
  RVIT_CHECK_ARG(x && gamma && beta && w1 && b1 && w2 && b2, "encoder_mlp: null operand");
  RVIT_CHECK_ARG(E == EN_E, "encoder_mlp: built for embedding width %d, got %d", EN_E, E);
  RVIT_CHECK_ARG(hidden > 0 && hidden % EN_HC == 0, "encoder_mlp: hidden width %d must be a multiple of %d", hidden, EN_HC);
  RVIT_CHECK_ARG(T >= 0 && T < (1L << 40), "encoder_mlp: bad token count %ld", T);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w1) && rvit_aligned16(w2) && rvit_aligned16(gamma) &&
                 rvit_aligned16(beta) && rvit_aligned16(b1) && rvit_aligned16(b2), "encoder_mlp: operands must be 16-byte aligned");
  if (T == 0) return RVIT_OK;
  // tokens per workgroup: one round over the 256 CUs when the batch allows it, at most 112 (7 MFMA row tiles)
  int rows = rows_per_wg;
  if (rows <= 0) {
    rows = (int)((T + 255) / 256);
    rows = rows < 16 ? 16 : rows > 112 ? 112 : rows;
  }
  RVIT_CHECK_ARG(rows >= 1 && rows <= 112, "encoder_mlp: rows_per_wg %d outside 1..112", rows);
  RVIT_CHECK_ARG((T + rows - 1) / rows < 0x7fffffffL, "encoder_mlp: too many workgroups");
  const bf16* W1 = (const bf16*)w1;
  const bf16* W2 = (const bf16*)w2;
  const int mt = (rows + 15) / 16;
  switch (mt) {
    case 1: return launch_mlp<1>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    case 2: return launch_mlp<2>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    case 3: return launch_mlp<3>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    case 4: return launch_mlp<4>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    case 5: return launch_mlp<5>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    case 6: return launch_mlp<6>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
    default: return launch_mlp<7>(x, gamma, beta, eps, W1, b1, W2, b2, T, rows, hidden, st);
  }
}

int rvit_encoder_attn_impl(float* x, const float* gamma, const float* beta, float eps, const void* wqkv,
                           const float* bqkv, const void* wout, const float* bout, int B, int S, int E, int H,
                           hipStream_t st) {
  RVIT_CHECK_ARG(x && gamma && beta && wqkv && bqkv && wout && bout, "encoder_attn: null operand");
  RVIT_CHECK_ARG(E == EN_E && H * 64 == E, "encoder_attn: built for embedding width %d with %d heads of 64, got E=%d H=%d",
                 EN_E, EN_E / 64, E, H);
  RVIT_CHECK_ARG(S >= 1 && S <= 64, "encoder_attn: sequence length %d outside 1..64", S);
  RVIT_CHECK_ARG(B >= 0 && B < (1 << 30), "encoder_attn: bad batch %d", B);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(wqkv) && rvit_aligned16(wout) && rvit_aligned16(gamma) &&
                 rvit_aligned16(beta) && rvit_aligned16(bqkv) && rvit_aligned16(bout), "encoder_attn: operands must be 16-byte aligned");
  if (B == 0) return RVIT_OK;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)enc_attn_kernel}, EN_ATTN_LDS, "encoder_attn")) return rc;
  hipLaunchKernelGGL(enc_attn_kernel, dim3((unsigned)((B + 1) / 2)), dim3(EN_THREADS), EN_ATTN_LDS, st, x, gamma, beta, eps,
                     (const bf16*)wqkv, bqkv, (const bf16*)wout, bout, B, S, 0.125f * 1.44269504088896340736f);
  RVIT_CHECK_LAUNCH("encoder_attn");
  return RVIT_OK;
}

int rvit_attn16_impl(const void* qkv, void* out, int B, int S, int H, long ldq, long ldo, float scale, hipStream_t st) {
  RVIT_CHECK_ARG(qkv && out, "attn_fwd_s64: null pointer");
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 64, "attn_fwd_s64: unsupported shape B=%d H=%d S=%d (S<=64)", B, H, S);
  RVIT_CHECK_ARG(ldq >= 3L * H * 64 && (ldq & 7) == 0 && ldo >= (long)H * 64 && (ldo & 3) == 0,
                 "attn_fwd_s64: bad pitches ldq=%ld ldo=%ld", ldq, ldo);
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && ((uintptr_t)out & 7) == 0 && (long)B * H < 0x7fffffffL, "attn_fwd_s64: alignment / size");
  hipLaunchKernelGGL(attn16_kernel, dim3((unsigned)(B * H)), dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, S, H, ldq, ldo,
                     scale * 1.44269504088896340736f);
  RVIT_CHECK_LAUNCH("attn_fwd_s64");
  return RVIT_OK;
}
