// Multi-head self-attention for short sequences (S <= 256, head_dim 64), forward.
// Replaces torch scaled_dot_product_attention as called from multi_head_attention_forward
// (torch functional.py:6569-6570, 6623-6631): softmax(q k^T / sqrt(Dh)) v, no mask, not causal.
//
// gfx950 structure: K and V of one (batch, head) are staged once into LDS (K rows XOR-swizzled
// for conflict-free ds_read_b128, V rows padded to 192 B for conflict-free ds_read_b64_tr_b16);
// each wave owns a 32-query block: scores are computed TRANSPOSED (S^T = K q^T, 32x32x16 MFMA) so
// that one lane owns one query column -> the softmax row reduction is a register reduction plus
// one cross-half shuffle; the exponentiated accumulator tile is fed straight back as the B
// operand of the PV product (O^T = V^T P^T) with V^T fragments produced by the hardware
// transposing LDS read.  The whole key range fits in registers (S <= 256), so no online rescale.
#include <stdlib.h>

#include "common.h"
#include "rvit_internal.h"
#include "mx_common.h"

namespace {

// -DRVIT_ATTN_STAMPS (diagnosis builds only, scripts/attn_stamps.py): s_memtime of wave 0 at the phase boundaries of
// the backward kernel, per workgroup
#ifdef RVIT_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[4096 * 8];
#define ATTN_PUT(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_attn_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef RVIT_ATTN_STAMPS_FWD   // the same array filled by the forward kernel instead (scripts/attn_fwd_stamps.py)
#define ATTN_FSTAMP(k) ATTN_PUT(k)
#define ATTN_STAMP(k) do {} while (0)
#else
#define ATTN_STAMP(k) ATTN_PUT(k)
#define ATTN_FSTAMP(k) do {} while (0)
#endif
#else
#define ATTN_STAMP(k) do {} while (0)
#define ATTN_FSTAMP(k) do {} while (0)
#endif

// 16-byte-per-lane LDS-DMA (global_load_lds_dwordx4): lane l of the wave writes lds_wave_base + 16 l
typedef __attribute__((address_space(3))) void attn_lds_void;
typedef __attribute__((address_space(1))) const void attn_gbl_cvoid;
__device__ __forceinline__ void attn_glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((attn_gbl_cvoid*)src, (attn_lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ uint4 g_attn_zero[1];   // source of the rows past the sequence end

// LDS image of a [rows][64] bf16 tensor that serves row reads (ds_read_b128) AND transposing reads (ds_read_b64_tr_b16)
// conflict-free: 128-byte rows, 16-byte chunk index ^= bitreverse3((row >> 1) & 7).  attn_stage_rows fills 8 rows per
// LDS-DMA instruction (lane-linear destination, so the swizzle goes on the source address); rows >= `rows_valid` read zeros.
__device__ __forceinline__ int du_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int du_off(int row, int ch) { return row * 128 + ((ch ^ du_swz(row)) << 4); }
__device__ __forceinline__ void attn_stage_rows(const bf16* src_row0, long ld, int rows_valid, char* image, int grp, int lane) {
  const int row = grp * 8 + (lane >> 3);
  const int c = (lane & 7) ^ du_swz(row);
  attn_glds16(row < rows_valid ? (const void*)(src_row0 + (long)row * ld + c * 8) : (const void*)g_attn_zero, image + grp * 1024);
}


__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, r);
}

// One 64-column row per lane pair (lane, lane + 32): each lane holds 4 consecutive columns (8 bytes) of eight column
// groups.  Stored as they are, that is sixteen 8-byte stores per lane and matrix, and the store tail of a workgroup is
// bound by store ISSUE, not bandwidth (MI355X_MICROARCH.md, "attention epilogue store tail").  v_permlane32_swap
// between the two lanes of a row pairs the groups up instead: lane h2 = 0 ends with columns 16j .. 16j + 7, lane
// h2 = 1 with 16j + 8 .. 16j + 15: eight 16-byte stores.
__device__ __forceinline__ void attn_store_row(bf16* row, const f32x16 (&acc)[2], int h2, float mul, bool ok) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int g0 = 2 * gp, g1 = 2 * gp + 1;
      const uint32_t x0 = pack_bf16x2(acc[db][4 * g0 + 0] * mul, acc[db][4 * g0 + 1] * mul);
      const uint32_t x1 = pack_bf16x2(acc[db][4 * g0 + 2] * mul, acc[db][4 * g0 + 3] * mul);
      const uint32_t y0 = pack_bf16x2(acc[db][4 * g1 + 0] * mul, acc[db][4 * g1 + 1] * mul);
      const uint32_t y1 = pack_bf16x2(acc[db][4 * g1 + 2] * mul, acc[db][4 * g1 + 3] * mul);
      // upper lanes of x <-> lower lanes of y
      const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
      uint4 w;
      w.x = s0[0]; w.y = s1[0]; w.z = s0[1]; w.w = s1[1];
      if (ok) *reinterpret_cast<uint4*>(row + db * 32 + 16 * gp + 8 * h2) = w;   // the swaps run on every lane
    }
}
// MXOUT (round 3; entry rvit_attn_fwd_mx, the MX-fp8 inference mode's default attention): the same kernel, but the
// output row is rounded to bf16 (the storage point of the per-op path) and written as the MXFP8 operand of the
// out_proj GEMM - e4m3 values in `out` (uint8 [T, ldo]) + one E8M0 scale per 32 columns in `out_s` (tiled, gemm_mx.hip)
// - exactly the bytes rvit_attn_fwd + rvit_quantize_mx produce, without the bf16 map and the quantisation pass.
template <int NKB, bool MXOUT = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                       float* __restrict__ lse, int total_pairs, int S, int H,
                                                       long ldq, long ldo, float scale_log2e, float scale,
                                                       DropKey dk, uint8_t* __restrict__ out_s = nullptr) {
  constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
  constexpr int SP = 32 * NKB;
  constexpr int PAIR_BYTES = SP * 256;   // K image, V image (du_off layout)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = H * 64;
  const int pair0 = blockIdx.x * PAIRS;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  // ---- stage K and V of every pair of this workgroup -------------------------------------------
#pragma unroll
  for (int pi = 0; pi < PAIRS; ++pi) {
    const int pair = pair0 + pi;
    const bool pv = pair < total_pairs;
    const int b = pv ? pair / H : 0, h = pv ? pair % H : 0;
    // K and V by LDS-DMA (no register round trip, all 14 requests of a wave in flight): -12 % kernel time at S = 197
    // against a loop of register loads + LDS stores
    const bf16* kbase = qkv + (long)b * S * ldq + E + h * 64;
    char* sK = smem + pi * PAIR_BYTES;
    constexpr int GROUPS = SP / 8;
    for (int gi = wave; gi < 2 * GROUPS; gi += 4) {   // wave-uniform
      const int tensor = gi / GROUPS, grp = gi - tensor * GROUPS;
      attn_stage_rows(kbase + tensor * E, ldq, pv ? S : 0, sK + tensor * SP * 128, grp, lane);
    }
  }
  ATTN_FSTAMP(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA has landed (explicit: not left to the fence of the barrier)
  __syncthreads();
  ATTN_FSTAMP(1);

  const int r32 = lane & 31, h2 = lane >> 5;
  for (int qi = wave; qi < PAIRS * NKB; qi += 4) {  // wave-uniform
    const int pi = qi / NKB, qb = qi - pi * NKB;
    const int pair = pair0 + pi;
    if (pair >= total_pairs) continue;
    if (qb * 32 >= S) continue;
    const int b = pair / H, h = pair - b * H;
    const char* sK = smem + pi * PAIR_BYTES;
    const char* sV = sK + SP * 128;
    const int query = qb * 32 + r32;
    const bool q_ok = query < S;
    const bf16* qrow = qkv + ((long)b * S + (q_ok ? query : 0)) * ldq + h * 64;

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 u = q_ok ? *reinterpret_cast<const uint4*>(qrow + ks * 16 + h2 * 8) : zero4;
      qf[ks] = __builtin_bit_cast(bf16x8, u);
    }

    if (qi == wave) ATTN_FSTAMP(2);
    // S^T[key][query] for all key blocks
    f32x16 sc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + du_off(kb * 32 + r32, ks * 2 + h2));
        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc[kb], 0, 0, 0);
      }
    }
    if (qi == wave) ATTN_FSTAMP(3);
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kb == NKB - 1) {  // only the last key block can hold padding (the launcher picks NKB = ceil(S / 32))
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (key >= S) sc[kb][r] = -INFINITY;
        }
        mx = fmaxf(mx, sc[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f((sc[kb][r] - mx) * scale_log2e);  // raw v_exp_f32: argument <= 0, flush-to-zero tail is fine
        sc[kb][r] = pe;
        sum += pe;
      }
    sum += __shfl_xor(sum, 32, 64);
    if (dk.thresh) {
      // dropout on the (normalised) probabilities.  Element index (pair * S + query) * S2 + key with the row stride
      // S2 = S rounded up to even, so that keys (2j, 2j + 1) of a query always share one hash (common.h): a lane
      // holds keys 4m .. 4m + 3 of its query, i.e. two whole hash pairs per group of four.
      const uint32_t base2 = (uint32_t)(pair * S + query) * (uint32_t)((S + 1) >> 1);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float m[4];
          drop_keep4(dk, base2 + kb * 16 + 4 * g + 2 * h2, dk.scale, m);
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[kb][4 * g + e] *= m[e];
        }
    }

    if (qi == wave) ATTN_FSTAMP(4);
    // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    const int tr_q = (lane >> 2) & 3;
    const int tr_ch = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);  // + 4*db
    const int tr_byte = (lane & 1) * 8;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (bf16)sc[kb][8 * s + j];
        const int k0 = kb * 32 + 16 * s + 4 * h2 + tr_q;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x4 lo = tr_read(sV + du_off(k0, 4 * db + tr_ch) + tr_byte);
          const bf16x4 hi = tr_read(sV + du_off(k0 + 8, 4 * db + tr_ch) + tr_byte);
          bf16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[db], 0, 0, 0);
        }
      }
    }
    if (qi == wave) ATTN_FSTAMP(5);
    const float inv = 1.0f / sum;
    if constexpr (MXOUT) {
      // a row's 64 columns sit in the lane pair (lane, lane + 32): per 32-column block db the lane holds columns
      // 8g + 4h2 + (0..3), g = 0..3 (as attn_fwd_fp8_kernel's epilogue)
      uint8_t* const out_q = reinterpret_cast<uint8_t*>(out);
      const long token = (long)b * S + query;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        float am = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[db][r] = (float)(bf16)(o[db][r] * inv);
          am = fmaxf(am, fabsf(o[db][r]));
        }
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const int sE = mx_scale_e8m0(am);
        const float qinv = mx_inv_scale(sE);
        if (q_ok) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint32_t*>(out_q + token * ldo + h * 64 + db * 32 + 8 * g + 4 * h2) =
                mx_pack4(o[db][4 * g + 0], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3], qinv);
          if (h2 == 0) out_s[mx_scale_offset(token, h * 64 + db * 32, E)] = (uint8_t)sE;
        }
      }
    } else {
      attn_store_row(out + ((long)b * S + (q_ok ? query : 0)) * ldo + h * 64, o, h2, inv, q_ok);
    }
    if (q_ok && lse && h2 == 0) lse[(long)pair * S + query] = mx * scale + logf(sum);
    if (qi == wave) ATTN_FSTAMP(6);
  }
  ATTN_FSTAMP(7);
}

template <int NKB>
int launch_attn_fwd(const bf16* qkv, bf16* out, float* lse, int B, int S, int H, long ldq, long ldo, float scale,
                    DropKey dk, hipStream_t st) {
  constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
  constexpr int LDS = PAIRS * 32 * NKB * 256;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)attn_fwd_kernel<NKB>}, LDS, "attn_fwd")) return rc;
  const int total = B * H;
  const unsigned grid = (unsigned)((total + PAIRS - 1) / PAIRS);
  hipLaunchKernelGGL((attn_fwd_kernel<NKB>), dim3(grid), dim3(256), LDS, st, qkv, out, lse, total, S, H, ldq, ldo,
                     scale * 1.44269504088896340736f, scale, dk);
  RVIT_CHECK_LAUNCH("attn_fwd");
  return RVIT_OK;
}

template <int NKB>
int launch_attn_fwd_mx(const bf16* qkv, uint8_t* out_q, uint8_t* out_s, int B, int S, int H, long ldq, long ldo,
                       float scale, hipStream_t st) {
  constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
  constexpr int LDS = PAIRS * 32 * NKB * 256;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)attn_fwd_kernel<NKB, true>}, LDS, "attn_fwd_mx")) return rc;
  const int total = B * H;
  const unsigned grid = (unsigned)((total + PAIRS - 1) / PAIRS);
  hipLaunchKernelGGL((attn_fwd_kernel<NKB, true>), dim3(grid), dim3(256), LDS, st, qkv, reinterpret_cast<bf16*>(out_q),
                     (float*)nullptr, total, S, H, ldq, ldo, scale * 1.44269504088896340736f, scale, make_drop_key(0.f, 0),
                     out_s);
  RVIT_CHECK_LAUNCH("attn_fwd_mx");
  return RVIT_OK;
}


// ---------------------------------------------------------------------------------------------
// fp8 attention for the MX-fp8 inference mode (BASELINE.json configs[4]: "fp8 MFMA attention + MLP"): the same
// dataflow as attn_fwd_kernel with both products on v_mfma_f32_32x32x16_fp8_fp8 and the output emitted directly as
// MXFP8 (the operand of the out_proj's rvit_gemm_mxfp8), so the bf16 attention map and its quantisation pass
// disappear.  q, k, v arrive as the bf16 in_proj output and are quantised to e4m3 while they are staged:
//   k: one power-of-two scale per (key, head) row of 64 (E8M0 rule of mx_common.h: 2^(floor(log2 amax) - 8));
//   q: the same per (query, head) row, in the wave that owns the query;
//   v: one scale per (image, head) (its contraction axis is the key axis, so a per-row scale could not be factored
//      out of the P V sum), stored TRANSPOSED in LDS as bytes, V^T[d][key];
//   p: exp2(..) in (0, 1] times 256, rounded to e4m3 (the row sum uses the unrounded values).
// scores = (q8 . k8) * sq * sk in fp32; O = (P8 . V8) * sv / (256 * sum), rounded to bf16 (the per-op path's storage
// point), then MX-quantised per 32 columns.  oracle/ref_mxfp8.py (fp8_attention=True) restates exactly this.
template <int NKB>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(const bf16* __restrict__ qkv, uint8_t* __restrict__ out_q,
                                                           uint8_t* __restrict__ out_s, int S, int H, long ldq,
                                                           long ldo, float scale_log2e) {
  constexpr int SP = 32 * NKB;
  constexpr int VP = SP + 4;  // bytes per V^T row: (SP + 4) / 4 odd -> the 32 rows of a b32 fragment read hit 32 banks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;                                   // [SP][64] e4m3, 8-byte chunk c of row r at c ^ ((r >> 2) & 7)
  float* const sKs = reinterpret_cast<float*>(smem + SP * 64);   // [SP] scale of each key row
  char* const sVt = smem + SP * 64 + SP * 4;               // [64][VP] e4m3
  float* const red = reinterpret_cast<float*>(sVt + 64 * VP);    // [4]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = H * 64;
  const int pair = blockIdx.x;
  const int b = pair / H, h = pair - b * H;
  const bf16* kbase = qkv + (long)b * S * ldq + E + h * 64;
  const bf16* vbase = kbase + E;

  // ---- stage K (quantised per row) and find V's absolute maximum ------------------------------------------------
  uint4 vreg[NKB];
  float vmax = 0.f;
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int id = t + it * 256, row = id >> 3, c = id & 7;
    const bool ok = row < S;
    uint4 kv = make_uint4(0, 0, 0, 0);
    vreg[it] = make_uint4(0, 0, 0, 0);
    if (ok) {
      kv = *reinterpret_cast<const uint4*>(kbase + (long)row * ldq + c * 8);
      vreg[it] = *reinterpret_cast<const uint4*>(vbase + (long)row * ldq + c * 8);
    }
    const float kf[8] = {bf16_lo(kv.x), bf16_hi(kv.x), bf16_lo(kv.y), bf16_hi(kv.y),
                         bf16_lo(kv.z), bf16_hi(kv.z), bf16_lo(kv.w), bf16_hi(kv.w)};
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(kf[e]));
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    am = fmaxf(am, __shfl_xor(am, 4, 64));
    const int sE = mx_scale_e8m0(am);
    const float inv = mx_inv_scale(sE);
    uint2 w;
    w.x = mx_pack4(kf[0], kf[1], kf[2], kf[3], inv);
    w.y = mx_pack4(kf[4], kf[5], kf[6], kf[7], inv);
    *reinterpret_cast<uint2*>(sK + row * 64 + ((c ^ ((row >> 2) & 7)) << 3)) = w;
    if (c == 0) sKs[row] = __uint_as_float((uint32_t)sE << 23);
    const uint4 vv = vreg[it];
    const float vf[8] = {bf16_lo(vv.x), bf16_hi(vv.x), bf16_lo(vv.y), bf16_hi(vv.y),
                         bf16_lo(vv.z), bf16_hi(vv.z), bf16_lo(vv.w), bf16_hi(vv.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) vmax = fmaxf(vmax, fabsf(vf[e]));
  }
  vmax = wave_max(vmax);
  if (lane == 0) red[wave] = vmax;
  __syncthreads();
  vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int sEv = mx_scale_e8m0(vmax);
  const float inv_v = mx_inv_scale(sEv);
  const float sv = __uint_as_float((uint32_t)sEv << 23);
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int id = t + it * 256, row = id >> 3, c = id & 7;
    const uint4 vv = vreg[it];
    const uint32_t w0 = mx_pack4(bf16_lo(vv.x), bf16_hi(vv.x), bf16_lo(vv.y), bf16_hi(vv.y), inv_v);
    const uint32_t w1 = mx_pack4(bf16_lo(vv.z), bf16_hi(vv.z), bf16_lo(vv.w), bf16_hi(vv.w), inv_v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sVt[(c * 8 + e) * VP + row] = (char)((w0 >> (8 * e)) & 0xff);
      sVt[(c * 8 + 4 + e) * VP + row] = (char)((w1 >> (8 * e)) & 0xff);
    }
  }
  __syncthreads();

  const int r32 = lane & 31, h2 = lane >> 5;
  for (int qb = wave; qb < NKB; qb += 4) {  // wave-uniform
    if (qb * 32 >= S) continue;
    const int query = qb * 32 + r32;
    const bool q_ok = query < S;
    const bf16* qrow = qkv + ((long)b * S + (q_ok ? query : 0)) * ldq + h * 64;
    float qv[4][8];
    float qam = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 u = make_uint4(0, 0, 0, 0);
      if (q_ok) u = *reinterpret_cast<const uint4*>(qrow + ks * 16 + h2 * 8);
      qv[ks][0] = bf16_lo(u.x); qv[ks][1] = bf16_hi(u.x); qv[ks][2] = bf16_lo(u.y); qv[ks][3] = bf16_hi(u.y);
      qv[ks][4] = bf16_lo(u.z); qv[ks][5] = bf16_hi(u.z); qv[ks][6] = bf16_lo(u.w); qv[ks][7] = bf16_hi(u.w);
#pragma unroll
      for (int e = 0; e < 8; ++e) qam = fmaxf(qam, fabsf(qv[ks][e]));
    }
    qam = fmaxf(qam, __shfl_xor(qam, 32, 64));
    const int sEq = mx_scale_e8m0(qam);
    const float inv_q = mx_inv_scale(sEq);
    const float sq = __uint_as_float((uint32_t)sEq << 23);
    long qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint32_t lo = mx_pack4(qv[ks][0], qv[ks][1], qv[ks][2], qv[ks][3], inv_q);
      const uint32_t hi = mx_pack4(qv[ks][4], qv[ks][5], qv[ks][6], qv[ks][7], inv_q);
      qf[ks] = (long)(((unsigned long)hi << 32) | lo);
    }

    f32x16 sc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
      const int krow = kb * 32 + r32;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const long kf = *reinterpret_cast<const long*>(sK + krow * 64 + (((ks * 2 + h2) ^ ((krow >> 2) & 7)) << 3));
        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], sc[kb], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 sk4 = *reinterpret_cast<const float4*>(sKs + kb * 32 + 8 * g + 4 * h2);
        const float skv[4] = {sk4.x, sk4.y, sk4.z, sk4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float v = sc[kb][r] * (sq * skv[e]);
          if (kb == NKB - 1 && kb * 32 + 8 * g + 4 * h2 + e >= S) v = -INFINITY;
          sc[kb][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f((sc[kb][r] - mx) * scale_log2e);
        sc[kb][r] = pe;
        sum += pe;
      }
    sum += __shfl_xor(sum, 32, 64);

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t plo = mx_pack4(sc[kb][8 * s + 0], sc[kb][8 * s + 1], sc[kb][8 * s + 2], sc[kb][8 * s + 3], 256.f);
        const uint32_t phi = mx_pack4(sc[kb][8 * s + 4], sc[kb][8 * s + 5], sc[kb][8 * s + 6], sc[kb][8 * s + 7], 256.f);
        const long pb = (long)(((unsigned long)phi << 32) | plo);
        const int key0 = kb * 32 + 16 * s + 4 * h2;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vrow = sVt + (db * 32 + r32) * VP + key0;
          const uint32_t lo = *reinterpret_cast<const uint32_t*>(vrow);
          const uint32_t hi = *reinterpret_cast<const uint32_t*>(vrow + 8);
          const long vf = (long)(((unsigned long)hi << 32) | lo);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pb, o[db], 0, 0, 0);
        }
      }
    }
    const float fin = sv * (1.0f / 256.0f) / sum;
    const long token = (long)b * S + query;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float am = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[db][r] = (float)(bf16)(o[db][r] * fin);   // the bf16 storage point of the per-op path
        am = fmaxf(am, fabsf(o[db][r]));
      }
      am = fmaxf(am, __shfl_xor(am, 32, 64));
      const int sE = mx_scale_e8m0(am);
      const float inv = mx_inv_scale(sE);
      if (q_ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint32_t*>(out_q + token * ldo + h * 64 + db * 32 + 8 * g + 4 * h2) =
              mx_pack4(o[db][4 * g + 0], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3], inv);
        if (h2 == 0) out_s[mx_scale_offset(token, h * 64 + db * 32, E)] = (uint8_t)sE;
      }
    }
  }
}

template <int NKB>
int launch_attn_fwd_fp8(const bf16* qkv, uint8_t* out_q, uint8_t* out_s, int B, int S, int H, long ldq, long ldo,
                        float scale, hipStream_t st) {
  constexpr int SP = 32 * NKB;
  constexpr int LDS = SP * 64 + SP * 4 + 64 * (SP + 4) + 16;
  hipLaunchKernelGGL((attn_fwd_fp8_kernel<NKB>), dim3((unsigned)(B * H)), dim3(256), LDS, st, qkv, out_q, out_s, S, H, ldq,
                     ldo, scale * 1.44269504088896340736f);
  RVIT_CHECK_LAUNCH("attn_fwd_fp8");
  return RVIT_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward (autograd of scaled_dot_product_attention, run by train_vit.py:939).  P is recomputed
// from q, k and the forward's log-sum-exp.  Everything of one (batch, head) — Q, K, V, dO — sits
// in LDS in ONE image per tensor that serves both row reads (ds_read_b128) and transposing reads
// (ds_read_b64_tr_b16) conflict-free: 128-byte rows, chunk ^= bitreverse3((row>>1)&7).
//   pass A: a wave owns 32 keys (key on the MFMA lane), sweeps the query blocks:
//           S = Q K^T, dP = dO V^T -> P, dS in registers -> dV^T += dO^T P, dK^T += Q^T dS
//           (the accumulator tiles are used directly as B operands, no LDS round trip)
//   pass B: a wave owns 32 queries (query on the lane), sweeps the key blocks:
//           S^T, dP^T recomputed -> dS^T -> dQ^T += K^T dS^T
// No cross-wave reduction and no atomics: dq, dk, dv are bitwise reproducible.

// Operands that both sweeps of the backward share
struct BwdCtx {
  int pair, S;
  uint32_t half_stride;   // dropout element index: (pair * S + query) * S2 + key, S2 = S rounded up to even
  float c_l2, scale, keep_ds;
  DropKey dk;
};

// pass A for the 32 keys a wave owns (K, V rows in registers): sweep the query blocks of the Q / dO images
template <int NKB, bool DROP>
__device__ __forceinline__ void bwd_sweep_queries(const char* sQ, const char* sD, const float* sLse, const float* sDel,
                                                  const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int kb, const BwdCtx& cx,
                                                  f32x16 (&dVt)[2], f32x16 (&dKt)[2]) {
  const int lane = threadIdx.x & 63, r32 = lane & 31, h2 = lane >> 5;
  const int tr_q = (lane >> 2) & 3;
  const int tr_ch = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);  // + 4*db
  const int tr_byte = (lane & 1) * 8;
  const int pair = cx.pair, S = cx.S;
  const uint32_t half_stride = cx.half_stride;
  const float c_l2 = cx.c_l2, scale = cx.scale, keep_ds = cx.keep_ds;
  const DropKey& dk = cx.dk;
  (void)scale; (void)keep_ds; (void)half_stride; (void)pair;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dVt[db][r] = 0.f; dKt[db][r] = 0.f; }
#pragma unroll 1
  for (int qb = 0; qb < NKB; ++qb) {
    if (qb * 32 >= S) break;
    f32x16 sa, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQ + du_off(qb * 32 + r32, ks * 2 + h2));
      const bf16x8 df = *reinterpret_cast<const bf16x8*>(sD + du_off(qb * 32 + r32, ks * 2 + h2));
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sa, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], dp, 0, 0, 0);
    }
    // rows of the accumulators are queries qb*32 + 8g + 4h2 + (0..3)
    uint32_t hq[DROP ? 16 : 1];
    if constexpr (DROP) {
      // this lane hashes 8 of the block's 16 accumulator rows (even key: rows 0-7, odd key: rows 8-15) for the key
      // pair it belongs to; quad_perm [0,0,2,2] / [1,1,3,3] hands every lane the even / the odd lane's result
      uint32_t mine[8];
      const int odd = lane & 1;
      const uint32_t kp = (uint32_t)(kb * 16 + (r32 >> 1));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int query = qb * 32 + (i & 3) + 8 * (i >> 2) + 16 * odd + 4 * h2;
        mine[i] = rvit_drop_hash(dk, (uint32_t)(pair * S + query) * half_stride + kp);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        hq[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[i], 0xA0 /* quad_perm [0,0,2,2] */, 0xf, 0xf, false);
        hq[i + 8] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[i], 0xF5 /* quad_perm [1,1,3,3] */, 0xf, 0xf, false);
      }
    }
    const int half_shift = 16 * (lane & 1);   // even key: low half of the pair's hash, odd key: high half
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(sLse + qb * 32 + 8 * g + 4 * h2);
      const float4 d4 = *reinterpret_cast<const float4*>(sDel + qb * 32 + 8 * g + 4 * h2);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
      const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const float pe = __builtin_amdgcn_exp2f(fmaf(sa[r], c_l2, -lv[e]));
        if constexpr (DROP) {
          const bool keep = ((hq[r] >> half_shift) & 0xffffu) >= dk.thresh;
          sa[r] = pe * (keep ? dk.scale : 0.f);                         // dropped probabilities feed dV
          dp[r] = pe * fmaf(dp[r], keep ? keep_ds : 0.f, -dl[e]);       // dS = P * (dP_drop * mask/(1-p) - delta) * scale
        } else {
          sa[r] = pe;
          dp[r] = pe * fmaf(dp[r], scale, -dl[e]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 pb, dsb;
#pragma unroll
      for (int j = 0; j < 8; ++j) { pb[j] = (bf16)sa[8 * s + j]; dsb[j] = (bf16)dp[8 * s + j]; }
      const int q0 = qb * 32 + 16 * s + 4 * h2 + tr_q;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int o_lo = du_off(q0, 4 * db + tr_ch) + tr_byte;
        const int o_hi = du_off(q0 + 8, 4 * db + tr_ch) + tr_byte;
        const bf16x4 dlo = tr_read(sD + o_lo), dhi = tr_read(sD + o_hi);
        const bf16x4 qlo = tr_read(sQ + o_lo), qhi = tr_read(sQ + o_hi);
        const bf16x8 dOt = bf16x8{dlo[0], dlo[1], dlo[2], dlo[3], dhi[0], dhi[1], dhi[2], dhi[3]};
        const bf16x8 Qt = bf16x8{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
        dVt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dOt, pb, dVt[db], 0, 0, 0);
        dKt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Qt, dsb, dKt[db], 0, 0, 0);
      }
    }
  }
}

// pass B for the 32 queries a wave owns (Q, dO rows in registers): sweep the key blocks of the K / V images
template <int NKB, bool DROP>
__device__ __forceinline__ void bwd_sweep_keys(const char* sK, const char* sV, const bf16x8 (&qf)[4], const bf16x8 (&df)[4],
                                               float lse_q, float del_q, int qb, const BwdCtx& cx, f32x16 (&dQt)[2]) {
  const int lane = threadIdx.x & 63, r32 = lane & 31, h2 = lane >> 5;
  const int tr_q = (lane >> 2) & 3;
  const int tr_ch = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);  // + 4*db
  const int tr_byte = (lane & 1) * 8;
  const int S = cx.S;
  const float c_l2 = cx.c_l2, scale = cx.scale, keep_ds = cx.keep_ds;
  const DropKey& dk = cx.dk;
  const uint32_t base2 = (uint32_t)(cx.pair * S + qb * 32 + r32) * cx.half_stride;
  (void)scale; (void)keep_ds; (void)base2; (void)dk;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQt[db][r] = 0.f;
#pragma unroll 1
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb * 32 >= S) break;
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(sK + du_off(kb * 32 + r32, ks * 2 + h2));
      const bf16x8 vfr = *reinterpret_cast<const bf16x8*>(sV + du_off(kb * 32 + r32, ks * 2 + h2));
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], st, 0, 0, 0);
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, df[ks], dpt, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // keys kb*32 + 8g + 4h2 + (0..3): two whole hash pairs
      float m[4];
      if constexpr (DROP) drop_keep4(dk, base2 + kb * 16 + 4 * g + 2 * h2, keep_ds, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        // padded keys (>= S) need no mask here: their K rows are zero in LDS, so whatever finite dS^T they get
        // contributes K^T dS^T = 0 to dQ
        const float pe = __builtin_amdgcn_exp2f(fmaf(st[r], c_l2, -lse_q));
        dpt[r] = pe * fmaf(dpt[r], DROP ? m[e] : scale, -del_q);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 dsb;
#pragma unroll
      for (int j = 0; j < 8; ++j) dsb[j] = (bf16)dpt[8 * s + j];
      const int k0 = kb * 32 + 16 * s + 4 * h2 + tr_q;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x4 lo = tr_read(sK + du_off(k0, 4 * db + tr_ch) + tr_byte);
        const bf16x4 hi = tr_read(sK + du_off(k0 + 8, 4 * db + tr_ch) + tr_byte);
        const bf16x8 Kt = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dQt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kt, dsb, dQt[db], 0, 0, 0);
      }
    }
  }
}

// ---- pass A with the operands still arriving (round 3, NKB >= 5: S > 128) -------------------------------------------
// The kernel below staged all four images (126 KB at S = 197) behind ONE barrier: 19k of the 42k cycles a workgroup
// lives (stamps, profiles/r02_gemm8p_notes.txt section 8), with nothing to hide them behind - the workgroup owns its CU.
// attn_bwd_pipe_kernel requests everything by LDS-DMA up front, in the order it is needed - K | V, then Q | dO block
// by block - one 1 KB piece per wave per block (waves 0-3 the first tensor of the pair, waves 4-7 the second), and pass
// A starts on query block 0 as soon as K, V and that block have landed: before every query block one counted
// s_waitcnt vmcnt (this wave's piece of the block) and one workgroup barrier (everybody's), the later blocks stay in
// flight behind the sweep.  delta / lse of all rows are computed first from the global dO and O rows (their loads are
// the oldest, so consuming them leaves every DMA in flight; dO is read twice, the second time from L2).
// The transposing reads of the sweep are inline assembly: a ds_read_tr the compiler can see makes its wait-count pass
// wait for EVERY LDS-DMA in flight (wgrad.hip tr_read_asm) - here that would be the blocks still arriving.
__device__ __forceinline__ void attn_tr_read_asm(s16x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
__device__ __forceinline__ void attn_tr_wait8(s16x4 (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
               :
               : "memory");
}
template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void attn_wait_vmcnt_rt(int n) {   // n in [0, 8], wave-uniform
  switch (n) {
    case 0: attn_wait_vmcnt<0>(); break;
    case 1: attn_wait_vmcnt<1>(); break;
    case 2: attn_wait_vmcnt<2>(); break;
    case 3: attn_wait_vmcnt<3>(); break;
    case 4: attn_wait_vmcnt<4>(); break;
    case 5: attn_wait_vmcnt<5>(); break;
    case 6: attn_wait_vmcnt<6>(); break;
    case 7: attn_wait_vmcnt<7>(); break;
    default: attn_wait_vmcnt<8>(); break;
  }
}

// one query block of pass A (the body of bwd_sweep_queries' loop; same arithmetic in the same order: same bits)
template <bool DROP>
__device__ __forceinline__ void bwd_query_block(const char* sQ, const char* sD, uint32_t aQ, uint32_t aD, uint32_t aLse,
                                                uint32_t del_off, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], int kb,
                                                int qb, const BwdCtx& cx, f32x16 (&dVt)[2], f32x16 (&dKt)[2]) {
  const int lane = threadIdx.x & 63, r32 = lane & 31, h2 = lane >> 5;
  const int tr_q = (lane >> 2) & 3;
  const int tr_ch = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);  // + 4*db
  const int tr_byte = (lane & 1) * 8;
  const int pair = cx.pair, S = cx.S;
  const uint32_t half_stride = cx.half_stride;
  const float c_l2 = cx.c_l2, scale = cx.scale, keep_ds = cx.keep_ds;
  const DropKey& dk = cx.dk;
  (void)scale; (void)keep_ds; (void)half_stride; (void)pair; (void)S; (void)kb;
  // every LDS read of the block goes out first (inline assembly: see above): the lse / delta quads and the sixteen
  // transposing reads of the two output MFMA groups land behind the score MFMAs and the vector work
  f32x4 l4[4], d4[4];
  s16x4 rd[2][8];   // [s][4 db + {0: dO lo, 1: dO hi, 2: Q lo, 3: Q hi}]
  {
    const uint32_t a0 = aLse + (uint32_t)((qb * 32 + 4 * h2) * 4);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(l4[g]) : "v"(a0 + g * 32) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(d4[g]) : "v"(a0 + g * 32 + del_off) : "memory");
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q0 = qb * 32 + 16 * s + 4 * h2 + tr_q;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const uint32_t o_lo = (uint32_t)(du_off(q0, 4 * db + tr_ch) + tr_byte);
        const uint32_t o_hi = (uint32_t)(du_off(q0 + 8, 4 * db + tr_ch) + tr_byte);
        attn_tr_read_asm(rd[s][4 * db + 0], aD + o_lo);
        attn_tr_read_asm(rd[s][4 * db + 1], aD + o_hi);
        attn_tr_read_asm(rd[s][4 * db + 2], aQ + o_lo);
        attn_tr_read_asm(rd[s][4 * db + 3], aQ + o_hi);
      }
    }
  }
  f32x16 sa, dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sQ + du_off(qb * 32 + r32, ks * 2 + h2));
    const bf16x8 df = *reinterpret_cast<const bf16x8*>(sD + du_off(qb * 32 + r32, ks * 2 + h2));
    sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sa, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], dp, 0, 0, 0);
  }
  uint32_t hq[DROP ? 16 : 1];
  if constexpr (DROP) {
    uint32_t mine[8];
    const int odd = lane & 1;
    const uint32_t kp = (uint32_t)(kb * 16 + (r32 >> 1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int query = qb * 32 + (i & 3) + 8 * (i >> 2) + 16 * odd + 4 * h2;
      mine[i] = rvit_drop_hash(dk, (uint32_t)(pair * S + query) * half_stride + kp);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      hq[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[i], 0xA0 /* quad_perm [0,0,2,2] */, 0xf, 0xf, false);
      hq[i + 8] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[i], 0xF5 /* quad_perm [1,1,3,3] */, 0xf, 0xf, false);
    }
  }
  const int half_shift = 16 * (lane & 1);
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(l4[0]), "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]), "+v"(d4[3])
               :
               : "memory");
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      const float pe = __builtin_amdgcn_exp2f(fmaf(sa[r], c_l2, -l4[g][e]));
      if constexpr (DROP) {
        const bool keep = ((hq[r] >> half_shift) & 0xffffu) >= dk.thresh;
        sa[r] = pe * (keep ? dk.scale : 0.f);
        dp[r] = pe * fmaf(dp[r], keep ? keep_ds : 0.f, -d4[g][e]);
      } else {
        sa[r] = pe;
        dp[r] = pe * fmaf(dp[r], scale, -d4[g][e]);
      }
    }
  }
  attn_tr_wait8(rd[0]);
  attn_tr_wait8(rd[1]);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 pb, dsb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pb[j] = (bf16)sa[8 * s + j]; dsb[j] = (bf16)dp[8 * s + j]; }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const bf16x4 dlo = __builtin_bit_cast(bf16x4, rd[s][4 * db + 0]), dhi = __builtin_bit_cast(bf16x4, rd[s][4 * db + 1]);
      const bf16x4 qlo = __builtin_bit_cast(bf16x4, rd[s][4 * db + 2]), qhi = __builtin_bit_cast(bf16x4, rd[s][4 * db + 3]);
      const bf16x8 dOt = bf16x8{dlo[0], dlo[1], dlo[2], dlo[3], dhi[0], dhi[1], dhi[2], dhi[3]};
      const bf16x8 Qt = bf16x8{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
      dVt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dOt, pb, dVt[db], 0, 0, 0);
      dKt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Qt, dsb, dKt[db], 0, 0, 0);
    }
  }
}

template <int NKB, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_pipe_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                               const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                               bf16* __restrict__ dqkv, int total_pairs, int S, int H,
                                                               long ldq, long ldo, long lddo, long lddq, float scale,
                                                               DropKey dk) {
  static_assert(NKB >= 5 && NKB <= 8, "one (image, head) pair per workgroup, eight waves");
  constexpr int SP = 32 * NKB;
  constexpr int TILE = SP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQ = smem;
  char* const sK = sQ + TILE;
  char* const sV = sK + TILE;
  char* const sD = sV + TILE;
  float* const sLse = reinterpret_cast<float*>(sD + TILE);
  float* const sDel = sLse + SP;
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_cchar*)smem;
  const uint32_t aQ = lds0, aD = lds0 + 3 * TILE;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = H * 64;
  const int pair = blockIdx.x;   // (an XCD-aware pair order, heads of an image on one XCD, measured no different)
  const int b = pair / H, h = pair - b * H;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const float c_l2 = scale * 1.44269504088896340736f;
  const uint32_t half_stride = (uint32_t)((S + 1) >> 1);
  const float keep_ds = dk.scale * scale;
  ATTN_STAMP(0);

  // ---- the global rows for delta / lse (oldest requests), then every LDS-DMA piece in the order of use ----------------
  constexpr int DIT = (SP * 8 + 511) / 512;   // (row, chunk) items per thread
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 dreg[DIT], oreg[DIT];
  float lreg[DIT];
  const bf16* const dbase = dO + (long)b * S * lddo + h * 64;
  const bf16* const obase = O + (long)b * S * ldo + h * 64;
  // As inline assembly: loads the compiler can see are waited for with s_waitcnt vmcnt(0) at their first use once an
  // LDS-DMA has been issued behind them (its wait-count pass gives up counting), which would drain every piece below
  // before pass A starts.  Rows >= S read row 0 (a valid address) and are zeroed / set to +inf afterwards.
#pragma unroll
  for (int it = 0; it < DIT; ++it) {
    const int id = t + it * 512, row = id >> 3, c = id & 7;
    const int rsafe = row < S ? row : 0;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dreg[it]) : "v"(dbase + (long)rsafe * lddo + c * 8) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(oreg[it]) : "v"(obase + (long)rsafe * ldo + c * 8) : "memory");
    asm volatile("global_load_dword %0, %1, off" : "=v"(lreg[it]) : "v"(lse + (long)pair * S + rsafe) : "memory");
  }
  const bf16* const qbase = qkv + (long)b * S * ldq + h * 64;
  {
    const bool second = wave >= 4;   // waves 0-3: K / Q, waves 4-7: V / dO
    const int sub = wave & 3;
    const bf16* kv_src = qbase + (second ? 2 * E : E);
    char* kv_img = second ? sV : sK;
#pragma unroll
    for (int j = 0; j < NKB; ++j) attn_stage_rows(kv_src, ldq, S, kv_img, j * 4 + sub, lane);
    const bf16* qd_src = second ? dbase : qbase;
    const long qd_ld = second ? lddo : ldq;
    char* qd_img = second ? sD : sQ;
#pragma unroll
    for (int j = 0; j < NKB; ++j) attn_stage_rows(qd_src, qd_ld, S, qd_img, j * 4 + sub, lane);
  }
  // the dO / O / lse rows are back; all 2 NKB pieces still in flight
#pragma unroll
  for (int it = 0; it < DIT; ++it)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(dreg[it]), "+v"(oreg[it]), "+v"(lreg[it]) : "n"(2 * NKB) : "memory");
#pragma unroll
  for (int it = 0; it < DIT; ++it) {
    const int id = t + it * 512, row = id >> 3, c = id & 7;
    const bool rok = row < S;
    const uint4 dv = rok ? make_uint4(dreg[it][0], dreg[it][1], dreg[it][2], dreg[it][3]) : zero4;
    const uint4 ov = make_uint4(oreg[it][0], oreg[it][1], oreg[it][2], oreg[it][3]);
    float d = bf16_lo(dv.x) * bf16_lo(ov.x) + bf16_hi(dv.x) * bf16_hi(ov.x) +
              bf16_lo(dv.y) * bf16_lo(ov.y) + bf16_hi(dv.y) * bf16_hi(ov.y) +
              bf16_lo(dv.z) * bf16_lo(ov.z) + bf16_hi(dv.z) * bf16_hi(ov.z) +
              bf16_lo(dv.w) * bf16_lo(ov.w) + bf16_hi(dv.w) * bf16_hi(ov.w);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (c == 0 && row < SP) {
      sDel[row] = d * scale;
      sLse[row] = row < S ? lreg[it] * 1.44269504088896340736f : INFINITY;   // rows >= S: +inf -> P = 0
    }
  }
  // this wave's K / V pieces have landed, its sDel / sLse stores are done; then everybody's (a raw barrier:
  // __syncthreads() would also wait for every piece still in flight)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NKB) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ATTN_STAMP(1);
  ATTN_STAMP(2);

  const int r32 = lane & 31, h2 = lane >> 5;
  const BwdCtx cx{pair, S, half_stride, c_l2, scale, keep_ds, dk};
  // ================================ pass A: dK, dV ================================================
  {
    const int kb = wave;
    const bool active = kb < NKB;
    bf16x8 kf[4], vf[4];
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(sK + du_off((active ? kb : 0) * 32 + r32, ks * 2 + h2));
      vf[ks] = *reinterpret_cast<const bf16x8*>(sV + du_off((active ? kb : 0) * 32 + r32, ks * 2 + h2));
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dVt[db][r] = 0.f; dKt[db][r] = 0.f; }
#pragma unroll 1
    for (int qb = 0; qb < NKB; ++qb) {
      attn_wait_vmcnt_rt(NKB - 1 - qb);   // this wave's piece of query block qb has landed
      __builtin_amdgcn_s_barrier();       // everybody's
      asm volatile("" ::: "memory");
      if (active) bwd_query_block<DROP>(sQ, sD, aQ, aD, lds0 + 4 * TILE, (uint32_t)(SP * 4), kf, vf, kb, qb, cx, dVt, dKt);
    }
    ATTN_STAMP(3);
    if (active) {
      const int key = kb * 32 + r32;
      const bool k_ok = key < S;
      bf16* krow = dqkv + ((long)b * S + (k_ok ? key : 0)) * lddq + E + h * 64;
      attn_store_row(krow, dKt, h2, 1.0f, k_ok);
      attn_store_row(krow + E, dVt, h2, 1.0f, k_ok);
    }
  }
  ATTN_STAMP(4);
  // ================================ pass B: dQ (everything is in LDS) ==============================
  if (wave < NKB) {
    const int qb = wave;
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + du_off(qb * 32 + r32, ks * 2 + h2));
      df[ks] = *reinterpret_cast<const bf16x8*>(sD + du_off(qb * 32 + r32, ks * 2 + h2));
    }
    const float lse_q = sLse[qb * 32 + r32];
    const float del_q = sDel[qb * 32 + r32];
    f32x16 dQt[2];
    bwd_sweep_keys<NKB, DROP>(sK, sV, qf, df, lse_q, del_q, qb, cx, dQt);
    ATTN_STAMP(5);
    const int query = qb * 32 + r32;
    const bool q_ok = query < S;
    attn_store_row(dqkv + ((long)b * S + (q_ok ? query : 0)) * lddq + h * 64, dQt, h2, 1.0f, q_ok);
  }
  ATTN_STAMP(6);
}

// Waves per workgroup of the backward kernel: its four LDS images allow only one workgroup per CU once S > 128
// (NKB >= 5), so those cases run 8 waves (two per SIMD, one round over the <= 8 key / query blocks).
template <int NKB> constexpr int bwd_waves() { return NKB >= 5 ? 8 : 4; }
template <int NKB> constexpr int bwd_image_bytes() { return (NKB >= 4 ? 1 : 4 / NKB) * (4 * 32 * NKB * 128 + 32 * NKB * 8); }

// The element-wise work between the products is what bounds this kernel (rocprofv3 + ISA count, S = 197: ~490 vector
// instructions per 32 x 32 block against 16 MFMAs; matrix pipe busy 0.12): dropout is a template parameter (no
// per-element branches), the log-sum-exp arrives pre-multiplied by log2(e) and delta by the softmax scale (one fma each
// at staging instead of one multiply per element), and the dropout hash is shared by the two keys of a pair (common.h):
// in pass B a lane holds four consecutive keys of its query (two whole pairs); in pass A keys sit on the lanes, so lanes
// 2j and 2j + 1 split the sixteen queries of a block between them and exchange the hashes by DPP (quad_perm).
template <int NKB, bool DROP>
__global__ __launch_bounds__(64 * bwd_waves<NKB>(), NKB >= 5 ? 2 : 1) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ O,
                                                       const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                       bf16* __restrict__ dqkv, int total_pairs, int S, int H,
                                                       long ldq, long ldo, long lddo, long lddq, float scale,
                                                       DropKey dk) {
  constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
  constexpr int SP = 32 * NKB;
  constexpr int TILE = SP * 128;
  constexpr int PAIR_BYTES = 4 * TILE + SP * 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = H * 64;
  const int pair0 = blockIdx.x * PAIRS;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const float c_l2 = scale * 1.44269504088896340736f;
  const uint32_t half_stride = (uint32_t)((S + 1) >> 1);   // dropout element index: (pair * S + query) * S2 + key, S2 even
  const float keep_ds = dk.scale * scale;                  // multiplier of dP for a kept element, softmax scale folded in
  ATTN_STAMP(0);

  // ---- stage Q, K, V, dO; delta = rowsum(dO * O) * scale; lse * log2(e) (padded queries: +inf -> P = 0) ------------
  // This staging is 19k of the 42k cycles a workgroup lives at S = 197 (s_memtime stamps, scripts/attn_stamps.py):
  // 126 KB per workgroup in 128-byte pieces 4.6 KB (qkv) / 1.5 KB (dO, O) apart, ~6 B per cycle and CU.  Measured
  // alternatives, all on one box against this loop: every load of the workgroup requested before the first LDS store
  // +9 % kernel time; the four images by LDS-DMA (attn_stage_rows, as the forward does) +2 %; two images at a time and
  // four waves so that two workgroups share a CU (one staging while the other computes) +2 %; first-round workgroups
  // started 3-20 us apart +1..5 %.  The forward, with half the bytes per workgroup and two workgroups per CU, does gain
  // from the DMA form (-12 %).
#pragma unroll
  for (int pi = 0; pi < PAIRS; ++pi) {
    const int pair = pair0 + pi;
    const bool pv = pair < total_pairs;
    const int b = pv ? pair / H : 0, h = pv ? pair % H : 0;
    char* sQ = smem + pi * PAIR_BYTES;
    char* sK = sQ + TILE;
    char* sV = sK + TILE;
    char* sD = sV + TILE;
    float* sLse = reinterpret_cast<float*>(sD + TILE);
    float* sDel = sLse + SP;
    for (int id = t; id < SP * 8; id += 64 * bwd_waves<NKB>()) {
      const int row = id >> 3, c = id & 7;
      const bool ok = pv && row < S;
      const long tok = (long)b * S + row;
      const bf16* qp = qkv + tok * ldq + h * 64 + c * 8;
      const uint4 qv = ok ? *reinterpret_cast<const uint4*>(qp) : zero4;
      const uint4 kv = ok ? *reinterpret_cast<const uint4*>(qp + E) : zero4;
      const uint4 vv = ok ? *reinterpret_cast<const uint4*>(qp + 2 * E) : zero4;
      const uint4 dv = ok ? *reinterpret_cast<const uint4*>(dO + tok * lddo + h * 64 + c * 8) : zero4;
      const uint4 ov = ok ? *reinterpret_cast<const uint4*>(O + tok * ldo + h * 64 + c * 8) : zero4;
      const int o = du_off(row, c);
      *reinterpret_cast<uint4*>(sQ + o) = qv;
      *reinterpret_cast<uint4*>(sK + o) = kv;
      *reinterpret_cast<uint4*>(sV + o) = vv;
      *reinterpret_cast<uint4*>(sD + o) = dv;
      float d = bf16_lo(dv.x) * bf16_lo(ov.x) + bf16_hi(dv.x) * bf16_hi(ov.x) +
                bf16_lo(dv.y) * bf16_lo(ov.y) + bf16_hi(dv.y) * bf16_hi(ov.y) +
                bf16_lo(dv.z) * bf16_lo(ov.z) + bf16_hi(dv.z) * bf16_hi(ov.z) +
                bf16_lo(dv.w) * bf16_lo(ov.w) + bf16_hi(dv.w) * bf16_hi(ov.w);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if (c == 0) {
        sDel[row] = d * scale;
        sLse[row] = ok ? lse[(long)pair * S + row] * 1.44269504088896340736f : INFINITY;
      }
    }
  }
  ATTN_STAMP(1);
  __syncthreads();
  ATTN_STAMP(2);

  const int r32 = lane & 31, h2 = lane >> 5;

  // ================================ pass A: dK, dV ================================================
  for (int ki = wave; ki < PAIRS * NKB; ki += bwd_waves<NKB>()) {
    const int pi = ki / NKB, kb = ki - pi * NKB;
    const int pair = pair0 + pi;
    if (pair >= total_pairs || kb * 32 >= S) continue;
    const int b = pair / H, h = pair - b * H;
    const char* sQ = smem + pi * PAIR_BYTES;
    const char* sK = sQ + TILE;
    const char* sV = sK + TILE;
    const char* sD = sV + TILE;
    const float* sLse = reinterpret_cast<const float*>(sD + TILE);
    const float* sDel = sLse + SP;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(sK + du_off(kb * 32 + r32, ks * 2 + h2));
      vf[ks] = *reinterpret_cast<const bf16x8*>(sV + du_off(kb * 32 + r32, ks * 2 + h2));
    }
    f32x16 dVt[2], dKt[2];
    const BwdCtx cx{pair, S, half_stride, c_l2, scale, keep_ds, dk};
    bwd_sweep_queries<NKB, DROP>(sQ, sD, sLse, sDel, kf, vf, kb, cx, dVt, dKt);
    ATTN_STAMP(3);
    // (Restaging the three outputs through a wave-private LDS area so that every store writes whole 128-byte rows was
    // measured too: +5 % kernel time.  The swapped 16-byte stores are -1.5 % against 8-byte ones here, -5 % in the forward.)
    const int key = kb * 32 + r32;
    const bool k_ok = key < S;
    bf16* krow = dqkv + ((long)b * S + (k_ok ? key : 0)) * lddq + E + h * 64;
    attn_store_row(krow, dKt, h2, 1.0f, k_ok);
    attn_store_row(krow + E, dVt, h2, 1.0f, k_ok);
  }

  ATTN_STAMP(4);
  // ================================ pass B: dQ =====================================================
  for (int qi = wave; qi < PAIRS * NKB; qi += bwd_waves<NKB>()) {
    const int pi = qi / NKB, qb = qi - pi * NKB;
    const int pair = pair0 + pi;
    if (pair >= total_pairs || qb * 32 >= S) continue;
    const int b = pair / H, h = pair - b * H;
    const char* sQ = smem + pi * PAIR_BYTES;
    const char* sK = sQ + TILE;
    const char* sV = sK + TILE;
    const char* sD = sV + TILE;
    const float* sLse = reinterpret_cast<const float*>(sD + TILE);
    const float* sDel = sLse + SP;
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + du_off(qb * 32 + r32, ks * 2 + h2));
      df[ks] = *reinterpret_cast<const bf16x8*>(sD + du_off(qb * 32 + r32, ks * 2 + h2));
    }
    const float lse_q = sLse[qb * 32 + r32];
    const float del_q = sDel[qb * 32 + r32];
    f32x16 dQt[2];
    const BwdCtx cx{pair, S, half_stride, c_l2, scale, keep_ds, dk};
    bwd_sweep_keys<NKB, DROP>(sK, sV, qf, df, lse_q, del_q, qb, cx, dQt);
    ATTN_STAMP(5);
    const int query = qb * 32 + r32;
    const bool q_ok = query < S;
    attn_store_row(dqkv + ((long)b * S + (q_ok ? query : 0)) * lddq + h * 64, dQt, h2, 1.0f, q_ok);
  }
  ATTN_STAMP(6);
}

template <int NKB, bool DROP>
int launch_attn_bwd_t(const bf16* qkv, const bf16* O, const bf16* dO, const float* lse, bf16* dqkv, int B, int S, int H,
                    long ldq, long ldo, long lddo, long lddq, float scale, DropKey dk, hipStream_t st) {
  const int total = B * H;
  constexpr int PAIRS = NKB >= 4 ? 1 : 4 / NKB;
  constexpr int LDS = bwd_image_bytes<NKB>();
  if constexpr (NKB >= 5) {
    static const bool pipe = [] { const char* e = getenv("RVIT_ATTN_BWD_PIPE"); return !e || atoi(e) != 0; }();
    if (pipe) {
      static RvitLdsLimit lim_p;
      if (const int rc = lim_p.ensure({(const void*)attn_bwd_pipe_kernel<NKB, DROP>}, LDS, "attn_bwd")) return rc;
      hipLaunchKernelGGL((attn_bwd_pipe_kernel<NKB, DROP>), dim3((unsigned)total), dim3(512), LDS, st, qkv, O, dO, lse, dqkv,
                         total, S, H, ldq, ldo, lddo, lddq, scale, dk);
      RVIT_CHECK_LAUNCH("attn_bwd");
      return RVIT_OK;
    }
  }
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)attn_bwd_kernel<NKB, DROP>}, LDS, "attn_bwd")) return rc;
  const unsigned grid = (unsigned)((total + PAIRS - 1) / PAIRS);
  hipLaunchKernelGGL((attn_bwd_kernel<NKB, DROP>), dim3(grid), dim3(64 * bwd_waves<NKB>()), LDS, st, qkv, O, dO, lse, dqkv, total, S, H, ldq,
                     ldo, lddo, lddq, scale, dk);
  RVIT_CHECK_LAUNCH("attn_bwd");
  return RVIT_OK;
}
template <int NKB>
int launch_attn_bwd(const bf16* qkv, const bf16* O, const bf16* dO, const float* lse, bf16* dqkv, int B, int S, int H,
                    long ldq, long ldo, long lddo, long lddq, float scale, DropKey dk, hipStream_t st) {
  return dk.thresh ? launch_attn_bwd_t<NKB, true>(qkv, O, dO, lse, dqkv, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st)
                   : launch_attn_bwd_t<NKB, false>(qkv, O, dO, lse, dqkv, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
}

}  // namespace

#ifdef RVIT_ATTN_STAMPS
extern "C" int rvit_debug_attn_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), (size_t)n * 8);
}
#endif

int rvit_attn_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, long ldq, long ldo, float scale,
                       float drop_p, uint64_t drop_seed, hipStream_t st) {
  RVIT_CHECK_ARG(qkv && out, "attn_fwd: null pointer");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attn_fwd: drop_p must be in [0, 1)");
  // the dropout element index is 32 bits wide; without dropout there is no such limit on the batch
  RVIT_CHECK_ARG(drop_p == 0.f || (long)B * H * S * (S + (S & 1)) < 0xffffffffL, "attn_fwd: dropout element index exceeds 32 bits");
  const DropKey dk = make_drop_key(drop_p, drop_seed);
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256, "attn_fwd: unsupported shape B=%d H=%d S=%d (S<=256)", B, H, S);
  RVIT_CHECK_ARG(ldq >= 3L * H * 64 && (ldq & 7) == 0 && ldo >= (long)H * 64 && (ldo & 3) == 0,
                 "attn_fwd: bad pitches ldq=%ld ldo=%ld", ldq, ldo);
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && ((uintptr_t)out & 7) == 0, "attn_fwd: alignment");
  const bf16* q = (const bf16*)qkv;
  bf16* o = (bf16*)out;
  switch ((S + 31) / 32) {
    case 1: return launch_attn_fwd<1>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 2: return launch_attn_fwd<2>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 3: return launch_attn_fwd<3>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 4: return launch_attn_fwd<4>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 5: return launch_attn_fwd<5>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 6: return launch_attn_fwd<6>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 7: return launch_attn_fwd<7>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
    case 8: return launch_attn_fwd<8>(q, o, lse, B, S, H, ldq, ldo, scale, dk, st);
  }
  rvit_set_error("attn_fwd: S=%d unsupported", S);
  return RVIT_EUNSUPPORTED;
}

int rvit_attn_bwd_impl(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H,
                       long ldq, long ldo, long lddo, long lddq, float scale, float drop_p, uint64_t drop_seed,
                       hipStream_t st) {
  RVIT_CHECK_ARG(qkv && O && dO && lse && dqkv, "attn_bwd: null pointer");
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attn_bwd: drop_p must be in [0, 1)");
  RVIT_CHECK_ARG(drop_p == 0.f || (long)B * H * S * (S + (S & 1)) < 0xffffffffL, "attn_bwd: dropout element index exceeds 32 bits");
  const DropKey dk = make_drop_key(drop_p, drop_seed);
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256, "attn_bwd: unsupported shape B=%d H=%d S=%d (S<=256)", B, H, S);
  RVIT_CHECK_ARG(ldq >= 3L * H * 64 && lddq >= 3L * H * 64 && ldo >= (long)H * 64 && lddo >= (long)H * 64 &&
                     ((ldq | lddq | ldo | lddo) & 7) == 0,
                 "attn_bwd: bad pitches");
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && rvit_aligned16(O) && rvit_aligned16(dO) && rvit_aligned16(dqkv),
                 "attn_bwd: alignment");
  const bf16 *q = (const bf16*)qkv, *o = (const bf16*)O, *d = (const bf16*)dO;
  bf16* dq = (bf16*)dqkv;
  switch ((S + 31) / 32) {
    case 1: return launch_attn_bwd<1>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 2: return launch_attn_bwd<2>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 3: return launch_attn_bwd<3>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 4: return launch_attn_bwd<4>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 5: return launch_attn_bwd<5>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 6: return launch_attn_bwd<6>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 7: return launch_attn_bwd<7>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
    case 8: return launch_attn_bwd<8>(q, o, d, lse, dq, B, S, H, ldq, ldo, lddo, lddq, scale, dk, st);
  }
  rvit_set_error("attn_bwd: S=%d unsupported", S);
  return RVIT_EUNSUPPORTED;
}

int rvit_attn_fwd_fp8_impl(const void* qkv, uint8_t* out_q, long ldo, uint8_t* out_s, int B, int S, int H, long ldq,
                           float scale, hipStream_t st) {
  RVIT_CHECK_ARG(qkv && out_q && out_s, "attn_fwd_fp8: null pointer");
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256, "attn_fwd_fp8: unsupported shape B=%d H=%d S=%d (S<=256)", B, H, S);
  RVIT_CHECK_ARG((H & 1) == 0, "attn_fwd_fp8: H * 64 must be a multiple of 128 (MX operand of the out_proj), H=%d", H);
  RVIT_CHECK_ARG(ldq >= 3L * H * 64 && (ldq & 7) == 0 && ldo >= (long)H * 64 && (ldo & 3) == 0,
                 "attn_fwd_fp8: bad pitches ldq=%ld ldo=%ld", ldq, ldo);
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && ((uintptr_t)out_q & 3) == 0, "attn_fwd_fp8: alignment");
  RVIT_CHECK_ARG((long)B * H < 0x7fffffffL, "attn_fwd_fp8: too many (image, head) pairs");
  const bf16* q = (const bf16*)qkv;
  switch ((S + 31) / 32) {
    case 1: return launch_attn_fwd_fp8<1>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 2: return launch_attn_fwd_fp8<2>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 3: return launch_attn_fwd_fp8<3>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 4: return launch_attn_fwd_fp8<4>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 5: return launch_attn_fwd_fp8<5>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 6: return launch_attn_fwd_fp8<6>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 7: return launch_attn_fwd_fp8<7>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 8: return launch_attn_fwd_fp8<8>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
  }
  rvit_set_error("attn_fwd_fp8: S=%d unsupported", S);
  return RVIT_EUNSUPPORTED;
}

int rvit_attn_fwd_mx_impl(const void* qkv, uint8_t* out_q, long ldo, uint8_t* out_s, int B, int S, int H, long ldq,
                          float scale, hipStream_t st) {
  RVIT_CHECK_ARG(qkv && out_q && out_s, "attn_fwd_mx: null pointer");
  RVIT_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 256, "attn_fwd_mx: unsupported shape B=%d H=%d S=%d (S<=256)", B, H, S);
  RVIT_CHECK_ARG((H & 1) == 0, "attn_fwd_mx: H * 64 must be a multiple of 128 (MX operand of the out_proj), H=%d", H);
  RVIT_CHECK_ARG(ldq >= 3L * H * 64 && (ldq & 7) == 0 && ldo >= (long)H * 64 && (ldo & 3) == 0,
                 "attn_fwd_mx: bad pitches ldq=%ld ldo=%ld", ldq, ldo);
  RVIT_CHECK_ARG(rvit_aligned16(qkv) && ((uintptr_t)out_q & 3) == 0, "attn_fwd_mx: alignment");
  RVIT_CHECK_ARG((long)B * H < 0x7fffffffL, "attn_fwd_mx: too many (image, head) pairs");
  const bf16* q = (const bf16*)qkv;
  switch ((S + 31) / 32) {
    case 1: return launch_attn_fwd_mx<1>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 2: return launch_attn_fwd_mx<2>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 3: return launch_attn_fwd_mx<3>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 4: return launch_attn_fwd_mx<4>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 5: return launch_attn_fwd_mx<5>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 6: return launch_attn_fwd_mx<6>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 7: return launch_attn_fwd_mx<7>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
    case 8: return launch_attn_fwd_mx<8>(q, out_q, out_s, B, S, H, ldq, ldo, scale, st);
  }
  rvit_set_error("attn_fwd_mx: S=%d unsupported", S);
  return RVIT_EUNSUPPORTED;
}
