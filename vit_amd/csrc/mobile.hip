// HBM-bound kernels of the MobileNetV3 stem (the reference's `backbone='mobilenetv3'` = timm mobilenetv3_small_100,
// features_only, train_vit.py:68-74,87,145-146; inference): depthwise k x k convolution with folded BatchNorm and
// activation, the squeeze-excite global average pool, and the per-(image, channel) gate multiply.  The pointwise
// convolutions and the two squeeze-excite projections are plain GEMMs (gemm.hip) over the NHWC maps.
// All maps are NHWC bf16 with C % 8 == 0; a thread owns 8 consecutive channels of one pixel (16-byte accesses).
#include "common.h"
#include "rvit_internal.h"

namespace {

__device__ __forceinline__ void unpack8f(const uint4 u, float (&v)[8]) {
  v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
  v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
}
// the activations these stems use (the launchers refuse GELU, whose polynomial would be inlined at every call site)
__device__ __forceinline__ float mb_act(float x, int act) {
  const float r6 = fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
  return act == RVIT_ACT_RELU ? fmaxf(x, 0.f) : act == RVIT_ACT_HSWISH ? x * r6 : act == RVIT_ACT_HSIGMOID ? r6 : x;
}
__device__ __forceinline__ uint4 pack8f(const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

// out[b,ho,wo,c] = act(scale[c] * sum_{kh,kw} x[b, ho*s - pad + kh, wo*s - pad + kw, c] * w[kh*K + kw][c] + bias[c])
// w: [K*K][C] bf16 (tap-major, so the 8 weights of a thread's channels are one 16-byte load per tap).
template <int K>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                                     bf16* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo,
                                                     int stride, int pad, int act) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * Wo * cpr;
  if (i >= total) return;
  const int chunk = (int)(i % cpr);
  long pix = i / cpr;
  const int wo = (int)(pix % Wo);
  pix /= Wo;
  const int ho = (int)(pix % Ho);
  const int b = (int)(pix / Ho);
  const int c0 = chunk * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int hi0 = ho * stride - pad, wi0 = wo * stride - pad;
#pragma unroll
  for (int kh = 0; kh < K; ++kh) {
    const int hi = hi0 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int wi = wi0 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      float xv[8], wv[8];
      unpack8f(*reinterpret_cast<const uint4*>(x + (((long)b * H + hi) * W + wi) * C + c0), xv);
      unpack8f(*reinterpret_cast<const uint4*>(w + (long)(kh * K + kw) * C + c0), wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
    }
  }
  const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = mb_act(fmaf(acc[e], sc[e], bi[e]), act);
  *reinterpret_cast<uint4*>(out + (((long)b * Ho + ho) * Wo + wo) * C + c0) = pack8f(acc);
}


// Strip form of the depthwise convolution: a thread owns DW_TW consecutive output columns of one output row for its 8
// channels, so an input pixel is loaded once per kernel row instead of once per tap that touches it (K = 5, stride 1:
// 16 loads per 7 outputs and kernel row instead of 70), which is what bounds the one-output-per-thread kernel above
// (25 + 25 16-byte L1 reads per 16 bytes stored).  Loads are unconditional at clamped addresses and zeroed by select:
// predicated loads behind branches issue one at a time.  Accumulation order per output is kh-major, kw-minor, as above.
constexpr int DW_TW = 7;
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           bf16* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo,
                                                           int pad, int act) {
  constexpr int NC = (DW_TW - 1) * S + K;
  const int cpr = C >> 3, strips = Wo / DW_TW;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * strips * cpr;
  if (i >= total) return;
  const int chunk = (int)(i % cpr);
  long t = i / cpr;
  const int strip = (int)(t % strips);
  t /= strips;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int c0 = chunk * 8;
  float acc[DW_TW][8];
#pragma unroll
  for (int o = 0; o < DW_TW; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
  const int hi0 = ho * S - pad, wi0 = strip * DW_TW * S - pad;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1  // one kernel row's loads in flight at a time: unrolled, the compiler hoists all K rows and spills
  for (int kh = 0; kh < K; ++kh) {
    const int hi = hi0 + kh;
    const bool rok = (unsigned)hi < (unsigned)H;
    const bf16* row = x + ((long)b * H + min(max(hi, 0), H - 1)) * W * C + c0;
    uint4 xr[NC], wr[K];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int wi = wi0 + j;
      xr[j] = *reinterpret_cast<const uint4*>(row + (long)min(max(wi, 0), W - 1) * C);
      if (!(rok && (unsigned)wi < (unsigned)W)) xr[j] = zero;
    }
#pragma unroll
    for (int kw = 0; kw < K; ++kw) wr[kw] = *reinterpret_cast<const uint4*>(w + (long)(kh * K + kw) * C + c0);
    float wv[K][8];
#pragma unroll
    for (int kw = 0; kw < K; ++kw) unpack8f(wr[kw], wv[kw]);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      float xv[8];
      unpack8f(xr[j], xv);
      // column j feeds output o through tap kw = j - o*S; descending o = ascending kw for a fixed output is kept
      // because each (o, kw) pair appears once and outputs accumulate independently
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        if ((j - kw) % S != 0 || j - kw < 0 || (j - kw) / S >= DW_TW) continue;
        const int o = (j - kw) / S;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(xv[e], wv[kw][e], acc[o][e]);
      }
    }
  }
  const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  bf16* orow = out + (((long)b * Ho + ho) * Wo + strip * DW_TW) * C + c0;
#pragma unroll
  for (int o = 0; o < DW_TW; ++o) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[o][e] = mb_act(fmaf(acc[o][e], sc[e], bi[e]), act);
    *reinterpret_cast<uint4*>(orow + (long)o * C) = pack8f(acc[o]);
  }
}

// out[b, c] = mean over the HW pixels of x[b, :, c]  (bf16 out: the A operand of the squeeze-excite reduce GEMM).
// One workgroup per image; thread -> (8-channel chunk, row lane); row lanes combined through LDS in lane order.
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int HW, int C) {
  extern __shared__ float ap_red[];  // [rows_per_pass][C]
  const int cpr = C >> 3;
  const int rpp = 256 / cpr;
  const int chunk = threadIdx.x % cpr, r = threadIdx.x / cpr;
  const bf16* base = x + (long)blockIdx.x * HW * C + chunk * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (r < rpp) {
    for (int p = r; p < HW; p += rpp) {
      float v[8];
      unpack8f(*reinterpret_cast<const uint4*>(base + (long)p * C), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ap_red[r * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  const float inv = 1.0f / (float)HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += ap_red[rr * C + c];
    out[(long)blockIdx.x * C + c] = (bf16)(t * inv);
  }
}

// x[b, p, c] *= gate[b, c]  (squeeze-excite), in place
__global__ __launch_bounds__(256) void channel_scale_kernel(bf16* __restrict__ x, const bf16* __restrict__ gate, long total,
                                                            int HW, int C) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i % cpr);
  const long pix = i / cpr;
  const long b = pix / HW;
  float v[8], g[8];
  bf16* p = x + pix * C + chunk * 8;
  unpack8f(*reinterpret_cast<const uint4*>(p), v);
  unpack8f(*reinterpret_cast<const uint4*>(gate + b * C + chunk * 8), g);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= g[e];
  *reinterpret_cast<uint4*>(p) = pack8f(v);
}


// Squeeze-excite in one kernel, one workgroup per image, the image's map resident in LDS between the pool and the gate
// multiply (the maps of mobilenetv3_small_100 that carry one are 25-94 KB): x is read once and written once, and the
// pool / reduce projection / expand projection / multiply of the four-launch form (avgpool, 2 x gemm, channel_scale)
// become one launch.  Same rounding points as that form: pooled mean, ReLU output and gate are rounded to bf16.
//   x [B, HW, C] bf16 in place; w1 [RD, C], w2 [C, RD] bf16; b1 [RD], b2 [C] f32.
constexpr int SE_THREADS = 512;
constexpr int SE_INFLIGHT = 8;
__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16)v; }
// out[o] = epi(sum_j w[o][j] * vin[j]) for o < nout: 2^lpo lanes share an output and split its row into 16-byte chunks,
// SE_INFLIGHT loads in flight per lane (clamped addresses, so the loads never sit behind a branch); the projections are a few
// hundred KB of L2 reads per image and latency, not bandwidth, is what they cost.
template <typename Epi>
__device__ __forceinline__ void se_matvec(const bf16* __restrict__ w, int nout, int nin, const float* vin, int lpo, Epi&& epi) {
  const int tid = threadIdx.x, g = tid >> lpo, gl = tid & ((1 << lpo) - 1), ngroups = SE_THREADS >> lpo, step = 1 << lpo;
  const int chunks = nin >> 3;
  for (int o0 = 0; o0 < nout; o0 += ngroups) {   // uniform trip count: every lane reaches the shuffles
    const int o = o0 + g;
    const bf16* row = w + (long)min(o, nout - 1) * nin;
    float a = 0.f;
    for (int j0 = gl; j0 < chunks; j0 += SE_INFLIGHT * step) {
      uint4 u[SE_INFLIGHT];
#pragma unroll
      for (int q = 0; q < SE_INFLIGHT; ++q) u[q] = *reinterpret_cast<const uint4*>(row + min(j0 + q * step, chunks - 1) * 8);
#pragma unroll
      for (int q = 0; q < SE_INFLIGHT; ++q) {
        const int j = j0 + q * step;
        if (j < chunks) {
          float wv[8];
          unpack8f(u[q], wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) a = fmaf(wv[e], vin[j * 8 + e], a);
        }
      }
    }
    for (int m = step >> 1; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
    if (gl == 0 && o < nout) epi(o, a);
  }
}

__global__ __launch_bounds__(SE_THREADS) void se_fused_kernel(bf16* __restrict__ x, const bf16* __restrict__ w1,
                                                              const float* __restrict__ b1, const bf16* __restrict__ w2,
                                                              const float* __restrict__ b2, int HW, int C, int RD,
                                                              int map_bytes, int lpo1, int lpo2) {
  extern __shared__ __align__(16) unsigned char se_smem[];
  const int cpr = C >> 3, rpp = SE_THREADS / cpr;
  uint4* smap = reinterpret_cast<uint4*>(se_smem);
  float* red = reinterpret_cast<float*>(se_smem + map_bytes);  // [rpp][C]
  float* pooled = red + (long)rpp * C;                          // [C]
  float* gate = pooled + C;                                     // [C]
  float* hid = gate + C;                                        // [RD]
  const int tid = threadIdx.x, chunk = tid % cpr, r = tid / cpr;
  bf16* base = x + (long)blockIdx.x * HW * C + chunk * 8;
  if (r < rpp) {
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int p = r; p < HW; p += rpp) {
      const uint4 u = *reinterpret_cast<const uint4*>(base + (long)p * C);
      smap[p * cpr + chunk] = u;
      float v[8];
      unpack8f(u, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r * C + chunk * 8 + e] = s[e];
  }
  __syncthreads();
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += SE_THREADS) {
    float t = 0.f;
    for (int rr = 0; rr < rpp; ++rr) t += red[rr * C + c];
    pooled[c] = bf16_round(t * inv);
  }
  __syncthreads();
  se_matvec(w1, RD, C, pooled, lpo1, [&](int o, float a) { hid[o] = bf16_round(fmaxf(a + b1[o], 0.f)); });
  __syncthreads();
  se_matvec(w2, C, RD, hid, lpo2, [&](int c, float a) { gate[c] = bf16_round(mb_act(a + b2[c], RVIT_ACT_HSIGMOID)); });
  __syncthreads();
  if (r < rpp) {
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gate[chunk * 8 + e];
    for (int p = r; p < HW; p += rpp) {
      float v[8];
      unpack8f(smap[p * cpr + chunk], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= g[e];
      *reinterpret_cast<uint4*>(base + (long)p * C) = pack8f(v);
    }
  }
}

// First layer of the MobileNetV3 stem straight from the model's input: conv 3x3 / stride 2 / pad 1, 3 -> 16 channels,
// over x f32 [B,3,H,W] (NCHW, H and W even) + folded BatchNorm + activation, out bf16 [B,H/2,W/2,16] (NHWC).  One thread
// per output pixel with its 16 accumulators in registers; the 432 weights are wave-uniform (scalar loads, used as the
// scalar operand of the FMAs), the 27 inputs are 9 (float, float2) pairs at clamped addresses.  Reads 12 B and writes
// 8 B per output channel-vector: HBM-bound once the 432 FMAs per pixel are hidden.
//   w f32 [27][16], k = (kh*3 + kw)*3 + ci.
__global__ __launch_bounds__(256) void conv3x3s2_rgb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            bf16* __restrict__ out, int B, int H, int W, int act) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * Ho * Wo) return;
  const int wo = (int)(i % Wo);
  const int ho = (int)((i / Wo) % Ho);
  const long b = i / ((long)Wo * Ho);
  // (kh, ci) is a real loop of 9 steps, 48 weights (three s_load_dwordx16) and 48 FMAs each: fully unrolled, the
  // scheduler walks channel by channel, wants all 432 weights in scalar registers at once and spills them through
  // vector lanes (a v_readlane per FMA: 1.07 ms at B = 512).  The next step's three inputs are loaded before this
  // step's FMAs.  Only hi = -1 and wi = -1 can fall outside the image (H, W even).
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  const float* img = x + b * 3 * (long)H * W;
  const long plane = (long)H * W;
  const int wl = max(2 * wo - 1, 0);
  auto load3 = [&](int step, float& l, float2& cr) {
    const int kh = step / 3, ci = step - kh * 3;
    const float* row = img + ci * plane + (long)max(2 * ho - 1 + kh, 0) * W;
    l = row[wl];
    cr = *reinterpret_cast<const float2*>(row + 2 * wo);
  };
  float l, ln;
  float2 cr, crn;
  load3(0, l, cr);
#pragma unroll 1
  for (int step = 0; step < 9; ++step) {
    load3(min(step + 1, 8), ln, crn);
    const int kh = step / 3, ci = step - kh * 3;
    const bool rok = 2 * ho - 1 + kh >= 0;
    const float x0 = (rok && wo > 0) ? l : 0.f, x1 = rok ? cr.x : 0.f, x2 = rok ? cr.y : 0.f;
    const float* wk = w + (kh * 9 + ci) * 16;   // k = (kh*3 + kw)*3 + ci
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = fmaf(x0, wk[c], acc[c]);
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = fmaf(x1, wk[48 + c], acc[c]);
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = fmaf(x2, wk[96 + c], acc[c]);
    l = ln;
    cr = crn;
  }
  float o0[8], o1[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    o0[c] = mb_act(fmaf(acc[c], scale[c], bias[c]), act);
    o1[c] = mb_act(fmaf(acc[c + 8], scale[c + 8], bias[c + 8]), act);
  }
  uint4* dst = reinterpret_cast<uint4*>(out + i * 16);
  dst[0] = pack8f(o0);
  dst[1] = pack8f(o1);
}

// Pointwise (1x1) convolution over an NHWC map viewed as X [M, K] with M = B*H*W in the millions and K, N <= a few
// hundred: out[m, n] = act(scale[n] * sum_k X[m, k] W[n, k] + bias[n] + residual[m, n]).  The general GEMM (gemm.hip) is
// built around a K loop; here K is 1-8 MFMA steps, so its tile prologue / barrier / epilogue per 256 x 64 tile is most of
// its time (16 -> 72 at 56 x 56 x 512: 154 us for 282 MB).  This kernel keeps the whole weight matrix in LDS as ready-made
// MFMA operands (weights on the A side of v_mfma_f32_16x16x32_bf16, so a lane ends with 4 consecutive output channels of
// one pixel), and each wave streams 16-pixel groups: KS 16-byte loads per lane (next group's issued before this group's
// MFMAs), NB*KS MFMAs, NB 8-byte stores.  No barrier after the prologue; HBM-bound: 2*(K + N) bytes per pixel.
constexpr int PW_THREADS = 256;
constexpr long RVIT_PW_TARGET_WAVES = 8192;
constexpr long RVIT_PW_STAGED_LDS_BYTES = 24 * 1024;
template <int KS>
__global__ __launch_bounds__(PW_THREADS) void pointwise_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                               const float* __restrict__ scale, const float* __restrict__ bias,
                                                               const bf16* __restrict__ res, bf16* __restrict__ out, long M,
                                                               int K, int N, int NB, int groups_per_wave, int act,
                                                               int staged) {
  extern __shared__ __align__(16) unsigned char pw_smem[];
  uint4* sw = reinterpret_cast<uint4*>(pw_smem);                       // [NB][KS][64 lanes] A operands
  float* ssc = reinterpret_cast<float*>(pw_smem + (long)NB * KS * 1024);  // [NB*16] scale, then [NB*16] bias
  float* sbi = ssc + NB * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16* tile = reinterpret_cast<bf16*>(sbi + NB * 16) + (long)wave * 16 * N;   // [16][N] per wave: output staging
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < NB * KS * 64; i += PW_THREADS) {
    const int l = i & 63, t = i >> 6, st = t % KS, nb = t / KS;
    const int n = nb * 16 + (l & 15), k0 = (st * 4 + (l >> 4)) * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(w + (long)min(n, N - 1) * K + min(k0, K - 8));
    sw[i] = (n < N && k0 < K) ? u : zero;
  }
  for (int i = tid; i < NB * 16; i += PW_THREADS) {
    ssc[i] = i < N ? (scale ? scale[i] : 1.f) : 0.f;
    sbi[i] = (i < N && bias) ? bias[i] : 0.f;
  }
  __syncthreads();
  const long ngroups = (M + 15) >> 4;
  const long g0 = ((long)blockIdx.x * (PW_THREADS / 64) + wave) * groups_per_wave;
  if (g0 >= ngroups) return;
  const int k_lane = (lane >> 4) * 8;
  auto loadx = [&](long g, uint4 (&xr)[KS]) {
    const bf16* row = x + min(g * 16 + (lane & 15), M - 1) * K;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
      const int k0 = st * 32 + k_lane;
      const uint4 u = *reinterpret_cast<const uint4*>(row + min(k0, K - 8));
      xr[st] = k0 < K ? u : zero;
    }
  };
  uint4 xc[KS], xn[KS];
  loadx(g0, xc);
  for (int gi = 0; gi < groups_per_wave; ++gi) {
    const long g = g0 + gi;
    if (g >= ngroups) break;  // wave-uniform
    loadx(min(g + 1, ngroups - 1), xn);
    const long m = g * 16 + (lane & 15);
    for (int nb = 0; nb < NB; ++nb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < KS; ++st)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sw[(nb * KS + st) * 64 + lane]),
                                                      __builtin_bit_cast(bf16x8, xc[st]), acc, 0, 0, 0);
      const int n0 = nb * 16 + (lane >> 4) * 4;
      if (n0 < N) {
        const float4 sc = *reinterpret_cast<const float4*>(ssc + n0), bi = *reinterpret_cast<const float4*>(sbi + n0);
        float v[4] = {fmaf(acc[0], sc.x, bi.x), fmaf(acc[1], sc.y, bi.y), fmaf(acc[2], sc.z, bi.z), fmaf(acc[3], sc.w, bi.w)};
        if (res) {
          const uint2 rr = *reinterpret_cast<const uint2*>(res + min(m, M - 1) * N + n0);
          v[0] += bf16_lo(rr.x); v[1] += bf16_hi(rr.x); v[2] += bf16_lo(rr.y); v[3] += bf16_hi(rr.y);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mb_act(v[e], act);
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        if (staged) *reinterpret_cast<uint2*>(tile + (lane & 15) * N + n0) = o;
        else if (m < M) *reinterpret_cast<uint2*>(out + m * N + n0) = o;
      }
    }
    if (!staged) {
#pragma unroll
      for (int st = 0; st < KS; ++st) xc[st] = xn[st];
      continue;
    }
    // the group's 16 x N outputs are one contiguous run of HBM: out through LDS as whole 16-byte pieces in address
    // order (8-byte stores straight from the MFMA layout are 32-byte pieces of 16 different rows each: 16 -> 72 at
    // 56 x 56 x 512 104 us, staged 89 us).  Only while the tiles do not cost occupancy (the launcher decides): at
    // N >= 240 they halve the workgroups per CU and the direct stores are faster (40 -> 240: 35 us vs 53 us).
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    {
      const int rows = (int)min((long)16, M - g * 16);
      const int pieces = rows * N >> 3;
      uint4* dst = reinterpret_cast<uint4*>(out + g * 16 * N);
      const uint4* src = reinterpret_cast<const uint4*>(tile);
      for (int i = lane; i < pieces; i += 64) dst[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < KS; ++st) xc[st] = xn[st];
  }
}

}  // namespace

int rvit_dwconv_impl(const void* x, const void* w, const float* scale, const float* bias, void* out, int B, int H, int W,
                     int C, int K, int stride, int pad, int act, hipStream_t st) {
  RVIT_CHECK_ARG(x && w && scale && bias && out, "dwconv: null pointer");
  RVIT_CHECK_ARG(B >= 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "dwconv: bad shape B=%d H=%d W=%d C=%d (C %% 8 == 0)", B, H, W, C);
  RVIT_CHECK_ARG((K == 3 || K == 5) && (stride == 1 || stride == 2) && pad >= 0 && pad < K, "dwconv: K in {3,5}, stride in {1,2}");
  RVIT_CHECK_ARG(act >= RVIT_ACT_NONE && act <= RVIT_ACT_HSIGMOID && act != RVIT_ACT_GELU, "dwconv: bad activation %d", act);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w) && rvit_aligned16(out) && rvit_aligned16(scale) && rvit_aligned16(bias),
                 "dwconv: operands must be 16-byte aligned");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  if (Ho <= 0 || Wo <= 0 || B == 0) return RVIT_OK;
  if (Wo % DW_TW == 0) {
    const long threads = (long)B * Ho * (Wo / DW_TW) * (C >> 3);
    RVIT_CHECK_ARG((threads + 255) / 256 < 0x7fffffffL, "dwconv: too many elements");
    const dim3 sgrid((unsigned)((threads + 255) / 256));
#define RVIT_DW_STRIP(KK, SS)                                                                                          \
  hipLaunchKernelGGL((dwconv_strip_kernel<KK, SS>), sgrid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, scale, bias, \
                     (bf16*)out, B, H, W, C, Ho, Wo, pad, act)
    if (K == 3 && stride == 1) RVIT_DW_STRIP(3, 1);
    else if (K == 3) RVIT_DW_STRIP(3, 2);
    else if (stride == 1) RVIT_DW_STRIP(5, 1);
    else RVIT_DW_STRIP(5, 2);
#undef RVIT_DW_STRIP
    RVIT_CHECK_LAUNCH("dwconv_strip");
    return RVIT_OK;
  }
  const long total = (long)B * Ho * Wo * (C >> 3);
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "dwconv: too many elements");
  const dim3 grid((unsigned)((total + 255) / 256));
  if (K == 3) hipLaunchKernelGGL(dwconv_kernel<3>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, scale, bias, (bf16*)out, B, H, W, C, Ho, Wo, stride, pad, act);
  else hipLaunchKernelGGL(dwconv_kernel<5>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, scale, bias, (bf16*)out, B, H, W, C, Ho, Wo, stride, pad, act);
  RVIT_CHECK_LAUNCH("dwconv");
  return RVIT_OK;
}

int rvit_avgpool_impl(const void* x, void* out, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && out && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0 && C <= 2048, "avgpool: bad arguments (C %% 8 == 0, C <= 2048)");
  RVIT_CHECK_ARG(rvit_aligned16(x), "avgpool: x must be 16-byte aligned");
  if (B == 0) return RVIT_OK;
  const int rpp = 256 / (C >> 3);
  RVIT_CHECK_ARG(rpp >= 1, "avgpool: C too large");
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)B), dim3(256), (size_t)rpp * C * sizeof(float), st, (const bf16*)x, (bf16*)out, HW, C);
  RVIT_CHECK_LAUNCH("avgpool");
  return RVIT_OK;
}

int rvit_channel_scale_impl(void* x, const void* gate, int B, int HW, int C, hipStream_t st) {
  RVIT_CHECK_ARG(x && gate && B >= 0 && HW > 0 && C > 0 && (C & 7) == 0, "channel_scale: bad arguments");
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(gate), "channel_scale: operands must be 16-byte aligned");
  const long total = (long)B * HW * (C >> 3);
  if (total <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "channel_scale: too many elements");
  hipLaunchKernelGGL(channel_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (bf16*)x, (const bf16*)gate, total, HW, C);
  RVIT_CHECK_LAUNCH("channel_scale");
  return RVIT_OK;
}

// lanes per output (log2) of se_matvec: fewest sequential L2 round trips (one per SE_INFLIGHT loads of a lane and per
// round of outputs), then fewest loads and shuffles
static int se_lanes_log2(int nout, int nin) {
  int best = 0;
  long best_cost = -1;
  for (int l = 0; l <= 6; ++l) {
    const int groups = SE_THREADS >> l, chunks = nin >> 3;
    const long rounds = (nout + groups - 1) / groups, loads = (chunks + (1 << l) - 1) >> l;
    const long cost = rounds * (700 * ((loads + SE_INFLIGHT - 1) / SE_INFLIGHT) + 20 * loads + 30 * l);
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best = l;
  }
  return best;
}

int rvit_squeeze_excite_impl(void* x, const void* w1, const float* b1, const void* w2, const float* b2, int B, int HW, int C,
                             int RD, hipStream_t st) {
  RVIT_CHECK_ARG(x && w1 && b1 && w2 && b2, "squeeze_excite: null pointer");
  RVIT_CHECK_ARG(B >= 0 && HW > 0 && C > 0 && RD > 0 && (C & 7) == 0 && (RD & 7) == 0 && C <= 8 * SE_THREADS,
                 "squeeze_excite: bad shape HW=%d C=%d RD=%d (C %% 8 == 0, RD %% 8 == 0, C <= %d)", HW, C, RD, 8 * SE_THREADS);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w1) && rvit_aligned16(w2), "squeeze_excite: operands must be 16-byte aligned");
  const long map_bytes = (long)HW * C * 2;
  const int rpp = SE_THREADS / (C >> 3);
  const long smem = map_bytes + ((long)rpp * C + 2 * C + RD) * (long)sizeof(float);
  RVIT_CHECK_ARG(smem <= RVIT_SE_MAX_LDS_BYTES, "squeeze_excite: one image's map (%ld bytes) does not fit in LDS; use "
                 "rvit_avgpool_nhwc + rvit_gemm_bf16 x 2 + rvit_channel_scale_nhwc", map_bytes);
  if (B == 0) return RVIT_OK;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)se_fused_kernel}, (int)smem, "squeeze_excite")) return rc;
  hipLaunchKernelGGL(se_fused_kernel, dim3((unsigned)B), dim3(SE_THREADS), (size_t)smem, st, (bf16*)x, (const bf16*)w1, b1,
                     (const bf16*)w2, b2, HW, C, RD, (int)map_bytes, se_lanes_log2(RD, C), se_lanes_log2(C, RD));
  RVIT_CHECK_LAUNCH("squeeze_excite");
  return RVIT_OK;
}

int rvit_conv3x3s2_rgb_impl(const float* x, const float* w, const float* scale, const float* bias, void* out, int B, int H,
                            int W, int Cout, int act, hipStream_t st) {
  RVIT_CHECK_ARG(x && w && scale && bias && out, "conv3x3s2_rgb: null pointer");
  RVIT_CHECK_ARG(B >= 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && Cout == 16,
                 "conv3x3s2_rgb: H and W even, Cout == 16 (got H=%d W=%d Cout=%d)", H, W, Cout);
  RVIT_CHECK_ARG(act >= RVIT_ACT_NONE && act <= RVIT_ACT_HSIGMOID && act != RVIT_ACT_GELU, "conv3x3s2_rgb: bad activation %d", act);
  RVIT_CHECK_ARG(((uintptr_t)x & 7) == 0 && rvit_aligned16(out), "conv3x3s2_rgb: x must be 8-byte and out 16-byte aligned");
  const long total = (long)B * (H / 2) * (W / 2);
  if (total <= 0) return RVIT_OK;
  RVIT_CHECK_ARG((total + 255) / 256 < 0x7fffffffL, "conv3x3s2_rgb: too many pixels");
  hipLaunchKernelGGL(conv3x3s2_rgb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, w, scale, bias,
                     (bf16*)out, B, H, W, act);
  RVIT_CHECK_LAUNCH("conv3x3s2_rgb");
  return RVIT_OK;
}

// LDS the pointwise kernel needs for an [N, K] weight matrix, or -1 when the shape is outside what it takes.
long rvit_pointwise_lds_bytes(int K, int N) {
  if (K < 8 || N < 8 || (K & 7) || (N & 7) || K > 256) return -1;
  int ks = (K + 31) / 32;
  if (ks > 5) ks = 8;
  const long nb = (N + 15) / 16;
  long bytes = nb * ks * 1024 + nb * 16 * 8;
  const long tiles = (PW_THREADS / 64) * 16L * N * 2;
  if (bytes + tiles <= RVIT_PW_STAGED_LDS_BYTES) bytes += tiles;   // output staging only while 6+ workgroups fit a CU
  return bytes <= RVIT_POINTWISE_MAX_LDS_BYTES ? bytes : -1;
}

int rvit_pointwise_impl(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* out,
                        long M, int K, int N, int act, hipStream_t st) {
  RVIT_CHECK_ARG(x && w && out, "pointwise_conv: null pointer");
  RVIT_CHECK_ARG(M >= 0, "pointwise_conv: M < 0");
  const long lds = rvit_pointwise_lds_bytes(K, N);
  RVIT_CHECK_ARG(lds > 0, "pointwise_conv: K=%d N=%d outside this kernel (K, N %% 8 == 0, K <= 256, weights <= %d bytes as "
                 "MFMA operands); use rvit_gemm_bf16", K, N, RVIT_POINTWISE_MAX_LDS_BYTES);
  RVIT_CHECK_ARG(act >= RVIT_ACT_NONE && act <= RVIT_ACT_HSIGMOID && act != RVIT_ACT_GELU, "pointwise_conv: bad activation %d", act);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(w) && rvit_aligned16(out) && (((uintptr_t)residual) & 7) == 0,
                 "pointwise_conv: x, w, out must be 16-byte aligned (residual 8)");
  if (M == 0) return RVIT_OK;
  int ks = (K + 31) / 32;
  if (ks > 5) ks = 8;
  const int nb = (N + 15) / 16;
  const long ngroups = (M + 15) / 16;
  // a wave has one 16-pixel group of loads in flight (1-3 KB): the stream needs all 8 wave slots of every SIMD filled
  // (8192 waves) before the bytes in flight cover the HBM round trip; more groups per wave only amortise the prologue
  long gpw = ngroups / RVIT_PW_TARGET_WAVES;
  gpw = gpw < 1 ? 1 : gpw > 32 ? 32 : gpw;
  const long wgs = (ngroups + 4 * gpw - 1) / (4 * gpw);
  RVIT_CHECK_ARG(wgs < 0x7fffffffL, "pointwise_conv: too many rows");
  const int staged = lds > (long)nb * ks * 1024 + nb * 16 * 8 ? 1 : 0;
  static RvitLdsLimit lim;
  if (const int rc = lim.ensure({(const void*)pointwise_kernel<1>, (const void*)pointwise_kernel<2>, (const void*)pointwise_kernel<3>,
                                 (const void*)pointwise_kernel<4>, (const void*)pointwise_kernel<5>, (const void*)pointwise_kernel<8>},
                                RVIT_POINTWISE_MAX_LDS_BYTES, "pointwise_conv")) return rc;
#define RVIT_PW(KSV)                                                                                                     \
  hipLaunchKernelGGL(pointwise_kernel<KSV>, dim3((unsigned)wgs), dim3(PW_THREADS), (size_t)lds, st, (const bf16*)x,    \
                     (const bf16*)w, scale, bias, (const bf16*)residual, (bf16*)out, M, K, N, nb, (int)gpw, act, staged)
  switch (ks) {
    case 1: RVIT_PW(1); break;
    case 2: RVIT_PW(2); break;
    case 3: RVIT_PW(3); break;
    case 4: RVIT_PW(4); break;
    case 5: RVIT_PW(5); break;
    default: RVIT_PW(8); break;
  }
#undef RVIT_PW
  RVIT_CHECK_LAUNCH("pointwise_conv");
  return RVIT_OK;
}
