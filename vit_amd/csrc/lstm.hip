// LSTM / GRU time step, forward and backward, both directions in one launch: the temporal encoder of the reference's
// sequence model (nn.LSTM(E, 256, num_layers=2, batch_first, bidirectional), train_vit.py:188-229,
// infer.py:186-227) and its autograd (train_vit.py:939).  torch gate order i, f, g, o:
//   pre = W_ih x_t + b_ih + W_hh h_prev + b_hh;  i,f,o = sigmoid, g = tanh;  c = f*c_prev + i*g;  h = o*tanh(c)
// The recurrence is tiny next to the spatial encoder (5 MFLOP against 3.97 GFLOP per frame) and strictly
// sequential, so it runs in fp32 on the vector ALUs straight from the fp32 parameters: no packing, no bf16
// rounding inside the recurrence.  A workgroup owns LS_JB hidden units of one direction for LS_BC samples:
// 256 threads = 32 gate rows x 8 K-parts (forward) or 8 units x 32 K-parts (backward); the step's input
// vectors sit in LDS, the weight rows stream through registers once per chunk of 8 samples.
#include "common.h"
#include "rvit_internal.h"

namespace {

constexpr int LS_JB = 8;   // hidden units per workgroup
constexpr int LS_BC = 8;   // samples per pass over the weights

__device__ __forceinline__ float ls_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct LstmDir {
  const float* w_ih;  // [4H, In]
  const float* w_hh;  // [4H, H]   (backward kernel: W_hh^T [H, 4H])
  const float* b_ih;  // [4H]
  const float* b_hh;  // [4H]
};

// One forward step s of both directions.  Direction d works on time t = d ? T-1-s : s and takes h_prev from the
// output sequence at the neighbouring time (zero at s == 0).  out [B,T,ndir*H]; c [ndir,B,H] updated in place
// (a (sample, unit) pair belongs to one thread); gates [B,T,ndir,4H] and cseq [B,T,ndir,H] saved when training.
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const float* __restrict__ x, LstmDir d0, LstmDir d1,
                                                            float* __restrict__ out, float* __restrict__ c,
                                                            float* __restrict__ gates, float* __restrict__ cseq, int B,
                                                            int T, int In, int H, int s, int ndir) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = In + H;
  float* xh = sm;                       // [LS_BC][K]
  float* pre = sm + LS_BC * K;          // [32][LS_BC]
  const int dir = blockIdx.y;
  const LstmDir w = dir ? d1 : d0;
  const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
  const int j0 = blockIdx.x * LS_JB;
  const int r = threadIdx.x >> 3, p = threadIdx.x & 7;   // gate row r = q*8 + jj, K part p
  const int q = r >> 3, jj = r & 7;
  const int row = q * H + j0 + jj;
  const float* wi = w.w_ih + (long)row * In;
  const float* wh = w.w_hh + (long)row * H;
  const int OW = ndir * H;
  {  // one chunk of LS_BC samples per workgroup: blockIdx.z (samples are independent)
    const int b0 = blockIdx.z * LS_BC;
    const int nb = B - b0 < LS_BC ? B - b0 : LS_BC;
    __syncthreads();   // previous chunk's cell update has read `pre`
    for (int i = threadIdx.x; i < LS_BC * K; i += 256) {
      const int bb = i / K, k = i - bb * K;
      float v = 0.f;
      if (bb < nb) {
        if (k < In) v = x[((long)(b0 + bb) * T + t) * In + k];
        else if (s > 0) v = out[((long)(b0 + bb) * T + tp) * OW + dir * H + (k - In)];
      }
      xh[i] = v;
    }
    __syncthreads();
    float acc[LS_BC];
#pragma unroll
    for (int bb = 0; bb < LS_BC; ++bb) acc[bb] = 0.f;
    for (int k = p * 4; k < K; k += 32) {   // In and H are multiples of 4: a float4 never straddles the two matrices
      const float4 wv = k < In ? *reinterpret_cast<const float4*>(wi + k) : *reinterpret_cast<const float4*>(wh + (k - In));
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) {
        const float4 xv = *reinterpret_cast<const float4*>(xh + bb * K + k);
        acc[bb] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
      }
    }
#pragma unroll
    for (int bb = 0; bb < LS_BC; ++bb) {
      float v = acc[bb];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      if (p == 0) pre[r * LS_BC + bb] = v + w.b_ih[row] + w.b_hh[row];
    }
    __syncthreads();
    if (threadIdx.x < LS_JB * LS_BC) {
      const int bb = threadIdx.x / LS_JB, j = threadIdx.x % LS_JB;
      if (bb < nb) {
        const int b = b0 + bb, hj = j0 + j;
        const float gi = ls_sigmoid(pre[(0 * 8 + j) * LS_BC + bb]), gf = ls_sigmoid(pre[(1 * 8 + j) * LS_BC + bb]);
        const float gg = tanhf(pre[(2 * 8 + j) * LS_BC + bb]), go = ls_sigmoid(pre[(3 * 8 + j) * LS_BC + bb]);
        float* cp = c + ((long)dir * B + b) * H + hj;
        const float cn = gf * (s > 0 ? *cp : 0.f) + gi * gg;
        *cp = cn;
        out[((long)b * T + t) * OW + dir * H + hj] = go * tanhf(cn);
        if (gates) {
          float* g = gates + (((long)b * T + t) * ndir + dir) * 4 * H + hj;
          g[0] = gi; g[H] = gf; g[2 * H] = gg; g[3 * H] = go;
          cseq[(((long)b * T + t) * ndir + dir) * H + hj] = cn;
        }
      }
    }
  }
}

// One backward step s (run s = T-1 .. 0).  dh = dout[b,t] + dG[b,t_next] . W_hh (the step processed just before,
// absent at s == T-1); the pointwise part turns dh and the carried dc into the pre-activation gradients
// dG[b,t,dir,4H] and the new carry.  whh_t = W_hh^T [H,4H] so that the rows a workgroup needs are contiguous.
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float* __restrict__ dout, LstmDir d0, LstmDir d1,
                                                            const float* __restrict__ gates,
                                                            const float* __restrict__ cseq, float* __restrict__ dG,
                                                            float* __restrict__ dc, int B, int T, int H, int s,
                                                            int ndir) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int G4 = 4 * H;
  float* dgn = sm;                      // [LS_BC][4H]: dG of the step processed before this one
  float* dhr = sm + LS_BC * G4;         // [LS_JB][LS_BC]
  const int dir = blockIdx.y;
  const float* whh_t = (dir ? d1 : d0).w_hh;
  const int t = dir ? T - 1 - s : s;
  const int tn = dir ? t - 1 : t + 1;   // time of forward step s + 1
  const int tpv = dir ? t + 1 : t - 1;  // time of forward step s - 1
  const int j0 = blockIdx.x * LS_JB;
  const int jj = threadIdx.x >> 5, p = threadIdx.x & 31;
  const float* wr = whh_t + (long)(j0 + jj) * G4;
  const int OW = ndir * H;
  {  // one chunk of LS_BC samples per workgroup: blockIdx.z (samples are independent)
    const int b0 = blockIdx.z * LS_BC;
    const int nb = B - b0 < LS_BC ? B - b0 : LS_BC;
    __syncthreads();
    if (s < T - 1) {
      for (int i = threadIdx.x; i < LS_BC * G4; i += 256) {
        const int bb = i / G4, k = i - bb * G4;
        dgn[i] = bb < nb ? dG[(((long)(b0 + bb) * T + tn) * ndir + dir) * G4 + k] : 0.f;
      }
      __syncthreads();
      float acc[LS_BC];
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) acc[bb] = 0.f;
      for (int k = p * 4; k < G4; k += 128) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
        for (int bb = 0; bb < LS_BC; ++bb) {
          const float4 gv = *reinterpret_cast<const float4*>(dgn + bb * G4 + k);
          acc[bb] += wv.x * gv.x + wv.y * gv.y + wv.z * gv.z + wv.w * gv.w;
        }
      }
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) {
        float v = acc[bb];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        if (p == 0) dhr[jj * LS_BC + bb] = v;
      }
      __syncthreads();
    }
    if (threadIdx.x < LS_JB * LS_BC) {
      const int bb = threadIdx.x / LS_JB, j = threadIdx.x % LS_JB;
      if (bb < nb) {
        const int b = b0 + bb, hj = j0 + j;
        const long gt = ((long)b * T + t) * ndir + dir;
        const float* g = gates + gt * G4 + hj;
        const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
        const float cn = cseq[gt * H + hj];
        const float cp = s > 0 ? cseq[(((long)b * T + tpv) * ndir + dir) * H + hj] : 0.f;
        const float th = tanhf(cn);
        float dh = dout[((long)b * T + t) * OW + dir * H + hj];
        if (s < T - 1) dh += dhr[j * LS_BC + bb];
        float* dcp = dc + ((long)dir * B + b) * H + hj;
        const float dcn = (s < T - 1 ? *dcp : 0.f) + dh * go * (1.f - th * th);
        float* o = dG + gt * G4 + hj;
        o[0] = dcn * gg * gi * (1.f - gi);
        o[H] = dcn * cp * gf * (1.f - gf);
        o[2 * H] = dcn * gi * (1.f - gg * gg);
        o[3 * H] = dh * th * go * (1.f - go);
        *dcp = dcn * gf;
      }
    }
  }
}

// ---- GRU (model_type='gru' of TemporalEncoder, train_vit.py:207-211) ----------------------------------------
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr);  z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   hn = W_hn h + b_hn;  n = tanh(W_in x + b_in + r * hn);  h' = (1 - z) * n + z * h
// Same workgroup shape as the LSTM step: 32 dot-product rows per 8 hidden units = r and z over [x | h], the
// n gate's x part and its h part separately (r multiplies only the latter).  saved[b,t,dir,4H] = (r, z, n, hn).
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ x, LstmDir d0, LstmDir d1,
                                                           float* __restrict__ out, float* __restrict__ saved, int B,
                                                           int T, int In, int H, int s, int ndir) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = In + H;
  float* xh = sm;
  float* pre = sm + LS_BC * K;
  const int dir = blockIdx.y;
  const LstmDir w = dir ? d1 : d0;
  const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
  const int j0 = blockIdx.x * LS_JB;
  const int r = threadIdx.x >> 3, p = threadIdx.x & 7;
  const int q = r >> 3, jj = r & 7;
  const int grow = (q == 3 ? 2 : q) * H + j0 + jj;     // row of the [3H, .] matrices
  const float* wi = w.w_ih + (long)grow * In;
  const float* wh = w.w_hh + (long)grow * H;
  const int k_lo = q == 3 ? In : 0, k_hi = q == 2 ? In : K;
  const int OW = ndir * H;
  {  // one chunk of LS_BC samples per workgroup: blockIdx.z (samples are independent)
    const int b0 = blockIdx.z * LS_BC;
    const int nb = B - b0 < LS_BC ? B - b0 : LS_BC;
    __syncthreads();
    for (int i = threadIdx.x; i < LS_BC * K; i += 256) {
      const int bb = i / K, k = i - bb * K;
      float v = 0.f;
      if (bb < nb) {
        if (k < In) v = x[((long)(b0 + bb) * T + t) * In + k];
        else if (s > 0) v = out[((long)(b0 + bb) * T + tp) * OW + dir * H + (k - In)];
      }
      xh[i] = v;
    }
    __syncthreads();
    float acc[LS_BC];
#pragma unroll
    for (int bb = 0; bb < LS_BC; ++bb) acc[bb] = 0.f;
    for (int k = k_lo + p * 4; k < k_hi; k += 32) {
      const float4 wv = k < In ? *reinterpret_cast<const float4*>(wi + k) : *reinterpret_cast<const float4*>(wh + (k - In));
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) {
        const float4 xv = *reinterpret_cast<const float4*>(xh + bb * K + k);
        acc[bb] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
      }
    }
#pragma unroll
    for (int bb = 0; bb < LS_BC; ++bb) {
      float v = acc[bb];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      if (p == 0) pre[r * LS_BC + bb] = v + (q != 3 ? w.b_ih[grow] : 0.f) + (q != 2 ? w.b_hh[grow] : 0.f);
    }
    __syncthreads();
    if (threadIdx.x < LS_JB * LS_BC) {
      const int bb = threadIdx.x / LS_JB, j = threadIdx.x % LS_JB;
      if (bb < nb) {
        const int b = b0 + bb, hj = j0 + j;
        const float gr = ls_sigmoid(pre[(0 * 8 + j) * LS_BC + bb]), gz = ls_sigmoid(pre[(1 * 8 + j) * LS_BC + bb]);
        const float hn = pre[(3 * 8 + j) * LS_BC + bb];
        const float gn = tanhf(pre[(2 * 8 + j) * LS_BC + bb] + gr * hn);
        const float hp = xh[bb * K + In + hj];
        out[((long)b * T + t) * OW + dir * H + hj] = (1.f - gz) * gn + gz * hp;
        if (saved) {
          float* g = saved + (((long)b * T + t) * ndir + dir) * 4 * H + hj;
          g[0] = gr; g[H] = gz; g[2 * H] = gn; g[3 * H] = hn;
        }
      }
    }
  }
}

// Backward GRU step (s = T-1 .. 0).  dh = dout[b,t] + dhd (direct path dh_next * z_next, carried in `dhd`
// [ndir,B,H]) + dGh[b,t_next] . W_hh.  dG[b,t,dir,6H] = input-side gate gradients (dr, dz, dn) then hidden-side
// (dr, dz, dn * r); whh_t = W_hh^T [H,3H].
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dout, LstmDir d0, LstmDir d1,
                                                           const float* __restrict__ saved,
                                                           const float* __restrict__ out, float* __restrict__ dG,
                                                           float* __restrict__ dhd, int B, int T, int H, int s,
                                                           int ndir) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int G3 = 3 * H;
  float* dgn = sm;                      // [LS_BC][3H]: hidden-side dG of the step processed before
  float* dhr = sm + LS_BC * G3;         // [LS_JB][LS_BC]
  const int dir = blockIdx.y;
  const float* whh_t = (dir ? d1 : d0).w_hh;
  const int t = dir ? T - 1 - s : s;
  const int tn = dir ? t - 1 : t + 1, tpv = dir ? t + 1 : t - 1;
  const int j0 = blockIdx.x * LS_JB;
  const int jj = threadIdx.x >> 5, p = threadIdx.x & 31;
  const float* wr = whh_t + (long)(j0 + jj) * G3;
  const int OW = ndir * H;
  {  // one chunk of LS_BC samples per workgroup: blockIdx.z (samples are independent)
    const int b0 = blockIdx.z * LS_BC;
    const int nb = B - b0 < LS_BC ? B - b0 : LS_BC;
    __syncthreads();
    if (s < T - 1) {
      for (int i = threadIdx.x; i < LS_BC * G3; i += 256) {
        const int bb = i / G3, k = i - bb * G3;
        dgn[i] = bb < nb ? dG[(((long)(b0 + bb) * T + tn) * ndir + dir) * 2 * G3 + G3 + k] : 0.f;
      }
      __syncthreads();
      float acc[LS_BC];
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) acc[bb] = 0.f;
      for (int k = p * 4; k < G3; k += 128) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
        for (int bb = 0; bb < LS_BC; ++bb) {
          const float4 gv = *reinterpret_cast<const float4*>(dgn + bb * G3 + k);
          acc[bb] += wv.x * gv.x + wv.y * gv.y + wv.z * gv.z + wv.w * gv.w;
        }
      }
#pragma unroll
      for (int bb = 0; bb < LS_BC; ++bb) {
        float v = acc[bb];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        if (p == 0) dhr[jj * LS_BC + bb] = v;
      }
      __syncthreads();
    }
    if (threadIdx.x < LS_JB * LS_BC) {
      const int bb = threadIdx.x / LS_JB, j = threadIdx.x % LS_JB;
      if (bb < nb) {
        const int b = b0 + bb, hj = j0 + j;
        const long gt = ((long)b * T + t) * ndir + dir;
        const float* g = saved + gt * 4 * H + hj;
        const float gr = g[0], gz = g[H], gn = g[2 * H], hn = g[3 * H];
        const float hp = s > 0 ? out[((long)b * T + tpv) * OW + dir * H + hj] : 0.f;
        float* carry = dhd + ((long)dir * B + b) * H + hj;
        float dh = dout[((long)b * T + t) * OW + dir * H + hj];
        if (s < T - 1) dh += *carry + dhr[j * LS_BC + bb];
        const float dn = dh * (1.f - gz) * (1.f - gn * gn);
        const float dz = dh * (hp - gn) * gz * (1.f - gz);
        const float dr = dn * hn * gr * (1.f - gr);
        float* o = dG + gt * 2 * G3 + hj;
        o[0] = dr; o[H] = dz; o[2 * H] = dn;
        o[G3] = dr; o[G3 + H] = dz; o[G3 + 2 * H] = dn * gr;
        *carry = dh * gz;
      }
    }
  }
}

}  // namespace

static int lstm_check(int B, int T, int In, int H, int s, int ndir) {
  RVIT_CHECK_ARG(B > 0 && B <= 65535 * 8 && T > 0 && s >= 0 && s < T && (ndir == 1 || ndir == 2), "lstm: bad B=%d T=%d step=%d ndir=%d", B, T, s, ndir);
  RVIT_CHECK_ARG(H > 0 && H % LS_JB == 0 && (H & 3) == 0 && In > 0 && (In & 3) == 0, "lstm: H must be a multiple of 8 and In of 4 (H=%d In=%d)", H, In);
  return RVIT_OK;
}

int rvit_lstm_step_fwd_impl(const float* x, const float* const* w, float* out, float* c, float* gates, float* cseq, int B,
                            int T, int In, int H, int s, int ndir, hipStream_t st) {
  if (int e = lstm_check(B, T, In, H, s, ndir)) return e;
  RVIT_CHECK_ARG(x && w && out && c && (!gates == !cseq), "lstm_step_fwd: null pointer (gates and cseq come together)");
  for (int i = 0; i < 4 * ndir; ++i) RVIT_CHECK_ARG(w[i] && rvit_aligned16(w[i]), "lstm_step_fwd: weight %d null or unaligned", i);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(out), "lstm_step_fwd: alignment");
  const LstmDir d0{w[0], w[1], w[2], w[3]};
  const LstmDir d1 = ndir == 2 ? LstmDir{w[4], w[5], w[6], w[7]} : d0;
  const size_t lds = (size_t)(LS_BC * (In + H) + 32 * LS_BC) * sizeof(float);
  RVIT_CHECK_ARG(lds <= 64 * 1024, "lstm_step_fwd: In + H = %d too wide for the LDS vector buffer", In + H);
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(H / LS_JB, ndir, (B + LS_BC - 1) / LS_BC), dim3(256), lds, st, x, d0, d1, out, c, gates, cseq, B,
                     T, In, H, s, ndir);
  RVIT_CHECK_LAUNCH("lstm_step_fwd");
  return RVIT_OK;
}

int rvit_lstm_step_bwd_impl(const float* dout, const float* whh_t0, const float* whh_t1, const float* gates,
                            const float* cseq, float* dG, float* dc, int B, int T, int H, int s, int ndir,
                            hipStream_t st) {
  if (int e = lstm_check(B, T, 4, H, s, ndir)) return e;
  RVIT_CHECK_ARG(dout && whh_t0 && (ndir == 1 || whh_t1) && gates && cseq && dG && dc, "lstm_step_bwd: null pointer");
  RVIT_CHECK_ARG(rvit_aligned16(whh_t0) && rvit_aligned16(whh_t1) && rvit_aligned16(dG), "lstm_step_bwd: alignment");
  const LstmDir d0{nullptr, whh_t0, nullptr, nullptr}, d1{nullptr, ndir == 2 ? whh_t1 : whh_t0, nullptr, nullptr};
  const size_t lds = (size_t)(LS_BC * 4 * H + LS_JB * LS_BC) * sizeof(float);
  RVIT_CHECK_ARG(lds <= 64 * 1024, "lstm_step_bwd: H = %d too wide for the LDS gate buffer", H);
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(H / LS_JB, ndir, (B + LS_BC - 1) / LS_BC), dim3(256), lds, st, dout, d0, d1, gates, cseq, dG, dc,
                     B, T, H, s, ndir);
  RVIT_CHECK_LAUNCH("lstm_step_bwd");
  return RVIT_OK;
}

int rvit_gru_step_fwd_impl(const float* x, const float* const* w, float* out, float* saved, int B, int T, int In, int H,
                           int s, int ndir, hipStream_t st) {
  if (int e = lstm_check(B, T, In, H, s, ndir)) return e;
  RVIT_CHECK_ARG(x && w && out, "gru_step_fwd: null pointer");
  for (int i = 0; i < 4 * ndir; ++i) RVIT_CHECK_ARG(w[i] && rvit_aligned16(w[i]), "gru_step_fwd: weight %d null or unaligned", i);
  RVIT_CHECK_ARG(rvit_aligned16(x) && rvit_aligned16(out), "gru_step_fwd: alignment");
  const LstmDir d0{w[0], w[1], w[2], w[3]};
  const LstmDir d1 = ndir == 2 ? LstmDir{w[4], w[5], w[6], w[7]} : d0;
  const size_t lds = (size_t)(LS_BC * (In + H) + 32 * LS_BC) * sizeof(float);
  RVIT_CHECK_ARG(lds <= 64 * 1024, "gru_step_fwd: In + H = %d too wide for the LDS vector buffer", In + H);
  hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / LS_JB, ndir, (B + LS_BC - 1) / LS_BC), dim3(256), lds, st, x, d0, d1, out, saved, B, T, In, H,
                     s, ndir);
  RVIT_CHECK_LAUNCH("gru_step_fwd");
  return RVIT_OK;
}

int rvit_gru_step_bwd_impl(const float* dout, const float* whh_t0, const float* whh_t1, const float* saved,
                           const float* out, float* dG, float* dhd, int B, int T, int H, int s, int ndir,
                           hipStream_t st) {
  if (int e = lstm_check(B, T, 4, H, s, ndir)) return e;
  RVIT_CHECK_ARG(dout && whh_t0 && (ndir == 1 || whh_t1) && saved && out && dG && dhd, "gru_step_bwd: null pointer");
  RVIT_CHECK_ARG(rvit_aligned16(whh_t0) && rvit_aligned16(whh_t1) && rvit_aligned16(dG), "gru_step_bwd: alignment");
  const LstmDir d0{nullptr, whh_t0, nullptr, nullptr}, d1{nullptr, ndir == 2 ? whh_t1 : whh_t0, nullptr, nullptr};
  const size_t lds = (size_t)(LS_BC * 3 * H + LS_JB * LS_BC) * sizeof(float);
  RVIT_CHECK_ARG(lds <= 64 * 1024, "gru_step_bwd: H = %d too wide for the LDS gate buffer", H);
  hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(H / LS_JB, ndir, (B + LS_BC - 1) / LS_BC), dim3(256), lds, st, dout, d0, d1, saved, out, dG, dhd,
                     B, T, H, s, ndir);
  RVIT_CHECK_LAUNCH("gru_step_bwd");
  return RVIT_OK;
}
