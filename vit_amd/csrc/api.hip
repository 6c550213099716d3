// extern "C" surface of librvit (declared in include/rvit.h).  Argument validation lives in the
// *_impl / *_dispatch functions next to each kernel; this file only adapts ABI types.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/rvit.h"
#include "common.h"
#include "rvit_internal.h"

static thread_local char g_err[512] = "";

void rvit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline bool aligned_to(const void* p, uintptr_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

static void conv_geometry(GemmArgs& a, const void* x, const void* w, int B, int H, int W, int Cpix, int seg_len, int KH,
                          int KW, int stride, int pad, int Ho, int Wo, int Cout) {
  a.A = (const bf16*)x; a.lda = 0;
  a.B = (const bf16*)w; a.ldb = (long)KH * KW * seg_len;
  a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * seg_len;
  a.mode = 1;
  a.H = H; a.W = W; a.Cpix = Cpix; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
  a.seg_len = seg_len;
}

extern "C" {

int rvit_version(void) { return RVIT_ABI_VERSION; }
const char* rvit_last_error(void) { return g_err; }

int rvit_gemm_bf16(const void* A, int64_t lda, int a_row_group, int a_row_extra, const void* B, int64_t ldb, void* C,
                   int64_t ldc, int c_dtype, int64_t M, int N, int K, const float* scale, const float* bias,
                   const void* gate, int64_t ldg, int gate_mode, const void* R, int64_t ldr, int r_dtype,
                   int r_period, int act, void* preact, int64_t ldp, int row_group, int row_extra, int drop_mode,
                   float drop_p, uint64_t drop_seed, int tile, void* stream) {
  GemmArgs a{};
  a.A = (const bf16*)A; a.lda = lda; a.a_row_group = a_row_group; a.a_row_extra = a_row_extra;
  a.B = (const bf16*)B; a.ldb = ldb;
  a.M = M; a.N = N; a.K = K;
  a.mode = 0;
  a.C = C; a.ldc = ldc; a.c_f32 = c_dtype == RVIT_F32;
  a.C2 = (bf16*)preact; a.ldc2 = ldp;
  a.scale = scale; a.bias = bias;
  a.G = (const bf16*)gate; a.ldg = ldg; a.gate = gate ? gate_mode : 0;
  a.R = R; a.ldr = ldr; a.r_f32 = r_dtype == RVIT_F32; a.r_period = r_period;
  a.act = act; a.row_group = row_group; a.row_extra = row_extra; a.tile = tile;
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "gemm: drop_p=%f outside [0,1)", (double)drop_p);
  a.drop_mode = drop_p > 0.f ? drop_mode : 0;
  a.drop = make_drop_key(drop_p, drop_seed);
  RVIT_CHECK_ARG(c_dtype == RVIT_BF16 || c_dtype == RVIT_F32, "gemm: bad c_dtype %d", c_dtype);
  RVIT_CHECK_ARG(act >= 0 && act <= RVIT_ACT_HSIGMOID, "gemm: bad act %d", act);
  RVIT_CHECK_ARG(!gate || gate_mode == RVIT_GATE_GELU_GRAD || gate_mode == RVIT_GATE_RELU_MASK, "gemm: bad gate mode");
  RVIT_CHECK_ARG(ldc >= N && (!R || ldr >= N) && (!preact || ldp >= N), "gemm: output pitch smaller than N");
  RVIT_CHECK_ARG(row_group >= 0 && row_extra >= 0 && r_period >= 0 && a_row_group >= 0 && a_row_extra >= 0,
                 "gemm: negative row remap");
  // vector epilogue: 8 consecutive columns per lane, 16-byte accesses
  bool vec = (N & 7) == 0 && (ldc & (a.c_f32 ? 3 : 7)) == 0 && aligned_to(C, 16);
  if (scale) vec = vec && aligned_to(scale, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  if (gate) vec = vec && (ldg & 7) == 0 && aligned_to(gate, 16);
  if (R) vec = vec && (ldr & (a.r_f32 ? 3 : 7)) == 0 && aligned_to(R, 16);
  if (preact) vec = vec && (ldp & 7) == 0 && aligned_to(preact, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_gemm_mxfp8(const void* A, int64_t lda, const void* SA, const void* B, int64_t ldb, const void* SB, void* C,
                    int64_t ldc, int c_dtype, int64_t M, int N, int K, const float* bias, const void* R, int64_t ldr,
                    int r_dtype, int act, void* Cq, int64_t ldcq, void* SCq, void* stream) {
  MxGemmArgs a{};
  a.A = (const uint8_t*)A; a.lda = lda; a.SA = (const uint8_t*)SA;
  a.B = (const uint8_t*)B; a.ldb = ldb; a.SB = (const uint8_t*)SB;
  a.M = M; a.N = N; a.K = K;
  a.C = C; a.ldc = ldc; a.c_f32 = c_dtype == RVIT_F32;
  a.bias = bias; a.R = R; a.ldr = ldr; a.r_f32 = r_dtype == RVIT_F32; a.act = act;
  a.Cq = (uint8_t*)Cq; a.ldcq = ldcq; a.SCq = (uint8_t*)SCq;
  RVIT_CHECK_ARG(!C || c_dtype == RVIT_BF16 || c_dtype == RVIT_F32, "gemm_mxfp8: bad c_dtype %d", c_dtype);
  return rvit_gemm_mxfp8_dispatch(a, (hipStream_t)stream);
}

int rvit_quantize_mxfp8(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t rows, int K,
                        void* stream) {
  RVIT_CHECK_ARG(x_dtype == RVIT_BF16 || x_dtype == RVIT_F32, "quantize_mxfp8: bad dtype %d", x_dtype);
  return rvit_quantize_mx_impl(x, x_dtype == RVIT_F32, ldx, (uint8_t*)q, ldq, (uint8_t*)scales, rows, K,
                               (hipStream_t)stream);
}

int rvit_conv2d_fwd_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cpix, int seg_len, int KH,
                         int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* scale,
                         const float* bias, const void* residual, int act, int tile, void* stream) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                 "conv: bad geometry");
  RVIT_CHECK_ARG((long)B * H * W < 0x7fffffffL, "conv: B*H*W exceeds 2^31 pixels");
  RVIT_CHECK_ARG((Ho - 1) * stride - pad + KH - 1 < H + pad + stride && (Wo - 1) * stride - pad < W,
                 "conv: output size inconsistent with input");
  GemmArgs a{};
  conv_geometry(a, x, w, B, H, W, Cpix, seg_len, KH, KW, stride, pad, Ho, Wo, Cout);
  a.C = y; a.ldc = Cout; a.c_f32 = 0;
  a.scale = scale; a.bias = bias;
  a.R = residual; a.ldr = Cout; a.r_f32 = 0; a.r_period = 0;
  a.act = act; a.tile = tile;
  RVIT_CHECK_ARG(act >= 0 && act <= RVIT_ACT_HSIGMOID, "conv: bad act %d", act);
  bool vec = (Cout & 7) == 0 && aligned_to(y, 16);
  if (scale) vec = vec && aligned_to(scale, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  if (residual) vec = vec && aligned_to(residual, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_stem_conv1_stats_f32_fwd(const float* x, const void* w, void* y, void* stat_ws, int64_t stat_ws_bytes,
                                  int* stat_slots, int B, void* stream) {
  return rvit_stem_conv1_rows_impl(x, w, y, (float*)stat_ws, stat_ws_bytes, stat_slots, B, (hipStream_t)stream);
}

int rvit_stem_conv1_wgrad_f32(const float* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes, int B,
                              void* stream) {
  return rvit_stem_conv1_wgrad_impl(x, dy, dw, (float*)workspace, workspace_bytes, B, (hipStream_t)stream);
}

int rvit_conv2d_fwd_stats_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cpix, int seg_len, int KH,
                               int KW, int stride, int pad, int Ho, int Wo, int Cout, void* stat_ws, int64_t stat_ws_bytes,
                               int* stat_slots, int tile, void* stream) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                 "conv: bad geometry");
  RVIT_CHECK_ARG((long)B * H * W < 0x7fffffffL, "conv: B*H*W exceeds 2^31 pixels");
  RVIT_CHECK_ARG((Ho - 1) * stride - pad + KH - 1 < H + pad + stride && (Wo - 1) * stride - pad < W,
                 "conv: output size inconsistent with input");
  RVIT_CHECK_ARG(stat_ws && stat_slots && aligned_to(stat_ws, 16) && stat_ws_bytes >= 0, "conv (stats): workspace / slot count missing");
  GemmArgs a{};
  conv_geometry(a, x, w, B, H, W, Cpix, seg_len, KH, KW, stride, pad, Ho, Wo, Cout);
  a.C = y; a.ldc = Cout; a.c_f32 = 0;
  a.ldr = Cout;
  a.tile = tile;
  a.vec = (Cout & 7) == 0 && aligned_to(y, 16);
  a.stat_part = (float*)stat_ws;
  a.stat_cap_slots = (long)(stat_ws_bytes / 4 / (2L * Cout));
  a.stat_slots = stat_slots;
  *stat_slots = 0;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_bn_stats_finalize(const void* stat_ws, int stat_slots, int C, float* sums, void* stage, void* stream) {
  return rvit_bn_stats_finalize_impl((const float*)stat_ws, stat_slots, C, sums, (float*)stage, (hipStream_t)stream);
}

int rvit_conv3x3_kernel(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (long)B * H * W >= 0x7fffffffL) return RVIT_CONV_GATHER;
  GemmArgs a{};
  conv_geometry(a, nullptr, nullptr, B, H, W, Cin, Cin, 3, 3, 1, 1, H, W, Cout);
  a.ldc = Cout; a.ldr = Cout; a.vec = (Cout & 7) == 0;
  return rvit_conv3x3_auto(a);
}

int rvit_conv2d_dgrad_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int KH, int KW,
                           int stride, int pad, int Hx, int Wx, int Cin, const void* residual, int tile,
                           void* stream) {
  RVIT_CHECK_ARG(B > 0 && Hy > 0 && Wy > 0 && Hx > 0 && Wx > 0 && KH > 0 && KW > 0 && pad >= 0, "conv dgrad: bad geometry");
  RVIT_CHECK_ARG((long)B * Hy * Wy < 0x7fffffffL && (long)B * Hx * Wx < 0x7fffffffL, "conv dgrad: too many pixels");
  RVIT_CHECK_ARG((Cout & 63) == 0, "conv dgrad: Cout=%d must be a multiple of 64", Cout);
  GemmArgs a{};
  // rows = pixels of dX, K = (kh, kw, cout) runs of dY pixels, N = Cin
  conv_geometry(a, dy, w_t, B, Hy, Wy, Cout, Cout, KH, KW, stride, pad, Hx, Wx, Cin);
  a.transposed = 1;
  a.C = dx; a.ldc = Cin; a.c_f32 = 0;
  a.R = residual; a.ldr = Cin; a.r_f32 = 0;
  a.tile = tile;
  a.vec = (Cin & 7) == 0 && aligned_to(dx, 16) && (!residual || aligned_to(residual, 16));
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_conv2d_dgrad_bnr_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int KH, int KW,
                               int stride, int pad, int Hx, int Wx, int Cin, const void* residual, const void* bn_y,
                               const void* bn_relu_mask, const float* bn_fwd_sums, float eps, void* stat_ws,
                               int64_t stat_ws_bytes, int* stat_slots, void* stream) {
  RVIT_CHECK_ARG(B > 0 && Hy > 0 && Wy > 0 && Hx > 0 && Wx > 0 && KH > 0 && KW > 0 && pad >= 0, "conv dgrad: bad geometry");
  RVIT_CHECK_ARG((long)B * Hy * Wy < 0x7fffffffL && (long)B * Hx * Wx < 0x7fffffffL, "conv dgrad: too many pixels");
  RVIT_CHECK_ARG((Cout & 63) == 0, "conv dgrad: Cout=%d must be a multiple of 64", Cout);
  RVIT_CHECK_ARG(bn_y && bn_fwd_sums && stat_ws && stat_slots && aligned_to(bn_y, 16) && aligned_to(stat_ws, 16) &&
                     stat_ws_bytes >= 0 && (Cin & 7) == 0,
                 "conv dgrad (bn sums): norm input / forward sums / workspace / slot count missing or misaligned");
  GemmArgs a{};
  conv_geometry(a, dy, w_t, B, Hy, Wy, Cout, Cout, KH, KW, stride, pad, Hx, Wx, Cin);
  a.transposed = 1;
  a.C = dx; a.ldc = Cin; a.c_f32 = 0;
  a.R = residual; a.ldr = Cin; a.r_f32 = 0;
  a.tile = 0;
  a.vec = (Cin & 7) == 0 && aligned_to(dx, 16) && (!residual || aligned_to(residual, 16));
  a.stat_part = (float*)stat_ws;
  a.stat_cap_slots = (long)(stat_ws_bytes / 4 / (2L * Cin));
  a.stat_slots = stat_slots;
  a.bnr_y = (const bf16*)bn_y;
  a.bnr_mask = (const uint8_t*)bn_relu_mask;
  a.bnr_fsums = bn_fwd_sums;
  a.bnr_inv_m = 1.0f / (float)((long)B * Hx * Wx);
  a.bnr_eps = eps;
  *stat_slots = 0;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_conv2d_dgrad_s2_bf16(const void* dy, const void* w_s2, void* dx, int B, int Hy, int Wy, int Cout, int Cin,
                              const void* residual, void* stream) {
  // Input gradient of a 3x3 / stride 2 / pad 1 conv by output parity.  dX pixel (2i+ph, 2j+pw) only receives the taps
  // with kh = 1 (ph = 0) or kh in {2, 0} (ph = 1), likewise kw: four dense stride-1 convs over dY with 1x1, 1x2,
  // 2x1 and 2x2 taps (9 taps in all instead of 4 x 9 with three quarters of them reading zeros).  Tap (kh', kw')
  // of class (ph, pw) reads dY[i + kh', j + kw'] (out of range = 0, the forward gather's bounds check); each class
  // writes its own lattice of dX: row pitch 2*Cin with the row remap (m / Wy) * 2*Wy + m % Wy + Wy.
  RVIT_CHECK_ARG(dy && w_s2 && dx && B > 0 && Hy > 0 && Wy > 0, "conv dgrad s2: bad geometry");
  RVIT_CHECK_ARG((Cout & 63) == 0 && (Cin & 7) == 0, "conv dgrad s2: Cout=%d must be a multiple of 64, Cin=%d of 8", Cout, Cin);
  RVIT_CHECK_ARG((long)B * Hy * Wy * 4 < 0x7fffffffL, "conv dgrad s2: too many pixels");
  RVIT_CHECK_ARG(aligned_to(dx, 16) && (!residual || aligned_to(residual, 16)), "conv dgrad s2: alignment");
  static const int k_off[4] = {0, 1, 3, 5};   // first K block (in taps) of class 2*ph + pw in w_s2
  const int Wx = 2 * Wy;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    GemmArgs a{};
    conv_geometry(a, dy, (const bf16*)w_s2 + (long)k_off[cls] * Cout, B, Hy, Wy, Cout, Cout, 1 + ph, 1 + pw, 1, 0, Hy, Wy,
                  Cin);
    a.ldb = 9L * Cout;
    const long base = ((long)ph * Wx + pw - Wx) * Cin;   // + remapped row * 2*Cin = pixel (2i+ph, 2j+pw)
    a.C = (bf16*)dx + base; a.ldc = 2L * Cin; a.c_f32 = 0;
    a.R = residual ? (const bf16*)residual + base : nullptr; a.ldr = 2L * Cin; a.r_f32 = 0;
    a.row_group = Wy; a.row_extra = Wy;
    a.tile = 0;
    a.vec = 1;
    const int rc = rvit_gemm_dispatch(a, (hipStream_t)stream);
    if (rc != RVIT_OK) return rc;
  }
  return RVIT_OK;
}

int rvit_conv1x1_dgrad_s2_acc_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int Cin,
                                   void* stream) {
  // Input gradient of a 1x1 / stride 2 conv (the resnet downsample branch), ADDED to dx: only the (even, even) pixels
  // of dx receive anything, dx[b, 2i, 2j, :] += dy[b, i, j, :] . w_t^T - a plain GEMM whose output rows are that lattice.
  RVIT_CHECK_ARG(dy && w_t && dx && B > 0 && Hy > 0 && Wy > 0, "conv1x1 dgrad s2: bad geometry");
  RVIT_CHECK_ARG((Cout & 7) == 0 && (Cin & 7) == 0 && aligned_to(dx, 16), "conv1x1 dgrad s2: channels must be multiples of 8");
  GemmArgs a{};
  a.A = (const bf16*)dy; a.lda = Cout;
  a.B = (const bf16*)w_t; a.ldb = Cout;
  a.M = (long)B * Hy * Wy; a.N = Cin; a.K = Cout;
  const long base = -2L * Wy * Cin;
  a.C = (bf16*)dx + base; a.ldc = 2L * Cin; a.c_f32 = 0;
  a.R = (const bf16*)dx + base; a.ldr = 2L * Cin; a.r_f32 = 0;
  a.row_group = Wy; a.row_extra = Wy;
  a.vec = 1;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_conv2d_tokens_fwd(const void* x, const void* w, float* tokens, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* bias,
                           const float* pos_embed, float drop_p, uint64_t drop_seed, int tile, void* stream) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                 "conv_tokens: bad geometry");
  RVIT_CHECK_ARG((long)B * H * W < 0x7fffffffL, "conv_tokens: B*H*W exceeds 2^31 pixels");
  RVIT_CHECK_ARG(tokens && pos_embed, "conv_tokens: null pointer");
  GemmArgs a{};
  conv_geometry(a, x, w, B, H, W, Cpix, seg_len, KH, KW, stride, pad, Ho, Wo, Cout);
  a.C = tokens; a.ldc = Cout; a.c_f32 = 1;
  a.bias = bias;
  a.R = pos_embed + Cout; a.ldr = Cout; a.r_f32 = 1; a.r_period = Ho * Wo;
  a.row_group = Ho * Wo; a.row_extra = 1;
  a.act = RVIT_ACT_NONE; a.tile = tile;
  RVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "conv_tokens: drop_p=%f outside [0,1)", (double)drop_p);
  a.drop_mode = drop_p > 0.f ? RVIT_DROP_POST_ACT : 0;  // pos_drop: after the position add
  a.drop = make_drop_key(drop_p, drop_seed);
  bool vec = (Cout & 7) == 0 && aligned_to(tokens, 16) && aligned_to(pos_embed, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_wgrad_bf16(const void* dY, int64_t ldy, int y_row_group, int y_row_extra, const void* X, int64_t ldx,
                    int x_row_group, int x_row_extra, float* dW, int64_t ldw, int64_t M, int N, int N_store, int K,
                    int splits, void* workspace, int64_t workspace_bytes, void* stream) {
  WgradArgs a{};
  a.ws = (float*)workspace; a.ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  a.dY = (const bf16*)dY; a.ldy = ldy; a.y_row_group = y_row_group; a.y_row_extra = y_row_extra;
  a.X = (const bf16*)X; a.ldx = ldx; a.x_row_group = x_row_group; a.x_row_extra = x_row_extra;
  a.dW = dW; a.ldw = ldw; a.M = M; a.N = N; a.N_store = N_store; a.K = K; a.mode = 0; a.splits = splits;
  return rvit_wgrad_dispatch(a, (hipStream_t)stream);
}

int rvit_linear_wgrad_bias_bf16(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw,
                                float* dbias, int64_t M, int N, int N_store, int K, int splits, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  RVIT_CHECK_ARG(dbias, "linear_wgrad_bias: null dbias");
  WgradArgs a{};
  a.ws = (float*)workspace; a.ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  a.dY = (const bf16*)dY; a.ldy = ldy;
  a.X = (const bf16*)X; a.ldx = ldx;
  a.dW = dW; a.ldw = ldw; a.M = M; a.N = N; a.N_store = N_store; a.K = K; a.mode = 0; a.splits = splits;
  if (N > 64) {
    a.dbias = dbias;
    return rvit_wgrad_dispatch(a, (hipStream_t)stream);
  }
  // narrow outputs (the classifier head): the four-wave narrow kernel has no spare wave for the sums
  const int rc = rvit_wgrad_dispatch(a, (hipStream_t)stream);
  if (rc != RVIT_OK) return rc;
  return rvit_colsum_impl(dY, ldy, dbias, M, N_store, 0, 0, a.ws, a.ws_bytes, (hipStream_t)stream);
}

int rvit_conv2d_wgrad_bf16(const void* dy, const void* x, float* dW, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, int splits,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                 "conv wgrad: bad geometry");
  RVIT_CHECK_ARG((long)B * H * W < 0x7fffffffL && (long)B * Ho * Wo < 0x7fffffffL, "conv wgrad: too many pixels");
  WgradArgs a{};
  a.dY = (const bf16*)dy; a.ldy = Cout;
  a.X = (const bf16*)x; a.ldx = 0;
  a.dW = dW; a.ldw = (long)KH * KW * seg_len;
  a.M = (long)B * Ho * Wo; a.N = Cout; a.N_store = Cout; a.K = KH * KW * seg_len;
  a.mode = 1;
  a.H = H; a.W = W; a.Cpix = Cpix; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
  a.seg_len = seg_len; a.splits = splits;
  a.ws = (float*)workspace; a.ws_bytes = workspace ? (size_t)workspace_bytes : 0;
  return rvit_wgrad_dispatch(a, (hipStream_t)stream);
}

int rvit_stem_conv1_pool_fwd(const void* img, const void* w, const float* scale, const float* bias, void* pooled, int B,
                             int relu, void* stream) {
  return rvit_stem_conv1_pool_impl(img, w, scale, bias, pooled, B, relu, 0, (hipStream_t)stream);
}

int rvit_stem_conv1_pool_f32_fwd(const float* x_nchw, const void* w, const float* scale, const float* bias, void* pooled,
                                 int B, int relu, void* stream) {
  return rvit_stem_conv1_pool_impl(x_nchw, w, scale, bias, pooled, B, relu, 1, (hipStream_t)stream);
}

int rvit_input_pack_nhwc4(const float* x_nchw, void* out, int B, int C, int H, int W, int Hp, int Wp, int pad_top,
                          int pad_left, void* stream) {
  return rvit_input_pack_impl(x_nchw, out, B, C, H, W, Hp, Wp, pad_top, pad_left, (hipStream_t)stream);
}

int rvit_maxpool3x3s2_nhwc(const void* x, void* y, void* argmax, int B, int H, int W, int C, void* stream) {
  return rvit_maxpool_impl(x, y, argmax, B, H, W, C, (hipStream_t)stream);
}

int rvit_embed_cls(const float* cls_token, const float* pos_embed, float* x, int B, int S, int E, float drop_p,
                   uint64_t drop_seed, void* stream) {
  return rvit_embed_cls_impl(cls_token, pos_embed, x, B, S, E, drop_p, drop_seed, (hipStream_t)stream);
}

int rvit_dropout_apply(const void* in, void* out, int dtype, int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                       float p, uint64_t seed, void* stream) {
  RVIT_CHECK_ARG(dtype == RVIT_BF16 || dtype == RVIT_F32, "dropout_apply: bad dtype %d", dtype);
  return rvit_dropout_apply_impl(in, out, dtype == RVIT_F32, rows, cols, ld_in, ld_out, p, seed, (hipStream_t)stream);
}

int rvit_dropout_mask(uint8_t* out, int64_t n, float p, uint64_t seed, void* stream) {
  return rvit_dropout_mask_impl(out, n, p, seed, (hipStream_t)stream);
}

int rvit_layernorm_mxfp8_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                             uint8_t* q, int64_t ldq, uint8_t* scales, int64_t rows, int E, float eps, void* stream) {
  RVIT_CHECK_ARG(x_dtype == RVIT_F32 || x_dtype == RVIT_BF16, "layernorm_mx: x must be f32 or bf16");
  return rvit_layernorm_mx_fwd_impl(x, x_dtype == RVIT_F32, ldx, gamma, beta, q, ldq, scales, rows, E, eps,
                                    (hipStream_t)stream);
}

int rvit_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y,
                       int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows, int E, float eps,
                       void* stream) {
  return rvit_layernorm_fwd_impl(x, x_dtype == RVIT_F32, ldx, gamma, beta, y, y_dtype == RVIT_F32, ldy, mean, rstd,
                                 rows, E, eps, (hipStream_t)stream);
}

int rvit_encoder_mlp_fwd(float* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                         const void* w2, const float* b2, int64_t T, int E, int hidden, int rows_per_wg,
                         void* stream) {
  return rvit_encoder_mlp_impl(x, gamma, beta, eps, w1, b1, w2, b2, T, E, hidden, rows_per_wg, (hipStream_t)stream);
}

int rvit_encoder_attn_fwd(float* x, const float* gamma, const float* beta, float eps, const void* wqkv,
                          const float* bqkv, const void* wout, const float* bout, int B, int S, int E, int H,
                          void* stream) {
  return rvit_encoder_attn_impl(x, gamma, beta, eps, wqkv, bqkv, wout, bout, B, S, E, H, (hipStream_t)stream);
}

int rvit_encoder_stack_fwd(float* x, const void* const* layer_params, int n_layers, float eps, int B, int S, int E, int H,
                           int hidden, void* stream) {
  return rvit_encoder_stack_impl(x, layer_params, n_layers, eps, B, S, E, H, hidden, (hipStream_t)stream);
}

int rvit_attn_fwd_fp8(const void* qkv, uint8_t* out_q, int64_t ldo, uint8_t* out_s, int B, int S, int H, int64_t ldq,
                      float scale, void* stream) {
  return rvit_attn_fwd_fp8_impl(qkv, out_q, ldo, out_s, B, S, H, ldq, scale, (hipStream_t)stream);
}

int rvit_attn_fwd_mx(const void* qkv, uint8_t* out_q, int64_t ldo, uint8_t* out_s, int B, int S, int H, int64_t ldq,
                     float scale, void* stream) {
  return rvit_attn_fwd_mx_impl(qkv, out_q, ldo, out_s, B, S, H, ldq, scale, (hipStream_t)stream);
}

int rvit_attn_fwd_s64(const void* qkv, void* out, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                      void* stream) {
  return rvit_attn16_impl(qkv, out, B, S, H, ldq, ldo, scale, (hipStream_t)stream);
}

int rvit_dwconv_fwd_bf16(const void* x, const void* w, const float* scale, const float* bias, void* out, int B, int H,
                         int W, int C, int K, int stride, int pad, int act, void* stream) {
  return rvit_dwconv_impl(x, w, scale, bias, out, B, H, W, C, K, stride, pad, act, (hipStream_t)stream);
}
int rvit_avgpool_nhwc(const void* x, void* out, int B, int HW, int C, void* stream) {
  return rvit_avgpool_impl(x, out, B, HW, C, (hipStream_t)stream);
}
int rvit_channel_scale_nhwc(void* x, const void* gate, int B, int HW, int C, void* stream) {
  return rvit_channel_scale_impl(x, gate, B, HW, C, (hipStream_t)stream);
}
int rvit_squeeze_excite_bf16(void* x, const void* w1, const float* b1, const void* w2, const float* b2, int B, int HW, int C,
                             int RD, void* stream) {
  return rvit_squeeze_excite_impl(x, w1, b1, w2, b2, B, HW, C, RD, (hipStream_t)stream);
}
int rvit_bn_generic_stats(const void* y, int64_t M, int C, float* sums, void* workspace, int64_t workspace_bytes, void* stream) {
  return rvit_mb_bn_stats_impl(y, (long)M, C, sums, workspace, (size_t)(workspace_bytes > 0 ? workspace_bytes : 0), (hipStream_t)stream);
}
int rvit_bn_generic_apply(const void* y, const float* sums, const float* gamma, const float* beta, const void* residual,
                          void* z, int64_t M, int C, float eps, int act, float momentum, float* running_mean,
                          float* running_var, void* stream) {
  return rvit_mb_bn_apply_impl(y, sums, gamma, beta, residual, z, (long)M, C, eps, act, momentum, running_mean, running_var,
                               (hipStream_t)stream);
}
int rvit_bn_generic_bwd(const void* dz, const void* y, const float* fsums, float* bsums, const float* gamma, const float* beta,
                        void* dy, float* dgamma, float* dbeta, int64_t M, int C, float eps, int act, void* workspace,
                        int64_t workspace_bytes, void* stream) {
  return rvit_mb_bn_bwd_impl(dz, y, fsums, bsums, gamma, beta, dy, dgamma, dbeta, (long)M, C, eps, act, workspace,
                             (size_t)(workspace_bytes > 0 ? workspace_bytes : 0), (hipStream_t)stream);
}
int rvit_dwconv_dgrad_bf16(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                           void* stream) {
  return rvit_dwconv_dgrad_impl(dy, w, dx, B, H, W, C, K, stride, pad, (hipStream_t)stream);
}
int rvit_dwconv_wgrad_bf16(const void* dy, const void* x, float* dw, int B, int H, int W, int C, int K, int stride, int pad,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  return rvit_dwconv_wgrad_impl(dy, x, dw, B, H, W, C, K, stride, pad, workspace,
                                (size_t)(workspace_bytes > 0 ? workspace_bytes : 0), (hipStream_t)stream);
}
int rvit_channel_affine_nhwc(const void* x, const void* gate, const void* add, void* out, int B, int HW, int C, void* stream) {
  return rvit_channel_affine_impl(x, gate, add, out, B, HW, C, (hipStream_t)stream);
}
int rvit_channel_dot_nhwc(const void* a, const void* x, const void* gate, void* out, int B, int HW, int C, void* stream) {
  return rvit_channel_dot_impl(a, x, gate, out, B, HW, C, (hipStream_t)stream);
}
int rvit_pointwise_conv_bf16(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                             void* out, int64_t M, int K, int N, int act, void* stream) {
  return rvit_pointwise_impl(x, w, scale, bias, residual, out, (long)M, K, N, act, (hipStream_t)stream);
}
int rvit_pointwise_conv_supported(int K, int N) { return rvit_pointwise_lds_bytes(K, N) > 0 ? 1 : 0; }
int rvit_conv3x3s2_rgb_f32_fwd(const float* x_nchw, const float* w, const float* scale, const float* bias, void* out, int B,
                               int H, int W, int Cout, int act, void* stream) {
  return rvit_conv3x3s2_rgb_impl(x_nchw, w, scale, bias, out, B, H, W, Cout, act, (hipStream_t)stream);
}

int rvit_layernorm_bwd(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                       const float* rstd, const float* gamma, float* dx, int64_t lddx, int accumulate,
                       void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta, int64_t rows, int E, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  return rvit_layernorm_bwd_impl(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, accumulate, dx_bf16, lddb, dgamma,
                                 dbeta, rows, E, (float*)workspace, workspace ? (size_t)workspace_bytes : 0, 0.f, 0,
                                 (hipStream_t)stream);
}
int rvit_layernorm_bwd_drop(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                            const float* rstd, const float* gamma, float* dx, int64_t lddx, int accumulate,
                            void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta, int64_t rows, int E,
                            void* workspace, int64_t workspace_bytes, float drop_p, uint64_t drop_seed, void* stream) {
  return rvit_layernorm_bwd_impl(dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, accumulate, dx_bf16, lddb, dgamma,
                                 dbeta, rows, E, (float*)workspace, workspace ? (size_t)workspace_bytes : 0, drop_p,
                                 drop_seed, (hipStream_t)stream);
}

int rvit_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                  float drop_p, uint64_t drop_seed, void* stream) {
  return rvit_attn_fwd_impl(qkv, out, lse, B, S, H, ldq, ldo, scale, drop_p, drop_seed, (hipStream_t)stream);
}

int rvit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S,
                  int H, int64_t ldq, int64_t ldo, int64_t lddo, int64_t lddq, float scale, float drop_p,
                  uint64_t drop_seed, void* stream) {
  return rvit_attn_bwd_impl(qkv, out, dout, lse, dqkv, B, S, H, ldq, ldo, lddo, lddq, scale, drop_p, drop_seed,
                            (hipStream_t)stream);
}

int rvit_cls_head_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w0,
                      const float* b0, const void* w3, const float* b3, float* logits, int64_t ldl, int B, int E,
                      int hidden, int num_classes, void* stream) {
  return rvit_cls_head_fwd_impl(x, ldx, gamma, beta, eps, w0, b0, w3, b3, logits, ldl, B, E, hidden, num_classes,
                                (hipStream_t)stream);
}

int rvit_cls_head_supported(int E, int hidden, int num_classes) { return rvit_cls_head_ok(E, hidden, num_classes) ? 1 : 0; }

int rvit_attn_dh_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int head_dim, int64_t ldq, int64_t ldo,
                     float scale, float drop_p, uint64_t drop_seed, void* stream) {
  return rvit_attn_dh_fwd_impl(qkv, out, lse, B, S, H, head_dim, ldq, ldo, scale, drop_p, drop_seed, (hipStream_t)stream);
}

int rvit_attn_dh_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S,
                     int H, int head_dim, int64_t ldq, int64_t ldo, int64_t lddo, int64_t lddq, float scale, float drop_p,
                     uint64_t drop_seed, void* stream) {
  return rvit_attn_dh_bwd_impl(qkv, out, dout, lse, dqkv, B, S, H, head_dim, ldq, ldo, lddo, lddq, scale, drop_p,
                               drop_seed, (hipStream_t)stream);
}

int rvit_ce_loss(const float* logits, int64_t ldl, const int64_t* labels, float* loss_sum, float* dlogits,
                 int64_t lddl, void* dlogits_bf16, int64_t lddb, int B, int C, int C_pad, float label_smoothing,
                 float grad_scale, void* stream) {
  return rvit_ce_loss_impl(logits, ldl, (const long*)labels, loss_sum, dlogits, lddl, dlogits_bf16, lddb, B, C, C_pad,
                           label_smoothing, grad_scale, (hipStream_t)stream);
}

int rvit_colsum_bf16(const void* a, int64_t lda, float* out, int64_t M, int N, int64_t row_group, int64_t row_extra,
                     void* workspace, int64_t workspace_bytes, void* stream) {
  return rvit_colsum_impl(a, lda, out, M, N, row_group, row_extra, (float*)workspace,
                          workspace ? (size_t)workspace_bytes : 0, (hipStream_t)stream);
}

int rvit_batch_sum_f32(const float* x, int64_t ldb, float* out, int B, int64_t n, void* stream) {
  return rvit_batch_sum_impl(x, ldb, out, B, n, (hipStream_t)stream);
}

int rvit_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  return rvit_sumsq_impl(x, n, out, (float*)workspace, workspace ? (size_t)workspace_bytes : 0, (hipStream_t)stream);
}

int rvit_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm_sq,
                    float max_grad_norm, void* params_bf16, void* stream) {
  return rvit_adamw_impl(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq,
                         max_grad_norm, (bf16*)params_bf16, (hipStream_t)stream);
}

int rvit_cast_bf16(const float* x, void* out, int64_t n, void* stream) {
  return rvit_cast_bf16_impl(x, (bf16*)out, n, (hipStream_t)stream);
}

int rvit_transpose_batched_bf16(const void* src, void* dst, const int64_t* table, int n_entries, int64_t total_tiles,
                                void* stream) {
  return rvit_transpose_batched_impl((const bf16*)src, (bf16*)dst, (const long*)table, n_entries, total_tiles,
                                     (hipStream_t)stream);
}

int rvit_preprocess_u8(const uint8_t* frames, int B, int H, int W, int C, const int32_t* hcoef, const int32_t* hbounds,
                       int hk, const int32_t* vcoef, const int32_t* vbounds, int vk, int resized, const int32_t* crop_tl,
                       int out_side, uint8_t* tmp, float noise_std, uint64_t noise_seed, float* out, void* stream) {
  return rvit_preprocess_impl(frames, B, H, W, C, hcoef, hbounds, hk, vcoef, vbounds, vk, resized, crop_tl, out_side, tmp,
                              noise_std, noise_seed, out, (hipStream_t)stream);
}

int rvit_preprocess_u8_static(const uint8_t* frames, int B, int H, int W, int C, const int32_t* hcoef,
                              const int32_t* hbounds, int hk, const int32_t* vcoef, const int32_t* vbounds, int vk,
                              int resized, const int32_t* crop_tl, int out_side, float noise_std, uint64_t noise_seed,
                              float* out, void* stream) {
  return rvit_preprocess_static_impl(frames, B, H, W, C, hcoef, hbounds, hk, vcoef, vbounds, vk, resized, crop_tl,
                                     out_side, noise_std, noise_seed, out, (hipStream_t)stream);
}

int rvit_lstm_step_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* out,
                       float* c, float* gates, float* cseq, int B, int T, int In, int H, int step, void* stream) {
  const float* w[8] = {w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r};
  return rvit_lstm_step_fwd_impl(x, w, out, c, gates, cseq, B, T, In, H, step, w_ih_r ? 2 : 1, (hipStream_t)stream);
}

int rvit_lstm_step_bwd(const float* dout, const float* w_hh_t, const float* w_hh_t_r, const float* gates,
                       const float* cseq, float* dgates, float* dc, int B, int T, int H, int step, void* stream) {
  return rvit_lstm_step_bwd_impl(dout, w_hh_t, w_hh_t_r, gates, cseq, dgates, dc, B, T, H, step, w_hh_t_r ? 2 : 1,
                                 (hipStream_t)stream);
}

int rvit_gru_step_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* out,
                      float* saved, int B, int T, int In, int H, int step, void* stream) {
  const float* w[8] = {w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r};
  return rvit_gru_step_fwd_impl(x, w, out, saved, B, T, In, H, step, w_ih_r ? 2 : 1, (hipStream_t)stream);
}

int rvit_gru_step_bwd(const float* dout, const float* w_hh_t, const float* w_hh_t_r, const float* saved, const float* out,
                      float* dgates, float* dh_carry, int B, int T, int H, int step, void* stream) {
  return rvit_gru_step_bwd_impl(dout, w_hh_t, w_hh_t_r, saved, out, dgates, dh_carry, B, T, H, step, w_hh_t_r ? 2 : 1,
                                (hipStream_t)stream);
}

}  // extern "C"
