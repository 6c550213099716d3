// extern "C" surface of librvit (declared in include/rvit.h).  Argument validation lives in the
// *_impl functions next to each kernel; this file only adapts ABI types.
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

extern "C" {

int rvit_version(void) { return RVIT_ABI_VERSION; }
const char* rvit_last_error(void) { return g_err; }

int rvit_gemm_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int c_dtype,
                   int64_t M, int N, int K, const float* scale, const float* bias, const void* R, int64_t ldr,
                   int r_dtype, int r_period, int act, void* preact, int64_t ldp, int row_group, int row_extra,
                   int tile, void* stream) {
  GemmArgs a{};
  a.A = (const bf16*)A; a.lda = lda;
  a.B = (const bf16*)B; a.ldb = ldb;
  a.M = M; a.N = N; a.K = K;
  a.mode = 0;
  a.C = C; a.ldc = ldc; a.c_f32 = c_dtype == RVIT_F32;
  a.C2 = (bf16*)preact; a.ldc2 = ldp;
  a.scale = scale; a.bias = bias;
  a.R = R; a.ldr = ldr; a.r_f32 = r_dtype == RVIT_F32; a.r_period = r_period;
  a.act = act; a.row_group = row_group; a.row_extra = row_extra; a.tile = tile;
  RVIT_CHECK_ARG(c_dtype == RVIT_BF16 || c_dtype == RVIT_F32, "gemm: bad c_dtype %d", c_dtype);
  RVIT_CHECK_ARG(act >= 0 && act <= 2, "gemm: bad act %d", act);
  RVIT_CHECK_ARG(ldc >= N && (!R || ldr >= N) && (!preact || ldp >= N), "gemm: output pitch smaller than N");
  RVIT_CHECK_ARG(row_group >= 0 && row_extra >= 0 && r_period >= 0, "gemm: negative row remap");
  // vector epilogue: 8 consecutive columns per lane, 16-byte accesses
  bool vec = (N & 7) == 0 && (ldc & (a.c_f32 ? 3 : 7)) == 0 && aligned_to(C, 16);
  if (scale) vec = vec && aligned_to(scale, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  if (R) vec = vec && (ldr & (a.r_f32 ? 3 : 7)) == 0 && aligned_to(R, 16);
  if (preact) vec = vec && (ldp & 7) == 0 && aligned_to(preact, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
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
  a.A = (const bf16*)x; a.lda = 0;
  a.B = (const bf16*)w; a.ldb = (long)KH * KW * seg_len;
  a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * seg_len;
  a.mode = 1;
  a.H = H; a.W = W; a.Cpix = Cpix; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
  a.seg_len = seg_len;
  a.C = y; a.ldc = Cout; a.c_f32 = 0;
  a.scale = scale; a.bias = bias;
  a.R = residual; a.ldr = Cout; a.r_f32 = 0; a.r_period = 0;
  a.act = act; a.tile = tile;
  RVIT_CHECK_ARG(act >= 0 && act <= 2, "conv: bad act %d", act);
  bool vec = (Cout & 7) == 0 && aligned_to(y, 16);
  if (scale) vec = vec && aligned_to(scale, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  if (residual) vec = vec && aligned_to(residual, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_conv2d_tokens_fwd(const void* x, const void* w, float* tokens, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* bias,
                           const float* pos_embed, int tile, void* stream) {
  RVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                 "conv_tokens: bad geometry");
  RVIT_CHECK_ARG((long)B * H * W < 0x7fffffffL, "conv_tokens: B*H*W exceeds 2^31 pixels");
  RVIT_CHECK_ARG(tokens && pos_embed, "conv_tokens: null pointer");
  GemmArgs a{};
  a.A = (const bf16*)x; a.lda = 0;
  a.B = (const bf16*)w; a.ldb = (long)KH * KW * seg_len;
  a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * seg_len;
  a.mode = 1;
  a.H = H; a.W = W; a.Cpix = Cpix; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
  a.seg_len = seg_len;
  a.C = tokens; a.ldc = Cout; a.c_f32 = 1;
  a.bias = bias;
  a.R = pos_embed + Cout; a.ldr = Cout; a.r_f32 = 1; a.r_period = Ho * Wo;
  a.row_group = Ho * Wo; a.row_extra = 1;
  a.act = RVIT_ACT_NONE; a.tile = tile;
  bool vec = (Cout & 7) == 0 && aligned_to(tokens, 16) && aligned_to(pos_embed, 16);
  if (bias) vec = vec && aligned_to(bias, 16);
  a.vec = vec;
  return rvit_gemm_dispatch(a, (hipStream_t)stream);
}

int rvit_input_pack_nhwc4(const float* x_nchw, void* out, int B, int C, int H, int W, int Hp, int Wp, int pad_top,
                          int pad_left, void* stream) {
  return rvit_input_pack_impl(x_nchw, out, B, C, H, W, Hp, Wp, pad_top, pad_left, (hipStream_t)stream);
}

int rvit_maxpool3x3s2_nhwc(const void* x, void* y, void* argmax, int B, int H, int W, int C, void* stream) {
  return rvit_maxpool_impl(x, y, argmax, B, H, W, C, (hipStream_t)stream);
}

int rvit_embed_cls(const float* cls_token, const float* pos_embed, float* x, int B, int S, int E, void* stream) {
  return rvit_embed_cls_impl(cls_token, pos_embed, x, B, S, E, (hipStream_t)stream);
}

int rvit_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y,
                       int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows, int E, float eps,
                       void* stream) {
  return rvit_layernorm_fwd_impl(x, x_dtype == RVIT_F32, ldx, gamma, beta, y, y_dtype == RVIT_F32, ldy, mean, rstd,
                                 rows, E, eps, (hipStream_t)stream);
}

int rvit_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                  void* stream) {
  return rvit_attn_fwd_impl(qkv, out, lse, B, S, H, ldq, ldo, scale, (hipStream_t)stream);
}

}  // extern "C"
