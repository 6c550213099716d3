/* librvit — C ABI of the MI355X (gfx950) kernels behind vit_amd.CNNViTHybrid.
 *
 * The reference (bheering007/ViT) has no FFI/plugin interface: its hot path is the nn.Module
 * `CNNViTHybrid` (train_vit.py:54-185) calling torch.nn arithmetic.  Every entry below replaces
 * one of those torch calls; the cited file:line is the reference (or torch) call site whose
 * arithmetic the entry reproduces.  The Python host (vit_amd/engine.py) binds these with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library borrows it for the
 *     duration of the enqueue and never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls only enqueue work;
 *   - return value 0 = ok, negative = RVIT_E*; rvit_last_error() returns a thread-local text;
 *   - dtype codes: RVIT_BF16 = 0, RVIT_F32 = 1; activation codes RVIT_ACT_*;
 *   - bf16 tensors are raw 16-bit words (torch.bfloat16 storage), row-major, pitches in ELEMENTS.
 */
#ifndef RVIT_H
#define RVIT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RVIT_ABI_VERSION 1

#define RVIT_BF16 0
#define RVIT_F32 1

#define RVIT_ACT_NONE 0
#define RVIT_ACT_RELU 1
#define RVIT_ACT_GELU 2 /* exact erf form, nn.GELU() / activation='gelu' */

#define RVIT_OK 0
#define RVIT_EINVAL (-1)
#define RVIT_ELAUNCH (-2)
#define RVIT_EUNSUPPORTED (-3)

int rvit_version(void);
const char* rvit_last_error(void);

/* C[M,N] = act( scale[n] * (A[M,K] . B[N,K]^T) + bias[n] + R ) on bf16 MFMA, fp32 accumulate.
 * Replaces every nn.Linear on the path: packed QKV in-proj (torch functional.py:5819-5833),
 * attention out-proj + residual (functional.py:6632-6637, transformer.py:947-949), linear1+GELU
 * and linear2+residual (transformer.py:980-982), the head Linears (train_vit.py:122-127), and the
 * 1x1 patch_embed conv with the flatten/transpose/cls-offset/pos_embed add folded into the
 * epilogue (train_vit.py:93,149-159).
 *   A: bf16 [M,lda]   B: bf16 [N,ldb] (nn.Linear weight layout)   K % 8 == 0
 *   C: bf16 or f32 [.,ldc];  preact (optional, bf16 [.,ldp]) receives the value before R/act
 *   R: optional residual (bf16 or f32), row = m % r_period when r_period > 0, else the C row
 *   row_group > 0 remaps the output row: (m / g) * (g + row_extra) + m % g + row_extra
 *   tile: 0 = automatic tile choice, 1/2/3 force 256x64 / 128x64 / 64x64 (testing). */
int rvit_gemm_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int c_dtype,
                   int64_t M, int N, int K, const float* scale, const float* bias, const void* R, int64_t ldr,
                   int r_dtype, int r_period, int act, void* preact, int64_t ldp, int row_group, int row_extra,
                   int tile, void* stream);

/* NHWC implicit-GEMM convolution, bf16 MFMA: y = act( scale[c]*conv(x, w) + bias[c] + residual ).
 * Replaces the timm resnet18 feature extractor convs + (eval-mode, folded) BatchNorm + ReLU +
 * BasicBlock residual add: `self.cnn_backbone(x)[-1]` (train_vit.py:87,145-146).
 *   x: bf16 [B,H,W,Cpix]; w: bf16 [Cout, K] with K = KH*KW*seg_len laid out (kh, kw, channel run);
 *   seg_len = contiguous bf16 run read per (kh,kw) tap (= Cin for ordinary convs; 32 with KW=1
 *   for the 7x7 stem conv reading 8 px x 4 ch from the haloed NHWC4 image);
 *   y / residual: bf16 [B,Ho,Wo,Cout]. Out-of-image taps read zero (padding). */
int rvit_conv2d_fwd_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cpix, int seg_len, int KH,
                         int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* scale,
                         const float* bias, const void* residual, int act, int tile, void* stream);

/* Patch-embedding convolution straight into the token stream (pure-ViT `patch16` layout; the
 * same flatten/transpose/cls-offset/pos_embed epilogue as the 1x1 case, train_vit.py:149-159):
 * tokens[b, 1+p, :] = conv(x, w)[b, p, :] + bias + pos_embed[1+p, :], fp32 [B, Ho*Wo+1, Cout].
 * Row 0 of every sample (cls) is left untouched (rvit_embed_cls writes it). */
int rvit_conv2d_tokens_fwd(const void* x, const void* w, float* tokens, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* bias,
                           const float* pos_embed, int tile, void* stream);

/* fp32 NCHW image batch -> zero-haloed bf16 NHWC4 (channel 3 = 0).  Input contract of
 * CNNViTHybrid.forward (train_vit.py:141-146: x is [B,3,224,224] float).  Writes only the
 * interior; the caller zeroes the buffer once. */
int rvit_input_pack_nhwc4(const float* x_nchw, void* out, int B, int C, int H, int W, int Hp, int Wp, int pad_top,
                          int pad_left, void* stream);

/* 3x3 / stride 2 / pad 1 max-pool on NHWC bf16 (resnet stem max-pool, train_vit.py:87).
 * argmax (optional, uint8 same shape as y) records the winning tap for the backward pass. */
int rvit_maxpool3x3s2_nhwc(const void* x, void* y, void* argmax, int B, int H, int W, int C, void* stream);

/* x[b,0,:] = cls_token + pos_embed[0] on the fp32 residual stream [B,S,E] (train_vit.py:155-159). */
int rvit_embed_cls(const float* cls_token, const float* pos_embed, float* x, int B, int S, int E, void* stream);

/* LayerNorm over the last axis, eps inside the sqrt, affine (torch transformer.py:774,947-950;
 * train_vit.py:119,166).  x: bf16/f32 [rows, ldx]; y: bf16/f32 [rows, ldy]; mean/rstd optional
 * fp32 [rows] (saved for backward). */
int rvit_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y,
                       int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows, int E, float eps,
                       void* stream);

/* softmax(q k^T * scale) v per (batch, head), head_dim 64, S <= 256, no mask
 * (torch functional.py:6623-6631 -> scaled_dot_product_attention).
 *   qkv: bf16 [B*S, ldq] holding q | k | v column blocks of width H*64 (the in-proj output);
 *   out: bf16 [B*S, ldo], head h at columns h*64..; lse (optional) fp32 [B,H,S]. */
int rvit_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RVIT_H */
