// Internal (non-ABI) declarations shared by the librvit translation units.
#pragma once
#include "common.h"


// Argument block of the MFMA GEMM / implicit-GEMM conv kernel (gemm.hip).
struct GemmArgs {
  const bf16* A;  // plain: [M, lda]; gather: NHWC activation [B, H, W, Cpix]
  long lda;
  const bf16* B;  // [N, ldb] (K-contiguous weights)
  long ldb;
  long M;
  int N, K;
  // gather (implicit-GEMM conv) geometry; mode 0 = plain row-major A
  int mode;
  int H, W, Cpix, Ho, Wo, KW, stride, pad, seg_len;
  // epilogue
  void* C;
  long ldc;
  int c_f32;         // 1: float output, 0: bf16
  bf16* C2;          // optional bf16 copy of the pre-activation value (training)
  long ldc2;
  const float* scale;  // per-column multiplier (folded BatchNorm), may be null
  const float* bias;   // per-column addend, may be null
  const void* R;       // residual, added before the activation, may be null
  long ldr;
  int r_f32;
  int r_period;      // >0: residual row = m % r_period (pos_embed), else the output row
  int act;
  int row_group;     // >0: output row = (m / g) * (g + row_extra) + m % g + row_extra
  int row_extra;
  int vec;           // 1: N, ld* and pointers allow 8-wide (16-byte) vector epilogue accesses
  int tile;          // 0 = auto, else forced tile config (tests)
};

int rvit_gemm_dispatch(const GemmArgs& a, hipStream_t stream);

int rvit_layernorm_fwd_impl(const void* x, int x_f32, long ldx, const float* gamma, const float* beta, void* y,
                            int y_f32, long ldy, float* mean, float* rstd, long rows, int E, float eps,
                            hipStream_t st);
int rvit_attn_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, long ldq, long ldo, float scale,
                       hipStream_t st);
int rvit_input_pack_impl(const float* x, void* out, int B, int C, int H, int W, int Hp, int Wp, int pt, int pl,
                         hipStream_t st);
int rvit_maxpool_impl(const void* x, void* y, void* argmax, int B, int H, int W, int C, hipStream_t st);
int rvit_embed_cls_impl(const float* cls, const float* pos, float* x, int B, int S, int E, hipStream_t st);
