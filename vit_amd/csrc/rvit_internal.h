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
  int transposed;    // gather: 1 = transposed conv (input-gradient of a strided conv)
  int a_row_group;   // plain: >0 remaps the A row like row_group/row_extra below
  int a_row_extra;
  // epilogue
  void* C;
  long ldc;
  int c_f32;         // 1: float output, 0: bf16
  bf16* C2;          // optional bf16 copy of the pre-activation value (training)
  long ldc2;
  const float* scale;  // per-column multiplier (folded BatchNorm), may be null
  const float* bias;   // per-column addend, may be null
  const bf16* G;       // gate operand (same rows as C): gelu'(G) or (G > 0) multiplies the value
  long ldg;
  int gate;
  const void* R;       // residual, added before the activation, may be null
  long ldr;
  int r_f32;
  int r_period;      // >0: residual row = m % r_period (pos_embed), else the output row
  int act;
  int row_group;     // >0: output row = (m / g) * (g + row_extra) + m % g + row_extra
  int row_extra;
  int vec;           // 1: N, ld* and pointers allow 8-wide (16-byte) vector epilogue accesses
  int tile;          // 0 = auto, else forced tile config (tests)
  int drop_mode;     // 0 none, RVIT_DROP_PRE_RESIDUAL, RVIT_DROP_POST_ACT; element index = out_row * N + col
  DropKey drop;
  // Optional batch statistics of the STORED (bf16-rounded) output, for the train-mode BatchNorm that follows a conv
  // (rvit_conv2d_fwd_stats_bf16): every M tile writes sum and sum of squares of its rows per column to its own slot,
  // stat_part[slot * 2N + n] / [slot * 2N + N + n]; rvit_bn_stats_finalize adds the slots in order.  Host side: the
  // dispatcher stores the slot count the chosen kernel will write to *stat_slots (0: the kernel has no such epilogue or
  // the slots exceed stat_cap_slots - the caller then runs rvit_bn_stats) and clears stat_part in that case.
  float* stat_part;
  long stat_cap_slots;
  int* stat_slots;
  // ... or, for a dgrad conv whose output is the gradient entering a BatchNorm (+ ReLU) (rvit_conv2d_dgrad_bnr_bf16), the
  // two sums of that norm's backward pass (bn.hip bn_bwd_reduce_kernel) over the rows of the tile, in the same slots:
  // sum d | rstd * sum d (y - mean), d = stored bf16 output where the ReLU bit is set (bnr_mask null: everywhere).
  // bnr_y: the norm's input map, [M, N] bf16; bnr_mask: one byte per 8 channels; bnr_fsums: the forward sums [2N].
  const bf16* bnr_y;
  const uint8_t* bnr_mask;
  const float* bnr_fsums;
  float bnr_inv_m, bnr_eps;
};

int rvit_gemm_dispatch(const GemmArgs& a, hipStream_t stream);
bool rvit_gemm8p_ok(const GemmArgs& a);                       // gemm8p.hip (256x256 tile, eight waves, 8-phase K loop)
int rvit_gemm8p(const GemmArgs& a, hipStream_t stream);
bool rvit_conv3x3_direct_ok(const GemmArgs& a);               // conv3x3.hip
int rvit_conv3x3_direct(const GemmArgs& a, int variant, hipStream_t stream);
enum { RVIT_CONV_GATHER = 0, RVIT_CONV_DIRECT = 1, RVIT_CONV_C64 = 2, RVIT_CONV_C256 = 3, RVIT_CONV_P8 = 4 };
int rvit_conv3x3_auto(const GemmArgs& a);                     // conv3x3.hip: the kernel tile 0 runs (RVIT_CONV_*)
bool rvit_conv3x3_p8_ok(const GemmArgs& a);                   // conv3x3_p8.hip (eight waves, four-phase taps, one workgroup per CU)
bool rvit_conv3x3_p8_pays(const GemmArgs& a);                 // ... and its tiles / rounds are full enough to win
int rvit_conv3x3_p8(const GemmArgs& a, hipStream_t stream);
int rvit_stem_conv1_wgrad_impl(const float* x, const void* dy, float* dw, float* ws, long ws_bytes, int B, hipStream_t st);
int rvit_stem_conv1_rows_impl(const float* x, const void* w, void* y, float* stat_ws, long stat_ws_bytes, int* stat_slots,
                              int B, hipStream_t st);   // conv1.hip
bool rvit_conv3x3_c64_ok(const GemmArgs& a);                  // conv3x3_c64.hip (64 -> 64 channels, weights resident)
int rvit_conv3x3_c64(const GemmArgs& a, hipStream_t stream);
int rvit_stem_conv1_pool_impl(const void* img, const void* w, const float* scale, const float* bias, void* pooled,
                              int B, int relu, int img_f32, hipStream_t st);  // conv1.hip

// Argument block of the MX-fp8 GEMM (gemm_mx.hip): C = act(A_q . B_q^T + bias) + R, optional MX copy of C.
struct MxGemmArgs {
  const uint8_t* A;   // e4m3 [M, lda]
  long lda;
  const uint8_t* SA;  // E8M0 scales, tiled [ceil(M/16)][ceil(K/512)][16][4][4] (gemm_mx.hip)
  const uint8_t* B;   // e4m3 [N, ldb]
  long ldb;
  const uint8_t* SB;  // E8M0 scales, tiled [ceil(N/16)][ceil(K/512)][16][4][4]
  long M;
  int N, K;
  void* C;            // bf16 / f32 [M, ldc], may be null when only the MX copy is wanted
  long ldc;
  int c_f32;
  const float* bias;
  const void* R;      // residual, added after the activation-free result (bf16 / f32), may be null
  long ldr;
  int r_f32;
  int act;
  uint8_t* Cq;        // optional e4m3 copy of the result [M, ldcq] ...
  long ldcq;
  uint8_t* SCq;       // ... with its E8M0 scales, tiled [ceil(M/16)][ceil(N/512)][16][4][4]
};
int rvit_sumsq_impl(const float* x, long n, float* out, float* ws, size_t ws_bytes, hipStream_t st);  // optim.hip
int rvit_adamw_impl(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                    float wd, int step, const float* gnorm_sq, float max_norm, bf16* mirror, hipStream_t st);
int rvit_preprocess_impl(const uint8_t* img, int B, int H, int W, int C, const int* hcoef, const int* hbounds, int hk,
                         const int* vcoef, const int* vbounds, int vk, int resized, const int* crop, int out_side,
                         uint8_t* tmp, float noise_std, uint64_t noise_seed, float* out, hipStream_t st);  // preproc.hip
int rvit_preprocess_static_impl(const uint8_t* img, int B, int H, int W, int C, const int* hcoef, const int* hbounds,
                                int hk, const int* vcoef, const int* vbounds, int vk, int resized, const int* crop,
                                int out_side, float noise_std, uint64_t noise_seed, float* out, hipStream_t st);
int rvit_lstm_step_fwd_impl(const float* x, const float* const* w, float* out, float* c, float* gates, float* cseq, int B,
                            int T, int In, int H, int s, int ndir, hipStream_t st);  // lstm.hip
int rvit_lstm_step_bwd_impl(const float* dout, const float* whh_t0, const float* whh_t1, const float* gates,
                            const float* cseq, float* dG, float* dc, int B, int T, int H, int s, int ndir,
                            hipStream_t st);
int rvit_gru_step_fwd_impl(const float* x, const float* const* w, float* out, float* saved, int B, int T, int In, int H,
                           int s, int ndir, hipStream_t st);
int rvit_gru_step_bwd_impl(const float* dout, const float* whh_t0, const float* whh_t1, const float* saved,
                           const float* out, float* dG, float* dhd, int B, int T, int H, int s, int ndir,
                           hipStream_t st);
int rvit_cast_bf16_impl(const float* x, bf16* out, long n, hipStream_t st);
int rvit_transpose_batched_impl(const bf16* src, bf16* dst, const long* table, int n_entries, long total_tiles,
                                hipStream_t st);
int rvit_gemm_mxfp8_dispatch(const MxGemmArgs& a, hipStream_t st);
int rvit_gemm_mx8p_mode(const MxGemmArgs& a);                  // gemm_mx8p.hip: -1 = not a case of the eight-wave kernel
int rvit_gemm_mx8p(const MxGemmArgs& a, int mode, hipStream_t st);
int rvit_quantize_mx_impl(const void* x, int x_f32, long ldx, uint8_t* q, long ldq, uint8_t* sc, long rows, int K,
                          hipStream_t st);

// Argument block of the weight-gradient kernel (wgrad.hip): dW[N,K] += dY^T X.
struct WgradArgs {
  const bf16* dY;  // [M, ldy]
  long ldy;
  int y_row_group, y_row_extra;  // optional row remap of dY (token rows past the cls row)
  const bf16* X;   // plain: [M, ldx]; gather: NHWC activation
  long ldx;
  int x_row_group, x_row_extra;
  float* dW;       // [N_store, ldw] fp32, accumulated with atomics
  long ldw;
  long M;
  int N, N_store, K;
  int mode;        // 0 plain, 1 gather
  int H, W, Cpix, Ho, Wo, KW, stride, pad, seg_len;
  float* dbias;    // optional [N_store] fp32: += column sums of dY (plain mode, N > 64; the dispatcher checks)
  int splits;      // 0 = auto; > 0 M splits of the implicit-GEMM kernel; -1 force that kernel; -2 force the direct 3x3 kernel
  // ordered mode (ordered.hip): host side, the caller's workspace ...
  float* ws;       // null: fp32 atomics into dW / dbias
  size_t ws_bytes;
  // ... device side, filled by the dispatcher: split s (sub-slot u of `sub`) stores its tile at
  // part + (s * sub + u) * part_stride + n * K + k and its bias sums at part_bias + (s * sub + u) * N_store + n
  float* part;
  long part_stride;
  float* part_bias;
  int part_sub;    // slots per split (2 where two wave groups share an output tile)
};

int rvit_ordered_finalize(const float* part, int nslots, long slot_stride, float* out, long rows, int cols, long ld,
                          hipStream_t st);
int rvit_ordered_finalize_pair(const float* part, int nslots, long slot_stride, float* out, long rows, int cols, long ld,
                               const float* partb, long strideb, float* outb, int nb, hipStream_t st);
int rvit_bn_stats_finalize_impl(const float* part, int nslots, int C, float* sums, float* stage, hipStream_t st);
int rvit_ordered_splits(long units, long splits, int sub, long elems, size_t ws_bytes, const char* what, long* out);

int rvit_wgrad_dispatch(const WgradArgs& a, hipStream_t st);
bool rvit_conv3x3_wgrad_direct_ok(const WgradArgs& a);            // wgrad3x3.hip
bool rvit_conv3x3_wgrad_direct_pays(const WgradArgs& a);
int rvit_conv3x3_wgrad_direct(const WgradArgs& a, hipStream_t st);

int rvit_layernorm_fwd_impl(const void* x, int x_f32, long ldx, const float* gamma, const float* beta, void* y,
                            int y_f32, long ldy, float* mean, float* rstd, long rows, int E, float eps,
                            hipStream_t st);
int rvit_layernorm_mx_fwd_impl(const void* x, int x_f32, long ldx, const float* gamma, const float* beta, uint8_t* q,
                               long ldq, uint8_t* sc, long rows, int E, float eps, hipStream_t st);
int rvit_layernorm_bwd_impl(const void* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                            const float* gamma, float* dxs, long lddx, int accumulate, void* dx_bf16, long lddb,
                            float* dgamma, float* dbeta, long rows, int E, float* ws, size_t ws_bytes,
                            float drop_p, uint64_t drop_seed, hipStream_t st);
int rvit_encoder_mlp_impl(float* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                          const void* w2, const float* b2, long T, int E, int hidden, int rows_per_wg, hipStream_t st);
int rvit_encoder_attn_impl(float* x, const float* gamma, const float* beta, float eps, const void* wqkv,
                           const float* bqkv, const void* wout, const float* bout, int B, int S, int E, int H,
                           hipStream_t st);
int rvit_encoder_stack_impl(float* x, const void* const* layer_ptrs, int n_layers, float eps, int B, int S, int E, int H,
                            int hidden, hipStream_t st);
int rvit_attn_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, long ldq, long ldo, float scale,
                       float drop_p, uint64_t drop_seed, hipStream_t st);
int rvit_attn_fwd_fp8_impl(const void* qkv, uint8_t* out_q, long ldo, uint8_t* out_s, int B, int S, int H, long ldq,
                           float scale, hipStream_t st);
int rvit_attn_fwd_mx_impl(const void* qkv, uint8_t* out_q, long ldo, uint8_t* out_s, int B, int S, int H, long ldq,
                          float scale, hipStream_t st);
int rvit_dwconv_impl(const void* x, const void* w, const float* scale, const float* bias, void* out, int B, int H, int W,
                     int C, int K, int stride, int pad, int act, hipStream_t st);
int rvit_avgpool_impl(const void* x, void* out, int B, int HW, int C, hipStream_t st);
int rvit_channel_scale_impl(void* x, const void* gate, int B, int HW, int C, hipStream_t st);
int rvit_squeeze_excite_impl(void* x, const void* w1, const float* b1, const void* w2, const float* b2, int B, int HW, int C,
                             int RD, hipStream_t st);
int rvit_mb_bn_stats_impl(const void* y, long M, int C, float* sums, void* ws, size_t ws_bytes, hipStream_t st);
int rvit_mb_bn_apply_impl(const void* y, const float* sums, const float* gamma, const float* beta, const void* res, void* z,
                          long M, int C, float eps, int act, float momentum, float* running_mean, float* running_var,
                          hipStream_t st);
int rvit_mb_bn_bwd_impl(const void* dz, const void* y, const float* fsums, float* bsums, const float* gamma, const float* beta,
                        void* dy, float* dgamma, float* dbeta, long M, int C, float eps, int act, void* ws, size_t ws_bytes,
                        hipStream_t st);
int rvit_dwconv_dgrad_impl(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                           hipStream_t st);
int rvit_dwconv_wgrad_impl(const void* dy, const void* x, float* dw, int B, int H, int W, int C, int K, int stride, int pad,
                           void* ws, size_t ws_bytes, hipStream_t st);
int rvit_channel_affine_impl(const void* x, const void* gate, const void* add, void* out, int B, int HW, int C, hipStream_t st);
int rvit_channel_dot_impl(const void* a, const void* x, const void* gate, void* out, int B, int HW, int C, hipStream_t st);
long rvit_pointwise_lds_bytes(int K, int N);
int rvit_pointwise_impl(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* out,
                        long M, int K, int N, int act, hipStream_t st);
int rvit_conv3x3s2_rgb_impl(const float* x, const float* w, const float* scale, const float* bias, void* out, int B, int H,
                            int W, int Cout, int act, hipStream_t st);
int rvit_attn_dh_fwd_impl(const void* qkv, void* out, float* lse, int B, int S, int H, int Dh, long ldq, long ldo,
                          float scale, float drop_p, uint64_t drop_seed, hipStream_t st);
int rvit_attn_dh_bwd_impl(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H,
                          int Dh, long ldq, long ldo, long lddo, long lddq, float scale, float drop_p, uint64_t drop_seed,
                          hipStream_t st);
bool rvit_cls_head_ok(int E, int HID, int C);   // head.hip
int rvit_cls_head_fwd_impl(const float* x, long ldx, const float* gamma, const float* beta, float eps, const void* w0,
                           const float* b0, const void* w3, const float* b3, float* logits, long ldl, int B, int E, int HID,
                           int C, hipStream_t st);
int rvit_attn16_impl(const void* qkv, void* out, int B, int S, int H, long ldq, long ldo, float scale, hipStream_t st);
int rvit_attn_bwd_impl(const void* qkv, const void* O, const void* dO, const float* lse, void* dqkv, int B, int S, int H,
                       long ldq, long ldo, long lddo, long lddq, float scale, float drop_p, uint64_t drop_seed,
                       hipStream_t st);
int rvit_dropout_apply_impl(const void* in, void* out, int f32, long rows, int cols, long ld_in, long ld_out,
                            float p, uint64_t seed, hipStream_t st);
int rvit_dropout_mask_impl(uint8_t* out, long n, float p, uint64_t seed, hipStream_t st);
int rvit_input_pack_impl(const float* x, void* out, int B, int C, int H, int W, int Hp, int Wp, int pt, int pl,
                         hipStream_t st);
int rvit_maxpool_impl(const void* x, void* y, void* argmax, int B, int H, int W, int C, hipStream_t st);
int rvit_embed_cls_impl(const float* cls, const float* pos, float* x, int B, int S, int E, float drop_p,
                        uint64_t drop_seed, hipStream_t st);
int rvit_ce_loss_impl(const float* logits, long ldl, const long* labels, float* loss_sum, float* dlogits, long lddl,
                      void* dlogits_bf16, long lddb, int B, int C, int C_pad, float eps, float gscale, hipStream_t st);
int rvit_colsum_impl(const void* a, long lda, float* out, long M, int N, long row_group, long row_extra,
                     float* ws, size_t ws_bytes, hipStream_t st);
int rvit_batch_sum_impl(const float* x, long ldb, float* out, int B, long n, hipStream_t st);
int rvit_bn_stats_impl(const void* y, long M, int C, float* sums, float* ws, hipStream_t st);
int rvit_bn_apply_impl(const void* y, const float* sums, const float* gamma, const float* beta, const void* res,
                       void* z, long M, int C, float eps, int relu, float momentum, float* running_mean,
                       float* running_var, void* relu_mask, hipStream_t st);
int rvit_bn_apply_pool_impl(const void* y, const float* sums, const float* gamma, const float* beta, void* pooled,
                            void* argmax, void* relu_mask, int B, int H, int W, int C, float eps, float momentum,
                            float* running_mean, float* running_var, hipStream_t st);
int rvit_bn_bwd_reduce_impl(const void* dz, const void* z, const void* y, const float* fsums, long M, int C, float eps,
                            int relu, float* bsums, float* ws, hipStream_t st);
int rvit_bn_bwd_apply_impl(const void* dz, const void* z, const void* y, const float* fsums, const float* bsums,
                           const float* gamma, void* dy, void* dzm_out, float* dgamma, float* dbeta, long M, int C,
                           float eps, int relu, hipStream_t st);
int rvit_bn_bwd_reduce_pool_impl(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                                 const float* fsums, int B, int H, int W, int C, float eps, float* bsums, float* ws,
                                 hipStream_t st);
int rvit_bn_bwd_apply_pool_impl(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                                const float* fsums, const float* bsums, const float* gamma, void* dy, float* dgamma,
                                float* dbeta, int B, int H, int W, int C, float eps, hipStream_t st);
int rvit_maxpool_bwd_impl(const void* dpool, const void* argmax, void* dx, int B, int H, int W, int C, hipStream_t st);
