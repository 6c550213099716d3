/* librvit — C ABI of the MI355X (gfx950) kernels behind vit_amd.CNNViTHybrid.
 *
 * The reference (bheering007/ViT) has no FFI/plugin interface: its hot path is the nn.Module
 * `CNNViTHybrid` (train_vit.py:54-185) calling torch.nn arithmetic.  Every entry below replaces
 * one of those torch calls (forward) or its autograd counterpart run by
 * `scaler.scale(loss).backward()` (train_vit.py:939); the cited file:line is the reference (or
 * torch) call site whose arithmetic the entry reproduces.  The Python host (vit_amd/engine.py)
 * binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library borrows it for the
 *     duration of the enqueue and never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls only enqueue work;
 *   - return value 0 = ok, negative = RVIT_E*; rvit_last_error() returns a thread-local text;
 *   - dtype codes: RVIT_BF16 = 0, RVIT_F32 = 1; activation codes RVIT_ACT_*;
 *   - bf16 tensors are raw 16-bit words (torch.bfloat16 storage), row-major, pitches in ELEMENTS;
 *   - "row remap (g, e)": logical row m maps to physical row (m / g) * (g + e) + m % g + e when
 *     g > 0 (the patch rows of the token stream, which skip the cls row of every sample).
 */
#ifndef RVIT_H
#define RVIT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RVIT_ABI_VERSION 29

#define RVIT_BF16 0
#define RVIT_F32 1

#define RVIT_ACT_NONE 0
#define RVIT_ACT_RELU 1
#define RVIT_ACT_GELU 2 /* exact erf form, nn.GELU() / activation='gelu' */
#define RVIT_ACT_HSWISH 3   /* x * relu6(x + 3) / 6  (nn.Hardswish; timm mobilenetv3) */
#define RVIT_ACT_HSIGMOID 4 /* relu6(x + 3) / 6      (timm hard_sigmoid: the squeeze-excite gate of mobilenetv3) */

#define RVIT_GATE_GELU_GRAD 1 /* value *= gelu'(gate)   (backward of nn.GELU) */
#define RVIT_GATE_RELU_MASK 2 /* value *= (gate > 0)    (backward of ReLU)    */

/* Dropout (nn.Dropout(p) at train_vit.py:100,107,125 = pos_drop, the encoder layers' dropout /
 * dropout1 / dropout2 and the head's; attention-probability dropout, torch functional.py:6569-6570).
 * A site is keyed by (drop_p, drop_seed).  The generator (csrc/common.h make_drop_key / rvit_drop_hash; restated in
 * NumPy by oracle/ref_dropout.py keep_mask_numpy, and rvit_dropout_mask below must equal that bit for bit):
 *   key      z = splitmix64 finaliser of drop_seed (z = seed + 0x9E3779B97F4A7C15; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;
 *            z = (z ^ z>>27) * 0x94D049BB133111EB; z ^= z>>31), k0 = low 32 bits, k1 = high 32 bits;
 *   pairs    ONE 32-bit hash decides the two neighbouring elements 2j and 2j+1 (pair index j = idx >> 1):
 *            h = mix24(j + k0; 0xB5297B, 0x68E31D); h ^= h >> 15; h = mix24(h ^ k1; 0x9E3779, 0x7FEB35); h ^= h >> 16,
 *            where mix24(x; a, b) = (x & 0xFFFFFF) * a + (x >> 8) * b mod 2^32 (each factor's low 24 bits);
 *   decision element 2j is kept iff (h & 0xFFFF) >= T, element 2j+1 iff (h >> 16) >= T, T = round(p * 65536): the drop
 *            probability is quantised to 1/65536; kept values are scaled by 1/(1-p) (of the unquantised p);
 *   index    idx = row * cols + col of the tensor the site acts on (GEMM epilogues: the output matrix; rows of an even
 *            width, so a pair never straddles rows; rvit_layernorm_bwd_drop uses (row * E + e) >> 1 likewise); the
 *            attention-probability site uses ((image * H + head) * S + query) * S2 + key with S2 = S rounded up to even.
 * Nothing is stored: the backward entry points regenerate the decision from the same (p, seed).  torch's Philox stream
 * is NOT reproduced (no kernel can): the distribution matches, the individual masks do not.
 * drop_p == 0 disables.  Position of the dropout in a GEMM epilogue: */
#define RVIT_DROP_PRE_RESIDUAL 1 /* C = act(drop(scale*acc + bias) + R): dropout1 / dropout2 */
#define RVIT_DROP_POST_ACT 2     /* C = drop(act(... + R)): FFN / head dropout, pos_drop; also the mask multiply of
                                    the matching backward GEMM (after `gate`)                     */

#define RVIT_OK 0
#define RVIT_EINVAL (-1)
#define RVIT_ELAUNCH (-2)
#define RVIT_EUNSUPPORTED (-3)

int rvit_version(void);
const char* rvit_last_error(void);

/* C[M,N] = act( gate( scale[n] * (A[M,K] . B[N,K]^T) + bias[n] ) + R ) on bf16 MFMA, fp32 accumulate.
 * Forward: every nn.Linear on the path — packed QKV in-proj (torch functional.py:5819-5833),
 * attention out-proj + residual (functional.py:6632-6637, transformer.py:947-949), linear1+GELU
 * and linear2+residual (transformer.py:980-982), the head Linears (train_vit.py:122-127), and the
 * 1x1 patch_embed conv with the flatten/transpose/cls-offset/pos_embed add folded into the
 * epilogue (train_vit.py:93,149-159).  Backward: the input-gradient of the same Linears
 * (B = the transposed weight), with the GELU derivative applied through `gate`.
 *   A: bf16 [.,lda], optional row remap (a_row_group, a_row_extra)   B: bf16 [N,ldb]   K % 8 == 0
 *   C: bf16 or f32 [.,ldc];  preact (optional, bf16 [.,ldp]) receives the value before R/act
 *   gate: optional bf16 [.,ldg] indexed like C, gate_mode RVIT_GATE_*
 *   R: optional residual (bf16 or f32), row = m % r_period when r_period > 0, else the C row
 *   row remap (row_group, row_extra) applies to C, preact, gate and R
 *   drop_mode / drop_p / drop_seed: dropout in the epilogue (above); element index = C row * N + column
 *   tile: 0 = automatic tile choice, 1/2/3/4 force 256x64 / 128x64 / 64x64 / 256x256 of sixteen 64x64 waves, 5 the
 *   eight-wave 256x256 kernel with the four-phase K tile (csrc/gemm8p.hip: plain A operand, 16-byte aligned rows, K % 64
 *   == 0; what products of >= 512 such tiles with K >= 512 run: the ViT-B / ViT-L Linears), 6 / 7 = 256x128 on sixteen /
 *   128x128 on eight waves (7: implicit-GEMM convs with N % 128 == 0 that fill the chip); for testing: every tile
 *   contracts K in the same order, so the choice never changes a bit of the result. */
int rvit_gemm_bf16(const void* A, int64_t lda, int a_row_group, int a_row_extra, const void* B, int64_t ldb, void* C,
                   int64_t ldc, int c_dtype, int64_t M, int N, int K, const float* scale, const float* bias,
                   const void* gate, int64_t ldg, int gate_mode, const void* R, int64_t ldr, int r_dtype,
                   int r_period, int act, void* preact, int64_t ldp, int row_group, int row_extra, int drop_mode,
                   float drop_p, uint64_t drop_seed, int tile, void* stream);

/* MX-fp8 GEMM on the block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4):
 *   C[M,N] = act( A_q[M,K] . B_q[N,K]^T + bias[n] ) + R
 * A_q / B_q: OCP e4m3 bytes; SA / SB: E8M0 scale bytes, one per 32 consecutive K elements (OCP Microscaling
 * MXFP8: element value = e4m3 * 2^(scale - 127)), stored TILED in the order the instruction consumes them:
 * as [ceil(rows/16)][ceil(K/512)][16][4][4] bytes, the scale of (row r, element k) at byte
 * (((r/16) * ceil(K/512) + k/512) * 64 + (r%16) * 4 + (k/32)%4) * 4 + (k/128)%4; the caller zero-fills the
 * padding.  fp32 accumulate.
 * BASELINE.json configs[4] (ViT-L/16-scale fp8 inference): the encoder Linears, torch
 * functional.py:5819-5833,6632-6637 and transformer.py:980-982.  No counterpart in the reference (fp32/fp16 only).
 *   K % 128 == 0, N % 8 == 0; C (optional) bf16 / f32; R optional residual bf16 / f32; act RVIT_ACT_*;
 *   Cq / SCq (optional, N % 128 == 0): MX-fp8 copy of the result (values + tiled scales) for the next GEMM. */
int rvit_gemm_mxfp8(const void* A, int64_t lda, const void* SA, const void* B, int64_t ldb, const void* SB, void* C,
                    int64_t ldc, int c_dtype, int64_t M, int N, int K, const float* bias, const void* R, int64_t ldr,
                    int r_dtype, int act, void* Cq, int64_t ldcq, void* SCq, void* stream);

/* MXFP8 quantisation: x (bf16 / f32) [rows, K] -> e4m3 q [rows, K] + E8M0 scales (tiled as above,
 * ceil(rows/16) * ceil(K/512) * 256 bytes); K % 128 == 0; scale = 2^(floor(log2 max|x_block|) - 8), values rounded to
 * nearest even and saturated to +-448. */
int rvit_quantize_mxfp8(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t rows, int K,
                        void* stream);

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

/* rvit_conv2d_fwd_bf16 without scale / bias / residual / activation (the conv of a train-mode conv -> BatchNorm pair,
 * timm resnet in train(): train_vit.py:87,935-937) whose epilogue ALSO produces the BatchNorm batch statistics of the
 * map it stores (round 3): every M tile writes the per-channel sum and sum of squares of its bf16-ROUNDED outputs to
 * its own slot, stat_ws[slot * 2 Cout + c] and [slot * 2 Cout + Cout + c] (fp32, a fixed association inside the tile, no
 * atomics); rvit_bn_stats_finalize then adds the slots in order into `sums` - the values rvit_bn_stats would compute from
 * the stored map, without the pass over it (104 us for the 411 MB map of the stem's first conv at B = 256).
 *   *stat_slots receives the number of slots the chosen kernel writes; 0 = this kernel has no statistics epilogue (or the
 *   slots do not fit stat_ws_bytes): the conv has run as rvit_conv2d_fwd_bf16 would and the caller falls back to
 *   rvit_bn_stats.  stat_ws 16-byte aligned.
 *   rvit_bn_stats_finalize: sums[0..2C) += slots; stage: scratch of >= 256 * 2C floats, needed above 512 slots. */
int rvit_conv2d_fwd_stats_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cpix, int seg_len, int KH,
                               int KW, int stride, int pad, int Ho, int Wo, int Cout, void* stat_ws, int64_t stat_ws_bytes,
                               int* stat_slots, int tile, void* stream);
int rvit_bn_stats_finalize(const void* stat_ws, int stat_slots, int C, float* sums, void* stage, void* stream);

/* The stem's first conv in a training step (timm resnet18 `conv1` 7x7 / stride 2 / pad 3, 3 -> 64, under
 * train_vit.py:87,145 in train()): y bf16 [B,112,112,64] = conv(x) read straight from the fp32 NCHW input
 * [B,3,224,224] (rounded to bf16 as rvit_input_pack_nhwc4 would), with the batch statistics of the stored map as
 * rvit_conv2d_fwd_stats_bf16 gives them: *stat_slots slots of 128 floats (sum[64] | sum of squares[64]) in stat_ws, to be
 * added by rvit_bn_stats_finalize(stat_ws, *stat_slots, 64, ...).  w: bf16 [64, 224], k = kh*32 + px*4 + ch (the layout
 * rvit_conv2d_fwd_bf16 takes for this conv with seg_len 32).  Same products in the same kh order as that call on the
 * packed image; one workgroup walks down an image (or a part of it), every input row is fetched once. */
int rvit_stem_conv1_stats_f32_fwd(const float* x, const void* w, void* y, void* stat_ws, int64_t stat_ws_bytes,
                                  int* stat_slots, int B, void* stream);

/* Weight gradient of that conv (autograd of `conv1` in loss.backward(), train_vit.py:949-955), from the same fp32 input:
 *   dw[co][kh*32 + px*4 + ch] += sum over images and output pixels (r, c) of dy[b][r][c][co] * bf16(x[b][ch][2r+kh-3][2c+px-3])
 * i.e. what rvit_conv2d_wgrad_bf16 gives on the packed image with seg_len 32 (the columns px = 7 and ch = 3 are not
 * weights of the conv: px = 7 holds the products with the neighbouring pixel, ch = 3 zeros, as there), without the packed
 * image being read 2 x 7 times.
 *   dy: bf16 [B,112,112,64]; dw: fp32 [64, 224]; workspace: the ordered-reduction scratch (required: one slot of 64 x 224
 *   floats per workgroup, >= B slots; added in slot order, so the result is the same on every run). */
int rvit_stem_conv1_wgrad_f32(const float* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes, int B,
                              void* stream);

/* Which kernel rvit_conv2d_fwd_bf16 / rvit_conv2d_dgrad_bf16 run (tile 0) for a 3x3 / stride 1 / pad 1 convolution of B
 * images of H x W x Cin -> Cout (16-byte aligned operands): 0 implicit-GEMM gather kernel, 1 generic direct kernel
 * (conv3x3.hip), 2 weights-resident 64 -> 64 kernel (conv3x3_c64.hip), 3 its 256-channel sixteen-wave variant, 4 the
 * eight-wave staggered kernel (conv3x3_p8.hip, chosen where its tiles fill whole rounds of the CUs).  Every choice gives the
 * same bits; the query exists so that profilers can label launches (torch has no counterpart: cudnn picks silently,
 * train_vit.py:711-714 `cudnn.benchmark`). */
int rvit_conv3x3_kernel(int B, int H, int W, int Cin, int Cout);

/* Input gradient of the convolution above (autograd of the stem convs): dx = conv_transpose(dy, w)
 * (+ residual, the gradient arriving over the other branch of a BasicBlock).
 *   dy: bf16 [B,Hy,Wy,Cout]; w_t: bf16 [Cin, KH, KW, Cout]; dx/residual: bf16 [B,Hx,Wx,Cin];
 *   stride 1 or 2; Cout % 64 == 0. */
int rvit_conv2d_dgrad_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int KH, int KW,
                           int stride, int pad, int Hx, int Wx, int Cin, const void* residual, int tile,
                           void* stream);

/* rvit_conv2d_dgrad_bf16 whose output dx is the gradient entering a train-mode BatchNorm (+ ReLU): the conv -> norm ->
 * ReLU chains of timm's BasicBlock under loss.backward() (train_vit.py:87,949-955).  Where the chosen kernel has the
 * hook (the direct 3x3 / stride 1 kernels), its epilogue also writes that norm's two backward sums over the rows of each
 * tile - what rvit_bn_bwd_reduce computes from dx, bn_y and the ReLU bits in a pass of its own - to per-tile slots in
 * stat_ws (layout of rvit_conv2d_fwd_stats_bf16: sum d | rstd * sum d (y - mean) per channel, d = the stored bf16
 * gradient where the ReLU bit is set); rvit_bn_stats_finalize(stat_ws, *stat_slots, Cin, bwd_sums, stage) then adds them
 * in order and rvit_bn_bwd_apply follows.  *stat_slots == 0: no hook for this geometry, dx is complete, the caller runs
 * rvit_bn_bwd_reduce.
 *   bn_y: the norm's input map, bf16 [B,Hx,Wx,Cin]; bn_relu_mask: the bits rvit_bn_apply wrote (one byte per 8 channels)
 *   or NULL (no ReLU); bn_fwd_sums: fp32 [2 Cin] forward sums of that norm; eps: its epsilon. */
int rvit_conv2d_dgrad_bnr_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int KH, int KW,
                               int stride, int pad, int Hx, int Wx, int Cin, const void* residual, const void* bn_y,
                               const void* bn_relu_mask, const float* bn_fwd_sums, float eps, void* stat_ws,
                               int64_t stat_ws_bytes, int* stat_slots, void* stream);

/* Input gradient of a 3x3 / stride 2 / pad 1 conv (the first conv of resnet layers 2-4, train_vit.py:87 through
 * autograd), decomposed by output parity into four dense stride-1 convs over dY (9 taps of work instead of 36):
 * dy [B, Hy, Wy, Cout] -> dx [B, 2*Hy, 2*Wy, Cin] (+ residual of the same shape).  w_s2: bf16 [Cin, 9*Cout], block p
 * of the K axis = transposed tap RVIT_S2_TAP_ORDER[p] = {4, 5, 3, 7, 1, 8, 6, 2, 0}[p] of the [Cout, 3, 3, Cin]
 * weight (tap = 3*kh + kw). */
int rvit_conv2d_dgrad_s2_bf16(const void* dy, const void* w_s2, void* dx, int B, int Hy, int Wy, int Cout, int Cin,
                              const void* residual, void* stream);

/* Input gradient of the 1x1 / stride 2 downsample conv, accumulated: dx[b, 2i, 2j, :] += dy[b, i, j, :] . w_t^T
 * (w_t bf16 [Cin, Cout]); the other three quarters of dx are left as they are. */
int rvit_conv1x1_dgrad_s2_acc_bf16(const void* dy, const void* w_t, void* dx, int B, int Hy, int Wy, int Cout, int Cin,
                                   void* stream);

/* Patch-embedding convolution straight into the token stream (pure-ViT `patch16` layout; the
 * same flatten/transpose/cls-offset/pos_embed epilogue as the 1x1 case, train_vit.py:149-159):
 * tokens[b, 1+p, :] = conv(x, w)[b, p, :] + bias + pos_embed[1+p, :], fp32 [B, Ho*Wo+1, Cout].
 * Row 0 of every sample (cls) is left untouched (rvit_embed_cls writes it).  drop_p > 0 applies
 * pos_drop (train_vit.py:160) with element index = token row * Cout + column. */
int rvit_conv2d_tokens_fwd(const void* x, const void* w, float* tokens, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, const float* bias,
                           const float* pos_embed, float drop_p, uint64_t drop_seed, int tile, void* stream);

/* Ordered (bit-reproducible) reductions.  The reference asks its backend for deterministic kernels
 * (train_vit.py:50).  The entry points below that take `workspace` / `workspace_bytes` end a reduction that is split
 * over workgroups either with fp32 atomics (workspace == NULL: arrival order, last bits differ from run to run) or,
 * given a caller-owned device workspace, by storing one partial per split and adding the partials in ascending split
 * order in a second kernel: same bits on every run.  The workspace is scratch (no state survives the call), 16-byte
 * aligned; calls that share it must be ordered on one stream.  Weight gradients size their split count to what it
 * holds (one split needs N_store * K * 4 bytes, two for the variants where two wave groups share a tile); the
 * column-sum entries need at most 4 MiB.  RVIT_ORDERED_WORKSPACE_BYTES serves every layer of the models built here. */
#define RVIT_ORDERED_WORKSPACE_BYTES (192u << 20)

/* Weight gradient of a Linear: dW[N_store,K] += dY[M,N]^T . X[M,K] (accumulates like autograd's .grad; fp32 atomics,
 * or ordered with a workspace).  dY: bf16 [.,ldy], X: bf16 [.,ldx], optional row remaps; N, K, ldy, ldx
 * multiples of 8; rows n >= N_store of the product are discarded (padded logits).
 * splits: number of M partitions (0 = automatic; -3 / -4 force / forbid the sixteen-wave 256x256-tile variant that
 * long plain products run, for tests). */
int rvit_wgrad_bf16(const void* dY, int64_t ldy, int y_row_group, int y_row_extra, const void* X, int64_t ldx,
                    int x_row_group, int x_row_extra, float* dW, int64_t ldw, int64_t M, int N, int N_store, int K,
                    int splits, void* workspace, int64_t workspace_bytes, void* stream);

/* Weight AND bias gradient of a Linear in one launch: dW[N_store,K] += dY^T . X as rvit_wgrad_bf16 (no row remaps) and
 * dbias[N_store] += column sums of dY (autograd of the encoder's nn.Linear layers, train_vit.py:141-171, run by
 * train_vit.py:939).  For N > 64 the sums ride the matrix pipe of the workgroups that own the first K tile (one extra
 * MFMA per four); for N <= 64 (the classifier head) it is the weight-gradient kernel followed by rvit_colsum_bf16. */
int rvit_linear_wgrad_bias_bf16(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw,
                                float* dbias, int64_t M, int N, int N_store, int K, int splits, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* Weight gradient of the NHWC convolution: dW[Cout, KH*KW*seg_len] += dy^T . gather(x), same
 * geometry arguments and weight layout as rvit_conv2d_fwd_bf16. */
int rvit_conv2d_wgrad_bf16(const void* dy, const void* x, float* dW, int B, int H, int W, int Cpix, int seg_len,
                           int KH, int KW, int stride, int pad, int Ho, int Wo, int Cout, int splits, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* MobileNetV3 stem pieces (the reference's backbone='mobilenetv3' = timm mobilenetv3_small_100 features_only,
 * train_vit.py:68-74,87,145-146; inference), NHWC bf16 maps, C % 8 == 0.  Pointwise convolutions and the squeeze-excite
 * projections are rvit_gemm_bf16 over the same maps.
 *   rvit_dwconv_fwd_bf16: depthwise K x K (K 3 or 5, stride 1 or 2) + folded BatchNorm (scale, bias) + activation;
 *     w: bf16 [K*K, C] (tap-major); out [B, Ho, Wo, C].
 *   rvit_avgpool_nhwc: out[b, c] = mean over the HW pixels (bf16 [B, C]).
 *   rvit_channel_scale_nhwc: x[b, p, c] *= gate[b, c] in place (bf16 gate [B, C]).
 *   rvit_squeeze_excite_bf16: the whole squeeze-excite block (timm SqueezeExcite: mean over HW, conv_reduce + ReLU,
 *     conv_expand + hard-sigmoid, multiply) in one launch and in place, one image's map resident in LDS: x [B, HW, C],
 *     w1 bf16 [RD, C], b1 f32 [RD], w2 bf16 [C, RD], b2 f32 [C]; RD % 8 == 0.  Rejects maps whose LDS need
 *     (2*HW*C + 4*(512/(C/8)*C + 2*C + RD) bytes) exceeds RVIT_SE_MAX_LDS_BYTES: use the three calls above then.
 *   rvit_pointwise_conv_bf16: 1x1 convolution + folded BatchNorm + residual + activation over the map viewed as
 *     x [M, K] (M = B*H*W): out[m, n] = act(scale[n] * sum_k x[m, k] w[n, k] + bias[n] + residual[m, n]); w bf16 [N, K],
 *     scale / bias / residual may be NULL.  For the small K and N of these stems (K, N % 8 == 0, K <= 256, the weights
 *     as MFMA operands within RVIT_POINTWISE_MAX_LDS_BYTES: rvit_pointwise_conv_supported says); same result as
 *     rvit_gemm_bf16 with the same epilogue up to summation order within one MFMA.
 *   rvit_conv3x3s2_rgb_f32_fwd: conv_stem + bn1 + activation straight from the model's input: 3x3 / stride 2 / pad 1,
 *     3 -> 16 channels, x_nchw f32 [B, 3, H, W] (H, W even), w f32 [27, 16] with k = (kh*3 + kw)*3 + ci,
 *     out bf16 [B, H/2, W/2, 16]. */
#define RVIT_SE_MAX_LDS_BYTES (160 * 1024)
#define RVIT_POINTWISE_MAX_LDS_BYTES (64 * 1024)
int rvit_dwconv_fwd_bf16(const void* x, const void* w, const float* scale, const float* bias, void* out, int B, int H,
                         int W, int C, int K, int stride, int pad, int act, void* stream);
int rvit_avgpool_nhwc(const void* x, void* out, int B, int HW, int C, void* stream);
int rvit_channel_scale_nhwc(void* x, const void* gate, int B, int HW, int C, void* stream);
int rvit_squeeze_excite_bf16(void* x, const void* w1, const float* b1, const void* w2, const float* b2, int B, int HW, int C,
                             int RD, void* stream);
int rvit_pointwise_conv_bf16(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                             void* out, int64_t M, int K, int N, int act, void* stream);
int rvit_pointwise_conv_supported(int K, int N);
int rvit_conv3x3s2_rgb_f32_fwd(const float* x_nchw, const float* w, const float* scale, const float* bias, void* out, int B,
                               int H, int W, int Cout, int act, void* stream);

/* Training pass of the MobileNetV3 stem (model.train() forward and its autograd, train_vit.py:145-146,939).
 *   rvit_bn_generic_{stats,apply,bwd}: train-mode nn.BatchNorm2d on an NHWC bf16 map [M, C] of any C % 8 == 0
 *     (rvit_bn_* needs the resnet widths), fused with the activation that follows it in timm's BatchNormAct2d
 *     (RVIT_ACT_NONE / RELU / HSWISH) and an optional residual:
 *       stats: sums[0:C] += sum_m y, sums[C:2C] += sum_m y^2;
 *       apply: z = act(gamma * (y - mean) * rstd + beta + residual), running statistics updated when given;
 *       bwd:   g = dz * act'(u) with u recomputed from y; bsums += (sum g, sum g * xhat);
 *              dy = gamma * rstd * (g - bsums1 / M - xhat * bsums2 / M); dgamma += bsums2; dbeta += bsums1.
 *     sums / bsums are zeroed by the caller.  The reductions are ordered (bit-reproducible): `workspace` is required.
 *   rvit_dwconv_dgrad_bf16 / rvit_dwconv_wgrad_bf16: gradients of the depthwise convolution of rvit_dwconv_fwd_bf16
 *     (same geometry arguments; H, W are the INPUT map's): dx [B, H, W, C]; dw f32 [C, K*K] += (the parameter's own
 *     layout [C, 1, K, K]); ordered, `workspace` required.
 *   rvit_channel_affine_nhwc: out[b, p, c] = x[b, p, c] * gate[b, c] + add[b, c] (add may be NULL, out may be x): the
 *     squeeze-excite multiply with a separate output, and its backward.
 *   rvit_channel_dot_nhwc: out[b, c] = (sum_p a[b, p, c] * x[b, p, c]) * hardsigmoid'(gate[b, c]) (gate may be NULL): the
 *     gradient of the squeeze-excite gate's pre-activation, bf16 [B, C]. */
int rvit_bn_generic_stats(const void* y, int64_t M, int C, float* sums, void* workspace, int64_t workspace_bytes, void* stream);
int rvit_bn_generic_apply(const void* y, const float* sums, const float* gamma, const float* beta, const void* residual,
                          void* z, int64_t M, int C, float eps, int act, float momentum, float* running_mean,
                          float* running_var, void* stream);
int rvit_bn_generic_bwd(const void* dz, const void* y, const float* fsums, float* bsums, const float* gamma, const float* beta,
                        void* dy, float* dgamma, float* dbeta, int64_t M, int C, float eps, int act, void* workspace,
                        int64_t workspace_bytes, void* stream);
int rvit_dwconv_dgrad_bf16(const void* dy, const void* w, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                           void* stream);
int rvit_dwconv_wgrad_bf16(const void* dy, const void* x, float* dw, int B, int H, int W, int C, int K, int stride, int pad,
                           void* workspace, int64_t workspace_bytes, void* stream);
int rvit_channel_affine_nhwc(const void* x, const void* gate, const void* add, void* out, int B, int HW, int C, void* stream);
int rvit_channel_dot_nhwc(const void* a, const void* x, const void* gate, void* out, int B, int HW, int C, void* stream);

/* Fused stem entry for inference: conv 7x7/2 pad 3 (3->64) + folded BatchNorm (scale, bias) + ReLU +
 * max-pool 3x3/2 pad 1 in one kernel; the 112x112x64 map never reaches HBM.  Replaces timm resnet
 * `conv1`/`bn1`/`act1`/`maxpool` under `self.cnn_backbone(x)[-1]` (train_vit.py:87,145-146) in eval mode.
 *   img: bf16 [B,230,232,4] from rvit_input_pack_nhwc4 (pad_top = pad_left = 3, 224x224 image)
 *   w:   bf16 [64,232], k = kh*32 + px*4 + ch for kh<7, px<7, ch<3, every other entry zero
 *   scale/bias: f32 [64];  pooled: bf16 [B,56,56,64];  relu != 0 applies the ReLU. */
int rvit_stem_conv1_pool_fwd(const void* img, const void* w, const float* scale, const float* bias, void* pooled, int B,
                             int relu, void* stream);

/* The same stem entry straight from the model's input tensor: x_nchw f32 [B,3,224,224] contiguous, 16-byte aligned
 * (CNNViTHybrid.forward's argument, train_vit.py:141-146).  The bf16 NHWC4 patch is built in LDS by the kernel itself,
 * so the separate rvit_input_pack_nhwc4 pass and its 427 KB/sample buffer are not needed; results are bit-identical
 * to rvit_input_pack_nhwc4 + rvit_stem_conv1_pool_fwd.  What eval-mode inference runs. */
int rvit_stem_conv1_pool_f32_fwd(const float* x_nchw, const void* w, const float* scale, const float* bias, void* pooled,
                                 int B, int relu, void* stream);

/* fp32 NCHW image batch -> zero-haloed bf16 NHWC4 (channel 3 = 0).  Input contract of
 * CNNViTHybrid.forward (train_vit.py:141-146: x is [B,3,224,224] float).  Writes only the
 * interior; the caller zeroes the buffer once. */
int rvit_input_pack_nhwc4(const float* x_nchw, void* out, int B, int C, int H, int W, int Hp, int Wp, int pad_top,
                          int pad_left, void* stream);

/* 3x3 / stride 2 / pad 1 max-pool on NHWC bf16 (resnet stem max-pool, train_vit.py:87).
 * argmax (optional, uint8 same shape as y) records the winning tap for the backward pass. */
int rvit_maxpool3x3s2_nhwc(const void* x, void* y, void* argmax, int B, int H, int W, int C, void* stream);

/* x[b,0,:] = drop(cls_token + pos_embed[0]) on the fp32 residual stream [B,S,E]
 * (train_vit.py:155-160); same dropout site and element indexing as the patch rows. */
int rvit_embed_cls(const float* cls_token, const float* pos_embed, float* x, int B, int S, int E, float drop_p,
                   uint64_t drop_seed, void* stream);

/* out = in * mask(p, seed) / (1-p), element index = row * cols + col; bf16 or f32, in place allowed.
 * Backward of the dropout sites whose forward ran in a GEMM epilogue before a residual add
 * (dropout1 / dropout2, torch transformer.py:977,982) and of pos_drop (train_vit.py:160). */
int rvit_dropout_apply(const void* in, void* out, int dtype, int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                       float p, uint64_t seed, void* stream);

/* out[i] = 1 if element i of site (p, seed) is kept, else 0.  Test/debug aid: lets a host
 * reference replay the masks the kernels used.  The generator (every drop_seed of this header): key words = halves of
 * splitmix64(seed), so any distinct seeds are independent sites; one 32-bit integer hash of (key, i >> 1) decides
 * elements 2j and 2j + 1 (low / high 16 bits >= round(p * 2^16) keeps).  oracle/ref_dropout.py restates it in NumPy and
 * this entry must equal it bit for bit. */
int rvit_dropout_mask(uint8_t* out, int64_t n, float p, uint64_t seed, void* stream);

/* LayerNorm over the last axis, eps inside the sqrt, affine (torch transformer.py:774,947-950;
 * train_vit.py:119,166).  x: bf16/f32 [rows, ldx]; y: bf16/f32 [rows, ldy]; mean/rstd optional
 * fp32 [rows] (saved for backward). */
int rvit_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y,
                       int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows, int E, float eps,
                       void* stream);

/* LayerNorm whose output is MXFP8, the operand format of rvit_gemm_mxfp8 (inference, BASELINE configs[4]): the
 * bytes equal rvit_layernorm_fwd (bf16 output) followed by rvit_quantize_mxfp8, in one pass.  E % 128 == 0;
 * q: e4m3 [rows, ldq]; scales: E8M0, tiled as for rvit_quantize_mxfp8. */
int rvit_layernorm_mxfp8_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                             uint8_t* q, int64_t ldq, uint8_t* scales, int64_t rows, int E, float eps, void* stream);

/* Feed-forward half of a pre-LN encoder layer for inference, in place on the fp32 residual stream:
 *   x += linear2(gelu(linear1(LayerNorm(x))))     (torch transformer.py:949-950, 980-982; erf GELU)
 * in ONE kernel: a workgroup keeps its tokens' LayerNorm output and GELU map in LDS, so the hidden map never
 * reaches HBM.  Same bits as rvit_layernorm_fwd -> rvit_gemm_bf16(gelu) -> rvit_gemm_bf16(residual).
 *   x: f32 [T, E], E == 256; gamma/beta: f32 [E]; w1: bf16 [hidden, E], b1: f32 [hidden]; w2: bf16 [E, hidden],
 *   b2: f32 [E]; hidden % 256 == 0.  rows_per_wg: tokens per workgroup, 0 = chosen from T (<= 112). */
int rvit_encoder_mlp_fwd(float* x, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                         const void* w2, const float* b2, int64_t T, int E, int hidden, int rows_per_wg,
                         void* stream);

/* Self-attention half of a pre-LN encoder layer for inference, in place on the fp32 residual stream:
 *   x += out_proj(softmax(q k^T / 8) v),  q | k | v = in_proj(LayerNorm(x))
 * (torch transformer.py:947-948; functional.py:5819-5833, 6569-6637), in ONE kernel: a workgroup owns two images,
 * q, k, v, the probabilities and the attention output never leave the CU.
 *   x: f32 [B*S, E], E == 256, H == 4 (head_dim 64), S <= 64; gamma/beta: f32 [E];
 *   wqkv: bf16 [3E, E] (in_proj_weight), bqkv: f32 [3E]; wout: bf16 [E, E], bout: f32 [E]. */
int rvit_encoder_attn_fwd(float* x, const float* gamma, const float* beta, float eps, const void* wqkv,
                          const float* bqkv, const void* wout, const float* bout, int B, int S, int E, int H,
                          void* stream);

/* The whole pre-LN encoder stack for inference in ONE launch, in place on the fp32 residual stream: for every layer
 *   x += out_proj(softmax(q k^T / 8) v), q | k | v = in_proj(LayerNorm(x));   x += linear2(gelu(linear1(LayerNorm(x))))
 * (`self.transformer(x)`, train_vit.py:112-116,163; torch transformer.py:539-553 the loop, :945-950 the layer).  A
 * workgroup owns two whole images from the first LayerNorm to the last residual add (attention mixes the tokens of one
 * image only), so no workgroup waits for another and the layers need no kernel boundary between them.  The arithmetic is
 * that of rvit_encoder_attn_fwd / rvit_encoder_mlp_fwd: same bits as calling those per layer.
 *   x: f32 [B*S, E], E == 256, H == 4, 2 S <= 112, hidden % 256 == 0, n_layers <= 16, one eps for every LayerNorm;
 *   layer_params: HOST array of 12 device pointers per layer: norm1.weight, norm1.bias (f32 [E]), in_proj_weight (bf16
 *   [3E, E]), in_proj_bias (f32 [3E]), out_proj.weight (bf16 [E, E]), out_proj.bias (f32 [E]), norm2.weight, norm2.bias,
 *   linear1.weight (bf16 [hidden, E]), linear1.bias (f32 [hidden]), linear2.weight (bf16 [E, hidden]), linear2.bias. */
int rvit_encoder_stack_fwd(float* x, const void* const* layer_params, int n_layers, float eps, int B, int S, int E, int H,
                           int hidden, void* stream);

/* LayerNorm backward fused with the residual-gradient accumulation of the pre-LN block
 * (x + f(LN(x)), torch transformer.py:947-950):  dx (+)= LN'(dy); dgamma += ..., dbeta += ...
 *   dy: bf16 [rows,lddy]; x: f32 [rows,ldx] (the LN input); mean/rstd from the forward;
 *   dx: f32 [rows,lddx], accumulate != 0 adds to its content; dx_bf16: optional bf16 copy of the
 *   updated dx; dgamma/dbeta: f32 [E], accumulated (atomics, or ordered with a workspace). */
int rvit_layernorm_bwd(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                       const float* rstd, const float* gamma, float* dx, int64_t lddx, int accumulate,
                       void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta, int64_t rows, int E, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* rvit_layernorm_bwd whose bf16 copy additionally carries the dropout of the site that consumes it: dx_bf16 =
 * bf16(dx) * mask(drop_p, drop_seed) / (1 - p), element index = row * E + col (the indexing of rvit_dropout_apply and of
 * the GEMM epilogues).  In the pre-LN block the gradient that reaches out_proj / linear2 is the residual gradient
 * times the mask of dropout1 / dropout2 (torch transformer.py:977,982), so the separate rvit_dropout_apply pass over
 * that copy disappears; dx (fp32) is not masked.  drop_p = 0 is rvit_layernorm_bwd; drop_p > 0 needs dx_bf16. */
int rvit_layernorm_bwd_drop(const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                            const float* rstd, const float* gamma, float* dx, int64_t lddx, int accumulate,
                            void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta, int64_t rows, int E,
                            void* workspace, int64_t workspace_bytes, float drop_p, uint64_t drop_seed, void* stream);

/* softmax(q k^T * scale) v per (batch, head), head_dim 64, S <= 256, no mask
 * (torch functional.py:6623-6631 -> scaled_dot_product_attention).
 *   qkv: bf16 [B*S, ldq] holding q | k | v column blocks of width H*64 (the in-proj output);
 *   out: bf16 [B*S, ldo], head h at columns h*64..; lse (optional) fp32 [B,H,S].
 *   drop_p > 0: dropout on the normalised probabilities (functional.py:6569-6570), element index =
 *   ((b*H + h)*S + query)*S2 + key with the row stride S2 = S rounded up to even (keys 2j, 2j+1 of a query share
 *   one hash of the generator; rvit_dropout_mask(n = B*H*S*S2) viewed [B,H,S,S2] replays the site). */
int rvit_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                  float drop_p, uint64_t drop_seed, void* stream);

/* rvit_attn_fwd (no dropout, no lse) for S <= 64 with the arithmetic of rvit_encoder_attn_fwd's attention core: the
 * per-op sequence rvit_layernorm_fwd -> rvit_gemm_bf16 -> rvit_attn_fwd_s64 -> rvit_gemm_bf16(residual) gives the SAME
 * BITS as rvit_encoder_attn_fwd, so a caller may choose between them by batch size (the fused kernel needs a few
 * hundred images to fill the chip) without a frame's result depending on the batch it travels in. */
int rvit_attn_fwd_s64(const void* qkv, void* out, int B, int S, int H, int64_t ldq, int64_t ldo, float scale,
                      void* stream);

/* rvit_attn_fwd for the MX-fp8 inference mode (BASELINE.json configs[4]): both products run on the fp8 MFMA and the
 * result is emitted as the MXFP8 operand of the out_proj GEMM (no bf16 attention map, no quantisation pass).
 *   qkv: bf16 [B*S, ldq] as for rvit_attn_fwd; q and k are quantised to e4m3 with one power-of-two scale per
 *   (token, head) row, v with one per (image, head), the probabilities as e4m3(256 p);
 *   out_q: e4m3 [B*S, ldo]; out_s: E8M0 scales of a [B*S, H*64] operand, tiled as for rvit_quantize_mxfp8
 *   (H even: K % 128 == 0).  The reference has no fp8 path; oracle/ref_mxfp8.py restates this arithmetic. */
int rvit_attn_fwd_fp8(const void* qkv, uint8_t* out_q, int64_t ldo, uint8_t* out_s, int B, int S, int H, int64_t ldq,
                      float scale, void* stream);

/* The MX-fp8 inference mode's default attention since round 3: rvit_attn_fwd (bf16 MFMA, torch functional.py:6623-6631)
 * with the output row rounded to bf16 and written as the MXFP8 operand of the out_proj GEMM: the same bytes as
 * rvit_attn_fwd followed by rvit_quantize_mxfp8, without the bf16 attention map and the quantisation pass.  (Un-scaled
 * fp8 MFMAs run at the bf16 rate on gfx950 and rvit_attn_fwd_fp8 pays for quantising q/k/v/p on the vector ALUs: 185
 * against 140 us per ViT-L layer at B = 256.)  Arguments as rvit_attn_fwd_fp8. */
int rvit_attn_fwd_mx(const void* qkv, uint8_t* out_q, int64_t ldo, uint8_t* out_s, int B, int S, int H, int64_t ldq,
                     float scale, void* stream);

/* Backward of rvit_attn_fwd: dqkv (same layout as qkv) from qkv, out, dout and the saved lse;
 * (drop_p, drop_seed) must be the forward's. */
int rvit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S,
                  int H, int64_t ldq, int64_t ldo, int64_t lddo, int64_t lddq, float scale, float drop_p,
                  uint64_t drop_seed, void* stream);

/* The classifier tail in one launch (train_vit.py:127-131, 166-185: `x = self.norm(x[:, 0]); return self.head(x)` with
 * head = Linear(E, E/2) -> GELU -> Dropout -> Linear(E/2, classes), eval mode):
 *   logits[b] = w3 . gelu(w0 . LayerNorm(x[b]) + b0) + b3
 * x: fp32 rows of pitch ldx (the class-token rows of the token stream: ldx = S*E), w0 bf16 [hidden, E], w3 bf16
 * [num_classes, hidden], logits fp32 [B, ldl].  The same bits as rvit_layernorm_fwd (bf16 out) -> rvit_gemm_bf16 (bias,
 * GELU, bf16 out) -> rvit_gemm_bf16 (bias, fp32 out).  E and hidden multiples of 32 up to 1024
 * (rvit_cls_head_supported says so; other shapes keep the three calls). */
int rvit_cls_head_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w0,
                      const float* b0, const void* w3, const float* b3, float* logits, int64_t ldl, int B, int E,
                      int hidden, int num_classes, void* stream);
int rvit_cls_head_supported(int E, int hidden, int num_classes);

/* rvit_attn_fwd / rvit_attn_bwd for the other head dims the reference's constructor accepts (train_vit.py:76-111 builds
 * nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads) for any embed_dim % num_heads == 0; torch
 * functional.py:6623-6631): head_dim in {16, 32, 48, 96, 128}, S <= 256.  Same layouts with column blocks of width
 * H*head_dim, same dropout site indexing (rvit_dropout_mask replays it), same lse.  `scale` is given by the caller
 * (torch: head_dim^-0.5) and must be positive.  head_dim 64 stays on rvit_attn_fwd / rvit_attn_bwd. */
int rvit_attn_dh_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int head_dim, int64_t ldq, int64_t ldo,
                     float scale, float drop_p, uint64_t drop_seed, void* stream);
int rvit_attn_dh_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S,
                     int H, int head_dim, int64_t ldq, int64_t ldo, int64_t lddo, int64_t lddq, float scale, float drop_p,
                     uint64_t drop_seed, void* stream);

/* Cross-entropy with label smoothing, mean over the batch, value and gradient in one pass
 * (nn.CrossEntropyLoss(label_smoothing=...), train_vit.py:894,937).
 *   logits f32 [B,ldl]; labels int64 [B]; *loss_sum += mean loss (caller zeroes it);
 *   dlogits (optional) f32 [B,lddl]; dlogits_bf16 (optional) bf16 [B,lddb] with columns C..C_pad-1
 *   zero-filled (operand of the head's backward GEMMs); both scaled by grad_scale. */
int rvit_ce_loss(const float* logits, int64_t ldl, const int64_t* labels, float* loss_sum, float* dlogits,
                 int64_t lddl, void* dlogits_bf16, int64_t lddb, int B, int C, int C_pad, float label_smoothing,
                 float grad_scale, void* stream);

/* out[n] += sum_m a[m,n]  (bias gradients); a bf16 [.,lda] with optional row remap; ordered with a workspace. */
int rvit_colsum_bf16(const void* a, int64_t lda, float* out, int64_t M, int N, int64_t row_group, int64_t row_extra,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* out[j] += sum_b x[b*ldb + j], j < n  (pos_embed / cls_token gradients: token gradients summed
 * over the batch; backward of the broadcast add at train_vit.py:155-159). */
int rvit_batch_sum_f32(const float* x, int64_t ldb, float* out, int B, int64_t n, void* stream);

/* Training-mode BatchNorm2d on NHWC bf16 maps viewed as [M = B*H*W, C] (timm resnet BatchNorm2d
 * in train(), train_vit.py:87; eps inside the sqrt, biased batch variance for normalisation,
 * unbiased for running_var, momentum update like torch).  C % 8 == 0, 256 % (C/8) == 0.
 *   rvit_bn_stats:  sums[0..C) += sum_m y, sums[C..2C) += sum_m y^2   (caller zeroes sums)
 *                   workspace == NULL: partial sums meet in float atomics (fastest; the last bits vary from run
 *                   to run).  workspace != NULL (RVIT_BN_WORKSPACE_BYTES, 4-byte aligned, its first word zero before
 *                   the first call; consecutive calls on one stream may share it): ORDERED reduction - partials per
 *                   workgroup, added in workgroup order by the workgroup that finishes last - the same bits on every
 *                   run, what `torch.backends.cudnn.deterministic = True` asks of the reference (train_vit.py:50).
 *                   rvit_bn_bwd_reduce takes the same argument.
 *   rvit_bn_apply:  z = relu?( gamma*(y-mean)*rstd + beta + residual ); updates running stats
 *                   when running_mean/var are given.  relu_mask (optional, relu only): uint8 [M, C/8],
 *                   bit k of byte (m, c/8) = z[m, 8*(c/8) + k] > 0; lets the backward kernels read one byte per
 *                   8 channels instead of z. */
#define RVIT_BN_WORKSPACE_BYTES ((64 + 262144) * 4)
int rvit_bn_stats(const void* y, int64_t M, int C, float* sums, void* workspace, void* stream);
int rvit_bn_apply(const void* y, const float* sums, const float* gamma, const float* beta, const void* residual,
                  void* z, int64_t M, int C, float eps, int relu, float momentum, float* running_mean,
                  float* running_var, void* relu_mask, void* stream);

/* Stem entry in train(): rvit_bn_apply with relu, followed by MaxPool2d(3, 2, 1), in one pass (timm resnet
 * bn1 -> act1 -> maxpool, train_vit.py:87): y [B, H, W, C] -> pooled [B, Ho, Wo, C] (Ho = (H-1)/2+1), argmax uint8
 * of the same shape (window tap 0..8, first maximum, as rvit_maxpool3x3s2_nhwc records it) and the relu_mask of the
 * normalised map, which itself is never written. */
int rvit_bn_apply_pool(const void* y, const float* sums, const float* gamma, const float* beta, void* pooled,
                       void* argmax, void* relu_mask, int B, int H, int W, int C, float eps, float momentum,
                       float* running_mean, float* running_var, void* stream);

/* Backward of the above (autograd of BatchNorm2d + ReLU + residual add):
 *   dzm = dz * (z > 0) when relu, else dz; relu = 1: z is the forward output (bf16 map), relu = 2: z is the
 *   relu_mask written by rvit_bn_apply
 *   rvit_bn_bwd_reduce: bwd_sums[0..C) += sum dzm, bwd_sums[C..2C) += sum dzm*xhat (caller zeroes)
 *   rvit_bn_bwd_apply:  dy = gamma*rstd*(dzm - mean(dzm) - xhat*mean(dzm*xhat)); dgamma += .., dbeta += ..;
 *                       dz_masked (optional) receives dzm = the gradient of the residual branch. */
int rvit_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* fwd_sums, int64_t M, int C,
                       float eps, int relu, float* bwd_sums, void* workspace, void* stream);
int rvit_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* fwd_sums, const float* bwd_sums,
                      const float* gamma, void* dy, void* dz_masked, float* dgamma, float* dbeta, int64_t M, int C,
                      float eps, int relu, void* stream);

/* rvit_bn_bwd_reduce / rvit_bn_bwd_apply (relu = 2) for a BatchNorm whose ReLU output went through rvit_bn_apply_pool
 * (the stem's bn1 -> relu -> 3x3 / stride-2 max-pool; timm resnet18 behind train_vit.py:87,145): dz is not read from
 * the [B,H,W,C] map rvit_maxpool3x3s2_bwd_nhwc would write but gathered from the pooled map's gradient dpool
 * [B,Ho,Wo,C] and the argmax taps.  The same bits as that call followed by the plain entries (same fp32 sum over the
 * windows, same bf16 rounding), without the map's round trip through HBM and without the launch.  B*H*W < 2^24. */
int rvit_bn_bwd_reduce_pool(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                            const float* fwd_sums, int B, int H, int W, int C, float eps, float* bwd_sums, void* workspace,
                            void* stream);
int rvit_bn_bwd_apply_pool(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                           const float* fwd_sums, const float* bwd_sums, const float* gamma, void* dy, float* dgamma,
                           float* dbeta, int B, int H, int W, int C, float eps, void* stream);

/* Backward of rvit_maxpool3x3s2_nhwc from the recorded argmax taps; dx bf16 [B,H,W,C]. */
int rvit_maxpool3x3s2_bwd_nhwc(const void* dpool, const void* argmax, void* dx, int B, int H, int W, int C,
                               void* stream);

/* *out += sum_i x[i]^2 (fp32; the caller zeroes *out).  Squared global gradient norm for
 * torch.nn.utils.clip_grad_norm_ (train_vit.py:942-943) over the flat gradient buffer; ordered with a workspace. */
int rvit_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* One AdamW step over flat fp32 buffers, torch.optim.AdamW semantics (train_vit.py:879-883,945: decoupled
 * weight decay, bias correction with `step` >= 1), fused with gradient clipping: when grad_norm_sq is not
 * null and max_grad_norm > 0 the gradients are scaled by min(1, max_grad_norm / (sqrt(*grad_norm_sq) + 1e-6))
 * exactly as clip_grad_norm_ does.  grads are read only.  params_bf16 (nullable, 8-byte aligned, n elements)
 * receives the updated parameters rounded to bf16: the operand copy the next forward multiplies with (the
 * reference keeps such copies implicitly under autocast, train_vit.py:407-417). */
int rvit_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm_sq,
                    float max_grad_norm, void* params_bf16, void* stream);

/* out[i] = bf16(x[i]) (round to nearest even) over a flat buffer: (re)builds the bf16 operand copy of the
 * parameters outside an optimiser step, e.g. after load_state_dict (train_vit.py:913-922, resume). */
int rvit_cast_bf16(const float* x, void* out, int64_t n, void* stream);

/* Transposes every matrix of a table in one launch: the K-major operand copies W^T that the data-gradient
 * GEMMs / convolutions of the backward pass (train_vit.py:939) multiply with, derived from the bf16 parameter
 * copy.  table: device int64 [n_entries][8] = {src_off, dst_off, rows, cols, src_ld, dst_ld, tile_start,
 * tiles_c} in elements; dst[dst_off + c*dst_ld + r] = src[src_off + r*src_ld + c]; rows, cols, offsets and
 * leading dimensions multiples of 8; entry e owns 64x64 tiles [tile_start, tile_start + tiles_r*tiles_c);
 * total_tiles = the sum over entries. */
int rvit_transpose_batched_bf16(const void* src, void* dst, const int64_t* table, int n_entries, int64_t total_tiles,
                                void* stream);

/* Input transforms of the reference on device, from decoded uint8 frames [B,H,W,C] (C = 1 luma, 3 RGB, 4 RGBA,
 * interleaved, 4-byte aligned base) to the fp32 NCHW tensor [B,3,out_side,out_side] the model takes:
 *   Grayscale(3) -> Resize((resized,resized)) [-> crop out_side x out_side at crop_tl[b] = (top,left)] -> ToTensor
 *   -> Normalize(ImageNet mean/std) [-> + noise_std * N(0,1)]
 * (train_vit.py:722-756 train_tf / eval_tf, 557-567 AddGaussianNoise; infer.py:346-353).  Integer part bit-exact
 * with Pillow: rgb2l, then the two-pass triangle resampling with 22-bit fixed-point coefficients.  hcoef [resized][hk]
 * / hbounds [resized][2] = (first tap, tap count) for W -> resized, vcoef / vbounds for H -> resized, as
 * Resample.c precompute_coeffs + normalize_coeffs_8bpc produce them (vit_amd/pipeline.py resample_tables).
 * crop_tl may be null when resized == out_side.  tmp: scratch uint8 [B][H][out_side] owned by the caller.
 * noise_std 0 disables the noise; otherwise element i gets a counter-based Gaussian keyed by (noise_seed, i). */
int rvit_preprocess_u8(const uint8_t* frames, int B, int H, int W, int C, const int32_t* hcoef, const int32_t* hbounds,
                       int hk, const int32_t* vcoef, const int32_t* vbounds, int vk, int resized, const int32_t* crop_tl,
                       int out_side, uint8_t* tmp, float noise_std, uint64_t noise_seed, float* out, void* stream);

/* rvit_preprocess_u8 for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its transforms run on the host, train_vit.py:722-756).
 * Same arguments and bit-identical output, without `tmp`: the uint8 intermediate [B][H][out_side] lives in a static
 * device array of RVIT_PREPROCESS_STATIC_MAX_BYTES inside the library (one copy per GPU the library runs on, held
 * until the process ends).  The call allocates and frees nothing and calls no allocating routine; the first call on
 * a device resolves the array's address on that device (hipGetSymbolAddress) - make it outside graph capture.
 *   - B * H * out_side > RVIT_PREPROCESS_STATIC_MAX_BYTES returns RVIT_EINVAL with a message (split the batch);
 *   - the array of the CURRENT device is used: `stream` must belong to it (checked), as must every pointer;
 *   - all calls on one GPU share the one array: calls that overlap on a GPU (two streams, two threads) race;
 *     calls on one stream are ordered and safe.  Nothing needs clearing between calls (pass 1 overwrites every
 *     byte pass 2 reads). */
#define RVIT_PREPROCESS_STATIC_MAX_BYTES (32u << 20)
int rvit_preprocess_u8_static(const uint8_t* frames, int B, int H, int W, int C, const int32_t* hcoef,
                              const int32_t* hbounds, int hk, const int32_t* vcoef, const int32_t* vbounds, int vk,
                              int resized, const int32_t* crop_tl, int out_side, float noise_std, uint64_t noise_seed,
                              float* out, void* stream);

/* One time step of nn.LSTM (batch_first, 1 or 2 directions: pass the *_r set or nulls), fp32, torch gate order
 * i,f,g,o: the temporal encoder of the sequence model (train_vit.py:188-229 TemporalEncoder, infer.py:186-227).
 * Call with step = 0 .. T-1 in order.  x [B,T,In]; weights as nn.LSTM stores them (w_ih [4H,In], w_hh [4H,H],
 * biases [4H]); out [B,T,ndir*H]: the forward direction writes time `step`, the reverse direction time
 * T-1-step, each reading its previous h from `out` (zero state at step 0); c [ndir,B,H] is the running cell
 * state.  gates [B,T,ndir,4H] (activated) and cseq [B,T,ndir,H] are saved for the backward pass when non-null
 * (both or neither).  H % 8 == 0, In % 4 == 0. */
int rvit_lstm_step_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* out,
                       float* c, float* gates, float* cseq, int B, int T, int In, int H, int step, void* stream);

/* Backward of one time step (autograd of nn.LSTM, train_vit.py:939); call with step = T-1 .. 0.  dout [B,T,ndir*H]
 * = gradient of `out`; w_hh_t = W_hh^T [H,4H] per direction; dgates [B,T,ndir,4H] receives the pre-activation
 * gradients of this step (and is read at the step processed before); dc [ndir,B,H] carries the cell-state
 * gradient.  Weight, bias and input gradients follow from dgates with plain GEMMs (rvit_wgrad_bf16 etc.). */
int rvit_lstm_step_bwd(const float* dout, const float* w_hh_t, const float* w_hh_t_r, const float* gates,
                       const float* cseq, float* dgates, float* dc, int B, int T, int H, int step, void* stream);

/* nn.GRU counterparts of the two LSTM entry points (TemporalEncoder model_type='gru', train_vit.py:207-211):
 * gate order r,z,n, weights [3H,In] / [3H,H].  saved [B,T,ndir,4H] = (r, z, n, W_hn h + b_hn), null in inference.
 * Backward: w_hh_t = W_hh^T [H,3H]; `out` is the forward's output sequence (h_prev); dgates [B,T,ndir,6H] =
 * input-side (dr,dz,dn) then hidden-side (dr,dz,dn*r) pre-activation gradients; dh_carry [ndir,B,H]. */
int rvit_gru_step_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r, float* out,
                      float* saved, int B, int T, int In, int H, int step, void* stream);
int rvit_gru_step_bwd(const float* dout, const float* w_hh_t, const float* w_hh_t_r, const float* saved, const float* out,
                      float* dgates, float* dh_carry, int B, int T, int H, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RVIT_H */
