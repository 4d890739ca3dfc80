/*
 * advstep_lcnn.h — C ABI of the LCNN max-feature-map kernels in libadvstep.so.
 *
 * They replace, inside the attacked LCNN's forward + input-backward pass, the eager chains of
 *   src/models/lcnn.py:76-95   MaxFeatureMap2D.forward:  inputs.view(N, 2, C, H, W).max(1)   (values + int64 indices;
 *                              backward = zeros + scatter_)
 *   src/models/lcnn.py:123,129,137,154   torch.nn.MaxPool2d((2, 2), (2, 2)) that follows four of the nine MFMs
 * which are pure HBM streaming over the largest activations of the model ((B, 64, 404, 80) f32 = 1.06 GB at
 * B = 128).  Selection is recorded in one byte per 4 outputs (MFM) / one byte per pooled output (MFM + pool)
 * instead of int64 indices, and the backward pass writes the (mostly zero) input gradient in one coalesced pass.
 *
 * Semantics are those of the ATen kernels being replaced, bit for bit, including ties and NaN:
 *   MFM     y = a if (isnan(a) || a >= b) else b,  a = x[n, c], b = x[n, c + C]      (lower index wins a tie)
 *   pool    scan (dh, dw) in row-major order, take v when (v > best || isnan(v)), best starts at -inf
 * Every forward entry point takes optional bn_mean / bn_invstd (C floats, both NULL or both set): the eval-mode
 * BatchNorm2d(affine=False) that follows the block in LCNN (lcnn.py:127,131,134,141,144,148), y = (y - mean[c]) * invstd[c]
 * with invstd = 1 / sqrt(running_var + eps), applied in the epilogue; the matching backward takes gscale (= invstd).
 * Conventions as in advstep.h: contiguous NCHW float32 device pointers, caller-owned outputs, stream-ordered,
 * status codes.
 */
#ifndef ADVSTEP_LCNN_H_
#define ADVSTEP_LCNN_H_

#include <stddef.h>
#include <stdint.h>

#include "advstep.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of the selection buffer for an MFM over x (N, 2C, HW): one byte per group of 4 consecutive outputs of a
 * sample, N * ceil(C*HW / 4). */
size_t advstep_mfm_sel_bytes(int64_t N, int64_t C, int64_t HW);

/* y (N, C, HW) = max-feature-map of x (N, 2C, HW) [+ bias (2C) when bias != NULL: the convolution's bias add
 * (one rounding per element, as ATen's separate add kernel) is folded into this pass]; sel receives, per group of 4
 * outputs, bit k = 1 when output k came from the second half of the channels. */
int advstep_mfm_forward_f32(const float *x, const float *bias, const float *bn_mean, const float *bn_invstd, float *y,
                            uint8_t *sel, int64_t N, int64_t C, int64_t HW, advstep_stream_t stream);

/* gx (N, 2C, HW) from gy (N, C, HW) [* gscale (C) when given]: the gradient goes to the selected half, the other
 * half gets 0. */
int advstep_mfm_backward_f32(const float *gy, const uint8_t *sel, const float *gscale, float *gx, int64_t N, int64_t C,
                             int64_t HW, advstep_stream_t stream);

/* y (N, C, H/2, W/2) = MaxPool2d(2, 2)(MFM(x)), x (N, 2C, H, W) (floor division: a trailing odd row / column is
 * dropped, as with ceil_mode=False); bias (2C) as above, may be NULL.  idx receives one byte per pooled output: bit 2 = second channel half,
 * bit 1 = dh, bit 0 = dw of the winning input. */
int advstep_mfm_pool2_forward_f32(const float *x, const float *bias, const float *bn_mean, const float *bn_invstd,
                                  float *y, uint8_t *idx, int64_t N, int64_t C, int64_t H, int64_t W,
                                  advstep_stream_t stream);

/* gx (N, 2C, H, W) from gy (N, C, H/2, W/2) and idx: every input position receives either the pooled gradient
 * (the winner) or 0 — including a trailing odd row / column. */
int advstep_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *gscale, float *gx, int64_t N,
                                   int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* ---- first block fused: Conv2d(1, 2C, (5,5), stride 1, padding 2) -> MaxFeatureMap2D -> MaxPool2d(2, 2) -----------
 * (src/models/lcnn.py:121-123).  With ONE input channel the convolution is not a GEMM (K = 25): it is a 1.06 GB
 * write whose only consumer is the max-feature-map + pool, so the three are one kernel and the conv output never
 * exists.  x (N, 1, H, W), weight (2C, 1, 5, 5), bias (2C) or NULL -> y (N, C, H/2, W/2) and one selection byte per
 * output (same encoding as advstep_mfm_pool2_forward_f32).  Convolution sums are formed tap by tap in row-major
 * order with fma (deterministic; within float rounding of MIOpen's result, not bit-equal to it). */
int advstep_conv5_mfm_pool2_forward_f32(const float *x, const float *weight, const float *bias, float *y, uint8_t *idx,
                                        int64_t N, int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* Input gradient of the block above: gx (N, 1, H, W) from gy (N, C, H/2, W/2), idx and the weights.  No atomics, fixed
 * summation order.  Round 5: for even W <= 128 a thread owns a pooled CELL, accumulates over the channels what its winner
 * sends to the 6x6 input window around it (the window, by channel and selection code, is read from a table in LDS) and
 * the 3x3 blocks of the windows meet in one exchange per tile; other shapes - and ADVSTEP_CONV0_BWD=gather - take the
 * kernel of rounds 1-4, where every 2x2 input patch gathers from the <= 9 pooled cells whose winner can reach it. */
int advstep_conv5_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *weight, float *gx, int64_t N,
                                         int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* ---- 1x1 blocks fused: Conv2d(Cin, 2C, (1,1)) -> MaxFeatureMap2D  (src/models/lcnn.py:125-126,132-133,139-140,146-147)
 * A skinny GEMM (K = Cin) run on the matrix cores (fp32-in / fp32-accumulate MFMA: exact f32).
 * x (N, Cin, P) with P = H*W, weight (2C, Cin), bias (2C) or NULL -> y (N, C, P); sel gets ONE bit per output, set
 * when the second channel half won, in the layout the kernels' lanes own them (round 6; opaque to callers, size from
 * advstep_conv1x1_mfm_sel_bytes, 4-byte aligned): [n][p / 32][p % 32][h] masks of 16 bits (C <= 32) or 32 bits, where
 * channel c sits in half h = (c >> 2) & 1 at bit 16 (c >> 5) + (c & 3) + 4 ((c & 31) >> 3) - one 128- / 256-byte store
 * per wave and 32-pixel tile, one load per lane in the backward.  (Rounds 1-5: [n][c][p / 32] ballot words, sixteen
 * partial-line stores per wave; -7 us of the first block's 67.)  The 2C-channel conv output is never written.  Cin must be one advstep_conv1x1_mfm_supported() accepts (32, 48, 64: LCNN's); C <= 64.
 * bn_mean / bn_invstd (C) or both NULL: the eval-mode BatchNorm2d(affine=False) that follows each of these blocks
 * (lcnn.py:127,134,141,148), y = (max - mean[c]) * invstd[c], folded into the epilogue. */
int advstep_conv1x1_mfm_supported(int64_t Cin);
size_t advstep_conv1x1_mfm_sel_bytes(int64_t N, int64_t C, int64_t P);
int advstep_conv1x1_mfm_forward_f32(const float *x, const float *weight, const float *bias, const float *bn_mean,
                                    const float *bn_invstd, float *y, void *sel, int64_t N, int64_t Cin, int64_t C,
                                    int64_t P, advstep_stream_t stream);

/* Input gradient of the block above: gx (N, Cin, P) = sum_c (gy[n, c, p] * gscale[c]) * weight[selected half of
 * pair c, :]; gscale (C) = the BatchNorm's invstd, or NULL. */
int advstep_conv1x1_mfm_backward_f32(const float *gy, const void *sel, const float *weight, const float *gscale,
                                     float *gx, int64_t N, int64_t Cin, int64_t C, int64_t P, advstep_stream_t stream);

/* ---- 3x3 blocks fused on the matrix cores: Conv2d(Cin, 2C, (3,3), padding 1) -> MaxFeatureMap2D -> MaxPool2d(2, 2)
 * [-> BatchNorm2d(eval, affine=False)]   (src/models/lcnn.py:128-131, 135-137, 149-154)
 * Winograd F(2x2, 3x3): the 16 per-position GEMMs run on v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate), the
 * max-feature-map pair maximum, the pool and the BatchNorm are the kernel's epilogue: the 2C-channel conv output is never
 * written.  Rounds like any fp32 Winograd convolution (MIOpen's included), i.e. not bit-equal to a direct convolution.
 *
 * The weights enter pre-transformed (U = G g G^T in the kernel's chunked LDS layout); prepare once per weight version:
 *   mode 0: forward operand, from weight (2C, Cin, 3, 3);
 *   mode 1: operand of the input-gradient convolution (rotated, transposed kernel), from the same weight tensor;
 *           gscale (C = Cout / 2 floats, or NULL) multiplies the rows of conv channels c and c + C: the invstd of the
 *           eval-mode BatchNorm folded behind the block, so the backward kernel needs no separate scaling pass;
 *   mode 2: mode 1 with the reduction channels in the order advstep_conv3x3_mfm_pool2_backward_f32 walks its compact source
 *           (the two halves of a max-feature-map channel in adjacent k-steps: one load of the pooled gradient serves both).
 * advstep_conv3x3_prepared_floats() floats are written.  Cin % 16 == 0, Cin >= 32, (2C) % 32 == 0 (LCNN: 32/48/64 ->
 * 96/128/64); every tensor of a call must be smaller than 2 GiB (the caller splits the batch otherwise). */
int advstep_conv3x3_supported(int64_t Cin, int64_t Cout);
size_t advstep_conv3x3_prepared_floats(int64_t Cin, int64_t Cout, int mode);
int advstep_conv3x3_prepare_f32(const float *weight, const float *gscale, float *U, int64_t Cin, int64_t Cout, int mode,
                                advstep_stream_t stream);

/* x (N, Cin, H, W), U (mode 0), bias (2C) or NULL, bn_mean / bn_invstd (C) or both NULL -> y (N, C, H/2, W/2) and one
 * selection byte per output (the encoding of advstep_mfm_pool2_forward_f32: its backward kernel consumes it). */
int advstep_conv3x3_mfm_pool2_forward_f32(const float *x, const float *U, const float *bias, const float *bn_mean,
                                          const float *bn_invstd, float *y, uint8_t *idx, int64_t N, int64_t Cin,
                                          int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* The same block WITHOUT the pool (src/models/lcnn.py:142-144: Conv2d -> MFM -> BatchNorm): y (N, C, H, W) and one byte
 * per 2x2 tile (advstep_conv3x3_mfm_sel_bytes() bytes, bit 2*row + col set when the second channel half won). */
size_t advstep_conv3x3_mfm_sel_bytes(int64_t N, int64_t C, int64_t H, int64_t W);
int advstep_conv3x3_mfm_forward_f32(const float *x, const float *U, const float *bias, const float *bn_mean,
                                    const float *bn_invstd, float *y, uint8_t *sel, int64_t N, int64_t Cin, int64_t C,
                                    int64_t H, int64_t W, advstep_stream_t stream);
/* ... and its max-feature-map backward: gout (N, 2C, H, W) = gy (N, C, H, W) [* gscale (C)] routed to the selected half,
 * zero in the other; the input gradient then follows from advstep_conv3x3_backward_data_f32. */
int advstep_conv3x3_mfm_backward_f32(const float *gy, const uint8_t *sel, const float *gscale, float *gout, int64_t N,
                                     int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* Input gradient of a Conv2d(Cin, Cout, (3,3), padding 1): gx (N, Cin, H, W) from gout (N, Cout, H, W) and U (mode 1). */
int advstep_conv3x3_backward_data_f32(const float *gout, const float *U, float *gx, int64_t N, int64_t Cin, int64_t Cout,
                                      int64_t H, int64_t W, advstep_stream_t stream);

/* Input gradient of the WHOLE block from its compact state: gy (N, C, H/2, W/2) and the forward's selection bytes.  The
 * (N, 2C, H, W) gradient of the conv output — gy routed to the winning position of the winning half, zero elsewhere —
 * is expanded on the fly inside the kernel's operand load and never written.  U: mode 2 (with the BatchNorm scale). */
int advstep_conv3x3_mfm_pool2_backward_f32(const float *gy, const uint8_t *idx, const float *U, float *gx, int64_t N,
                                           int64_t Cin, int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* ---- recurrent part of a (bi)directional LSTM layer  (src/models/lcnn.py:24-46: nn.LSTM(160, 80, bidirectional)) ------
 * The input projections are computed by the caller with one GEMM:
 *   gx (T, B, D, 4H) = W_ih x_t + b_ih + b_hh per direction d (D = 1 or 2; d = 1 runs over time in reverse),
 * w_hh (D, 4H, H) as torch stores weight_hh_l0 / weight_hh_l0_reverse (gate order i, f, g, o).
 * out (T, B, D*H) receives h_t (direction d in columns [d*H, (d+1)*H)); gates (T, B, D, 4H) the ACTIVATED gates and
 * cell (T, B, D, H) the cell states, both kept for the backward pass.  One workgroup per (utterance, direction).
 * H must satisfy advstep_lstm_supported() (80: LCNN's). */
int advstep_lstm_supported(int64_t H);
int advstep_lstm_forward_f32(const float *gx, const float *w_hh, float *out, float *gates, float *cell, int64_t T,
                             int64_t B, int64_t D, int64_t H, advstep_stream_t stream);

/* dgx (T, B, D, 4H): gradient w.r.t. the gate pre-activations, from dout (T, B, D*H); the caller maps it back to
 * the layer input with one GEMM (dgx . W_ih). */
int advstep_lstm_backward_f32(const float *dout, const float *w_hh, const float *gates, const float *cell, float *dgx,
                              int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream);
/* The same with ONE gradient row per (utterance, direction, unit) for every frame: dout_row (B, D*H) — what the mean over
 * frames hands back (src/models/lcnn.py:205) — read with a zero frame stride, so the (T, B, D*H) expansion never exists. */
int advstep_lstm_backward_bcast_f32(const float *dout_row, const float *w_hh, const float *gates, const float *cell,
                                    float *dgx, int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream);
/* The same with dout[t][b][f] = dz[b] * row[f] formed inside the kernel: the mean's gradient dz (B) times the Linear's row / T
 * (D*H) - what `dz.reshape(B, 1) * w_over_t` -> advstep_lstm_backward_bcast_f32 computes with one elementwise launch more. */
int advstep_lstm_backward_outer_f32(const float *dz, const float *row, const float *w_hh, const float *gates, const float *cell,
                                    float *dgx, int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream);

/* ---- around the two BLSTM layers  (src/models/lcnn.py:196-205) ---------------------------------------------------------------
 *   hidden = conv_out.permute(0, 2, 1, 3).contiguous().view(B, T, C*W);  lstm = blstm2(blstm1(hidden));
 *   z = Linear(C*W, 1)((lstm + hidden).mean(1))
 * pack:        x4 (B, C, T, W) -> xt (T, B, C*W), the sequence-first layout the recurrent kernels read (one copy, not two);
 * forward:     z (B) = bias + sum_k w[k] * mean_t(a[t][b][k] + xt[t][b][k]),  a = the second layer's output (T, B, F), F <= 256:
 *              skip connection + mean over frames + one-row Linear in one pass;
 * unpack_add:  dx4 (B, C, T, W) = dxt (T, B, C*W) + g0 (B, C*W): the two gradients of `hidden` (through the recurrent layers, and
 *              the mean's row) summed while they go back to the convolution's layout. */
int advstep_lcnn_tail_pack_f32(const float *x4, float *xt, int64_t B, int64_t C, int64_t T, int64_t W, advstep_stream_t stream);
int advstep_lcnn_tail_forward_f32(const float *a, const float *xt, const float *w, const float *bias, float *z, int64_t T,
                                  int64_t B, int64_t F, advstep_stream_t stream);
int advstep_lcnn_tail_unpack_add_f32(const float *dxt, const float *g0, float *dx4, int64_t B, int64_t C, int64_t T, int64_t W,
                                     advstep_stream_t stream);
/* unpack_add with g0[b][k] = dz[b] * row[k] formed inside the kernel (see advstep_lstm_backward_outer_f32) */
int advstep_lcnn_tail_unpack_add_outer_f32(const float *dxt, const float *dz, const float *row, float *dx4, int64_t B, int64_t C,
                                           int64_t T, int64_t W, advstep_stream_t stream);

/* ---- recurrent part of a (bi)directional GRU layer  (src/models/specrnet.py:121-127,176-177: nn.GRU(64, 64, 2 layers,
 * bidirectional); MIOpen runs it as ~400 kernels of ~4 us per forward + backward) --------------------------------------
 *   gx (T, B, D, 3H) = W_ih x_t + b_ih per direction d (gate order r, z, n; d = 1 runs over time in reverse), computed
 *   by the caller with one GEMM; w_hh (D, 3H, H), b_hh (D, 3H) as torch stores weight_hh_l* / bias_hh_l*.
 * out (T, B, D*H) receives h_t; saved (T, B, D, 4H) = r, z, n and a_n = W_hn h + b_hn, kept for the backward pass.
 * One workgroup per (utterance, direction).  H must satisfy advstep_gru_supported() (64: SpecRNet's). */
int advstep_gru_supported(int64_t H);
int advstep_gru_forward_f32(const float *gx, const float *w_hh, const float *b_hh, float *out, float *saved, int64_t T,
                            int64_t B, int64_t D, int64_t H, advstep_stream_t stream);
/* dgx (T, B, D, 3H): gradient w.r.t. gx, from dout (T, B, D*H), the saved gates and the forward's own output (h_{t-1});
 * the caller maps it back to the layer input with one GEMM (dgx . W_ih). */
int advstep_gru_backward_f32(const float *dout, const float *w_hh, const float *saved, const float *out, float *dgx,
                             int64_t T, int64_t B, int64_t D, int64_t H, advstep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVSTEP_LCNN_H_ */
