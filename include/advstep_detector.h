/* advstep_detector.h — C ABI of the fused elementwise / pooling kernels and the residual-block convolutions of the SpecRNet
 * and RawNet3 detectors (csrc/detector_elem.hip, csrc/lcnn_wino.hip; SURVEY.md section 8, rows a12 / a13: these op chains
 * run 40-100 times per batch inside `model(adv)`; the elementwise ones are pure HBM streaming over the detectors' largest
 * activations).
 *
 * Conventions as in advstep.h: raw device pointers, contiguous float32 NCHW / NCL tensors, caller-owned outputs, launches on
 * the given HIP stream, status code returned, no global state.  P = H * W (or L) is the per-channel plane size. */
#ifndef ADVSTEP_DETECTOR_H
#define ADVSTEP_DETECTOR_H

#include <stddef.h>
#include <stdint.h>

#include "advstep.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-channel affine + activation --------------------------------------------------------------------------------
 * mode 0  y = leaky_relu(x * scale[c] + shift[c], slope)    SpecRNet: Conv2d bias + BatchNorm2d(eval) -> LeakyReLU(0.3)
 *                                                            (src/models/specrnet.py:76-81: bn2 -> lrelu between conv1, conv2)
 * mode 1  y = relu(x + pre[c]) * scale[c] + shift[c]         RawNet3: Conv1d bias -> ReLU -> BatchNorm1d(eval)
 *                                                            (src/models/rawnet3.py:240-242, 252-254, 262-264)
 * mode 2  y = selu(x * scale[c] + shift[c])                   SpecRNet: BatchNorm2d(eval) -> SELU (src/models/specrnet.py:159, 173)
 * x, y (N, C, P); scale, shift, pre (C) (pre may be NULL = 0).  One read + one write instead of two or three of each. */
int advstep_affine_act_forward_f32(const float *x, const float *scale, const float *shift, const float *pre, float *y,
                                   int64_t N, int64_t C, int64_t P, int mode, float slope, advstep_stream_t stream);
/* gx = gy * d y / d x, recomputed from x (mode 0: slope where x * scale + shift <= 0; mode 1: 0 where x + pre <= 0;
 * mode 2: selu'(x * scale + shift) * scale). */
int advstep_affine_act_backward_f32(const float *gy, const float *x, const float *scale, const float *shift,
                                    const float *pre, float *gx, int64_t N, int64_t C, int64_t P, int mode, float slope,
                                    advstep_stream_t stream);

/* ---- one link of RawNet3's Res2Net chain (src/models/rawnet3.py:244-258: `sp = sp + spx[i]; sp = bns[i](relu(convs[i](sp)))`,
 * `out = cat(out, sp)`) -----------------------------------------------------------------------------------------------------
 * forward:  y = relu(h + pre[c]) * scale[c] + shift[c] written straight into the branch's channel slice of the concatenated
 *           tensor (plane (n, c) of y at n * y_bs + c * P), and, when `other` (the next group: plane at n * other_bs + c * P) is
 *           given, z (N, C, P) = y + other — the next branch's input — in the same pass.  h (N, C, P) = the branch convolution's
 *           output without its bias (pre = that bias, may be NULL).
 * backward: gx (N, C, P) = (h + pre <= 0) ? 0 : (g1 + g2) * scale, g1 = the slice of d(concatenated tensor) (batch stride g1_bs),
 *           g2 = the gradient coming back from the next branch's input (batch stride g2_bs), NULL for the last branch.
 * Batch strides are in elements and >= C * P.  Replaces, per branch, a strided add, the activation pass, a share of `torch.cat`
 * and (backward) autograd's gradient-accumulation add. */
int advstep_res2net_link_forward_f32(const float *h, const float *scale, const float *shift, const float *pre, float *y,
                                     int64_t y_bs, const float *other, int64_t other_bs, float *z, int64_t N, int64_t C, int64_t P,
                                     advstep_stream_t stream);
int advstep_res2net_link_backward_f32(const float *g1, int64_t g1_bs, const float *g2, int64_t g2_bs, const float *h,
                                      const float *scale, const float *shift, const float *pre, float *gx, int64_t N, int64_t C,
                                      int64_t P, advstep_stream_t stream);

/* ---- residual add + MaxPool2d(2) --------------------------------------------------------------------------------------
 * y (N, C, H/2, W/2) = MaxPool2d(2)(a + b + bias[c]), sel = one byte per pooled output (bit 1 = dh, bit 0 = dw of the
 * winner; ATen's scan order and NaN rule).  SpecRNet: `mp(out + identity)` (src/models/specrnet.py:83-90) with the two
 * convolutions' bias adds folded in (bias = bias_conv2 + bias_downsample, or NULL).  b may be NULL (plain pooling). */
int advstep_add_maxpool2_forward_f32(const float *a, const float *b, const float *bias, float *y, uint8_t *sel, int64_t N,
                                     int64_t C, int64_t H, int64_t W, advstep_stream_t stream);
/* g (N, C, H, W): gy at the winner of every 2x2 window, 0 elsewhere (trailing odd row / column included) — the gradient
 * of both a and b. */
int advstep_maxpool2_backward_f32(const float *gy, const uint8_t *sel, float *g, int64_t N, int64_t C, int64_t H, int64_t W,
                                  advstep_stream_t stream);

/* ---- residual add + MaxPool1d(k), kernel = stride = k (2 <= k <= 8) -----------------------------------------------------
 * y (N, C, L/k) = MaxPool1d(k)(a + b); b may be NULL; sel = winner's offset inside its window, one byte per output.
 * RawNet3: `out += residual` -> `MaxPool1d(5 | 3)` (src/models/rawnet3.py:266-269) and `mp3(x1)` (:96, 102).
 * Backward: g (N, C, L) = gy at the winners, 0 elsewhere (the dropped tail of L % k elements included). */
int advstep_add_maxpool1d_forward_f32(const float *a, const float *b, float *y, uint8_t *sel, int64_t N, int64_t C, int64_t L,
                                      int64_t k, advstep_stream_t stream);
int advstep_maxpool1d_backward_f32(const float *gy, const uint8_t *sel, float *g, int64_t N, int64_t C, int64_t L, int64_t k,
                                   advstep_stream_t stream);

/* ---- RawNet3's feature normalisation after the sinc encoder (src/models/rawnet3.py:80-85, log_sinc + norm_sinc == "mean") ------
 * forward:  x (rows, L) = log(|y| + eps) - mean over L of the same      (rows = N * C; L <= advstep_log_meannorm_max_length())
 * backward: gy = (gx - mean_L(gx)) / (|y| + eps) * sign(y)               (torch's abs / log / mean gradients; sign(0) = 0)
 * One read and one write of the 0.4 GB feature map each way instead of 5 / 9 ATen passes; the row sum is taken in a fixed
 * order (deterministic; differs from ATen's reduction order in the last bits). */
int advstep_log_meannorm_max_length(void);
int advstep_log_meannorm_forward_f32(const float *y, float eps, float *x, int64_t rows, int64_t L, advstep_stream_t stream);
int advstep_log_meannorm_backward_f32(const float *gx, const float *y, float eps, float *gy, int64_t rows, int64_t L,
                                      advstep_stream_t stream);

/* ---- RawNet3's AFMS (src/models/rawnet3.py:161-182): y = sigmoid(fc(mean_t x)), out = (x + alpha[c]) * y[n, c] ------------------
 * Row kernels over (rows = N * C, L) tensors; the (N, C)-sized fc / sigmoid stay with the caller:
 *   mode 0  out[row] = mean_L a                                  mode 1  out = (a + alpha[c]) * r0[row]
 *   mode 2  out[row] = sum_L a * (b + alpha[c])   (a = d out, b = x: the gate's gradient)
 *   mode 3  out = a * r0[row] + r1[row]           (a = d out, r0 = y, r1 = the mean's share of d x)
 * c = row % C.  Two passes each way over the block's output instead of three / six ATen kernels; row sums in a fixed order. */
int advstep_afms_row_f32(int mode, const float *a, const float *b, const float *alpha, const float *r0, const float *r1, float *out,
                         int64_t rows, int64_t C, int64_t L, advstep_stream_t stream);

/* The same backward in two halves, for a gate that is itself a function of x (SpecRNet: gate = sigmoid(fc(mean x)), specrnet.py:145-149):
 * first the gate's partial sums alone, then — once the caller has pushed them through sigmoid' and fc^T — gx = gate * scatter(gy) +
 * addc[n, c], addc = the mean's share of d x (may be NULL).  Replaces the full backward + a broadcast + an add over x-sized tensors. */
int advstep_gate_maxpool2_backward_gate_f32(const float *gy, const uint8_t *sel, const float *x, float *ggate_partial, int64_t N,
                                            int64_t C, int64_t H, int64_t W, advstep_stream_t stream);
int advstep_gate_maxpool2_backward_input_f32(const float *gy, const uint8_t *sel, const float *gate, const float *addc, float *gx,
                                             int64_t N, int64_t C, int64_t H, int64_t W, advstep_stream_t stream);

/* ---- RawNet3's attentive statistics (src/models/rawnet3.py:131-132: `sum(x * w, dim=2)`, `sum((x ** 2) * w, dim=2)`) ------------
 * forward:  mu[row] = sum_L x w,  m2[row] = sum_L (x x) w   over (rows = N * C, L) tensors x (features) and w (attention weights)
 * backward: gx = gmu[row] w + 2 gm2[row] x w,   gw = gmu[row] x + gm2[row] x x
 * One pass each way instead of five / ten ATen kernels over the 0.17 GB tensors; row sums in a fixed order. */
int advstep_weighted_stats_forward_f32(const float *x, const float *w, float *mu, float *m2, int64_t rows, int64_t L,
                                       advstep_stream_t stream);
int advstep_weighted_stats_backward_f32(const float *x, const float *w, const float *gmu, const float *gm2, float *gx, float *gw,
                                        int64_t rows, int64_t L, advstep_stream_t stream);

/* ---- the tail of a RawNet3 Bottle2neck (src/models/rawnet3.py:262-269: `bn3(relu(conv3(.)))`, `out += residual`, `mp(out)`) ----
 * forward:  y (N, C, L/k) = MaxPool1d(k)(relu(h + pre[c]) * scale[c] + shift[c] + res), sel as advstep_add_maxpool1d_forward_f32;
 *           h = conv3's output without its bias (pre = that bias or NULL), res = the residual branch; 2 <= k <= 8.
 * backward: g_res (N, C, L) = unpool(gy, sel) (dropped tail 0) and g_h = (h + pre <= 0) ? 0 : g_res * scale in the same pass.
 * Same arithmetic as advstep_affine_act_*(mode 1) followed by advstep_add_maxpool1d_* — the activated tensor is never written,
 * and the windows are gathered through LDS so that global accesses are contiguous. */
int advstep_tail_pool1d_forward_f32(const float *h, const float *res, const float *scale, const float *shift, const float *pre,
                                    float *y, uint8_t *sel, int64_t N, int64_t C, int64_t L, int64_t k, advstep_stream_t stream);
int advstep_tail_pool1d_backward_f32(const float *gy, const uint8_t *sel, const float *h, const float *scale, const float *pre,
                                     float *g_h, float *g_res, int64_t N, int64_t C, int64_t L, int64_t k, advstep_stream_t stream);

/* The attention gate itself on (N, C), C <= 256 (specrnet.py:145-149 with a frozen fc):
 *   forward:  gate = sigmoid(mean (N, C) . w^T (C, C) + bias)                       — addmm + sigmoid in one launch
 *   backward: g_mean (N, C) = inv_hw * ((sum_blocks ggate_partial) * gate * (1 - gate)) . w   — the partial sums' reduction,
 *             sigmoid', fc^T and the mean's 1 / (H W) in one launch (six ATen launches before). */
int advstep_gate_fc_forward_f32(const float *mean, const float *w, const float *bias, float *gate, int64_t N, int64_t C,
                                advstep_stream_t stream);
int advstep_gate_fc_backward_f32(const float *ggate_partial, int64_t blocks, const float *gate, const float *w, float inv_hw,
                                 float *g_mean, int64_t N, int64_t C, advstep_stream_t stream);
/* ---- channel gate + MaxPool2d(2) --------------------------------------------------------------------------------------
 * y = MaxPool2d(2)(x * gate[n, c] + gate[n, c]) (src/models/specrnet.py:145-149 followed by `self.pool`, :163-172).
 * Backward: gx = gate * scatter(gy); ggate_partial (N * C, blocks) holds per-workgroup partial sums of
 * gy * (x_winner + 1) in a fixed order — the caller sums the last dimension (advstep_gate_maxpool2_blocks gives it). */
size_t advstep_gate_maxpool2_blocks(int64_t H, int64_t W);
int advstep_gate_maxpool2_forward_f32(const float *x, const float *gate, float *y, uint8_t *sel, int64_t N, int64_t C,
                                      int64_t H, int64_t W, advstep_stream_t stream);
int advstep_gate_maxpool2_backward_f32(const float *gy, const uint8_t *sel, const float *x, const float *gate, float *gx,
                                       float *ggate_partial, int64_t N, int64_t C, int64_t H, int64_t W,
                                       advstep_stream_t stream);
/* The forward that also writes xw (N, C, H/2, W/2) = the winner's UN-gated input (may be NULL), and the gate's gradient from it:
 * ggate (N * C) = sum over the pooled plane of gy * (xw + 1), fixed order — reads two pooled-size tensors instead of gathering
 * the winners out of x (a quarter of the bytes; x need not be kept for backward). */
int advstep_gate_maxpool2_forward_xw_f32(const float *x, const float *gate, float *y, uint8_t *sel, float *xw, int64_t N,
                                         int64_t C, int64_t H, int64_t W, advstep_stream_t stream);
int advstep_gate_maxpool2_backward_gate_pooled_f32(const float *gy, const float *xw, float *ggate, int64_t N, int64_t C, int64_t H,
                                                   int64_t W, advstep_stream_t stream);

/* ---- 3x3 convolutions of SpecRNet's residual blocks on the fp32 matrix cores (csrc/lcnn_wino.hip) ------------------------
 * Replaces the ATen / MIOpen calls behind `Residual_block2D.forward` (src/models/specrnet.py:73-91: conv1 -> bn2 -> lrelu ->
 * conv2, `out += conv_downsample(x)`, `mp(out)`) and their input gradients while an attack runs (weights frozen).
 *
 * One operator: a stride-1, zero-padded 3x3 convolution over the K1 channels of x1 (N, K1, H, W) PLUS, optionally, a 1x1
 * convolution over the K2 channels of x2 (N, K2, H, W) — one reduction of length K1 + K2 (Winograd F(2x2, 3x3), the 1x1
 * part as centre-tap-only weights) — into `rows` output channels.  K1 % 4 == 0 when K2 > 0; K1 + K2, rows <= 256;
 * every tensor < 2 GiB.  The weights are given once, transformed (advstep_resconv_prepare_f32) into U:
 *   transpose 0   w3 (rows, K1, 3, 3), w1 (rows, K2)     the forward convolution(s)
 *   transpose 1   w3 (K1, rows, 3, 3), w1 (K2, rows)     the input gradient of a convolution with these forward weights
 *   rscale (rows) or NULL: factor on an output row (a folded eval BatchNorm scale);
 *   kscale (K1) or NULL: factor on a reduction channel of the 3x3 part (the same scale, seen from the gradient side). */
int advstep_resconv_supported(int64_t K1, int64_t K2, int64_t rows);
size_t advstep_resconv_prepared_floats(int64_t K1, int64_t K2, int64_t rows);
int advstep_resconv_prepare_f32(const float *w3, const float *w1, const float *rscale, const float *kscale, float *U,
                                int64_t rows, int64_t K1, int64_t K2, int transpose, advstep_stream_t stream);
/* y (N, rows, H, W) = leaky_relu(conv + shift[row], slope); shift may be NULL, slope = 1 is the plain convolution. */
int advstep_resconv_forward_f32(const float *x1, const float *x2, const float *U, const float *shift, float slope, float *y,
                                int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                advstep_stream_t stream);
/* The same, also writing the activation's SIGN BYTES act (N, rows, ceil(H/2), ceil(W/2)): bit 2 i + j of the byte of a 2x2
 * tile = y > 0 at tile position (i, j) — all advstep_resconv_pooled_grad_act_f32 needs of y (1/16 of its bytes).  act may be
 * NULL (= advstep_resconv_forward_f32). */
int advstep_resconv_forward_act_f32(const float *x1, const float *x2, const float *U, const float *shift, float slope, float *y,
                                    uint8_t *act, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                    advstep_stream_t stream);
/* y (N, rows, H/2, W/2) = MaxPool2d(2)(conv + bias[row]) with the selection bytes of advstep_add_maxpool2_forward_f32
 * (consumed by advstep_maxpool2_backward_f32): the full-resolution convolution output is never written. */
int advstep_resconv_pool2_forward_f32(const float *x1, const float *x2, const float *U, const float *bias, float *y,
                                      uint8_t *sel, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H, int64_t W,
                                      advstep_stream_t stream);
/* The same with the 1x1 part over FEW channels (K2 = 1 or 2: the downsample convolution of SpecRNet's first block, whose input
 * is the 1- or 2-channel spectrogram): U is prepared from the 3x3 weights alone (K2 = 0) and the 1x1 weights wd (rows, K2) are
 * applied in the epilogue on the vector ALUs — as reduction channels they would pad a whole k-step (4 channels) of matrix
 * instructions.  Sums in another order than advstep_resconv_pool2_forward_f32 (fp32 rounding differences only). */
int advstep_resconv_pool2_forward_few_f32(const float *x1, const float *x2, const float *U, const float *wd, const float *bias,
                                          float *y, uint8_t *sel, int64_t N, int64_t K1, int64_t K2, int64_t rows, int64_t H,
                                          int64_t W, advstep_stream_t stream);

/* g (N, rows, H, W) = conv3x3(unpool(gy, sel)) [* (h > 0 ? 1 : slope)]: the input gradient of a convolution whose OUTPUT went
 * through MaxPool2d(2), from the pooled gradient gy (N, K, H/2, W/2) and the selection bytes — the full-resolution
 * d(conv out) is expanded inside the operand load and never written — with U prepared with transpose 1; h (N, rows, H, W),
 * when given, is the LeakyReLU output that fed the convolution (its sign is its input's sign): the activation's backward
 * in the epilogue (specrnet.py:80-81 on the way back). */
int advstep_resconv_pooled_grad_f32(const float *gy, const uint8_t *sel, const float *U, const float *h, float slope, float *g,
                                    int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W, advstep_stream_t stream);

/* The same with the activation given as sign bytes (advstep_resconv_forward_act_f32 / advstep_conv3x3_fewin_forward_act_f32):
 * g = conv3x3(unpool(gy, sel)) * (bit ? 1 : slope).  Bit-identical to the h form for slope > 0; the epilogue reads 1 byte per
 * 2x2 tile and row instead of 16 (SpecRNet block0, B = 128: 21 MB instead of 331 MB). */
int advstep_resconv_pooled_grad_act_f32(const float *gy, const uint8_t *sel, const float *U, const uint8_t *act, float slope,
                                        float *g, int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W,
                                        advstep_stream_t stream);

/* ---- the spectrogram end of SpecRNet's first block: 3x3 convolutions with 1-2 channels on one side (csrc/detector_conv.hip) --
 * Vector-ALU kernels (a matrix tile would be mostly padding), thread = one 2x2 block of positions.
 * forward:  y (N, Cout, H, W) = leaky_relu(conv3x3(x (N, Cin, H, W), w (Cout, Cin, 3, 3), pad 1) + shift[co], slope), Cin in {1, 2};
 *           a folded BatchNorm scale is expected INSIDE w (specrnet.py:76-81: conv1 -> bn2 -> lrelu of block0).
 * grad:     gx (N, rows, H, W), rows in {1, 2}: the input gradient of conv3x3(x, w3 (K, rows, 3, 3)) given d(out) = g1 (N, K, H, W),
 *           plus — when gp / sel / wd are given — the input gradient of the 1x1 convolution wd (K, rows) whose d(out) is the
 *           unpooled gradient of MaxPool2d(2) in compact form (gp (N, K, H/2, W/2), sel as advstep_add_maxpool2_forward_f32 writes
 *           it): `conv_downsample`'s share of d x (specrnet.py:83-90 on the way back).  Every sample < 2 GiB per tensor. */
int advstep_conv3x3_fewin_supported(int64_t channels);
int advstep_conv3x3_fewin_forward_f32(const float *x, const float *w, const float *shift, float slope, float *y, int64_t N,
                                      int64_t Cin, int64_t Cout, int64_t H, int64_t W, advstep_stream_t stream);
/* forward that also writes the sign bytes act (N, Cout, ceil(H/2), ceil(W/2)) of its output (see advstep_resconv_forward_act_f32);
 * act may be NULL. */
int advstep_conv3x3_fewin_forward_act_f32(const float *x, const float *w, const float *shift, float slope, float *y,
                                          uint8_t *act, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                          advstep_stream_t stream);
int advstep_conv3x3_fewout_grad_f32(const float *g1, const float *w3, const float *gp, const uint8_t *sel, const float *wd,
                                    float *gx, int64_t N, int64_t K, int64_t rows, int64_t H, int64_t W,
                                    advstep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
