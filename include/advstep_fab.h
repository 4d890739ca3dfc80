/*
 * advstep_fab.h — C ABI of the FAB (Fast Adaptive Boundary) kernels in libadvstep.so.
 *
 * They replace the per-iteration tensor work of the reference's FAB attack
 *   adversarial_attacks/torchattacks/attacks/fab.py:208-292  (attack_single_run's loop body)
 *   adversarial_attacks/torchattacks/attacks/fab.py:562-717  (projection_linf / projection_l2 / projection_l1)
 * which the reference runs as argsort + gather + cumsum + a log2(T)-step bisection over (2B, T) tensors — about 60 ATen
 * launches and an O(T log T) segmented sort per iteration.  The kernels solve the same per-row problem
 *      smallest ||d||_p  with  w.(t + d) = b,  0 <= t + d <= 1      (nearest box corner when out of reach)
 * WITHOUT sorting: the row's optimality condition  sum_i weight_i * min(cap_i, lam) = |w.t - b|  is a concave
 * piecewise-linear equation in one unknown; a monotone fixed-point (Newton) iteration on it converges in a handful of
 * streaming passes over the row (Linf, L2), and a radix selection over the float keys |1/w| (3 key bits per streaming
 * pass, 11 passes) does the greedy L1 selection.  One workgroup of 1024 threads owns one row (T = 64 600 floats = 258 KB, L2-resident between passes);
 * all reductions are fixed-order trees: results are deterministic.
 *
 * Parity: floating point.  The reference's own CPU path (float64-accumulated cumsum) and CUDA path (float32 parallel
 * scan) already differ in summation order; tests/test_gpu_fab.py states the tolerances against oracle/fab.py, which is
 * pinned to reference-generated fixtures.
 *
 * norm_kind: 0 = "Linf", 1 = "L2", 2 = "L1" (the attack's norm; the hyperplane distance uses its dual).
 * Conventions as in advstep.h: contiguous row-major float32 device pointers, caller-owned outputs, stream-ordered,
 * no host synchronisation, status codes.
 */
#ifndef ADVSTEP_FAB_H_
#define ADVSTEP_FAB_H_

#include <stddef.h>
#include <stdint.h>

#include "advstep.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ADVSTEP_FAB_LINF = 0, ADVSTEP_FAB_L2 = 1, ADVSTEP_FAB_L1 = 2 };

/* fab.py:90-112 + :210-229 for the reference's two-column logits cat([-z, z]):  the linearised decision boundary
 * closest to x.  gz (B, T) = d(sum z)/dx from ONE backward pass (the reference runs one per column; the columns'
 * gradients are -gz and +gz exactly), x (B, T) the current points, z (B) the logits, labels (B) int64 in {0, 1}.
 * Per row:  df_k = y_k - y_la,  dg_k = c_k * gz with c_k in {0, +-2},  dist_k = |df_k| / (1e-12 + dualnorm(dg_k)),
 * df_la = 1e10,  ind = argmin_k dist_k.   Outputs (B each):  wscale = c_ind  (the hyperplane normal is wscale * gz),
 * b = -df_ind + wscale * sum(gz * x),  and the raw row statistics  gnorm = dualnorm(gz)  (sum|g| for Linf,
 * sqrt(sum g^2) for L2, max|g| for L1),  gdot = sum(gz * x).  z / labels may be NULL: only gnorm / gdot are written. */
int advstep_fab_hyperplane_f32(const float *gz, const float *x, const float *z, const int64_t *labels, float *wscale,
                               float *b, float *gnorm, float *gdot, int64_t B, int64_t T, int norm_kind,
                               advstep_stream_t stream);

/* fab.py:562-614 / :617-669 / :672-717  projection_{linf,l2,l1}(points, w, b) -> d, for R rows.
 * t (R, T) points; the hyperplane normal of row r is  wscale[r % w_rows] * w[r % w_rows, :]  (w (w_rows, T); wscale may
 * be NULL = 1) — the reference's torch.cat((w, w), 0) is never materialised: pass w_rows = R / 2; b (R).
 * d (R, T) receives the move (it must not alias t or w: the L1 kernel keeps its sort keys there between passes);
 * dnorm (R) its attack norm (max|d|, sqrt(sum d^2), sum|d|: fab.py:248-256). */
int advstep_fab_projection_f32(const float *t, const float *w, const float *wscale, const float *b, float *d,
                               float *dnorm, int64_t R, int64_t w_rows, int64_t T, int norm_kind,
                               advstep_stream_t stream);

/* fab.py:257-267:  a = max(norm, 1e-8);  alpha = min(max(a1 / (a1 + a2), 0), alpha_max)  per row;
 *   out = clamp((x1 + eta * d1) * (1 - alpha) + (x0 + d2 * eta) * alpha, 0, 1)
 * d1 / n1: move and norm of the projection from x1, d2 / n2: from the clean point x0.  out may alias x1. */
int advstep_fab_combine_f32(const float *x1, const float *x0, const float *d1, const float *d2, const float *n1,
                            const float *n2, float *out, int64_t B, int64_t T, float eta, float alpha_max,
                            advstep_stream_t stream);

/* fab.py:271-290 for the rows with is_adv[r] != 0 (others untouched):
 *   t = norm(x1 - x0);  if t < res2: adv = x1, res2 = t  (the reference's mask arithmetic: a NaN norm zeroes adv);
 *   x1 = x0 + (x1 - x0) * beta          (the backward step towards the clean point)
 * x1, adv (B, T) and res2 (B) are updated in place; no host synchronisation (the reference syncs on is_adv.sum()). */
int advstep_fab_backward_step_f32(float *x1, const float *x0, float *adv, float *res2, const uint8_t *is_adv, int64_t B,
                                  int64_t T, float beta, int norm_kind, advstep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVSTEP_FAB_H_ */
