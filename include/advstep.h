/*
 * advstep.h — C ABI of libadvstep.so: the MI355X (gfx950) waveform-perturbation kernels behind the
 * torchattacks.Attack plugin API of piotrkawa/audio-deepfake-adversarial-attacks.
 *
 * Every entry point replaces one eager ATen op chain of the reference's adversarial-evaluation hot loop
 * (evaluate_models_on_adversarial_attacks.py:211-265).  The reference interface each one replaces is cited
 * as  <reference file>:<lines>  relative to the reference tree.
 *
 * Conventions (all entry points)
 *   - plain C symbols, raw DEVICE pointers, sizes as int64_t, scalars by value, no torch / C++ types;
 *   - tensors are contiguous row-major float32: waveforms are (B, T) (T = 64 600 for the repo's 4 s cut),
 *     "flat" entry points take n = B*T;  per-row scalars are (B);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); kernels are stream-ordered, never
 *     synchronise the device and keep no state in the library or the process: the ONLY state that survives a call is what
 *     the caller's own `ws` buffer holds (next item).  Calls on different streams with different buffers are independent;
 *   - the caller owns every buffer, including the row-reduction scratch `ws`: >= advstep_row_workspace_bytes(B, T) bytes,
 *     16-byte aligned, ZERO-FILLED ONCE by the caller before its first use, and from then on used by ONE stream at a time and
 *     for ONE (B, T).  The single-pass PGD-L2 calls keep their in-launch exchange state in it — a call counter in its first
 *     word, tagged 8-byte granules, per-row flags — in an area the float partial sums of the other entry points do not touch
 *     for the same (B, T); nothing has to be cleaned between calls (a granule counts only if it carries the tag of THIS call:
 *     counter + phase), and flags that are not zero merely send their rows through the repair pass (slower, same bits;
 *     advstep_pgd_l2_repaired_rows counts them).  Sharing one `ws` between two (B, T) or between two streams that may run
 *     concurrently is OUTSIDE the contract and the results of the single-pass PGD-L2 calls are then UNDEFINED: the layouts
 *     overlap, the counter is advanced by a plain read-modify-write, and a word another layout or another stream's call left
 *     behind is taken for this call's whenever it equals the 32-bit tag.  The library cannot diagnose it (a status code
 *     would need a device synchronisation); give every (stream, B, T) its own buffer, as the Python binding does.  ABI 3);
 *   - `out` may alias the first waveform input of the same call (in-place update) unless stated otherwise;
 *   - return value: ADVSTEP_OK or an ADVSTEP_E* code; nothing is thrown across the ABI.
 *   - Python-float hyper-parameters of the reference enter as float32 (that is how ATen applies a Python
 *     scalar to a float32 tensor); arithmetic order and rounding follow the reference expression by
 *     expression: no FMA contraction, IEEE division, NaN-propagating clamps, sign(NaN) = sign(0) = 0.
 */
#ifndef ADVSTEP_H_
#define ADVSTEP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): the row workspace has a new layout (header word + exchange area + float planes: advstep_row_workspace_bytes
 * grew again) and the single-pass PGD-L2 calls tag their exchange with a call counter instead of cleaning it.
 * 2 (round 4): advstep_conv3x3_mfm_pool2_backward_f32 takes a mode-2 prepared U (a mode-1 U — version 1's contract — gives wrong
 * gradients without an error) and needs 128 KB of LDS per workgroup for K > 64; advstep_row_workspace_bytes grew in rounds 3
 * and 4, and `ws` must be zero-filled once before its first use (no memset node inside the PGD-L2 calls any more).
 * A binding built against another version must refuse the library (the Python binding does, for every build it is pointed at). */
#define ADVSTEP_ABI_VERSION 3

enum {
    ADVSTEP_OK = 0,
    ADVSTEP_EINVAL = 1,     /* null pointer, negative size, bad scalar                      */
    ADVSTEP_EWORKSPACE = 2, /* ws == NULL or ws_bytes < advstep_row_workspace_bytes(B, T)   */
    ADVSTEP_ELAUNCH = 3,    /* hipGetLastError() != hipSuccess after a launch               */
    ADVSTEP_ENODEVICE = 4   /* no HIP device visible to this process                        */
};

typedef void *advstep_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------------------------------------- */

int advstep_abi_version(void);
const char *advstep_status_string(int status);
/* Number of HIP devices visible (0 on a CPU-only box); never fails. */
int advstep_device_count(void);
/* Bytes of scratch any row-reducing entry point needs for a (B, T) batch. */
size_t advstep_row_workspace_bytes(int64_t B, int64_t T);

/* ---- a1 / a2: per-utterance min-max normalisation ---------------------------------------------------- */

/* src/aa/utils.py:4-9  to_minmax:  mn = min_t x, mx = max_t x, x01 = (x - mn) / (mx - mn)  per row.
 * A constant row yields NaN exactly like the reference (0/0).  x01 must not alias x. */
int advstep_minmax_normalize_f32(const float *x, float *x01, float *mn, float *mx, int64_t B, int64_t T,
                                 void *ws, size_t ws_bytes, advstep_stream_t stream);

/* src/aa/utils.py:12-14  revert_minmax:  out = (x01 * (mx - mn)) + mn  (two roundings). */
int advstep_minmax_revert_f32(const float *x01, const float *mn, const float *mx, float *out, int64_t B,
                              int64_t T, advstep_stream_t stream);

/* ---- a4: FGSM ------------------------------------------------------------------------------------------ */

/* adversarial_attacks/torchattacks/attacks/fgsm.py:59-60
 *   out = clamp(x + eps * sign(grad), lo, hi)          (reference: lo = 0, hi = 1) */
int advstep_fgsm_step_f32(const float *x, const float *grad, float *out, int64_t n, float eps, float lo,
                          float hi, advstep_stream_t stream);

/* ---- a5: PGD (L-inf) ----------------------------------------------------------------------------------- */

/* adversarial_attacks/torchattacks/attacks/pgd.py:54-57 with the uniform(-eps, eps) draw supplied by the
 * caller (the reference's own noise tensor for parity runs):  out = clamp(x + noise, lo, hi). */
int advstep_pgd_linf_init_noise_f32(const float *x, const float *noise, float *out, int64_t n, float lo,
                                    float hi, advstep_stream_t stream);

/* Same start with the noise generated in-kernel: Philox4x32-10, key = seed, counter = (i / 4, offset),
 * u = (bits >> 8) * 2^-24 in [0, 1), noise = u * (eps - (-eps)) + (-eps).  8 B/sample instead of 12. */
int advstep_pgd_linf_init_philox_f32(const float *x, float *out, int64_t n, float eps, float lo, float hi,
                                     uint64_t seed, uint64_t offset, advstep_stream_t stream);

/* adversarial_attacks/torchattacks/attacks/pgd.py:74-76  (the 7-launch chain fused into one pass)
 *   a   = adv + alpha * sign(grad)
 *   d   = clamp(a - orig, -eps, eps)
 *   out = clamp(orig + d, lo, hi) */
int advstep_pgd_linf_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t n,
                              float alpha, float eps, float lo, float hi, advstep_stream_t stream);

/* ---- a6: PGD (L2) -------------------------------------------------------------------------------------- */

/* adversarial_attacks/torchattacks/attacks/pgdl2.py:55-62 with caller-supplied draws
 *   normal (B, T) ~ N(0, 1),  r (B) ~ U(0, 1):
 *   nrm = ||normal_b||_2 ;  d = normal * ((r / nrm) * eps) ;  out = clamp(x + d, lo, hi). */
int advstep_pgd_l2_init_noise_f32(const float *x, const float *normal, const float *r, float *out, int64_t B,
                                  int64_t T, float eps, float lo, float hi, void *ws, size_t ws_bytes,
                                  advstep_stream_t stream);

/* Same start with in-kernel Philox4x32-10 + Box-Muller normals (key = seed, counter = (i / 4, offset));
 * r_b comes from counter (b, offset + 1).  The normals are regenerated, never stored. */
int advstep_pgd_l2_init_philox_f32(const float *x, float *out, int64_t B, int64_t T, float eps, float lo,
                                   float hi, uint64_t seed, uint64_t offset, void *ws, size_t ws_bytes,
                                   advstep_stream_t stream);

/* adversarial_attacks/torchattacks/attacks/pgdl2.py:78-88  (>= 12 launches fused into 3 row passes)
 *   gn  = ||grad_b||_2 + eps_div ;  a = adv + alpha * (grad / gn)
 *   d   = a - orig ;  dn = ||d_b||_2 ;  f = min((1 / dn) * eps, 1) ;  out = clamp(orig + d * f, lo, hi)
 * gnorm / dnorm (B) receive ||grad_b||_2 (before eps_div) and dn; either may be NULL. */
int advstep_pgd_l2_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t B,
                            int64_t T, float alpha, float eps, float eps_div, float lo, float hi,
                            float *gnorm, float *dnorm, void *ws, size_t ws_bytes, advstep_stream_t stream);

/* Diagnostics for the single-pass forms of the two calls above (no reference counterpart).  When a (B, T) launch fits the
 * device's resident capacity, advstep_pgd_l2_step_f32 / advstep_pgd_l2_init_philox_f32 exchange the row norms inside ONE
 * launch; a row whose exchange could not complete within its bound (another stream or process holding the compute units)
 * is recomputed by a repair kernel queued behind the launch — same arithmetic, same bits, never a wrong or NaN row.
 * This writes to *count (device int) how many rows of the LAST such call on `ws` went through the repair kernel. */
int advstep_pgd_l2_repaired_rows(const void *ws, size_t ws_bytes, int64_t B, int64_t T, int *count,
                                 advstep_stream_t stream);

/* ---- a7: Carlini-Wagner (L2, tanh space, Adam) ---------------------------------------------------------- */

/* adversarial_attacks/torchattacks/attacks/cw.py:57,117-122   w = 0.5 * log((1 + y) / (1 - y)), y = x*2 - 1 */
int advstep_cw_init_w_f32(const float *x, float *w, int64_t n, advstep_stream_t stream);

/* cw.py:72,75-76,114-115   adv = 1/2 * (tanh(w) + 1) ;  l2[b] = sum_t (adv - x)^2 */
int advstep_cw_tanh_sqdist_f32(const float *w, const float *x, float *adv, float *l2, int64_t B, int64_t T,
                               void *ws, size_t ws_bytes, advstep_stream_t stream);

/* cw.py:68,87-91  one torch.optim.Adam(lr, betas, adam_eps) step on w for
 * cost = sum_b l2[b] + c * sum_b f_b, given grad_adv = d(c * sum f)/d adv from the model's backward pass.
 * tanh(w) is recomputed (bit-identical to advstep_cw_tanh_sqdist_f32), adv is not re-read:
 *   y = tanh(w) ; a = 1/2 * (y + 1) ; g = ((2 * (a - x) + grad_adv) * 0.5) * (1 - y * y)
 *   m = m + (1 - beta1) * (g - m) ;  v = v * beta2 + ((1 - beta2) * g) * g
 *   w = w + (-(lr / (1 - beta1^step))) * (m / (sqrt(v) / sqrt(1 - beta2^step) + adam_eps))
 * Hyper-parameters are doubles: the bias corrections are formed in double on the host exactly as the
 * Python optimiser does, then applied as float32 scalars.  `step` is the 1-based Adam step count.
 * w, m, v are updated in place. */
int advstep_cw_adam_step_f32(float *w, float *m, float *v, const float *x, const float *grad_adv, int64_t n,
                             int64_t step, double lr, double beta1, double beta2, double adam_eps,
                             advstep_stream_t stream);

/* cw.py:99-103   best = mask * adv + (1 - mask) * best, mask (B) in {0, 1} broadcast over the row. */
int advstep_cw_best_update_f32(const float *adv, const float *mask, float *best, int64_t B, int64_t T,
                               advstep_stream_t stream);

/* ---- a8: one-logit -> two-logit adapter + mean cross-entropy, closed form ------------------------------- */

/* pgd.py:62,50,68 (same lines in fgsm.py:47, pgdl2.py:67):  out = cat([-z, z], 1), cost = CE(out, y) (mean).
 *   u_b = (1 - 2 y_b) * 2 z_b ;  loss_b = softplus(u_b) ;
 *   dz[b] = scale * (2 / B) * (1 - 2 y_b) * sigmoid(u_b)      ( == (2 / B) * (sigmoid(2 z_b) - y_b), no cancellation )
 * loss (1) receives scale * mean_b loss_b; `scale` = +1 (untargeted) or -1 (targeted, cost = -CE).
 * Single-workgroup kernel (B is a batch size); labels are int64 in {0, 1}. */
int advstep_ce2_loss_grad_f32(const float *z, const int64_t *labels, float *dz, float *loss, int64_t B,
                              float scale, advstep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVSTEP_H_ */
