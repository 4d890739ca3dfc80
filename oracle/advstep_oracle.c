/*
 * advstep_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the waveform arithmetic on the adversarial-evaluation hot path of
 * piotrkawa/audio-deepfake-adversarial-attacks.  It exists to CHECK the HIP kernels (tests/, __graft_entry__.smoke())
 * and to serve as the scalar leg of bench.py's cpu_baseline; nothing in the shipped package may call it.
 *
 * Each function cites the reference lines it follows (paths relative to the reference tree).  Rounding rules
 * (validated bit-for-bit against the importable Python reference, see tests/golden/generate_golden.py and
 * tests/test_oracle_golden.py): Python-float hyper-parameters are applied as float32 scalars; one rounding per
 * ATen op; no FMA contraction (build with -ffp-contract=off); IEEE float32 division; clamp = min(max(v, lo), hi)
 * with NaN propagation; sign(v) = (0 < v) - (v < 0).
 * Row L2 norms accumulate in double and round once to float32: the reference's torch.norm accumulates in
 * float32 in an implementation-defined order, so anything downstream of a norm is compared with a tolerance
 * (1e-6 relative), never bit-for-bit.
 *
 * Parity status: PINNED against golden vectors generated from the reference itself (the .npz files under tests/golden).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline float sgnf(float g) { return (float)(0.0f < g) - (float)(g < 0.0f); }

static inline float clampf_(float v, float lo, float hi) {
    v = (v < lo) ? lo : v;
    return (v > hi) ? hi : v;
}

static inline float min_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a < b ? a : b;
}

static inline float max_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}

static float row_l2(const float *v, int64_t T) {
    double s = 0.0;
    for (int64_t t = 0; t < T; ++t) s += (double)v[t] * (double)v[t];
    return (float)sqrt(s);
}

/* src/aa/utils.py:4-9  to_minmax */
void oracle_minmax_normalize_f32(const float *x, float *x01, float *mn, float *mx, int64_t B, int64_t T) {
    for (int64_t b = 0; b < B; ++b) {
        const float *row = x + b * T;
        float lo = row[0], hi = row[0];
        for (int64_t t = 1; t < T; ++t) {
            lo = min_nan(lo, row[t]);
            hi = max_nan(hi, row[t]);
        }
        const float r = hi - lo; /* :8  r = mx - mn */
        for (int64_t t = 0; t < T; ++t) x01[b * T + t] = (row[t] - lo) / r; /* :9 */
        mn[b] = lo;
        mx[b] = hi;
    }
}

/* src/aa/utils.py:12-14  revert_minmax */
void oracle_minmax_revert_f32(const float *x01, const float *mn, const float *mx, float *out, int64_t B, int64_t T) {
    for (int64_t b = 0; b < B; ++b) {
        const float r = mx[b] - mn[b];
        for (int64_t t = 0; t < T; ++t) {
            const float p = x01[b * T + t] * r;
            out[b * T + t] = p + mn[b];
        }
    }
}

/* adversarial_attacks/torchattacks/attacks/fgsm.py:59-60 */
void oracle_fgsm_step_f32(const float *x, const float *grad, float *out, int64_t n, float eps, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) {
        const float s = eps * sgnf(grad[i]);
        out[i] = clampf_(x[i] + s, lo, hi);
    }
}

/* adversarial_attacks/torchattacks/attacks/pgd.py:56-57 (noise given) */
void oracle_pgd_linf_init_noise_f32(const float *x, const float *noise, float *out, int64_t n, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) out[i] = clampf_(x[i] + noise[i], lo, hi);
}

/* adversarial_attacks/torchattacks/attacks/pgd.py:74-76 */
void oracle_pgd_linf_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t n,
                              float alpha, float eps, float lo, float hi) {
    for (int64_t i = 0; i < n; ++i) {
        const float a = adv[i] + alpha * sgnf(grad[i]);     /* :74 */
        const float d = clampf_(a - orig[i], -eps, eps);    /* :75 */
        out[i] = clampf_(orig[i] + d, lo, hi);              /* :76 */
    }
}

/* adversarial_attacks/torchattacks/attacks/pgdl2.py:57-62 (draws given) */
void oracle_pgd_l2_init_noise_f32(const float *x, const float *normal, const float *r, float *out, int64_t B,
                                  int64_t T, float eps, float lo, float hi) {
    for (int64_t b = 0; b < B; ++b) {
        const float nrm = row_l2(normal + b * T, T);  /* :59 */
        const float q = r[b] / nrm;                   /* :61  r/n        */
        const float scale = q * eps;                  /* :61  (r/n)*eps  */
        for (int64_t t = 0; t < T; ++t) {
            const float d = normal[b * T + t] * scale;
            out[b * T + t] = clampf_(x[b * T + t] + d, lo, hi); /* :62 */
        }
    }
}

/* adversarial_attacks/torchattacks/attacks/pgdl2.py:78-88 */
void oracle_pgd_l2_step_f32(const float *adv, const float *grad, const float *orig, float *out, int64_t B, int64_t T,
                            float alpha, float eps, float eps_div, float lo, float hi, float *gnorm, float *dnorm,
                            float *scratch /* T floats */) {
    for (int64_t b = 0; b < B; ++b) {
        const float *a = adv + b * T, *g = grad + b * T, *x = orig + b * T;
        const float gn_raw = row_l2(g, T);
        const float gn = gn_raw + eps_div; /* :78 */
        for (int64_t t = 0; t < T; ++t) {
            const float gh = g[t] / gn;          /* :79 */
            const float st = alpha * gh;         /* :80 */
            const float an = a[t] + st;          /* :80 */
            scratch[t] = an - x[t];              /* :82 */
        }
        const float dn = row_l2(scratch, T);     /* :83 */
        const float rcp = 1.0f / dn;             /* :84  eps / t  ==  t.reciprocal() * eps (Tensor.__rtruediv__) */
        const float f = min_nan(rcp * eps, 1.0f);/* :85 */
        for (int64_t t = 0; t < T; ++t) {
            const float d = scratch[t] * f;      /* :86 */
            out[b * T + t] = clampf_(x[t] + d, lo, hi); /* :88 */
        }
        if (gnorm) gnorm[b] = gn_raw;
        if (dnorm) dnorm[b] = dn;
    }
}

/* adversarial_attacks/torchattacks/attacks/cw.py:117-122 */
void oracle_cw_init_w_f32(const float *x, float *w, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const float y = x[i] * 2.0f - 1.0f;
        w[i] = 0.5f * logf((1.0f + y) / (1.0f - y));
    }
}

/* cw.py:72,75-76,114-115 */
void oracle_cw_tanh_sqdist_f32(const float *w, const float *x, float *adv, float *l2, int64_t B, int64_t T) {
    for (int64_t b = 0; b < B; ++b) {
        double s = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            const float a = 0.5f * (tanhf(w[b * T + t]) + 1.0f);
            const float d = a - x[b * T + t];
            adv[b * T + t] = a;
            s += (double)(d * d);
        }
        l2[b] = (float)s;
    }
}

/* cw.py:68,87-91: autograd of (sum l2 + c*sum f) through tanh_space, then one torch.optim.Adam step
 * (torch/optim/adam.py _single_tensor_adam, amsgrad=False, weight_decay=0, maximize=False). */
void oracle_cw_adam_step_f32(float *w, float *m, float *v, const float *x, const float *grad_adv, int64_t n,
                             int64_t step, double lr, double beta1, double beta2, double adam_eps) {
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
    const float neg_step = (float)(-(lr / bc1)), bc2s = (float)sqrt(bc2), eps = (float)adam_eps;
    for (int64_t i = 0; i < n; ++i) {
        const float y = tanhf(w[i]);
        const float a = 0.5f * (y + 1.0f);
        const float ga = 2.0f * (a - x[i]) + grad_adv[i]; /* d/d adv of MSE-sum + model term */
        const float g = (ga * 0.5f) * (1.0f - y * y);     /* through 1/2*(tanh+1) */
        m[i] = m[i] + w1 * (g - m[i]);                    /* exp_avg.lerp_(grad, 1 - beta1) */
        v[i] = v[i] * b2 + (omb2 * g) * g;                /* exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1-beta2) */
        const float denom = sqrtf(v[i]) / bc2s + eps;
        w[i] = w[i] + neg_step * (m[i] / denom);          /* param.addcdiv_(exp_avg, denom, value=-step_size) */
    }
}

/* cw.py:99-103 */
void oracle_cw_best_update_f32(const float *adv, const float *mask, float *best, int64_t B, int64_t T) {
    for (int64_t b = 0; b < B; ++b) {
        const float mk = mask[b], inv = 1.0f - mk;
        for (int64_t t = 0; t < T; ++t) best[b * T + t] = mk * adv[b * T + t] + inv * best[b * T + t];
    }
}

/* pgd.py:62,50,68: CE(cat([-z, z], 1), y), mean over the batch; closed form. */
void oracle_ce2_loss_grad_f32(const float *z, const int64_t *labels, float *dz, float *loss, int64_t B, float scale) {
    double acc = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        const double flip = 1.0 - 2.0 * (double)labels[b];
        const double u = flip * 2.0 * (double)z[b];
        acc += (u > 0 ? u : 0) + log1p(exp(-fabs(u)));
        const double sig = 1.0 / (1.0 + exp(-u));
        dz[b] = (float)((double)scale * (2.0 / (double)B) * (flip * sig));
    }
    loss[0] = (float)((double)scale * acc / (double)B);
}

/* ---- Philox4x32-10: the build's own random-start stream (no reference counterpart: the reference draws from
 * torch's global generator, pgd.py:56 / pgdl2.py:57,60).  Defined identically here and in csrc/advstep.hip. ---- */

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0;
        c[1] = (uint32_t)p1;
        c[2] = n2;
        c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

static inline float u01(uint32_t bits) { return (float)(bits >> 8) * 5.9604644775390625e-08f; }
static inline float u01_open0(uint32_t bits) { return (float)((bits >> 8) + 1u) * 5.9604644775390625e-08f; }

void oracle_philox_raw(uint64_t c_lo, uint64_t c_hi, uint64_t seed, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)c_lo, (uint32_t)(c_lo >> 32), (uint32_t)c_hi, (uint32_t)(c_hi >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

void oracle_pgd_linf_init_philox_f32(const float *x, float *out, int64_t n, float eps, float lo, float hi,
                                     uint64_t seed, uint64_t offset) {
    const float from = -eps, range = eps - from;
    for (int64_t q = 0; q * 4 < n; ++q) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int k = 0; k < 4 && q * 4 + k < n; ++k) {
            const float nz = u01(c[k]) * range + from;
            out[q * 4 + k] = clampf_(x[q * 4 + k] + nz, lo, hi);
        }
    }
}

static void philox_normal_row(float *nz, int64_t T, uint32_t b, uint64_t seed, uint64_t offset) {
    for (int64_t q = 0; q * 4 < T; ++q) {
        uint32_t c[4] = {(uint32_t)q, b, (uint32_t)offset, (uint32_t)(offset >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const float r0 = sqrtf(-2.0f * logf(u01_open0(c[0])));
        const float r1 = sqrtf(-2.0f * logf(u01_open0(c[2])));
        const float t0 = 6.283185307179586f * u01(c[1]);
        const float t1 = 6.283185307179586f * u01(c[3]);
        const float z[4] = {r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1)};
        for (int k = 0; k < 4 && q * 4 + k < T; ++k) nz[q * 4 + k] = z[k];
    }
}

void oracle_pgd_l2_init_philox_f32(const float *x, float *out, int64_t B, int64_t T, float eps, float lo, float hi,
                                   uint64_t seed, uint64_t offset, float *scratch /* T floats */) {
    for (int64_t b = 0; b < B; ++b) {
        philox_normal_row(scratch, T, (uint32_t)b, seed, offset);
        const uint64_t off1 = offset + 1;
        uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)off1, (uint32_t)(off1 >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const float r = u01(c[0]);
        oracle_pgd_l2_init_noise_f32(x + b * T, scratch, &r, out + b * T, 1, T, eps, lo, hi);
    }
}
