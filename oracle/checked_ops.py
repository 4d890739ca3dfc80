"""TEST INFRASTRUCTURE — `hip_ops` with every launch re-computed by the plain-C oracle and compared.

`CheckedOps(hip_ops)` is an op table: each call runs the HIP kernel on the device tensors, copies the SAME inputs
to the host, runs the oracle (oracle/kernels.py) and asserts agreement — bit-for-bit where the arithmetic is
order-independent, within the stated tolerance where a row norm or a libm/OCML transcendental is involved.
Installed into an attack (`atk.ops = CheckedOps(hip_ops)`) it verifies every kernel launch of a real attack, with
real model gradients, in situ.  Used by `pytest -m gpu` and `__graft_entry__.smoke()`."""
from collections import Counter

import numpy as np
import torch

from . import kernels as K

# tolerances (documented in DESIGN.md "Parity")
L2_ATOL = 3e-7        # anything downstream of a row L2 norm (f32 reduction order differs): |a - b| <= 3e-7 in [0,1]
NORM_RTOL = 2e-6      # the row norms themselves
TANH_ATOL = 2e-7      # 1/2(tanh w + 1): OCML vs glibc tanhf differ by <= 2 ulp
LOG_RTOL = 2e-6       # atanh via logf
ADAM_RTOL = 2e-5      # Adam moments
ADAM_W_ATOL = 2e-6    # Adam parameter update where the gradient is well above rounding noise
CE_RTOL = 1e-5


def _np(t):
    return t.detach().cpu().numpy()


def _exact(name, got, want):
    got, want = _np(got), np.asarray(want)
    if not np.array_equal(got, want.reshape(got.shape), equal_nan=True):
        bad = np.flatnonzero(~((got == want.reshape(got.shape)) | (np.isnan(got) & np.isnan(want.reshape(got.shape)))))
        raise AssertionError(f"{name}: HIP != oracle at {bad.size}/{got.size} samples "
                             f"(first {bad[:4]}, max abs diff {np.nanmax(np.abs(got - want.reshape(got.shape)))})")


def _close(name, got, want, atol=0.0, rtol=0.0):
    got, want = _np(got), np.asarray(want).reshape(_np(got).shape)
    both_nan = np.isnan(got) & np.isnan(want)
    same_inf = np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want))
    err = np.abs(got - want)
    ok = (err <= atol + rtol * np.abs(want)) | both_nan | same_inf
    if not ok.all():
        raise AssertionError(f"{name}: {np.count_nonzero(~ok)}/{got.size} samples beyond atol={atol} rtol={rtol}; "
                             f"max abs err {np.nanmax(np.where(ok, 0, err))}")


class CheckedOps:
    """`every` > 1 samples: of each entry point's launches only number 0, every, 2 * every ... are re-computed by the
    oracle (the others run unchecked) — for full-size workloads, where one check copies four (128, 64600) arrays to
    the host.  `calls` counts all launches, `checked` the compared ones."""
    NAME = "hip+oracle-check"
    SAMPLED = ("to_minmax", "revert_minmax", "fgsm_step", "pgd_linf_init", "pgd_linf_step", "pgd_l2_init", "pgd_l2_step",
               "cw_init_w", "cw_tanh_sqdist", "cw_adam_step", "cw_best_update", "ce2_loss_grad")

    def __init__(self, hip_ops, every: int = 1):
        self.hip = hip_ops
        self.calls = Counter()
        self.checked = Counter()
        self.every = max(int(every), 1)
        for name in self.SAMPLED:
            setattr(self, name, self._sampled(name, getattr(self, name), getattr(hip_ops, name)))

    def _sampled(self, name, checked_fn, plain_fn):
        seen = Counter()

        def run(*args, **kwargs):
            n = seen[name]
            seen[name] += 1
            if n % self.every == 0:
                self.checked[name] += 1
                return checked_fn(*args, **kwargs)
            self.calls[name] += 1
            return plain_fn(*args, **kwargs)

        return run

    # a1 / a2 -------------------------------------------------------------------------------------------
    def to_minmax(self, batch_x):
        x01, mn, mx = self.hip.to_minmax(batch_x)
        w01, wmn, wmx = K.minmax_normalize(_np(batch_x).reshape(batch_x.shape[0], -1))
        _exact("to_minmax.mn", mn, wmn), _exact("to_minmax.mx", mx, wmx), _exact("to_minmax.x01", x01, w01)
        self.calls["to_minmax"] += 1
        return x01, mn, mx

    def revert_minmax(self, batch_x, mn, mx, out=None):
        xin = _np(batch_x).reshape(batch_x.shape[0], -1).copy()
        got = self.hip.revert_minmax(batch_x, mn, mx, out=out)
        _exact("revert_minmax", got, K.minmax_revert(xin, _np(mn), _np(mx)))
        self.calls["revert_minmax"] += 1
        return got

    # a4 / a5 -------------------------------------------------------------------------------------------
    def fgsm_step(self, x, grad, eps, lo=0.0, hi=1.0, out=None):
        xin = _np(x).copy()
        got = self.hip.fgsm_step(x, grad, eps, lo, hi, out=out)
        _exact("fgsm_step", got, K.fgsm_step(xin, _np(grad), eps, lo, hi))
        self.calls["fgsm_step"] += 1
        return got

    def pgd_linf_init(self, x, eps, noise=None, seed=None, offset=0, lo=0.0, hi=1.0, out=None):
        xin = _np(x).copy()
        got = self.hip.pgd_linf_init(x, eps, noise=noise, seed=seed, offset=offset, lo=lo, hi=hi, out=out)
        want = (K.pgd_linf_init_noise(xin, _np(noise), lo, hi) if noise is not None
                else K.pgd_linf_init_philox(xin, eps, seed, offset, lo, hi))
        _exact("pgd_linf_init", got, want)
        self.calls["pgd_linf_init"] += 1
        return got

    def pgd_linf_step(self, adv, grad, orig, alpha, eps, lo=0.0, hi=1.0, out=None):
        ain = _np(adv).copy()
        got = self.hip.pgd_linf_step(adv, grad, orig, alpha, eps, lo, hi, out=out)
        _exact("pgd_linf_step", got, K.pgd_linf_step(ain, _np(grad), _np(orig), alpha, eps, lo, hi))
        self.calls["pgd_linf_step"] += 1
        return got

    # a6 --------------------------------------------------------------------------------------------------
    def pgd_l2_init(self, x, eps, draws=None, seed=None, offset=0, lo=0.0, hi=1.0, out=None):
        B = x.shape[0]
        xin = _np(x).reshape(B, -1).copy()
        got = self.hip.pgd_l2_init(x, eps, draws=draws, seed=seed, offset=offset, lo=lo, hi=hi, out=out)
        if draws is not None:
            want = K.pgd_l2_init_noise(xin, _np(draws[0]).reshape(B, -1), _np(draws[1]), eps, lo, hi)
        else:
            want = K.pgd_l2_init_philox(xin, eps, seed, offset, lo, hi)
        # Philox start: logf/cosf/sinf differ by a few ulp between OCML and glibc; the draw is scaled by ~eps/sqrt(T)
        _close("pgd_l2_init", got, want, atol=L2_ATOL if draws is not None else 1e-6 * max(eps, 1e-3))
        self.calls["pgd_l2_init"] += 1
        return got

    def pgd_l2_step(self, adv, grad, orig, alpha, eps, eps_div=1e-10, lo=0.0, hi=1.0, out=None, return_norms=False):
        B = adv.shape[0]
        ain = _np(adv).reshape(B, -1).copy()
        got, gn, dn = self.hip.pgd_l2_step(adv, grad, orig, alpha, eps, eps_div, lo, hi, out=out, return_norms=True)
        want, wgn, wdn = K.pgd_l2_step(ain, _np(grad).reshape(B, -1), _np(orig).reshape(B, -1), alpha, eps, eps_div, lo, hi)
        _close("pgd_l2_step.gnorm", gn, wgn, rtol=NORM_RTOL), _close("pgd_l2_step.dnorm", dn, wdn, rtol=NORM_RTOL)
        _close("pgd_l2_step", got, want, atol=L2_ATOL)
        self.calls["pgd_l2_step"] += 1
        return (got, gn, dn) if return_norms else got

    # a7 --------------------------------------------------------------------------------------------------
    def cw_init_w(self, x, out=None):
        got = self.hip.cw_init_w(x, out=out)
        _close("cw_init_w", got, K.cw_init_w(_np(x)), atol=1e-6, rtol=LOG_RTOL)
        self.calls["cw_init_w"] += 1
        return got

    def cw_tanh_sqdist(self, w, x, adv_out=None):
        B = w.shape[0]
        adv, l2 = self.hip.cw_tanh_sqdist(w, x, adv_out=adv_out)
        wadv, wl2 = K.cw_tanh_sqdist(_np(w).reshape(B, -1), _np(x).reshape(B, -1))
        _close("cw_tanh_sqdist.adv", adv, wadv, atol=TANH_ATOL)
        T = wadv.shape[1]
        _close("cw_tanh_sqdist.l2", l2, wl2, atol=4 * TANH_ATOL * np.sqrt(T) * np.sqrt(np.maximum(wl2, 0)).max() + 1e-10,
               rtol=1e-5)
        self.calls["cw_tanh_sqdist"] += 1
        return adv, l2

    def cw_adam_step(self, w, m, v, x, grad_adv, step, lr=0.01, beta1=0.9, beta2=0.999, adam_eps=1e-8):
        w0, m0, v0 = _np(w).copy(), _np(m).copy(), _np(v).copy()
        self.hip.cw_adam_step(w, m, v, x, grad_adv, step, lr, beta1, beta2, adam_eps)
        ww, wm, wv = K.cw_adam_step(w0, m0, v0, _np(x), _np(grad_adv), step, lr, beta1, beta2, adam_eps)
        _close("cw_adam_step.m", m, wm, atol=2e-8, rtol=ADAM_RTOL)
        _close("cw_adam_step.v", v, wv, atol=1e-12, rtol=10 * ADAM_RTOL)
        # the update direction m/(sqrt(v)+eps) is ill-conditioned where the gradient is at rounding-noise level
        # (Adam normalises it to O(1)); compare w where the second moment says the gradient is resolved
        resolved = np.sqrt(wv) > 1e-5
        got_w = _np(w)
        err = np.abs(got_w - ww.reshape(got_w.shape))[resolved.reshape(got_w.shape)]
        if err.size and err.max() > ADAM_W_ATOL + 1e-5 * lr:
            raise AssertionError(f"cw_adam_step.w: max abs err {err.max()} on resolved coordinates")
        noise = np.abs(got_w - w0.reshape(got_w.shape))
        if np.nanmax(np.where(np.isfinite(noise), noise, 0)) > 4 * lr:
            raise AssertionError("cw_adam_step.w: an update exceeds the Adam step bound")
        self.calls["cw_adam_step"] += 1

    def cw_best_update(self, adv, mask, best):
        B = adv.shape[0]
        b0 = _np(best).reshape(B, -1).copy()
        self.hip.cw_best_update(adv, mask, best)
        _exact("cw_best_update", best, K.cw_best_update(_np(adv).reshape(B, -1), _np(mask), b0))
        self.calls["cw_best_update"] += 1

    # a8 --------------------------------------------------------------------------------------------------
    def ce2_loss_grad(self, z, labels, scale=1.0):
        dz, loss = self.hip.ce2_loss_grad(z, labels, scale)
        wdz, wloss = K.ce2_loss_grad(_np(z), _np(labels), scale)
        _close("ce2_loss_grad.dz", dz, wdz.reshape(_np(dz).shape), atol=1e-9, rtol=CE_RTOL)
        _close("ce2_loss_grad.loss", loss, np.array([wloss]), atol=1e-7, rtol=CE_RTOL)
        self.calls["ce2_loss_grad"] += 1
        return dz, loss

    # f3: FAB — floating-point parity against the float64 oracle (tolerances: oracle/fab.py header) ------------------
    def fab_hyperplane(self, gz, x, z=None, labels=None, norm="Linf"):
        from . import torch_ops as O

        got = self.hip.fab_hyperplane(gz, x, z, labels, norm)
        want = O.fab_hyperplane(gz.cpu(), x.cpu(), None if z is None else z.cpu(), None if labels is None else labels.cpu(), norm)
        for name, g, w in zip(("wscale", "b", "gnorm", "gdot"), got, want):
            if g is not None:
                scale = float(np.abs(_np(want[2])).max()) if name in ("b", "gdot") else 0.0
                _close("fab_hyperplane." + name, g, _np(w), atol=1e-6 * scale + 1e-6, rtol=2e-5)
        self.calls["fab_hyperplane"] += 1
        return got

    def fab_projection(self, points, w, b, norm="Linf", wscale=None, out=None):
        from . import torch_ops as O

        pin, win, bin_ = points.cpu().clone(), w.cpu().clone(), b.cpu().clone()
        sin = None if wscale is None else wscale.cpu().clone()
        d, dn = self.hip.fab_projection(points, w, b, norm, wscale, out)
        wd, wdn = O.fab_projection(pin, win, bin_, norm, sin)
        tol = {"Linf": 2e-5, "L2": 2e-5, "L1": 8e-3}[norm]
        _close("fab_projection.d", d, _np(wd), atol=tol)
        _close("fab_projection.norm", dn, _np(wdn), atol=tol, rtol=1e-4)
        self.calls["fab_projection"] += 1
        return d, dn

    def fab_combine(self, x1, x0, d1, d2, n1, n2, eta, alpha_max, out=None):
        from . import torch_ops as O

        want = O.fab_combine(x1.cpu(), x0.cpu(), d1.cpu(), d2.cpu(), n1.cpu(), n2.cpu(), eta, alpha_max)
        got = self.hip.fab_combine(x1, x0, d1, d2, n1, n2, eta, alpha_max, out)
        _exact("fab_combine", got, _np(want))
        self.calls["fab_combine"] += 1
        return got

    def fab_backward_step(self, x1, x0, adv, res2, is_adv, beta, norm="Linf"):
        from . import torch_ops as O

        c1, ca, cr = x1.cpu().clone(), adv.cpu().clone(), res2.cpu().clone()
        O.fab_backward_step(c1, x0.cpu(), ca, cr, is_adv.cpu(), beta, norm)
        self.hip.fab_backward_step(x1, x0, adv, res2, is_adv, beta, norm)
        _exact("fab_backward_step.x1", x1, _np(c1))
        _close("fab_backward_step.res2", res2, _np(cr), rtol=1e-5)
        _exact("fab_backward_step.adv", adv, _np(ca))
        self.calls["fab_backward_step"] += 1


class TracingOps:
    """TEST INFRASTRUCTURE — an op table that forwards every call to `ops` unchanged and keeps, per attack iteration, what
    the reference's loop would show at that point: the iterate entering the update step, the input gradient, the loss and
    the logits (PGD / PGDL2: oracle/attacks.py `trace` holds the CPU twin), and for CW the per-utterance squared
    distances.  Used by the end-to-end GPU-vs-CPU-oracle tests (tests/test_gpu_e2e_parity.py)."""

    def __init__(self, ops):
        self.ops = ops
        self.NAME = getattr(ops, "NAME", "ops") + "+trace"
        self.steps = []          # (adv_in, grad, cost (1,), z (B, 1)) per update launch
        self.cw_l2 = []          # (B,) per CW iteration
        self.cw_adv = []         # (B, T) per CW iteration: 1/2 (tanh w + 1)
        self.cw_mask = []        # (B,) per CW iteration: the rows whose iterate the best-so-far blend took (cw.py:94-103)
        self._pending = None

    def __getattr__(self, name):
        return getattr(self.ops, name)

    def ce2_loss_grad(self, z, labels, scale=1.0):
        dz, loss = self.ops.ce2_loss_grad(z, labels, scale)
        self._pending = (loss.detach().clone(), z.detach().clone())
        return dz, loss

    def _note(self, adv, grad):
        cost, z = self._pending if self._pending is not None else (None, None)
        self.steps.append((adv.detach().clone(), grad.detach().clone(), cost, z))
        self._pending = None

    def pgd_linf_step(self, adv, grad, orig, *args, **kwargs):
        self._note(adv, grad)
        return self.ops.pgd_linf_step(adv, grad, orig, *args, **kwargs)

    def pgd_l2_step(self, adv, grad, orig, *args, **kwargs):
        self._note(adv, grad)
        return self.ops.pgd_l2_step(adv, grad, orig, *args, **kwargs)

    def fgsm_step(self, x, grad, *args, **kwargs):
        self._note(x, grad)
        return self.ops.fgsm_step(x, grad, *args, **kwargs)

    def cw_best_update(self, adv, mask, best):
        self.cw_mask.append(mask.detach().clone())
        return self.ops.cw_best_update(adv, mask, best)

    def cw_tanh_sqdist(self, w, x, adv_out=None):
        adv, l2 = self.ops.cw_tanh_sqdist(w, x, adv_out=adv_out)
        self.cw_l2.append(l2.detach().clone())
        self.cw_adv.append(adv.detach().clone())
        return adv, l2
