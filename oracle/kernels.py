"""TEST INFRASTRUCTURE — numpy front-end of the plain-C oracle (oracle/advstep_oracle.c).

One function per C-ABI entry point of include/advstep.h, same argument meaning, host numpy arrays instead
of device pointers.  Reference citations live next to each C function.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i64, _f32, _f64, _u64 = ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_uint64
_optf32p = ctypes.c_void_p


def build(force: bool = False) -> Path:
    """Compile liboracle.so with gcc (seconds)."""
    so = _HERE / "liboracle.so"
    src = _HERE / "advstep_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(str(build()))
        sig = {
            "oracle_minmax_normalize_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _i64],
            "oracle_minmax_revert_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _i64],
            "oracle_fgsm_step_f32": [_f32p, _f32p, _f32p, _i64, _f32, _f32, _f32],
            "oracle_pgd_linf_init_noise_f32": [_f32p, _f32p, _f32p, _i64, _f32, _f32],
            "oracle_pgd_linf_step_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _f32, _f32, _f32, _f32],
            "oracle_pgd_l2_init_noise_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _f32, _f32, _f32],
            "oracle_pgd_l2_step_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _f32, _f32, _f32, _f32, _f32,
                                       _f32p, _f32p, _f32p],
            "oracle_cw_init_w_f32": [_f32p, _f32p, _i64],
            "oracle_cw_tanh_sqdist_f32": [_f32p, _f32p, _f32p, _f32p, _i64, _i64],
            "oracle_cw_adam_step_f32": [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _f64, _f64, _f64, _f64],
            "oracle_cw_best_update_f32": [_f32p, _f32p, _f32p, _i64, _i64],
            "oracle_ce2_loss_grad_f32": [_f32p, _i64p, _f32p, _f32p, _i64, _f32],
            "oracle_philox_raw": [_u64, _u64, _u64, _u32p],
            "oracle_pgd_linf_init_philox_f32": [_f32p, _f32p, _i64, _f32, _f32, _f32, _u64, _u64],
            "oracle_pgd_l2_init_philox_f32": [_f32p, _f32p, _i64, _i64, _f32, _f32, _f32, _u64, _u64, _f32p],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = None
        _LIB = L
    return _LIB


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def minmax_normalize(x):
    x = _c(x)
    B, T = x.shape
    out, mn, mx = np.empty_like(x), np.empty(B, np.float32), np.empty(B, np.float32)
    lib().oracle_minmax_normalize_f32(x, out, mn, mx, B, T)
    return out, mn, mx


def minmax_revert(x01, mn, mx):
    x01 = _c(x01)
    B, T = x01.shape
    out = np.empty_like(x01)
    lib().oracle_minmax_revert_f32(x01, _c(mn).reshape(-1), _c(mx).reshape(-1), out, B, T)
    return out


def fgsm_step(x, grad, eps, lo=0.0, hi=1.0):
    x = _c(x)
    out = np.empty_like(x)
    lib().oracle_fgsm_step_f32(x, _c(grad), out, x.size, eps, lo, hi)
    return out


def pgd_linf_init_noise(x, noise, lo=0.0, hi=1.0):
    x = _c(x)
    out = np.empty_like(x)
    lib().oracle_pgd_linf_init_noise_f32(x, _c(noise), out, x.size, lo, hi)
    return out


def pgd_linf_init_philox(x, eps, seed, offset, lo=0.0, hi=1.0):
    x = _c(x)
    out = np.empty_like(x)
    lib().oracle_pgd_linf_init_philox_f32(x, out, x.size, eps, lo, hi, seed, offset)
    return out


def pgd_linf_step(adv, grad, orig, alpha, eps, lo=0.0, hi=1.0):
    adv = _c(adv)
    out = np.empty_like(adv)
    lib().oracle_pgd_linf_step_f32(adv, _c(grad), _c(orig), out, adv.size, alpha, eps, lo, hi)
    return out


def pgd_l2_init_noise(x, normal, r, eps, lo=0.0, hi=1.0):
    x = _c(x)
    B, T = x.shape
    out = np.empty_like(x)
    lib().oracle_pgd_l2_init_noise_f32(x, _c(normal), _c(r).reshape(-1), out, B, T, eps, lo, hi)
    return out


def pgd_l2_init_philox(x, eps, seed, offset, lo=0.0, hi=1.0):
    x = _c(x)
    B, T = x.shape
    out = np.empty_like(x)
    lib().oracle_pgd_l2_init_philox_f32(x, out, B, T, eps, lo, hi, seed, offset, np.empty(T, np.float32))
    return out


def pgd_l2_step(adv, grad, orig, alpha, eps, eps_div=1e-10, lo=0.0, hi=1.0):
    adv = _c(adv)
    B, T = adv.shape
    out, gn, dn = np.empty_like(adv), np.empty(B, np.float32), np.empty(B, np.float32)
    lib().oracle_pgd_l2_step_f32(adv, _c(grad), _c(orig), out, B, T, alpha, eps, eps_div, lo, hi, gn, dn,
                                 np.empty(T, np.float32))
    return out, gn, dn


def cw_init_w(x):
    x = _c(x)
    w = np.empty_like(x)
    lib().oracle_cw_init_w_f32(x, w, x.size)
    return w


def cw_tanh_sqdist(w, x):
    w = _c(w)
    B, T = w.shape
    adv, l2 = np.empty_like(w), np.empty(B, np.float32)
    lib().oracle_cw_tanh_sqdist_f32(w, _c(x), adv, l2, B, T)
    return adv, l2


def cw_adam_step(w, m, v, x, grad_adv, step, lr=0.01, beta1=0.9, beta2=0.999, adam_eps=1e-8):
    w, m, v = _c(w).copy(), _c(m).copy(), _c(v).copy()
    lib().oracle_cw_adam_step_f32(w, m, v, _c(x), _c(grad_adv), w.size, step, lr, beta1, beta2, adam_eps)
    return w, m, v


def cw_best_update(adv, mask, best):
    adv = _c(adv)
    B, T = adv.shape
    best = _c(best).copy()
    lib().oracle_cw_best_update_f32(adv, _c(mask).reshape(-1), best, B, T)
    return best


def ce2_loss_grad(z, labels, scale=1.0):
    z = _c(z).reshape(-1)
    y = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
    dz, loss = np.empty_like(z), np.empty(1, np.float32)
    lib().oracle_ce2_loss_grad_f32(z, y, dz, loss, z.size, scale)
    return dz, float(loss[0])


def philox_raw(counter_lo: int, counter_hi: int, seed: int) -> np.ndarray:
    out = np.empty(4, np.uint32)
    lib().oracle_philox_raw(counter_lo, counter_hi, seed, out)
    return out
