"""Final aggregate of the evaluation loop (reference: src/metrics.py:9-14 and the sklearn calls of
evaluate_models_on_adversarial_attacks.py:267-293), restated in numpy so the GPU box needs neither sklearn nor
scipy.  Each function documents the library routine it reproduces; tests pin them against values the
reference's own calls produced (tests/golden/metrics.npz)."""
from typing import Tuple

import numpy as np


def roc_curve(y_true: np.ndarray, y_score: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """sklearn.metrics.roc_curve(y_true, y_score) with drop_intermediate=True, positive label 1."""
    y_true = np.asarray(y_true).ravel() == 1
    y_score = np.asarray(y_score, dtype=np.float64).ravel()
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    thresholds = y_score[idx]
    if len(fps) > 2:  # drop collinear interior points
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thresholds = fps[keep], tps[keep], thresholds[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thresholds = np.r_[np.inf, thresholds]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full_like(fps, np.nan)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full_like(tps, np.nan)
    return fpr, tpr, thresholds


def _interp_linear(xs: np.ndarray, ys: np.ndarray, x: float) -> float:
    """scipy.interpolate.interp1d(xs, ys)(x) for 1-D linear interpolation, which scipy evaluates with np.interp:
    on a run of equal xs (vertical ROC segments) the LAST point of the run is used."""
    return float(np.interp(x, xs, ys))


def _brentq(f, a: float, b: float, xtol: float = 2e-12, rtol: float = 8.881784197001252e-16, maxiter: int = 100):
    """scipy.optimize.brentq (Brent 1973, as in scipy/optimize/Zeros/brentq.c)."""
    xpre, xcur = a, b
    fpre, fcur = f(xpre), f(xcur)
    if fpre == 0:
        return xpre
    if fcur == 0:
        return xcur
    if np.sign(fpre) == np.sign(fcur):
        raise ValueError("f(a) and f(b) must have different signs")
    xblk = fblk = spre = scur = 0.0
    for _ in range(maxiter):
        if fpre != 0 and fcur != 0 and np.sign(fpre) != np.sign(fcur):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur, xblk = xcur, xblk, xcur
            fpre, fcur, fblk = fcur, fblk, fcur
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)  # secant
            else:  # inverse quadratic extrapolation
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre = scur = sbis
        else:
            spre = scur = sbis
        xpre, fpre = xcur, fcur
        xcur += scur if abs(scur) > delta else (delta if sbis > 0 else -delta)
        fcur = f(xcur)
    raise RuntimeError("brentq failed to converge")


def calculate_eer(y, y_score) -> Tuple[float, float, np.ndarray, np.ndarray]:
    """src/metrics.py:9-14: ROC of (y, -y_score); EER = root of 1 - x - tpr(x) on [0, 1]; threshold at the EER."""
    fpr, tpr, thresholds = roc_curve(y, -np.asarray(y_score, dtype=np.float64))
    eer = _brentq(lambda x: 1.0 - x - _interp_linear(fpr, tpr, x), 0.0, 1.0)
    thresh = _interp_linear(fpr, thresholds, eer)
    return thresh, eer, fpr, tpr


def roc_auc_score(y_true, y_score) -> float:
    """sklearn.metrics.roc_auc_score for binary labels: trapezoidal area under roc_curve(drop_intermediate=False)."""
    y_true = np.asarray(y_true).ravel() == 1
    y_score = np.asarray(y_score, dtype=np.float64).ravel()
    if y_true.all() or not y_true.any():
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    idx = np.r_[np.where(np.diff(y_score))[0], y_true.size - 1]
    tps = np.r_[0, np.cumsum(y_true, dtype=np.float64)[idx]]
    fps = np.r_[0, 1 + idx - tps[1:]]
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(trapezoid(tps / tps[-1], fps / fps[-1]))


def precision_recall_f1_binary(y_true, y_pred) -> Tuple[float, float, float]:
    """sklearn.metrics.precision_recall_fscore_support(average='binary', beta=1.0), positive label 1
    (zero_division -> 0.0, as sklearn's default 'warn' returns)."""
    y_true = np.asarray(y_true).ravel() == 1
    y_pred = np.asarray(y_pred).ravel() == 1
    tp = float(np.sum(y_true & y_pred))
    pred_pos, true_pos = float(np.sum(y_pred)), float(np.sum(y_true))
    precision = tp / pred_pos if pred_pos > 0 else 0.0
    recall = tp / true_pos if true_pos > 0 else 0.0
    f1 = 2 * precision * recall / (precision + recall) if (precision + recall) > 0 else 0.0
    return precision, recall, f1


def adversarial_report(y: np.ndarray, y_pred: np.ndarray, y_pred_label: np.ndarray) -> dict:
    """The six numbers of the reference's final log line (evaluate_models_on_adversarial_attacks.py:267-298)."""
    y = np.asarray(y).ravel()
    accuracy = 100.0 * float(np.sum(np.asarray(y_pred_label).ravel() == y)) / max(len(y), 1)
    precision, recall, f1 = precision_recall_f1_binary(y, y_pred_label)
    auc = roc_auc_score(y, y_pred)
    _, eer, _, _ = calculate_eer(y=1 - y, y_score=y_pred)  # "For EER flip values" (:282-283)
    return {"adv_eval/eer": eer, "adv_eval/accuracy": accuracy, "adv_eval/precision": precision,
            "adv_eval/recall": recall, "adv_eval/f1_score": f1, "adv_eval/auc": auc}
