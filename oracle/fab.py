"""TEST INFRASTRUCTURE — CPU restatement of the reference's FAB attack.

Reference: adversarial_attacks/torchattacks/attacks/fab.py
    :562-614  projection_linf      :617-669  projection_l2      :672-717  projection_l1
    :90-112   get_diff_logits_grads_batch (one backward per logit column of cat([-z, z]))
    :131-307  attack_single_run    :495-559  perturb            :70-78    forward

The three projections solve, per row, "smallest ||d||_p with  w.(t + d) = b  and  0 <= t + d <= 1" (and the
nearest feasible corner move when the hyperplane does not cross the box).  The reference does it with
argsort / gather / cumsum / a log2(n)-step bisection over sorted breakpoints; this restatement solves the same
piecewise-linear equation from the sorted breakpoints in float64 (one stable sort, one prefix sum, one searchsorted
per row) — the exact solution both the reference (float32 sums + float64-accumulated cumsum on CPU, float32 parallel
scans on CUDA) and the HIP kernels (float32 tree sums, sort-free fixed-point iteration) approximate.  It is pinned
against outputs of the reference itself (tests/golden/fab_*.npz, tolerances stated in tests/test_oracle_golden.py):
floating-point parity, not bit parity — the reference's own CPU and CUDA paths differ in summation order too.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np
import torch

BIG = 1e12


def _orient(t, w, b):
    """fab.py:566-568 / :621-624 / :676-679: flip the hyperplane so the point lies on its non-negative side.
    The reference evaluates (w*t).sum(1) - b in float32; the float32 sum is kept so that the sign decision and the
    residual see the same cancellation the reference sees (b is typically -df + (w*t).sum())."""
    dot = (w.astype(np.float32) * t.astype(np.float32)).sum(axis=1, dtype=np.float32)
    c = dot - b.astype(np.float32)
    sg = np.where(c >= 0, 1.0, -1.0)
    return w.astype(np.float64) * sg[:, None], np.abs(c.astype(np.float64))


def _waterfill(weight, cap, target):
    """Per row: the level lam >= 0 with  sum_i weight_i * min(cap_i, lam) = target  (weight, cap >= 0), or +inf when
    even lam = inf falls short.  Sorted-breakpoint solution: with caps ascending, the equation is linear between
    consecutive caps."""
    R, n = weight.shape
    lam = np.full(R, np.inf)
    for r in range(R):
        order = np.argsort(cap[r], kind="stable")
        cs, ws = cap[r][order], weight[r][order]
        filled = np.concatenate(([0.0], np.cumsum(ws * cs)))          # mass of the k smallest caps, saturated
        tail = np.concatenate((np.cumsum(ws[::-1])[::-1], [0.0]))     # weight of the caps from position k on
        # value of the left-hand side at lam = cs[k]:  filled[k] + cs[k] * tail[k]
        at_cap = filled[:-1] + cs * tail[:-1]
        if not target[r] < filled[-1]:
            continue                                                  # unreachable inside the box (or NaN)
        k = int(np.searchsorted(at_cap, target[r], side="right"))     # caps [0, k) are saturated
        if k >= n or tail[k] <= 0:
            continue
        lam[r] = (target[r] - filled[k]) / tail[k]
    return lam


def projection_linf(points, w, b):
    """fab.py:562-614.  d_i = s_i * min(lam, p_i) with s_i = +1 where the flipped w_i < 0 else -1, p_i the room to the
    box face in that direction, and  sum |w_i| min(lam, p_i) = |w.t - b|."""
    t = points.astype(np.float64)
    wf, resid = _orient(points, w, b)
    toward_one = wf < 0
    room = np.where(toward_one, 1.0 - t, t)
    lam = _waterfill(np.abs(wf), room, resid)
    d = np.where(toward_one, 1.0, -1.0) * np.minimum(lam[:, None], room)
    return np.where(wf != 0, d, 0.0).astype(np.float32)


def projection_l2(points, w, b):
    """fab.py:617-669.  d_i = -w_i min(alpha, r_i), r_i = room_i / |w_i|, with  sum w_i^2 min(alpha, r_i) = |w.t - b|;
    coordinates with |w_i| < 1e-8 do not move (fab.py:627,638,669) but keep r = 1e12 in the sums."""
    t = points.astype(np.float64)
    wf, resid = _orient(points, w, b)
    live = np.abs(wf) >= 1e-8
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.maximum(t / wf, (t - 1.0) / wf)
    r = np.clip(r, -BIG, BIG)
    r[~live] = BIG
    r[r == -BIG] = BIG
    alpha = _waterfill(wf * wf, r, resid)
    d = -wf * np.minimum(alpha[:, None], r)
    return np.where(live, d, 0.0).astype(np.float32)


def projection_l1(points, w, b):
    """fab.py:672-717.  Coordinates move to their box face in order of decreasing |w| (stable in the index for equal
    |w|) while the residual stays positive; the coordinate that would overshoot moves by residual / w; the rest stay."""
    t = points.astype(np.float64)
    wf, resid = _orient(points, w, b)
    R, n = wf.shape
    with np.errstate(divide="ignore"):
        key = np.minimum(np.abs(1.0 / wf), BIG)
    face = np.where(wf < 0, 1.0, 0.0) - t                      # full move (fab.py:685-686)
    gain = np.minimum(-wf * t, wf * (1.0 - t))                  # change of the residual for a full move (<= 0)
    d = np.where(wf != 0, face, 0.0)
    for r in range(R):
        order = np.argsort(key[r], kind="stable")
        s = resid[r] + np.concatenate(([0.0], np.cumsum(gain[r][order])))   # residual before sorted position k
        if not s[-1] < 0:
            continue                                            # the box corner does not reach the hyperplane
        # fab.py:693-705: bisection for the last position whose residual-before is still positive (lb stays 0 if none)
        pos = np.nonzero(s[:-1] > 0)[0]
        lb = int(pos[-1]) if pos.size else 0
        row = np.zeros(n)
        row[order[:lb]] = d[r][order[:lb]]
        row[order[lb]] = -s[lb] / wf[r][order[lb]]
        d[r] = row
    return np.where(np.abs(wf) > 1e-8, d, 0.0).astype(np.float32)


PROJECTIONS = {"Linf": projection_linf, "L2": projection_l2, "L1": projection_l1}


def row_norm(v, norm):
    v = v.reshape(v.shape[0], -1)
    if norm == "Linf":
        return v.abs().max(dim=1)[0]
    if norm == "L2":
        return (v ** 2).sum(dim=1).sqrt()
    return v.abs().sum(dim=1)


def dual_norm(g, norm):
    """fab.py:211-222: the norm of the hyperplane normal that turns |df| into a distance."""
    g = g.reshape(g.shape[0], -1)
    if norm == "Linf":
        return g.abs().sum(dim=1)
    if norm == "L2":
        return (g ** 2).sum(dim=1).sqrt()
    return g.abs().max(dim=1)[0]


def logits2(model, x):
    z = model(x)
    return torch.cat([-z, z], dim=1)


def predicted(model, x):
    with torch.no_grad():
        return logits2(model, x).max(dim=1)[1]


def boundary_hyperplane(model, x1, la, norm):
    """fab.py:90-112 + :210-229 for the two-column logits cat([-z, z]): returns (w, b) of the linearised decision
    boundary closest to x1.  One backward pass gives gz; the column gradients are -gz and +gz exactly."""
    im = x1.clone().requires_grad_()
    with torch.enable_grad():
        z = model(im)
        gz = torch.autograd.grad(z.sum(), im)[0]
    y = torch.cat([-z, z], dim=1).detach()
    g2 = torch.stack([-gz, gz], dim=1)
    u = torch.arange(x1.shape[0])
    df = y - y[u, la].unsqueeze(1)
    dg = g2 - g2[u, la].unsqueeze(1)
    df[u, la] = 1e10
    dist = df.abs() / (1e-12 + torch.stack([dual_norm(dg[:, k], norm) for k in range(2)], dim=1))
    ind = dist.min(dim=1)[1]
    w = dg[u, ind]
    b = -df[u, ind] + (w * x1).reshape(x1.shape[0], -1).sum(dim=-1)
    return w, b


def fab_iteration(model, x1, x0, la, adv, res2, norm, eta, beta, alpha_max):
    """One pass of fab.py:208-292; returns the new (x1, adv, res2)."""
    bs = x1.shape[0]
    w, b = boundary_hyperplane(model, x1, la, norm)
    pts = torch.cat((x1, x0), 0).numpy()
    d3 = torch.from_numpy(PROJECTIONS[norm](pts, torch.cat((w, w), 0).numpy(), torch.cat((b, b), 0).numpy()))
    d1, d2 = d3[:bs], d3[bs:]
    a0 = torch.clamp_min(row_norm(d3, norm), 1e-8).unsqueeze(1)
    a1, a2 = a0[:bs], a0[bs:]
    alpha = torch.clamp(a1 / (a1 + a2), min=0.0, max=alpha_max)
    x1 = ((x1 + eta * d1) * (1 - alpha) + (x0 + d2 * eta) * alpha).clamp(0.0, 1.0)
    is_adv = predicted(model, x1) != la
    if is_adv.any():
        rows = is_adv.nonzero().reshape(-1)
        t = row_norm(x1[rows] - x0[rows], norm)
        better = t < res2[rows]
        adv = adv.clone()
        res2 = res2.clone()
        adv[rows[better]] = x1[rows[better]]
        res2[rows[better]] = t[better]
        x1 = x1.clone()
        x1[rows] = x0[rows] + (x1[rows] - x0[rows]) * beta
    return x1, adv, res2


def random_start(x0, res2, norm, eps, draw):
    """fab.py:174-205 with the draw (uniform(0,1) for Linf, normal otherwise) supplied by the caller."""
    radius = torch.minimum(res2, torch.full_like(res2, eps)).unsqueeze(1)
    if norm == "Linf":
        t = 2 * draw - 1
        x1 = x0 + radius * t / t.abs().max(dim=1, keepdim=True)[0] * 0.5
    elif norm == "L2":
        x1 = x0 + radius * draw / (draw ** 2).sum(dim=1, keepdim=True).sqrt() * 0.5
    else:
        x1 = x0 + radius * draw / draw.abs().sum(dim=1, keepdim=True) / 2
    return x1.clamp(0.0, 1.0)


def attack_single_run(model, x, y, norm="Linf", eps=0.3, steps=100, alpha_max=0.1, eta=1.05, beta=0.9, start_draw=None,
                      trace=None):
    """fab.py:131-307."""
    x = x.detach().clone().float()
    y = y.detach().clone().long()
    ok = predicted(model, x) == y
    if ok.sum() == 0:
        return x
    rows = ok.nonzero().reshape(-1)
    x0 = x[rows].clone()
    la = y[rows].clone()
    adv = x0.clone()
    res2 = torch.full((x0.shape[0],), 1e10)
    x1 = x0.clone() if start_draw is None else random_start(x0, res2, norm, eps, start_draw)
    for _ in range(steps):
        if trace is not None:
            trace.append(x1.clone())
        x1, adv, res2 = fab_iteration(model, x1, x0, la, adv, res2, norm, eta, beta, alpha_max)
    out = x.clone()
    won = res2 < 1e10
    out[rows[won]] = adv[won]
    return out


def fab(model, images, labels, norm="Linf", eps=None, steps=100, n_restarts=1, alpha_max=0.1, eta=1.05, beta=0.9,
        start_draws=None, trace=None):
    """fab.py:70-78 + :495-530 (untargeted: `self.targeted` is always False in the reference, fab.py:63)."""
    eps = {"Linf": 0.3, "L2": 1.0, "L1": 5.0}[norm] if eps is None else eps
    x = images.clone().detach()
    y = labels.clone().detach()
    adv = x.clone()
    with torch.no_grad():
        acc = logits2(model, x).max(1)[1] == y
        for counter in range(n_restarts):
            todo = acc.nonzero().reshape(-1)
            if todo.numel() == 0:
                continue
            xs, ys = x[todo].clone(), y[todo].clone()
            draw = None if counter == 0 else start_draws[counter - 1]
            cur = attack_single_run(model, xs, ys, norm, eps, steps, alpha_max, eta, beta, draw, trace)
            still = logits2(model, cur).max(1)[1] == ys
            if norm == "L1":
                # fab.py:518-522 never assigns `res` for L1: the reference raises UnboundLocalError here
                raise UnboundLocalError("local variable 'res' referenced before assignment")
            still = still | (row_norm(xs - cur, norm) > eps)
            fooled = (~still).nonzero().reshape(-1)
            acc[todo[fooled]] = False
            adv[todo[fooled]] = cur[fooled].clone()
    return adv
