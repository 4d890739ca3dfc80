"""End-to-end comparison of the product's iterated attacks against the CPU oracle ON THE REAL DETECTORS (SURVEY.md
section 7 "Parity definition", items (iii) / (iv); reference pgd.py:59-76, pgdl2.py:64-88, cw.py:70-110).

The product attack (Attack subclass + op table, any device) and the oracle (oracle/attacks.py on a CPU copy of the SAME
weights, the SAME recorded random start) run the whole attack; both sides keep, per iteration, the iterate, the input
gradient, the loss and the logits.  What is compared:

  free-running     the two trajectories as they are: loss per iteration, share of identical samples / max-abs distance of
                   the iterates, sign agreement of the two gradients, final waveform, scores / labels / accuracy / EER
  teacher-forced   the product's gradient AT THE ORACLE'S iterate of iteration k (so earlier differences cannot
                   compound): sign agreement, where the flips sit (|grad| relative to the utterance's largest), loss

Shared by tests/test_gpu_e2e_parity.py (MI355X: hip_ops) and tests/test_e2e_parity_harness.py (CPU: the product's host
logic with the oracle's op table, where everything must agree exactly)."""
import copy
from contextlib import contextmanager

import numpy as np
import torch

from audio_deepfake_adversarial_attacks_amd import metrics
from audio_deepfake_adversarial_attacks_amd.torchattacks.attack import Attack
from oracle import attacks as OA
from oracle.checked_ops import TracingOps


@contextmanager
def cpu_threads(n):
    """The oracle's CPU passes: a bounded thread count (the GPU box has 256 hardware threads; small batches run slower
    on all of them than on 16)."""
    before = torch.get_num_threads()
    torch.set_num_threads(min(n, before) if before else n)
    try:
        yield
    finally:
        torch.set_num_threads(before)


class GradProbe(Attack):
    """The product's forward + input-backward of one iterate, through Attack.__call__ (mode juggling, parameter freeze,
    closed-form loss gradient): returns (grad, cost, z)."""

    def __init__(self, model):
        super().__init__("GradProbe", model)
        self.z = None

    def forward(self, images, labels):
        images, labels, target = self._prepare(images, labels)
        hook = self.model.register_forward_hook(lambda m, i, o: setattr(self, "z", o.detach().clone()))
        try:
            grad, cost = self._input_gradient(images, labels, target)
        finally:
            hook.remove()
        return grad, cost, self.z


def armed(atk, ops=None, noise=None):
    atk.set_training_mode(model_training=True, batchnorm_training=False)      # evaluate_...:170
    if ops is not None:
        atk.ops = ops
    if noise is not None:
        atk.set_init_noise(noise)
    return atk


def _f(t):
    return float(t.detach().cpu().reshape(-1)[0]) if isinstance(t, torch.Tensor) else float(t)


def score_both(target_dev, target_cpu, got, want, y):
    """evaluate_...:236-238 on both sides + the report of :267-298 over this batch."""
    from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
    p_dev, l_dev = score_batch(target_dev.eval(), got)
    p_cpu, l_cpu = score_batch(target_cpu.eval(), want)
    p_dev, l_dev = p_dev.cpu(), l_dev.cpu()
    fig = {"score_max_abs": (p_dev - p_cpu).abs().max().item(), "labels_equal": bool(torch.equal(l_dev, l_cpu)),
           "accuracy_product": (l_dev == y.int()).float().mean().item(),
           "accuracy_oracle": (l_cpu == y.int()).float().mean().item()}
    if 0 < int(y.sum()) < len(y):
        fig["eer_product"] = float(metrics.calculate_eer(y.numpy(), p_dev.numpy())[1])
        fig["eer_oracle"] = float(metrics.calculate_eer(y.numpy(), p_cpu.numpy())[1])
        fig["eer_abs_diff"] = abs(fig["eer_product"] - fig["eer_oracle"])
    return fig


def _flip_stats(g_dev, g_cpu):
    same = g_dev.sign() == g_cpu.sign()
    rel = g_cpu.abs() / g_cpu.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    flipped = rel[~same]
    return same, {"agreement": same.float().mean().item(), "flips": int((~same).sum()),
                  "flip_rel_worst": flipped.max().item() if flipped.numel() else 0.0}


def nudged(draw):
    """The same random start moved by ONE float32 ulp towards zero at every sample: the smallest change any other
    arithmetic (another summation order, another thread count, another device) can make to an iterate."""
    if isinstance(draw, (tuple, list)):
        return (nudged(draw[0]),) + tuple(draw[1:])
    return torch.nextafter(draw, torch.zeros_like(draw))


def _divergence(a01, b01, x01):
    d = (a01 - b01).abs()
    pert = (b01 - x01).abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    return {"differing_samples": (a01 != b01).float().mean().item(), "max_abs": d.max().item(),
            "max_abs_over_row_linf": (d / pert).max().item(),
            "rel_l2": (((a01 - b01).norm(dim=1)) / (b01 - x01).norm(dim=1).clamp_min(1e-30)).max().item()}


def run_gradient_attack(kind, model, target, ops, x, y, hyper, draw, device, forced_every=4, threads=16,
                        self_sensitivity=True):
    """kind = "PGD" | "PGDL2".  model / target: the attacked and the scoring detector on `device` (may be the same
    object); x (B, T) raw waveforms and y on the CPU.  Returns (figures, product adv01 on cpu, oracle adv01).
    self_sensitivity: also run the ORACLE from the start moved by one ulp (`nudged`) — how far the reference's own
    trajectory moves under a last-bit change is the scale the free-running figures are judged against."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    steps = hyper["steps"]
    model_cpu = copy.deepcopy(model).cpu()
    target_cpu = model_cpu if target is model else copy.deepcopy(target).cpu()

    # ---- the oracle's run (CPU) ------------------------------------------------------------------------------------------
    x01, mn, mx = OA.to_minmax(x)
    trace = []
    with cpu_threads(threads), OA.attack_mode(model_cpu):
        if kind == "PGD":
            want01 = OA.pgd(model_cpu, x01, y, noise=draw, trace=trace, **hyper)
        else:
            want01 = OA.pgdl2(model_cpu, x01, y, draws=draw, trace=trace, **hyper)
    want = OA.revert_minmax(want01, mn, mx)
    own01 = own_trace = None
    if self_sensitivity:
        own_trace = []
        with cpu_threads(threads), OA.attack_mode(model_cpu):
            if kind == "PGD":
                own01 = OA.pgd(model_cpu, x01, y, noise=nudged(draw), trace=own_trace, **hyper)
            else:
                own01 = OA.pgdl2(model_cpu, x01, y, draws=nudged(draw), trace=own_trace, **hyper)

    # ---- the product's run ------------------------------------------------------------------------------------------------
    tops = TracingOps(ops)
    cls = torchattacks.PGD if kind == "PGD" else torchattacks.PGDL2
    atk = armed(cls(model, **hyper), tops, draw)
    xg, yg = x.to(device), y.to(device)
    g01, gmn, gmx = ops.to_minmax(xg)
    assert torch.equal(g01.cpu(), x01), "to_minmax differs"
    got01 = atk(g01, yg)
    got = ops.revert_minmax(got01, gmn, gmx)
    assert len(tops.steps) == steps == len(trace)

    fig = {"kind": kind, "batch": int(x.shape[0]), "steps": steps, "hyper": {k: float(v) for k, v in hyper.items()}}
    free = {"loss_rel": [], "identical_samples": [], "iterate_max_abs": [], "grad_sign_agreement": [], "grad_rel_l2": [],
            "logit_max_abs": []}
    for (a_d, g_d, c_d, z_d), (a_c, g_c, c_c, z_c) in zip(tops.steps, trace):
        a_d, g_d = a_d.cpu(), g_d.cpu()
        free["loss_rel"].append(abs(_f(c_d) - _f(c_c)) / abs(_f(c_c)))
        free["identical_samples"].append((a_d == a_c).float().mean().item())
        free["iterate_max_abs"].append((a_d - a_c).abs().max().item())
        free["grad_sign_agreement"].append((g_d.sign() == g_c.sign()).float().mean().item())
        free["grad_rel_l2"].append(((g_d - g_c).norm() / g_c.norm()).item())
        free["logit_max_abs"].append((z_d.cpu() - z_c).abs().max().item())
    fig["free_running"] = {"per_iteration": free,
                           "loss_rel_worst": max(free["loss_rel"]), "logit_max_abs_worst": max(free["logit_max_abs"]),
                           "identical_samples_worst": min(free["identical_samples"]),
                           "iterate_max_abs_worst": max(free["iterate_max_abs"]),
                           "grad_sign_agreement_worst": min(free["grad_sign_agreement"]),
                           "grad_rel_l2_worst": max(free["grad_rel_l2"])}

    # ---- teacher-forced: the product's gradient and update at the oracle's iterates ---------------------------------------------
    probe = armed(GradProbe(model), ops)
    forced = {"iteration": [], "loss_rel": [], "logit_max_abs": [], "grad_sign_agreement": [], "flips": [],
              "flip_rel_worst": [], "grad_rel_l2": [], "update_max_abs_on_agreeing": [], "update_max_abs": [],
              "update_given_oracle_grad_max_abs": []}
    orig = x01.to(device)
    for k in list(range(0, steps, forced_every)) + ([steps - 1] if (steps - 1) % forced_every else []):
        a_c, g_c, c_c, z_c = trace[k]
        g_d, c_d, z_d = probe(a_c.to(device), yg)
        same, st = _flip_stats(g_d.cpu(), g_c)
        nxt_c = trace[k + 1][0] if k + 1 < steps else want01
        if kind == "PGD":
            nxt_d = ops.pgd_linf_step(a_c.to(device), g_d, orig, hyper["alpha"], hyper["eps"]).cpu()
            agree_err = (nxt_d - nxt_c).abs()[same].max().item()
        else:
            nxt_d = ops.pgd_l2_step(a_c.to(device), g_d, orig, hyper["alpha"], hyper["eps"],
                                    hyper.get("eps_for_division", 1e-10)).cpu()
            agree_err = (nxt_d - nxt_c).abs().max().item()
        g_up = g_c.to(device)
        if kind == "PGD":
            same_grad = ops.pgd_linf_step(a_c.to(device), g_up, orig, hyper["alpha"], hyper["eps"]).cpu()
        else:
            same_grad = ops.pgd_l2_step(a_c.to(device), g_up, orig, hyper["alpha"], hyper["eps"],
                                        hyper.get("eps_for_division", 1e-10)).cpu()
        forced["update_given_oracle_grad_max_abs"].append((same_grad - nxt_c).abs().max().item())
        forced["iteration"].append(k)
        forced["loss_rel"].append(abs(_f(c_d) - _f(c_c)) / abs(_f(c_c)))
        forced["logit_max_abs"].append((z_d.cpu() - z_c).abs().max().item())
        forced["grad_sign_agreement"].append(st["agreement"]), forced["flips"].append(st["flips"])
        forced["flip_rel_worst"].append(st["flip_rel_worst"])
        forced["grad_rel_l2"].append(((g_d.cpu() - g_c).norm() / g_c.norm()).item())
        forced["update_max_abs_on_agreeing"].append(agree_err)
        forced["update_max_abs"].append((nxt_d - nxt_c).abs().max().item())
    fig["teacher_forced"] = {"per_iteration": forced, "loss_rel_worst": max(forced["loss_rel"]),
                             "logit_max_abs_worst": max(forced["logit_max_abs"]),
                             "grad_sign_agreement_worst": min(forced["grad_sign_agreement"]),
                             "flip_rel_worst": max(forced["flip_rel_worst"]),
                             "grad_rel_l2_worst": max(forced["grad_rel_l2"]),
                             "update_max_abs_on_agreeing_worst": max(forced["update_max_abs_on_agreeing"]),
                             "update_given_oracle_grad_max_abs_worst": max(forced["update_given_oracle_grad_max_abs"])}

    # ---- the final waveform and what the target makes of it -------------------------------------------------------------------
    got01c = got01.cpu()
    d_dev, d_cpu = got01c - x01, want01 - x01
    fig["final"] = {"identical_samples": (got01c == want01).float().mean().item(),
                    "max_abs": (got01c - want01).abs().max().item(),
                    "perturbation_sign_agreement": (d_dev.sign() == d_cpu.sign()).float().mean().item(),
                    "linf_product": d_dev.abs().max().item(), "linf_oracle": d_cpu.abs().max().item(),
                    "l2_product_max": d_dev.norm(dim=1).max().item(), "l2_oracle_max": d_cpu.norm(dim=1).max().item(),
                    "l2_rel_diff_worst": ((d_dev.norm(dim=1) - d_cpu.norm(dim=1)).abs() / d_cpu.norm(dim=1)).max().item(),
                    "box_ok": bool(got01c.min() >= 0 and got01c.max() <= 1)}
    fig["final"]["divergence_product_vs_oracle"] = _divergence(got01c, want01, x01)
    if own01 is not None:
        fig["final"]["divergence_oracle_vs_oracle_one_ulp"] = _divergence(own01, want01, x01)
        fig["free_running"]["oracle_one_ulp"] = {
            "loss_rel_worst": max(abs(_f(a[2]) - _f(b[2])) / abs(_f(b[2])) for a, b in zip(own_trace, trace)),
            "grad_sign_agreement_worst": min((a[1].sign() == b[1].sign()).float().mean().item()
                                             for a, b in zip(own_trace, trace)),
            "grad_rel_l2_worst": max(((a[1] - b[1]).norm() / b[1].norm()).item() for a, b in zip(own_trace, trace)),
            "iterate_max_abs_worst": max((a[0] - b[0]).abs().max().item() for a, b in zip(own_trace, trace)),
            "logit_max_abs_worst": max((a[3] - b[3]).abs().max().item() for a, b in zip(own_trace, trace))}
    with cpu_threads(threads):
        fig["target"] = score_both(target, target_cpu, got, want, y)
        if own01 is not None:
            from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
            p_own, _ = score_batch(target_cpu.eval(), OA.revert_minmax(own01, mn, mx))
            p_want, _ = score_batch(target_cpu.eval(), want)
            fig["target"]["score_max_abs_oracle_one_ulp"] = (p_own - p_want).abs().max().item()
    return fig, got01c, want01


def _cw_f(z, y, kappa):
    """cw.py:125-134 on outputs = cat([-z, z]): f = clamp(out_true - max((1 - onehot) * out), min=-kappa).  The masked
    maximum runs over {out_other, 0} — the true class's entry is zeroed, not removed (kept as in the reference)."""
    yy = y.float() * 2 - 1
    true, other = yy * z, -yy * z
    return torch.clamp(true - torch.clamp(other, min=0), min=-kappa)


def _cw_model_grad(model, adv, y, c, kappa):
    """d (c * sum f) / d adv on the CPU (cw.py:79-89 without the distance term) — the oracle side of the teacher-forced
    gradient comparison."""
    adv = adv.clone().requires_grad_(True)
    z = model(adv)
    outputs = torch.cat([-z, z], dim=1)
    loss = c * OA._cw_f(outputs, y, kappa).sum()
    (g,) = torch.autograd.grad(loss, adv)
    return g, z.detach().reshape(-1)


class CWGradProbe(Attack):
    """The product's side of the same gradient (its CW.forward's autograd call, through Attack.__call__)."""

    def __init__(self, model, c, kappa):
        super().__init__("CWGradProbe", model)
        self.c, self.kappa = c, kappa

    def forward(self, adv, labels):
        from audio_deepfake_adversarial_attacks_amd.torchattacks import CW
        adv, labels, _ = self._prepare(adv, labels)
        adv.requires_grad_(True)
        z = self.model(adv)
        outputs = torch.cat([-z, z], dim=1)
        f_loss = CW.f(self, outputs, labels).sum()
        (g,) = torch.autograd.grad(self.c * f_loss, adv)
        return g, z.detach().reshape(-1)


def _cw_compare(trace_a, l2_a, z_a, adv_a, trace_b, hyper, y):
    """Per-iteration figures of run a (lists of tensors) against the oracle trace b."""
    n = min(len(l2_a), len(trace_b))
    per = {"cost_rel": [], "l2_rel_worst": [], "logit_max_abs": [], "adv_max_abs": [], "adv_mean_abs": []}
    for k in range(n):
        cost_c, l2_c, z_c, adv_c = trace_b[k]
        l2_d, z_d = l2_a[k], z_a[k]
        cost_d = l2_d.sum() + hyper["c"] * _cw_f(z_d, y, hyper.get("kappa", 0)).sum()
        # iteration 0: adv = 1/2 (tanh(atanh(2x - 1)) + 1) = x up to rounding, so its squared distance (~1e-11) is rounding
        # noise on both sides; relative errors are taken against a floor well above that
        per["cost_rel"].append(abs(_f(cost_d) - _f(cost_c)) / max(abs(_f(cost_c)), 1e-3))
        per["l2_rel_worst"].append(((l2_d - l2_c).abs() / l2_c.clamp_min(1e-6)).max().item())
        per["logit_max_abs"].append((z_d - z_c).abs().max().item())
        per["adv_max_abs"].append((adv_a[k] - adv_c).abs().max().item())
        per["adv_mean_abs"].append((adv_a[k] - adv_c).abs().mean().item())
    return {"per_iteration": per, "cost_rel_worst": max(per["cost_rel"]), "l2_rel_worst": max(per["l2_rel_worst"]),
            "logit_max_abs_worst": max(per["logit_max_abs"]), "adv_max_abs_worst": max(per["adv_max_abs"]),
            "adv_mean_abs_worst": max(per["adv_mean_abs"])}


def run_cw(model, target, ops, x, y, hyper, device, threads=16, forced_at=(0, 5, 10), self_sensitivity=True):
    """CW on `model` (attacked) scored by `target`; hyper = dict(c, kappa, steps, lr).  Compares the iteration count at
    which cw.py:107-110 stops, the per-iteration cost / squared distances / logits / iterates, teacher-forced model
    gradients at the oracle's iterates, the final best adversarials and the target's scores.  self_sensitivity: the oracle
    against itself from inputs moved by one ulp (Adam turns rounding-level gradient entries into +-lr moves of arbitrary
    sign, in the reference too: that run shows how much of the free-running difference is the attack's own)."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    model_cpu = copy.deepcopy(model).cpu()
    target_cpu = copy.deepcopy(target).cpu()
    x01, mn, mx = OA.to_minmax(x)
    trace = []
    with cpu_threads(threads), OA.attack_mode(model_cpu):
        want01 = OA.cw(model_cpu, x01, y, trace=trace, **hyper)
    want = OA.revert_minmax(want01, mn, mx)

    tops = TracingOps(ops)
    atk = armed(torchattacks.CW(model, **hyper), tops)
    zs = []
    hook = model.register_forward_hook(lambda m, i, o: zs.append(o.detach().clone()))
    try:
        xg, yg = x.to(device), y.to(device)
        g01, gmn, gmx = ops.to_minmax(xg)
        got01 = atk(g01, yg)
    finally:
        hook.remove()
    got = ops.revert_minmax(got01, gmn, gmx)

    fig = {"kind": "CW", "batch": int(x.shape[0]), "hyper": {k: float(v) for k, v in hyper.items()},
           "iterations_product": len(tops.cw_l2), "iterations_oracle": len(trace)}
    fig["free_running"] = _cw_compare(None, [t.cpu() for t in tops.cw_l2], [z.cpu().reshape(-1) for z in zs],
                                      [a.cpu() for a in tops.cw_adv], trace, hyper, y)
    if self_sensitivity:
        own = []
        with cpu_threads(threads), OA.attack_mode(model_cpu):
            OA.cw(model_cpu, torch.nextafter(x01, torch.full_like(x01, 0.5)), y, trace=own, **hyper)
        fig["iterations_oracle_one_ulp"] = len(own)
        fig["free_running"]["oracle_one_ulp"] = {
            k: v for k, v in _cw_compare(None, [t[1] for t in own], [t[2] for t in own], [t[3] for t in own], trace, hyper,
                                         y).items() if k != "per_iteration"}

    # teacher-forced: the model-term gradient at the oracle's iterates
    probe = armed(CWGradProbe(model, hyper["c"], hyper.get("kappa", 0)), ops)
    forced = {"iteration": [], "logit_max_abs": [], "grad_rel_l2": [], "grad_rel_err_median": [], "grad_sign_agreement": [],
              "oracle_one_ulp_grad_rel_l2": [], "oracle_one_ulp_grad_sign_agreement": []}
    for k in [k for k in forced_at if k < len(trace)]:
        adv_c = trace[k][3]
        with cpu_threads(threads), OA.attack_mode(model_cpu):
            g_c, z_c = _cw_model_grad(model_cpu, adv_c, y, hyper["c"], hyper.get("kappa", 0))
            if self_sensitivity:      # the oracle's own gradient one ulp away from the same iterate
                g_u, _ = _cw_model_grad(model_cpu, torch.nextafter(adv_c, torch.full_like(adv_c, 0.5)), y, hyper["c"],
                                        hyper.get("kappa", 0))
                if (g_c.abs().amax(dim=1) > 0).any():
                    lv = g_c.abs().amax(dim=1) > 0
                    forced["oracle_one_ulp_grad_rel_l2"].append(((g_u[lv] - g_c[lv]).norm() / g_c[lv].norm()).item())
                    forced["oracle_one_ulp_grad_sign_agreement"].append((g_u[lv].sign() == g_c[lv].sign()).float().mean().item())
        g_d, z_d = probe(adv_c.to(device), yg)
        g_d, z_d = g_d.cpu(), z_d.cpu()
        live = g_c.abs().amax(dim=1) > 0                 # rows whose f is clamped have a zero model gradient on both sides
        forced["iteration"].append(k)
        forced["logit_max_abs"].append((z_d - z_c).abs().max().item())
        if live.any():
            forced["grad_rel_l2"].append(((g_d[live] - g_c[live]).norm() / g_c[live].norm()).item())
            forced["grad_rel_err_median"].append(((g_d[live] - g_c[live]).abs() / g_c[live].abs().clamp_min(1e-30)).median().item())
            forced["grad_sign_agreement"].append((g_d[live].sign() == g_c[live].sign()).float().mean().item())
        assert torch.equal(g_d[~live], g_c[~live])
    fig["teacher_forced"] = {"per_iteration": forced, "logit_max_abs_worst": max(forced["logit_max_abs"]),
                             "grad_rel_l2_worst": max(forced["grad_rel_l2"], default=0.0),
                             "grad_rel_err_median_worst": max(forced["grad_rel_err_median"], default=0.0),
                             "grad_sign_agreement_worst": min(forced["grad_sign_agreement"], default=1.0)}

    got01c = got01.cpu()
    err = (got01c - want01).abs()
    fig["final"] = {"mean_abs": err.mean().item(), "max_abs": err.max().item(),
                    "frac_off_by_1e-4": (err > 1e-4).float().mean().item(),
                    "rows_changed_product": ((got01c - x01).abs().amax(1) > 1e-6).tolist(),
                    "rows_changed_oracle": ((want01 - x01).abs().amax(1) > 1e-6).tolist(),
                    "box_ok": bool(got01c.min() >= 0 and got01c.max() <= 1)}
    with cpu_threads(threads):
        fig["target"] = score_both(target, target_cpu, got, want, y)
    return fig, got01c, want01


def slim(fig):
    """The record kept under profiles/: per-iteration lists rounded, nothing else dropped."""
    def r(v):
        if isinstance(v, dict):
            return {k: r(x) for k, x in v.items()}
        if isinstance(v, list):
            return [r(x) for x in v]
        if isinstance(v, float):
            return float(np.format_float_scientific(v, precision=3)) if v != 0 else 0.0
        return v
    return r(fig)
