"""`-m gpu`: the iterated attacks END TO END on the real detectors at BASELINE.json's settings — the shipped GPU path
(hip_ops kernels + the fused model kernels) against the CPU oracle (oracle/attacks.py on a CPU copy of the same weights,
same recorded random start).  Protocol and figures: tests/e2e_parity.py; SURVEY.md section 7 "Parity definition".

    configs[1]  LCNN + LFCC, PGD L-inf eps = 0.003, alpha = 2/255, 40 iterations, B = 8      pgd.py:59-76
    configs[2]  SpecRNet + mel-spec, PGDL2 eps = 0.1, alpha = 0.2, 40 iterations, B = 8        pgdl2.py:64-88
    configs[3]  RawNet3 -> LCNN + LFCC, CW c = 1, lr = 0.01, steps = 100 (stops on its own cost), B = 2   cw.py:70-110
    configs[1] at full size: B = 128, PGD-40, every 10th launch of every kernel re-computed by the C oracle

What "stated tolerance" means here.  sign() and the detectors' max-feature-map / max-pool / LeakyReLU routing are
discontinuous: an iterate that differs in its last bit re-routes a winner within a few iterations, and from there the two
trajectories are different (equally valid) runs of the same attack — the ORACLE does this to itself when its random start
is moved by one float32 ulp (`*_one_ulp` figures, recorded next to the product's).  So the element-wise bounds are asserted
TEACHER-FORCED (the product's gradient and update at the oracle's own iterates, every 4th iteration of the real
trajectory), and the free-running runs are bound by what the attack is for: loss per iteration, logits, the target's
scores, predicted labels, accuracy, EER, the eps-ball / box invariants, and a divergence of the same order as the oracle's
own.  Every bound is <= 3x the figure measured on MI355X (profiles/r03_parity.json / r04_parity.json; round 3: 10x)."""
import pytest
import torch

from tests import e2e_parity as E

pytestmark = pytest.mark.gpu


def _model(name, cfg, cuda, seed=0):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(seed)
    return get_model(name, cfg, str(cuda)).to(cuda).eval()


@pytest.fixture(scope="module")
def hip(cuda):
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def lcnn(cuda):
    return _model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, cuda)


def _batch(n, seed):
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    x, y = synthetic_waveforms(n, seed=seed)
    y[: n // 2], y[n // 2:] = 0, 1               # both classes present: the batch has an EER
    return x, y


def test_pgd40_on_lcnn_lfcc_matches_cpu_oracle(cuda, hip, lcnn, parity_record):
    """configs[1] at B = 8.  PGD with alpha = 2/255 > 2 eps is bang-bang: the iterate after a step is
    clamp(x +- eps) by the gradient's sign alone, so `identical_samples` is the share of equal gradient signs."""
    x, y = _batch(8, 1234)
    hyper = dict(eps=0.003, alpha=2 / 255, steps=40)
    noise = torch.empty_like(x).uniform_(-hyper["eps"], hyper["eps"], generator=torch.Generator().manual_seed(77))
    fig, got01, want01 = E.run_gradient_attack("PGD", lcnn, lcnn, hip, x, y, hyper, noise, cuda)
    parity_record["configs1_pgd40_lcnn_lfcc_gpu_vs_cpu_oracle"] = E.slim(fig)
    tf, fr, fin, tg = fig["teacher_forced"], fig["free_running"], fig["final"], fig["target"]
    # teacher-forced, iterations 0, 4, ..., 36, 39 of the oracle's trajectory.  Measured: 0 - 140 of 516 800 signs differ per
    # iteration (agreement >= 0.9997), every flip at |grad| <= 2.9 % of its utterance's largest entry; gradient relative L2
    # 1e-6 where no max-feature-map / pooling winner re-routes (iteration 32) and 1e-3 .. 7e-3 where some do (the fused LFCC
    # differs from the CPU chain by ~1e-6 of scale, enough to turn a near-tie)
    # Round 4 (tests/parity_attribution.py, profiles/r04_parity_attribution.txt): every one of these flips sits behind a
    # re-routed max-feature-map / pool winner, and the shipped kernels re-route no more winners against this oracle than the
    # oracle's float32 does against its own float64 run; bounds = 3x the measured figures (0.9998, 1.7 %, 5.2e-3, 8.6e-8, 6e-8)
    assert tf["grad_sign_agreement_worst"] >= 0.9994, tf
    assert tf["flip_rel_worst"] <= 0.052, tf                      # flips only where |grad| is small for its utterance
    assert tf["grad_rel_l2_worst"] <= 1.6e-2, tf
    assert tf["update_max_abs_on_agreeing_worst"] == 0.0, tf      # the step kernel is exact given the gradient's sign
    assert tf["loss_rel_worst"] <= 3e-7 and tf["logit_max_abs_worst"] <= 2e-7, tf
    # free-running.  The sign pattern itself is chaotic (measured: 75 % of the final samples identical; the oracle started one
    # ulp away from itself ends 3 % different after the same 40 iterations and is still diverging), so it is recorded and only
    # loosely bounded; the loss per iteration (measured <= 3.3e-4 relative), the logits (<= 7.9e-4) and what the target makes
    # of the result (scores <= 6.5e-5, same labels, same EER) are the stated tolerance
    assert fin["box_ok"] and fin["linf_product"] <= hyper["eps"] + 1e-7, fin
    assert fr["loss_rel_worst"] <= 1.6e-3 and fr["logit_max_abs_worst"] <= 2.6e-3, fr          # measured 5.1e-4, 8.4e-4
    assert fin["identical_samples"] >= 0.6 and fin["max_abs"] <= 2 * hyper["eps"] * (1 + 1e-4), fin
    assert tg["labels_equal"] and tg["accuracy_product"] == tg["accuracy_oracle"], tg
    assert tg["score_max_abs"] <= 2e-4 and tg["eer_abs_diff"] <= 1e-9, tg                        # measured 6.5e-5


def test_pgdl2_40_on_specrnet_mel_matches_cpu_oracle(cuda, hip, parity_record):
    """configs[2] at B = 8."""
    model = _model("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, cuda)
    x, y = _batch(8, 4321)
    hyper = dict(eps=0.1, alpha=0.2, steps=40)
    g = torch.Generator().manual_seed(78)
    draws = (torch.randn(x.shape, generator=g), torch.rand(x.shape[0], generator=g))
    fig, got01, want01 = E.run_gradient_attack("PGDL2", model, model, hip, x, y, hyper, draws, cuda)
    parity_record["configs2_pgdl2_40_specrnet_mel_gpu_vs_cpu_oracle"] = E.slim(fig)
    tf, fr, fin, tg = fig["teacher_forced"], fig["free_running"], fig["final"], fig["target"]
    # teacher-forced.  The update kernel given the ORACLE's gradient: the row-norm tolerance of DESIGN.md section 5 (3e-7).
    # The product's own gradient at the oracle's iterate: relative L2 <= 4.9e-3 measured (MaxPool / LeakyReLU / |.|, angle
    # routing at near ties, as for LCNN), which moves the update by alpha * that (8.6e-5 measured)
    # bounds = 3x measured (6e-8; 4.9e-3, 0.9997, 2.2 %; 8.6e-5; 1.7e-7, 8.4e-7)
    assert tf["update_given_oracle_grad_max_abs_worst"] <= 2e-7, tf
    assert tf["grad_rel_l2_worst"] <= 1.5e-2 and tf["grad_sign_agreement_worst"] >= 0.9991 and tf["flip_rel_worst"] <= 0.066, tf
    assert tf["update_max_abs_on_agreeing_worst"] <= 2.6e-4, tf
    assert tf["loss_rel_worst"] <= 5.2e-7 and tf["logit_max_abs_worst"] <= 2.6e-6, tf
    # free-running: the iterates themselves end 0.48 of the perturbation's norm apart in the worst utterance — exactly the
    # oracle's distance from ITSELF started one ulp away (`divergence_oracle_vs_oracle_one_ulp`, same 0.48); bound: what
    # the attack optimises and what the target sees
    assert fin["box_ok"] and fin["l2_product_max"] <= hyper["eps"] * (1 + 1e-4), fin
    assert fin["l2_rel_diff_worst"] <= 7e-6, fin                                                  # measured 2.2e-6
    assert fr["loss_rel_worst"] <= 2.6e-4 and fr["logit_max_abs_worst"] <= 1.9e-3, fr          # measured 8.5e-5, 6.2e-4
    assert tg["labels_equal"] and tg["accuracy_product"] == tg["accuracy_oracle"], tg
    assert tg["score_max_abs"] <= 1e-5 and tg["eer_abs_diff"] <= 1e-9, tg                        # measured 3.1e-6


def test_cw_transfer_rawnet3_to_lcnn_matches_cpu_oracle(cuda, hip, lcnn, parity_record):
    """configs[3] at B = 2: CW as AttackEnum.CW configures it (c = 1, 100 steps, lr = 0.01) on RawNet3, scored by LCNN.
    cw.py:107-110 compares the batch cost every 10 iterations and stops when it rose: both sides must stop at the same
    iteration."""
    raw = _model("rawnet3", {}, cuda)
    x, y = _batch(2, 31)
    hyper = dict(c=1.0, kappa=0, steps=100, lr=0.01)
    fig, got01, want01 = E.run_cw(raw, lcnn, hip, x, y, hyper, cuda)
    parity_record["configs3_cw_rawnet3_to_lcnn_gpu_vs_cpu_oracle"] = E.slim(fig)
    fr, tf = fig["free_running"], fig["teacher_forced"]
    # both sides (and the oracle started one ulp away) stop at iteration 11 of 100
    assert fig["iterations_product"] == fig["iterations_oracle"] >= 10, fig
    # teacher-forced model gradient at the oracle's iterates 0, 5, 10.  RawNet3 differentiates log(|y| + 1e-6) of its sinc
    # encoder's output: where y is at rounding level the factor 1 / (|y| + 1e-6) is ~1e6 and follows y's last bits, those
    # few entries carry the gradient's norm (relative L2 0.3 - 1.6 measured — and the ORACLE's own gradient one ulp away from
    # the same iterate is recorded next to it); the robust statistics are the stated tolerance: median relative error
    # (measured 0.6 %), sign agreement (98.9 %), logits (3.7e-6)
    assert tf["logit_max_abs_worst"] <= 2e-5 and tf["grad_rel_err_median_worst"] <= 0.03, tf
    assert tf["grad_sign_agreement_worst"] >= 0.97, tf
    # free-running: Adam turns every coordinate's gradient SIGN into a +-lr move, rounding-level entries included (in the
    # reference too), so the iterates drift apart by up to lr per iteration (measured: max 0.043, mean 3.5e-3 after 11).  The
    # oracle started one ulp away drifts from itself by the same amounts (cost 4.6 %, distances 7.5 %, logits 1.2e-3); the
    # product's figures (3.8 %, 6.8 %, 1.8e-3) are bounded at ~3x
    assert fr["cost_rel_worst"] <= 0.12 and fr["l2_rel_worst"] <= 0.2 and fr["logit_max_abs_worst"] <= 6e-3, fr
    assert fr["adv_max_abs_worst"] <= 11 * hyper["lr"] and fr["adv_mean_abs_worst"] <= 1e-2, fr
    assert fig["final"]["box_ok"] and fig["final"]["rows_changed_product"] == fig["final"]["rows_changed_oracle"], fig["final"]
    assert fig["target"]["labels_equal"] and fig["target"]["score_max_abs"] <= 1e-6, fig["target"]


def test_pgd40_at_full_batch_with_sampled_oracle_checks(cuda, hip, lcnn, parity_record):
    """The headline workload itself (configs[1]: B = 128, PGD-40, Philox start) through attack_batch's order of operations,
    with launch 0, 10, 20, 30 of every kernel re-computed by the C oracle on the same inputs (bit-exact rules of
    oracle/checked_ops.py), plus the invariants of the result."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
    from oracle.checked_ops import CheckedOps
    x, y = synthetic_waveforms(128, seed=1234)
    x, y = x.to(cuda), y.to(cuda)
    ops = CheckedOps(hip, every=10)
    atk = E.armed(torchattacks.PGD(lcnn, eps=0.003, steps=40), ops)
    torch.manual_seed(42)
    x01, mn, mx = ops.to_minmax(x)
    adv01 = atk(x01, y)
    adv = ops.revert_minmax(adv01, mn, mx)
    preds, labels = score_batch(lcnn.eval(), adv)
    assert ops.calls["pgd_linf_step"] == 40 and ops.checked["pgd_linf_step"] == 4
    assert ops.checked["ce2_loss_grad"] == 4 and ops.checked["pgd_linf_init"] == 1
    assert (adv01 - x01).abs().max().item() <= 0.003 + 1e-7 and adv01.min() >= 0 and adv01.max() <= 1
    assert ((adv01 - x01).abs() > 0).float().mean().item() > 0.99
    assert torch.isfinite(preds).all()
    parity_record["configs1_full_batch_sampled_checks"] = {"batch": 128, "steps": 40, "checked": dict(ops.checked),
                                                           "launches": dict(ops.calls)}


def test_pgdl2_40_at_full_batch_on_specrnet_with_sampled_oracle_checks(cuda, hip, parity_record):
    """configs[2] at full size: SpecRNet + mel-spec, PGDL2-40 (eps 0.1, alpha 0.2, Philox start), B = 128 — exactly the
    single-pass L2 kernels' residency limit (B x C = 2 048 workgroups) — through attack_batch's order of operations, launch
    0, 10, 20, 30 of every kernel re-computed by the C oracle, no row handed to the repair kernel, and the ball / box
    invariants of the result (pgdl2.py:64-88)."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
    from oracle.checked_ops import CheckedOps
    model = _model("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, cuda)
    x, y = synthetic_waveforms(128, seed=4321)
    x, y = x.to(cuda), y.to(cuda)
    ops = CheckedOps(hip, every=10)
    atk = E.armed(torchattacks.PGDL2(model, eps=0.1, alpha=0.2, steps=40), ops)
    torch.manual_seed(43)
    x01, mn, mx = ops.to_minmax(x)
    adv01 = atk(x01, y)
    repaired = hip.pgd_l2_repaired_rows(adv01)
    adv = ops.revert_minmax(adv01, mn, mx)
    preds, labels = score_batch(model.eval(), adv)
    assert ops.calls["pgd_l2_step"] == 40 and ops.checked["pgd_l2_step"] == 4
    assert ops.checked["ce2_loss_grad"] == 4 and ops.checked["pgd_l2_init"] == 1
    assert repaired == 0                                  # every row's norm exchange completed inside the launch
    norms = (adv01 - x01).norm(dim=1)
    assert norms.max().item() <= 0.1 * (1 + 1e-5) and adv01.min() >= 0 and adv01.max() <= 1
    assert norms.min().item() > 0.05                      # 40 steps of alpha = 2 eps: every utterance sits near the sphere
    assert torch.isfinite(preds).all()
    parity_record["configs2_full_batch_sampled_checks"] = {"batch": 128, "steps": 40, "checked": dict(ops.checked),
                                                           "launches": dict(ops.calls), "pgd_l2_repaired_rows": repaired,
                                                           "l2_norm_max": norms.max().item(), "l2_norm_min": norms.min().item()}


def test_cw_on_rawnet3_where_the_blend_takes_iterates_matches_cpu_oracle(cuda, hip, parity_record):
    """VERDICT r05, What's missing 4: in every configs[3] run CW stops after 11 iterations without fooling RawNet3 on a row
    it classified correctly, so the best-so-far blend (cw.py:94-103) only ever took iteration 0 (where adv == x) - on a
    real detector the masked blend was covered by the surrogate fixture alone.  Here the attack SUCCEEDS: c = 1e4 makes the
    model term outweigh the distances (the seeded RawNet3 emits z ~ +0.02 for every row, so the rows labelled 1 need z < 0),
    steps = 20 compares the cost every 2 iterations.  WHICH iterate the blend takes is as chaotic as the trajectory (Adam turns
    rounding-level gradient entries into +-lr moves): the CPU oracle on 8 threads flips rows 2 / 3 at iterations 9 / 9 and stops
    after 11, started one ulp away at 8 / 10, on the GPU box's 16 threads at 9 / 7; the product at 7 / 10, stopping after 13.
      * free-running, product vs oracle: the same rows are taken, the taken iterates fool the attacked model on both sides,
        within a few iterations of each other (squared distances grow by ~2 per iteration) - the oracle's own spread;
      * teacher-forced: the product's blend kernel driven along the ORACLE's trajectory (its iterates, distances and
        logits, from the trace) reproduces the oracle's best adversarials and best distances bit for bit."""
    import copy
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from oracle import attacks as OA
    from oracle.checked_ops import TracingOps
    raw = _model("rawnet3", {}, cuda)
    x, y = _batch(4, 31)
    hyper = dict(c=1e4, kappa=0, steps=20, lr=0.01)
    model_cpu = copy.deepcopy(raw).cpu()
    x01, _, _ = OA.to_minmax(x)
    trace = []
    with E.cpu_threads(16), OA.attack_mode(model_cpu):
        want01 = OA.cw(model_cpu, x01, y, trace=trace, **hyper)

    # the oracle's blend history from its trace: cw.py:94-103 restated on (l2, z) per iteration
    best_l2 = 1e10 * torch.ones(len(x01))
    best = x01.clone()
    took = []
    for _, l2, z, adv in trace:
        correct = ((z > 0).long() == y).float()          # argmax of (-z, z) is 1 iff z > 0 (ties -> index 0, as torch.max)
        mask = (1 - correct) * (best_l2 > l2).float()
        best_l2 = mask * l2 + (1 - mask) * best_l2
        best = mask.view(-1, 1) * adv + (1 - mask.view(-1, 1)) * best
        took.append(mask.clone())
    assert torch.equal(best, want01)                     # the restatement IS the oracle's blend
    changed_oracle = ((want01 - x01).abs().amax(1) > 1e-6)
    assert changed_oracle.tolist() == [False, False, True, True], "the oracle run of this test no longer flips rows 2 and 3"

    # product, free-running
    tops = TracingOps(hip)
    atk = E.armed(torchattacks.CW(raw, **hyper), tops)
    xg, yg = x.to(cuda), y.to(cuda)
    g01, _, _ = hip.to_minmax(xg)
    got01 = atk(g01, yg)
    got = got01.cpu()
    changed_product = ((got - x01).abs().amax(1) > 1e-6)
    first_oracle = [int(torch.stack(took)[:, r].argmax()) if torch.stack(took)[:, r].any() else -1 for r in range(4)]
    pm = torch.stack([m.cpu() for m in tops.cw_mask])
    last_product = [int(pm[:, r].nonzero().max()) if pm[:, r].any() else -1 for r in range(4)]
    last_oracle = [int(torch.stack(took)[:, r].nonzero().max()) if torch.stack(took)[:, r].any() else -1 for r in range(4)]
    l2_product = ((got - x01) ** 2).sum(1)
    l2_oracle = ((want01 - x01) ** 2).sum(1)
    with torch.no_grad(), OA.attack_mode(model_cpu):
        z_oracle_best = model_cpu(want01).view(-1)
    raw.train()
    for mod in raw.modules():
        if "BatchNorm" in mod.__class__.__name__ or "Dropout" in mod.__class__.__name__:
            mod.eval()
    with torch.no_grad():
        z_product_best = raw(got01).view(-1).cpu()
    raw.eval()
    fig = {"hyper": hyper, "iterations_product": len(tops.cw_l2), "iterations_oracle": len(trace),
           "rows_changed_product": changed_product.tolist(), "rows_changed_oracle": changed_oracle.tolist(),
           "first_taken_iteration_oracle": first_oracle, "last_taken_iteration_oracle": last_oracle,
           "last_taken_iteration_product": last_product,
           "best_l2_product": l2_product.tolist(), "best_l2_oracle": l2_oracle.tolist(),
           "logit_of_best_product": z_product_best.tolist(), "logit_of_best_oracle": z_oracle_best.tolist(),
           "blended_rows_max_abs": (got - want01).abs().amax(1).tolist(),
           "oracle_spread_note": "the CPU oracle itself: 8 threads takes rows 2 / 3 at iterations 9 / 9 (squared distances 17.4 / "
                                 "17.8), started one ulp away at 8 / 10 (15.6 / 19.6), 16 threads on the GPU box's host at 9 / 7"}
    parity_record["configs3_cw_rows_that_flip_rawnet3_gpu_vs_cpu_oracle"] = fig
    assert changed_product.tolist() == changed_oracle.tolist(), fig
    assert abs(fig["iterations_product"] - fig["iterations_oracle"]) <= 4, fig       # two cost checks (every 2 iterations) apart at most
    for r in (2, 3):
        # the taken iterate fools the attacked model (label 1 needs z > 0 to be classified correctly) on both sides
        assert z_product_best[r] <= 0 and z_oracle_best[r] <= 0, fig
        # taken within a few iterations of the oracle's (measured 2 and 3; the oracle against itself: 1 and 2), at a squared
        # distance that follows the iteration count (measured 22 % and 42 % apart)
        assert abs(last_product[r] - last_oracle[r]) <= 5, fig
        assert abs(l2_product[r] - l2_oracle[r]).item() <= 0.6 * l2_oracle[r].item(), fig
        assert 5.0 <= l2_product[r].item() <= 30.0, fig
    assert got.min() >= 0 and got.max() <= 1
    # rows the model misclassifies from the start are taken at iteration 0 only (distance ~ 0): unchanged on both sides
    assert pm[0, :2].tolist() == [1.0, 1.0] and pm[1:, :2].sum().item() == 0, fig

    # teacher-forced: the product's blend kernel along the oracle's trajectory
    best_d = x01.clone().to(cuda)
    best_l2_d = 1e10 * torch.ones(len(x01), device=cuda)
    for (_, l2, z, adv), mask_o in zip(trace, took):
        l2d, zd = l2.to(cuda), z.to(cuda)
        correct = ((zd > 0).long() == yg).float()
        mask = (1 - correct) * (best_l2_d > l2d).float()
        assert torch.equal(mask.cpu(), mask_o)
        best_l2_d = mask * l2d + (1 - mask) * best_l2_d
        hip.cw_best_update(adv.to(cuda).contiguous(), mask.contiguous(), best_d)
    assert torch.equal(best_d.cpu(), want01) and torch.equal(best_l2_d.cpu(), best_l2)


def test_fgsm_and_cw_at_full_batch_rawnet3_to_lcnn_with_sampled_oracle_checks(cuda, hip, lcnn, parity_record):
    """configs[3] at full size: attack model RawNet3, target LCNN + LFCC, B = 64; FGSM (AttackEnum.FGSM: eps 0.0005) and CW
    (AttackEnum.CW: c 1, 100 steps, lr 0.01, early stop on) with every 10th launch of every kernel re-computed by the C
    oracle, and the invariants of both results (fgsm.py:59-60, cw.py:70-110)."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
    from oracle.checked_ops import CheckedOps
    raw = _model("rawnet3", {}, cuda)
    x, y = synthetic_waveforms(64, seed=31)
    x, y = x.to(cuda), y.to(cuda)
    record = {"batch": 64}
    for member in ("FGSM", "CW"):
        cls, params = AttackEnum[member].value
        ops = CheckedOps(hip, every=10)
        atk = E.armed(cls(raw, **params), ops)
        x01, mn, mx = ops.to_minmax(x)
        adv01 = atk(x01, y)
        adv = ops.revert_minmax(adv01, mn, mx)
        preds, labels = score_batch(lcnn.eval(), adv)
        assert adv01.min() >= 0 and adv01.max() <= 1 and torch.isfinite(adv).all() and torch.isfinite(preds).all()
        if member == "FGSM":
            assert ops.calls["fgsm_step"] + 0 == 1 and ops.checked["fgsm_step"] == 1
            assert (adv01 - x01).abs().max().item() <= params["eps"] + 1e-7
            assert ((adv01 - x01).abs() > 0).float().mean().item() > 0.99
        else:
            iters = ops.calls["cw_adam_step"]
            # cw.py:107-110: the cost is compared every steps // 10 iterations, so the attack stops at 10 k + 1 or runs all 100
            assert iters == params["steps"] or (iters % (params["steps"] // 10) == 1 and iters >= 11), iters
            assert ops.checked["cw_adam_step"] == (iters + 9) // 10 and ops.checked["cw_tanh_sqdist"] >= 1
            assert ops.calls["cw_best_update"] == iters
            changed = (adv01 - x01).abs().amax(dim=1) > 1e-6
            # the best-so-far blend only takes an iterate that fools the attacked model: changed rows moved by < 11 lr
            assert (adv01 - x01).abs().max().item() <= iters * params["lr"]
            record["cw_iterations"] = iters
            record["cw_rows_changed"] = int(changed.sum())
        record[member] = {"checked": dict(ops.checked), "launches": dict(ops.calls)}
    parity_record["configs3_full_batch_sampled_checks"] = record
