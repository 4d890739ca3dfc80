"""CPU: the end-to-end comparison harness of tests/e2e_parity.py run with the product's Attack classes driven by the
ORACLE's op table on the real LCNN + LFCC detector — both sides are then the same torch CPU arithmetic, so every figure
the GPU tests bound by a tolerance must be exact here (a harness that manufactures differences would show up)."""
import torch

from oracle import torch_ops
from tests import e2e_parity as E


def _lcnn():
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    return get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, "cpu").eval()


def test_pgd_and_pgdl2_harness_is_exact_on_cpu():
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    model = _lcnn()
    x, y = synthetic_waveforms(2, seed=1234)
    y[0], y[1] = 0, 1
    g = torch.Generator().manual_seed(5)
    noise = torch.empty_like(x).uniform_(-0.003, 0.003, generator=g)
    fig, got, want = E.run_gradient_attack("PGD", model, model, torch_ops, x, y, dict(eps=0.003, alpha=2 / 255, steps=3),
                                           noise, "cpu", forced_every=2, threads=8)
    assert torch.equal(got, want)
    assert fig["free_running"]["loss_rel_worst"] <= 1e-6 and fig["free_running"]["identical_samples_worst"] == 1.0
    assert fig["free_running"]["grad_sign_agreement_worst"] == 1.0
    assert fig["teacher_forced"]["grad_sign_agreement_worst"] == 1.0
    assert fig["teacher_forced"]["update_max_abs_on_agreeing_worst"] == 0.0
    assert fig["target"]["labels_equal"] and fig["target"]["score_max_abs"] == 0.0 and fig["target"]["eer_abs_diff"] == 0.0
    assert fig["final"]["linf_product"] <= 0.003 + 1e-7 and fig["final"]["box_ok"]

    draws = (torch.randn(x.shape, generator=g), torch.rand(2, generator=g))
    fig, got, want = E.run_gradient_attack("PGDL2", model, model, torch_ops, x, y, dict(eps=0.1, alpha=0.2, steps=3),
                                           draws, "cpu", forced_every=2, threads=8)
    # the C oracle's row norms sum in another order than torch.norm: <= 3e-7 per update.  The free-running
    # iterates are NOT bounded that way: a 6e-8 difference re-routes a max-feature-map / pooling winner in LCNN within a few
    # iterations (the oracle does the same to itself from a start moved by one ulp: `divergence_oracle_vs_oracle_one_ulp`)
    assert fig["teacher_forced"]["update_max_abs_on_agreeing_worst"] <= 3e-7
    assert fig["teacher_forced"]["grad_rel_l2_worst"] <= 1e-5 and fig["teacher_forced"]["logit_max_abs_worst"] <= 1e-7
    assert "divergence_oracle_vs_oracle_one_ulp" in fig["final"]
    assert fig["free_running"]["loss_rel_worst"] <= 1e-5 and fig["final"]["l2_product_max"] <= 0.1 * (1 + 1e-4)
    assert fig["target"]["labels_equal"]
    assert isinstance(E.slim(fig)["free_running"]["per_iteration"]["loss_rel"], list)


def test_sampled_checked_ops_counts():
    """CheckedOps(every=N): launch 0, N, 2N ... of each entry point are compared, the rest run unchecked."""
    from oracle.checked_ops import CheckedOps
    ops = CheckedOps(torch_ops, every=3)
    x = torch.rand(2, 64)
    g = torch.randn(2, 64)
    for _ in range(7):
        ops.pgd_linf_step(x, g, x, 0.01, 0.003)
    assert ops.calls["pgd_linf_step"] == 7 and ops.checked["pgd_linf_step"] == 3
    full = CheckedOps(torch_ops)
    full.fgsm_step(x, g, 0.001)
    assert full.calls["fgsm_step"] == 1 and full.checked["fgsm_step"] == 1
