"""`-m gpu`: SURVEY.md section 8(d)'s secondary shape, T = 64 000 (exactly 4.000 s: 401 frames).  With 401 frames the
LCNN planes are 401 x 80 -> 200 x 40 -> 100 x 20 -> 50 x 10 -> 25 x 5 — an ODD first plane (the first block's last
input row has no pooling partner; its input gradient takes the cell-centric kernel's odd-height tile seam), 8 000 / 2 000 /
500-pixel 1x1 blocks (whole 32-pixel tiles, no ragged tile as at 8 080 = 252.5 tiles), 13 + 1 frames in the STFT kernels'
last four-frame group — none of which T = 64 600 (404 frames) exercises.  VERDICT r05, What's weak 10."""
import pytest
import torch

pytestmark = pytest.mark.gpu

T = 64_000
SWITCHES = ("ADVSTEP_LCNN_FUSED", "ADVSTEP_LCNN_CONV0", "ADVSTEP_LCNN_CONV1X1", "ADVSTEP_LCNN_CONV3X3", "ADVSTEP_LCNN_LSTM",
            "ADVSTEP_LCNN_BN", "ADVSTEP_FUSED_LFCC", "ADVSTEP_FUSED_STFT")


def attack_mode(model):
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def test_lcnn_lfcc_at_401_frames_fused_kernels_vs_plain_pytorch_rocm(cuda, monkeypatch, parity_record):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = attack_mode(get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda))
    x = (torch.randn(4, T, generator=torch.Generator().manual_seed(3)) * 0.05).clamp(-1, 1).to(cuda)

    def run(fused):
        for s in SWITCHES:
            monkeypatch.setenv(s, "1" if fused else "0")
        a = x.clone().requires_grad_(True)
        spec = model._compute_frontend(a)
        z = model._compute_embedding(spec)
        (g,) = torch.autograd.grad(z.sum(), a)
        return spec.detach(), z.detach(), g

    s0, z0, g0 = run(False)
    s1, z1, g1 = run(True)
    assert s0.shape == s1.shape == (4, 1, 80, 401)                       # lcnn.py:252 pins (B, 1, 80, frames)
    scale = s0.abs().max().item()
    fig = {"frames": 401, "lfcc_max_abs_over_scale": (s0 - s1).abs().max().item() / scale,
           "logit_max_abs": (z0 - z1).abs().max().item(),
           "grad_rel_l2": ((g0 - g1).norm() / g0.norm()).item(),
           "grad_entries_off_by_1e-3_of_max": int(((g0 - g1).abs() > 1e-3 * g0.abs().max()).sum())}
    parity_record["lcnn_lfcc_T64000_fused_vs_plain"] = fig
    # measured (profiles/r06_parity.json): frontend 3.5e-7 of its scale, logits 2.6e-8, waveform gradient 7.8e-7 relative L2, no
    # entry off by 1e-3 of the largest (no near-tie winner re-routed on this input) - the T = 64 600 figures; bounds = 10 x
    assert fig["lfcc_max_abs_over_scale"] <= 5e-6, fig
    assert fig["logit_max_abs"] <= 5e-7, fig
    assert fig["grad_rel_l2"] <= 1e-5 and fig["grad_entries_off_by_1e-3_of_max"] == 0, fig
    assert torch.isfinite(g1).all() and g1.shape == (4, T)


def test_pgd4_at_401_frames_every_launch_checked_by_the_oracle(cuda):
    """The smoke run's recipe (__graft_entry__.smoke) at T = 64 000: min-max -> PGD-4 on LCNN + LFCC -> revert, every launch of
    the attack kernels re-computed by the CPU oracle (oracle/checked_ops.py raises on the first mismatch)."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops, torchattacks
    from audio_deepfake_adversarial_attacks_amd.evaluation import score_batch
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from oracle.checked_ops import CheckedOps
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(3, T, generator=g) * 0.05).clamp(-1, 1).to(cuda)
    y = torch.randint(0, 2, (3,), generator=g).to(cuda)
    ops = CheckedOps(hip_ops)
    atk = torchattacks.PGD(model, eps=0.003, steps=4)
    atk.set_training_mode(model_training=True, batchnorm_training=False)
    atk.ops = ops
    x01, mn, mx = ops.to_minmax(x)
    adv01 = atk(x01, y)
    adv = ops.revert_minmax(adv01, mn, mx)
    preds, _ = score_batch(model.eval(), adv)
    torch.cuda.synchronize()
    assert adv.shape == (3, T) and (adv01 - x01).abs().max().item() <= 0.003 + 1e-7
    assert 0.0 <= adv01.min().item() and adv01.max().item() <= 1.0 and torch.isfinite(preds).all()
    assert ops.calls["pgd_linf_step"] == 4 and ops.calls["pgd_linf_init"] == 1 and ops.calls["ce2_loss_grad"] == 4
