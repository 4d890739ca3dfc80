"""`-m gpu`: the detectors' SHIPPED device paths (fused HIP kernels with frozen parameters, i.e. what runs inside an
attack) against the logits and input gradients the REFERENCE's own model classes produced on CPU
(tests/golden/{lcnn,specrnet,rawnet3}_body.npz — reference src/models/lcnn.py:166-208, specrnet.py:141-181,
rawnet3.py:81-137).  Tolerances are float32 cross-device bounds, stated per test and kept within 10x of the figures
measured on MI355X (profiles/r02_parity.json, r03_parity.json), so a kernel that loses three decimal digits fails."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def sd_of(fixture):
    return {k[3:]: T(v) for k, v in fixture.items() if k.startswith("sd_")}


def attack_mode_frozen(model):
    """attack.py:311-319 + the parameter freeze Attack.__call__ applies: the fused kernels engage."""
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def gradient_figures(got, want):
    d = (got - want).abs()
    scale = want.abs().max().item()
    return {"grad_rel_l2": ((got - want).norm() / want.norm()).item(), "grad_max_abs_over_max": d.max().item() / scale,
            "grad_frac_off_by_1e-3_of_max": (d > 1e-3 * scale).float().mean().item(), "grad_max": scale}


def test_lcnn_fused_device_path_matches_reference_body(cuda, golden, parity_record):
    """BaseLCNN: first-block kernel, Winograd 3x3 blocks on the matrix cores, 1x1 blocks, folded BatchNorm, persistent
    LSTM — vs the reference's CPU logits / grad_spec.  Measured: logits 3e-8, gradient relative L2 6.4e-7, worst entry
    6.5e-7 of the largest, no max-feature-map / pool winner re-routed on this input.  Bounds: logits 3e-7 abs
    (|logit| ~ 0.1); gradient relative L2 6e-6, no entry off by more than 1e-5 of the largest."""
    from audio_deepfake_adversarial_attacks_amd.models import lcnn
    g = golden("lcnn_body")
    body = lcnn.BaseLCNN(input_channels=1, num_coefficients=80)
    body.load_state_dict(sd_of(g), strict=True)
    body = attack_mode_frozen(body.to(cuda))
    spec = T(g["spec"]).to(cuda).requires_grad_(True)
    out = body(spec)
    (grad,) = torch.autograd.grad(out.sum(), spec)
    want_z, want_g = T(g["logits_attackmode"]).to(cuda), T(g["grad_spec"]).to(cuda)
    fig = gradient_figures(grad, want_g)
    fig["logit_max_abs"] = (out - want_z).abs().max().item()
    with torch.no_grad():
        fig["logit_eval_max_abs"] = (body.eval()(spec.detach()) - T(g["logits"]).to(cuda)).abs().max().item()
    parity_record["lcnn_body_fused_vs_reference"] = fig
    assert fig["logit_max_abs"] <= 3e-7 and fig["logit_eval_max_abs"] <= 3e-7, fig
    assert fig["grad_rel_l2"] <= 6e-6 and fig["grad_max_abs_over_max"] <= 1e-5, fig


def test_specrnet_device_path_matches_reference_body(cuda, golden, parity_record):
    """BaseSpecRNet on the device — residual blocks on the Winograd matrix-core kernel (`advstep_resconv_*`) and the
    vector-ALU kernels for the 2-channel ends, fused elementwise / pooling passes, fused GRU — vs the reference's CPU logits
    / grad_spec.  Measured since the blocks left MIOpen (round 2): logits 1.5e-8, gradient relative L2 6.4e-7, worst entry
    6.7e-7 of the largest, no pooling winner re-routed on this input.  Bounds: logits 2e-7 abs; gradient relative L2 6e-6,
    no entry off by more than 1e-5 of the largest."""
    from audio_deepfake_adversarial_attacks_amd.models import specrnet
    g = golden("specrnet_body")
    body = specrnet.BaseSpecRNet(specrnet.get_config(2), device=str(cuda))
    body.load_state_dict(sd_of(g), strict=True)
    body = attack_mode_frozen(body.to(cuda))
    spec = T(g["spec"]).to(cuda).requires_grad_(True)
    out = body(spec)
    (grad,) = torch.autograd.grad(out.sum(), spec)
    fig = gradient_figures(grad, T(g["grad_spec"]).to(cuda))
    fig["logit_max_abs"] = (out - T(g["logits_attackmode"]).to(cuda)).abs().max().item()
    parity_record["specrnet_body_device_vs_reference"] = fig
    assert fig["logit_max_abs"] <= 2e-7, fig
    assert fig["grad_rel_l2"] <= 6e-6 and fig["grad_max_abs_over_max"] <= 1e-5, fig


def test_rawnet3_device_path_matches_reference_body(cuda, golden, parity_record):
    """RawNet3 after its first layer on the device — the dilated Res2Net convolutions as GEMMs over shifted views
    (models/rawnet3.py:_same_conv1d), library GEMMs for the 1x1 convolutions — vs the reference class's CPU logits and
    gradient w.r.t. the tensor leaving conv1.  Measured: logits 4.8e-7, gradient relative L2 3.5e-5, worst entry 4e-5 of the
    largest (2 300 GEMM-accumulated channels ahead of the statistics pooling; Tensile's fp32 GEMMs sum in another order than
    the CPU's).  Bounds: logits 5e-6 abs; gradient relative L2 3.5e-4, no entry off by more than 4e-4 of the largest."""
    from tests.test_models import rawnet3_like_fixture
    g = golden("rawnet3_body")
    model = attack_mode_frozen(rawnet3_like_fixture(g).to(cuda))
    h = T(g["h"]).to(cuda).requires_grad_(True)
    model.conv1.h = h
    out = model(T(g["x"]).to(cuda))
    (grad,) = torch.autograd.grad(out.sum(), h)
    fig = gradient_figures(grad, T(g["grad_h"]).to(cuda))
    fig["logit_max_abs"] = (out - T(g["logits_attackmode"]).to(cuda)).abs().max().item()
    parity_record["rawnet3_body_device_vs_reference"] = fig
    assert fig["logit_max_abs"] <= 5e-6, fig
    assert fig["grad_rel_l2"] <= 3.5e-4 and fig["grad_max_abs_over_max"] <= 4e-4, fig


def test_lcnn_fused_tail_steps_aside_for_hooks(cuda):
    """the one-node tail (lcnn_ops.lcnn_tail) bypasses m_before_pooling / m_output_act's forward(); a hook on any
    of them must keep firing, so the plain modules run then — with the same logits (reference lcnn.py:196-205)."""
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(3)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.randn(4, 64_600, device=cuda) * 0.05
    fused = model(x)
    seen = []
    handle = model.m_output_act.register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
    try:
        hooked = model(x)
    finally:
        handle.remove()
    assert len(seen) == 1 and torch.equal(seen[0], hooked)
    assert (hooked - fused).abs().max().item() <= 2e-6          # separate ops vs one node: summation order only
    seen.clear()
    again = model(x)
    assert not seen and torch.equal(again, fused)               # hook gone: the fused node is back, bit for bit
    # hooks registered for EVERY module fire from Module.__call__ as well - same rule
    calls = []
    handle = torch.nn.modules.module.register_module_forward_hook(
        lambda m, i, o: calls.append(m) if m is model.m_output_act else None)
    try:
        hooked = model(x)
    finally:
        handle.remove()
    assert len(calls) == 1 and (hooked - fused).abs().max().item() <= 2e-6
    assert torch.equal(model(x), fused)
    # the weight's w / T row is cached off the Parameter: nothing extra is pickled with the model
    assert not hasattr(model.m_output_act.weight, "_advstep_over_t")
