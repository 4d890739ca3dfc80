"""CPU: the detector restatements against logits / input-gradients produced by the reference's own BaseLCNN,
BaseSpecRNet and RawNet3-after-its-first-layer (tests/golden/*_body.npz), plus structure checks for RawNet3's sinc
encoder (third-party, parity-unpinned)."""
import numpy as np
import pytest
import torch

from audio_deepfake_adversarial_attacks_amd.models import lcnn, models, rawnet3, sincfb, specrnet

T = torch.from_numpy


@pytest.fixture(autouse=True)
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def sd_of(fixture):
    return {k[3:]: T(v) for k, v in fixture.items() if k.startswith("sd_")}


def attack_mode(model):
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    return model


def test_lcnn_body_equals_reference(golden):
    g = golden("lcnn_body")
    body = lcnn.BaseLCNN(input_channels=1, num_coefficients=80).eval()
    body.load_state_dict(sd_of(g), strict=True)           # the reference's key names
    spec = T(g["spec"])
    with torch.no_grad():
        assert torch.equal(body(spec), T(g["logits"]))
    s = spec.clone().requires_grad_(True)
    out = attack_mode(body)(s)
    assert torch.equal(out, T(g["logits_attackmode"]))
    (grad,) = torch.autograd.grad(out.sum(), s)
    assert torch.equal(grad, T(g["grad_spec"]))


def test_lcnn_state_dict_layout():
    m = models.get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, "cpu")
    keys = set(m.state_dict())
    for i in (0, 3, 6, 10, 13, 16, 19, 22, 25):                       # SURVEY.md section 5 (checkpoint key names)
        assert f"m_transform.{i}.weight" in keys and f"m_transform.{i}.bias" in keys
    for i in (5, 9, 12, 18, 21, 24):
        assert f"m_transform.{i}.running_mean" in keys and f"m_transform.{i}.weight" not in keys  # affine=False
    assert {"m_before_pooling.0.l_blstm.weight_ih_l0", "m_before_pooling.1.l_blstm.weight_hh_l0_reverse",
            "m_output_act.weight", "frontend.filter_mat", "frontend.dct_mat", "frontend.Spectrogram.window"} <= keys
    assert sum(p.numel() for p in m.parameters()) == 467_425
    out = m(torch.rand(2, 64_600))
    assert out.shape == (2, 1)


def test_specrnet_body_equals_reference(golden):
    g = golden("specrnet_body")
    body = specrnet.BaseSpecRNet(specrnet.get_config(2), device="cpu").eval()
    body.load_state_dict(sd_of(g), strict=True)
    with torch.no_grad():
        assert torch.equal(body(T(g["spec"])), T(g["logits"]))
    assert sum(p.numel() for p in body.parameters()) == 278_165       # SURVEY.md section 2 (2 input channels)
    s = T(g["spec"]).clone().requires_grad_(True)
    out = attack_mode(body)(s)
    assert torch.equal(out, T(g["logits_attackmode"]))
    (grad,) = torch.autograd.grad(out.sum(), s)
    assert torch.equal(grad, T(g["grad_spec"]))


def test_specrnet_unused_bn1_still_tracks_statistics_in_train_mode():
    """specrnet.py:75-78: the reference evaluates bn1(x) and drops the result; in train mode the running statistics
    (which are saved in checkpoints) still move."""
    block = specrnet.Residual_block2D([4, 4], first=False).train()
    before = block.bn1.running_mean.clone()
    x = torch.randn(2, 4, 8, 8) + 3.0
    block(x)
    assert not torch.equal(block.bn1.running_mean, before) and int(block.bn1.num_batches_tracked) == 1
    block.eval()
    frozen = block.bn1.running_mean.clone()
    block(x)
    assert torch.equal(block.bn1.running_mean, frozen)


def test_specrnet_full_model_and_registry():
    m = models.get_model("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, "cpu")
    assert m(torch.rand(2, 64_600)).shape == (2, 1)
    assert not any(k.startswith("frontend") for k in m.state_dict())
    with pytest.raises(ValueError, match="not supported"):
        models.get_model("frontend_specrnet", {}, "cpu")              # the reference's own broken config name


def test_mfm_validates_its_input():
    mfm = lcnn.MaxFeatureMap2D()
    x = torch.rand(2, 6, 3, 3)
    assert torch.equal(mfm(x), torch.maximum(x[:, :3], x[:, 3:]))
    with pytest.raises(ValueError):
        mfm(torch.rand(2, 5, 3, 3))
    with pytest.raises(ValueError):
        lcnn.MaxFeatureMap2D(max_dim=4)(x)
    with pytest.raises(ValueError):
        lcnn.BLSTMLayer(8, 7)


def test_sinc_filterbank_properties():
    fb = sincfb.ParamSincFB(256, 251, stride=10)
    filt = fb.filters()
    assert filt.shape == (256, 1, 251) and torch.isfinite(filt).all()
    cos, sin = filt[:128, 0], filt[128:, 0]
    assert torch.allclose(cos, cos.flip(1), atol=1e-6)                # even filters
    assert torch.allclose(sin, -sin.flip(1), atol=1e-6)               # odd filters
    assert torch.allclose(cos[:, 125], torch.ones(128), atol=1e-6)    # centre tap 2*band / (2*band)
    assert sorted(fb.state_dict()) == ["band_hz_", "low_hz_", "n_", "window_"]
    low = 50 + fb.low_hz_.abs().squeeze()
    assert (low[1:] > low[:-1]).all() and low[0] >= 80 - 1e-3          # mel-spaced from 30 Hz + min_low_hz
    # pass-band check: filter k responds to a tone inside its band far more than to one outside
    t = torch.arange(251) / 16_000.0
    k = 40
    lo_hz = float(low[k])
    hi_hz = lo_hz + 50 + float(fb.band_hz_.abs()[k])
    inside = torch.cos(2 * torch.pi * (lo_hz + hi_hz) / 2 * t)
    outside = torch.cos(2 * torch.pi * (hi_hz + 1500) * t)
    assert (cos[k] * inside).sum().abs() > 10 * (cos[k] * outside).sum().abs()
    enc = sincfb.Encoder(fb)
    assert enc(torch.rand(2, 64_600)).shape == (2, 256, 6435)


def test_sinc_filterbank_against_scipy_firwin():
    """Independent anchor for the restated asteroid filterbank: its even filters are scipy's Hamming-windowed ideal band-pass
    (`firwin(scale=False)`) rescaled by fs / (2 band), and even + i odd is an analytic band-pass (one-sided spectrum)."""
    from scipy.signal import firwin
    fb = sincfb.ParamSincFB(256, 251, stride=10)
    filt = fb.filters().detach().double().numpy()[:, 0]
    low = (50 + fb.low_hz_.abs()).detach().double().numpy()[:, 0]
    high = np.minimum(low + 50 + fb.band_hz_.abs().detach().double().numpy()[:, 0], 8000.0)
    for k in range(128):
        ref = firwin(251, [low[k], min(high[k], 8000.0 - 1e-3)], pass_zero=False, window="hamming", scale=False, fs=16000.0)
        ref = ref * 16000.0 / (2 * (high[k] - low[k]))
        assert np.abs(ref - filt[k]).max() < 5e-5, k                  # float32 evaluation of the closed form
    spectrum = np.fft.fft(filt[:128] + 1j * filt[128:], 4096, axis=1)
    positive = (np.abs(spectrum[:, 1:2048]) ** 2).sum(1)
    negative = (np.abs(spectrum[:, 2049:]) ** 2).sum(1)
    ratio = negative / positive
    assert ratio[:-1].max() < 1e-3 and ratio[-1] < 0.05               # the last band touches Nyquist and wraps


def test_rawnet3_structure_and_forward():
    m = rawnet3.prepare_model().eval()
    n_params = sum(p.numel() for p in m.parameters())
    assert 15_000_000 < n_params < 16_500_000                         # "~15.5 M params" 
    keys = set(m.state_dict())
    assert {"preprocess.0.flipped_filter", "preprocess.1.weight", "conv1.filterbank.low_hz_", "bn1.weight",
            "layer1.conv1.weight", "layer1.convs.6.weight", "layer1.afms.alpha", "layer1.residual.0.weight",
            "layer4.weight", "attention.0.weight", "attention.3.weight", "bn5.running_mean", "fc6.weight"} <= keys
    x = torch.rand(2, 16_000, requires_grad=True)
    out = m(x)
    assert out.shape == (2, 1) and torch.isfinite(out).all()
    (g,) = torch.autograd.grad(out.sum(), x)
    assert torch.isfinite(g).all() and g.abs().max() > 0


def rawnet3_like_fixture(g):
    """The in-tree RawNet3 carrying the fixture's weights (seeded recipe, verified tensor by tensor against the SHA-256
    digests taken from the reference class) with the recorded tensor standing in for the sinc encoder's output."""
    import json
    from tests.helpers import FixedEncoder, rawnet3_fixture_weights, tensor_digests
    model = rawnet3_fixture_weights(rawnet3.prepare_model)
    want = json.loads(str(g["digests"]))
    got = tensor_digests(model.state_dict())
    assert set(got) == set(want)                                       # the reference's key names, conv1.* aside
    assert [k for k in want if got[k] != want[k]] == []
    model.conv1 = FixedEncoder()
    return model.eval()


def test_rawnet3_body_equals_reference(golden):
    """src/models/rawnet3.py:81-137 (+ Bottle2neck / AFMS :161-274): logits and the attack-mode gradient w.r.t. the
    tensor leaving conv1, bit for bit."""
    g = golden("rawnet3_body")
    model = rawnet3_like_fixture(g)
    model.conv1.h = T(g["h"])
    with torch.no_grad():
        assert torch.equal(model(T(g["x"])), T(g["logits"]))
    h = T(g["h"]).clone().requires_grad_(True)
    model.conv1.h = h
    out = attack_mode(model)(T(g["x"]))
    assert torch.equal(out, T(g["logits_attackmode"]))
    (grad,) = torch.autograd.grad(out.sum(), h)
    assert torch.equal(grad, T(g["grad_h"]))


def test_preemphasis_matches_definition():
    x = torch.rand(2, 50)
    y = rawnet3.PreEmphasis()(x)
    want = x.clone()
    want[:, 1:] = x[:, 1:] - 0.97 * x[:, :-1]
    want[:, 0] = x[:, 0] - 0.97 * x[:, 1]       # reflect padding on the left
    assert torch.allclose(y.squeeze(1), want, atol=1e-6)


# ---- round 2: RawNet3's GEMM formulations are plain torch ops, so their arithmetic is pinned on the CPU as well -------------------

def test_rawnet3_dilated_gemm_equals_conv1d_and_its_input_gradient():
    """models/rawnet3.py:_dilated_gemm (k GEMMs accumulating in place into sub-ranges, `out=` channel slices, transposed taps)
    against nn.Conv1d with autograd, in float64."""
    import torch.nn as nn
    from audio_deepfake_adversarial_attacks_amd.models import rawnet3 as R
    torch.manual_seed(0)
    for C, k, d, T in [(8, 3, 2, 40), (8, 3, 4, 9), (4, 5, 3, 30), (4, 3, 4, 5)]:
        conv = nn.Conv1d(C, C, k, dilation=d, padding=(k // 2) * d).double()
        big = torch.randn(3, 3 * C, T, dtype=torch.double)
        x = big[:, C:2 * C].detach().clone().requires_grad_(True)
        y0 = conv(x)
        g = torch.randn_like(y0)
        (g0,) = torch.autograd.grad(y0, x, g)
        out = torch.zeros(3, 3 * C, T, dtype=torch.double)
        y1 = R._dilated_gemm(big[:, C:2 * C], conv.weight.detach(), conv.bias.detach(), d, out=out[:, C:2 * C])
        assert y1.data_ptr() == out[:, C:2 * C].data_ptr() and not out[:, :C].any() and not out[:, 2 * C:].any()
        assert (y0 - y1).abs().max().item() < 1e-12
        taps = [conv.weight.detach()[:, :, j].contiguous() for j in range(k)]
        g1 = R._dilated_gemm(g, taps, None, d, transpose=True)
        assert (g0 - g1).abs().max().item() < 1e-12
        x2 = x.detach().clone().requires_grad_(True)
        y2 = R._SameConv1dFrozen.apply(x2, conv.weight.detach(), conv.bias.detach(), d)
        (g2,) = torch.autograd.grad(y2, x2, g)
        assert (y0 - y2).abs().max().item() < 1e-12 and (g0 - g2).abs().max().item() < 1e-12


def test_sinc_encoder_gemm_formulation_equals_conv1d():
    import torch.nn.functional as F
    from audio_deepfake_adversarial_attacks_amd.models import sincfb
    torch.manual_seed(1)
    for B, T, nf, K, st in [(3, 2000, 8, 251, 10), (2, 251, 4, 251, 10), (2, 1003, 6, 31, 7)]:
        x = torch.randn(B, 1, T, dtype=torch.double, requires_grad=True)
        w = torch.randn(nf, 1, K, dtype=torch.double)
        y0 = F.conv1d(x, w, stride=st)
        g = torch.randn_like(y0)
        (g0,) = torch.autograd.grad(y0, x, g)
        x1 = x.detach().clone().requires_grad_(True)
        y1 = sincfb._StridedCorrelationFrozen.apply(x1, w, st)
        (g1,) = torch.autograd.grad(y1, x1, g)
        assert y1.shape == y0.shape and (y0 - y1).abs().max().item() < 1e-12 and (g0 - g1).abs().max().item() < 1e-12


def test_context_attention_split_equals_the_concatenated_convolution():
    """W [x; mean 1^T; std 1^T] = W_x x + (W_mean mean + W_std std + b) 1^T  (models/rawnet3.py, RawNet3.forward)."""
    import torch.nn as nn
    torch.manual_seed(2)
    B, C, T = 3, 16, 9
    x = torch.randn(B, C, T, dtype=torch.double)
    conv = nn.Conv1d(3 * C, 8, 1).double()
    mean = x.mean(2, keepdim=True).repeat(1, 1, T)
    std = torch.sqrt(x.var(2, keepdim=True).clamp(min=1e-4, max=1e4)).repeat(1, 1, T)
    ref = conv(torch.cat((x, mean, std), 1))
    W = conv.weight[:, :, 0]
    const = x.mean(2) @ W[:, C:2 * C].t() + torch.sqrt(x.var(2).clamp(min=1e-4, max=1e4)) @ W[:, 2 * C:].t() + conv.bias
    h = torch.baddbmm(const.unsqueeze(2), W[:, :C].unsqueeze(0).expand(B, -1, -1), x)
    assert (ref - h).abs().max().item() < 1e-12
