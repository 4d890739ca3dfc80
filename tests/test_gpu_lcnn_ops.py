"""`-m gpu`: the LCNN max-feature-map kernels (include/advstep_lcnn.h) against the ATen ops they replace, on the GPU,
bit for bit — values, input gradients, ties, NaN, odd sizes — and the fused LCNN against the plain one."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(cuda):
    from audio_deepfake_adversarial_attacks_amd import lcnn_ops
    return lcnn_ops


def ref_mfm(x, bias=None):
    if bias is not None:
        x = x + bias.view(1, -1, 1, 1)
    n, c2, h, w = x.shape
    return x.view(n, 2, c2 // 2, h, w).max(1)[0]


def ref_mfm_pool(x, bias=None):
    return torch.nn.functional.max_pool2d(ref_mfm(x, bias), (2, 2), (2, 2))


def make(shape, cuda, seed, ties=False, nans=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if ties:
        x = (x * 2).round() / 2          # many exact ties, between channel halves and inside pooling windows
    if nans:
        flat = x.view(-1)
        flat[torch.randint(0, flat.numel(), (max(flat.numel() // 50, 1),), generator=g)] = float("nan")
    return x.to(cuda)


def same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=1234.5), torch.nan_to_num(b, nan=1234.5)) and \
        torch.equal(torch.isnan(a), torch.isnan(b))


SHAPES = [(2, 4, 6, 8), (3, 2, 1, 4), (2, 6, 5, 7), (1, 2, 3, 3), (2, 8, 101, 20), (2, 64, 404, 80), (3, 10, 7, 12),
          (1, 2, 2, 2), (2, 4, 9, 16)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "ties", "nans"])
@pytest.mark.parametrize("with_bias", [False, True])
def test_mfm_matches_aten(L, cuda, shape, mode, with_bias):
    x = make(shape, cuda, 1, ties=mode == "ties", nans=mode == "nans").requires_grad_(True)
    bias = (torch.randn(shape[1], generator=torch.Generator().manual_seed(2)).to(cuda) if with_bias else None)
    if with_bias and mode == "ties":
        bias = (bias * 2).round() / 2
    y_ref = ref_mfm(x, bias)
    gy = make(tuple(y_ref.shape), cuda, 3)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    y = L.mfm(x, bias)
    (gx,) = torch.autograd.grad(y, x, gy)
    assert same(y, y_ref)
    assert same(gx, gx_ref)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "ties", "nans"])
@pytest.mark.parametrize("with_bias", [False, True])
def test_mfm_pool2_matches_aten(L, cuda, shape, mode, with_bias):
    if shape[2] < 2 or shape[3] < 2:
        pytest.skip("ATen rejects an empty pooled output")
    x = make(shape, cuda, 4, ties=mode == "ties", nans=mode == "nans").requires_grad_(True)
    bias = (torch.randn(shape[1], generator=torch.Generator().manual_seed(5)).to(cuda) if with_bias else None)
    if with_bias and mode == "ties":
        bias = (bias * 2).round() / 2
    y_ref = ref_mfm_pool(x, bias)
    gy = make(tuple(y_ref.shape), cuda, 6)
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    y = L.mfm_pool2(x, bias)
    (gx,) = torch.autograd.grad(y, x, gy)
    assert y.shape == y_ref.shape and same(y, y_ref)
    assert same(gx, gx_ref)


def test_bias_gradient_when_requested(L, cuda):
    x = make((2, 6, 8, 8), cuda, 7).requires_grad_(True)
    bias = torch.randn(6, device=cuda, requires_grad=True)
    for fn, ref in ((L.mfm, ref_mfm), (L.mfm_pool2, ref_mfm_pool)):
        gy = torch.randn_like(ref(x, bias))
        gx_ref, gb_ref = torch.autograd.grad(ref(x, bias), (x, bias), gy)
        gx, gb = torch.autograd.grad(fn(x, bias), (x, bias), gy)
        assert torch.equal(gx, gx_ref) and torch.allclose(gb, gb_ref, rtol=1e-5, atol=1e-6)


def test_input_validation(L, cuda):
    from audio_deepfake_adversarial_attacks_amd._lib import AdvstepError
    with pytest.raises(AdvstepError, match="no CPU fallback"):
        L.mfm(torch.rand(1, 2, 4, 4))
    with pytest.raises(ValueError):
        L.mfm(torch.rand(1, 3, 4, 4, device=cuda))
    with pytest.raises(ValueError):
        L.mfm(torch.rand(1, 4, 4, 4, device=cuda), torch.rand(3, device=cuda))
    assert L.mfm(torch.rand(0, 4, 4, 4, device=cuda)).shape == (0, 2, 4, 4)


def test_fused_lcnn_is_bit_identical_to_plain_lcnn(cuda, monkeypatch):
    """Same weights, same input: logits and input-gradient of the attack-mode model with and without the kernels.
    Compared at the spectrogram (the LFCC backward uses atomic index_add, so waveform gradients are not run-to-run
    reproducible even for the plain model; they are compared against that noise floor instead)."""
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda)
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    x = (torch.randn(4, 64_600, generator=torch.Generator().manual_seed(1)) * 0.05).to(cuda)
    spec = model._compute_frontend(x).detach()

    # the convolution-fusing kernels round differently from MIOpen; this test isolates the max-feature-map kernels
    monkeypatch.setenv("ADVSTEP_LCNN_CONV0", "0")
    monkeypatch.setenv("ADVSTEP_LCNN_CONV1X1", "0")
    monkeypatch.setenv("ADVSTEP_LCNN_CONV3X3", "0")
    monkeypatch.setenv("ADVSTEP_LCNN_LSTM", "0")
    monkeypatch.setenv("ADVSTEP_LCNN_BN", "0")
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "0")

    def run(fused, frozen, waveform=False):
        monkeypatch.setenv("ADVSTEP_LCNN_FUSED", "1" if fused else "0")
        for p in model.parameters():
            p.requires_grad_(not frozen)
        a = (x if waveform else spec).clone().requires_grad_(True)
        z = model(a) if waveform else model._compute_embedding(a)
        (g,) = torch.autograd.grad(z.sum(), a)
        return z.detach(), g

    z0, g0 = run(False, False)
    for fused, frozen in ((True, False), (True, True)):     # without and with the bias add folded in
        z, g = run(fused, frozen)
        assert torch.equal(z, z0), (fused, frozen)
        assert torch.equal(g, g0), (fused, frozen)
    zw0, gw0 = run(False, False, waveform=True)
    zw1, gw1 = run(False, False, waveform=True)
    zw2, gw2 = run(True, True, waveform=True)
    assert torch.equal(zw0, zw2)
    noise = (gw0 - gw1).abs().max().item()
    assert (gw0 - gw2).abs().max().item() <= max(4 * noise, 1e-6 * gw0.abs().max().item())
    for p in model.parameters():
        p.requires_grad_(True)


# ---- fused first block: Conv2d(1, 2C, 5x5, pad 2) -> MFM -> MaxPool2d(2, 2) ------------------------------------------------

def ref_block0(x, weight, bias):
    return ref_mfm_pool(torch.nn.functional.conv2d(x, weight, bias, stride=1, padding=2))


@pytest.mark.parametrize("shape", [(2, 1, 12, 16), (3, 1, 9, 7), (1, 1, 2, 2), (2, 1, 101, 20), (4, 1, 404, 80), (2, 1, 5, 33),
                                   (3, 1, 14, 70), (2, 1, 9, 64), (5, 1, 11, 79), (1, 1, 2, 80), (37, 1, 6, 66)])
@pytest.mark.parametrize("C,with_bias", [(32, True), (3, False), (8, True), (32, False)])
def test_conv5_mfm_pool2_matches_float64_reference(L, cuda, shape, C, with_bias):
    g = torch.Generator().manual_seed(shape[2] * 131 + shape[3] + C)
    x = torch.randn(shape, generator=g).to(cuda)
    weight = (torch.randn(2 * C, 1, 5, 5, generator=g) * 0.3).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda) if with_bias else None
    xr = x.double().requires_grad_(True)
    y_ref = ref_block0(xr, weight.double(), bias.double() if with_bias else None)
    gy = torch.randn(y_ref.shape, generator=g).to(cuda)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy.double())
    xa = x.clone().requires_grad_(True)
    y = L.conv5_mfm_pool2(xa, weight, bias)
    (gx,) = torch.autograd.grad(y, xa, gy)
    assert y.shape == y_ref.shape
    # the same winners must be selected unless two candidates are within float rounding of each other (not at these seeds)
    assert (y.double() - y_ref).abs().max().item() <= 2e-5
    assert (gx.double() - gx_ref).abs().max().item() <= 2e-4 * max(gx_ref.abs().max().item(), 1.0)
    # and against the float32 ATen composition it replaces
    xf = x.clone().requires_grad_(True)
    yf = ref_block0(xf, weight, bias)
    (gxf,) = torch.autograd.grad(yf, xf, gy)
    assert (y - yf).abs().max().item() <= 2e-5
    close = (gx - gxf).abs() <= 2e-4 * max(gxf.abs().max().item(), 1.0)
    assert close.float().mean().item() >= 0.999       # MIOpen's own rounding may flip a near-tie winner


@pytest.mark.parametrize("shape,C", [((4, 1, 404, 80), 32), ((3, 1, 9, 16), 8), ((2, 1, 101, 20), 3), ((5, 1, 38, 128), 32),
                                     ((2, 1, 7, 2), 32), ((1, 1, 64, 66), 96)])
def test_conv5_backward_cell_centric_equals_the_gather(L, cuda, monkeypatch, shape, C):
    """Round 5: the first block's input gradient with a thread per pooled CELL (window table in LDS, one exchange per tile) against
    the patch-centric gather of rounds 1-4 on the same (gradient, selection bytes, weights): the same sums in another order - equal
    to float rounding, both finite, the tile seams (rows 17 / 34 / ... of the pooled plane at width 40), the trailing odd row and
    the narrow / wide / many-channel shapes included.  ADVSTEP_CONV0_BWD=gather selects the old kernel."""
    g = torch.Generator().manual_seed(shape[2] * 7 + shape[3] + C)
    x = torch.randn(shape, generator=g).to(cuda)
    weight = (torch.randn(2 * C, 1, 5, 5, generator=g) * 0.3).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda)
    outs = {}
    for mode in ("cells", "gather"):
        monkeypatch.setenv("ADVSTEP_CONV0_BWD", mode)
        xa = x.clone().requires_grad_(True)
        y = L.conv5_mfm_pool2(xa, weight, bias)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(cuda)
        (outs[mode],) = torch.autograd.grad(y, xa, gy)
    scale = outs["gather"].abs().max().item()
    assert torch.isfinite(outs["cells"]).all()
    assert (outs["cells"] - outs["gather"]).abs().max().item() <= 4e-6 * max(scale, 1.0)
    # a pooled gradient that is zero except ONE cell next to a tile seam: its 5x5 footprint must come out whole
    if shape[2] >= 40:
        gy1 = torch.zeros_like(gy)
        gy1[0, 0, 17, min(3, gy.shape[3] - 1)] = 1.0
        res = {}
        for mode in ("cells", "gather"):
            monkeypatch.setenv("ADVSTEP_CONV0_BWD", mode)
            xa = x.clone().requires_grad_(True)
            (res[mode],) = torch.autograd.grad(L.conv5_mfm_pool2(xa, weight, bias), xa, gy1)
        assert torch.equal(res["cells"], res["gather"]) and int((res["cells"] != 0).sum()) == 25


@pytest.mark.parametrize("kind", ["conv5", "conv3x3"])
def test_fused_conv_pool_selection_with_nans_ties_and_infinities(L, cuda, kind):
    """The pooling epilogues of the convolution kernels take a short path when no candidate in a wave is NaN and the
    exact ATen rule otherwise: NaNs must propagate exactly as max-feature-map + MaxPool2d do, ties (integer-valued data:
    the convolution is exact) and infinities must select like ATen, and the selection bytes must route the gradient the
    same way."""
    g = torch.Generator().manual_seed(11)
    if kind == "conv5":
        N, Cin, C, H, W, k = 3, 1, 32, 12, 70, 5
    else:
        N, Cin, C, H, W, k = 2, 32, 48, 10, 12, 3
    # small integers: every product and partial sum is exact in float32, so equal candidates are exactly equal
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    weight = torch.randint(-1, 2, (2 * C, Cin, k, k), generator=g).float()
    bias = torch.randint(-1, 2, (2 * C,), generator=g).float()
    if kind == "conv5":       # (a Winograd convolution — MIOpen's and this library's 3x3 — smears non-finite inputs over
        x[0, 0, 3, 5] = float("nan")              # neighbouring tiles through inf - inf in its transforms: no common reference)
        x[1, 0, 7, 30] = float("inf")
        x[1, 0, 2, 3] = float("-inf")
    x, weight, bias = x.to(cuda), weight.to(cuda), bias.to(cuda)
    # reference convolution on the CPU in float64: a direct convolution, so a NaN / inf reaches exactly its receptive
    # field (MIOpen's Winograd / FFT algorithms smear them over neighbouring outputs)
    y_ref = ref_mfm_pool(torch.nn.functional.conv2d(x.cpu().double(), weight.cpu().double(), bias.cpu().double(),
                                                    padding=k // 2)).float().to(cuda)
    xa = x.clone().requires_grad_(True)
    y = L.conv5_mfm_pool2(xa, weight, bias) if kind == "conv5" else L.conv3x3_mfm_pool2(xa, weight, bias)
    assert torch.equal(torch.isnan(y), torch.isnan(y_ref)) and (kind != "conv5" or torch.isnan(y).any())
    finite = torch.isfinite(y_ref)
    assert torch.equal(torch.nan_to_num(y, nan=0.0, posinf=1e30, neginf=-1e30),
                       torch.nan_to_num(y_ref, nan=0.0, posinf=1e30, neginf=-1e30))
    # gradient routing on the samples without NaN / inf (a NaN anywhere poisons ATen's whole convolution backward)
    clean = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float().to(cuda)
    ca, cr = clean.clone().requires_grad_(True), clean.clone().requires_grad_(True)
    yc = L.conv5_mfm_pool2(ca, weight, bias) if kind == "conv5" else L.conv3x3_mfm_pool2(ca, weight, bias)
    yr = ref_mfm_pool(torch.nn.functional.conv2d(cr, weight, bias, padding=k // 2))
    assert torch.equal(yc, yr)
    gy = torch.randint(-2, 3, yr.shape, generator=g).float().to(cuda)
    (ga,) = torch.autograd.grad(yc, ca, gy)
    (gr,) = torch.autograd.grad(yr, cr, gy)
    assert torch.equal(ga, gr) and finite.any()


def test_conv5_mfm_pool2_requires_frozen_weights(L, cuda):
    x = torch.randn(1, 1, 8, 8, device=cuda, requires_grad=True)
    w = torch.randn(4, 1, 5, 5, device=cuda, requires_grad=True)
    y = L.conv5_mfm_pool2(x, w, None)
    with pytest.raises(RuntimeError, match="input gradient only"):
        y.sum().backward()
    with pytest.raises(ValueError):
        L.conv5_mfm_pool2(torch.randn(1, 2, 8, 8, device=cuda), w.detach(), None)


def test_lcnn_with_fused_first_block_agrees_with_miopen_path(cuda, monkeypatch, parity_record):
    """Whole LCNN, attack mode, frozen parameters: fused first block + fused 1x1 blocks vs MIOpen convolutions.  The two
    convolutions round differently, so logits agree to float tolerance and input gradients to a small relative error.
    Measured (profiles/r02_parity.json): logits 3e-8, gradient relative L2 7e-7, no entry off by 1e-3 of the maximum."""
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda)
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    x = (torch.randn(4, 64_600, generator=torch.Generator().manual_seed(1)) * 0.05).to(cuda)
    spec = model._compute_frontend(x).detach()

    def run(conv0):
        monkeypatch.setenv("ADVSTEP_LCNN_CONV0", "1" if conv0 else "0")
        monkeypatch.setenv("ADVSTEP_LCNN_CONV1X1", "1" if conv0 else "0")
        monkeypatch.setenv("ADVSTEP_LCNN_CONV3X3", "1" if conv0 else "0")
        monkeypatch.setenv("ADVSTEP_LCNN_LSTM", "1" if conv0 else "0")
        a = spec.clone().requires_grad_(True)
        z = model._compute_embedding(a)
        (g,) = torch.autograd.grad(z.sum(), a)
        return z.detach(), g

    z0, g0 = run(False)
    z1, g1 = run(True)
    assert (z0 - z1).abs().max().item() <= 3.5e-7
    # different (equally valid) float rounding inside the convolution flips a handful of near-tie winners among the
    # 33 M max-feature-map / pool decisions; each flip re-routes one gradient entry.  Sparse, bounded differences:
    off = (g0 - g1).abs() > 1e-3 * g0.abs().max()
    fig = {"logit_max_abs": (z0 - z1).abs().max().item(), "grad_rel_l2": (g0 - g1).norm().item() / g0.norm().item(),
           "grad_entries": g0.numel(), "grad_entries_off_by_1e-3_of_max": int(off.sum()),
           "grad_max_abs_over_max": ((g0 - g1).abs().max() / g0.abs().max()).item()}
    parity_record["lcnn_fused_kernels_vs_miopen_path"] = fig
    # a near-tie max-feature-map / pool winner going the other way would re-route one gradient entry (~1e-3 of the relative
    # L2 each); none does on this input.  Bounds = 10x the measured figures (logits 3.4e-8, relative L2 6.9e-7, worst entry
    # 6.7e-7 of the largest): a kernel that loses three digits fails
    assert fig["grad_entries_off_by_1e-3_of_max"] == 0, fig
    assert fig["grad_rel_l2"] <= 7e-6 and fig["grad_max_abs_over_max"] <= 7e-6, fig
    for p in model.parameters():
        p.requires_grad_(True)


# ---- fused 1x1 blocks: Conv2d(Cin, 2C, 1x1) -> MFM -----------------------------------------------------------------------------

@pytest.mark.parametrize("N,Cin,C,H,W", [(2, 32, 32, 12, 10), (3, 48, 48, 7, 9), (2, 64, 64, 50, 10), (1, 32, 5, 1, 1),
                                         (2, 32, 32, 202, 40), (2, 48, 7, 3, 67)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_conv1x1_mfm_matches_float64_reference(L, cuda, N, Cin, C, H, W, with_bias):
    g = torch.Generator().manual_seed(N * 1000 + Cin * 10 + C + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    weight = (torch.randn(2 * C, Cin, 1, 1, generator=g) * 0.2).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda) if with_bias else None
    xr = x.double().requires_grad_(True)
    y_ref = ref_mfm(torch.nn.functional.conv2d(xr, weight.double(), bias.double() if with_bias else None))
    gy = torch.randn(y_ref.shape, generator=g).to(cuda)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy.double())
    xa = x.clone().requires_grad_(True)
    y = L.conv1x1_mfm(xa, weight, bias)
    (gx,) = torch.autograd.grad(y, xa, gy)
    assert y.shape == y_ref.shape
    assert (y.double() - y_ref).abs().max().item() <= 2e-5
    assert (gx.double() - gx_ref).abs().max().item() <= 2e-5 * max(gx_ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("Cin,C", [(48, 48), (32, 5), (48, 7), (64, 64), (64, 33)])
def test_conv1x1_mfm_writes_nothing_outside_its_tensors(cuda, Cin, C):
    """The kernels address channel rows through scalar offsets of a buffer descriptor, which the hardware does not
    range-check: accumulator rows beyond C (forward) / Cin (backward) — present whenever these are not multiples of 32 —
    must not be stored.  Outputs are carved out of sentinel-filled buffers and the sentinels checked afterwards."""
    from audio_deepfake_adversarial_attacks_amd import _lib
    lib = _lib.load()
    N, P = 3, 77
    g = torch.Generator().manual_seed(Cin + C)
    x = torch.randn(N, Cin, P, generator=g).to(cuda)
    w = (torch.randn(2 * C, Cin, generator=g) * 0.2).to(cuda)
    nsel = lib.advstep_conv1x1_mfm_sel_bytes(N, C, P) // 4       # opaque layout: its size comes from the library
    assert nsel == N * ((P + 31) // 32) * 64 * (2 if C <= 32 else 4) // 4
    pad = 64 * P
    ybuf = torch.full((pad + N * C * P + pad,), 7.5, device=cuda)
    sbuf = torch.full((pad + nsel + pad,), 0x5A5A5A5A, dtype=torch.int32, device=cuda)
    y, sel = ybuf[pad:pad + N * C * P], sbuf[pad:pad + nsel]
    stream = torch.cuda.current_stream(cuda).cuda_stream
    assert lib.advstep_conv1x1_mfm_forward_f32(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sel.data_ptr(),
                                               N, Cin, C, P, stream) == 0
    torch.cuda.synchronize()
    assert (ybuf[:pad] == 7.5).all() and (ybuf[pad + N * C * P:] == 7.5).all()
    assert (sbuf[:pad] == 0x5A5A5A5A).all() and (sbuf[pad + nsel:] == 0x5A5A5A5A).all()
    conv = torch.einsum("ok,nkp->nop", w.double(), x.double())
    want = torch.maximum(conv[:, :C], conv[:, C:])
    assert (y.view(N, C, P).double() - want).abs().max().item() <= 2e-5
    gy = torch.randn(N, C, P, generator=g).to(cuda)
    xpad = 64 * P
    gbuf = torch.full((xpad + N * Cin * P + xpad,), -3.25, device=cuda)
    gx = gbuf[xpad:xpad + N * Cin * P]
    assert lib.advstep_conv1x1_mfm_backward_f32(gy.data_ptr(), sel.data_ptr(), w.data_ptr(), None, gx.data_ptr(), N, Cin,
                                                C, P, stream) == 0
    torch.cuda.synchronize()
    assert (gbuf[:xpad] == -3.25).all() and (gbuf[xpad + N * Cin * P:] == -3.25).all()
    took_b = conv[:, C:] > conv[:, :C]
    gfull = torch.cat([torch.where(took_b, 0.0, gy.double()), torch.where(took_b, gy.double(), 0.0)], dim=1)
    want_gx = torch.einsum("ok,nop->nkp", w.double(), gfull)
    assert (gx.view(N, Cin, P).double() - want_gx).abs().max().item() <= 2e-5 * max(want_gx.abs().max().item(), 1.0)


def test_conv1x1_mfm_rejects_unsupported_and_unfrozen(L, cuda):
    w = torch.randn(8, 16, 1, 1, device=cuda)
    with pytest.raises(ValueError, match="Cin"):
        L.conv1x1_mfm(torch.randn(1, 16, 4, 4, device=cuda), w, None)
    x = torch.randn(1, 32, 4, 4, device=cuda, requires_grad=True)
    w = torch.randn(8, 32, 1, 1, device=cuda, requires_grad=True)
    with pytest.raises(RuntimeError, match="input gradient only"):
        L.conv1x1_mfm(x, w, None).sum().backward()


# ---- LSTM layer --------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("T,B,I", [(25, 6, 160), (1, 2, 160), (7, 128, 160), (3, 1, 32)])
def test_lstm_layer_matches_torch_lstm(L, cuda, T, B, I):
    torch.manual_seed(T * 100 + B)
    H = 80
    ref = torch.nn.LSTM(I, H, bidirectional=True).double()
    x = torch.randn(T, B, I, dtype=torch.float64, requires_grad=True)
    out_ref, _ = ref(x)
    dout = torch.randn_like(out_ref)
    (dx_ref,) = torch.autograd.grad(out_ref, x, dout)

    w_ih = torch.cat([ref.weight_ih_l0, ref.weight_ih_l0_reverse]).float().to(cuda).contiguous()
    w_hh = torch.stack([ref.weight_hh_l0, ref.weight_hh_l0_reverse]).float().to(cuda).contiguous()
    bias = torch.cat([ref.bias_ih_l0 + ref.bias_hh_l0, ref.bias_ih_l0_reverse + ref.bias_hh_l0_reverse]).float().to(cuda)
    xg = x.detach().float().to(cuda).requires_grad_(True)
    out = L.lstm_layer(xg, w_ih.detach(), w_hh.detach(), bias.detach())
    (dx,) = torch.autograd.grad(out, xg, dout.float().to(cuda))
    assert out.shape == (T, B, 2 * H)
    assert (out.double().cpu() - out_ref).abs().max().item() <= 2e-6
    assert (dx.double().cpu() - dx_ref).abs().max().item() <= 2e-5 * max(dx_ref.abs().max().item(), 1.0)


def test_blstm_layer_uses_kernel_when_frozen_and_matches_miopen(cuda, monkeypatch):
    from audio_deepfake_adversarial_attacks_amd.models.lcnn import BLSTMLayer
    torch.manual_seed(3)
    layer = BLSTMLayer(160, 160).to(cuda).train()
    x = torch.randn(16, 25, 160, device=cuda)

    def run(kernel, frozen):
        monkeypatch.setenv("ADVSTEP_LCNN_LSTM", "1" if kernel else "0")
        for p in layer.parameters():
            p.requires_grad_(not frozen)
        a = x.clone().requires_grad_(True)
        y = layer(a)
        (g,) = torch.autograd.grad(y, a, torch.ones_like(y))
        return y.detach(), g

    y0, g0 = run(False, False)          # MIOpen
    y1, g1 = run(True, True)            # HIP kernel
    y2, g2 = run(True, False)           # parameters need grad -> falls back to MIOpen, identical to y0
    assert torch.equal(y2, y0)
    assert (y1 - y0).abs().max().item() <= 2e-6 and (g1 - g0).abs().max().item() <= 2e-5 * max(g0.abs().max().item(), 1.0)
    for p in layer.parameters():
        p.requires_grad_(True)


# ---- eval-mode BatchNorm folded into the block kernels ---------------------------------------------------------------------------

def _bn_ref(y, mean, var, eps=1e-5):
    return torch.nn.functional.batch_norm(y, mean, var, None, None, False, 0.1, eps)


@pytest.mark.parametrize("kind", ["mfm", "mfm_pool2", "conv1x1"])
def test_folded_batchnorm_matches_aten(L, cuda, kind):
    g = torch.Generator().manual_seed(11)
    C = 32
    mean = torch.randn(C, generator=g).to(cuda)
    var = (torch.rand(C, generator=g) + 0.5).to(cuda)
    bn = (mean, (1.0 / torch.sqrt(var + 1e-5)).contiguous())
    if kind == "conv1x1":
        x = torch.randn(3, 32, 9, 10, generator=g).to(cuda).requires_grad_(True)
        w = (torch.randn(2 * C, 32, 1, 1, generator=g) * 0.2).to(cuda)
        b = torch.randn(2 * C, generator=g).to(cuda)
        ref = _bn_ref(ref_mfm(torch.nn.functional.conv2d(x, w, b)), mean, var)
        got = L.conv1x1_mfm(x, w, b, bn)
        tol = 2e-5
    else:
        x = make((3, 2 * C, 10, 12), cuda, 12).requires_grad_(True)
        b = torch.randn(2 * C, generator=g).to(cuda)
        ref = _bn_ref((ref_mfm_pool if kind == "mfm_pool2" else ref_mfm)(x, b), mean, var)
        got = (L.mfm_pool2 if kind == "mfm_pool2" else L.mfm)(x, b, bn)
        tol = 1e-6   # same operations; ATen may use rsqrt for invstd
    gy = torch.randn(ref.shape, generator=g).to(cuda)
    (g_ref,) = torch.autograd.grad(ref, x, gy)
    (g_got,) = torch.autograd.grad(got, x, gy)
    assert (got - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)
    assert (g_got - g_ref).abs().max().item() <= max(tol, 2e-6) * max(g_ref.abs().max().item(), 1.0)


# ---- fused 3x3 blocks on the matrix cores: Conv2d(Cin, 2C, 3x3, pad 1) -> MFM -> MaxPool2d(2, 2) [-> BN] ---------------------

@pytest.mark.parametrize("N,Cin,C,H,W", [(2, 32, 48, 12, 10), (1, 48, 64, 11, 9), (2, 64, 32, 50, 10), (3, 32, 32, 6, 8),
                                         (2, 32, 48, 202, 40), (1, 48, 64, 101, 20), (5, 32, 16, 2, 2)])
@pytest.mark.parametrize("with_bias,with_bn", [(True, True), (False, False)])
def test_conv3x3_mfm_pool2_matches_float64_reference(L, cuda, N, Cin, C, H, W, with_bias, with_bn):
    """Winograd F(2x2, 3x3) in fp32 against a float64 direct convolution: values within 1e-5 of the output scale (the
    same error class as MIOpen's own fp32 Winograd kernel); the input gradient is compared through the kernel's OWN
    selection (a last-bit difference may pick another pooling winner at a near tie, which is a different, equally valid
    subgradient), so the float64 reference is evaluated with the recorded winners."""
    g = torch.Generator().manual_seed(N * 1000 + Cin * 10 + C + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    weight = (torch.randn(2 * C, Cin, 3, 3, generator=g) * 0.1).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda) if with_bias else None
    bn = None
    if with_bn:
        mean = torch.randn(C, generator=g).to(cuda)
        var = (torch.rand(C, generator=g) + 0.5).to(cuda)
        bn = (mean, (1.0 / torch.sqrt(var + 1e-5)).contiguous())
    xr = x.double().requires_grad_(True)
    conv = torch.nn.functional.conv2d(xr, weight.double(), bias.double() if with_bias else None, padding=1)
    y_ref = ref_mfm_pool(conv)
    if with_bn:
        y_ref = (y_ref - bn[0].double().view(1, -1, 1, 1)) * bn[1].double().view(1, -1, 1, 1)
    xa = x.clone().requires_grad_(True)
    y = L.conv3x3_mfm_pool2(xa, weight, bias, bn)
    assert y.shape == y_ref.shape == (N, C, H // 2, W // 2)
    scale = max(y_ref.abs().max().item(), 1.0)
    assert (y.double() - y_ref).abs().max().item() <= 1e-5 * scale
    if y.numel() == 0:
        return
    gy = torch.randn(y_ref.shape, generator=g).to(cuda)
    (gx,) = torch.autograd.grad(y, xa, gy)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy.double())
    err = (gx.double() - gx_ref).abs()
    tol = 2e-5 * max(gx_ref.abs().max().item(), 1.0)
    # at most a handful of pooling windows may resolve a near tie differently
    assert (err > tol).float().mean().item() <= 1e-3, ((err > tol).sum().item(), err.max().item())
    assert (err.max().item() <= tol) or (err > tol).sum().item() <= 9 * 8 * Cin


@pytest.mark.parametrize("N,C,H,W", [(4, 32, 50, 10), (3, 32, 11, 9), (2, 64, 50, 10), (3, 64, 7, 13), (1, 64, 2, 2)])
def test_conv3x3_pool2_backward_half_slice_launch_equals_the_single_launch(L, cuda, monkeypatch, N, C, H, W):
    """a one-slice input gradient (Cin = 32 output rows) on a plane that does not fill the chip is launched as
    its two 16-row halves (round 5, `ga.halves`; every NT = 1 kernel then reads its A operand as one component of the
    stored pair).  A/B through the switch the launcher reads at call time: the same bits, for the resident weights
    (K = 2 C = 64) and the streamed ones (K = 128), on even and odd planes."""
    g = torch.Generator().manual_seed(C * 100 + H * 10 + W)
    Cin = 32
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    weight = (torch.randn(2 * C, Cin, 3, 3, generator=g) * 0.1).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda)
    mean, var = torch.randn(C, generator=g).to(cuda), (torch.rand(C, generator=g) + 0.5).to(cuda)
    bn = (mean, (1.0 / torch.sqrt(var + 1e-5)).contiguous())
    gy = torch.randn(N, C, H // 2, W // 2, generator=g).to(cuda)
    got = {}
    for halves in ("1", "0"):
        monkeypatch.setenv("ADVSTEP_WINO_HALVES", halves)
        xa = x.clone().requires_grad_(True)
        y = L.conv3x3_mfm_pool2(xa, weight, bias, bn)
        (got[halves],) = torch.autograd.grad(y, xa, gy)
    assert torch.equal(got["1"], got["0"])
    # and the gradient is the right one (float64 direct convolution through the kernel's own winners, as in the test above)
    xr = x.double().requires_grad_(True)
    conv = torch.nn.functional.conv2d(xr, weight.double(), bias.double(), padding=1)
    y_ref = (ref_mfm_pool(conv) - mean.double().view(1, -1, 1, 1)) * bn[1].double().view(1, -1, 1, 1)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy.double())
    err = (got["1"].double() - gx_ref).abs()
    tol = 2e-5 * max(gx_ref.abs().max().item(), 1.0)
    assert (err.max().item() <= tol) or (err > tol).sum().item() <= 9 * 8 * Cin


def test_conv3x3_backward_data_matches_aten(L, cuda):
    """The input-gradient convolution on its own (K = 96 and K = 128: the double-buffered weight stream)."""
    from audio_deepfake_adversarial_attacks_amd import _lib
    from audio_deepfake_adversarial_attacks_amd.lcnn_ops import _prepared_weights
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for N, Cin, Cout, H, W in ((2, 32, 96, 20, 12), (1, 48, 128, 13, 7), (3, 32, 64, 10, 10)):
        weight = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).to(cuda)
        gout = torch.randn(N, Cout, H, W, generator=g).to(cuda)
        want = torch.nn.grad.conv2d_input((N, Cin, H, W), weight.double(), gout.double(), padding=1)
        gx = torch.empty(N, Cin, H, W, device=cuda)
        st = lib.advstep_conv3x3_backward_data_f32(gout.data_ptr(), _prepared_weights(weight, 1).data_ptr(), gx.data_ptr(), N, Cin,
                                                   Cout, H, W, torch.cuda.current_stream().cuda_stream)
        assert st == 0
        assert (gx.double() - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0), (N, Cin, Cout)


def test_conv3x3_rejects_unsupported(L, cuda):
    assert L.conv3x3_supported(32, 96) and L.conv3x3_supported(48, 128) and L.conv3x3_supported(64, 64)
    assert not L.conv3x3_supported(1, 64) and not L.conv3x3_supported(32, 48) and not L.conv3x3_supported(40, 64)
    x = torch.randn(1, 32, 4, 4, device=cuda, requires_grad=True)
    w = torch.randn(64, 32, 3, 3, device=cuda, requires_grad=True)
    y = L.conv3x3_mfm_pool2(x, w, None)
    (gx,) = torch.autograd.grad(y.sum(), x)          # input gradient only: the weight gets none
    assert gx.shape == x.shape


@pytest.mark.parametrize("N,Cin,C,H,W", [(2, 64, 32, 50, 10), (3, 32, 16, 7, 9), (1, 48, 64, 4, 6)])
@pytest.mark.parametrize("with_bn", [True, False])
def test_conv3x3_mfm_without_pool_matches_float64_reference(L, cuda, N, Cin, C, H, W, with_bn):
    """The un-pooled block (lcnn.py:142-144): same Winograd kernel, max-feature-map + BatchNorm epilogue, one selection
    byte per 2x2 tile; 1e-5 of the output scale, gradient within 2e-5 except at near-tie flips."""
    g = torch.Generator().manual_seed(N * 100 + Cin + C + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    weight = (torch.randn(2 * C, Cin, 3, 3, generator=g) * 0.1).to(cuda)
    bias = torch.randn(2 * C, generator=g).to(cuda)
    bn = None
    if with_bn:
        mean = torch.randn(C, generator=g).to(cuda)
        var = (torch.rand(C, generator=g) + 0.5).to(cuda)
        bn = (mean, (1.0 / torch.sqrt(var + 1e-5)).contiguous())
    xr = x.double().requires_grad_(True)
    y_ref = ref_mfm(torch.nn.functional.conv2d(xr, weight.double(), bias.double(), padding=1))
    if with_bn:
        y_ref = (y_ref - bn[0].double().view(1, -1, 1, 1)) * bn[1].double().view(1, -1, 1, 1)
    xa = x.clone().requires_grad_(True)
    y = L.conv3x3_mfm(xa, weight, bias, bn)
    assert y.shape == y_ref.shape
    assert (y.double() - y_ref).abs().max().item() <= 1e-5 * max(y_ref.abs().max().item(), 1.0)
    gy = torch.randn(y_ref.shape, generator=g).to(cuda)
    (gx,) = torch.autograd.grad(y, xa, gy)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy.double())
    err = (gx.double() - gx_ref).abs()
    tol = 2e-5 * max(gx_ref.abs().max().item(), 1.0)
    assert (err > tol).float().mean().item() <= 1e-3


def test_conv3x3_batch_is_split_for_the_32bit_buffer_descriptor(L, cuda, monkeypatch):
    """The Winograd kernels address a tensor through a 32-bit buffer descriptor (< 2 GiB per call); larger batches are
    split by the wrapper.  Forcing the split on a small problem must not change a single bit."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 32, 12, 10, generator=g).to(cuda)
    w = (torch.randn(96, 32, 3, 3, generator=g) * 0.1).to(cuda)
    b = torch.randn(96, generator=g).to(cuda)
    gy = torch.randn(5, 48, 6, 5, generator=g).to(cuda)

    def run():
        xa = x.clone().requires_grad_(True)
        y = L.conv3x3_mfm_pool2(xa, w, b)
        (gx,) = torch.autograd.grad(y, xa, gy)
        return y.detach(), gx

    y0, g0 = run()
    assert L._batch_chunks(5, 1) == [(0, 5)] and L._batch_chunks(5, (1 << 29) // 2) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
    monkeypatch.setattr(L, "_batch_chunks", lambda N, per: [(0, 2), (2, 5)])
    y1, g1 = run()
    assert torch.equal(y0, y1) and torch.equal(g0, g1)


# ---- GRU layer (SpecRNet) --------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("T,B,I,layers", [(50, 6, 64, 2), (1, 2, 64, 1), (7, 128, 64, 2), (3, 1, 128, 1)])
def test_gru_layer_matches_torch_gru(L, cuda, T, B, I, layers):
    torch.manual_seed(T * 100 + B)
    H = 64
    ref = torch.nn.GRU(I, H, num_layers=layers, bidirectional=True).double()
    x = torch.randn(T, B, I, dtype=torch.float64, requires_grad=True)
    out_ref, _ = ref(x)
    dout = torch.randn_like(out_ref)
    (dx_ref,) = torch.autograd.grad(out_ref, x, dout)

    xg = x.detach().float().to(cuda).requires_grad_(True)
    seq = xg
    for layer in range(layers):
        sfx = [f"_l{layer}", f"_l{layer}_reverse"]
        w_ih = torch.cat([getattr(ref, "weight_ih" + s) for s in sfx]).float().to(cuda).contiguous()
        w_hh = torch.stack([getattr(ref, "weight_hh" + s) for s in sfx]).float().to(cuda).contiguous()
        b_ih = torch.cat([getattr(ref, "bias_ih" + s) for s in sfx]).float().to(cuda).contiguous()
        b_hh = torch.stack([getattr(ref, "bias_hh" + s) for s in sfx]).float().to(cuda).contiguous()
        seq = L.gru_layer(seq, w_ih.detach(), w_hh.detach(), b_ih.detach(), b_hh.detach())
    (dx,) = torch.autograd.grad(seq, xg, dout.float().to(cuda))
    assert seq.shape == (T, B, 2 * H)
    assert (seq.double().cpu() - out_ref).abs().max().item() <= 3e-6
    assert (dx.double().cpu() - dx_ref).abs().max().item() <= 3e-5 * max(dx_ref.abs().max().item(), 1.0)


def test_specrnet_uses_gru_kernel_when_frozen_and_matches_miopen(cuda, monkeypatch):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    monkeypatch.setenv("ADVSTEP_SPECRNET_ELEM", "0")     # this test isolates the GRU kernels (the elementwise fusions have their own)
    torch.manual_seed(5)
    model = get_model("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, str(cuda)).to(cuda)
    model.train()                      # attack mode (attack.py:311-319): train, BatchNorm / Dropout in eval
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    spec = torch.randn(6, 2, 80, 404, device=cuda)

    def run(kernel, frozen):
        monkeypatch.setenv("ADVSTEP_SPECRNET_GRU", "1" if kernel else "0")
        for p in model.parameters():
            p.requires_grad_(not frozen)
        a = spec.clone().requires_grad_(True)
        z = model._compute_embedding(a)
        (g,) = torch.autograd.grad(z.sum(), a)
        return z.detach(), g

    z0, g0 = run(False, True)           # MIOpen
    z1, g1 = run(True, True)            # HIP kernels
    z2, g2 = run(True, False)           # parameters need grad -> torch.nn.GRU, identical to z0
    assert torch.equal(z2, z0)
    assert (z1 - z0).abs().max().item() <= 2e-5 * max(z0.abs().max().item(), 1.0)
    assert (g1 - g0).abs().max().item() <= 1e-4 * max(g0.abs().max().item(), 1e-6)
    for p in model.parameters():
        p.requires_grad_(True)


def test_lcnn_tail_as_one_node_matches_the_separate_ops(cuda, monkeypatch, parity_record):
    """lcnn_ops.lcnn_tail (pack -> two recurrent layers -> skip + mean + Linear as ONE autograd node, the mean's gradient read
    with a zero frame stride, the two gradients of `hidden` summed while they return to the convolution's layout) against the
    same fused LSTM kernels driven by the separate torch ops (src/models/lcnn.py:196-205): logits and the gradient w.r.t. the
    convolution trunk's output.  Same kernels for the recurrences; only the summation order of the mean / Linear differs."""
    from audio_deepfake_adversarial_attacks_amd.models import lcnn
    torch.manual_seed(7)
    body = lcnn.BaseLCNN(input_channels=1, num_coefficients=80).to(cuda)
    body.train()
    for m in body.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    for p in body.parameters():
        p.requires_grad_(False)
    spec = torch.randn(6, 1, 80, 404, generator=torch.Generator().manual_seed(8)).to(cuda)

    def run(tail):
        monkeypatch.setenv("ADVSTEP_LCNN_TAIL", "1" if tail else "0")
        a = spec.clone().requires_grad_(True)
        z = body(a)
        (g,) = torch.autograd.grad((z * torch.arange(1, 7, device=cuda).view(6, 1)).sum(), a)
        return z.detach(), g

    z0, g0 = run(False)
    z1, g1 = run(True)
    fig = {"logit_max_abs": (z0 - z1).abs().max().item(), "grad_rel_l2": ((g0 - g1).norm() / g0.norm()).item(),
           "grad_max_abs_over_max": ((g0 - g1).abs().max() / g0.abs().max()).item()}
    parity_record["lcnn_tail_one_node_vs_separate_ops"] = fig
    assert z1.shape == (6, 1) and fig["logit_max_abs"] <= 3e-7, fig
    assert fig["grad_rel_l2"] <= 5e-6 and fig["grad_max_abs_over_max"] <= 5e-6, fig
    with torch.no_grad():                                  # the scoring pass (no autograd) takes the same node
        monkeypatch.setenv("ADVSTEP_LCNN_TAIL", "1")
        assert (body.eval()(spec) - body(spec)).abs().max().item() == 0.0
