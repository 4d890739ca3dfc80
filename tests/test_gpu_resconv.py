"""`-m gpu`: the residual-block convolutions of include/advstep_detector.h (advstep_resconv_*, Winograd F(2x2, 3x3) on the
fp32 matrix cores) against ATen / MIOpen: the operator itself (3x3 over x1 + 1x1 over x2, shift, LeakyReLU), its pooled
form, the transposed (input-gradient) preparation, and SpecRNet's whole Residual_block2D forward + input gradient.
Winograd in fp32 rounds differently from a direct convolution, so values are compared with a tolerance relative to the
output scale (written at each assert) and pooling selections are compared where the window's top two are not a near tie."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL = 2e-5          # max |difference| / max |reference| of one convolution


@pytest.fixture(scope="module")
def D(cuda):
    from audio_deepfake_adversarial_attacks_amd import detector_ops
    return detector_ops


def rnd(shape, seed, cuda, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(cuda)


# (N, K1, K2, rows, H, W): SpecRNet's layers at small sizes, odd sizes, padded reductions (K not a multiple of 8),
# half-empty and single-tile last slices, a streamed reduction (K1 + K2 > 64)
SHAPES = [(2, 20, 2, 20, 16, 24), (3, 64, 20, 64, 20, 101), (2, 64, 0, 64, 5, 25), (1, 20, 0, 64, 7, 9), (2, 64, 64, 20, 10, 12),
          (1, 4, 0, 2, 6, 8), (2, 2, 0, 20, 8, 10), (2, 128, 0, 48, 9, 11), (1, 1, 0, 1, 3, 3), (2, 40, 0, 2, 6, 6),
          (1, 20, 20, 33, 4, 6)]


def reference(x1, x2, w3, w1, shift, slope):
    y = F.conv2d(x1.double(), w3.double(), None, 1, 1)
    if x2 is not None:
        y = y + F.conv2d(x2.double(), w1.double()[:, :, None, None])
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    return F.leaky_relu(y, slope)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("slope,with_shift", [(1.0, False), (0.3, True)])
def test_resconv_matches_direct_convolution(D, cuda, shape, slope, with_shift):
    N, K1, K2, R, H, W = shape
    x1, w3 = rnd((N, K1, H, W), 1, cuda), rnd((R, K1, 3, 3), 2, cuda, 0.2)
    x2, w1 = (rnd((N, K2, H, W), 3, cuda), rnd((R, K2), 4, cuda, 0.3)) if K2 else (None, None)
    shift = rnd((R,), 5, cuda) if with_shift else None
    U = D.resconv_prepare(w3, w1)
    y = D.resconv(x1, x2, U, R, shift, slope)
    ref = reference(x1, x2, w3, w1, shift, slope)
    assert y.shape == ref.shape
    err = (y.double() - ref).abs().max().item()
    assert err <= REL * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("shape", SHAPES)
def test_resconv_transposed_preparation_is_the_input_gradient(D, cuda, shape):
    """transpose=True with forward weights w3 (K1, rows, 3, 3) [, w1 (K2, rows)] and kscale: d/dx of
    sum(g1 * (conv3x3(x, w3) * kscale)) + sum(g2 * conv1x1(x, w1))."""
    N, K1, K2, R, H, W = shape
    g1, w3, ks = rnd((N, K1, H, W), 1, cuda), rnd((K1, R, 3, 3), 2, cuda, 0.2), rnd((K1,), 6, cuda)
    g2, w1 = (rnd((N, K2, H, W), 3, cuda), rnd((K2, R), 4, cuda, 0.3)) if K2 else (None, None)
    U = D.resconv_prepare(w3, w1, kscale=ks, transpose=True)
    gx = D.resconv(g1, g2, U, R)
    x = torch.zeros((N, R, H, W), dtype=torch.float64, device=cuda, requires_grad=True)
    out = (F.conv2d(x, w3.double(), None, 1, 1) * ks.double().view(1, -1, 1, 1) * g1.double()).sum()
    if K2:
        out = out + (F.conv2d(x, w1.double()[:, :, None, None]) * g2.double()).sum()
    (ref,) = torch.autograd.grad(out, x)
    err = (gx.double() - ref).abs().max().item()
    assert err <= REL * ref.abs().max().item(), (err, ref.abs().max().item())


def test_resconv_row_scale_is_folded_into_the_weights(D, cuda):
    x, w3, rs = rnd((2, 20, 9, 10), 1, cuda), rnd((64, 20, 3, 3), 2, cuda, 0.2), rnd((64,), 3, cuda)
    y = D.resconv(x, None, D.resconv_prepare(w3, rscale=rs), 64)
    ref = F.conv2d(x.double(), w3.double(), None, 1, 1) * rs.double().view(1, -1, 1, 1)
    assert (y.double() - ref).abs().max().item() <= REL * ref.abs().max().item()


def test_padded_reduction_channels_do_not_read_the_next_sample(D, cuda):
    """K1 + K2 = 22 is padded to 24: k-step 5 holds two channels that do not exist.  Their taps must come from the buffer
    descriptor's out-of-range zero, not from the neighbouring sample's memory (here: NaN)."""
    N, K1, K2, R, H, W = 2, 20, 2, 20, 8, 12
    x1, x2 = rnd((N, K1, H, W), 1, cuda), rnd((N, K2, H, W), 2, cuda)
    x1[1], x2[1] = float("nan"), float("nan")
    w3, w1 = rnd((R, K1, 3, 3), 3, cuda, 0.2), rnd((R, K2), 4, cuda)
    y = D.resconv(x1, x2, D.resconv_prepare(w3, w1), R)
    assert torch.isfinite(y[0]).all() and torch.isnan(y[1]).all()
    ref = reference(x1[:1], x2[:1], w3, w1, None, 1.0)
    assert (y[:1].double() - ref).abs().max().item() <= REL * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 20, 2, 20, 16, 24), (3, 64, 20, 64, 20, 101), (2, 64, 0, 64, 5, 25), (1, 20, 2, 20, 7, 9),
                                   (2, 32, 0, 40, 2, 2), (1, 8, 0, 3, 1, 6)])
def test_resconv_pool2_matches_maxpool_of_the_convolution(D, cuda, shape):
    N, K1, K2, R, H, W = shape
    x1, w3 = rnd((N, K1, H, W), 1, cuda), rnd((R, K1, 3, 3), 2, cuda, 0.2)
    x2, w1 = (rnd((N, K2, H, W), 3, cuda), rnd((R, K2), 4, cuda, 0.3)) if K2 else (None, None)
    bias = rnd((R,), 5, cuda)
    y, sel = D.resconv_pool2(x1, x2, D.resconv_prepare(w3, w1), R, bias)
    full = reference(x1, x2, w3, w1, bias, 1.0)
    Ho, Wo = H // 2, W // 2
    assert y.shape == (N, R, Ho, Wo)
    if Ho * Wo == 0:
        return
    ref, ref_idx = F.max_pool2d(full, 2, return_indices=True)
    tol = REL * full.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= tol
    # selection byte = 2 * dh + dw of the winner; compare where the window's best beats its runner-up by more than 2 tol
    win = full[:, :, :2 * Ho, :2 * Wo].reshape(N, R, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, R, Ho, Wo, 4)
    top = win.topk(2, dim=-1).values
    clear = (top[..., 0] - top[..., 1]) > 2 * tol
    ref_code = (((ref_idx // W) % 2) * 2 + (ref_idx % W) % 2).to(torch.uint8)
    got = sel[:N * R * Ho * Wo].view(N, R, Ho, Wo)
    assert clear.float().mean().item() > 0.9
    assert torch.equal(got[clear], ref_code[clear])
    assert (got < 4).all()
    # and the byte is consistent with the value the kernel wrote, everywhere
    picked = win.gather(-1, got.long().unsqueeze(-1)).squeeze(-1)
    assert (picked - y.double()).abs().max().item() <= tol


@pytest.mark.parametrize("shape", [(2, 20, 20, 16, 24), (3, 64, 64, 20, 101), (2, 64, 64, 5, 25), (1, 20, 20, 7, 9), (2, 12, 40, 2, 2),
                                   (1, 8, 3, 1, 6), (2, 4, 2, 6, 8)])
@pytest.mark.parametrize("with_h", [False, True])
def test_pooled_gradient_source_equals_the_dense_path(D, cuda, shape, with_h):
    """advstep_resconv_pooled_grad_f32 expands the pooled gradient inside the operand load and applies LeakyReLU's backward in
    the epilogue: the same numbers in the same order as unpooling (advstep_maxpool2_backward_f32), the dense convolution and
    the elementwise backward pass — bit for bit."""
    N, K, R, H, W = shape
    full = rnd((N, K, H, W), 1, cuda)
    U = D.resconv_prepare(rnd((K, R, 3, 3), 2, cuda, 0.2), transpose=True)
    h = rnd((N, R, H, W), 3, cuda) if with_h else None
    if h is not None:
        h[0, 0, 0, 0] = 0.0                      # leaky_relu_backward at exactly 0 takes the slope
    if H // 2 == 0 or W // 2 == 0:
        gy = torch.zeros((N, K, H // 2, W // 2), device=cuda)
        sel = torch.zeros(1, dtype=torch.uint8, device=cuda)
        assert not D.resconv_pooled_grad(gy, sel, U, R, H, W, h, 0.3).any()
        return
    _, sel = D._add_maxpool2_raw(full, None, None)
    gy = rnd((N, K, H // 2, W // 2), 4, cuda)
    got = D.resconv_pooled_grad(gy, sel, U, R, H, W, h, 0.3)
    dense = torch.empty_like(full)
    from audio_deepfake_adversarial_attacks_amd import _lib
    from audio_deepfake_adversarial_attacks_amd.hip_ops import _stream
    _lib.check(_lib.load().advstep_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), dense.data_ptr(), N, K, H, W, _stream(cuda)),
               "advstep_maxpool2_backward_f32")
    ref = D.resconv(dense, None, U, R)
    if h is not None:
        ref = ref * torch.where(h > 0, 1.0, 0.3).to(ref.dtype)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(2, 2, 20, 16, 24), (2, 1, 20, 7, 9), (1, 2, 5, 1, 1), (2, 2, 20, 80, 404), (3, 1, 3, 2, 3), (1, 2, 7, 5, 4)])
def test_few_input_channel_convolution_matches_aten(D, cuda, shape):
    N, Cin, Cout, H, W = shape
    x, w, shift = rnd((N, Cin, H, W), 1, cuda), rnd((Cout, Cin, 3, 3), 2, cuda, 0.3), rnd((Cout,), 3, cuda)
    y = D.conv3x3_fewin(x, w, shift, 0.3)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), shift.double(), 1, 1), 0.3)
    assert (y.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()      # direct fp32 sums of 9-18 terms
    y1 = D.conv3x3_fewin(x, w, None, 1.0)
    ref1 = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert (y1.double() - ref1).abs().max().item() <= 2e-6 * ref1.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 20, 2, 16, 24), (2, 20, 1, 7, 9), (1, 5, 2, 1, 1), (2, 20, 2, 80, 404), (3, 3, 1, 2, 3), (1, 7, 2, 5, 4)])
@pytest.mark.parametrize("pooled", [False, True])
def test_few_output_row_gradient_matches_autograd(D, cuda, shape, pooled):
    """gx = d/dx [ sum(g1 * conv3x3(x, w3)) + sum(unpool(gp, sel) * conv1x1(x, wd)) ]."""
    N, K, R, H, W = shape
    g1, w3 = rnd((N, K, H, W), 1, cuda), rnd((K, R, 3, 3), 2, cuda, 0.3)
    x = torch.zeros((N, R, H, W), dtype=torch.float64, device=cuda, requires_grad=True)
    out = (F.conv2d(x, w3.double(), None, 1, 1) * g1.double()).sum()
    gp = sel = wd = None
    if pooled:
        full = rnd((N, K, H, W), 3, cuda)
        gp, wd = rnd((N, K, H // 2, W // 2), 4, cuda), rnd((K, R), 5, cuda)
        _, sel = D._add_maxpool2_raw(full, None, None)
        if gp.numel():
            dense = torch.autograd.grad(F.max_pool2d(full.requires_grad_(True), 2), full, gp)[0]
            out = out + (F.conv2d(x, wd.double()[:, :, None, None]) * dense.double()).sum()
    (ref,) = torch.autograd.grad(out, x)
    gx = D.conv3x3_fewout_grad(g1, w3, gp, sel, wd)
    assert (gx.double() - ref).abs().max().item() <= 4e-6 * max(ref.abs().max().item(), 1e-30)   # sums of up to 9 K + K terms


def make_block(cin, cout, first, cuda, seed):
    from audio_deepfake_adversarial_attacks_amd.models.specrnet import Residual_block2D
    torch.manual_seed(seed)
    blk = Residual_block2D([cin, cout], first=first).to(cuda).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.empty(m.running_mean.shape).uniform_(-0.2, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.running_var.shape).uniform_(0.5, 1.5, generator=g))
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(-1.0, 1.5, generator=g))   # some negative scales
                m.bias.copy_(torch.empty(m.bias.shape).uniform_(-0.3, 0.3, generator=g))
    for p in blk.parameters():
        p.requires_grad_(False)
    return blk


@pytest.mark.parametrize("cin,cout,first,hw", [(2, 20, True, (16, 24)), (20, 64, False, (20, 101)), (64, 64, False, (5, 25)),
                                               (2, 20, True, (80, 404)), (1, 20, True, (9, 13)), (4, 20, True, (8, 12))])
def test_residual_block_matches_plain_modules(D, cuda, monkeypatch, parity_record, cin, cout, first, hw):
    """Residual_block2D with frozen parameters: the matrix-core block (ADVSTEP_SPECRNET_CONV=1) against the plain torch modules
    in float64 — output and input gradient.  A pooling winner at a near tie may go the other way, which moves single
    gradient entries; the bound on the gradient is therefore on its relative L2 error."""
    blk = make_block(cin, cout, first, cuda, 11)
    x = rnd((2, cin) + hw, 7, cuda)
    gy = rnd((2, cout, hw[0] // 2, hw[1] // 2), 8, cuda)

    def run(module, inp, go):
        a = inp.clone().requires_grad_(True)
        y = module(a)
        (gr,) = torch.autograd.grad(y, a, go)
        return y.detach(), gr

    monkeypatch.setenv("ADVSTEP_SPECRNET_CONV", "1")
    monkeypatch.setenv("ADVSTEP_SPECRNET_ELEM", "1")
    y1, g1 = run(blk, x, gy)
    assert getattr(blk, "_advstep_plan", None) is not None, "the matrix-core block did not run"
    import copy
    ref = copy.deepcopy(blk).double()
    monkeypatch.setenv("ADVSTEP_SPECRNET_ELEM", "0")
    y0, g0 = run(ref, x.double(), gy.double())
    fig = {"out_max_abs_over_max": ((y1.double() - y0).abs().max() / y0.abs().max()).item(),
           "grad_rel_l2": ((g1.double() - g0).norm() / g0.norm()).item()}
    parity_record[f"specrnet_block_{cin}_{cout}_{hw[0]}x{hw[1]}_matrix_cores_vs_float64_modules"] = fig
    assert fig["out_max_abs_over_max"] <= 5e-5, fig
    assert fig["grad_rel_l2"] <= 2e-3, fig


@pytest.mark.parametrize("shape", [(2, 20, 2, 20, 16, 24), (2, 20, 1, 20, 7, 9), (1, 20, 2, 20, 80, 404), (2, 8, 2, 5, 6, 10), (1, 64, 1, 33, 5, 25),
                                   (2, 20, 2, 20, 1, 6)])
def test_few_channel_downsample_in_the_epilogue_matches_the_reference(D, cuda, shape):
    """advstep_resconv_pool2_forward_few_f32: the 1x1 convolution over 1-2 channels applied in the epilogue (vector ALUs) instead of
    as reduction channels — same pooled values to REL of the convolution's scale, same selections away from near ties, and the same
    values as the all-matrix form (advstep_resconv_pool2_forward_f32) to fp32 rounding."""
    N, K1, K2, R, H, W = shape
    x1, w3 = rnd((N, K1, H, W), 1, cuda), rnd((R, K1, 3, 3), 2, cuda, 0.2)
    x2, w1 = rnd((N, K2, H, W), 3, cuda), rnd((R, K2), 4, cuda, 0.3)
    bias = rnd((R,), 5, cuda)
    y, sel = D.resconv_pool2_few(x1, x2, D.resconv_prepare(w3), w1.contiguous(), R, bias)
    Ho, Wo = H // 2, W // 2
    assert y.shape == (N, R, Ho, Wo)
    if Ho * Wo == 0:
        return
    full = reference(x1, x2, w3, w1, bias, 1.0)
    ref, ref_idx = F.max_pool2d(full, 2, return_indices=True)
    tol = REL * full.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= tol
    win = full[:, :, :2 * Ho, :2 * Wo].reshape(N, R, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, R, Ho, Wo, 4)
    top = win.topk(2, dim=-1).values
    clear = (top[..., 0] - top[..., 1]) > 2 * tol
    ref_code = (((ref_idx // W) % 2) * 2 + (ref_idx % W) % 2).to(torch.uint8)
    got = sel[:N * R * Ho * Wo].view(N, R, Ho, Wo)
    assert torch.equal(got[clear], ref_code[clear]) and (got < 4).all()
    y_all, _ = D.resconv_pool2(x1, x2, D.resconv_prepare(w3, w1), R, bias)
    assert (y - y_all).abs().max().item() <= tol
    with pytest.raises(ValueError):
        D.resconv_pool2_few(x1, rnd((N, 3, H, W), 3, cuda), D.resconv_prepare(w3), rnd((R, 3), 4, cuda), R, bias)


def sign_bytes(y):
    """(N, C, ceil(H/2), ceil(W/2)) bytes, bit 2 i + j = y > 0 at position (i, j) of the 2x2 tile (0 outside the plane)."""
    N, C, H, W = y.shape
    pos = F.pad((y > 0).to(torch.uint8), (0, W % 2, 0, H % 2))
    t = pos.view(N, C, (H + 1) // 2, 2, (W + 1) // 2, 2)
    return t[:, :, :, 0, :, 0] | (t[:, :, :, 0, :, 1] << 1) | (t[:, :, :, 1, :, 0] << 2) | (t[:, :, :, 1, :, 1] << 3)


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_sign_bytes_are_the_signs_of_the_output_it_wrote(D, cuda, shape):
    """with_act=True: the same y as without, and one byte per 2x2 tile whose bits are exactly y > 0 (odd sizes: 0 outside)."""
    N, K1, K2, R, H, W = shape
    x1, w3 = rnd((N, K1, H, W), 1, cuda), rnd((R, K1, 3, 3), 2, cuda, 0.2)
    x2, w1 = (rnd((N, K2, H, W), 3, cuda), rnd((R, K2), 4, cuda, 0.3)) if K2 else (None, None)
    shift, U = rnd((R,), 5, cuda), None
    U = D.resconv_prepare(w3, w1)
    y0 = D.resconv(x1, x2, U, R, shift, 0.3)
    y, act = D.resconv(x1, x2, U, R, shift, 0.3, with_act=True)
    assert torch.equal(y, y0) and act.dtype == torch.uint8 and act.shape == (N, R, (H + 1) // 2, (W + 1) // 2)
    assert torch.equal(act, sign_bytes(y))


@pytest.mark.parametrize("shape", [(2, 2, 20, 16, 24), (2, 1, 20, 7, 9), (1, 2, 5, 1, 1), (2, 2, 20, 80, 404), (3, 1, 3, 2, 3), (1, 2, 7, 5, 4)])
def test_few_input_channel_forward_sign_bytes(D, cuda, shape):
    N, Cin, Cout, H, W = shape
    x, w, shift = rnd((N, Cin, H, W), 1, cuda), rnd((Cout, Cin, 3, 3), 2, cuda, 0.3), rnd((Cout,), 3, cuda)
    y0 = D.conv3x3_fewin(x, w, shift, 0.3)
    y, act = D.conv3x3_fewin(x, w, shift, 0.3, with_act=True)
    assert torch.equal(y, y0) and torch.equal(act, sign_bytes(y))


@pytest.mark.parametrize("shape", [(2, 20, 20, 16, 24), (3, 64, 64, 20, 101), (2, 64, 64, 5, 25), (1, 20, 20, 7, 9), (2, 12, 40, 2, 2),
                                   (1, 8, 3, 1, 6), (2, 4, 2, 6, 8), (2, 128, 48, 9, 11)])
def test_pooled_gradient_from_sign_bytes_equals_the_one_from_the_activation(D, cuda, shape):
    """advstep_resconv_pooled_grad_act_f32 == advstep_resconv_pooled_grad_f32 given h, bit for bit (zeros and negative zeros of h
    take the slope in both)."""
    N, K, R, H, W = shape
    full = rnd((N, K, H, W), 1, cuda)
    U = D.resconv_prepare(rnd((K, R, 3, 3), 2, cuda, 0.2), transpose=True)
    h = rnd((N, R, H, W), 3, cuda)
    h.view(-1)[::7] = 0.0
    h.view(-1)[3::11] = -0.0
    act = sign_bytes(h).contiguous()
    if H // 2 == 0 or W // 2 == 0:
        gy = torch.zeros((N, K, H // 2, W // 2), device=cuda)
        assert not D.resconv_pooled_grad(gy, torch.zeros(1, dtype=torch.uint8, device=cuda), U, R, H, W, None, 0.3, act=act).any()
        return
    _, sel = D._add_maxpool2_raw(full, None, None)
    gy = rnd((N, K, H // 2, W // 2), 4, cuda)
    assert torch.equal(D.resconv_pooled_grad(gy, sel, U, R, H, W, None, 0.3, act=act), D.resconv_pooled_grad(gy, sel, U, R, H, W, h, 0.3))
    with pytest.raises(ValueError):
        D.resconv_pooled_grad(gy, sel, U, R, H, W, h, 0.3, act=act)
    with pytest.raises(ValueError):
        D.resconv_pooled_grad(gy, sel, U, R, H, W, None, 0.3, act=act.view(-1))


def test_first_block_with_the_downsample_in_the_epilogue_matches_the_all_matrix_block(D, cuda, monkeypatch):
    """ADVSTEP_RESBLOCK_FEW=1 (default) vs 0 on a 2-channel first block: the two forms differ by fp32 summation order only."""
    monkeypatch.setenv("ADVSTEP_SPECRNET_CONV", "1")
    blk = make_block(2, 20, True, cuda, 11)
    x = rnd((2, 2, 16, 24), 7, cuda)
    gy = rnd((2, 20, 8, 12), 8, cuda)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ADVSTEP_RESBLOCK_FEW", mode)
        a = x.clone().requires_grad_(True)
        y = blk(a)
        assert blk._advstep_plan.few == (mode == "1")
        (g,) = torch.autograd.grad(y, a, gy)
        out[mode] = (y.detach(), g)
    assert (out["1"][0] - out["0"][0]).abs().max().item() <= 2e-6 * out["0"][0].abs().max().item()
    assert (out["1"][1] - out["0"][1]).norm().item() <= 1e-4 * out["0"][1].norm().item()       # a near-tie winner may move


@pytest.mark.parametrize("cin,cout,first,hw", [(2, 20, True, (16, 24)), (20, 64, False, (20, 101)), (64, 64, False, (5, 25)),
                                               (1, 20, True, (9, 13))])
def test_residual_block_saving_sign_bytes_equals_the_one_saving_the_activation(D, cuda, monkeypatch, cin, cout, first, hw):
    blk = make_block(cin, cout, first, cuda, 11)
    x = rnd((2, cin) + hw, 7, cuda)
    gy = rnd((2, cout, hw[0] // 2, hw[1] // 2), 8, cuda)
    monkeypatch.setenv("ADVSTEP_SPECRNET_CONV", "1")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ADVSTEP_RESBLOCK_ACT", mode)
        a = x.clone().requires_grad_(True)
        y = blk(a)
        saved, stack = [], [y.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None:
                continue
            saved += list(getattr(fn, "saved_tensors", ()))
            stack += [nf for nf, _ in fn.next_functions]
        (g,) = torch.autograd.grad(y, a, gy)
        out[mode] = (y.detach(), g, sorted(t.dtype == torch.uint8 and t.dim() == 4 for t in saved))
    assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1])
    assert any(out["1"][2]) and not any(out["0"][2])        # the byte tensor replaced the activation among the saved tensors


def test_plan_is_rebuilt_when_a_parameter_changes(D, cuda, monkeypatch):
    monkeypatch.setenv("ADVSTEP_SPECRNET_CONV", "1")
    blk = make_block(20, 64, False, cuda, 5)
    x = rnd((1, 20, 8, 10), 1, cuda)
    y0 = blk(x)
    plan0 = blk._advstep_plan
    assert blk(x) is not None and blk._advstep_plan is plan0
    with torch.no_grad():
        blk.conv2.weight.mul_(2.0)
    y1 = blk(x)
    assert blk._advstep_plan is not plan0
    assert not torch.allclose(y0, y1)


def test_unsupported_shapes_are_refused(D, cuda):
    assert not D.resconv_supported(0, 0, 4) and not D.resconv_supported(6, 2, 4) and not D.resconv_supported(200, 60, 4)
    assert not D.resconv_supported(4, 0, 0) and not D.resconv_supported(4, 0, 257)
    with pytest.raises(ValueError):
        D.resconv_prepare(torch.zeros(4, 300, 3, 3, device=cuda))
