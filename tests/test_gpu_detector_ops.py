"""`-m gpu`: the fused elementwise / pooling kernels of include/advstep_detector.h (SpecRNet, RawNet3) against the ATen op
chains they replace — same arithmetic in the same order, so forward values, selections and input gradients are compared
bit for bit where the order is the same, and within float rounding where a reduction order differs (the gate gradient)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D(cuda):
    from audio_deepfake_adversarial_attacks_amd import detector_ops
    return detector_ops


def rnd(shape, seed, cuda, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(cuda)


@pytest.mark.parametrize("shape", [(3, 20, 16, 24), (2, 5, 7, 9), (1, 64, 1, 3), (4, 128, 1001), (2, 3, 6435)])
def test_affine_lrelu_and_relu_affine_match_aten(D, cuda, shape):
    C = shape[1]
    x = rnd(shape, 1, cuda)
    x.view(-1)[::97] = float("nan")                                  # NaN inputs follow ATen's rules too
    x.view(-1)[5::89] = 0.0
    x.requires_grad_(True)
    scale, shift, pre = rnd((C,), 2, cuda).abs() + 0.5, rnd((C,), 3, cuda), rnd((C,), 4, cuda)
    gy = rnd(shape, 5, cuda)
    view = (1, C) + (1,) * (len(shape) - 2)
    # mode 0: leaky_relu(x * s + t, 0.3)
    ref = F.leaky_relu(x * scale.view(view) + shift.view(view), 0.3)
    (g_ref,) = torch.autograd.grad(ref, x, gy)
    got = D.affine_lrelu(x, scale, shift, 0.3)
    (g_got,) = torch.autograd.grad(got, x, gy)
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    assert torch.allclose(g_got, g_ref, rtol=1e-6, atol=0, equal_nan=True)   # (gy * d) * s vs autograd's grouping: one rounding apart
    # mode 1: relu(x + pre) * s + t
    ref = torch.relu(x + pre.view(view)) * scale.view(view) + shift.view(view)
    (g_ref,) = torch.autograd.grad(ref, x, gy)
    got = D.relu_affine(x, scale, shift, pre)
    (g_got,) = torch.autograd.grad(got, x, gy)
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    assert torch.equal(torch.nan_to_num(g_got, nan=7.0), torch.nan_to_num(g_ref, nan=7.0))
    got = D.relu_affine(x, scale, shift, None)
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(torch.relu(x) * scale.view(view) + shift.view(view), nan=7.0))


@pytest.mark.parametrize("shape", [(3, 2, 80, 404), (2, 64, 1, 25), (2, 5, 7, 9)])
def test_affine_selu_matches_batchnorm_then_selu(D, cuda, shape):
    """selu(bn_eval(x)) in one pass: forward to 2e-6 of the scale (the BatchNorm arithmetic is folded into one multiply-add),
    input gradient to 1e-5 relative (ATen differentiates the in-place SELU from its result, the kernel from its input)."""
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        bn.running_mean.copy_(torch.empty(C).uniform_(-0.5, 0.5, generator=g))
        bn.running_var.copy_(torch.empty(C).uniform_(0.5, 2.0, generator=g))
        bn.weight.copy_(torch.empty(C).uniform_(-1.0, 1.5, generator=g))
        bn.bias.copy_(torch.empty(C).uniform_(-0.5, 0.5, generator=g))
    x, gy = rnd(shape, 2, cuda, 2.0), rnd(shape, 3, cuda)
    a = x.clone().requires_grad_(True)
    y0 = torch.nn.SELU(inplace=True)(bn(a))
    (g0,) = torch.autograd.grad(y0, a, gy)
    scale, shift = D.bn_eval_affine(bn)
    b = x.clone().requires_grad_(True)
    y1 = D.affine_selu(b, scale, shift)
    (g1,) = torch.autograd.grad(y1, b, gy)
    assert (y0 - y1).abs().max().item() <= 2e-6 * y0.abs().max().item()
    assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()


def test_bn_eval_affine_equals_batch_norm(D, cuda):
    bn = torch.nn.BatchNorm2d(6).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-1, 1), bn.running_var.uniform_(0.5, 2), bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-1, 1)
    x = rnd((3, 6, 5, 7), 7, cuda)
    scale, shift = D.bn_eval_affine(bn)
    want = bn(x)
    got = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert D.bn_eval_affine(bn)[0] is scale                         # cached
    with torch.no_grad():
        bn.weight.mul_(2.0)
    assert D.bn_eval_affine(bn)[0] is not scale                      # refreshed when a parameter changes


@pytest.mark.parametrize("shape", [(2, 20, 80, 404), (3, 5, 7, 9), (1, 2, 2, 2), (2, 64, 20, 101), (2, 3, 5, 25), (1, 4, 3, 2)])
@pytest.mark.parametrize("kind", ["ab_bias", "a_only", "ties", "nans"])
def test_add_maxpool2_matches_aten(D, cuda, shape, kind):
    N, C, H, W = shape
    a = rnd(shape, 11, cuda)
    b = rnd(shape, 12, cuda) if kind != "a_only" else None
    bias = rnd((C,), 13, cuda) if kind == "ab_bias" else None
    if kind == "ties":
        a = torch.round(a)                                           # many equal values inside a window
        b = torch.round(b)
    if kind == "nans":
        a = a.clone()
        a.view(-1)[::7] = float("nan")
    a.requires_grad_(True)
    s = a if b is None else a + b
    if bias is not None:
        s = s + bias.view(1, -1, 1, 1)
    ref = F.max_pool2d(s, 2)
    got = D.add_maxpool2(a, b, bias)
    assert got.shape == ref.shape == (N, C, H // 2, W // 2)
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
    if ref.numel() == 0:
        return
    gy = rnd(tuple(ref.shape), 14, cuda)
    (g_ref,) = torch.autograd.grad(ref, a, gy)
    (g_got,) = torch.autograd.grad(got, a, gy)
    assert torch.equal(g_got, g_ref)                                  # same winner at ties and NaNs, zeros in odd tails


@pytest.mark.parametrize("shape,k", [((2, 16, 6435), 5), ((3, 7, 1287), 3), ((1, 2, 17), 5), ((2, 3, 4), 5), ((2, 4, 30), 2)])
@pytest.mark.parametrize("kind", ["ab", "a_only", "ties", "nans"])
def test_add_maxpool1d_matches_aten(D, cuda, shape, k, kind):
    a = rnd(shape, 31, cuda)
    b = rnd(shape, 32, cuda) if kind != "a_only" else None
    if kind == "ties":
        a, b = torch.round(a), torch.round(b)
    if kind == "nans":
        a = a.clone()
        a.view(-1)[::11] = float("nan")
    a.requires_grad_(True)
    s = a if b is None else a + b
    got = D.add_maxpool1d(a, b, k)
    if shape[2] < k:
        assert got.shape == (shape[0], shape[1], 0)
        return
    ref = F.max_pool1d(s, k)
    assert got.shape == ref.shape
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
    gy = rnd(tuple(ref.shape), 33, cuda)
    (g_ref,) = torch.autograd.grad(ref, a, gy)
    (g_got,) = torch.autograd.grad(got, a, gy)
    assert torch.equal(g_got, g_ref)


@pytest.mark.parametrize("shape", [(2, 20, 40, 202), (3, 5, 7, 9), (2, 64, 10, 50), (1, 3, 2, 12)])
def test_gate_maxpool2_matches_aten(D, cuda, shape):
    N, C, H, W = shape
    x = rnd(shape, 21, cuda).requires_grad_(True)
    gate = torch.sigmoid(rnd((N, C), 22, cuda)).requires_grad_(True)
    g4 = gate.view(N, C, 1, 1)
    ref = F.max_pool2d(x * g4 + g4, 2)
    got = D.gate_maxpool2(x, gate)
    assert torch.equal(got, ref)
    gy = rnd(tuple(ref.shape), 23, cuda)
    gx_ref, gg_ref = torch.autograd.grad(ref, (x, gate), gy)
    gx, gg = torch.autograd.grad(got, (x, gate), gy)
    assert torch.equal(gx, gx_ref)
    assert torch.allclose(gg, gg_ref, rtol=2e-5, atol=2e-5 * gg_ref.abs().max().item())
    # fixed summation order: bit-reproducible
    gx2, gg2 = torch.autograd.grad(D.gate_maxpool2(x, gate), (x, gate), gy)
    assert torch.equal(gg, gg2) and torch.equal(gx, gx2)


def attack_mode_frozen(model):
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return model


@pytest.mark.parametrize("name,cfg,switch", [("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, "ADVSTEP_SPECRNET_ELEM"),
                                             ("rawnet3", {}, "ADVSTEP_RAWNET3_ELEM")])
def test_detectors_with_fused_elementwise_passes_agree_with_plain_modules(cuda, monkeypatch, parity_record, name, cfg, switch):
    """Whole detector, attack mode, frozen parameters: fused elementwise / pooling passes vs the plain torch modules.  Bias
    adds move behind the convolution (one rounding apart), so logits agree to float tolerance; a max-pool winner at a
    near tie may go the other way (SpecRNet), so the gradient bound is relative."""
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(3)
    model = attack_mode_frozen(get_model(name, dict(cfg), str(cuda)).to(cuda))
    with torch.no_grad():                                             # non-trivial BatchNorm statistics
        g = torch.Generator().manual_seed(4)
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.copy_(torch.empty(m.running_mean.shape).uniform_(-0.2, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.running_var.shape).uniform_(0.5, 1.5, generator=g))
    x = (torch.randn(2, 64_600, generator=torch.Generator().manual_seed(5)) * 0.05).to(cuda)

    def run(on):
        monkeypatch.setenv(switch, "1" if on else "0")
        a = x.clone().requires_grad_(True)
        z = model(a)
        (gr,) = torch.autograd.grad(z.sum(), a)
        return z.detach(), gr

    z0, g0 = run(False)
    z1, g1 = run(True)
    fig = {"logit_max_abs": (z0 - z1).abs().max().item(), "logit_scale": z0.abs().max().item(),
           "grad_rel_l2": ((g0 - g1).norm() / g0.norm()).item()}
    parity_record[f"{name}_fused_elementwise_vs_plain_modules"] = fig
    # measured (profiles/r02_parity.json): SpecRNet logits 1.5e-8, gradient relative L2 3.2e-7; RawNet3 8.9e-8 / 7.9e-6
    bound = {"specrnet": (1.5e-7, 3.2e-6), "rawnet3": (9e-7, 8e-5)}[name]
    assert fig["logit_max_abs"] <= bound[0] and fig["grad_rel_l2"] <= bound[1], fig
    for p in model.parameters():
        p.requires_grad_(True)


@pytest.mark.parametrize("N,C,P,Ct", [(2, 128, 6435, 1024), (3, 4, 429, 12), (1, 8, 17, 8), (2, 4, 1287, 12), (2, 3, 5, 7)])
def test_res2net_link_kernels_match_the_torch_chain(D, cuda, N, C, P, Ct):
    """advstep_res2net_link_{forward,backward}_f32 on channel slices of (N, Ct, P) tensors against
    `y = bn(relu(h + bias)); z = y + next_group` and `relu_affine`'s backward of (g1 + g2): same arithmetic, bit for bit;
    nothing outside the slices is written.  (Ct = 7, P = 5: batch strides not a multiple of 4 -> the element-wise path.)"""
    h = rnd((N, C, P), 1, cuda)
    scale, shift, pre = rnd((C,), 2, cuda), rnd((C,), 3, cuda), rnd((C,), 4, cuda)
    h[0, 0, 0] = -pre[0]                                        # relu at exactly 0
    lo = C if Ct >= 3 * C else 0                                # the slice under test
    groups = rnd((N, Ct, P), 5, cuda)
    other = groups[:, Ct - C:]
    cat = torch.full((N, Ct, P), 7.0, device=cuda)
    z = torch.empty((N, C, P), device=cuda)
    D.res2net_link_forward(h, scale, shift, pre, cat[:, lo:lo + C], other, z)
    y_ref = torch.relu(h + pre.view(1, -1, 1)) * scale.view(1, -1, 1) + shift.view(1, -1, 1)
    assert torch.equal(cat[:, lo:lo + C], y_ref) and torch.equal(z, y_ref + other)
    untouched = torch.ones(Ct, dtype=torch.bool, device=cuda)
    untouched[lo:lo + C] = False
    assert (cat[:, untouched] == 7.0).all()
    cat2 = torch.full((N, Ct, P), 7.0, device=cuda)
    D.res2net_link_forward(h, scale, shift, None, cat2[:, lo:lo + C], None, None)                 # last branch, no bias
    assert torch.equal(cat2[:, lo:lo + C], torch.relu(h) * scale.view(1, -1, 1) + shift.view(1, -1, 1))

    g_cat, g_in = rnd((N, Ct, P), 6, cuda), rnd((N, Ct, P), 7, cuda)
    g1, g2 = g_cat[:, lo:lo + C], g_in[:, Ct - C:]
    gx = D.res2net_link_backward(g1, g2, h, scale, shift, pre)
    ref = torch.where((h + pre.view(1, -1, 1)) <= 0, torch.zeros_like(h), (g1 + g2) * scale.view(1, -1, 1))
    assert torch.equal(gx, ref)
    gx1 = D.res2net_link_backward(g1, None, h, scale, shift, pre)
    assert torch.equal(gx1, torch.where((h + pre.view(1, -1, 1)) <= 0, torch.zeros_like(h), (g1 + 0.0) * scale.view(1, -1, 1)))
    with pytest.raises(ValueError):
        D.res2net_link_backward(g_cat.transpose(1, 2)[:, :C], None, h, scale, shift, pre)


def test_rawnet3_res2net_chain_equals_the_separate_ops(cuda, monkeypatch, parity_record):
    """Whole RawNet3, frozen: the branches of every Bottle2neck through `_Res2NetChain` (ADVSTEP_RAWNET3_CHAIN=1) against the
    same kernels launched as separate torch ops with autograd in between.  Same GEMMs and the same elementwise arithmetic:
    logits and the waveform gradient agree to float rounding (the order of the two-term gradient sums is the same)."""
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(3)
    model = attack_mode_frozen(get_model("rawnet3", {}, str(cuda)).to(cuda))
    x = (torch.randn(2, 64_600, generator=torch.Generator().manual_seed(5)) * 0.05).to(cuda)

    def run(on):
        monkeypatch.setenv("ADVSTEP_RAWNET3_CHAIN", "1" if on else "0")
        a = x.clone().requires_grad_(True)
        z = model(a)
        (gr,) = torch.autograd.grad(z.sum(), a)
        return z.detach(), gr

    z0, g0 = run(False)
    z1, g1 = run(True)
    assert model.layer1._chain_val["nums"] == 7
    fig = {"logit_max_abs": (z0 - z1).abs().max().item(), "grad_rel_l2": ((g0 - g1).norm() / g0.norm()).item()}
    parity_record["rawnet3_res2net_chain_vs_separate_ops"] = fig
    assert fig["logit_max_abs"] <= 1e-6 * max(z0.abs().max().item(), 1.0), fig
    assert fig["grad_rel_l2"] <= 1e-5, fig
    for p in model.parameters():
        p.requires_grad_(True)


@pytest.mark.parametrize("shape,k", [((2, 16, 6435), 5), ((3, 7, 1287), 3), ((1, 2, 17), 5), ((2, 3, 4), 5), ((2, 4, 30), 2), ((1, 3, 2700), 3)])
def test_tail_pool1d_equals_activation_then_add_maxpool1d(D, cuda, shape, k):
    """advstep_tail_pool1d_* = relu_affine followed by add_maxpool1d, in one pass each way: bit for bit, values, winners and both
    gradients (several tiles per row, a dropped tail, rows shorter than a window)."""
    N, C, L = shape
    h, res = rnd(shape, 1, cuda), rnd(shape, 2, cuda)
    scale, shift, pre = rnd((C,), 3, cuda), rnd((C,), 4, cuda), rnd((C,), 5, cuda)
    h[0, 0, :min(L, 3)] = -pre[0]                                   # relu at exactly 0; ties inside a window
    res[0, 0, :min(L, 3)] = 0.25

    def run(fused, with_pre):
        a, b = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
        p = pre if with_pre else None
        y = D.tail_pool1d(a, b, scale, shift, p, k) if fused else D.add_maxpool1d(D.relu_affine(a, scale, shift, p), b, k)
        if y.numel() == 0:
            return y.detach(), torch.zeros_like(a), torch.zeros_like(b)
        gy = rnd(tuple(y.shape), 6, cuda)
        ga, gb = torch.autograd.grad(y, (a, b), gy)
        return y.detach(), ga, gb

    for with_pre in (True, False):
        y0, ga0, gb0 = run(False, with_pre)
        y1, ga1, gb1 = run(True, with_pre)
        assert y1.shape == (N, C, L // k)
        assert torch.equal(y0, y1) and torch.equal(ga0, ga1) and torch.equal(gb0, gb1)


@pytest.mark.parametrize("shape", [(2, 16, 6435), (3, 5, 429), (1, 2, 1), (2, 3, 8192), (1, 4, 257)])
def test_log_meannorm_matches_the_aten_chain(D, cuda, shape):
    """x = log(|y| + 1e-6) - mean_t(.) and its gradient against torch's abs / add / log / mean / sub with autograd (the row sum
    runs in another order: values to 2e-6 of the row's log range, gradients to 1e-5 relative).  Zeros and a NaN are in the input:
    sign(0) = 0 and NaN poisons exactly its own row."""
    y = rnd(shape, 1, cuda)
    y[0, 0, 0] = 0.0
    g = rnd(shape, 2, cuda)

    def chain(t):
        x = torch.log(torch.abs(t) + 1e-6)
        return x - torch.mean(x, dim=-1, keepdim=True)

    a = y.clone().requires_grad_(True)
    x0 = chain(a)
    (g0,) = torch.autograd.grad(x0, a, g)
    b = y.clone().requires_grad_(True)
    x1 = D.log_meannorm(b)
    (g1,) = torch.autograd.grad(x1, b, g)
    assert (x0 - x1).abs().max().item() <= 2e-6 * max(x0.abs().max().item(), 1.0)
    assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()
    if shape[-1] > 1:
        z = y.clone()
        z[-1, -1, -1] = float("nan")
        xn = D.log_meannorm(z)
        assert torch.isnan(xn[-1, -1]).all() and torch.isfinite(xn.reshape(-1, shape[-1])[:-1]).all()
    assert not D.log_meannorm_supported(8193) and D.log_meannorm_supported(6435)


@pytest.mark.parametrize("shape", [(2, 16, 1287), (3, 5, 429), (1, 2, 1), (2, 1024, 33)])
def test_afms_matches_the_torch_module(D, cuda, shape):
    """detector_ops.afms against RawNet3's AFMS module (mean -> fc -> sigmoid -> (x + alpha) * y) with autograd: values and input
    gradient to float rounding (row sums in another order)."""
    from audio_deepfake_adversarial_attacks_amd.models.rawnet3 import AFMS
    N, C, L = shape
    torch.manual_seed(7)
    mod = AFMS(C).to(cuda)
    with torch.no_grad():
        mod.alpha.copy_(torch.randn(C, 1))
    x, g = rnd(shape, 1, cuda), rnd(shape, 2, cuda)
    a = x.clone().requires_grad_(True)
    y0 = mod(a)                                                   # parameters require grad -> the torch ops
    (g0,) = torch.autograd.grad(y0, a, g)
    b = x.clone().requires_grad_(True)
    y1 = D.afms(b, mod.alpha.detach(), mod.fc.weight.detach(), mod.fc.bias.detach())
    (g1,) = torch.autograd.grad(y1, b, g)
    assert (y0 - y1).abs().max().item() <= 2e-6 * max(y0.abs().max().item(), 1.0)
    assert (g0 - g1).abs().max().item() <= 2e-5 * max(g0.abs().max().item(), 1e-30)
    for p in mod.parameters():
        p.requires_grad_(False)
    c = x.clone().requires_grad_(True)
    assert torch.equal(mod(c), y1)                                # frozen module takes the same path


@pytest.mark.parametrize("shape", [(2, 1536, 429), (3, 5, 7), (1, 2, 1), (2, 3, 130)])
def test_weighted_stats_match_the_torch_expressions(D, cuda, shape):
    """(sum x w, sum x^2 w) and both gradients against torch with autograd; row sums run in another order (2e-6 of the scale),
    the gradients are elementwise (bit-identical products, one rounding apart at most)."""
    x, w = rnd(shape, 1, cuda), torch.softmax(rnd(shape, 2, cuda), dim=2)
    gmu, gm2 = rnd(shape[:2], 3, cuda), rnd(shape[:2], 4, cuda)
    a, b = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    mu0, m20 = torch.sum(a * b, dim=2), torch.sum((a ** 2) * b, dim=2)
    ga0, gb0 = torch.autograd.grad((mu0, m20), (a, b), (gmu, gm2))
    c, d = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    mu1, m21 = D.weighted_stats(c, d)
    ga1, gb1 = torch.autograd.grad((mu1, m21), (c, d), (gmu, gm2))
    for r, o in ((mu0, mu1), (m20, m21)):
        assert (r - o).abs().max().item() <= 2e-6 * max(r.abs().max().item(), 1.0)
    for r, o in ((ga0, ga1), (gb0, gb1)):
        assert (r - o).abs().max().item() <= 2e-6 * max(r.abs().max().item(), 1e-30)


@pytest.mark.parametrize("shape", [(2, 20, 40, 202), (3, 5, 7, 9), (2, 64, 10, 50), (1, 3, 2, 12), (1, 4, 5, 25)])
@pytest.mark.parametrize("winners", ["1", "0"])
def test_attend_pool_matches_the_torch_chain(D, cuda, monkeypatch, shape, winners):
    """detector_ops.attend_pool (mean -> fc -> sigmoid -> x * g + g -> MaxPool2d(2), frozen fc) against the torch ops with
    autograd: pooled values to 2e-6 of the scale, input gradient to 2e-5 relative (the mean and the gate's gradient are row
    sums in another order).  winners = 1 (default): the gate's gradient from the winners the forward wrote (pooled size), x not
    kept; 0: gathered out of x in backward."""
    monkeypatch.setenv("ADVSTEP_ATTEND_XW", winners)
    N, C, H, W = shape
    torch.manual_seed(9)
    fc = torch.nn.Linear(C, C).to(cuda)
    x, gy = rnd(shape, 1, cuda), rnd((N, C, H // 2, W // 2), 2, cuda)
    a = x.clone().requires_grad_(True)
    g0 = torch.sigmoid(fc(a.mean(dim=(2, 3)))).view(N, C, 1, 1)
    y0 = F.max_pool2d(a * g0 + g0, 2)
    (ga0,) = torch.autograd.grad(y0, a, gy)
    b = x.clone().requires_grad_(True)
    y1 = D.attend_pool(b, fc.weight.detach(), fc.bias.detach())
    saved = y1.grad_fn.saved_tensors
    assert any(t.shape == x.shape for t in saved) == (winners == "0")          # x itself is kept only on the gather path
    (ga1,) = torch.autograd.grad(y1, b, gy)
    assert (y0 - y1).abs().max().item() <= 2e-6 * max(y0.abs().max().item(), 1.0)
    assert (ga0 - ga1).norm().item() <= 2e-5 * max(ga0.norm().item(), 1e-30)


def test_gate_pool_forward_writes_the_winners(D, cuda):
    """advstep_gate_maxpool2_forward_xw_f32: xw = x at the position the selection byte names, bit for bit; y and sel as without."""
    from audio_deepfake_adversarial_attacks_amd import _lib
    from audio_deepfake_adversarial_attacks_amd.hip_ops import _stream
    lib = _lib.load()
    for N, C, H, W in [(2, 20, 40, 202), (3, 5, 7, 9), (1, 3, 2, 12)]:
        x, gate = rnd((N, C, H, W), 1, cuda), torch.rand(N, C, device=cuda) + 0.1
        Ho, Wo = H // 2, W // 2
        y0, y1, xw = (torch.empty((N, C, Ho, Wo), device=cuda) for _ in range(3))
        s0, s1 = (torch.empty(N * C * Ho * Wo, dtype=torch.uint8, device=cuda) for _ in range(2))
        _lib.check(lib.advstep_gate_maxpool2_forward_f32(x.data_ptr(), gate.data_ptr(), y0.data_ptr(), s0.data_ptr(), N, C, H, W,
                                                         _stream(cuda)), "forward")
        _lib.check(lib.advstep_gate_maxpool2_forward_xw_f32(x.data_ptr(), gate.data_ptr(), y1.data_ptr(), s1.data_ptr(), xw.data_ptr(),
                                                            N, C, H, W, _stream(cuda)), "forward_xw")
        assert torch.equal(y0, y1) and torch.equal(s0, s1)
        code = s1.view(N, C, Ho, Wo).long()
        win = x[:, :, :2 * Ho, :2 * Wo].reshape(N, C, Ho, 2, Wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, Ho, Wo, 4)
        assert torch.equal(xw, win.gather(-1, code.unsqueeze(-1)).squeeze(-1))
        gy, gg = rnd((N, C, Ho, Wo), 2, cuda), torch.empty(N * C, device=cuda)
        _lib.check(lib.advstep_gate_maxpool2_backward_gate_pooled_f32(gy.data_ptr(), xw.data_ptr(), gg.data_ptr(), N, C, H, W,
                                                                      _stream(cuda)), "gate_pooled")
        ref = (gy.double() * (xw.double() + 1.0)).sum(dim=(2, 3)).view(-1)
        assert (gg.double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), 1.0)


def test_round2_detector_ops_accept_empty_batches(D, cuda):
    """N = 0 (an empty last shard) through every round-2 detector operator: correctly shaped empty outputs, no launch."""
    e3 = torch.empty((0, 8, 33), device=cuda)
    c = torch.ones(8, device=cuda)
    assert D.tail_pool1d(e3, e3, c, c, c, 3).shape == (0, 8, 11)
    assert D.log_meannorm(e3).shape == (0, 8, 33)
    mu, m2 = D.weighted_stats(e3, e3)
    assert mu.shape == (0, 8) and m2.shape == (0, 8)
    assert D.afms(e3, c, torch.eye(8, device=cuda), c).shape == (0, 8, 33)
    assert D.affine_selu(e3, c, c).shape == (0, 8, 33)
    e4 = torch.empty((0, 20, 8, 12), device=cuda)
    assert D.attend_pool(e4, torch.eye(20, device=cuda), torch.zeros(20, device=cuda)).shape == (0, 20, 4, 6)
    U = D.resconv_prepare(torch.zeros(20, 20, 3, 3, device=cuda))
    assert D.resconv(e4, None, U, 20).shape == (0, 20, 8, 12)
    y, _ = D.resconv_pool2(e4, None, U, 20)
    assert y.shape == (0, 20, 4, 6)
    assert D.conv3x3_fewin(torch.empty((0, 2, 8, 12), device=cuda), torch.zeros(20, 2, 3, 3, device=cuda), None, 1.0).shape == (0, 20, 8, 12)
    assert D.conv3x3_fewout_grad(e4, torch.zeros(20, 2, 3, 3, device=cuda)).shape == (0, 2, 8, 12)
