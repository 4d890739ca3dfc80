"""`-m gpu`: every C-ABI entry point of libadvstep.so against (i) the reference's golden vectors and (ii) the
plain-C oracle on seeded inputs, through the product's ctypes binding (hip_ops -> libadvstep.so)."""
import os

import numpy as np
import pytest
import torch

from oracle import kernels as K
from tests.helpers import dev, rand01, randn

pytestmark = pytest.mark.gpu

T_FULL = 64_600
SHAPES = [(1, 1), (2, 3), (3, 255), (2, 4096), (3, 4099), (5, 8192), (4, 12_289), (2, T_FULL), (1, 70_001)]


@pytest.fixture(scope="module")
def ops(cuda):
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    return hip_ops


def host(t):
    return t.detach().cpu().numpy()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


# ---- golden vectors produced by the reference itself ----------------------------------------------------------------

def test_minmax_golden(ops, cuda, golden):
    g = golden("minmax")
    for tag in ("full", "ragged", "const"):
        x01, mn, mx = ops.to_minmax(dev(g[f"{tag}_x"], cuda))
        assert same(host(x01), g[f"{tag}_x01"]) and same(host(mn), g[f"{tag}_mn"]) and same(host(mx), g[f"{tag}_mx"]), tag
        if tag != "const":
            back = ops.revert_minmax(dev(g[f"{tag}_p"], cuda), dev(g[f"{tag}_mn"], cuda), dev(g[f"{tag}_mx"], cuda))
            assert same(host(back), g[f"{tag}_revert"]), tag
    assert np.isnan(g["const_x01"][1]).all()  # the reference's own NaN row (src/aa/utils.py:8-9)


def test_fgsm_golden(ops, cuda, golden):
    g = golden("fgsm")
    for tag in ("ragged", "small"):
        for e in ("e0005", "e00075", "e001"):
            p = f"{tag}_{e}_"
            out = ops.fgsm_step(dev(g[p + "x"], cuda), dev(g[p + "grad"], cuda), float(g[p + "eps"]))
            assert same(host(out), g[p + "adv"]), p  # bit-exact; north-star bound is 1e-5 max-abs


def test_pgd_linf_golden(ops, cuda, golden):
    g = golden("pgd_linf")
    for tag in ("ragged_rs", "small_nors", "full_rs"):
        eps, alpha, steps = float(g[tag + "_eps"]), float(g[tag + "_alpha"]), int(g[tag + "_steps"])
        x = dev(g[tag + "_x"], cuda)
        if tag + "_noise" in g:
            a0 = ops.pgd_linf_init(x, eps, noise=dev(g[tag + "_noise"], cuda))
            assert same(host(a0), g[tag + "_a0"]), tag
        for k in range(steps):
            out = ops.pgd_linf_step(dev(g[f"{tag}_a{k}"], cuda), dev(g[f"{tag}_g{k}"], cuda), x, alpha, eps)
            assert same(host(out), g[f"{tag}_a{k + 1}"]), (tag, k)


def test_pgd_l2_golden(ops, cuda, golden):
    g = golden("pgd_l2")
    for tag in ("ragged_rs", "small_nors"):
        eps, alpha, steps = float(g[tag + "_eps"]), float(g[tag + "_alpha"]), int(g[tag + "_steps"])
        eps_div = float(g[tag + "_eps_div"])
        x = dev(g[tag + "_x"], cuda)
        if tag + "_normal" in g:
            a0 = ops.pgd_l2_init(x, eps, draws=(dev(g[tag + "_normal"], cuda), dev(g[tag + "_r"], cuda)))
            np.testing.assert_allclose(host(a0), g[tag + "_a0"], atol=3e-7, rtol=0)
        for k in range(steps):
            out, gn, dn = ops.pgd_l2_step(dev(g[f"{tag}_a{k}"], cuda), dev(g[f"{tag}_g{k}"], cuda), x, alpha, eps, eps_div,
                                          return_norms=True)
            # stated tolerance: the reference's torch.norm and the kernel reduce in different f32 orders
            np.testing.assert_allclose(host(gn), g[f"{tag}_gnorm{k}"], rtol=2e-6)
            np.testing.assert_allclose(host(dn), g[f"{tag}_dnorm{k}"], rtol=2e-6)
            np.testing.assert_allclose(host(out), g[f"{tag}_a{k + 1}"], atol=3e-7, rtol=0)


def test_cw_golden(ops, cuda, golden):
    g = golden("cw")
    x = dev(g["x"], cuda)
    w0 = host(ops.cw_init_w(x))
    fin = np.isfinite(g["w0"])
    assert same(np.isposinf(w0), np.isposinf(g["w0"])) and same(np.isneginf(w0), np.isneginf(g["w0"]))
    np.testing.assert_allclose(w0[fin], g["w0"][fin], rtol=2e-6, atol=1e-6)
    m = torch.zeros_like(x)
    v = torch.zeros_like(x)
    for k in range(4):
        w = dev(g[f"s{k}_w"], cuda)
        adv, l2 = ops.cw_tanh_sqdist(w, x)
        np.testing.assert_allclose(host(adv), g[f"s{k}_adv"], atol=2e-7, rtol=0)
        np.testing.assert_allclose(host(l2), g[f"s{k}_l2"], rtol=1e-4, atol=1e-9)
        ops.cw_adam_step(w, m, v, x, dev(g[f"s{k}_grad_adv"], cuda), k + 1, lr=float(g["lr"]))
        np.testing.assert_allclose(host(m), g[f"s{k}_m_after"], rtol=2e-5, atol=3e-8)
        np.testing.assert_allclose(host(v), g[f"s{k}_v_after"], rtol=2e-4, atol=1e-12)
        resolved = np.abs(g[f"s{k}_grad_w"]) > 1e-4  # elsewhere Adam amplifies rounding noise to +-lr
        np.testing.assert_allclose(host(w)[resolved], g[f"s{k}_w_after"][resolved], atol=2e-6, rtol=0)
        m, v = dev(g[f"s{k}_m_after"], cuda), dev(g[f"s{k}_v_after"], cuda)


# ---- oracle on seeded inputs, ragged / unaligned / large shapes ----------------------------------------------------------

@pytest.mark.parametrize("B,T", SHAPES)
def test_minmax_and_revert_vs_oracle(ops, cuda, B, T):
    x = randn((B, T), 100 + T, 0.05)
    x01, mn, mx = ops.to_minmax(dev(x, cuda))
    w01, wmn, wmx = K.minmax_normalize(x)
    assert same(host(x01), w01) and same(host(mn).ravel(), wmn) and same(host(mx).ravel(), wmx)
    p = rand01((B, T), 200 + T)
    assert same(host(ops.revert_minmax(dev(p, cuda), mn, mx)), K.minmax_revert(p, wmn, wmx))


def test_minmax_nan_and_constant_rows(ops, cuda):
    x = randn((3, 5000), 5, 0.05)
    x[1, 1234] = np.nan
    x[2, :] = 0.125
    x01, mn, mx = ops.to_minmax(dev(x, cuda))
    w01, wmn, wmx = K.minmax_normalize(x)
    assert same(host(x01), w01) and same(host(mn).ravel(), wmn) and same(host(mx).ravel(), wmx)
    assert np.isnan(host(x01)[1]).all() and np.isnan(host(x01)[2]).all() and np.isfinite(host(x01)[0]).all()


@pytest.mark.parametrize("B,T", SHAPES)
def test_flat_steps_vs_oracle(ops, cuda, B, T):
    x, a, g = rand01((B, T), 1 + T), rand01((B, T), 2 + T), randn((B, T), 3 + T, 1e-3)
    g[0, 0] = 0.0
    g.flat[-1] = np.nan
    for eps in (0.0005, 0.001, 0.003):
        assert same(host(ops.fgsm_step(dev(x, cuda), dev(g, cuda), eps)), K.fgsm_step(x, g, eps))
        a_near = np.clip(x + randn((B, T), 4 + T, eps), 0, 1).astype(np.float32)
        for alpha in (2 / 255, eps / 4):
            got = ops.pgd_linf_step(dev(a_near, cuda), dev(g, cuda), dev(x, cuda), alpha, eps)
            assert same(host(got), K.pgd_linf_step(a_near, g, x, alpha, eps))
    nz = randn((B, T), 5 + T, 0.003)
    assert same(host(ops.pgd_linf_init(dev(x, cuda), 0.003, noise=dev(nz, cuda))), K.pgd_linf_init_noise(x, nz))
    for seed, off in ((0, 0), (123456789012345, 7), (2 ** 62 - 1, 2 ** 40)):
        got = host(ops.pgd_linf_init(dev(x, cuda), 0.003, seed=seed, offset=off))
        assert same(got, K.pgd_linf_init_philox(x, 0.003, seed, off))
    assert np.abs(got - x).max() <= 0.003 + 1e-7


def test_flat_steps_unaligned_pointers(ops, cuda):
    """Views that start 4 bytes into an allocation take the scalar path of every flat kernel."""
    n = 10_001
    base = [dev(np.concatenate([[0], a]).astype(np.float32), cuda) for a in
            (rand01(n, 1), randn(n, 2, 1e-3), rand01(n, 3))]
    a, g, x = (b[1:] for b in base)
    assert a.data_ptr() % 16 != 0
    got = ops.pgd_linf_step(a, g, x, 2 / 255, 0.003)
    assert same(host(got), K.pgd_linf_step(host(a), host(g), host(x), 2 / 255, 0.003))
    out = torch.empty(n + 1, device=cuda)[1:]
    ops.fgsm_step(x, g, 0.001, out=out)
    assert same(host(out), K.fgsm_step(host(x), host(g), 0.001))


def test_in_place_step(ops, cuda):
    a, g, x = rand01((4, 4099), 1), randn((4, 4099), 2), rand01((4, 4099), 3)
    da = dev(a, cuda)
    ops.pgd_linf_step(da, dev(g, cuda), dev(x, cuda), 2 / 255, 0.003, out=da)
    assert same(host(da), K.pgd_linf_step(a, g, x, 2 / 255, 0.003))
    da = dev(a, cuda)
    ops.pgd_l2_step(da, dev(g, cuda), dev(x, cuda), 0.2, 0.1, out=da)
    np.testing.assert_allclose(host(da), K.pgd_l2_step(a, g, x, 0.2, 0.1)[0], atol=3e-7, rtol=0)


@pytest.mark.parametrize("B,T", SHAPES)
def test_pgd_l2_vs_oracle(ops, cuda, B, T):
    x = rand01((B, T), 11 + T)
    a = np.clip(x + randn((B, T), 12 + T, 1e-3), 0, 1).astype(np.float32)
    g = randn((B, T), 13 + T, 1e-4)
    for alpha, eps in ((0.2, 0.1), (0.2, 1.0), (0.01, 0.2)):
        got, gn, dn = ops.pgd_l2_step(dev(a, cuda), dev(g, cuda), dev(x, cuda), alpha, eps, return_norms=True)
        want, wgn, wdn = K.pgd_l2_step(a, g, x, alpha, eps)
        np.testing.assert_allclose(host(gn), wgn, rtol=2e-6)
        np.testing.assert_allclose(host(dn), wdn, rtol=2e-6)
        np.testing.assert_allclose(host(got), want, atol=3e-7, rtol=0)
        # the L2-ball invariant, with float slack
        assert (np.linalg.norm((host(got) - x).astype(np.float64), axis=1) <= eps * (1 + 1e-4) + 1e-6).all()
    normal, r = randn((B, T), 14 + T), rand01((B,), 15 + T)
    got = ops.pgd_l2_init(dev(x, cuda), 0.1, draws=(dev(normal, cuda), dev(r, cuda)))
    np.testing.assert_allclose(host(got), K.pgd_l2_init_noise(x, normal, r, 0.1), atol=3e-7, rtol=0)
    got = ops.pgd_l2_init(dev(x, cuda), 0.1, seed=424242, offset=3)
    np.testing.assert_allclose(host(got), K.pgd_l2_init_philox(x, 0.1, 424242, 3), atol=1e-6, rtol=0)


def test_pgd_l2_zero_gradient_and_zero_delta(ops, cuda):
    """g = 0 -> g / (0 + 1e-10) = 0; a == x -> dn = 0 -> (1/0)*eps = inf -> min(inf, 1) = 1 (pgdl2.py:78-86)."""
    x = rand01((2, 4096), 1)
    g = np.zeros_like(x)
    got, gn, dn = ops.pgd_l2_step(dev(x, cuda), dev(g, cuda), dev(x, cuda), 0.2, 0.1, return_norms=True)
    assert same(host(got), x) and (host(gn) == 0).all() and (host(dn) == 0).all()


@pytest.mark.parametrize("B,T", [(2, 3), (3, 4099), (2, T_FULL)])
def test_cw_vs_oracle(ops, cuda, B, T):
    x = rand01((B, T), 21 + T)
    x[0, 0], x[0, 1 % T] = 0.0, 1.0  # after to_minmax every row holds an exact 0 and 1 -> w = -+inf (cw.py:117-122)
    w = host(ops.cw_init_w(dev(x, cuda)))
    ww = K.cw_init_w(x)
    assert np.isneginf(w[0, 0]) and np.isposinf(w[0, 1 % T])
    fin = np.isfinite(ww)
    np.testing.assert_allclose(w[fin], ww[fin], rtol=2e-6, atol=1e-6)
    wv = randn((B, T), 22 + T, 2.0)
    adv, l2 = ops.cw_tanh_sqdist(dev(wv, cuda), dev(x, cuda))
    wadv, wl2 = K.cw_tanh_sqdist(wv, x)
    np.testing.assert_allclose(host(adv), wadv, atol=2e-7, rtol=0)
    np.testing.assert_allclose(host(l2), wl2, rtol=1e-5)
    m, v, gm = randn((B, T), 23 + T, 0.1), np.abs(randn((B, T), 24 + T, 0.01)), randn((B, T), 25 + T, 0.5)
    dw, dm, dv = dev(wv, cuda), dev(m, cuda), dev(v, cuda)
    ops.cw_adam_step(dw, dm, dv, dev(x, cuda), dev(gm, cuda), 3, lr=0.01)
    ow, om, ov = K.cw_adam_step(wv, m, v, x, gm, 3, lr=0.01)
    np.testing.assert_allclose(host(dm), om, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(host(dv), ov, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(host(dw), ow, atol=2e-6, rtol=0)
    mask = (np.arange(B) % 2).astype(np.float32)
    best = rand01((B, T), 26 + T)
    db = dev(best, cuda)
    ops.cw_best_update(adv, dev(mask, cuda), db)
    assert same(host(db), K.cw_best_update(host(adv), mask, best))


@pytest.mark.parametrize("B", [1, 2, 64, 128, 1000])
def test_ce2_loss_grad_vs_oracle_and_torch(ops, cuda, B):
    z = randn((B, 1), 31 + B, 3.0)
    y = (np.random.default_rng(B).integers(0, 2, B)).astype(np.int64)
    for scale in (1.0, -1.0):
        dz, loss = ops.ce2_loss_grad(dev(z, cuda), dev(y, cuda), scale)
        wdz, wloss = K.ce2_loss_grad(z, y, scale)
        np.testing.assert_allclose(host(dz).ravel(), wdz, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(host(loss)[0], wloss, rtol=1e-5, atol=1e-7)
    # and against the reference's formulation, torch fp32: CE(cat([-z, z], 1), y)   (pgd.py:62,50,68)
    zt = torch.from_numpy(z).requires_grad_(True)
    cost = torch.nn.CrossEntropyLoss()(torch.cat([-zt, zt], dim=1), torch.from_numpy(y))
    (gz,) = torch.autograd.grad(cost, zt)
    dz, loss = ops.ce2_loss_grad(dev(z, cuda), dev(y, cuda), 1.0)
    # torch forms softmax - onehot in f32 (cancellation at saturated logits): absolute tolerance there
    np.testing.assert_allclose(host(dz), gz.numpy(), rtol=1e-4, atol=3e-8)
    np.testing.assert_allclose(host(loss)[0], cost.item(), rtol=1e-5)


# ---- full benchmark size: size-independent properties --------------------------------------------------------------------

def test_full_size_properties(ops, cuda):
    """B = 128, T = 64 600 (BASELINE.json config 2): idempotence, ball / box invariants, revert round trip, and
    equality with the oracle on a strided sample of rows."""
    B, T = 128, T_FULL
    gen = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.randn(B, T, generator=gen) * 0.05).clamp_(-1, 1).to(cuda)
    x01, mn, mx = ops.to_minmax(x)
    assert x01.min().item() == 0.0 and x01.max().item() == 1.0
    assert (x01.min(dim=1)[0] == 0).all() and (x01.max(dim=1)[0] == 1).all()
    back = ops.revert_minmax(x01, mn, mx)
    assert (back - x).abs().max().item() <= 1e-7                              # 4.5e-8 observed
    g = torch.randn(B, T, generator=gen).to(cuda)
    eps, alpha = 0.003, 2 / 255
    adv = ops.pgd_linf_init(x01, eps, seed=99)
    for _ in range(3):
        adv = ops.pgd_linf_step(adv, g, x01, alpha, eps)
    assert (adv - x01).abs().max().item() <= eps + 1e-7
    assert adv.min().item() >= 0.0 and adv.max().item() <= 1.0
    # alpha > eps: one step saturates the ball, so a second step with the same gradient is a fixed point
    assert torch.equal(ops.pgd_linf_step(adv, g, x01, alpha, eps), adv)
    rows = [0, 1, 63, 64, 127]
    sub = lambda t: host(t[rows])
    assert same(sub(adv), K.pgd_linf_step(sub(adv), sub(g), sub(x01), alpha, eps))
    assert same(sub(x01), K.minmax_normalize(sub(x))[0])
    l2 = ops.pgd_l2_step(x01, g, x01, 0.2, 0.1)
    assert ((l2 - x01).norm(dim=1) <= 0.1 * (1 + 1e-4)).all()  # f32 rounding of x + d*f moves the norm by ~1e-5 relative


def test_error_behaviour(ops, cuda):
    from audio_deepfake_adversarial_attacks_amd._lib import AdvstepError
    x = torch.rand(2, 100)
    with pytest.raises(AdvstepError, match="no CPU fallback"):
        ops.to_minmax(x)                                   # CPU tensor: refused, never silently computed
    with pytest.raises(TypeError):
        ops.fgsm_step(x.double().to(cuda), x.double().to(cuda), 0.1)
    with pytest.raises(ValueError):
        ops.pgd_linf_step(x.to(cuda), x.to(cuda)[:, :50], x.to(cuda), 0.1, 0.1)
    with pytest.raises(ValueError):
        ops.fgsm_step(x.to(cuda).t(), x.to(cuda).t(), 0.1)  # non-contiguous
    # empty batch is a no-op, not an error
    e = torch.empty(0, 100, device=cuda)
    assert ops.pgd_linf_step(e, e, e, 0.1, 0.1).shape == (0, 100)


@pytest.mark.parametrize("B,T", [(128, 64_600), (5, 64_600), (3, 4_099), (2, 100), (1, 1)])
def test_pgd_l2_single_pass_equals_three_kernel_path(cuda, monkeypatch, B, T):
    """The single-launch PGD-L2 step (row norms exchanged inside the launch, 16 B per sample) forms the same partial sums
    and re-reduces them with the same code as the three-kernel path (32 B per sample): bit-identical outputs and norms,
    also at the full bench shape where all 2 048 workgroups spin on each other, and run to run."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    g = torch.Generator().manual_seed(B * 31 + T)
    orig = torch.rand(B, T, generator=g).to(cuda)
    adv = (orig + (torch.rand(B, T, generator=g).to(cuda) - 0.5) * 0.01).clamp(0, 1)
    grad = (torch.randn(B, T, generator=g) * 1e-3).to(cuda)
    grad[0] = 0.0                                                   # a zero-gradient row: gn = eps_div
    outs = []
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", mode)
        outs.append(hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1, return_norms=True))
    for out, gn, dn in outs[1:]:
        assert torch.equal(out, outs[0][0]) and torch.equal(gn, outs[0][1]) and torch.equal(dn, outs[0][2])
    assert ((outs[0][0] - orig).norm(dim=1) <= 0.1 * (1 + 1e-4)).all()


@pytest.mark.parametrize("B,T", [(128, 64_600), (3, 4_099), (2, 100)])
def test_pgd_l2_philox_start_single_pass_equals_two_kernel_path(cuda, monkeypatch, B, T):
    """The single-launch Philox random start keeps the normals in registers across the in-launch exchange of their row norm;
    same draws, same partial sums: bit-identical to the two-kernel path (which generates the normals twice)."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    x = torch.rand(B, T, generator=torch.Generator().manual_seed(B + T)).to(cuda)
    outs = []
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", mode)
        outs.append(hip_ops.pgd_l2_init(x, 0.1, seed=1234, offset=5))
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
    assert ((outs[0] - x).norm(dim=1) <= 0.1 * (1 + 1e-5)).all() and not torch.equal(outs[0], x)


def _l2_inputs(B, T, cuda):
    g = torch.Generator().manual_seed(B * 31 + T)
    orig = torch.rand(B, T, generator=g).to(cuda)
    adv = (orig + (torch.rand(B, T, generator=g).to(cuda) - 0.5) * 0.01).clamp(0, 1)
    grad = (torch.randn(B, T, generator=g) * 1e-3).to(cuda)
    return adv, grad, orig


@pytest.mark.parametrize("B,T", [(128, 64_600), (5, 64_600), (3, 4_099), (2, 100)])
def test_pgd_l2_repair_pass_when_the_exchange_is_abandoned(cuda, monkeypatch, B, T):
    """ADVICE r02 / VERDICT r02 item 7: a row whose in-launch norm exchange does not complete must not come out NaN.  With
    the wait bound forced to 0 sweeps EVERY workgroup abandons its exchange, flags its row and leaves; the repair kernel
    queued behind the launch recomputes the rows with the three-kernel path's arithmetic: bit-identical outputs and norms,
    and the diagnostic counter says all B rows were repaired (0 in a normal launch)."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    adv, grad, orig = _l2_inputs(B, T, cuda)
    monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "0")
    want = hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1, return_norms=True)
    want_init = hip_ops.pgd_l2_init(orig, 0.1, seed=99, offset=3)
    monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "1")
    got = hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1, return_norms=True)
    assert hip_ops.pgd_l2_repaired_rows(adv) == 0
    monkeypatch.setenv("ADVSTEP_L2_SPIN_LIMIT", "0")
    rep = hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1, return_norms=True)
    assert hip_ops.pgd_l2_repaired_rows(adv) == B
    rep_init = hip_ops.pgd_l2_init(orig, 0.1, seed=99, offset=3)
    assert hip_ops.pgd_l2_repaired_rows(orig) == B
    for a, b, c in zip(want, got, rep):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(want_init, rep_init) and torch.isfinite(rep[0]).all()


def test_pgd_l2_single_pass_respects_device_capacity_and_aliasing(cuda, monkeypatch):
    """The single-pass path is taken only when the launch fits the device's resident capacity for the kernel (CU count x
    occupancy, queried from the runtime — ADVSTEP_L2_CAPACITY stands in for a smaller / partitioned device) and when `out`
    does not alias an input (the repair pass re-reads them); otherwise the three-kernel path runs.  Same bits every way."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    B, T = 16, 64_600
    adv, grad, orig = _l2_inputs(B, T, cuda)
    monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "0")
    want = hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1)
    monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "1")
    monkeypatch.setenv("ADVSTEP_L2_SPIN_LIMIT", "0")        # if the single-pass path ran, every row would be flagged
    monkeypatch.setenv("ADVSTEP_L2_CAPACITY", str(B * 16 - 1))
    hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1)           # leaves the flags of an earlier call untouched ...
    monkeypatch.setenv("ADVSTEP_L2_CAPACITY", str(B * 16))
    assert torch.equal(hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1), want)
    assert hip_ops.pgd_l2_repaired_rows(adv) == B            # ... this one fits: single pass, all rows repaired
    monkeypatch.delenv("ADVSTEP_L2_CAPACITY")
    monkeypatch.delenv("ADVSTEP_L2_SPIN_LIMIT")
    assert torch.equal(hip_ops.pgd_l2_step(adv, grad, orig, 0.2, 0.1), want)
    assert hip_ops.pgd_l2_repaired_rows(adv) == 0
    inplace = adv.clone()
    hip_ops.pgd_l2_step(inplace, grad, orig, 0.2, 0.1, out=inplace)      # in-place update (include/advstep.h allows it)
    assert torch.equal(inplace, want)


def _raw_l2_step(lib, ws, adv, grad, orig, stream=None):
    out = torch.empty_like(adv)
    B, T = adv.shape
    st = lib.advstep_pgd_l2_step_f32(adv.data_ptr(), grad.data_ptr(), orig.data_ptr(), out.data_ptr(), B, T, 0.2, 0.1, 1e-10,
                                     0.0, 1.0, None, None, ws.data_ptr(), ws.numel(), stream)
    assert st == 0
    return out


@pytest.mark.parametrize("fill", ["zeros", "garbage"])
def test_pgd_l2_single_pass_on_one_buffer_shared_between_shapes_and_entry_points(cuda, monkeypatch, fill):
    """ONE caller-owned scratch buffer used through the C ABI for several (B, T) and by the entry points that write
    float partial sums into it (min-max, CW's distance, the multi-kernel PGD-L2 path), with calls whose every row went through the
    repair pass in between - under round 4's layout the `last = 1` words such a call leaves lay, for another shape, on granule
    tags (1 was a phase tag).  The exchange is tagged by a call counter now: every single-pass result equals the three-kernel
    path bit for bit whatever the buffer held before, also when it was never zero-filled (random bytes: rows may take the repair
    pass, never another value)."""
    from audio_deepfake_adversarial_attacks_amd import _lib
    lib = _lib.load()
    T = 64_600
    shapes = [4, 16, 3, 128, 16, 4, 128]
    need = max(lib.advstep_row_workspace_bytes(B, T) for B in shapes)
    g = torch.Generator().manual_seed(7)
    if fill == "zeros":
        ws = torch.zeros(need, dtype=torch.uint8, device=cuda)
    else:
        ws = torch.randint(0, 256, (need,), dtype=torch.uint8, generator=g).to(cuda)
    for round_, B in enumerate(shapes * 2):
        adv, grad, orig = _l2_inputs(B, T, cuda)
        grad = grad.roll(round_, dims=1)
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "0")
        want = _raw_l2_step(lib, ws, adv, grad, orig)                   # writes float partials over the planes
        mn, mx = torch.empty(B, device=cuda), torch.empty(B, device=cuda)
        assert lib.advstep_minmax_normalize_f32(orig.data_ptr(), torch.empty_like(orig).data_ptr(), mn.data_ptr(), mx.data_ptr(),
                                                B, T, ws.data_ptr(), ws.numel(), None) == 0
        assert lib.advstep_cw_tanh_sqdist_f32(grad.data_ptr(), orig.data_ptr(), torch.empty_like(orig).data_ptr(), mn.data_ptr(),
                                              B, T, ws.data_ptr(), ws.numel(), None) == 0
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "1")
        assert torch.equal(_raw_l2_step(lib, ws, adv, grad, orig), want)
        if round_ % 3 == 0:                                              # a call that leaves every row's `last` word raised
            monkeypatch.setenv("ADVSTEP_L2_SPIN_LIMIT", "0")
            assert torch.equal(_raw_l2_step(lib, ws, adv, grad, orig), want)
            monkeypatch.delenv("ADVSTEP_L2_SPIN_LIMIT")
        assert torch.equal(_raw_l2_step(lib, ws, adv, grad, orig), want)
        x0 = torch.empty_like(orig)
        assert lib.advstep_pgd_l2_init_philox_f32(orig.data_ptr(), x0.data_ptr(), B, T, 0.1, 0.0, 1.0, 11, round_, ws.data_ptr(),
                                                  ws.numel(), None) == 0
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "0")
        x1 = torch.empty_like(orig)
        assert lib.advstep_pgd_l2_init_philox_f32(orig.data_ptr(), x1.data_ptr(), B, T, 0.1, 0.0, 1.0, 11, round_, ws.data_ptr(),
                                                  ws.numel(), None) == 0
        assert torch.equal(x0, x1)
    if fill == "zeros":     # properly used buffer, one shape at the end: the last call repaired nothing
        count = torch.zeros(1, dtype=torch.int32, device=cuda)
        monkeypatch.setenv("ADVSTEP_L2_SINGLE_PASS", "1")
        B = shapes[-1]
        adv, grad, orig = _l2_inputs(B, T, cuda)
        _raw_l2_step(lib, ws, adv, grad, orig)
        _raw_l2_step(lib, ws, adv, grad, orig)
        assert lib.advstep_pgd_l2_repaired_rows(ws.data_ptr(), ws.numel(), B, T, count.data_ptr(), None) == 0
        assert int(count.item()) == 0


_TWO_PROCESS_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["ADVSTEP_REPO"])
from audio_deepfake_adversarial_attacks_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(sys.argv[1]))
B, T = 128, 64600
orig = torch.rand(B, T, generator=g).to(dev)
adv = (orig + (torch.rand(B, T, generator=g).to(dev) - 0.5) * 0.01).clamp(0, 1)
grad = (torch.randn(B, T, generator=g) * 1e-3).to(dev)
repaired = 0
cur = adv
for it in range(int(sys.argv[2])):
    os.environ["ADVSTEP_L2_SINGLE_PASS"] = "1"
    got = hip_ops.pgd_l2_step(cur, grad, orig, 0.2, 0.1)
    repaired += hip_ops.pgd_l2_repaired_rows(cur)
    os.environ["ADVSTEP_L2_SINGLE_PASS"] = "0"
    want = hip_ops.pgd_l2_step(cur, grad, orig, 0.2, 0.1)
    if not torch.equal(got, want) or not torch.isfinite(got).all():
        print("MISMATCH at iteration", it, flush=True)
        sys.exit(3)
    cur = got
    grad = grad.roll(1, dims=1)
print("ok repaired_rows", repaired, flush=True)
"""


def test_pgd_l2_single_pass_with_two_processes_sharing_the_device(cuda, tmp_path):
    """Two processes on the ONE device, both running the single-pass PGDL2 step at B = 128 (2 048 workgroups each: together
    twice the resident capacity) for 150 iterations, each comparing every result with the three-kernel path: bit-identical
    and finite, whether or not a row had to go through the repair pass (the count is printed)."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    script = tmp_path / "two_proc.py"
    script.write_text(_TWO_PROCESS_SCRIPT)
    env = dict(os.environ, ADVSTEP_REPO=str(ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(seed), "150"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for seed in (1, 2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok repaired_rows" in o, o[-2000:]
