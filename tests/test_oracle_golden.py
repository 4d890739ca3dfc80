"""CPU: the oracle (plain-C kernels + torch-level whole-attack restatement) against the golden vectors that the
REFERENCE ITSELF produced (tests/golden/generate_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import attacks as OA
from oracle import kernels as K
from tests.helpers import surrogate_from

T = torch.from_numpy


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(autouse=True)
def one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)  # the fixtures were generated single-threaded; the reference is bit-reproducible then
    yield
    torch.set_num_threads(n)


def test_minmax_bit_exact(golden):
    g = golden("minmax")
    for tag in ("full", "ragged", "const"):
        x01, mn, mx = K.minmax_normalize(g[f"{tag}_x"])
        assert same(x01, g[f"{tag}_x01"]) and same(mn, g[f"{tag}_mn"].ravel()) and same(mx, g[f"{tag}_mx"].ravel())
        if tag != "const":
            assert same(K.minmax_revert(g[f"{tag}_p"], g[f"{tag}_mn"], g[f"{tag}_mx"]), g[f"{tag}_revert"])
    assert np.isnan(g["const_x01"][1]).all() and np.isfinite(g["const_x01"][0]).all()


def test_fgsm_step_bit_exact(golden):
    g = golden("fgsm")
    for tag in ("ragged", "small"):
        for e in ("e0005", "e00075", "e001"):
            p = f"{tag}_{e}_"
            assert same(K.fgsm_step(g[p + "x"], g[p + "grad"], float(g[p + "eps"])), g[p + "adv"]), p


def test_pgd_linf_bit_exact(golden):
    g = golden("pgd_linf")
    for tag in ("ragged_rs", "small_nors", "full_rs"):
        eps, alpha, steps = float(g[tag + "_eps"]), float(g[tag + "_alpha"]), int(g[tag + "_steps"])
        if tag + "_noise" in g:
            assert same(K.pgd_linf_init_noise(g[tag + "_x"], g[tag + "_noise"]), g[tag + "_a0"])
        for k in range(steps):
            got = K.pgd_linf_step(g[f"{tag}_a{k}"], g[f"{tag}_g{k}"], g[tag + "_x"], alpha, eps)
            assert same(got, g[f"{tag}_a{k + 1}"]), (tag, k)


def test_pgd_l2_within_norm_tolerance(golden):
    g = golden("pgd_l2")
    for tag in ("ragged_rs", "small_nors"):
        eps, alpha, steps = float(g[tag + "_eps"]), float(g[tag + "_alpha"]), int(g[tag + "_steps"])
        if tag + "_normal" in g:
            np.testing.assert_allclose(K.pgd_l2_init_noise(g[tag + "_x"], g[tag + "_normal"], g[tag + "_r"], eps),
                                       g[tag + "_a0"], atol=1e-7, rtol=0)
        for k in range(steps):
            got, gn, dn = K.pgd_l2_step(g[f"{tag}_a{k}"], g[f"{tag}_g{k}"], g[tag + "_x"], alpha, eps,
                                        float(g[tag + "_eps_div"]))
            np.testing.assert_allclose(gn, g[f"{tag}_gnorm{k}"], rtol=1e-6)
            np.testing.assert_allclose(dn, g[f"{tag}_dnorm{k}"], rtol=1e-6)
            np.testing.assert_allclose(got, g[f"{tag}_a{k + 1}"], atol=1.2e-7, rtol=0)  # <= 2 ulp at 1.0


def test_cw_kernels_within_tolerance(golden):
    g = golden("cw")
    x = g["x"]
    w0 = K.cw_init_w(x)
    fin = np.isfinite(g["w0"])
    assert same(np.isinf(w0), np.isinf(g["w0"])) and (~fin).sum() == 2 * x.shape[0]  # the exact 0 and 1 of each row
    np.testing.assert_allclose(w0[fin], g["w0"][fin], rtol=1e-6, atol=1e-7)
    m, v = np.zeros_like(x), np.zeros_like(x)
    for k in range(4):
        adv, l2 = K.cw_tanh_sqdist(g[f"s{k}_w"], x)
        np.testing.assert_allclose(adv, g[f"s{k}_adv"], atol=1.2e-7, rtol=0)
        np.testing.assert_allclose(l2, g[f"s{k}_l2"], rtol=1e-4, atol=1e-9)
        w, m2, v2 = K.cw_adam_step(g[f"s{k}_w"], m, v, x, g[f"s{k}_grad_adv"], k + 1, lr=float(g["lr"]))
        np.testing.assert_allclose(m2, g[f"s{k}_m_after"], rtol=1e-5, atol=3e-8)
        np.testing.assert_allclose(v2, g[f"s{k}_v_after"], rtol=1e-4, atol=1e-12)
        resolved = np.abs(g[f"s{k}_grad_w"]) > 1e-4
        np.testing.assert_allclose(w[resolved], g[f"s{k}_w_after"][resolved], atol=1e-6, rtol=0)
        m, v = g[f"s{k}_m_after"], g[f"s{k}_v_after"]
    # best-so-far blend: exact
    mask = np.array([1, 0, 1, 0], np.float32)
    want = mask[:, None] * g["s1_adv"] + (1 - mask[:, None]) * g["s0_adv"]
    assert same(K.cw_best_update(g["s1_adv"], mask, g["s0_adv"]), want)


def test_ce2_closed_form_matches_reference_loss(golden):
    """a8: CE(cat([-z, z], 1), y) in closed form vs the torch ops the reference calls (pgd.py:62,50,68)."""
    g = golden("fgsm")
    z = g["small_e001_logits"]
    y = g["small_e001_y"]
    zt = T(z).clone().requires_grad_(True)
    cost = torch.nn.CrossEntropyLoss()(torch.cat([-zt, zt], dim=1), T(y))
    (gz,) = torch.autograd.grad(cost, zt)
    dz, loss = K.ce2_loss_grad(z, y)
    np.testing.assert_allclose(dz, gz.numpy().ravel(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(loss, cost.item(), rtol=1e-6)
    dz_t, loss_t = K.ce2_loss_grad(z, y, scale=-1.0)
    np.testing.assert_allclose(dz_t, -dz, rtol=1e-6) and np.testing.assert_allclose(loss_t, -loss, rtol=1e-6)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert [hex(v) for v in K.philox_raw(0, 0, 0)] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ones = 0xFFFFFFFFFFFFFFFF
    assert [hex(v) for v in K.philox_raw(ones, ones, ones)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    pi = [hex(v) for v in K.philox_raw(0x85a308d3243f6a88, 0x0370734413198a2e, 0x299f31d0a4093822)]
    assert pi == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_random_start_statistics():
    x = np.full((4, 50_000), 0.5, np.float32)
    a = K.pgd_linf_init_philox(x, 0.003, seed=7, offset=0)
    d = a - x
    assert np.abs(d).max() <= 0.003 + 1e-7 and abs(d.mean()) < 3e-5 and abs(d.std() - 0.003 / np.sqrt(3)) < 2e-5
    assert not same(a, K.pgd_linf_init_philox(x, 0.003, seed=8, offset=0))
    assert same(a, K.pgd_linf_init_philox(x, 0.003, seed=7, offset=0))
    b = K.pgd_l2_init_philox(x, 0.1, seed=7, offset=0)
    n = np.linalg.norm((b - x).astype(np.float64), axis=1)
    assert (n <= 0.1 + 1e-6).all() and (n > 0).all() and len(set(np.round(n, 6))) == 4  # radius r*eps, r ~ U(0,1) per row


# ---- whole attacks: oracle/attacks.py (torch ops in the reference's order) == reference, bit for bit --------------------

def test_whole_attacks_bit_identical_to_reference(golden):
    g = golden("fgsm")
    m = surrogate_from(g)
    for e, eps in (("e0005", 0.0005), ("e00075", 0.00075), ("e001", 0.001)):
        with OA.attack_mode(m):
            adv = OA.fgsm(m, T(g[f"small_{e}_x"]), T(g[f"small_{e}_y"]), eps=eps)
        assert torch.equal(adv, T(g[f"small_{e}_adv"]))
    g = golden("pgd_linf")
    m = surrogate_from(g)
    for tag in ("ragged_rs", "small_nors"):
        with OA.attack_mode(m):
            adv = OA.pgd(m, T(g[tag + "_x"]), T(g[tag + "_y"]), eps=float(g[tag + "_eps"]), steps=int(g[tag + "_steps"]),
                         random_start=tag.endswith("_rs"), noise=T(g[tag + "_noise"]) if tag + "_noise" in g else None)
        assert torch.equal(adv, T(g[tag + "_adv"])), tag
    g = golden("pgd_l2")
    m = surrogate_from(g)
    for tag in ("ragged_rs", "small_nors"):
        draws = (T(g[tag + "_normal"]), T(g[tag + "_r"])) if tag + "_normal" in g else None
        with OA.attack_mode(m):
            adv = OA.pgdl2(m, T(g[tag + "_x"]), T(g[tag + "_y"]), eps=float(g[tag + "_eps"]), steps=int(g[tag + "_steps"]),
                           random_start=tag.endswith("_rs"), draws=draws)
        assert torch.equal(adv, T(g[tag + "_adv"])), tag
    g = golden("cw")
    m = surrogate_from(g)
    with OA.attack_mode(m):
        best = OA.cw(m, T(g["x"]), T(g["y"]), c=float(g["c"]), steps=int(g["steps"]), lr=float(g["lr"]))
    assert torch.equal(best, T(g["best"]))
    assert (best != T(g["x"])).any()  # the blend fired: the fixture is not the trivial "nothing fooled the model" case


def test_torch_oracle_minmax_matches_c_oracle():
    x = torch.randn(5, 777) * 0.05
    x01, mn, mx = OA.to_minmax(x)
    c01, cmn, cmx = K.minmax_normalize(x.numpy())
    assert same(x01.numpy(), c01) and same(mn.numpy().ravel(), cmn) and same(mx.numpy().ravel(), cmx)
    p = torch.rand(5, 777)
    assert same(OA.revert_minmax(p, mn, mx).numpy(), K.minmax_revert(p.numpy(), cmn, cmx))


# ---- FAB: floating-point parity, tolerances stated per check -------------------------------------------
def _fab_rows(golden, T):
    from tests.helpers import fab_projection_inputs

    g = golden("fab_projection")
    t, w, b = fab_projection_inputs(T, int(g[f"T{T}_seed"]))
    # the inputs are regenerated from the seed: make sure this torch build draws the same numbers
    assert np.allclose([t.double().sum().item(), w.double().sum().item(), b.double().sum().item()], g[f"T{T}_checksum"],
                       rtol=1e-12, atol=1e-9)
    assert same(b.numpy(), g[f"T{T}_b"])
    return g, t.numpy(), w.numpy(), b.numpy()


@pytest.mark.parametrize("T", [257, 4099, 64600])
def test_fab_projections_match_reference(golden, T):
    """oracle/fab.py's float64 sorted-breakpoint solution against the reference's float32 sort/cumsum/bisection
    (fab.py:562-717) on the same rows.  Linf / L2: 2e-6 max-abs on moves of size <= 1; L1: the one partially moved
    coordinate is residual / w_i with float32 cancellation in the residual, 5e-4."""
    from oracle import fab as OF

    g, t, w, b = _fab_rows(golden, T)
    for name, tol in (("linf", 2e-6), ("l2", 2e-6), ("l1", 5e-4)):
        got = getattr(OF, "projection_" + name)(t, w, b)
        want = g[f"T{T}_{name}"]
        assert np.abs(got - want).max() <= tol, (name, np.abs(got - want).max())


def test_fab_attack_matches_reference(golden):
    """Whole FAB runs of the reference (surrogate detector, eta = 1.05) against oracle/fab.py: 5e-6 max-abs on the
    adversarial waveform after 8-12 iterations; rows the detector already misclassifies stay bit-identical."""
    from oracle import fab as OF

    g = golden("fab_attack")
    model = surrogate_from(g)
    x01, y = T(g["x01"]), T(g["labels"])
    for name, norm in (("linf", "Linf"), ("linf_tight", "Linf"), ("l2", "L2")):
        eta, steps, eps = g[f"{name}_params"]
        adv = OF.fab(model, x01, y, norm=norm, eps=float(eps), steps=int(steps), eta=float(eta)).numpy()
        assert np.abs(adv - g[f"{name}_adv"]).max() <= 5e-6, name
        assert same(adv[2], g["x01"][2])
        run = OF.attack_single_run(model, x01, y, norm, float(eps), int(steps), 0.1, float(eta), 0.9).numpy()
        assert np.abs(run - g[f"{name}_single_run"]).max() <= 5e-6, name
    # eps = 0.26 rejects every row whose best adversarial point is farther than eps (fab.py:518-526)
    assert same(g["linf_tight_adv"], g["x01"]) and not same(g["linf_tight_single_run"], g["x01"])
    # L1 moves coordinates one at a time in order of |w|: a last-bit difference in the residual changes WHICH coordinates
    # move, so rows agree either coordinate-wise (2e-4) or in the size of the perturbation (5e-4 relative L1 norm)
    run = OF.attack_single_run(model, x01, y, "L1", 5.0, 12, 0.1, 1.05, 0.9).numpy()
    err = np.abs(run - g["l1_single_run"]).max(axis=1)
    n_mine, n_ref = np.abs(run - g["x01"]).sum(axis=1), np.abs(g["l1_single_run"] - g["x01"]).sum(axis=1)
    assert (err <= 2e-4).sum() >= 5 and np.allclose(n_mine, n_ref, rtol=5e-4)
    assert bool(g["l1_forward_raises"])
    with pytest.raises(UnboundLocalError):
        OF.fab(model, x01, y, norm="L1", steps=2)


def test_fab_iterations_replay_reference_trace(golden):
    """Step by step: from the reference's own x1 at iteration k, one oracle iteration lands on the reference's x1 at
    k + 1 (2e-6), including AttackEnum.FAB's eta = 10 overshoot, whose whole-run trajectory is too sensitive to compare
    end to end."""
    from oracle import fab as OF

    g = golden("fab_attack")
    model = surrogate_from(g)
    x01, y = T(g["x01"]), T(g["labels"])
    rows = torch.tensor([0, 1, 3, 4, 5])  # row 2 is misclassified from the start and never enters the run
    x0, la = x01[rows], y[rows]
    for name in ("linf", "linf_eta10"):
        eta = float(g[f"{name}_params"][0])
        trace = T(g[f"{name}_x1"])
        adv, res2 = x0.clone(), torch.full((5,), 1e10)
        for k in range(trace.shape[0] - 1):
            nxt, adv, res2 = OF.fab_iteration(model, trace[k], x0, la, adv, res2, "Linf", eta, 0.9, 0.1)
            assert (nxt - trace[k + 1]).abs().max() <= 2e-6, (name, k, (nxt - trace[k + 1]).abs().max())
