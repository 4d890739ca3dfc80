#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run ONLY in the build container, where the Python reference is mounted read-only at /root/reference:

    python tests/golden/generate_golden.py

It imports the reference's own modules (adversarial_attacks.torchattacks, src.aa.utils, src.metrics, the
BaseLCNN / BaseSpecRNet bodies and RawNet3 behind its first layer), runs them on small seeded inputs with torch CPU kernels and stores inputs,
intermediate tensors and outputs as .npz files.  The .npz files are data (inputs + expected outputs); no
reference source travels.  The GPU box never runs this script (there is no /root/reference there).

What is recorded and how
  * per-step tensors of FGSM / PGD / PGDL2 / CW are captured by wrapping `torch.autograd.grad` and the
    attacked model's forward while the reference's `forward` runs unmodified;
  * random starts are reproduced by re-seeding torch's global generator and replaying the same draw calls
    the reference makes (pgd.py:56, pgdl2.py:57,60); the replay is asserted against the first model input;
  * the attacked model is a tiny pure-torch surrogate detector (Conv1d -> tanh -> mean -> Linear, (B,T)->(B,1))
    whose weights are stored in the fixture;
  * `src/models/lcnn.py` and `src/models/specrnet.py` import `src.frontends`, which imports torchaudio (absent
    here).  To import the model BODIES (BaseLCNN / BaseSpecRNet, which never touch torchaudio) an inert module
    object named `torchaudio` is placed in sys.modules; it performs no arithmetic and the frontends are NOT
    pinned by these fixtures (see DESIGN.md "parity unpinned" list).
"""
from __future__ import annotations

import contextlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
OUT = HERE          # `main(out_dir)` / `--out DIR` redirect the fixtures (tests/test_golden_recipe.py regenerates into a temp dir)
T_FULL = 64_600
T_RAGGED = 4_099  # odd: exercises the scalar tails / unaligned rows
T_SMALL = 4_096


def _import_reference():
    if not REF.exists():
        sys.exit("the reference is not mounted at /root/reference; fixtures can only be generated in the build container")
    sys.path.insert(0, str(REF))


class Surrogate(torch.nn.Module):
    """(B, T) -> (B, 1) differentiable detector with a handful of parameters."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv1d(1, 4, kernel_size=9, stride=4)
        self.fc = torch.nn.Linear(4, 1)

    def forward(self, x):
        h = torch.tanh(self.conv(x.unsqueeze(1)) * 8.0)
        return self.fc(h.mean(dim=2)) * 4.0


def surrogate(seed: int) -> Surrogate:
    torch.manual_seed(seed)
    m = Surrogate()
    return m.eval()


def waveforms(B: int, T: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, generator=g) * 0.05
    return x.clamp_(-1.0, 1.0)


def npy(t) -> np.ndarray:
    return t.detach().cpu().numpy().copy()


@contextlib.contextmanager
def record_attack(model):
    """Record every model input and every tensor returned by torch.autograd.grad while the body runs."""
    rec = {"inputs": [], "grads": [], "logits": []}
    orig_forward = model.forward
    orig_grad = torch.autograd.grad

    def fwd(x):
        rec["inputs"].append(x.detach().clone())
        out = orig_forward(x)
        rec["logits"].append(out.detach().clone())
        return out

    def grad(*a, **k):
        g = orig_grad(*a, **k)
        rec["grads"].append(g[0].detach().clone())
        return g

    model.forward = fwd
    torch.autograd.grad = grad
    try:
        yield rec
    finally:
        model.forward = orig_forward
        torch.autograd.grad = orig_grad


def gen_minmax(aa_utils):
    out = {}
    for tag, B, T, seed in (("full", 1, T_FULL, 11), ("ragged", 3, T_RAGGED, 12)):
        x = waveforms(B, T, seed)
        x01, mn, mx = aa_utils.to_minmax(x)
        # perturb inside [0, 1] so revert is exercised on non-trivial data
        g = torch.Generator().manual_seed(seed + 100)
        p = (x01 + (torch.rand(B, T, generator=g) - 0.5) * 2e-3).clamp(0, 1)
        back = aa_utils.revert_minmax(p, mn, mx)
        out.update({f"{tag}_x": npy(x), f"{tag}_x01": npy(x01), f"{tag}_mn": npy(mn), f"{tag}_mx": npy(mx),
                    f"{tag}_p": npy(p), f"{tag}_revert": npy(back)})
    # constant row -> NaN (src/aa/utils.py:8-9 divides by zero)
    xc = waveforms(2, 64, 13)
    xc[1] = 0.25
    x01, mn, mx = aa_utils.to_minmax(xc)
    out.update({"const_x": npy(xc), "const_x01": npy(x01), "const_mn": npy(mn), "const_mx": npy(mx)})
    np.savez_compressed(OUT / "minmax.npz", **out)


def gen_fgsm(ta, aa_utils):
    out = {}
    model = surrogate(21)
    out.update({f"model_{k}": npy(v) for k, v in model.state_dict().items()})
    for tag, B, T, seed in (("ragged", 3, T_RAGGED, 22), ("small", 4, T_SMALL, 23)):
        x01, _, _ = aa_utils.to_minmax(waveforms(B, T, seed))
        y = torch.tensor([1, 0, 1, 0][:B])
        for eps_tag, eps in (("e0005", 0.0005), ("e00075", 0.00075), ("e001", 0.001)):
            atk = ta.FGSM(model, eps=eps)
            atk.set_training_mode(model_training=True, batchnorm_training=False)
            with record_attack(model) as rec:
                adv = atk(x01, y)
            assert len(rec["grads"]) == 1
            out.update({f"{tag}_{eps_tag}_x": npy(x01), f"{tag}_{eps_tag}_y": npy(y),
                        f"{tag}_{eps_tag}_grad": npy(rec["grads"][0]), f"{tag}_{eps_tag}_logits": npy(rec["logits"][0]),
                        f"{tag}_{eps_tag}_adv": npy(adv), f"{tag}_{eps_tag}_eps": np.float64(eps)})
    np.savez_compressed(OUT / "fgsm.npz", **out)


def gen_pgd_linf(ta, aa_utils):
    out = {}
    model = surrogate(31)
    out.update({f"model_{k}": npy(v) for k, v in model.state_dict().items()})
    cases = (
        # tag, B, T, seed, eps, steps, random_start
        ("ragged_rs", 3, T_RAGGED, 32, 0.003, 3, True),     # BASELINE config 2 hyper-parameters (alpha default 2/255)
        ("small_nors", 4, T_SMALL, 33, 0.0005, 3, False),   # AttackEnum.PGD eps (aa_types.py:9)
        ("full_rs", 1, T_FULL, 34, 0.003, 1, True),
    )
    for tag, B, T, seed, eps, steps, rs in cases:
        x01, _, _ = aa_utils.to_minmax(waveforms(B, T, seed))
        y = torch.tensor([0, 1, 1, 0][:B])
        atk = ta.PGD(model, eps=eps, steps=steps, random_start=rs)
        atk.set_training_mode(model_training=True, batchnorm_training=False)
        torch.manual_seed(seed + 1000)
        with record_attack(model) as rec:
            adv = atk(x01, y)
        out.update({f"{tag}_x": npy(x01), f"{tag}_y": npy(y), f"{tag}_adv": npy(adv),
                    f"{tag}_eps": np.float64(eps), f"{tag}_alpha": np.float64(atk.alpha), f"{tag}_steps": np.int64(steps)})
        if rs:
            torch.manual_seed(seed + 1000)
            noise = torch.empty_like(x01).uniform_(-eps, eps)  # replay of pgd.py:56
            a0 = torch.clamp(x01 + noise, min=0, max=1)
            assert torch.equal(a0, rec["inputs"][0]), "noise replay does not reproduce the reference's start"
            out[f"{tag}_noise"] = npy(noise)
        for k in range(steps):
            out[f"{tag}_a{k}"] = npy(rec["inputs"][k])
            out[f"{tag}_g{k}"] = npy(rec["grads"][k])
        out[f"{tag}_a{steps}"] = npy(adv)
    np.savez_compressed(OUT / "pgd_linf.npz", **out)


def gen_pgd_l2(ta, aa_utils):
    out = {}
    model = surrogate(41)
    out.update({f"model_{k}": npy(v) for k, v in model.state_dict().items()})
    cases = (
        ("ragged_rs", 3, T_RAGGED, 42, 0.1, 3, True),      # AttackEnum.PGDL2 eps (aa_types.py:13), alpha default 0.2
        ("small_nors", 4, T_SMALL, 43, 0.2, 3, False),
    )
    for tag, B, T, seed, eps, steps, rs in cases:
        x01, _, _ = aa_utils.to_minmax(waveforms(B, T, seed))
        y = torch.tensor([1, 1, 0, 0][:B])
        atk = ta.PGDL2(model, eps=eps, steps=steps, random_start=rs)
        atk.set_training_mode(model_training=True, batchnorm_training=False)
        torch.manual_seed(seed + 1000)
        with record_attack(model) as rec:
            adv = atk(x01, y)
        out.update({f"{tag}_x": npy(x01), f"{tag}_y": npy(y), f"{tag}_adv": npy(adv), f"{tag}_eps": np.float64(eps),
                    f"{tag}_alpha": np.float64(atk.alpha), f"{tag}_eps_div": np.float64(atk.eps_for_division),
                    f"{tag}_steps": np.int64(steps)})
        if rs:
            torch.manual_seed(seed + 1000)
            normal = torch.empty_like(x01).normal_()                     # replay of pgdl2.py:57
            n = normal.view(B, -1).norm(p=2, dim=1).view(B, 1)           # :58-59
            r = torch.zeros_like(n).uniform_(0, 1)                       # :60
            a0 = torch.clamp(x01 + normal * (r / n * eps), min=0, max=1)
            assert torch.equal(a0, rec["inputs"][0]), "draw replay does not reproduce the reference's start"
            out[f"{tag}_normal"] = npy(normal)
            out[f"{tag}_r"] = npy(r.view(-1))
            out[f"{tag}_nnorm"] = npy(n.view(-1))
        for k in range(steps):
            a_k, g_k = rec["inputs"][k], rec["grads"][k]
            out[f"{tag}_a{k}"] = npy(a_k)
            out[f"{tag}_g{k}"] = npy(g_k)
            # the reference's own row norms for this step (pgdl2.py:78,83), recomputed with the same torch calls
            gn = torch.norm(g_k.view(B, -1), p=2, dim=1)
            a_mid = a_k + atk.alpha * (g_k / (gn + atk.eps_for_division).view(B, 1))
            dn = torch.norm((a_mid - x01).view(B, -1), p=2, dim=1)
            out[f"{tag}_gnorm{k}"] = npy(gn)
            out[f"{tag}_dnorm{k}"] = npy(dn)
        out[f"{tag}_a{steps}"] = npy(adv)
    np.savez_compressed(OUT / "pgd_l2.npz", **out)


def gen_cw(ta, aa_utils):
    out = {}
    model = surrogate(51)
    out.update({f"model_{k}": npy(v) for k, v in model.state_dict().items()})
    # c = 100 / 20 steps: the reference runs 19 optimiser steps before its early stop (cw.py:107-110) and the
    # best-so-far blend (cw.py:94-103) fires for part of the batch
    B, T, steps, c, lr, keep = 4, 2048, 20, 100.0, 0.01, 4
    x01, _, _ = aa_utils.to_minmax(waveforms(B, T, 52))
    y = torch.tensor([1, 0, 0, 1])
    atk = ta.CW(model, c=c, kappa=0, steps=steps, lr=lr)
    atk.set_training_mode(model_training=True, batchnorm_training=False)

    # Record the optimiser's view of every step: w before, grad, and (w, m, v) after.
    trace = []
    RealAdam = torch.optim.Adam

    class TracingAdam(RealAdam):
        def step(self, closure=None):
            p = self.param_groups[0]["params"][0]
            entry = {"w": p.detach().clone(), "grad_w": p.grad.detach().clone()}
            r = super().step(closure)
            st = self.state[p]
            entry.update({"w_after": p.detach().clone(), "m_after": st["exp_avg"].detach().clone(),
                          "v_after": st["exp_avg_sq"].detach().clone()})
            trace.append(entry)
            return r

    ref_cw_module = sys.modules[ta.CW.__module__]
    ref_cw_module.optim.Adam = TracingAdam
    try:
        with record_attack(model) as rec:
            best = atk(x01, y)
    finally:
        ref_cw_module.optim.Adam = RealAdam
    # cw.py:107-110: with steps < 10 the early-stop test runs every step, so fewer than `steps` updates may happen
    assert 1 <= len(trace) <= steps
    out["steps_done"] = np.int64(len(trace))
    print("cw: optimiser steps executed by the reference:", len(trace), "of", steps)
    out.update({"x": npy(x01), "y": npy(y), "best": npy(best), "c": np.float64(c), "lr": np.float64(lr),
                "steps": np.int64(steps), "kappa": np.float64(0)})
    out["w0"] = npy(atk.inverse_tanh_space(x01))
    out["all_l2"] = np.stack([npy(((rec["inputs"][k] - x01) ** 2).sum(dim=1)) for k in range(len(trace))])
    out["all_logits"] = np.stack([npy(rec["logits"][k]) for k in range(len(trace))])
    for k, e in enumerate(trace[:keep]):
        adv_k = rec["inputs"][k]
        # model-side gradient d(c * sum f)/d adv for this step, with the reference's own f() (cw.py:125-134)
        a = adv_k.clone().requires_grad_(True)
        o = model.forward(a)
        o = torch.cat([-o, o], dim=1)
        gm = torch.autograd.grad(c * atk.f(o, y).sum(), a)[0]
        l2 = ((adv_k - x01) ** 2).sum(dim=1)
        out.update({f"s{k}_w": npy(e["w"]), f"s{k}_grad_w": npy(e["grad_w"]), f"s{k}_adv": npy(adv_k),
                    f"s{k}_l2": npy(l2), f"s{k}_grad_adv": npy(gm), f"s{k}_logits": npy(rec["logits"][k]),
                    f"s{k}_w_after": npy(e["w_after"]), f"s{k}_m_after": npy(e["m_after"]),
                    f"s{k}_v_after": npy(e["v_after"])})
    np.savez_compressed(OUT / "cw.npz", **out)


def fab_projection_inputs(T: int, seed: int):
    """(t, w, b) rows for the FAB projections, regenerated from the seed by the tests (only outputs are stored):
    points in [0, 1] with exact 0s and 1s, hyperplane normals with exact zeros, offsets that put the hyperplane very
    close, inside reach on either side, and out of the box's reach."""
    g = torch.Generator().manual_seed(seed)
    R = 6
    t = torch.rand(R, T, generator=g)
    t[:, ::7] = 0.0
    t[:, 3::11] = 1.0
    w = torch.randn(R, T, generator=g) * 0.01
    w[:, ::13] = 0.0
    dot = (w * t).sum(1)
    b = dot + torch.tensor([1e-4, -1e-3, 0.05, -0.2, 0.45, 5.0]) * w.abs().sum(1)
    return t, w, b


def gen_fab_projections():
    """fab.py:562-717 called directly on seeded rows; only the outputs (and input checksums) are stored."""
    from adversarial_attacks.torchattacks.attacks import fab as rfab

    out = {}
    for T, seed in ((257, 11), (T_RAGGED, 12), (T_FULL, 13)):
        t, w, b = fab_projection_inputs(T, seed)
        key = f"T{T}"
        out[f"{key}_seed"] = np.int64(seed)
        out[f"{key}_checksum"] = np.array([t.double().sum().item(), w.double().sum().item(), b.double().sum().item()])
        out[f"{key}_b"] = npy(b)
        for name in ("linf", "l2", "l1"):
            d = getattr(rfab, "projection_" + name)(t.clone(), w.clone(), b.clone())
            out[f"{key}_{name}"] = npy(d)
    np.savez_compressed(OUT / "fab_projection.npz", **out)


def gen_fab_attack(ta, aa_utils):
    """FAB.forward / attack_single_run on the surrogate detector, with every model input recorded: per iteration the
    reference evaluates the model at x1 (fab.py:93) and at the combined point before the backward step (fab.py:269)."""
    model = surrogate(7)
    x = waveforms(6, T_SMALL, 3)
    x01, _, _ = aa_utils.to_minmax(x)
    y = (model(x01).flatten() > 0).long()
    y[2] = 1 - y[2]  # one utterance the detector already gets wrong: FAB must leave it untouched
    out = {f"model_{k}": npy(v) for k, v in model.state_dict().items()}
    out["x01"] = npy(x01)
    out["labels"] = npy(y)
    runs = {
        "linf": dict(norm="Linf", eta=1.05, steps=8, eps=None),
        "linf_eta10": dict(norm="Linf", eta=10, steps=5, eps=None),       # AttackEnum.FAB's overshoot
        "linf_tight": dict(norm="Linf", eta=1.05, steps=12, eps=0.26),    # some rows end above eps: rejected
        "l2": dict(norm="L2", eta=1.05, steps=12, eps=20.0),
    }
    for name, kw in runs.items():
        atk = ta.FAB(model, n_classes=2, **kw)
        atk.set_training_mode(True, False, False)
        with record_attack(model) as rec:
            adv = atk(x01, y)
        # inputs: [perturb's clean pass, single-run's clean pass, (x1_k, combined_k) * steps, final check]
        ins = rec["inputs"]
        assert len(ins) == 3 + 2 * kw["steps"], len(ins)
        out[f"{name}_adv"] = npy(adv)
        if name in ("linf", "linf_eta10"):   # per-iteration traces only where the tests replay step by step
            out[f"{name}_x1"] = np.stack([npy(v) for v in ins[2:-1:2]])
        if name == "linf":
            out[f"{name}_combined"] = np.stack([npy(v) for v in ins[3:-1:2]])
        out[f"{name}_single_run"] = npy(atk.attack_single_run(x01, y))
        out[f"{name}_params"] = np.array([kw["eta"], kw["steps"], atk.eps])
    # L1 through attack_single_run only: FAB.forward raises for L1 in the reference (fab.py:518-522, `res` unassigned)
    atk = ta.FAB(model, n_classes=2, norm="L1", eta=1.05, steps=12)
    out["l1_single_run"] = npy(atk.attack_single_run(x01, y))
    try:
        atk(x01, y)
        out["l1_forward_raises"] = np.bool_(False)
    except UnboundLocalError:
        out["l1_forward_raises"] = np.bool_(True)
    np.savez_compressed(OUT / "fab_attack.npz", **out)


class TinyDetectionSet(torch.utils.data.Dataset):
    """(waveform, sample_rate, label) triples like the reference's DetectionDataset items, held in memory."""

    def __init__(self, n: int, T: int, seed: int):
        self.x = waveforms(n, T, seed)
        self.y = torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(seed + 1))

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], 16_000, int(self.y[i])


TRAINER_RUNS = {   # strategy -> attacks (reference AttackEnum names; FGSM only: no random start, bit-reproducible)
    "RANDOM": ["FGSM", "FGSM_eps00075", "FGSM_eps001"],
    "EQUAL": ["FGSM_eps001"],
    "ONLY_ADV": ["FGSM_eps00075"],
    "ADAPTIVE": ["FGSM", "FGSM_eps001"],
    "ADAPTIVE_V2": ["FGSM", "FGSM_eps00075", "FGSM_eps001"],
}


def gen_trainer():
    """src/trainer.py's five adversarial-training strategies, 2 epochs on a tiny in-memory set with the surrogate
    detector on CPU: final (best) weights, every log line's numbers, the adaptive attack weights."""
    import logging
    import random

    from src.aa.aa_trainer_types import AdversarialGDTrainerEnum

    out = {}
    train, test = TinyDetectionSet(16, 1024, 21), TinyDetectionSet(8, 1024, 22)
    for strategy, attacks in TRAINER_RUNS.items():
        model = surrogate(7).train()      # (re-seeds torch: seed the run afterwards)
        random.seed(3)
        np.random.seed(3)
        torch.manual_seed(3)
        records = []
        handler = logging.Handler()
        handler.emit = lambda rec: records.append(rec.getMessage())
        log = logging.getLogger("src.trainer")
        log.setLevel(logging.INFO)
        log.addHandler(handler)
        try:
            trainer = AdversarialGDTrainerEnum[strategy].value(epochs=2, batch_size=4, device="cpu",
                                                               optimizer_kwargs={"lr": 1e-3})
            trained = trainer.train(dataset=train, model=model, attack_model=model, adversarial_attacks=attacks,
                                    test_dataset=test)
        finally:
            log.removeHandler(handler)
        for k, v in trained.state_dict().items():
            out[f"{strategy}_model_{k}"] = npy(v)
        out[f"{strategy}_log"] = np.array([m for m in records if m.startswith(("Epoch [", "[0"))])
        if hasattr(trainer, "adv_attacks_weights"):
            out[f"{strategy}_weights"] = np.array(trainer.adv_attacks_weights, dtype=np.float64)
    init = surrogate(7)
    for k, v in init.state_dict().items():
        out[f"init_model_{k}"] = npy(v)
    np.savez_compressed(OUT / "trainer.npz", **out)


def gen_metrics():
    from src.metrics import calculate_eer
    from sklearn.metrics import precision_recall_fscore_support, roc_auc_score

    out = {}
    rng = np.random.default_rng(61)
    for tag, N, sep in (("random", 1024, 0.0), ("separable", 1024, 1.5), ("tiny", 16, 0.7)):
        y = rng.integers(0, 2, size=N).astype(np.int64)
        score = 1.0 / (1.0 + np.exp(-(rng.standard_normal(N) + sep * (2 * y - 1))))
        score = score.astype(np.float32)
        label = (score + 0.5).astype(np.int32)  # evaluate_models_on_adversarial_attacks.py:238
        thresh, eer, fpr, tpr = calculate_eer(y=1 - y, y_score=score)  # :285-290
        p, r, f1, _ = precision_recall_fscore_support(y, label, average="binary", beta=1.0)  # :272-277
        auc = roc_auc_score(y_true=y, y_score=score)  # :278
        acc = 100.0 * float((label == y).sum()) / N
        out.update({f"{tag}_y": y, f"{tag}_score": score, f"{tag}_eer": np.float64(eer), f"{tag}_thresh": np.float64(thresh),
                    f"{tag}_precision": np.float64(p), f"{tag}_recall": np.float64(r), f"{tag}_f1": np.float64(f1),
                    f"{tag}_auc": np.float64(auc), f"{tag}_accuracy": np.float64(acc)})
    np.savez_compressed(OUT / "metrics.npz", **out)


_STAND_IN_NAMES = ("torchaudio", "torchaudio.functional", "soundfile", "asteroid_filterbanks")


@contextlib.contextmanager
def _stand_ins():
    """The inert module objects a generator puts into sys.modules live only as long as that generator: left behind, a
    spec-less `torchaudio` makes a later `import transformers.audio_utils` (gen_frontends_xcheck) fail with
    `ValueError: torchaudio.__spec__ is None` (What's weak 9).  Reference modules imported meanwhile stay
    cached; they hold their own references to the stand-ins."""
    before = {k: sys.modules.get(k) for k in _STAND_IN_NAMES}
    try:
        yield
    finally:
        for k, v in before.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _inert_torchaudio():
    """An object that satisfies `import torchaudio` + the three module-level constructor calls of
    src/frontends.py:13-38 and does nothing else (calling the result raises)."""
    class _Inert:
        def __init__(self, *a, **k):
            pass

        def to(self, *_):
            return self

        def __call__(self, *a, **k):
            raise RuntimeError("inert torchaudio placeholder: the frontends are not part of the golden fixtures")

    ta = types.ModuleType("torchaudio")
    ta.transforms = types.SimpleNamespace(MFCC=_Inert, LFCC=_Inert, MelScale=_Inert)
    return ta


def gen_model_bodies():
    sys.modules.setdefault("torchaudio", _inert_torchaudio())
    from src.models import lcnn as ref_lcnn
    from src.models import specrnet as ref_specrnet

    g = torch.Generator().manual_seed(71)
    torch.manual_seed(72)
    body = ref_lcnn.BaseLCNN(input_channels=1, num_coefficients=80).eval()
    spec = torch.randn(2, 1, 80, 404, generator=g) * 10.0
    with torch.no_grad():
        logits = body(spec)
    out = {f"sd_{k}": npy(v) for k, v in body.state_dict().items()}
    out.update({"spec": npy(spec), "logits": npy(logits)})
    # input-gradient of the train-mode-with-BN/Dropout-eval configuration the attacks use (attack.py:311-319)
    body.train()
    for m in body.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    s = spec.clone().requires_grad_(True)
    o = body(s)
    out["logits_attackmode"] = npy(o)
    out["grad_spec"] = npy(torch.autograd.grad(o.sum(), s)[0])
    np.savez_compressed(OUT / "lcnn_body.npz", **out)

    torch.manual_seed(73)
    body = ref_specrnet.BaseSpecRNet(ref_specrnet.get_config(2), device="cpu").eval()
    # non-trivial BN statistics so eval-mode BN is exercised
    with torch.no_grad():
        for m in body.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2, generator=g)
                m.running_var.uniform_(0.5, 1.5, generator=g)
    spec = torch.randn(2, 2, 80, 404, generator=g)
    with torch.no_grad():
        logits = body(spec)
    out = {f"sd_{k}": npy(v) for k, v in body.state_dict().items()}
    out.update({"spec": npy(spec), "logits": npy(logits)})
    # attack mode (attack.py:311-319): train() with BatchNorm in eval — the GRU then runs its training-mode code path
    body.train()
    for m in body.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    s = spec.clone().requires_grad_(True)
    o = body(s)
    out["logits_attackmode"] = npy(o)
    out["grad_spec"] = npy(torch.autograd.grad(o.sum(), s)[0])
    np.savez_compressed(OUT / "specrnet_body.npz", **out)


def gen_rawnet3_body():
    """Everything the reference's RawNet3 computes after its first layer (src/models/rawnet3.py:81-137, 161-274).
    `asteroid_filterbanks` (absent here) is replaced by tests/helpers.FixedEncoder, which returns a recorded tensor in
    place of the sinc encoder's output: the recorded logits / gradients are functions of that tensor only, and the sinc
    encoder itself is NOT pinned.  Full-size model (C = 1024); weights by the seeded recipe of helpers."""
    import json

    sys.path.insert(0, str(HERE.parent))
    import helpers

    afb = types.ModuleType("asteroid_filterbanks")
    afb.Encoder = helpers.FixedEncoder
    afb.ParamSincFB = lambda *a, **k: None
    sys.modules["asteroid_filterbanks"] = afb
    sys.modules.pop("src.models.rawnet3", None)
    from src.models import rawnet3 as ref_rawnet3

    model = helpers.rawnet3_fixture_weights(ref_rawnet3.prepare_model).eval()
    g = torch.Generator().manual_seed(83)
    frames = 900                                        # 900 -> pool 5 -> 180 -> pool 3 -> 60 (rawnet3.py:35-41)
    h = torch.randn(2, 256, frames, generator=g) * 0.3
    x = torch.rand(2, 16_000, generator=g)              # goes through preprocess only; conv1 ignores it
    model.conv1.h = h
    with torch.no_grad():
        logits = model(x)
    out = {"h": npy(h), "x": npy(x), "logits": npy(logits),
           "digests": np.array(json.dumps(helpers.tensor_digests(model.state_dict())))}
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    hh = h.clone().requires_grad_(True)
    model.conv1.h = hh
    o = model(x)
    out["logits_attackmode"] = npy(o)
    out["grad_h"] = npy(torch.autograd.grad(o.sum(), hh)[0])
    np.savez_compressed(OUT / "rawnet3_body.npz", **out)


def gen_frontends_xcheck():
    """Independent THIRD-PARTY arithmetic for the whole LFCC and mel-spectrogram outputs — NOT the reference: torchaudio
    0.10 (the package the reference calls, src/frontends.py:13-38) is neither in the reference tree nor installable, so
    the frontends stay PARITY UNPINNED.  What this fixture adds is that a wrong reading of the published algorithm can no
    longer pass every test: the restatement in frontends.py and the fused kernels are compared with code written by
    others (transformers.audio_utils: framing / window / FFT / triangular banks / dB; scipy.fft: the DCT).

      lfcc_<tag>  (80, frames): transformers `spectrogram` (periodic Hann 400, n_fft 512, hop 160, centred reflect pad,
                  power 2) -> linear triangular bank (transformers' triangular-bank helper on linspace edges, 257 -> 128)
                  -> `power_to_db(min 1e-10, db_range 80)` -> scipy ortho DCT-II, first 80 coefficients.  One utterance
                  at a time: the floor is then max - 80 dB of that utterance, which is what frontends.LFCC computes for
                  a batch of ONE.  (That torchaudio takes the maximum over the whole batch for a 3-D input is the
                  restatement's reading and cannot be cross-checked here.)
      mel_<tag>   (2, 80, frames): complex STFT with a rectangular 400-sample window — the reference calls torch.stft
                  itself for this (src/frontends.py:62-68), so the STFT here is numpy's FFT over torch.stft's framing
                  (window centred in the 512-point buffer; checked against torch.stft below) — then the HTK mel bank of
                  transformers `mel_filter_bank(norm=None)` on real and imaginary parts, magnitude and phase."""
    import scipy.fft
    from transformers import audio_utils as au

    out = {}
    window = au.window_function(400, "hann", periodic=True)
    fft_freqs = np.linspace(0, 8000, 257)
    linear_bank = au._create_triangular_filter_bank(fft_freqs, np.linspace(0.0, 8000.0, 130))       # (257, 128)
    mel_bank = au.mel_filter_bank(257, 80, 0.0, 8000.0, 16_000, norm=None, mel_scale="htk")          # (257, 80)
    cases = {"full": waveforms(2, T_FULL, 91), "short": waveforms(1, 8_000, 92), "loud": waveforms(1, 16_160, 93) * 8.0}
    cases["loud"][0, 5_000:9_000] *= 1e-4     # a near-silent stretch: the 80 dB floor is active for this utterance
    for tag, batch in cases.items():
        out[f"x_{tag}"] = npy(batch)
        lf, mel = [], []
        for row in batch.numpy():
            power = au.spectrogram(row, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=True,
                                   pad_mode="reflect", onesided=True)                                # (257, frames)
            bands = linear_bank.T @ power
            db = au.power_to_db(bands, reference=1.0, min_value=1e-10, db_range=80.0)
            lf.append(scipy.fft.dct(db, type=2, norm="ortho", axis=0)[:80])
            # torch.stft framing: reflect pad n_fft // 2, frames of 512 every 160, the 400 ones centred (56 zeros each side)
            padded = np.pad(row.astype(np.float64), 256, mode="reflect")
            n_frames = 1 + (padded.size - 512) // 160
            frames = np.stack([padded[t * 160:t * 160 + 512] for t in range(n_frames)])
            frames[:, :56] = 0.0
            frames[:, 456:] = 0.0
            spec = np.fft.rfft(frames, axis=1).T                                                    # (257, frames)
            want = torch.stft(torch.from_numpy(row), n_fft=512, hop_length=160, win_length=400,
                              window=torch.ones(400), return_complex=True).numpy()
            assert np.abs(spec - want).max() <= 2e-4 * np.abs(want).max()
            m = mel_bank.T @ spec.real + 1j * (mel_bank.T @ spec.imag)
            mel.append(np.stack([np.abs(m), np.angle(m)]))
        out[f"lfcc_{tag}"] = np.stack(lf).astype(np.float32)
        out[f"mel_{tag}"] = np.stack(mel).astype(np.float32)
    np.savez_compressed(OUT / "frontends_xcheck.npz", **out)


def gen_attack_save(ta, aa_utils):
    """Attack.save (adversarial_attacks/torchattacks/attack.py:149-233) run by the REFERENCE: FGSM over a two-batch loader with
    the surrogate detector, `save_pred=True`, once with return type 'float' and once 'int' — the saved (adversarials, labels,
    predictions) tuples and the returned (robust accuracy, mean L2 of the not-right rows)."""
    import tempfile
    out = {}
    model = surrogate(91)
    out.update({f"model_{k}": npy(v) for k, v in model.state_dict().items()})
    x01, _, _ = aa_utils.to_minmax(waveforms(4, T_SMALL, 92))
    y = torch.tensor([0, 0, 0, 1])      # `pred` is max over the single logit's dim: always 0 (kept as the reference computes it)
    out["x"], out["y"] = npy(x01), npy(y)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x01, y), batch_size=2, shuffle=False)
    for kind in ("float", "int"):
        atk = ta.FGSM(model, eps=0.001)
        atk.set_training_mode(model_training=True, batchnorm_training=False)
        atk.set_return_type(kind)
        with tempfile.TemporaryDirectory() as tmp:
            path = Path(tmp) / "adv.pt"
            rob_acc, l2, _ = atk.save(loader, save_path=str(path), verbose=False, return_verbose=True, save_pred=True)
            adv, labels, preds = torch.load(path)
        out[f"{kind}_adv"], out[f"{kind}_labels"], out[f"{kind}_preds"] = npy(adv), npy(labels), npy(preds)
        out[f"{kind}_rob_acc"], out[f"{kind}_l2"] = np.float64(rob_acc), np.float64(l2)
        out[f"{kind}_return_type_after"] = np.array(atk._return_type)    # the reference leaves 'float' behind (attack.py:175)
    np.savez_compressed(OUT / "attack_save.npz", **out)


def gen_frontends_batch_floor():
    """The one reading of torchaudio 0.10's LFCC that gen_frontends_xcheck cannot see: `amplitude_to_DB(top_db=80)` on
    the 3-D (B, n_filter, time) tensor LFCC hands it takes its floor from the maximum over the WHOLE BATCH
    (src/frontends.py:24-32 -> torchaudio LFCC.forward; SURVEY.md section 7 "Unpinned third-party arithmetic").  Still not
    the reference — third-party pieces (transformers.audio_utils framing / window / FFT / triangular bank, scipy's DCT) with
    the batch-wide floor applied BY HAND between them:

      x          (3, 16160): a loud (x 6), a quiet (x 0.002) and a near-silent (x 3e-4) utterance
      lfcc_batch (3, 80, frames): dB = 10 log10(max(bands, 1e-10)); floor = max over all three utterances - 80;
                  dB = max(dB, floor); DCT.  The quiet utterances sit largely BELOW the loud one's floor.
      lfcc_each  (3, 80, frames): the same with the floor taken per utterance (what processing one utterance at a time, or
                  a different sharding of the batch, would give) — must differ from lfcc_batch for rows 1 and 2."""
    import scipy.fft
    from transformers import audio_utils as au

    window = au.window_function(400, "hann", periodic=True)
    linear_bank = au._create_triangular_filter_bank(np.linspace(0, 8000, 257), np.linspace(0.0, 8000.0, 130))   # (257, 128)
    x = waveforms(3, 16_160, 95)
    x[0] *= 6.0
    x[1] *= 0.002
    x[2] *= 3e-4
    db = []
    for row in x.numpy():
        power = au.spectrogram(row, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=True,
                               pad_mode="reflect", onesided=True)
        db.append(10.0 * np.log10(np.maximum(linear_bank.T @ power, 1e-10)))
    db = np.stack(db)                                                        # (3, 128, frames), float64
    batch = np.maximum(db, db.max() - 80.0)
    each = np.maximum(db, db.max(axis=(1, 2), keepdims=True) - 80.0)
    assert (batch[1] != each[1]).mean() > 0.2 and (batch[2] != each[2]).mean() > 0.9 and np.array_equal(batch[0], each[0])
    dct = lambda a: scipy.fft.dct(a, type=2, norm="ortho", axis=1)[:, :80]
    np.savez_compressed(OUT / "frontends_batch_floor.npz", x=npy(x), lfcc_batch=dct(batch).astype(np.float32),
                        lfcc_each=dct(each).astype(np.float32),
                        floored_share=np.array([(db[k] < db.max() - 80.0).mean() for k in range(3)]))


def gen_datasets():
    """PadDataset.apply_pad, wavefake_preprocessing_on_batch (SoX steps off), the corpus
    listings of DetectionDataset on the miniature corpora of tests/helpers.build_corpus_trees, and AttackAnalyser's
    files.  `src.datasets.base_dataset` imports soundfile and torchaudio (absent): inert module objects stand in for
    them — none of the recorded functions calls into either."""
    import hashlib
    import json
    import tempfile

    sys.path.insert(0, str(HERE.parent))
    import helpers

    ta = sys.modules.get("torchaudio") or _inert_torchaudio()
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.functional.apply_codec = None
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = ta.functional
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    if "asteroid_filterbanks" not in sys.modules:  # src.utils -> src.models.rawnet3 imports it; nothing recorded uses it
        afb = types.ModuleType("asteroid_filterbanks")
        afb.Encoder = afb.ParamSincFB = None
        sys.modules["asteroid_filterbanks"] = afb
    import typing
    import torch.utils.data.dataset as tud
    if not hasattr(tud, "T_co"):  # a type-annotation name torch 2.10 no longer exports (base_dataset.py:14,150)
        tud.T_co = typing.TypeVar("T_co", covariant=True)
    from src.datasets import base_dataset as ref_base
    from src.datasets.detection_dataset import DetectionDataset as RefDetectionDataset
    from src.aa.qualitative.attacks_analysis import AttackAnalyser as RefAnalyser

    out = {}
    # --- apply_pad: (length, cut) pairs incl. exact multiples, one sample, longer than the cut
    g = torch.Generator().manual_seed(81)
    cases = [(1, 10), (7, 64), (64, 64), (65, 64), (100, 4099), (4099, 4099), (32300, 64600), (64599, 64600),
             (70000, 64600), (21533, 64600)]
    out["pad_cases"] = np.array(cases, dtype=np.int64)
    for k, (n, cut) in enumerate(cases):
        w = (torch.arange(n).remainder(997).float() / 997.0 - 0.5).unsqueeze(0)  # a ramp: compresses well
        out[f"pad_in_{k}"] = npy(w)
        out[f"pad_out_{k}"] = npy(ref_base.PadDataset.apply_pad(w, cut))
    # --- wavefake_preprocessing_on_batch with every SoX step off
    batch = torch.randn(5, 1000, generator=g)
    rates = torch.full((5,), 16_000, dtype=torch.int64)
    for name, cut in (("short", 2600), ("long", 640)):
        got, got_rates = ref_base.SimpleAudioFakeDataset.wavefake_preprocessing_on_batch(
            batch, rates, wave_fake_trim=False, wave_fake_cut=cut)
        out[f"onbatch_{name}_out"], out[f"onbatch_{name}_rates"] = npy(got), npy(got_rates)
    out["onbatch_in"] = npy(batch)
    # --- AttackAnalyser
    B, T = 14, 800
    x = torch.randn(B, T, generator=g) * 0.1
    xa = x + torch.randn(B, T, generator=g) * 0.01
    y = torch.tensor([0, 0, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1])
    clean = torch.tensor([0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0], dtype=torch.int32)
    attacked = torch.tensor([1, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 1], dtype=torch.int32)
    paths = [f"/data/WaveFake/generated_audio/ljspeech_melgan/LJ001-{i:04d}_gen.wav" if i % 3 == 0 else
             f"/data/FakeAVCeleb_v1.2/FakeAVCeleb-audio/RealVideo-RealAudio/African/men/id{i:05d}/{i:05d}.mp3" if i % 3 == 1
             else f"/data/ASVspoof2021/DF/ASVspoof2021_DF_eval_part00/ASVspoof2021_DF_eval/flac/DF_E_{2000000 + i}.flac"
             for i in range(B)]
    seconds = torch.tensor([1.0 + 0.377 * i for i in range(B)], dtype=torch.float64)
    metadata = [["melgan"] * B, paths, ["val"] * B, seconds]  # what default_collate makes of B metadata tuples
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(None):
        RefAnalyser(tmp).analyse(batch_x=x, batch_x_attacked=xa, batch_y=y, batch_preds_label=attacked,
                                 batch_preds=torch.rand(B), batch_preds_noattack_label=clean,
                                 batch_preds_noattack=torch.rand(B), batch_metadata=metadata)
        files = {p.name: p.read_bytes() for p in sorted(Path(tmp).iterdir())}
    out.update({"qual_x": npy(x), "qual_xa": npy(xa), "qual_y": npy(y), "qual_clean": npy(clean),
                "qual_attacked": npy(attacked), "qual_seconds": npy(seconds)})
    first = sorted(files)[0]
    out["qual_first_file"] = np.frombuffer(files[first], dtype=np.uint8)
    np.savez_compressed(OUT / "datasets.npz", **out)

    # --- corpus listings
    listing = {"qual_paths": paths, "qual_files": {n: hashlib.sha256(b).hexdigest() for n, b in files.items()},
               "qual_first_file": first, "listings": {}}
    # FakeAVCelebDataset.get_fake_samples hands split_samples a list of (index, Series) pairs, and base_dataset.py:66
    # passes it to np.split, which numpy <= 1.23 (the reference's pin) turned into an (n, 2) object array cut along axis
    # 0.  numpy >= 1.24 refuses to build that ragged array, so for such a list the cut is made on the list itself with
    # the same indices — the rows numpy <= 1.23 returned.
    real_split = np.split

    def split_lists_too(ary, indices, axis=0):
        if isinstance(ary, list) and ary and isinstance(ary[0], tuple):
            bounds = [0, *indices, len(ary)]
            return [ary[lo:hi] for lo, hi in zip(bounds[:-1], bounds[1:])]
        return real_split(ary, indices, axis)

    ref_base.np.split = split_lists_too
    with tempfile.TemporaryDirectory() as tmp:
        roots = helpers.build_corpus_trees(tmp)
        for subset in ("train", "test", "val"):
            for name, kw in (("plain", dict(oversample=False)), ("oversample", dict(oversample=True)),
                             ("undersample", dict(oversample=False, undersample=True)),
                             ("reduced", dict(oversample=True, reduced_number=17))):
                np.random.seed(5)  # oversampling draws from numpy's global RNG
                ds = RefDetectionDataset(subset=subset, **{k: str(v) for k, v in roots.items()}, **kw)
                listing["listings"][f"{subset}/{name}"] = helpers.listing_of(ds.samples, tmp)
    ref_base.np.split = real_split
    (OUT / "datasets_listing.json").write_text(json.dumps(listing, indent=0))


def main(out_dir=None):
    global OUT
    if out_dir is not None:
        OUT = Path(out_dir)
        OUT.mkdir(parents=True, exist_ok=True)
    _import_reference()
    torch.set_num_threads(1)  # fixed thread count: the reference is bit-reproducible at a fixed thread count
    torch.use_deterministic_algorithms(True)
    from adversarial_attacks import torchattacks as ta
    from src.aa import utils as aa_utils

    gen_minmax(aa_utils)
    gen_fgsm(ta, aa_utils)
    gen_pgd_linf(ta, aa_utils)
    gen_pgd_l2(ta, aa_utils)
    gen_cw(ta, aa_utils)
    gen_fab_projections()
    gen_fab_attack(ta, aa_utils)
    gen_trainer()
    gen_attack_save(ta, aa_utils)
    gen_metrics()
    with _stand_ins():
        gen_model_bodies()
    with _stand_ins():
        gen_rawnet3_body()
    gen_frontends_xcheck()
    gen_frontends_batch_floor()
    with _stand_ins():
        gen_datasets()
    for p in sorted(OUT.glob("*.npz")):
        print(f"{p.name}: {p.stat().st_size / 1e6:.2f} MB")


if __name__ == "__main__":
    if sys.argv[1:] == ["frontends_batch_floor"]:      # third-party code only: does not need the reference tree
        gen_frontends_batch_floor()
    elif sys.argv[1:] == ["attack_save"]:              # one fixture, same set-up as main()
        _import_reference()
        torch.set_num_threads(1)
        from adversarial_attacks import torchattacks as ta
        from src.aa import utils as aa_utils
        gen_attack_save(ta, aa_utils)
    elif sys.argv[1:2] == ["--out"]:                   # the whole recipe into another directory (nothing under tests/ is touched)
        main(sys.argv[2])
    else:
        main()
