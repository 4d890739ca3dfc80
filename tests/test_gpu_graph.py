"""`-m gpu`: hipGraph replay of the PGD / PGDL2 inner loop (torchattacks/graphed.py) against the eager launches — same
kernels, same order, so the adversarial waveforms must be bit-identical; stale-graph hazards (changed weights, changed
train/eval flags, active launch profiling) must fall back to eager."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def armed(cls, model, **kw):
    atk = cls(model, **kw)
    atk.set_training_mode(model_training=True, batchnorm_training=False)
    return atk


@pytest.fixture()
def fresh_graphs():
    from audio_deepfake_adversarial_attacks_amd.torchattacks import graphed
    graphed.clear()
    yield graphed
    graphed.clear()


def data(cuda, n, seed):
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.aa import utils as aa_utils
    x, y = synthetic_waveforms(n, seed=seed)
    x01, _, _ = aa_utils.to_minmax(x.to(cuda))
    return x01, y.to(cuda)


def test_pgd_on_lcnn_graph_replay_is_bit_identical(cuda, fresh_graphs, monkeypatch):
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda).eval()
    atk = armed(torchattacks.PGD, model, eps=0.003, steps=7, random_start=False)     # odd: 3 replays + 1 eager iteration
    x01, y = data(cuda, 4, 11)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "0")
    want = atk(x01, y)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "1")
    first = atk(x01, y)                      # key seen once: eager
    assert len(fresh_graphs._GRAPHS) == 0
    second = atk(x01, y)                     # key seen twice: captured, replayed
    assert len(fresh_graphs._GRAPHS) == 1
    third = atk(x01, y)                      # replay of the cached graph
    for got in (first, second, third):
        assert torch.equal(got, want)
    # another batch of the same shape through the same graph
    x2, y2 = data(cuda, 4, 12)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "0")
    want2 = atk(x2, y2)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "1")
    assert torch.equal(atk(x2, y2), want2) and len(fresh_graphs._GRAPHS) == 1
    # random starts are drawn outside the graph: two calls differ, both stay in the eps-ball
    rnd = armed(torchattacks.PGD, model, eps=0.003, steps=4, random_start=True)
    a, b, c = rnd(x01, y), rnd(x01, y), rnd(x01, y)
    assert not torch.equal(b, c) and (c - x01).abs().max().item() <= 0.003 + 1e-7

    # changed weights -> changed key -> no stale prepared weights: equal to a pure eager run on the new weights
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "0")
    want3 = atk(x01, y)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "1")
    stale = set(fresh_graphs._GRAPHS)
    assert len(stale) == 1
    assert torch.equal(atk(x01, y), want3) and set(fresh_graphs._GRAPHS) == stale         # first sight: eager
    # second sight: a new graph — and the capture of the same workload under the OLD weights, which can never be replayed
    # again, is dropped with its memory pool (an adversarial-training run would otherwise pile them up)
    assert torch.equal(atk(x01, y), want3)
    assert len(fresh_graphs._GRAPHS) == 1 and not (set(fresh_graphs._GRAPHS) & stale)
    assert torch.equal(atk(x01, y), want3)                                                # replay of the new graph
    assert not torch.equal(want3, want)


def test_profiling_keeps_the_loop_eager(cuda, fresh_graphs):
    from audio_deepfake_adversarial_attacks_amd import hip_ops, torchattacks
    from tests.helpers import Surrogate
    torch.manual_seed(1)
    model = Surrogate().to(cuda).eval()
    atk = armed(torchattacks.PGD, model, eps=0.01, steps=4, random_start=False)
    x01, y = data(cuda, 3, 5)
    hip_ops.start_profile("pgd_linf_step")
    for _ in range(3):
        atk(x01, y)
    ms = hip_ops.stop_profile()
    assert len(ms["pgd_linf_step"]) == 12 and len(fresh_graphs._GRAPHS) == 0


def test_pgdl2_on_specrnet_graph_replay_is_bit_identical(cuda, fresh_graphs, monkeypatch):
    """The captured iteration contains MIOpen convolutions and library GEMMs next to the repository's kernels."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}, str(cuda)).to(cuda).eval()
    atk = armed(torchattacks.PGDL2, model, eps=0.1, steps=4, random_start=False)
    x01, y = data(cuda, 3, 21)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "0")
    want = atk(x01, y)
    monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", "1")
    outs = [atk(x01, y) for _ in range(3)]
    assert len(fresh_graphs._GRAPHS) == 1
    for got in outs:
        assert torch.equal(got, want)
