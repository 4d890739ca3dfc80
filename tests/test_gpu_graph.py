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


def test_split_form_equals_the_fused_graph_and_eager_launches(cuda, fresh_graphs, monkeypatch):
    """Round 6: the model part of an iteration replays from two graphs and the update step is launched between them
    (ADVSTEP_ATTACK_GRAPH=1, default); rounds 2-5's single graph of two whole iterations stays behind `fused`.  Same
    kernels in the same order on the same buffers: bit-identical to each other and to eager launches, for PGD (one step
    launch) and PGDL2 (single-pass step + repair node)."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda).eval()
    x01, y = data(cuda, 4, 31)
    for cls, kw in ((torchattacks.PGD, dict(eps=0.003, steps=6)), (torchattacks.PGDL2, dict(eps=0.1, steps=5))):
        atk = armed(cls, model, random_start=False, **kw)
        got = {}
        for mode in ("0", "1", "fused"):
            monkeypatch.setenv("ADVSTEP_ATTACK_GRAPH", mode)
            fresh_graphs.clear()
            outs = [atk(x01, y) for _ in range(3)]
            assert len(fresh_graphs._GRAPHS) == (0 if mode == "0" else 1)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
            got[mode] = outs[2]
        assert torch.equal(got["1"], got["0"]) and torch.equal(got["fused"], got["0"])


def test_step_brackets_go_with_graph_replay_when_asked_for(cuda, fresh_graphs):
    """bench.py's timed region: start_profile(..., graph_ok=True) prices launches that stay outside the captured model
    part, so the loop keeps replaying (plain start_profile keeps it eager: test_profiling_keeps_the_loop_eager)."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops, torchattacks
    from tests.helpers import Surrogate
    torch.manual_seed(1)
    model = Surrogate().to(cuda).eval()
    atk = armed(torchattacks.PGD, model, eps=0.01, steps=4, random_start=False)
    x01, y = data(cuda, 3, 5)
    want = atk(x01, y)                                   # eager (first sight)
    hip_ops.start_profile("pgd_linf_step", "ce2_loss_grad", graph_ok=True)
    outs = [atk(x01, y) for _ in range(3)]               # second sight: captured under profiling; then replays
    ms = hip_ops.stop_profile()
    assert len(fresh_graphs._GRAPHS) == 1
    for got in outs:
        assert torch.equal(got, want)
    # call 1: the warm-up's 2 eager iterations (4 brackets: 2 steps ... of the pair) + 2 replayed pairs x 2 steps; calls 2, 3: 4 each
    assert len(ms["pgd_linf_step"]) == 2 + 4 + 4 + 4 and all(v > 0 for v in ms["pgd_linf_step"])
    # the loss gradient sits inside the captured model part: bracketed only while launched eagerly (the warm-up)
    assert len(ms["ce2_loss_grad"]) == 2


def test_two_batches_in_flight_give_the_same_scores(cuda, fresh_graphs):
    """evaluation.generate_attacks with batch i on stream i % 2 (round 6) against one batch at a time: 8 batches, so both
    streams pass their serialised first two batches, capture, and then run side by side — every score bit-identical."""
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
    from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks
    cfg = {"data": {"seed": 42}, "checkpoint": {"path": ""},
           "model": {"name": "lcnn", "parameters": {"frontend_algorithm": ["lfcc"], "input_channels": 1}}}

    def evaluate(in_flight):
        torch.manual_seed(5)                             # the random starts' Philox keys come from the global generator
        fresh_graphs.clear()
        rep = generate_attacks([None, None, None], cfg, str(cuda), attack_model_config=cfg, attack_method=torchattacks.PGD,
                               attack_params={"eps": 0.003, "steps": 6}, batch_size=4, dataset=SyntheticDetectionDataset(32),
                               share_weights=True, shuffle=False, num_workers=0, return_scores=True, in_flight=in_flight)
        return rep, len(fresh_graphs._GRAPHS)

    one, g1 = evaluate(1)
    two, g2 = evaluate(2)
    assert (g1, g2) == (1, 2)                            # a capture per launch stream
    for k in ("y_pred", "y_pred_label", "y"):
        assert torch.equal(torch.as_tensor(one["scores"][k]), torch.as_tensor(two["scores"][k])), k
    assert one["adv_eval/accuracy"] == two["adv_eval/accuracy"] and one["num_total"] == two["num_total"] == 32
    # the default for a graph-replayed attack is two in flight
    default, g3 = evaluate(None)
    assert g3 == 2 and torch.equal(torch.as_tensor(default["scores"]["y_pred"]), torch.as_tensor(one["scores"]["y_pred"]))
