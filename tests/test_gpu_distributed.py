"""BASELINE.json configs[4] rehearsed on ONE device: two and EIGHT ranks (one process each, gloo — RCCL refuses two ranks on the
same GPU) run the shipped sharded loop, and `bench.py --gpus 2 / 8` launches its own ranks.  What runs here is everything
of the 8-GPU path except the RCCL transport itself: rank discovery, contiguous shards of every global batch, per-rank
models / streams / workspaces, the end-of-run all-reduce + all-gather, rank 0's report (SURVEY.md section 8-e;
replaces nn.DataParallel, reference evaluate_models_on_adversarial_attacks.py:163,167)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
SEED = 42


def _sizes(world):
    """(global batch, items): 8 utterances per rank and batch, 2 complete global batches; 8 utterances are dropped (drop_last)."""
    return 8 * world, 16 * world + 8


GLOBAL_BATCH, N_ITEMS = _sizes(2)
ATTACK = {"eps": 0.003, "steps": 3, "random_start": False}   # deterministic: the shard replay needs no RNG agreement


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _config():
    import yaml
    return yaml.safe_load((ROOT / "configs" / "aa_evaluation" / "lcnn.yaml").read_text())


def _evaluate(dataset, batch_size, shuffle, device="cuda:0"):
    from audio_deepfake_adversarial_attacks_amd import torchattacks
    from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    cfg = _config()
    set_seed(SEED)                                     # equal replicas on every rank and in the replay
    return generate_attacks([None, None, None], cfg, device, attack_model_config=cfg, attack_method=torchattacks.PGD,
                            attack_params=dict(ATTACK), batch_size=batch_size, dataset=dataset, share_weights=True,
                            shuffle=shuffle, num_workers=0, return_scores=True)


def _worker(rank, world, port, out_dir, backend="gloo"):
    import torch.distributed as dist
    from torch.utils.data import Subset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
    from audio_deepfake_adversarial_attacks_amd.evaluation import ShardedBatchSampler
    # nccl: one rank per device (RCCL over xGMI) — the rank's device is addressed by its index (the visibility variables cannot
    # be changed once the HIP runtime is up in this process); gloo: every rank shares cuda:0
    device = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(torch.device(device))
    n_batch, n_items = _sizes(world)
    data = SyntheticDetectionDataset(n_items)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = _evaluate(data, n_batch, shuffle=True, device=device)
    finally:
        dist.destroy_process_group()
    # the same rows through the single-process loop: this rank's contiguous slices, batch = the shard size, so the
    # LFCC batch-wide dB floor sees the same rows as in the sharded run
    mine = sum(ShardedBatchSampler(n_items, n_batch, rank, world, shuffle=True, seed=SEED), [])
    alone = _evaluate(Subset(data, mine), n_batch // world, shuffle=False, device=device)
    np.savez(Path(out_dir) / f"rank{rank}.npz", mine=np.array(mine),
             sharded_pred=sharded["scores"]["y_pred"], sharded_label=sharded["scores"]["y_pred_label"],
             sharded_y=sharded["scores"]["y"], alone_pred=alone["scores"]["y_pred"],
             alone_label=alone["scores"]["y_pred_label"], alone_y=alone["scores"]["y"],
             report=json.dumps({k: v for k, v in sharded.items() if k != "scores"}))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 2), ("gloo", 8)])
def test_ranks_on_one_gpu_equal_single_process_shards(cuda, tmp_path, backend, world):
    """backend = "gloo": `world` ranks sharing the one device of this pool's boxes — 2, and 8 = BASELINE.json configs[4]'s rank
    count: launcher-free rendezvous, `ShardedBatchSampler`'s eight contiguous shards and `aggregate_across_ranks` run at the real
    world size before they ever meet an 8-GPU node.  backend = "nccl": the same comparison with one rank per DEVICE over RCCL
    — what the 8-GPU job runs; skipped where fewer than two devices are visible (every box of this pool)."""
    import torch.multiprocessing as mp
    from audio_deepfake_adversarial_attacks_amd import metrics
    from audio_deepfake_adversarial_attacks_amd.evaluation import shard_bounds
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip("needs two HIP devices (RCCL refuses two ranks on one device)")
    n_batch, n_items = _sizes(world)
    try:
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    except Exception as exc:       # keep the workers' traceback where a truncated console log cannot lose it
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / f"distributed_{backend}_{world}_failure.txt").write_text(repr(exc) + "\n" + str(exc))
        raise
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    per_rank = (n_items // n_batch) * n_batch // world
    seen = set()
    for k in range(world):                                  # `world` disjoint shards of equal size ...
        assert len(r[k]["mine"]) == per_rank and not seen & set(r[k]["mine"].tolist())
        seen |= set(r[k]["mine"].tolist())
    # ... each of them rank k's CONTIGUOUS slice of every global batch (DataParallel's scatter, reference :163)
    order = torch.randperm(n_items, generator=torch.Generator().manual_seed(SEED)).tolist()
    for k in range(world):
        lo, hi = shard_bounds(n_batch, k, world)
        want = sum((order[b * n_batch + lo:b * n_batch + hi] for b in range(n_items // n_batch)), [])
        assert r[k]["mine"].tolist() == want, f"rank {k} shard"
    # every rank ends with the same complete table and the same report
    for key in ("sharded_pred", "sharded_label", "sharded_y", "report"):
        for k in range(1, world):
            assert np.array_equal(r[0][key], r[k][key]), (key, k)
    # rank k's rows of the gathered table == a single-process run over rank k's shard, bit for bit
    for k in range(world):
        rows = slice(k * per_rank, (k + 1) * per_rank)
        assert np.array_equal(r[0]["sharded_pred"][rows], r[k]["alone_pred"]), f"rank {k} scores"
        assert np.array_equal(r[0]["sharded_label"][rows], r[k]["alone_label"])
        assert np.array_equal(r[0]["sharded_y"][rows], r[k]["alone_y"])
    assert np.unique(r[0]["sharded_pred"]).size > per_rank      # not a constant detector: the comparison has teeth
    # the gathered report == the report over the concatenated single-process shard runs
    pred = np.concatenate([r[k]["alone_pred"] for k in range(world)])
    label = np.concatenate([r[k]["alone_label"] for k in range(world)])
    y = np.concatenate([r[k]["alone_y"] for k in range(world)])
    want = metrics.adversarial_report(y, pred, label)
    got = json.loads(str(r[0]["report"]))
    for key, value in want.items():
        if key == "adv_eval/accuracy":
            assert got[key] == pytest.approx(100.0 * float((label == y).mean()), abs=1e-12)
        else:
            assert got[key] == pytest.approx(value, abs=1e-12), key
    assert got["num_total"] == world * per_rank


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks(cuda, world):
    """`python bench.py --gpus N` with no launcher in the environment must start N ranks itself and print ONE JSON
    line (rehearsed on one device: --share-device --backend gloo; on a >= N-GPU node the defaults use RCCL).  N = 8 is
    BASELINE.json configs[4]'s rank count (B = 8 per rank here instead of 128)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1", "--batch", "8",
           "--share-device", "--backend", "gloo"]
    proc = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=1400)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["global_batch"] == 8 * world and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(8 * world / (line["ms_per_step"] * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in line and line["roofline"]["launches_timed"] == 40
    assert f"{world} independent contiguous shards" in line["config"]["sharding"]


# ---- RCCL itself, on the one device this pool has ---------------------------------------------------------

_RCCL_SCRIPT = r"""
import json, os, sys, warnings
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ADVSTEP_REPO"])
import yaml
from audio_deepfake_adversarial_attacks_amd import torchattacks
from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
from audio_deepfake_adversarial_attacks_amd.evaluation import aggregate_across_ranks, generate_attacks
from audio_deepfake_adversarial_attacks_amd.torchattacks import graphed
from audio_deepfake_adversarial_attacks_amd.utils import set_seed

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = yaml.safe_load(open(os.path.join(os.environ["ADVSTEP_REPO"], "configs", "aa_evaluation", "lcnn.yaml")).read())
data = SyntheticDetectionDataset(32)
attack = {"eps": 0.003, "steps": 6, "random_start": False}


def evaluate():
    set_seed(42)
    graphed.clear()
    return generate_attacks([None, None, None], cfg, "cuda:0", attack_model_config=cfg, attack_method=torchattacks.PGD,
                            attack_params=dict(attack), batch_size=8, dataset=data, share_weights=True, shuffle=True,
                            num_workers=0, return_scores=True)


def table(n, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, generator=g).to(dev)
    y = torch.randint(0, 2, (n,), generator=g).to(dev)
    lab = (p + 0.5).int()
    return p, lab, y, (lab == y.int()).sum(), torch.tensor(n, device=dev)


plain_table = aggregate_across_ranks(*table(50, 3))
plain = evaluate()                                             # no process group: the non-distributed path
plain_graphs = len(graphed._GRAPHS)

dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # RCCL communicator on cuda:0
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    t = torch.arange(4, dtype=torch.float32, device=dev)
    dist.all_reduce(t)                                          # the collective kernels really run
    parts = [torch.empty_like(t)]
    dist.all_gather(parts, t)
    dist.barrier()
    rccl_table = aggregate_across_ranks(*table(50, 3))
    sharded = evaluate()                                        # generate_attacks under the process group, hipGraph path on
    rccl_graphs = len(graphed._GRAPHS)
dist.destroy_process_group()
out = {"all_reduce": t.tolist(), "all_gather": parts[0].tolist(), "graphs_plain": plain_graphs, "graphs_rccl": rccl_graphs,
       "capture_warnings": [str(w.message) for w in caught if "capture" in str(w.message)],
       "tables_equal": all(np.array_equal(a, b) for a, b in zip(plain_table, rccl_table)),
       "scores_equal": all(np.array_equal(plain["scores"][k], sharded["scores"][k]) for k in plain["scores"]),
       "report_plain": {k: v for k, v in plain.items() if k != "scores"},
       "report_rccl": {k: v for k, v in sharded.items() if k != "scores"},
       "backend": dist.Backend.NCCL, "nccl_version": list(torch.cuda.nccl.version())}
print("RESULT " + json.dumps(out), flush=True)
"""


@pytest.mark.timeout(900)
def test_rccl_process_group_on_one_device(cuda, tmp_path):
    """`backend="nccl"` (= RCCL) with ONE rank on cuda:0: communicator set-up, all_reduce / all_gather / barrier on device
    tensors, `aggregate_across_ranks` through it (same table as without a process group), and the shipped
    `generate_attacks` loop under it with the hipGraph replay on — the capture must succeed while RCCL's watchdog thread
    is alive (graphed.py captures in thread_local error mode for exactly that) and the scores must equal the
    non-distributed run bit for bit.  What this does NOT cover is ranks on distinct devices (xGMI transport): this pool
    has single-GPU boxes."""
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(ADVSTEP_REPO=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=800)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-3000:])
    res = json.loads([l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["all_reduce"] == [0.0, 1.0, 2.0, 3.0] and res["all_gather"] == [0.0, 1.0, 2.0, 3.0]
    assert res["tables_equal"] and res["scores_equal"], res
    assert res["report_plain"] == res["report_rccl"] and res["report_rccl"]["num_total"] == 32
    assert res["graphs_plain"] >= 1 and res["graphs_rccl"] >= 1, res          # 4 batches: captured at the second
    assert not res["capture_warnings"], res
