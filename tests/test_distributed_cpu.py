"""CPU, world_size = 2 over gloo: the only exchange of the hot path — the end-of-run aggregate — and the rank
sharding that feeds it.  On the GPU box the same code runs over RCCL (backend "nccl")."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, global_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_deepfake_adversarial_attacks_amd import metrics
        from audio_deepfake_adversarial_attacks_amd.evaluation import (ShardedBatchSampler, aggregate_across_ranks,
                                                                        rank_and_world)
        assert rank_and_world() == (rank, world)
        gen = torch.Generator().manual_seed(77)               # every rank derives the same global score table
        score = torch.rand(n_total, generator=gen)
        y = torch.randint(0, 2, (n_total,), generator=gen)
        mine = sum(ShardedBatchSampler(n_total, global_batch, rank, world, shuffle=True, seed=5), [])
        idx = torch.tensor(mine)
        pred, label, truth = score[idx], (score[idx] + 0.5).int(), y[idx]
        correct = (label == truth.int()).sum()
        total = torch.tensor(len(mine))
        all_pred, all_label, all_y, n_correct, n_total_seen = aggregate_across_ranks(pred, label, truth, correct, total)
        report = metrics.adversarial_report(all_y, all_pred, all_label)
        np.savez(Path(out_dir) / f"rank{rank}.npz", pred=all_pred, label=all_label, y=all_y, n_correct=n_correct,
                 n_total=n_total_seen, eer=report["adv_eval/eer"], mine=np.array(mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_aggregate_equals_single_process(tmp_path):
    world, n_total, global_batch = 2, 200, 32
    mp.spawn(_worker, args=(world, _free_port(), n_total, global_batch, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank ends with the same, complete table
    for k in ("pred", "label", "y", "n_correct", "n_total", "eer"):
        assert np.array_equal(r0[k], r1[k]), k
    used = (n_total // global_batch) * global_batch
    assert int(r0["n_total"]) == used and len(r0["pred"]) == used
    assert not set(r0["mine"]) & set(r1["mine"]) and len(r0["mine"]) == len(r1["mine"]) == used // 2

    # single-process truth on the same rows
    from audio_deepfake_adversarial_attacks_amd import metrics
    gen = torch.Generator().manual_seed(77)
    score = torch.rand(n_total, generator=gen).numpy()
    y = torch.randint(0, 2, (n_total,), generator=gen).numpy()
    rows = np.concatenate([r0["mine"], r1["mine"]])          # all_gather concatenates in rank order
    assert np.array_equal(r0["pred"], score[rows]) and np.array_equal(r0["y"], y[rows])
    want = metrics.adversarial_report(y[rows], score[rows], (score[rows] + 0.5).astype(np.int32))
    assert abs(float(r0["eer"]) - want["adv_eval/eer"]) < 1e-12
    assert int(r0["n_correct"]) == int(((score[rows] + 0.5).astype(np.int32) == y[rows]).sum())


# ---- adversarial training, 2 ranks: DistributedDataParallel over gloo (RCCL on the GPU box) ---------------------------------

def _train_worker(rank, world, port, out_dir):
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_deepfake_adversarial_attacks_amd import trainer as T
        from audio_deepfake_adversarial_attacks_amd.aa.aa_trainer_types import AdversarialGDTrainerEnum
        from oracle import torch_ops
        from tests.helpers import Surrogate, TinyDetectionSet
        torch.manual_seed(11)                      # same initial weights on every rank (DDP would broadcast rank 0's anyway)
        model = Surrogate().train()
        random.seed(3)
        torch.manual_seed(3)
        seen = []
        tr = AdversarialGDTrainerEnum.ADAPTIVE.value(epochs=2, batch_size=8, device="cpu", optimizer_kwargs={"lr": 1e-2})
        tr.attack_ops = torch_ops
        orig = tr.apply_adv_attack
        tr.apply_adv_attack = lambda bx, by: (seen.append((bx.shape[0], tr.last_adv_attack)), orig(bx, by))[1]
        out = tr.train(dataset=TinyDetectionSet(32, 1024, 21), model=model, attack_model=model,
                       adversarial_attacks=["FGSM", "FGSM_eps001"], test_dataset=TinyDetectionSet(16, 1024, 22))
        assert isinstance(out, torch.nn.parallel.DistributedDataParallel)
        state = {k: v.numpy() for k, v in T.unwrap(out).state_dict().items()}
        np.savez(Path(out_dir) / f"train_rank{rank}.npz", weights=np.array(tr.adv_attacks_weights, dtype=np.float64),
                 shard=np.array([s[0] for s in seen]), **state)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_adversarial_training_stays_in_lockstep(tmp_path):
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "train_rank0.npz"), np.load(tmp_path / "train_rank1.npz")
    # every rank saw its half of each global batch of 8, 4 steps per epoch
    assert r0["shard"].tolist() == [4] * 8 and r1["shard"].tolist() == [4] * 8
    # gradients are averaged over ranks and the adaptive strategy is fed the rank-averaged loss:
    # identical replicas and identical attack weights at the end
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
    from tests.helpers import Surrogate
    torch.manual_seed(11)
    init = Surrogate().state_dict()
    assert any(not np.array_equal(init[k].numpy(), r0[k]) for k in init)
    assert abs(r0["weights"].sum() - 1.0) < 1e-6
