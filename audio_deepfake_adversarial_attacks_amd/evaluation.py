"""The adversarial-evaluation loop (reference: evaluate_models_on_adversarial_attacks.py:146-298).

`generate_attacks` keeps the reference's signature and per-batch order of operations
(min-max -> attack -> revert -> target forward -> sigmoid / threshold -> accumulate -> final report) and its
log line.  What is different is how the work is placed on the machine:

  * one process per GPU instead of `nn.DataParallel` (reference :163,:167): every rank owns a replica of both
    models and a CONTIGUOUS slice of each global batch — the same chunking DataParallel's scatter would do, so
    the LFCC batch-wide dB floor sees the same rows — and there is NO per-step traffic between GPUs;
  * batches reach the device one batch ahead of the attack, through pinned staging buffers on a side stream
    (`_device_batches`): the reference's blocking `.to(device)` per batch would idle the GPU for a copy + collation;
  * scores stay on the device until the end (the reference synchronises with `.item()` / `.cpu()` every batch,
    :261-265); one D2H copy per run;
  * the final aggregate is the only communication: an all-reduce(SUM) of the two counters and one all-gather of
    the packed (score, predicted label, label) rows over RCCL/xGMI (`backend="nccl"`), or gloo on CPU tests.
"""
from __future__ import annotations

import logging
import contextlib
import os
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset, Sampler

from . import metrics
from .aa import utils as aa_utils
from .datasets.base_dataset import WAVE_FAKE_CUT, SimpleAudioFakeDataset, ragged_collate
from .datasets.detection_dataset import DetectionDataset
from .datasets.synthetic import SyntheticDetectionDataset
from .datasets.wave_ops import RaggedWaveBatch
from .utils import load_model

LOGGER = logging.getLogger()


# ---------------------------------------------------------------------------------------------------------
# ranks and shards
# ---------------------------------------------------------------------------------------------------------

def rank_and_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank` when n rows are cut into `world` chunks (torch.chunk / DataParallel
    scatter semantics: every chunk has ceil(n / world) rows except possibly the last ones)."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


class ShardedBatchSampler(Sampler[List[int]]):
    """Yields, for every GLOBAL batch (drop_last=True as in the reference's DataLoader, :197-203), the indices
    of this rank's contiguous slice.  The permutation comes from a dedicated generator seeded identically on
    every rank, so all ranks agree on the global batches without communicating."""

    def __init__(self, n_items: int, global_batch: int, rank: int = 0, world: int = 1, shuffle: bool = True,
                 seed: int = 42):
        if global_batch % world != 0:
            raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
        self.n_items, self.global_batch, self.rank, self.world = n_items, global_batch, rank, world
        self.shuffle, self.seed = shuffle, seed

    def __len__(self) -> int:
        return self.n_items // self.global_batch

    def __iter__(self) -> Iterator[List[int]]:
        if self.shuffle:
            order = torch.randperm(self.n_items, generator=torch.Generator().manual_seed(self.seed)).tolist()
        else:
            order = list(range(self.n_items))
        lo, hi = shard_bounds(self.global_batch, self.rank, self.world)
        for b in range(len(self)):
            batch = order[b * self.global_batch:(b + 1) * self.global_batch]
            yield batch[lo:hi]


def aggregate_across_ranks(y_pred: torch.Tensor, y_pred_label: torch.Tensor, y: torch.Tensor,
                           num_correct: torch.Tensor, num_total: torch.Tensor):
    """End-of-run exchange.  Inputs are this rank's 1-D tensors / 0-d counters on the
    compute device; returns numpy arrays for the whole job, ordered by rank, plus the two global counters."""
    rank, world = rank_and_world()
    packed = torch.stack([y_pred.float(), y_pred_label.float(), y.float()], dim=1).contiguous()  # (n_local, 3)
    counters = torch.stack([num_correct.to(torch.int64), num_total.to(torch.int64)])
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed)
        packed = torch.cat(parts, dim=0)
    host = packed.cpu().numpy()
    c = counters.cpu().numpy()
    return host[:, 0].astype(np.float32), host[:, 1].astype(np.int32), host[:, 2].astype(np.int64), int(c[0]), int(c[1])


# ---------------------------------------------------------------------------------------------------------
# per-batch body
# ---------------------------------------------------------------------------------------------------------

def attack_batch(atk, batch_x: torch.Tensor, batch_y: torch.Tensor) -> torch.Tensor:
    """Reference :218-221 — normalise each utterance to [0, 1], attack, map back to the original range."""
    x01, mn, mx = aa_utils.to_minmax(batch_x)
    adv01 = atk(x01, batch_y)
    return aa_utils.revert_minmax(adv01, mn, mx)


@torch.no_grad()
def score_batch(model, batch_x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference :236-238 — probability of the bonafide class and the hard label ((p + .5).int() == p >= .5)."""
    preds = torch.sigmoid(model(batch_x).squeeze(1).detach())
    return preds, (preds + 0.5).int()


def format_report(report: Dict[str, float]) -> str:
    """The reference's final log line (:295-298)."""
    order = ("eer", "accuracy", "precision", "recall", "f1_score", "auc")
    return ", ".join(f"adv_eval/{k}: {report['adv_eval/' + k]:.4f}" for k in order)


# ---------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------

def _device_batches(loader, device):
    """The loader's batches with `batch_x` / `batch_y` already on the device, ONE BATCH AHEAD of the consumer.

    A pageable host->device copy on the compute stream waits for everything queued there — the previous batch's whole
    attack — and blocks the host meanwhile, so neither the copy nor the collation of the next batch overlaps the GPU.
    Here the next batch is staged through one of two persistent pinned buffers and copied on a side stream (ragged
    `device_pad` batches: uploaded and decoded + padded there) while the current batch's kernels run; the compute stream
    only waits for that stream's event.  CPU devices pass through unchanged."""
    dev = torch.device(device)
    if dev.type != "cuda":
        for batch_x, batch_sr, batch_y, meta in loader:
            yield batch_x.to(dev), batch_sr, batch_y.to(dev), meta, None
        return
    side = torch.cuda.Stream(dev)
    pinned, done, slot = {}, [None, None], 0

    def stage(batch):
        nonlocal slot
        batch_x, batch_sr, batch_y, meta = batch
        with torch.cuda.stream(side):
            if isinstance(batch_x, RaggedWaveBatch):
                # device_pad datasets: the file payloads go up as they are; decode + first channel + pad/tile on the device
                x_dev = batch_x.to_padded(dev, WAVE_FAKE_CUT)
            else:
                key = (tuple(batch_x.shape), batch_x.dtype)
                if key not in pinned:
                    pinned[key] = [torch.empty(batch_x.shape, dtype=batch_x.dtype, pin_memory=True) for _ in range(2)]
                if done[slot] is not None:
                    done[slot].synchronize()          # the previous upload from this staging buffer has finished
                buf = pinned[key][slot]
                buf.copy_(batch_x)
                x_dev = buf.to(dev, non_blocking=True)
            y_dev = batch_y.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        done[slot] = ev
        slot ^= 1
        return x_dev, batch_sr, y_dev, meta, ev

    it = iter(loader)
    try:
        ahead = stage(next(it))
    except StopIteration:
        return
    while ahead is not None:
        x_dev, batch_sr, y_dev, meta, ev = ahead
        # the consumer's stream for this batch waits for the upload (`_Lanes.batch`): with two batches in flight that is not
        # the stream this generator happens to run under
        yield x_dev, batch_sr, y_dev, meta, ev
        # resumed when the consumer asks for the next batch, i.e. after it has queued this batch's kernels: the
        # collation + upload below run behind them
        try:
            ahead = stage(next(it))
        except StopIteration:
            ahead = None


class _Lanes:
    """Batches in flight (round 6).  Batches of the evaluation loop are independent (reference
    :211-265: per-row min-max, per-batch dB floor, per-row attack state), and one PGD iteration leaves most of the 256 CUs
    idle for ~ 10 % of its time (the recurrent layers: 256 workgroups of 25 dependent steps; launch-sized GEMMs; the last,
    small convolution blocks).  With `n` = 2 batch i runs on stream i % 2, each stream replaying its own captured graph
    (torchattacks/graphed.py keys captures by launch stream), and the hardware fills one batch's idle CUs with the other's
    convolutions: configs[1] 69.1 -> 63.0 ms per batch (+ 9.6 %), scores bit-identical (tools/two_stream_probe.py).

    Ordering rules.  (1) A batch waits for its own upload event.  (2) Caches derived from the weights or from a batch shape
    (prepared Winograd weights, fragment tables, captured graphs) are built by whichever stream meets them first and are
    NOT ordered against other streams; so a batch runs behind everything queued before it — it waits for the previous
    batch's end event — until its (stream, batch shape) has been seen twice, i.e. until after that stream's capture.  From
    then on the two streams only read those caches.  (3) `join()` puts the caller's stream behind every lane before the
    scores are concatenated.  n = 1: the caller's stream, no extra synchronisation (rounds 1-5 behaviour)."""

    def __init__(self, device, n: int):
        self.dev = torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.n = max(1, int(n)) if self.cuda else 1
        self.main = torch.cuda.current_stream(self.dev) if self.cuda else None
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.n)] if self.n > 1 else [self.main]
        self.seen: Dict[tuple, int] = {}
        self.prev_end = None
        self.held = []                                  # tensors produced on a lane and read by the caller's stream after join()
        if self.n > 1:
            for s in self.streams:
                s.wait_stream(self.main)

    def lane_of(self, i: int) -> int:
        return i % self.n

    @contextlib.contextmanager
    def batch(self, i: int, shape_key, ready=None, inputs=()):
        if not self.cuda:
            yield 0
            return
        k = self.lane_of(i)
        stream = self.streams[k]
        if ready is not None:
            stream.wait_event(ready)
        for t in inputs:
            t.record_stream(stream)
        if self.n == 1:
            yield 0
            return
        key = (k, shape_key)
        warm = self.seen.get(key, 0) >= 2
        self.seen[key] = self.seen.get(key, 0) + 1
        if not warm and self.prev_end is not None:
            stream.wait_event(self.prev_end)            # rule (2): behind everything queued so far
        with torch.cuda.stream(stream):
            yield k
            end = torch.cuda.Event()
            end.record(stream)
        self.prev_end = end

    def keep(self, *tensors):
        if self.n > 1:
            self.held.extend(tensors)

    def join(self):
        if self.n > 1:
            for s in self.streams:
                self.main.wait_stream(s)
            for t in self.held:
                t.record_stream(self.main)
            self.held = []


def default_in_flight(atk, device, has_callback: bool) -> int:
    """Two batches in flight when the attack's inner loop replays from a hipGraph (PGD, PGDL2: the host queues a batch in a
    few milliseconds); host-driven attacks (CW: a host read every steps // 10 iterations; FAB) and callback runs (the
    analyser reads every batch back) stay at one.  ADVSTEP_IN_FLIGHT overrides."""
    env = os.environ.get("ADVSTEP_IN_FLIGHT")
    if env:
        return max(1, int(env))
    if atk is None or has_callback or torch.device(device).type != "cuda":
        return 1
    from .torchattacks import graphed
    return 2 if (getattr(atk, "replays_from_graph", False) and graphed.enabled()) else 1


def get_dataset(datasets_paths: List[Union[str, os.PathLike, None]], amount_to_use: Optional[int],
                raw_sample_from_dataset: bool = False, device_pad: bool = False,
                wave_fake_trim: Optional[bool] = None) -> DetectionDataset:
    """Reference :301-317 — the validation part of the three corpora, class-balanced, optionally reduced."""
    return DetectionDataset(asvspoof_path=datasets_paths[0], wavefake_path=datasets_paths[1],
                            fakeavceleb_path=datasets_paths[2], subset="val", reduced_number=amount_to_use,
                            return_label=True, return_meta=True, return_raw=raw_sample_from_dataset,
                            device_pad=device_pad, wave_fake_trim=wave_fake_trim)


def generate_attacks(
    datasets_paths: List[Union[str, os.PathLike, None]],
    model_config: Dict,
    device: str,
    attack_model_config: Optional[Dict] = None,
    attack_method: Optional[Any] = None,
    attack_params: Dict = {},
    amount_to_use: Optional[int] = None,
    batch_size: int = 64,
    on_attack_end_callback: Optional[Callable] = None,
    raw_sample_from_dataset: bool = False,
    dataset: Optional[Dataset] = None,
    share_weights: bool = False,
    shuffle: bool = True,
    num_workers: int = 3,
    device_pad: bool = True,
    wave_fake_trim: Optional[bool] = None,
    return_scores: bool = False,
    on_batch_queued: Optional[Callable[[int], None]] = None,
    in_flight: Optional[int] = None,
) -> Dict[str, float]:
    """Reference signature (:146-157) plus additive keywords: `dataset` (a ready Dataset yielding the reference's
    4-tuple; without it the `DetectionDataset` over `datasets_paths` is built as in the reference's `get_dataset`,
    :301-317, or — with no corpus path — `amount_to_use` synthetic utterances are generated), `share_weights`
    (white-box runs without checkpoints: copy the target's weights into the attack model), `shuffle`, `num_workers` (3 like the reference, :202 — the loop is launch-bound on the host,
    so collation belongs in worker processes),
    `device_pad` (real corpora: ship undecoded payloads and pad on the device) and `wave_fake_trim` (None = the
    reference's default, the SoX silence trim, which needs a registered backend) and `return_scores` (adds the whole job's
    per-utterance `y_pred`, `y_pred_label`, `y` arrays, in rank order, to the returned report under "scores") and
    `on_batch_queued(i)` (called when batch i's kernels have been queued, under the stream they were queued on — no
    synchronisation, no extra work: measurements) and `in_flight` (batches processed concurrently on streams of their own,
    `_Lanes`; None = `default_in_flight`: 2 for the graph-replayed attacks, else 1; same scores either way).
    `batch_size` is the GLOBAL batch."""
    rank, world = rank_and_world()
    LOGGER.info("Loading data...")

    model = load_model(model_config, device)                       # :162  (no DataParallel: one process per GPU)
    if attack_model_config is not None and attack_method is not None:
        attack_model = load_model(attack_model_config, device)     # :166
        if share_weights:
            attack_model.load_state_dict(model.state_dict())
        atk = attack_method(attack_model, **attack_params)         # :169
        atk.set_training_mode(model_training=True, batchnorm_training=False)  # :170
    else:
        attack_model, atk = None, None

    if dataset is None:
        if any(p is not None for p in datasets_paths):
            dataset = get_dataset(datasets_paths, amount_to_use, raw_sample_from_dataset, device_pad=device_pad,
                                  wave_fake_trim=wave_fake_trim)
        else:
            dataset = SyntheticDetectionDataset(amount_to_use if amount_to_use else 4 * batch_size)
    data_val = dataset
    collate_fn = ragged_collate if getattr(data_val, "device_pad", False) else None

    LOGGER.info(f"Testing '{model.__class__.__name__}' model, weights path: '{model.weights_path}', "
                f"on {len(data_val)} audio files.")
    if attack_model is not None:
        LOGGER.info(f"Attack using '{attack_model.__class__.__name__}' model and '{atk.__class__.__name__}' method "
                    f"({attack_params}), weights path: '{attack_model.weights_path}'")
    else:
        LOGGER.info("No attack applied")

    seed = model_config.get("data", {}).get("seed", 42)
    sampler = ShardedBatchSampler(len(data_val), batch_size, rank, world, shuffle=shuffle, seed=seed)
    # (pinning every (B, 64600) float batch was measured and costs more than it saves — a fresh pinned allocation per batch
    # on the launching thread: 1 082 -> 737 utt/s without workers; only the small ragged payloads are pinned)
    test_loader = DataLoader(data_val, batch_sampler=sampler, num_workers=num_workers, collate_fn=collate_fn,
                             pin_memory=collate_fn is not None)
    if world > 1:
        # decorrelate the random starts of different ranks (all ranks were seeded alike to build equal replicas)
        torch.manual_seed(seed + rank)

    lanes = _Lanes(device, in_flight if in_flight is not None
                   else default_in_flight(atk, device, on_attack_end_callback is not None))
    num_correct = [torch.zeros((), dtype=torch.int64, device=device) for _ in range(lanes.n)]
    seen_total = 0
    y_pred, y_pred_label, y = [], [], []

    for i, (batch_x, batch_sr, batch_y, batch_metadata, ready) in enumerate(_device_batches(test_loader, device)):
      with lanes.batch(i, tuple(batch_x.shape), ready, (batch_x, batch_y)) as lane:
        model.eval()
        seen_total += batch_x.size(0)

        if attack_model is not None:
            batch_x_attacked = attack_batch(atk, batch_x, batch_y)
        else:
            batch_x_attacked = torch.clone(batch_x)
        batch_x_noproc, batch_x_attacked_noproc = batch_x, batch_x_attacked  # :223-224 (nothing below writes in place)

        if raw_sample_from_dataset:  # :229-234 — the dataset's default preprocessing, after the attack
            batch_x_attacked, _ = SimpleAudioFakeDataset.wavefake_preprocessing_on_batch(
                batch_x_attacked, batch_sr, wave_fake_trim=wave_fake_trim)
        batch_preds, batch_preds_label = score_batch(model, batch_x_attacked)

        if on_attack_end_callback is not None:  # :240-259
            if raw_sample_from_dataset:
                batch_x, _ = SimpleAudioFakeDataset.wavefake_preprocessing_on_batch(
                    batch_x, batch_sr, wave_fake_trim=wave_fake_trim)
            batch_preds_noattack, batch_preds_noattack_label = score_batch(model, batch_x)
            on_attack_end_callback(batch_x=batch_x_noproc, batch_x_attacked=batch_x_attacked_noproc, batch_y=batch_y,
                                   batch_preds_label=batch_preds_label, batch_preds=batch_preds,
                                   batch_preds_noattack_label=batch_preds_noattack_label,
                                   batch_preds_noattack=batch_preds_noattack, batch_metadata=batch_metadata)

        num_correct[lane] += (batch_preds_label == batch_y.int()).sum()
        y_pred.append(batch_preds)
        y_pred_label.append(batch_preds_label)
        y.append(batch_y)
        lanes.keep(batch_preds, batch_preds_label, batch_y)
        if on_batch_queued is not None:
            on_batch_queued(len(y) - 1)

    lanes.keep(*num_correct)
    lanes.join()
    num_correct = num_correct[0] if lanes.n == 1 else torch.stack(num_correct).sum()
    num_total = torch.tensor(seen_total, dtype=torch.int64, device=device)
    if not y:
        raise ValueError(f"no complete batch: {len(data_val)} items < global batch {batch_size} (drop_last=True)")
    all_pred, all_label, all_y, n_correct, n_total = aggregate_across_ranks(
        torch.cat(y_pred), torch.cat(y_pred_label), torch.cat(y), num_correct, num_total)

    report = metrics.adversarial_report(all_y, all_pred, all_label)
    report["adv_eval/accuracy"] = (n_correct / n_total) * 100  # :267 (from the all-reduced counters)
    report["num_total"] = n_total
    if return_scores:
        report["scores"] = {"y_pred": all_pred, "y_pred_label": all_label, "y": all_y}
    if rank == 0:
        LOGGER.info(format_report(report))
    return report
