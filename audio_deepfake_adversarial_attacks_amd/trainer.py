"""Adversarial training — the second caller of the attack API (reference: src/trainer.py:20-581).

Same classes, constructor / `train` signatures, strategy rules and log lines as the reference:

    Trainer, GDTrainer                               src/trainer.py:36-210
    AdversarialGDTrainer            (RANDOM)         :213-466   attack a batch with probability n/(n+1), attack chosen uniformly
    EqualAdversarialGDTrainer       (EQUAL)          :469-487   attack a random half of every batch with the first attack
    OnlyOneAdversarialGDTrainer     (ONLY_ADV)       :490-504   always attack, exactly one attack allowed
    AdaptiveAdversarialGDTrainer    (ADAPTIVE)       :507-545   attack drawn with loss-driven weights
    AdaptiveV2AdversarialGDTrainer  (ADAPTIVE_V2)    :548-581   same with a 2/3 : 1/3 attack : clean prior

The strategies draw from Python's `random` exactly where the reference does, so a seeded run makes the same choices.
What differs is how the work is placed on the machine (MI355X-first, one process per GPU):

  * the attacks run through the HIP kernels (`atk.ops`: min-max, attack steps, revert) on the rank's own shard; while an
    attack runs the attacked model's parameters are frozen (`Attack.__call__`), so the model's fused forward +
    input-backward kernels are used for the 10-100 attack iterations and plain autograd only for the one training step;
  * `nn.DataParallel` (train_models_on_adversarial_attacks.py:100) becomes `DistributedDataParallel` when
    `torch.distributed` is initialised: gradients are all-reduced in buckets over RCCL / xGMI while the backward pass
    is still running, every rank owns a contiguous shard of each global batch (`ShardedBatchSampler`) and attacks only
    its shard — there is no scatter / gather of waveforms;
  * losses and counters are summed across ranks once per epoch (and the scalar loss once per step for the adaptive
    strategies, so every rank keeps the same attack weights and makes the same draws);
  * scalars are read back once per step (the reference calls `.item()` three times).
"""
from __future__ import annotations

import functools
import logging
import random
from copy import deepcopy
from pathlib import Path
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .aa.aa_types import AttackEnum
from .datasets.base_dataset import WAVE_FAKE_CUT, ragged_collate
from .datasets.wave_ops import RaggedWaveBatch
from .evaluation import ShardedBatchSampler, rank_and_world

LOGGER = logging.getLogger(__name__)


def save_model(model: torch.nn.Module, model_dir: Union[Path, str], name: str, epoch: Optional[int] = None) -> None:
    """src/trainer.py:20-33; rank 0 writes.  Keys are saved as the model presents them: `module.`-prefixed when it is
    wrapped (the reference always saves its DataParallel wrapper), bare otherwise — `utils.load_model` reads both."""
    if rank_and_world()[0] != 0:
        return
    full_model_dir = Path(f"{model_dir}/{name}")
    full_model_dir.mkdir(parents=True, exist_ok=True)
    epoch_str = f"_{epoch:02d}" if epoch is not None else ""
    torch.save(model.state_dict(), f"{full_model_dir}/ckpt{epoch_str}.pth")
    LOGGER.info(f"Training model saved under: {full_model_dir}/ckpt{epoch}.pth")


def unwrap(model: torch.nn.Module) -> torch.nn.Module:
    return model.module if isinstance(model, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)) else model


class Trainer:
    """Lightweight wrapper storing the training set-up (src/trainer.py:36-67)."""

    loader_workers = 0      # the reference forks 6 DataLoader workers to decode audio; tensors already in memory need none
    corpus_loader_workers = 6   # ... audio files on disk do (src/trainer.py:154,162)
    attack_ops = None       # test seam: op table handed to every attack (None = the HIP kernels)

    def __init__(self, epochs: int = 20, batch_size: int = 32, device: str = "cpu",
                 optimizer_fn: Callable = torch.optim.Adam, optimizer_kwargs: dict = {"lr": 1e-3},
                 use_scheduler: bool = False) -> None:
        self.epochs = epochs
        self.batch_size = batch_size
        self.device = device
        self.optimizer_fn = optimizer_fn
        self.optimizer_kwargs = optimizer_kwargs
        self.epoch_test_losses: List[float] = []
        self.use_scheduler = use_scheduler

    # ---- shared plumbing -------------------------------------------------------------------------------------------

    def _split(self, dataset, test_len, test_dataset):
        if test_dataset is not None:
            return dataset, test_dataset
        n_test = int(len(dataset) * test_len)
        return torch.utils.data.random_split(dataset, [len(dataset) - n_test, n_test])

    def _upload(self, batch_x):
        """Batch to the device; `device_pad` corpora (datasets/base_dataset.py) arrive as undecoded payloads and are
        decoded + padded there."""
        if isinstance(batch_x, RaggedWaveBatch):
            return batch_x.to_padded(self.device, WAVE_FAKE_CUT)
        return batch_x.to(self.device)

    def _loader(self, data, epoch_seed: int) -> DataLoader:
        """Single process: the reference's DataLoader(shuffle=True, drop_last=True).  Distributed: this rank's
        contiguous shard of every global batch of `batch_size`, permutation shared through the seed."""
        rank, world = rank_and_world()
        base = data.dataset if isinstance(data, torch.utils.data.Subset) else data
        extra, workers = {}, self.loader_workers
        if getattr(base, "device_pad", False):  # audio files: decode in workers (reference: 6), pad on the device
            extra = dict(collate_fn=ragged_collate, pin_memory=True)
            workers = self.loader_workers or self.corpus_loader_workers
        if world == 1:
            return DataLoader(data, batch_size=self.batch_size, shuffle=True, drop_last=True, num_workers=workers, **extra)
        sampler = ShardedBatchSampler(len(data), self.batch_size, rank, world, shuffle=True, seed=epoch_seed)
        return DataLoader(data, batch_sampler=sampler, num_workers=workers, **extra)

    @staticmethod
    def _sum_over_ranks(*values: float) -> List[float]:
        _, world = rank_and_world()
        if world == 1:
            return list(values)
        t = torch.tensor(values, dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.tolist()

    def _scheduler(self, optim, batches_per_epoch: int, eta_min: float):
        return torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer=optim, T_0=batches_per_epoch, T_mult=1,
                                                                    eta_min=eta_min)


def forward_and_loss(model, criterion, batch_x, batch_y, **kwargs):
    batch_out = model(batch_x)
    batch_loss = criterion(batch_out, batch_y)
    return batch_out, batch_loss


def _maybe_ddp(model: torch.nn.Module, device) -> torch.nn.Module:
    """One replica per process; gradients all-reduced in buckets while backward runs (RCCL over xGMI on the GPU box)."""
    _, world = rank_and_world()
    if world == 1 or isinstance(model, torch.nn.parallel.DistributedDataParallel):
        return model
    dev = torch.device(device)
    ids = [dev.index if dev.index is not None else torch.cuda.current_device()] if dev.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(unwrap(model), device_ids=ids, bucket_cap_mb=64,
                                                     gradient_as_bucket_view=True)


class GDTrainer(Trainer):
    """Plain gradient-descent training (src/trainer.py:76-210)."""

    def train(self, dataset: torch.utils.data.Dataset, model: torch.nn.Module, test_len: Optional[float] = None,
              test_dataset: Optional[torch.utils.data.Dataset] = None):
        trainer = AdversarialGDTrainer(self.epochs, self.batch_size, self.device, self.optimizer_fn, self.optimizer_kwargs,
                                       self.use_scheduler)
        trainer.loader_workers = self.loader_workers
        # src/trainer.py:118 restarts the cosine schedule every second epoch for clean training
        return trainer._fit(dataset, model, None, [], test_len, test_dataset, None, None, scheduler_period=2,
                            banner=f"Starting training for {self.epochs} epochs!")


class AdversarialGDTrainer(Trainer):

    def __init__(self, *args, **kwargs):
        super(AdversarialGDTrainer, self).__init__(*args, **kwargs)
        self.attacks = None

    @staticmethod
    def multi_f1_score(results):
        s = sum(results)
        m = functools.reduce(lambda x, y: x * y, results)
        return len(results) * m / s

    def train(self, dataset: torch.utils.data.Dataset, model: torch.nn.Module, attack_model: torch.nn.Module,
              adversarial_attacks: List[str], test_len: Optional[float] = None,
              test_dataset: Optional[torch.utils.data.Dataset] = None, model_dir: Optional[str] = None,
              save_model_name: Optional[str] = None):
        return self._fit(dataset, model, attack_model, adversarial_attacks, test_len, test_dataset, model_dir,
                         save_model_name, scheduler_period=1,
                         banner=f"Starting adversarial training for {self.epochs} epochs!")

    # ---- the loop (src/trainer.py:224-401) -----------------------------------------------------------------------------

    def _fit(self, dataset, model, attack_model, adversarial_attacks, test_len, test_dataset, model_dir, save_model_name,
             scheduler_period: int, banner: str):
        train, test = self._split(dataset, test_len, test_dataset)
        adversarial = attack_model is not None
        train_loader = self._loader(train, epoch_seed=0)
        test_loader = self._loader(test, epoch_seed=1)

        criterion = torch.nn.BCEWithLogitsLoss()
        optim = self.optimizer_fn(model.parameters(), **self.optimizer_kwargs)
        best_model, best_acc = None, 0
        LOGGER.info(banner)

        scheduler = None
        if self.use_scheduler:
            if adversarial:
                LOGGER.info("Using optimizer scheduler!")
            scheduler = self._scheduler(optim, len(train_loader) * scheduler_period,
                                        self.optimizer_kwargs.get("eta_min", 5e-6) if adversarial else 5e-6)

        if adversarial:
            self.init_adv_attacks(unwrap(attack_model), adversarial_attacks)
        model = _maybe_ddp(model, self.device)
        rank, world = rank_and_world()
        if world > 1:
            # every rank was seeded alike so the replicas start equal; from here on only torch's generators are
            # decorrelated — random starts (Philox keys come from torch's CPU generator) and dropout masks differ per
            # shard, as evaluation.generate_attacks does.  Python's `random` stays shared: the strategies' draws and the
            # adaptive attack weights must be the same on every rank.
            base = torch.initial_seed()
            torch.manual_seed(base + rank)
            if torch.cuda.is_available():
                torch.cuda.manual_seed(base + rank)

        for epoch in range(self.epochs):
            LOGGER.info(f"Epoch num: {epoch}")
            running_loss, num_correct, num_total = 0.0, 0.0, 0.0
            model.train()
            if world > 1:
                train_loader = self._loader(train, epoch_seed=2 * epoch)

            for i, (batch_x, _, batch_y) in enumerate(train_loader):
                batch_size = batch_x.size(0)
                num_total += batch_size
                batch_x = self._upload(batch_x)
                if adversarial:
                    batch_x = self.apply_adv_attack(batch_x, batch_y).detach()
                batch_y = batch_y.unsqueeze(1).type(torch.float32).to(self.device)

                batch_out, batch_loss = forward_and_loss(model=model, criterion=criterion, batch_x=batch_x, batch_y=batch_y)
                batch_pred = (torch.sigmoid(batch_out) + .5).int()
                # one read-back per step: [loss, correct]
                stats = torch.stack([batch_loss.detach().float(), (batch_pred == batch_y.int()).sum().float()]).cpu()
                loss_value = stats[0].numpy()
                num_correct += stats[1].item()
                running_loss += loss_value.item() * batch_size

                if i % 100 == 0:
                    LOGGER.info(f"[{epoch:04d}][{i:05d}]: {running_loss / num_total} {num_correct / num_total * 100}")

                optim.zero_grad()
                batch_loss.backward()
                optim.step()
                if scheduler is not None:
                    scheduler.step()

                if adversarial:
                    if world > 1:   # every rank must update its attack weights with the same number
                        loss_value = np.float32(self._sum_over_ranks(float(loss_value))[0] / world)
                    self.update_adv_attack(loss_value, batch_pred, iter=i, epoch=epoch)

            weighted, num_correct, num_total = self._sum_over_ranks(running_loss, num_correct, num_total)
            running_loss = weighted / num_total
            train_accuracy = (num_correct / num_total) * 100
            LOGGER.info(f"Epoch [{epoch+1}/{self.epochs}]: train/loss: {running_loss}, train/accuracy: {train_accuracy}")

            test_running_loss, test_acc, eer_val = self.validation_epoch(model=model, criterion=criterion,
                                                                         test_loader=test_loader, attack=None)
            test_acc_results = [test_acc / 100]
            LOGGER.info(f"Epoch [{epoch+1}/{self.epochs}]: test/loss: {test_running_loss}, "
                        f"test/accuracy: {test_acc}, test/eer: {eer_val}")

            for (attack_name, attack_method) in (self.attacks or []):
                # a fresh loader per attack, as the reference (:364-370); single process: a new shuffle drawn from
                # torch's global generator, distributed: the same shared permutation for every attack
                test_loader = self._loader(test, epoch_seed=1)
                adv_loss, adv_acc, adv_eer = self.validation_epoch(model=model, criterion=criterion,
                                                                   test_loader=test_loader, attack=attack_method)
                test_acc_results.append(adv_acc / 100)
                LOGGER.info(f"Epoch [{epoch+1}/{self.epochs}]: adv_test/{attack_name}__loss: {adv_loss},"
                            f" adv_test/{attack_name}__accuracy: {adv_acc}, adv_test/{attack_name}__eer: {adv_eer}.")

            LOGGER.info(f"[{epoch:04d}]: loss {running_loss}, train acc: {train_accuracy}, test_acc: {test_acc}")
            if adversarial:
                test_acc = self.multi_f1_score(test_acc_results)
                LOGGER.info(f"[{epoch:04d}]: multi_f1_score: {test_acc}")

            if best_model is None or test_acc > best_acc:
                best_acc = test_acc
                best_model = deepcopy(model.state_dict())
                if adversarial:
                    LOGGER.info(f"[{epoch:04d}]: update best model")

            if model_dir is not None:
                save_model(model=model, model_dir=model_dir, name=save_model_name, epoch=epoch)

        model.load_state_dict(best_model)
        return model

    def validation_epoch(self, model, test_loader, criterion, attack: Optional[Callable]):
        """src/trainer.py:403-447 (EER reporting is disabled upstream: always 0)."""
        model.eval()
        test_running_loss, num_correct, num_total = 0.0, 0.0, 0.0
        eer_val = 0

        for batch_x, _, batch_y in test_loader:
            batch_size = batch_x.size(0)
            num_total += batch_size
            batch_x = self._upload(batch_x)
            if attack:
                batch_x = self._attack_batch(attack, batch_x, batch_y)
            batch_y = batch_y.unsqueeze(1).type(torch.float32).to(self.device)
            with torch.no_grad():
                batch_pred = model(batch_x)
            batch_loss = criterion(batch_pred, batch_y)
            batch_pred_label = (torch.sigmoid(batch_pred) + .5).int()
            stats = torch.stack([batch_loss.detach().float(), (batch_pred_label == batch_y.int()).sum().float()]).cpu()
            test_running_loss += stats[0].item() * batch_size
            num_correct += stats[1].item()

        test_running_loss, num_correct, num_total = self._sum_over_ranks(test_running_loss, num_correct, num_total)
        if num_total == 0:
            num_total = 1
        return test_running_loss / num_total, 100 * (num_correct / num_total), eer_val

    # ---- attacks -------------------------------------------------------------------------------------------------------

    @staticmethod
    def _attack_batch(attack, batch_x, batch_y):
        """to_minmax -> attack -> revert_minmax (src/trainer.py:424-426, 470-472, ...) through the attack's op table."""
        ops = attack.ops
        x01, mn, mx = ops.to_minmax(batch_x.contiguous())
        return ops.revert_minmax(attack(x01, batch_y).contiguous(), mn, mx)

    def init_adv_attacks(self, attack_model, adversarial_attacks):
        self.attacks = []
        for attack_method_name in adversarial_attacks:
            attack_method, attack_params = AttackEnum[attack_method_name].value
            atk = attack_method(attack_model, **attack_params)
            atk.set_training_mode(model_training=True, batchnorm_training=False)
            if self.attack_ops is not None:
                atk.ops = self.attack_ops
            self.attacks.append((attack_method_name, atk))
        LOGGER.info(f"Adversarial attacks: {adversarial_attacks}")
        return self.attacks

    def apply_adv_attack(self, batch_x, batch_y):
        if random.random() > 1 / (len(self.attacks) + 1):
            attack_index = random.randint(0, len(self.attacks) - 1)
            _, attack_for_batch = self.attacks[attack_index]
            batch_x = self._attack_batch(attack_for_batch, batch_x, batch_y)
        return batch_x

    def update_adv_attack(self, batch_loss, batch_pred, iter=None, epoch=None):
        ...


class EqualAdversarialGDTrainer(AdversarialGDTrainer):
    def apply_adv_attack(self, batch_x, batch_y):
        _, attack_for_batch = self.attacks[0]
        indices_to_attack = random.sample(range(len(batch_x)), len(batch_x) // 2)
        attacked = self._attack_batch(attack_for_batch, batch_x[indices_to_attack], batch_y[indices_to_attack])
        batch_x[indices_to_attack, ...] = attacked
        return batch_x


class OnlyOneAdversarialGDTrainer(AdversarialGDTrainer):

    def init_adv_attacks(self, attack_model, adversarial_attacks):
        assert len(adversarial_attacks) == 1, "Method allows to apply only one attack"
        self.attacks = super().init_adv_attacks(attack_model, adversarial_attacks)
        return self.attacks

    def apply_adv_attack(self, batch_x, batch_y):
        _, attack_for_batch = self.attacks[0]
        return self._attack_batch(attack_for_batch, batch_x, batch_y)


class AdaptiveAdversarialGDTrainer(AdversarialGDTrainer):

    def __init__(self, *args, **kwargs):
        super(AdaptiveAdversarialGDTrainer, self).__init__(*args, **kwargs)
        self.adv_attacks_weights = None
        self.last_adv_attack = None

    def init_adv_attacks(self, attack_model, adversarial_attacks):
        self.attacks = super().init_adv_attacks(attack_model, adversarial_attacks)
        self.adv_attacks_weights = [1 / (len(self.attacks) + 1)] * (len(self.attacks) + 1)
        return self.attacks

    def apply_adv_attack(self, batch_x, batch_y):
        attack_idx, = random.choices(range(len(self.attacks) + 1), weights=self.adv_attacks_weights, k=1)
        self.last_adv_attack = attack_idx
        if attack_idx < len(self.attacks):
            _, attack_for_batch = self.attacks[attack_idx]
            batch_x = self._attack_batch(attack_for_batch, batch_x, batch_y)
        return batch_x

    def _blend_last(self, batch_loss, max_val, proportion_val):
        loss = min(batch_loss, max_val)
        k = self.last_adv_attack
        self.adv_attacks_weights[k] = proportion_val * loss + (1 - proportion_val) * self.adv_attacks_weights[k]
        return np.sum(self.adv_attacks_weights)

    def update_adv_attack(self, batch_loss, batch_pred, max_val=1, proportion_val=0.2, iter=None, epoch=None):
        weights_sum = self._blend_last(batch_loss, max_val, proportion_val)
        n = len(self.adv_attacks_weights)
        self.adv_attacks_weights = [0.5 * (w / weights_sum) + 0.5 * (1.0 / n) for w in self.adv_attacks_weights]
        if iter is not None and iter % 100 == 0:
            LOGGER.info(f"[{epoch:04d}][{iter:05d}]: Adversarial attack weights: {self.adv_attacks_weights}")


class AdaptiveV2AdversarialGDTrainer(AdaptiveAdversarialGDTrainer):
    def update_adv_attack(self, batch_loss, batch_pred, max_val=1, proportion_val=0.2, iter=None, epoch=None):
        weights_sum = self._blend_last(batch_loss, max_val, proportion_val)
        halves = [0.5 * (w / weights_sum) for w in self.adv_attacks_weights]
        non_attack_ratio = 1 / 3
        attack_ratio = (2 / 3) / len(self.attacks)
        last = len(halves) - 1
        self.adv_attacks_weights = [w + 0.5 * (attack_ratio if i < last else non_attack_ratio) for i, w in enumerate(halves)]
        if iter is not None and iter % 100 == 0:
            LOGGER.info(f"[{epoch:04d}][{iter:05d}]: Adversarial attack weights: {self.adv_attacks_weights}")
