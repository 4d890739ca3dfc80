#!/usr/bin/env python
"""CLI of adversarial training — drop-in for the reference script of the same name
(train_models_on_adversarial_attacks.py:50-294): same flags, same YAML schema (`data.adversarial_attacks`,
`model.optimizer`, `checkpoint.path`), same strategy names, same checkpoint / test-config outputs.

Additive: --synthetic N_TRAIN,N_TEST (seeded synthetic utterances instead of the corpora), --no_trim (corpora without
the SoX silence trim, which needs a registered backend).  Multi-GPU: one process per GPU,
`python -m torch.distributed.run --nproc-per-node 8 train_models_on_adversarial_attacks.py ...` — the model is wrapped in
DistributedDataParallel (bucketed gradient all-reduce over RCCL, overlapped with backward) instead of the reference's
nn.DataParallel (:100, :110), --batch_size is the GLOBAL batch, every rank attacks and trains on its contiguous shard."""
import argparse
import logging
import os
import sys
import time
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import yaml

from audio_deepfake_adversarial_attacks_amd.aa.aa_trainer_types import AdversarialGDTrainerEnum
from audio_deepfake_adversarial_attacks_amd.datasets.detection_dataset import DetectionDataset
from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
from audio_deepfake_adversarial_attacks_amd.models import models
from audio_deepfake_adversarial_attacks_amd.trainer import save_model
from audio_deepfake_adversarial_attacks_amd.utils import load_model, set_seed

LOGGER = logging.getLogger()
LOGGER.setLevel(logging.INFO)
_handler = logging.StreamHandler()
_handler.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
LOGGER.addHandler(_handler)


def get_datasets(datasets_paths: List[Optional[str]], amount_to_use: Tuple[int, int],
                 synthetic: Optional[Tuple[int, int]] = None, wave_fake_trim: Optional[bool] = None):
    """train_models_on_adversarial_attacks.py:27-48 — the train / test parts of the three corpora, class-balanced;
    `--synthetic` replaces them with seeded noise."""
    if synthetic is not None:
        n_train = min(synthetic[0], amount_to_use[0]) if amount_to_use[0] else synthetic[0]
        n_test = min(synthetic[1], amount_to_use[1]) if amount_to_use[1] else synthetic[1]
        return (SyntheticDetectionDataset(n_train, seed=1234, return_meta=False),
                SyntheticDetectionDataset(n_test, seed=4321, return_meta=False))
    if all(p is None for p in datasets_paths):
        raise SystemExit("no data: pass --asv_path / --wavefake_path / --celeb_path, or --synthetic N_TRAIN,N_TEST")
    common = dict(asvspoof_path=datasets_paths[0], wavefake_path=datasets_paths[1], fakeavceleb_path=datasets_paths[2],
                  oversample=True, device_pad=True, wave_fake_trim=wave_fake_trim)
    return (DetectionDataset(subset="train", reduced_number=amount_to_use[0], **common),
            DetectionDataset(subset="test", reduced_number=amount_to_use[1], **common))


def train_nn(batch_size: int, epochs: int, device: str, config: Dict, attack_config: Optional[Dict],
             adversarial_attacks: List[str], model_dir: Optional[Path] = None,
             amount_to_use: Tuple[int, int] = (None, None), config_save_path: str = "configs",
             adv_training_strategy: str = AdversarialGDTrainerEnum.RANDOM.name, is_finetune: bool = False,
             synthetic: Optional[Tuple[int, int]] = None, datasets_paths: List[Optional[str]] = (None, None, None),
             wave_fake_trim: Optional[bool] = None):
    """train_models_on_adversarial_attacks.py:50-160."""
    model_config = config["model"]
    model_name = model_config["name"]
    optimizer_config = model_config["optimizer"]

    LOGGER.info("Loading data...")
    timestamp = time.time()
    checkpoint_paths = []
    data_train, data_test = get_datasets(list(datasets_paths), amount_to_use, synthetic, wave_fake_trim)

    current_model = models.get_model(model_name=model_name, config=model_config["parameters"], device=device)
    if is_finetune:
        assert config["checkpoint"]["path"], "Finetune requires to provide checkpoint"
        weights_path = config["checkpoint"]["path"]
        lr = config["model"]["optimizer"]["lr"]
        LOGGER.info(f"Adversarial finetuning! Architecture: '{model_name}', lr: {lr}, weights: '{weights_path}'!")
        current_model.load_state_dict(torch.load(weights_path, map_location="cpu"))
    current_model = current_model.to(device)

    use_scheduler = "rawnet3" in model_name.lower()

    if attack_config is not None:
        LOGGER.info("Load attack model based on attack config")
        attack_model_name = attack_config["model"]["name"]
        attack_model = load_model(attack_config, device)
        attack_info = f"{attack_model_name} (pretrained) {attack_config['checkpoint'].get('path', '')}"
    else:
        LOGGER.info("Use target model as attack model")
        attack_model = current_model
        attack_info = model_name

    LOGGER.info(f"Training '{model_name}', attacking using: '{attack_info}' model on {len(data_train)} audio files.")
    LOGGER.info(f"Adversarial training strategy: {adv_training_strategy}")
    save_name = f"aad__{model_name}_{timestamp}"

    current_model = AdversarialGDTrainerEnum[adv_training_strategy].value(
        device=device, batch_size=batch_size, epochs=epochs, optimizer_kwargs=optimizer_config, use_scheduler=use_scheduler,
    ).train(dataset=data_train, model=current_model, attack_model=attack_model, test_dataset=data_test,
            adversarial_attacks=adversarial_attacks, model_dir=model_dir, save_model_name=save_name)

    rank = int(os.environ.get("RANK", "0"))
    if model_dir is not None:
        save_model(model=current_model, model_dir=model_dir, name=save_name)
        checkpoint_paths.append(str(model_dir.resolve() / save_name / "ckpt.pth"))
    LOGGER.info("Training model done!")

    if model_dir is not None and rank == 0:
        config["checkpoint"] = {"paths": checkpoint_paths, "path": checkpoint_paths[0]}
        config_save_path = str(Path(config_save_path) / f"aad__{model_name}__{timestamp}.yaml")
        with open(config_save_path, "w") as f:
            yaml.dump(config, f)
        LOGGER.info(f"Test config saved at location '{config_save_path}'!")
    return current_model


def main(args):
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)
    attack_model_config = None
    if args.attack_model_config is not None:
        with open(args.attack_model_config, "r") as f:
            attack_model_config = yaml.safe_load(f)

    set_seed(config["data"].get("seed", 42))

    if args.cpu or not torch.cuda.is_available():
        raise SystemExit("No HIP device (or --cpu): the attack kernels of this build run on MI355X only (no CPU fallback).")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend="nccl", device_id=torch.device(device))  # "nccl" is RCCL on ROCm

    if not args.synthetic:
        from audio_deepfake_adversarial_attacks_amd.datasets import backends
        backends.require_for(args.asv_path, args.wavefake_path, args.celeb_path, trim=not args.no_trim)
    model_dir = Path(args.ckpt)
    model_dir.mkdir(parents=True, exist_ok=True)
    synthetic = tuple(int(v) for v in args.synthetic.split(",")) if args.synthetic else None

    train_nn(device=device, amount_to_use=(args.train_amount, args.test_amount), batch_size=args.batch_size,
             epochs=args.epochs, model_dir=model_dir, config=config, attack_config=attack_model_config,
             adversarial_attacks=config["data"].get("adversarial_attacks", []),
             adv_training_strategy=args.adv_training_strategy, is_finetune=args.finetune, synthetic=synthetic,
             config_save_path=args.config_save_path,
             datasets_paths=[args.asv_path, args.wavefake_path, args.celeb_path],
             wave_fake_trim=False if args.no_trim else None)
    if world > 1:
        dist.destroy_process_group()


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--asv_path", type=str, default=None, help="Path to ASVspoof2021 dataset directory")
    parser.add_argument("--wavefake_path", type=str, default=None, help="Path to WaveFake dataset directory")
    parser.add_argument("--celeb_path", type=str, default=None, help="Path to FakeAVCeleb dataset directory")
    parser.add_argument("--config", help="Model config file path (default: config.yaml)", type=str, default="config.yaml")
    parser.add_argument("--attack_model_config", type=str, default=None,
                        help="Model config file path - if not provided, training will proceed using weights of the trained model")
    parser.add_argument("--train_amount", "-a", help="Amount of files to load for training.", type=int, default=100_000)
    parser.add_argument("--test_amount", "-ta", help="Amount of files to load for testing.", type=int, default=10_000)
    parser.add_argument("--batch_size", "-b", help="Batch size (default: 64).", type=int, default=64)
    parser.add_argument("--epochs", "-e", help="Epochs (default: 5).", type=int, default=5)
    parser.add_argument("--ckpt", help="Checkpoint directory (default: trained_models).", type=str, default="trained_models")
    parser.add_argument("--adv_training_strategy", help="Adversarial training strategy", type=str,
                        default=AdversarialGDTrainerEnum.RANDOM.name, choices=[e.name for e in AdversarialGDTrainerEnum])
    parser.add_argument("--cpu", "-c", help="Force using cpu?", action="store_true")
    parser.add_argument("--finetune", help="Finetune using checkpoint provided in a config", action="store_true")
    # additive
    parser.add_argument("--synthetic", type=str, default=None, metavar="N_TRAIN,N_TEST",
                        help="train / test on seeded synthetic 64 600-sample utterances")
    parser.add_argument("--no_trim", default=False, action="store_true",
                        help="real corpora: skip the SoX silence trim (needs a registered backend otherwise)")
    parser.add_argument("--config_save_path", type=str, default="configs", help="where the test config is written")
    return parser.parse_args(argv)


if __name__ == "__main__":
    logging.basicConfig(stream=sys.stdout, level=logging.INFO)
    main(parse_args())
