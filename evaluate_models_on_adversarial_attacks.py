#!/usr/bin/env python
"""CLI of the adversarial-evaluation hot loop — drop-in for the reference script of the same name
(evaluate_models_on_adversarial_attacks.py:38-143): same flags, same YAML schema, same final log line.

Additive flags (defaults keep the reference behaviour): --batch_size (the reference hard-codes 64, :154),
--synthetic N (seeded synthetic utterances instead of the corpora), --no_trim (real corpora without the SoX silence
trim, which needs a registered backend), --num_workers, --share_weights (white-box runs with random-init models).  Multi-GPU: launch one process per GPU, e.g.
`python -m torch.distributed.run --nproc-per-node 8 evaluate_models_on_adversarial_attacks.py ...`;
each rank attacks a contiguous slice of every global batch and only the final scores cross xGMI (RCCL).

The attack kernels run on a HIP device only: without a GPU this script stops with an error (the CPU
restatement used as parity oracle / CPU baseline is test infrastructure: `python -m oracle.cpu_eval`)."""
import argparse
import logging
import os
from datetime import datetime
from pathlib import Path

import torch
import torch.distributed as dist
import yaml

from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
from audio_deepfake_adversarial_attacks_amd.aa.qualitative.attacks_analysis import AttackAnalyser
from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks
from audio_deepfake_adversarial_attacks_amd.utils import set_seed

LOGGER = logging.getLogger()


def setup_logging():
    LOGGER.setLevel(logging.INFO)
    formatter = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    sh = logging.StreamHandler()
    sh.setFormatter(formatter)
    LOGGER.addHandler(sh)
    if int(os.environ.get("RANK", "0")) == 0:
        Path("logs").mkdir(exist_ok=True)
        fh = logging.FileHandler(f"logs/{datetime.now()}.log")
        fh.setFormatter(formatter)
        LOGGER.addHandler(fh)


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser()
    # dataset roots (reference :40-53 hard-codes the authors' paths as defaults; here the default is "not given", so a
    # run names its data explicitly: corpus roots and/or --synthetic N).  WAVE corpora decode in-tree; FLAC / MP3 and the
    # SoX silence trim use soundfile / torchaudio when importable (datasets/backends.py checks at start-up)
    parser.add_argument("--asv_path", type=str, default=None)
    parser.add_argument("--wavefake_path", type=str, default=None)
    parser.add_argument("--celeb_path", type=str, default=None)
    parser.add_argument("--attack", help="Attack name", type=str, default=AttackEnum.NO_ATTACK.name,
                        choices=[e.name for e in AttackEnum])
    parser.add_argument("--attack_model_config", help="Attack model config file path", type=str, default=None)
    parser.add_argument("--config", help="Model config file path", type=str, default="configs/lcnn.yaml")
    parser.add_argument("--amount", "-a", type=int, default=None,
                        help="Amount of files to load from each directory (default: None - use all).")
    parser.add_argument("--qual", help="Generate qualitative results", default=False, action="store_true")
    parser.add_argument("--raw_from_dataset", help="Return raw sample from the dataset", default=False,
                        action="store_true")
    # additive
    parser.add_argument("--batch_size", type=int, default=64, help="GLOBAL batch size (reference: fixed 64)")
    parser.add_argument("--synthetic", type=int, default=None, metavar="N",
                        help="evaluate on N seeded synthetic 64 600-sample utterances")
    parser.add_argument("--no_trim", default=False, action="store_true",
                        help="real corpora: skip the SoX silence trim.  DEPARTS from the reference preprocessing (which "
                             "trims silence, base_dataset.py:29-33); without this flag a SoX backend must be importable "
                             "(datasets/backends.py)")
    parser.add_argument("--num_workers", type=int, default=3, help="DataLoader workers (reference :202: 3)")
    parser.add_argument("--in_flight", type=int, default=None, metavar="N",
                        help="batches processed concurrently, each on a stream of its own (default: 2 for the attacks whose "
                             "iteration replays from a hipGraph - PGD, PGDL2 -, else 1; same scores either way)")
    parser.add_argument("--share_weights", default=False, action="store_true",
                        help="copy the target model's weights into the attack model (white-box, no checkpoints)")
    return parser.parse_args(argv)


def main(args):
    print(args)
    if not torch.cuda.is_available():
        raise SystemExit("No HIP device: the attack kernels of this build run on MI355X only (no CPU fallback). "
                         "For the CPU oracle / baseline use `python -m oracle.cpu_eval`.")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend="nccl", device_id=torch.device(device))  # "nccl" is RCCL on ROCm

    attack_model_config = None
    if args.attack_model_config is not None:
        with open(args.attack_model_config, "r") as f:
            attack_model_config = yaml.safe_load(f)
    with open(args.config, "r") as f:
        config = yaml.safe_load(f)

    set_seed(config["data"].get("seed", 42))
    attack_method, attack_params = AttackEnum[args.attack].value
    on_attack_end_callback = None
    if args.qual:  # reference :126-129
        if args.attack_model_config is None:
            raise SystemExit("--qual names its folder after the attack model: pass --attack_model_config")
        results_folder = f"attack_{args.attack}_{Path(args.attack_model_config).stem}_on_{Path(args.config).stem}"
        if world > 1:   # one folder per rank: ranks analyse disjoint shards and must not overwrite each other's files
            results_folder = f"{results_folder}/rank{int(os.environ.get('RANK', '0'))}"
        attack_analyser = AttackAnalyser(Path("qualitative_results") / results_folder)
        on_attack_end_callback = attack_analyser.analyse
    corpora = [args.asv_path, args.wavefake_path, args.celeb_path]
    if args.synthetic is None and all(p is None for p in corpora):
        raise SystemExit("no data: pass --asv_path / --wavefake_path / --celeb_path, or --synthetic N")
    if args.synthetic is None:
        from audio_deepfake_adversarial_attacks_amd.datasets import backends
        backends.require_for(*corpora, trim=not args.no_trim)      # one clear message now, not a worker exception later

    report = generate_attacks(
        datasets_paths=[args.asv_path, args.wavefake_path, args.celeb_path],
        model_config=config,
        attack_model_config=attack_model_config,
        attack_method=attack_method,
        attack_params=attack_params,
        amount_to_use=args.amount,
        batch_size=args.batch_size,
        device=device,
        on_attack_end_callback=on_attack_end_callback,
        raw_sample_from_dataset=args.raw_from_dataset,
        dataset=SyntheticDetectionDataset(args.synthetic) if args.synthetic is not None else None,
        share_weights=args.share_weights,
        wave_fake_trim=False if args.no_trim else None,
        num_workers=args.num_workers,
        in_flight=args.in_flight,
    )
    if world > 1:
        dist.destroy_process_group()
    return report


if __name__ == "__main__":
    setup_logging()
    main(parse_arguments())
