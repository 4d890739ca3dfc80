#!/usr/bin/env python
"""Throughput of the OTHER BASELINE.json configurations through the shipped evaluation loop (`generate_attacks`), one
GPU, synthetic data — informational companions of the bench line (which is configs[1]):

  configs[2]  SpecRNet + mel-spec frontend, PGDL2-40 (eps 0.1), B = 128
  configs[3]  RawNet3 attack model -> LCNN + LFCC target (transferability), FGSM and CW-100, B = 64
  (plus configs[1] through the same loop, for a like-for-like number, and FAB on LCNN)

Prints steady-state utterances/s (median per-batch time inside one call of the loop)."""
import argparse
import sys
import time
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.utils import set_seed  # noqa: E402


def cfg(name):
    return yaml.safe_load((ROOT / "configs" / "aa_evaluation" / f"{name}.yaml").read_text())


def run(label, target, attack_model, attack, batch, batches, workers=0):
    """One call of the shipped loop over `batches` batches; the per-batch time is the median distance between the
    (synchronised) ends of consecutive batches after the second one — the first two carry model construction, worker start-up,
    weight transforms and the hipGraph capture (the loop captures an iteration the second time it sees it).  The time stamps are
    taken in `on_attack_end_callback`, which also makes the loop score the unattacked batch (one extra forward pass per batch)."""
    import statistics
    cls, params = AttackEnum[attack].value
    data = SyntheticDetectionDataset(batch * batches)
    stamps = []

    def stamp(**_):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())

    set_seed(42)
    rep = generate_attacks([None, None, None], cfg(target), "cuda:0", attack_model_config=cfg(attack_model), attack_method=cls,
                           attack_params=params, batch_size=batch, dataset=data, share_weights=target == attack_model,
                           shuffle=False, num_workers=workers, on_attack_end_callback=stamp)
    gaps = [b - a for a, b in zip(stamps[1:-1], stamps[2:])]
    per_batch = statistics.median(gaps)
    print(f"{label:58s} B={batch:4d}  {batch / per_batch:9.1f} utt/s  {per_batch * 1e3:9.1f} ms/batch  "
          f"(median of {len(gaps)}, {min(gaps) * 1e3:.1f}-{max(gaps) * 1e3:.1f} ms; acc {rep['adv_eval/accuracy']:.1f} %)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=6)
    args = ap.parse_args()
    run("configs[1] LCNN+LFCC, PGD-40 eps 0.003 (white box)", "lcnn", "lcnn", "PGD40_eps003", 128, 2 * args.batches)
    run("configs[1] ... with 3 DataLoader workers (CLI default)", "lcnn", "lcnn", "PGD40_eps003", 128, 2 * args.batches, 3)
    run("configs[2] SpecRNet+mel, PGDL2-40 eps 0.1 (white box)", "specrnet_melspec", "specrnet_melspec", "PGDL2_40", 128, args.batches)
    run("configs[3] RawNet3 -> LCNN+LFCC, FGSM eps 0.0005", "lcnn", "rawnet3", "FGSM", 64, 2 * args.batches)
    run("configs[3] RawNet3 -> LCNN+LFCC, CW-100 c = 1", "lcnn", "rawnet3", "CW", 64, 5)
    run("           LCNN+LFCC, FAB (eta 10, 100 steps)", "lcnn", "lcnn", "FAB", 128, args.batches)


if __name__ == "__main__":
    main()
