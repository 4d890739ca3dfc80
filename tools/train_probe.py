#!/usr/bin/env python
"""Adversarial-training step throughput on one MI355X: LCNN + LFCC, B = 64 (the reference CLI's default batch), one
attack per step (ONLY_ADV strategy), synthetic utterances resident in HBM.

    python tools/train_probe.py [--batch 64] [--steps 6]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import trainer as T  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.models.models import get_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, dev).to(dev).train()
    x, y = synthetic_waveforms(a.batch)
    x, y = x.to(dev), y
    criterion = torch.nn.BCEWithLogitsLoss()
    optim = torch.optim.Adam(model.parameters(), lr=1e-4)
    for name in ("NO_ATTACK", "FGSM", "PGDL2", "PGD40_eps003", "FAB"):
        tr = T.OnlyOneAdversarialGDTrainer(device=dev, batch_size=a.batch)
        if name != "NO_ATTACK":
            tr.init_adv_attacks(model, [name])

        def step():
            bx = tr.apply_adv_attack(x.clone(), y).detach() if name != "NO_ATTACK" else x
            by = y.unsqueeze(1).float().to(dev)
            out, loss = T.forward_and_loss(model, criterion, bx, by)
            optim.zero_grad()
            loss.backward()
            optim.step()
            return loss

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        print(f"{name:14s} {ms:9.2f} ms / training step   {a.batch / ms * 1e3:8.1f} utt/s")


if __name__ == "__main__":
    main()
