#!/usr/bin/env python
"""A/B of the hipGraph replay of the PGD / PGDL2 inner loop (torchattacks/graphed.py) against eager launches, interleaved in
one process: wall time per batch of the loop body and the launching thread's CPU time per batch.

    python tools/graph_probe.py [--config 1|2] [--batches 4] [--rounds 3]
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import attack_batch, score_batch
    dev = torch.device("cuda:0")
    spec = bench.WORKLOADS[a.config]
    target, attacked = bench.build_models(spec, dev)
    cls, params = AttackEnum[spec["attacks"][0]].value
    atk = cls(attacked, **params)
    atk.set_training_mode(model_training=True, batchnorm_training=False)
    B = spec["batch"]
    x, y = synthetic_waveforms(B * a.batches, bench.T, seed=1234)
    x, y = x.to(dev), y.to(dev)

    def run(mode):
        os.environ["ADVSTEP_ATTACK_GRAPH"] = mode
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        for i in range(a.batches):
            adv = attack_batch(atk, x[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
            score_batch(target, adv)
        c1 = time.process_time()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        return (t1 - t0) / a.batches * 1e3, (c1 - c0) / a.batches * 1e3

    for mode in ("0", "1", "1"):            # warm both paths (the graph is captured on the second sight of a key)
        run(mode)
    for r in range(a.rounds):
        for mode, name in (("0", "eager"), ("1", "graph")):
            wall, cpu = run(mode)
            print(f"round {r} {name:6s} {wall:8.2f} ms/batch wall   {cpu:8.2f} ms/batch launching-thread CPU   "
                  f"{B / wall * 1e3:8.1f} utt/s", flush=True)


if __name__ == "__main__":
    main()
