"""TEST INFRASTRUCTURE — the CPU leg of bench.py: the oracle ("port" of the reference's CPU path) timed on a bounded
sample of the headline workload (PGD-40, LCNN + LFCC, T = 64 600), in its own process so the host-thread pool is
configured before any torch work and a CPU-side problem cannot take the GPU measurement down with it.

    python -m oracle.cpu_baseline --utterances 8 --threads 64
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=8)
    ap.add_argument("--threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    threads = a.threads or (os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    os.environ["HIP_VISIBLE_DEVICES"] = ""  # this leg must not touch the GPU

    import torch
    torch.set_num_threads(threads)
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    from oracle import attacks as oracle_attacks

    set_seed(42)
    cfg = {"frontend_algorithm": ["lfcc"], "input_channels": 1}
    target = get_model("lcnn", dict(cfg), "cpu").eval()
    attacked = get_model("lcnn", dict(cfg), "cpu").eval()
    attacked.load_state_dict(target.state_dict())
    x, y = synthetic_waveforms(a.utterances, 64_600, seed=1234)
    params = {"eps": 0.003, "alpha": 2 / 255, "steps": a.steps, "random_start": True}
    # untimed: one short pass so thread pools / oneDNN primitives are created
    oracle_attacks.attack_and_score(target, attacked, "PGD", dict(params, steps=1), x, y)
    t0 = time.perf_counter()
    oracle_attacks.attack_and_score(target, attacked, "PGD", params, x, y)
    dt = time.perf_counter() - t0
    print(json.dumps({
        "value": a.utterances / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{a.utterances} utterances (one batch) x PGD-{a.steps} LCNN+LFCC T=64600 via oracle/attacks.py "
                  f"(torch CPU ops in the reference's order), {dt:.1f} s on {os.cpu_count()} host hardware threads",
    }))


if __name__ == "__main__":
    main()
