#!/usr/bin/env python
"""The priced kernel of a bench.py workload, launched over rotating buffer sets — the target of bench.py's own
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (`roofline.traffic`, measured in the run) and of its cold-regime
timing (`roofline.frac_cold`).  5 sets of (B, 64600) float32 operands: 0.5-0.8 GB, beyond the 256 MiB Infinity Cache, so a
launch's reads come from HBM.

    python tools/traffic_probe.py --entry pgd_linf_step --batch 128 [--launches 10] [--sets 5]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

T = 64_600


def launcher(entry: str, B: int, sets: int, device):
    """callable(i): one launch of `entry` on buffer set i % sets."""
    from audio_deepfake_adversarial_attacks_amd import hip_ops as ops
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda scale=1.0, shift=0.0: [torch.rand(B, T, device=device, generator=g) * scale + shift for _ in range(sets)]
    grad = [torch.randn(B, T, device=device, generator=g) * 1e-3 for _ in range(sets)]
    if entry == "pgd_linf_step":
        x, adv, out = rnd(), rnd(), rnd()
        return lambda i: ops.pgd_linf_step(adv[i % sets], grad[i % sets], x[i % sets], 2 / 255, 3e-3, out=out[i % sets])
    if entry == "pgd_l2_step":
        x, adv, out = rnd(), rnd(), rnd()
        return lambda i: ops.pgd_l2_step(adv[i % sets], grad[i % sets], x[i % sets], 0.2, 0.1, out=out[i % sets])
    if entry == "cw_adam_step":
        x, w, m, v = rnd(), rnd(4.0, -2.0), rnd(0.1), rnd(0.01)
        return lambda i: ops.cw_adam_step(w[i % sets], m[i % sets], v[i % sets], x[i % sets], grad[i % sets], 3)
    raise SystemExit(f"unknown entry point {entry!r}")


def cold_launch_ms(entry: str, B: int, device, launches: int = 30, sets: int = 5) -> float:
    """Average HIP-event bracket of ONE launch (events on the launch stream, one pair per launch — the same clock as
    bench.py's hot `avg_launch_ms`) with the operands rotated past the Infinity Cache."""
    fn = launcher(entry, B, sets, device)
    for i in range(sets):
        fn(i)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(device)
    pairs = []
    for i in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fn(i)
        b.record(stream)
        pairs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entry", required=True)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--sets", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fn = launcher(a.entry, a.batch, a.sets, dev)
    for i in range(a.launches):
        fn(i)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
