#!/usr/bin/env python
"""Mel-spec frontend (SpecRNet's input), forward + waveform-backward at B = 128, T = 64 600: torch op chain
(torch.stft + 2 matmuls + abs / angle + their autograd) vs the fused in-LDS-FFT kernels.

    python tools/mel_frontend_probe.py [--batch 128] [--iters 20]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd.frontends import MelSpecFrontend  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fe = MelSpecFrontend().to(dev)
    x = (torch.rand(a.batch, 64_600, device=dev) - 0.5)
    gy = None
    for name, flag in (("torch op chain", "0"), ("fused kernels", "1")):
        os.environ["ADVSTEP_FUSED_MEL"] = flag

        def step():
            nonlocal gy
            xa = x.clone().requires_grad_(True)
            y = fe(xa)
            if gy is None:
                gy = torch.randn_like(y)
            torch.autograd.grad(y, xa, gy)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            step()
        e1.record()
        torch.cuda.synchronize()
        print(f"mel-spec frontend fwd + bwd, B = {a.batch}: {name:16s} {e0.elapsed_time(e1) / a.iters * 1e3:9.1f} us")


if __name__ == "__main__":
    main()
