#!/usr/bin/env python
"""A/B probe of ONE PGD iteration's model work (LCNN + LFCC forward + input-backward, B = 128, T = 64 600) under
different settings: plain ATen max-feature-map vs the HIP kernels, frozen parameters (bias folding), MIOpen
benchmark (find) mode.  Prints ms per iteration for each arm (interleaved repeats).

    python tools/lcnn_step_probe.py [--batch 128] [--iters 10] [--repeats 3] [--arms plain,fused,fused_frozen,...]
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--arms", default="plain,fused_frozen_miopen3x3,fused_frozen")
    a = ap.parse_args()
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    dev = torch.device("cuda:0")
    set_seed(42)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, "cuda:0").to(dev)
    model.train()
    for m in model.modules():
        if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
            m.eval()
    x = (torch.randn(a.batch, 64_600, generator=torch.Generator().manual_seed(1)) * 0.05).clamp(-1, 1).to(dev)
    x = (x - x.amin(1, keepdim=True)) / (x.amax(1, keepdim=True) - x.amin(1, keepdim=True))

    def configure(arm):
        os.environ["ADVSTEP_LCNN_FUSED"] = "0" if arm.startswith("plain") else "1"
        os.environ["ADVSTEP_LCNN_LSTM"] = "0" if ("nolstm" in arm or "noconv0" in arm or "no1x1" in arm) else "1"
        os.environ["ADVSTEP_FUSED_LFCC"] = "0" if ("nolfcc" in arm or arm.startswith("plain")) else "1"
        os.environ["ADVSTEP_FUSED_STFT"] = "0" if "nostft" in arm else "1"
        os.environ["ADVSTEP_DIRECT_FFT"] = "0" if "torchfft" in arm else "1"
        os.environ["ADVSTEP_INLDS_FFT"] = "0" if ("hipfft" in arm or "torchfft" in arm) else "1"
        os.environ["ADVSTEP_LCNN_BN"] = "0" if "nobn" in arm else "1"
        os.environ["ADVSTEP_LCNN_CONV0"] = "0" if "noconv0" in arm else "1"
        os.environ["ADVSTEP_LCNN_CONV3X3"] = "0" if ("miopen3x3" in arm or arm.startswith("plain")) else "1"
        os.environ["ADVSTEP_LCNN_CONV1X1"] = "0" if ("no1x1" in arm or "noconv0" in arm) else "1"
        frozen = "frozen" in arm
        for p in model.parameters():
            p.requires_grad_(not frozen)
        torch.backends.cudnn.benchmark = "benchmark" in arm
        torch.backends.cudnn.deterministic = "benchmark" not in arm

    def one_iter():
        adv = x.clone().requires_grad_(True)
        z = model(adv)
        (g,) = torch.autograd.grad(z, adv, grad_outputs=torch.ones_like(z))
        return g

    graphs = {}

    def runner(arm):
        """`_graph`: the whole iteration replayed from a captured HIP graph; `_cl`: channels_last trunk."""
        if arm.endswith("_cl"):
            model.m_transform.to(memory_format=torch.channels_last)
        else:
            model.m_transform.to(memory_format=torch.contiguous_format)
        if not arm.endswith("_graph"):
            return one_iter
        if arm not in graphs:
            static_x = x.clone().requires_grad_(True)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    z = model(static_x)
                    torch.autograd.grad(z, static_x, grad_outputs=torch.ones_like(z))
            torch.cuda.current_stream().wait_stream(s)
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                z = model(static_x)
                (gr,) = torch.autograd.grad(z, static_x, grad_outputs=torch.ones_like(z))
            graphs[arm] = (gph, gr)
        gph, gr = graphs[arm]

        def replay():
            gph.replay()
            return gr
        return replay

    arms = a.arms.split(",")
    results = {arm: [] for arm in arms}
    runners = {}
    for arm in arms:  # warm-up (MIOpen solver selection / kernel compilation) per arm
        configure(arm)
        try:
            runners[arm] = runner(arm)
            for _ in range(3):
                runners[arm]()
            torch.cuda.synchronize()
        except Exception as e:  # e.g. an op that cannot be captured
            print(f"{arm}: unavailable ({type(e).__name__}: {str(e)[:200]})", flush=True)
            runners[arm] = None
    arms = [arm for arm in arms if runners[arm] is not None]
    for _ in range(a.repeats):
        for arm in arms:
            configure(arm)
            runner(arm)
            one_iter = runners[arm]
            one_iter()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                one_iter()
            torch.cuda.synchronize()
            results[arm].append(1e3 * (time.perf_counter() - t0) / a.iters)
    for arm, v in results.items():
        print(f"{arm:28s} {min(v):8.3f} ms/iter (min of {len(v)}; all: {', '.join(f'{t:.3f}' for t in v)})", flush=True)


if __name__ == "__main__":
    main()
