#!/usr/bin/env python
"""First block's input gradient: the patch-centric gather kernel against the cell-centric kernel (round 5) at LCNN's shape,
B = 128 - same inputs, both timed with HIP events, outputs compared with each other and with float64 autograd on a slice.

    python tools/conv0_bwd_probe.py [--batch 128] [--launches 20]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--launches", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, H, W, C = a.batch, 404, 80, 32
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 1, H, W, generator=g).to(dev)
    w = (torch.randn(2 * C, 1, 5, 5, generator=g) * 0.2).to(dev)
    b = torch.randn(2 * C, generator=g).to(dev)
    y = torch.empty(B, C, H // 2, W // 2, device=dev)
    idx = torch.empty(y.numel(), dtype=torch.uint8, device=dev)
    lib.advstep_conv5_mfm_pool2_forward_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), idx.data_ptr(), B, C, H, W, st)
    gy = torch.randn(y.shape, generator=g).to(dev)
    outs = {}
    for mode in ("gather", "cells", "gather", "cells"):
        os.environ["ADVSTEP_CONV0_BWD"] = mode
        gx = torch.full_like(x, float("nan"))
        fn = lambda: lib.advstep_conv5_mfm_pool2_backward_f32(gy.data_ptr(), idx.data_ptr(), w.data_ptr(), gx.data_ptr(), B, C, H, W, st)
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{mode:8s} {e0.elapsed_time(e1) / a.launches * 1e3:8.1f} us", flush=True)
        outs[mode] = gx
    d = (outs["gather"] - outs["cells"]).abs().max().item()
    print("max |gather - cells| =", d, " scale", outs["gather"].abs().max().item(), " finite", bool(torch.isfinite(outs["cells"]).all()))
    # float64 autograd on two utterances
    xs = x[:2].double().requires_grad_(True)
    conv = torch.nn.functional.conv2d(xs, w.double(), b.double(), padding=2)
    m = torch.maximum(conv[:, :C], conv[:, C:])
    yr = torch.nn.functional.max_pool2d(m, 2, 2)
    (gr,) = torch.autograd.grad(yr, xs, gy[:2].double())
    for mode in ("gather", "cells"):
        print(mode, "vs float64:", (outs[mode][:2].double() - gr).abs().max().item())


if __name__ == "__main__":
    main()
