#!/usr/bin/env python
"""Per-layer cost of LCNN's convolutions under PyTorch-ROCm / MIOpen at the benchmark shape (B = 128):
forward and input-backward (bwd-data) of each Conv2d, timed with HIP events over back-to-back calls."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

LAYERS = [  # name, Cin, Cout, k, pad, H, W
    ("L0  5x5   1->64  404x80", 1, 64, 5, 2, 404, 80),
    ("L3  1x1  32->64  202x40", 32, 64, 1, 0, 202, 40),
    ("L6  3x3  32->96  202x40", 32, 96, 3, 1, 202, 40),
    ("L10 1x1  48->96  101x20", 48, 96, 1, 0, 101, 20),
    ("L13 3x3  48->128 101x20", 48, 128, 3, 1, 101, 20),
    ("L16 1x1  64->128  50x10", 64, 128, 1, 0, 50, 10),
    ("L19 3x3  64->64   50x10", 64, 64, 3, 1, 50, 10),
    ("L22 1x1  32->64   50x10", 32, 64, 1, 0, 50, 10),
    ("L25 3x3  32->64   50x10", 32, 64, 3, 1, 50, 10),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    tot_f = tot_b = 0.0
    for name, cin, cout, k, pad, H, W in LAYERS:
        x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
        w = torch.randn(cout, cin, k, k, device=dev)
        b = torch.randn(cout, device=dev)
        y = F.conv2d(x, w, b, 1, pad)
        gy = torch.randn_like(y)
        f_us = timeit(lambda: F.conv2d(x, w, b, 1, pad))
        fnb_us = timeit(lambda: F.conv2d(x, w, None, 1, pad))
        b_us = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
        flops = 2.0 * B * H * W * cin * cout * k * k
        tot_f += fnb_us
        tot_b += b_us
        print(f"{name}: fwd {f_us:8.1f} us (no bias {fnb_us:8.1f} us = {flops / fnb_us / 1e6:6.1f} TFLOP/s) | "
              f"bwd-data {b_us:8.1f} us ({flops / b_us / 1e6:6.1f} TFLOP/s) | {flops / 1e9:6.2f} GFLOP "
              f"out {B * cout * H * W * 4 / 1e6:7.1f} MB", flush=True)
    print(f"total fwd (no bias) {tot_f / 1e3:.2f} ms, bwd-data {tot_b / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
