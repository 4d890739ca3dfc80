#!/usr/bin/env python
"""Per-shape time of the 3x3 convolutions of SpecRNet (B = 128, mel-spec 80 x 404) through ATen / MIOpen: forward and
input gradient, HIP events, median of 20.  Decides which of them are worth a hand-written kernel."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch
import torch.nn.functional as F

SHAPES = [  # (name, Cin, Cout, H, W, kernel)
    ("block0.conv1", 2, 20, 80, 404, 3), ("block0.conv2", 20, 20, 80, 404, 3), ("block0.down", 2, 20, 80, 404, 1),
    ("block2.conv1", 20, 64, 20, 101, 3), ("block2.conv2", 64, 64, 20, 101, 3), ("block2.down", 20, 64, 20, 101, 1),
    ("block4.conv1", 64, 64, 5, 25, 3), ("block4.conv2", 64, 64, 5, 25, 3),
]
# every block is followed by its own MaxPool2d(2) AND the attention's pooling: 80x404 -> 20x101 -> 5x25

# the same layers through advstep_resconv_* (N, K1, K2, rows, H, W, pooled epilogue)
RESCONV = [
    ("block0 conv2 + down + pool   ", 20, 2, 20, 80, 404, True), ("block0 conv2^T               ", 20, 0, 20, 80, 404, False),
    ("block0 conv1 (2 -> 20)       ", 2, 0, 20, 80, 404, False), ("block0 conv1^T + down^T      ", 20, 20, 2, 80, 404, False),
    ("block2 conv1                 ", 20, 0, 64, 20, 101, False), ("block2 conv2 + down + pool   ", 64, 20, 64, 20, 101, True),
    ("block2 conv2^T               ", 64, 0, 64, 20, 101, False), ("block2 conv1^T + down^T      ", 64, 64, 20, 20, 101, False),
    ("block4 conv (64 -> 64)       ", 64, 0, 64, 5, 25, False),
]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = "cuda"
    for name, ci, co, H, W, k in SHAPES:
        x = torch.randn(B, ci, H, W, device=dev)
        w = torch.randn(co, ci, k, k, device=dev) * 0.1
        gy = torch.randn(B, co, H, W, device=dev)
        pad = k // 2
        fwd = timed(lambda: F.conv2d(x, w, None, 1, pad))
        bwd = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                                [True, False, False]))
        gflop = 2 * k * k * ci * co * B * H * W / 1e9
        mb = (ci + co) * B * H * W * 4 / 1e6
        print(f"{name:14s} {ci:3d}->{co:3d} {H}x{W} k{k}  fwd {fwd:7.1f} us ({gflop / fwd * 1e3:6.1f} TF/s, {mb / fwd * 1e3:6.0f} GB/s)   "
              f"bwd-data {bwd:7.1f} us ({gflop / bwd * 1e3:6.1f} TF/s)", flush=True)


def resconv(B):
    from audio_deepfake_adversarial_attacks_amd import detector_ops as D
    dev = "cuda"
    for name, k1, k2, rows, H, W, pool in RESCONV:
        x1 = torch.randn(B, k1, H, W, device=dev)
        x2 = torch.randn(B, k2, H, W, device=dev) if k2 else None
        U = D.resconv_prepare(torch.randn(rows, k1, 3, 3, device=dev) * 0.1, torch.randn(rows, k2, device=dev) if k2 else None)
        bias = torch.randn(rows, device=dev)
        fn = (lambda: D.resconv_pool2(x1, x2, U, rows, bias)) if pool else (lambda: D.resconv(x1, x2, U, rows, bias, 0.3))
        t = timed(fn)
        gflop = 2 * (9 * k1 + k2) * rows * B * H * W / 1e9
        print(f"resconv {name} K {k1:3d}+{k2:<3d} -> {rows:3d} {H}x{W}  {t:7.1f} us ({gflop / t * 1e3:6.1f} TF/s direct-equivalent)",
              flush=True)


def direct(B):
    """block0's 2-channel ends on the vector ALUs (csrc/detector_conv.hip)."""
    from audio_deepfake_adversarial_attacks_amd import detector_ops as D
    dev, H, W = "cuda", 80, 404
    x = torch.randn(B, 2, H, W, device=dev)
    w = torch.randn(20, 2, 3, 3, device=dev) * 0.1
    shift = torch.randn(20, device=dev)
    t = timed(lambda: D.conv3x3_fewin(x, w, shift, 0.3))
    print(f"direct  block0 conv1 (2 -> 20) + shift + lrelu                {t:7.1f} us ({(2 + 20) * B * H * W * 4 / t / 1e3:6.0f} GB/s)", flush=True)
    g1 = torch.randn(B, 20, H, W, device=dev)
    full = torch.randn(B, 20, H, W, device=dev)
    gp = torch.randn(B, 20, H // 2, W // 2, device=dev)
    _, sel = D._add_maxpool2_raw(full, None, None)
    wd = torch.randn(20, 2, device=dev)
    t = timed(lambda: D.conv3x3_fewout_grad(g1, w, gp, sel, wd))
    print(f"direct  block0 conv1^T + down^T (20 + pooled 20 -> 2)          {t:7.1f} us ({(20 + 5.25 + 2) * B * H * W * 4 / t / 1e3:6.0f} GB/s)", flush=True)


def pooled(B):
    """The input gradients that start from a pooled gradient + selection bytes (what an attack iteration runs)."""
    from audio_deepfake_adversarial_attacks_amd import detector_ops as D
    dev = "cuda"
    for name, K, rows, H, W in [("block0 conv2^T from pooled gy * lrelu'", 20, 20, 80, 404),
                                ("block2 conv2^T from pooled gy * lrelu'", 64, 64, 20, 101),
                                ("block4 conv2^T from pooled gy * lrelu'", 64, 64, 5, 25)]:
        full = torch.randn(B, K, H, W, device=dev)
        _, sel = D._add_maxpool2_raw(full, None, None)
        gy = torch.randn(B, K, H // 2, W // 2, device=dev)
        h = torch.randn(B, rows, H, W, device=dev)
        U = D.resconv_prepare(torch.randn(K, rows, 3, 3, device=dev) * 0.1, transpose=True)
        t = timed(lambda: D.resconv_pooled_grad(gy, sel, U, rows, H, W, h, 0.3))
        t0 = timed(lambda: D.resconv_pooled_grad(gy, sel, U, rows, H, W))
        act = torch.randint(0, 16, (B, rows, (H + 1) // 2, (W + 1) // 2), dtype=torch.uint8, device=dev)
        ta = timed(lambda: D.resconv_pooled_grad(gy, sel, U, rows, H, W, None, 0.3, act=act))
        gflop = 2 * 9 * K * rows * B * H * W / 1e9
        mb = (K * (H // 2) * (W // 2) * 5 + 2 * rows * H * W * 4) * B / 1e6
        print(f"pooled  {name} K {K:3d} -> {rows:3d} {H}x{W}  {t:7.1f} us ({gflop / t * 1e3:6.1f} TF/s direct-equivalent, "
              f"{mb / t * 1e3:6.0f} GB/s)   from sign bytes {ta:7.1f} us   without lrelu' {t0:7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "pooled":
        pooled(int(sys.argv[1]))
        resconv(int(sys.argv[1]))
        sys.exit(0)
    main()
    pooled(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
    resconv(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
    direct(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
