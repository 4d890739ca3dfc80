#!/usr/bin/env python
"""Microbenchmark of the waveform-batch kernels at the BASELINE batch: B = 128 utterances cut /
tiled to 64 600 samples.  Prints the kernel time (HIP events on the launch stream), the algorithmic HBM rate
(bytes written + payload bytes read once) and, beside it, the reference's way of doing the same step —
`wavefake_preprocessing_on_batch`'s batch.cpu() -> per-row apply_pad -> stack -> .to(device) round trip
(src/datasets/base_dataset.py:122-148) restated with torch CPU ops."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from audio_deepfake_adversarial_attacks_amd.datasets import wave_ops

B, CUT = 128, 64_600


def timed(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters * 1e3  # us


def reference_round_trip(batch):
    rows = []
    for row in batch.cpu():
        w = row.unsqueeze(0).squeeze(0)
        n = w.shape[0]
        rows.append(w[:CUT] if n >= CUT else torch.tile(w.unsqueeze(0), (1, int(CUT / n) + 1))[:, :CUT][0])
    return torch.stack(rows).to(batch.device)


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    lengths = rng.integers(16_000, 96_000, B)
    for name, dtype in (("pcm16 ragged", np.int16), ("f32 ragged", np.float32)):
        waves = [rng.integers(-3000, 3000, n).astype(np.int16) if dtype == np.int16
                 else (rng.standard_normal(n) * 0.1).astype(np.float32) for n in lengths]
        ragged = wave_ops.RaggedWaveBatch.from_arrays(waves).pin_memory()
        payload, offsets, lens = ragged.payload.to(dev), ragged.offsets.to(dev), ragged.lengths.to(dev)
        us = timed(lambda: wave_ops.pad_tile(payload, offsets, lens, CUT))
        read = int(np.minimum(lengths, CUT).sum()) * payload.element_size()
        total = read + B * CUT * 4
        print(f"wave_pad_tile {name:14s} B={B} cut={CUT}: {us:8.1f} us  {total / us / 1e3:7.0f} GB/s "
              f"({total / 1e6:.1f} MB algorithmic)")
        t0 = time.perf_counter()
        for _ in range(5):
            out = ragged.to_padded(dev, CUT)
        torch.cuda.synchronize()
        print(f"  incl. H2D of the pinned payload ({ragged.payload.numel() * ragged.payload.element_size() / 1e6:.1f} MB): "
              f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms / batch")
    batch = torch.randn(B, CUT, device=dev)
    us = timed(lambda: wave_ops.apply_pad_batch(batch, CUT))
    print(f"wave_pad_tile (B, T) identity   B={B} cut={CUT}: {us:8.1f} us  {2 * B * CUT * 4 / us / 1e3:7.0f} GB/s")
    short = batch[:, :20_000].contiguous()
    us = timed(lambda: wave_ops.apply_pad_batch(short, CUT))
    print(f"wave_pad_tile (B, 20000) tiled  B={B} cut={CUT}: {us:8.1f} us  {(B * 20_000 + B * CUT) * 4 / us / 1e3:7.0f} GB/s")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ref = reference_round_trip(short)
    torch.cuda.synchronize()
    print(f"reference-style CPU round trip of the same batch: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms "
          f"(equal: {torch.equal(ref, wave_ops.apply_pad_batch(short, CUT))})")
    y = torch.randint(0, 2, (B,), device=dev)
    clean = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32)
    attacked = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32)
    us = timed(lambda: wave_ops.qual_select(y, clean, attacked))
    rows, counts = wave_ops.qual_select(y, clean, attacked)
    n = int(counts.sum())
    print(f"qual_select B={B}: {us:.1f} us ({n} rows flipped)")
    us = timed(lambda: wave_ops.gather_rows(batch, rows, n))
    print(f"wave_gather_rows n={n}: {us:.1f} us  {2 * n * CUT * 4 / us / 1e3:.0f} GB/s")


if __name__ == "__main__":
    main()
