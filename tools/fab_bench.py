#!/usr/bin/env python
"""FAB kernels at the benchmark shape (B = 128 utterances -> R = 256 projection rows, T = 64 600) + one whole FAB
iteration on LCNN + LFCC.

For scale, the Linf projection is also timed as the eager sort-based formulation the reference runs on a GPU
(argsort / gather / cumsum / searchsorted torch ops over the (2B, T) tensors, written here from the algorithm's
description — the reference itself cannot travel to the GPU box).

    python tools/fab_bench.py [--batch 128] [--launches 20] [--json out.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import hip_ops as ops  # noqa: E402

T = 64_600


def timed(fn, launches):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(launches):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / launches  # us


def eager_sorted_linf(t, w, b):
    """Sort-based water-filling in eager torch ops (what a GPU run of the reference's projection_linf amounts to)."""
    c = (w * t).sum(1) - b
    sg = torch.where(c >= 0, 1.0, -1.0).unsqueeze(1)
    w = w * sg
    beta = c.abs()
    up = w < 0
    room = torch.where(up, 1 - t, t)
    aw = w.abs()
    cap, order = torch.sort(room, dim=1)
    ws = aw.gather(1, order)
    filled = torch.cumsum(ws * cap, 1)
    tail = ws.flip(1).cumsum(1).flip(1)
    at_cap = filled - ws * cap + cap * tail
    k = torch.searchsorted(at_cap, beta.unsqueeze(1), right=True).clamp(max=t.shape[1] - 1)
    prev = torch.where(k > 0, filled.gather(1, (k - 1).clamp(min=0)), torch.zeros_like(beta).unsqueeze(1))
    lam = (beta.unsqueeze(1) - prev) / tail.gather(1, k)
    lam = torch.where(beta.unsqueeze(1) < filled[:, -1:], lam, torch.full_like(lam, float("inf")))
    d = torch.where(up, 1.0, -1.0) * torch.minimum(lam, room)
    return d * (w != 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--skip-model", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    g = torch.Generator(device=dev).manual_seed(0)
    x0 = (torch.randn(B, T, device=dev, generator=g) * 0.08 + 0.5).clamp(0, 1)       # min-max normalised audio: near 0.5
    x1 = (x0 + torch.randn(B, T, device=dev, generator=g) * 0.01).clamp(0, 1)
    gz = torch.randn(B, T, device=dev, generator=g) * 1e-3
    z = torch.randn(B, device=dev, generator=g) * 2
    la = (z > 0).long()
    pts = torch.cat((x1, x0))
    d3 = torch.empty_like(pts)
    adv = x0.clone()
    flags = torch.ones(B, dtype=torch.uint8, device=dev)
    out = {"batch": B, "T": T, "kernels": {}}

    def report(name, us, bytes_per_sample, rows):
        gbs = bytes_per_sample * rows * T / us / 1e3
        out["kernels"][name] = {"us": round(us, 1), "algorithmic_GBps": round(gbs, 1)}
        print(f"{name:34s} {us:9.1f} us   {gbs:8.1f} GB/s algorithmic ({bytes_per_sample} B/sample x {rows} rows)")

    for norm in ("Linf", "L2", "L1"):
        wscale, b, _, _ = ops.fab_hyperplane(gz, x1, z, la, norm)
        bb = b.repeat(2)
        report(f"fab_hyperplane[{norm}]", timed(lambda: ops.fab_hyperplane(gz, x1, z, la, norm), a.launches), 8, B)
        report(f"fab_projection[{norm}]", timed(lambda: ops.fab_projection(pts, gz, bb, norm, wscale, out=d3), a.launches),
               12, 2 * B)
        _, n3 = ops.fab_projection(pts, gz, bb, norm, wscale, out=d3)
        res2 = torch.full((B,), 1e10, device=dev)
        x1w = x1.clone()
        report(f"fab_backward_step[{norm}]",
               timed(lambda: ops.fab_backward_step(x1w, x0, adv, res2, flags, 0.9, norm), a.launches), 20, B)
    report("fab_combine", timed(lambda: ops.fab_combine(x1, x0, d3[:B], d3[B:], n3[:B], n3[B:], 1.05, 0.1, out=d3[:B]),
                                a.launches), 20, B)

    # eager sort-based Linf projection on the same rows
    wscale, b, _, _ = ops.fab_hyperplane(gz, x1, z, la, "Linf")
    wfull = (gz * wscale[:, None]).repeat(2, 1)
    bb = b.repeat(2)
    d_k, _ = ops.fab_projection(pts, gz, bb, "Linf", wscale)
    d_e = eager_sorted_linf(pts, wfull, bb)
    err = (d_k - d_e).abs().max().item()
    us = timed(lambda: eager_sorted_linf(pts, wfull, bb), max(3, a.launches // 4))
    out["eager_sorted_linf_us"] = round(us, 1)
    out["eager_vs_kernel_max_abs"] = err
    print(f"{'eager sort-based projection_linf':34s} {us:9.1f} us   (max |d_kernel - d_eager| = {err:.2e})")

    if not a.skip_model:
        from audio_deepfake_adversarial_attacks_amd import torchattacks
        from audio_deepfake_adversarial_attacks_amd.aa.utils import to_minmax
        from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
        from audio_deepfake_adversarial_attacks_amd.models.models import get_model
        torch.manual_seed(0)
        model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, "cuda:0").to(dev).eval()
        x, _ = synthetic_waveforms(B, T)
        x01, _, _ = to_minmax(x.to(dev))
        with torch.no_grad():
            y = (model(x01).reshape(-1) > 0).long()
        for steps in (5, 25):
            atk = torchattacks.FAB(model, n_classes=2, eta=10, steps=steps)
            atk.set_training_mode(True, False, False)
            atk(x01, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            atk(x01, y)
            torch.cuda.synchronize()
            out[f"fab_lcnn_steps{steps}_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        per_iter = (out["fab_lcnn_steps25_ms"] - out["fab_lcnn_steps5_ms"]) / 20
        out["fab_lcnn_ms_per_iteration"] = round(per_iter, 3)
        out["fab100_utt_per_s"] = round(B / ((out["fab_lcnn_steps5_ms"] - 5 * per_iter + 100 * per_iter) / 1e3), 1)
        print(f"FAB on LCNN+LFCC, B = {B}: {per_iter:.3f} ms / iteration (1 fwd+bwd, 1 fwd, 4 FAB kernels); "
              f"AttackEnum.FAB (100 steps): {out['fab100_utt_per_s']} utt/s")
    if a.json:
        Path(a.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
