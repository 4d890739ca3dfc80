#!/usr/bin/env python
"""Kernel-level throughput of every libadvstep.so entry point at the benchmark shape (B = 128, T = 64 600).

Two regimes per kernel: "hot" re-uses one set of buffers (working set 66-165 MB: fits the 256 MB Infinity Cache,
which is the situation inside a PGD loop) and "cold" rotates over enough buffer sets to exceed it (true HBM rate).
Times come from one HIP-event pair around a burst of back-to-back launches on torch's current stream.
Also the target of the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE per launch; see profiles/README.md).

    python tools/kernel_microbench.py [--batch 128] [--launches 50] [--only pgd_linf_step] [--json out.json]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import hip_ops as ops  # noqa: E402

T = 64_600


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--sets", type=int, default=5, help="buffer sets for the cold regime")
    ap.add_argument("--only", default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, n_sets = a.batch, a.sets
    n = B * T
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(scale=1.0, shift=0.0):
        return [torch.rand(B, T, device=dev, generator=g) * scale + shift for _ in range(n_sets)]

    x, adv, out, w, m, v, best = rnd(), rnd(), rnd(), rnd(4.0, -2.0), rnd(0.1), rnd(0.01), rnd()
    grad = [torch.randn(B, T, device=dev, generator=g) * 1e-3 for _ in range(n_sets)]
    mn = torch.rand(B, 1, device=dev) - 1.0
    mx = mn + 1.5
    mask = (torch.arange(B, device=dev) % 2).float()
    r = torch.rand(B, device=dev)

    # name -> (callable(set index), algorithmic bytes per sample)
    cases = {
        "minmax_normalize": (lambda i: ops.to_minmax(x[i]), 12),
        "minmax_revert": (lambda i: ops.revert_minmax(x[i], mn, mx, out=out[i]), 8),
        "fgsm_step": (lambda i: ops.fgsm_step(x[i], grad[i], 1e-3, out=out[i]), 12),
        "pgd_linf_init_philox": (lambda i: ops.pgd_linf_init(x[i], 3e-3, seed=1, out=out[i]), 8),
        "pgd_linf_step": (lambda i: ops.pgd_linf_step(adv[i], grad[i], x[i], 2 / 255, 3e-3, out=out[i]), 16),
        "pgd_l2_init_philox": (lambda i: ops.pgd_l2_init(x[i], 0.1, seed=1, out=out[i]), 8),
        "pgd_l2_init_noise": (lambda i: ops.pgd_l2_init(x[i], 0.1, draws=(grad[i], r), out=out[i]), 12),
        "pgd_l2_step": (lambda i: ops.pgd_l2_step(adv[i], grad[i], x[i], 0.2, 0.1, out=out[i]), 16),
        "cw_init_w": (lambda i: ops.cw_init_w(x[i], out=out[i]), 8),
        "cw_tanh_sqdist": (lambda i: ops.cw_tanh_sqdist(w[i], x[i], adv_out=out[i]), 12),
        "cw_adam_step": (lambda i: ops.cw_adam_step(w[i], m[i], v[i], x[i], grad[i], 3), 32),
        "cw_best_update": (lambda i: ops.cw_best_update(adv[i], mask, best[i]), 12),
    }
    # LCNN max-feature-map kernels at the first (largest) layer's shape: conv output (B, 64, 404, 80)
    from audio_deepfake_adversarial_attacks_amd import _lib, lcnn_ops  # noqa: F401
    lib = _lib.load()
    C, H, W = 32, 404, 80
    nsets2 = 2  # 1.06 GB per conv-output tensor: two sets already exceed the Infinity Cache
    cx = [torch.randn(B, 2 * C, H, W, device=dev, generator=g) for _ in range(nsets2)]
    cbias = torch.randn(2 * C, device=dev)
    y_m = torch.empty(B, C, H, W, device=dev)
    sel = torch.empty(lib.advstep_mfm_sel_bytes(B, C, H * W), dtype=torch.uint8, device=dev)
    y_p = torch.empty(B, C, H // 2, W // 2, device=dev)
    idx = torch.empty(y_p.numel(), dtype=torch.uint8, device=dev)
    gxb = [torch.empty(B, 2 * C, H, W, device=dev) for _ in range(nsets2)]
    stream = torch.cuda.current_stream().cuda_stream
    lib.advstep_mfm_forward_f32(cx[0].data_ptr(), None, None, None, y_m.data_ptr(), sel.data_ptr(), B, C, H * W, stream)
    lib.advstep_mfm_pool2_forward_f32(cx[0].data_ptr(), None, None, None, y_p.data_ptr(), idx.data_ptr(), B, C, H, W, stream)
    nm, npool = y_m.numel(), y_p.numel()
    lcnn_cases = {
        "mfm_forward(+bias)": (lambda i: lib.advstep_mfm_forward_f32(cx[i % nsets2].data_ptr(), cbias.data_ptr(), None, None,
                                                                     y_m.data_ptr(), sel.data_ptr(), B, C, H * W, stream),
                               nm * 12.25),
        "mfm_backward": (lambda i: lib.advstep_mfm_backward_f32(y_m.data_ptr(), sel.data_ptr(), None, gxb[i % nsets2].data_ptr(), B,
                                                                C, H * W, stream), nm * 12.25),
        "mfm_pool2_forward(+bias)": (lambda i: lib.advstep_mfm_pool2_forward_f32(cx[i % nsets2].data_ptr(), cbias.data_ptr(),
                                                                                 None, None, y_p.data_ptr(), idx.data_ptr(),
                                                                                 B, C, H, W, stream), npool * 37.0),
        "mfm_pool2_backward": (lambda i: lib.advstep_mfm_pool2_backward_f32(y_p.data_ptr(), idx.data_ptr(), None,
                                                                            gxb[i % nsets2].data_ptr(), B, C, H, W, stream),
                               npool * 37.0),
    }
    results = {}
    for name, (fn, total_bytes) in lcnn_cases.items():
        if a.only and a.only != name:
            continue
        row = {"algorithmic_bytes_per_launch": total_bytes}
        for regime, sets in (("hot", 1), ("cold", nsets2)):
            for i in range(2):
                fn(i % sets)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                fn(i % sets)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row[regime + "_us"] = 1e3 * ms
            row[regime + "_GBps"] = total_bytes / (ms * 1e-3) / 1e9
        results[name] = row
        print(f"{name:24s} hot {row['hot_us']:8.1f} us {row['hot_GBps']:7.0f} GB/s | cold {row['cold_us']:8.1f} us "
              f"{row['cold_GBps']:7.0f} GB/s  (B x 64 x 404 x 80 conv output; {total_bytes / 1e6:.0f} MB algorithmic)", flush=True)
    del cx, gxb, y_m, y_p
    for name, (fn, bytes_per_sample) in cases.items():
        if a.only and a.only != name:
            continue
        row = {"algorithmic_bytes_per_launch": bytes_per_sample * n}
        for regime, sets in (("hot", 1), ("cold", n_sets)):
            for i in range(3):
                fn(i % sets)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.launches):
                fn(i % sets)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.launches
            row[regime + "_us"] = 1e3 * ms
            row[regime + "_GBps"] = bytes_per_sample * n / (ms * 1e-3) / 1e9
        results[name] = row
        print(f"{name:24s} hot {row['hot_us']:8.1f} us {row['hot_GBps']:7.0f} GB/s | cold {row['cold_us']:8.1f} us "
              f"{row['cold_GBps']:7.0f} GB/s  ({bytes_per_sample} B/sample algorithmic)", flush=True)
    if a.json:
        Path(a.json).write_text(json.dumps({"batch": B, "T": T, "launches": a.launches, "kernels": results}, indent=1))


if __name__ == "__main__":
    main()
