#!/usr/bin/env python
"""What two batches in flight can and cannot overlap (round 6): pairs of the LCNN iteration's kernels at B = 128, each looping
on a stream of its own, alone and together.  `together / (alone_a + alone_b)` = 1.0 means the two simply take turns (no
overlap), 0.5 * (1 + min / max) ... means the shorter one ran entirely under the longer.

    python tools/coresidency_probe.py [--reps 40]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--only-wino", action="store_true", help="just the pairs with a Winograd kernel (ADVSTEP_WINO_RANGE_MULT A/B)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B = a.batch
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    kernels = {}

    # LSTM forward / backward (T = 25, H = 80, 2 directions): 256 workgroups of 320 threads, 25 dependent steps
    T, Hh, D = 25, 80, 2
    gxl = torch.randn(T, B, D, 4 * Hh, device=dev)
    whh = torch.randn(D, 4 * Hh, Hh, device=dev) * 0.1
    out = torch.empty(T, B, D * Hh, device=dev)
    gates, cell = torch.empty(T, B, D, 4 * Hh, device=dev), torch.empty(T, B, D, Hh, device=dev)
    dgx = torch.empty_like(gxl)
    kernels["lstm_forward"] = lambda st: lib.advstep_lstm_forward_f32(
        gxl.data_ptr(), whh.data_ptr(), out.data_ptr(), gates.data_ptr(), cell.data_ptr(), T, B, D, Hh, st)
    kernels["lstm_backward"] = lambda st: lib.advstep_lstm_backward_f32(
        out.data_ptr(), whh.data_ptr(), gates.data_ptr(), cell.data_ptr(), dgx.data_ptr(), T, B, D, Hh, st)
    # the same kernel on a fraction of the chip: B / 2 and B / 4 utterances = 128 / 64 workgroups (what a recurrent kernel with
    # two / four (utterance, direction) pairs per workgroup would occupy)
    for div in (2, 4):
        kernels[f"lstm_forward_{256 // div}wg"] = (lambda st, n=B // div: lib.advstep_lstm_forward_f32(
            gxl.data_ptr(), whh.data_ptr(), out.data_ptr(), gates.data_ptr(), cell.data_ptr(), T, n, D, Hh, st))

    # Winograd L6 forward (32 -> 96, 202 x 40) and its compact-source input gradient
    cin, c, h, wd = 32, 48, 202, 40
    xx = torch.randn(B, cin, h, wd, device=dev)
    ww = torch.randn(2 * c, cin, 3, 3, device=dev) * 0.1
    bb = torch.randn(2 * c, device=dev)
    mean, invstd = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
    yy = torch.empty(B, c, h // 2, wd // 2, device=dev)
    ii = torch.empty(yy.numel(), dtype=torch.uint8, device=dev)
    st0 = torch.cuda.current_stream().cuda_stream
    u0 = torch.empty(lib.advstep_conv3x3_prepared_floats(cin, 2 * c, 0), device=dev)
    u2 = torch.empty(lib.advstep_conv3x3_prepared_floats(cin, 2 * c, 2), device=dev)
    lib.advstep_conv3x3_prepare_f32(ww.data_ptr(), None, u0.data_ptr(), cin, 2 * c, 0, st0)
    lib.advstep_conv3x3_prepare_f32(ww.data_ptr(), invstd.data_ptr(), u2.data_ptr(), cin, 2 * c, 2, st0)
    gxx = torch.empty_like(xx)
    xx2, yy2, ii2, gxx2 = torch.randn_like(xx), torch.empty_like(yy), torch.empty_like(ii), torch.empty_like(xx)
    kernels["wino_L6_forward"] = lambda st: lib.advstep_conv3x3_mfm_pool2_forward_f32(
        xx.data_ptr(), u0.data_ptr(), bb.data_ptr(), mean.data_ptr(), invstd.data_ptr(), yy.data_ptr(), ii.data_ptr(), B, cin, c, h, wd, st)
    kernels["wino_L6_forward_2"] = lambda st: lib.advstep_conv3x3_mfm_pool2_forward_f32(
        xx2.data_ptr(), u0.data_ptr(), bb.data_ptr(), mean.data_ptr(), invstd.data_ptr(), yy2.data_ptr(), ii2.data_ptr(), B, cin, c, h, wd, st)
    kernels["wino_L6_backward"] = lambda st: lib.advstep_conv3x3_mfm_pool2_backward_f32(
        yy.data_ptr(), ii.data_ptr(), u2.data_ptr(), gxx.data_ptr(), B, cin, c, h, wd, st)

    # 1x1 block L3 forward, first block forward
    P = 202 * 40
    x1 = torch.randn(B, 32, 202, 40, device=dev)
    w1 = torch.randn(64, 32, 1, 1, device=dev) * 0.1
    b1 = torch.randn(64, device=dev)
    y1 = torch.empty(B, 32, 202, 40, device=dev)
    sel = torch.empty(lib.advstep_conv1x1_mfm_sel_bytes(B, 32, P), dtype=torch.uint8, device=dev)
    kernels["conv1x1_L3_forward"] = lambda st: lib.advstep_conv1x1_mfm_forward_f32(
        x1.data_ptr(), w1.data_ptr(), b1.data_ptr(), None, None, y1.data_ptr(), sel.data_ptr(), B, 32, 32, P, st)
    x0 = torch.randn(B, 1, 404, 80, device=dev)
    w0 = torch.randn(64, 1, 5, 5, device=dev) * 0.2
    b0 = torch.randn(64, device=dev)
    y0 = torch.empty(B, 32, 202, 40, device=dev)
    i0 = torch.empty(y0.numel(), dtype=torch.uint8, device=dev)
    kernels["conv5_forward"] = lambda st: lib.advstep_conv5_mfm_pool2_forward_f32(
        x0.data_ptr(), w0.data_ptr(), b0.data_ptr(), y0.data_ptr(), i0.data_ptr(), B, 32, 404, 80, st)
    torch.cuda.synchronize()

    def run(pairs):
        """pairs: [(stream, kernel name, launches)]; wall time in ms of all of them queued at once"""
        for stream, name, n in pairs:
            for _ in range(2):
                kernels[name](stream.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        e0.record(cur)
        for stream, _, _ in pairs:
            stream.wait_event(e0)
        for i in range(max(n for _, _, n in pairs)):
            for stream, name, n in pairs:
                if i < n:
                    kernels[name](stream.cuda_stream)
        for stream, _, _ in pairs:
            cur.wait_stream(stream)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    solo = {}
    for name in kernels:
        solo[name] = run([(sa, name, a.reps)]) / a.reps
        print(f"{name:24s} alone {solo[name] * 1e3:8.1f} us", flush=True)
    print()
    for na, nb in (("lstm_forward_128wg", "wino_L6_forward"), ("lstm_forward_64wg", "wino_L6_forward"),
                   ("lstm_forward_128wg", "conv5_forward"), ("lstm_forward_128wg", "conv1x1_L3_forward"),
                   ("lstm_forward", "wino_L6_forward"), ("lstm_backward", "wino_L6_backward"), ("lstm_forward", "conv1x1_L3_forward"),
                   ("lstm_forward", "conv5_forward"), ("wino_L6_forward", "wino_L6_forward_2"), ("wino_L6_forward", "conv1x1_L3_forward"),
                   ("wino_L6_forward", "conv5_forward"), ("conv5_forward", "conv1x1_L3_forward"), ("lstm_forward", "lstm_backward")):
        if a.only_wino and "wino" not in na + nb:
            continue
        # equal total work on both streams: launch counts in inverse proportion to the solo durations
        ta, tb = solo[na], solo[nb]
        ra = a.reps
        rb = max(1, round(a.reps * ta / tb))
        both = run([(sa, na, ra), (sb, nb, rb)])
        serial = ra * ta + rb * tb
        print(f"{na:20s} x{ra:3d} || {nb:20s} x{rb:3d}: together {both:8.2f} ms, one after the other {serial:8.2f} ms "
              f"-> {both / serial:5.2f} of serial (0.50 = perfect overlap)", flush=True)


if __name__ == "__main__":
    main()
