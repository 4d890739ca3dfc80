#!/usr/bin/env python
"""Per-kernel timings of the model-side kernels (include/advstep_lcnn.h, advstep_frontend.h) at LCNN's layer shapes,
B = 128, through the C ABI, HIP events around bursts of launches.

    python tools/model_kernel_bench.py [--launches 20] [--json out.json]
"""
import argparse
import json
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
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B = a.batch
    st = torch.cuda.current_stream().cuda_stream
    res = {}

    def timeit(name, fn, bytes_moved=None, flops=None):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.launches * 1e3
        row = {"us": us}
        extra = ""
        if bytes_moved:
            row["GBps"] = bytes_moved / us / 1e3
            extra += f" {row['GBps']:7.0f} GB/s"
        if flops:
            row["TFLOPs"] = flops / us / 1e6
            extra += f" {row['TFLOPs']:6.1f} TFLOP/s"
        res[name] = row
        print(f"{name:44s} {us:8.1f} us{extra}", flush=True)

    # ---- first block
    H, W, C = 404, 80, 32
    x = torch.randn(B, 1, H, W, device=dev)
    w = torch.randn(2 * C, 1, 5, 5, device=dev) * 0.2
    b = torch.randn(2 * C, device=dev)
    y = torch.empty(B, C, H // 2, W // 2, device=dev)
    idx = torch.empty(y.numel(), dtype=torch.uint8, device=dev)
    gx = torch.empty_like(x)
    fl = 2.0 * B * H * W * 2 * C * 25
    timeit("conv5_mfm_pool2_forward  (1->64, 404x80)", lambda: lib.advstep_conv5_mfm_pool2_forward_f32(
        x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), idx.data_ptr(), B, C, H, W, st), flops=fl)
    timeit("conv5_mfm_pool2_backward (1->64, 404x80)", lambda: lib.advstep_conv5_mfm_pool2_backward_f32(
        y.data_ptr(), idx.data_ptr(), w.data_ptr(), gx.data_ptr(), B, C, H, W, st))

    # ---- 1x1 blocks
    for name, cin, c, h, wd in (("L3 ", 32, 32, 202, 40), ("L10", 48, 48, 101, 20), ("L16", 64, 64, 50, 10), ("L22", 32, 32, 50, 10)):
        P = h * wd
        xx = torch.randn(B, cin, h, wd, device=dev)
        ww = torch.randn(2 * c, cin, 1, 1, device=dev) * 0.1
        bb = torch.randn(2 * c, device=dev)
        mean, invstd = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
        yy = torch.empty(B, c, h, wd, device=dev)
        sel = torch.empty(lib.advstep_conv1x1_mfm_sel_bytes(B, c, P), dtype=torch.uint8, device=dev)
        gxx = torch.empty_like(xx)
        moved = 4.0 * B * P * (cin + c)
        timeit(f"conv1x1_mfm_forward  {name} ({cin}->{2 * c}, {h}x{wd})", lambda: lib.advstep_conv1x1_mfm_forward_f32(
            xx.data_ptr(), ww.data_ptr(), bb.data_ptr(), mean.data_ptr(), invstd.data_ptr(), yy.data_ptr(), sel.data_ptr(),
            B, cin, c, P, st), bytes_moved=moved, flops=2.0 * B * P * cin * 2 * c)
        timeit(f"conv1x1_mfm_backward {name} ({cin}->{2 * c}, {h}x{wd})", lambda: lib.advstep_conv1x1_mfm_backward_f32(
            yy.data_ptr(), sel.data_ptr(), ww.data_ptr(), invstd.data_ptr(), gxx.data_ptr(), B, cin, c, P, st),
            bytes_moved=moved)

    # ---- 3x3 blocks on the matrix cores (Winograd): conv + bias + MFM + pool (+BN) forward, input-gradient convolution
    for name, cin, c, h, wd in (("L6 ", 32, 48, 202, 40), ("L13", 48, 64, 101, 20), ("L25", 32, 32, 50, 10)):
        xx = torch.randn(B, cin, h, wd, device=dev)
        ww = torch.randn(2 * c, cin, 3, 3, device=dev) * 0.1
        bb = torch.randn(2 * c, device=dev)
        mean, invstd = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
        yy = torch.empty(B, c, h // 2, wd // 2, device=dev)
        ii = torch.empty(yy.numel(), dtype=torch.uint8, device=dev)
        u0 = torch.empty(lib.advstep_conv3x3_prepared_floats(cin, 2 * c, 0), device=dev)
        u1 = torch.empty(lib.advstep_conv3x3_prepared_floats(cin, 2 * c, 1), device=dev)
        lib.advstep_conv3x3_prepare_f32(ww.data_ptr(), None, u0.data_ptr(), cin, 2 * c, 0, st)
        lib.advstep_conv3x3_prepare_f32(ww.data_ptr(), invstd.data_ptr(), u1.data_ptr(), cin, 2 * c, 1, st)
        u2 = torch.empty(lib.advstep_conv3x3_prepared_floats(cin, 2 * c, 2), device=dev)       # the compact source's channel order
        lib.advstep_conv3x3_prepare_f32(ww.data_ptr(), invstd.data_ptr(), u2.data_ptr(), cin, 2 * c, 2, st)
        gout = torch.randn(B, 2 * c, h, wd, device=dev)
        gxx = torch.empty_like(xx)
        fl = 2.0 * B * h * wd * cin * 2 * c * 9
        timeit(f"conv3x3_mfm_pool2_forward {name} ({cin}->{2 * c}, {h}x{wd})", lambda: lib.advstep_conv3x3_mfm_pool2_forward_f32(
            xx.data_ptr(), u0.data_ptr(), bb.data_ptr(), mean.data_ptr(), invstd.data_ptr(), yy.data_ptr(), ii.data_ptr(), B, cin,
            c, h, wd, st), flops=fl)
        timeit(f"conv3x3_backward_data     {name} ({2 * c}->{cin}, {h}x{wd})", lambda: lib.advstep_conv3x3_backward_data_f32(
            gout.data_ptr(), u1.data_ptr(), gxx.data_ptr(), B, cin, 2 * c, h, wd, st), flops=fl)
        timeit(f"conv3x3_mfm_pool2_backward {name} ({2 * c}->{cin}, {h}x{wd})", lambda: lib.advstep_conv3x3_mfm_pool2_backward_f32(
            yy.data_ptr(), ii.data_ptr(), u2.data_ptr(), gxx.data_ptr(), B, cin, c, h, wd, st), flops=fl)

    # ---- MFM + pool after the 3x3 convolutions
    for name, c, h, wd in (("L6 ", 48, 202, 40), ("L13", 64, 101, 20), ("L25", 32, 50, 10)):
        xx = torch.randn(B, 2 * c, h, wd, device=dev)
        bb = torch.randn(2 * c, device=dev)
        yy = torch.empty(B, c, h // 2, wd // 2, device=dev)
        ii = torch.empty(yy.numel(), dtype=torch.uint8, device=dev)
        gxx = torch.empty_like(xx)
        moved = xx.numel() * 4.0 + yy.numel() * 5.0
        timeit(f"mfm_pool2_forward  {name} ({2 * c}ch, {h}x{wd})", lambda: lib.advstep_mfm_pool2_forward_f32(
            xx.data_ptr(), bb.data_ptr(), None, None, yy.data_ptr(), ii.data_ptr(), B, c, h, wd, st), bytes_moved=moved)
        timeit(f"mfm_pool2_backward {name} ({2 * c}ch, {h}x{wd})", lambda: lib.advstep_mfm_pool2_backward_f32(
            yy.data_ptr(), ii.data_ptr(), None, gxx.data_ptr(), B, c, h, wd, st), bytes_moved=moved)

    # ---- LSTM layer (T = 25, H = 80, bidirectional)
    T, Hh, D = 25, 80, 2
    gxl = torch.randn(T, B, D, 4 * Hh, device=dev)
    whh = torch.randn(D, 4 * Hh, Hh, device=dev) * 0.1
    out = torch.empty(T, B, D * Hh, device=dev)
    gates, cell = torch.empty(T, B, D, 4 * Hh, device=dev), torch.empty(T, B, D, Hh, device=dev)
    dgx = torch.empty_like(gxl)
    timeit("lstm_forward  (T=25, H=80, 2 dirs)", lambda: lib.advstep_lstm_forward_f32(
        gxl.data_ptr(), whh.data_ptr(), out.data_ptr(), gates.data_ptr(), cell.data_ptr(), T, B, D, Hh, st))
    timeit("lstm_backward (T=25, H=80, 2 dirs)", lambda: lib.advstep_lstm_backward_f32(
        out.data_ptr(), whh.data_ptr(), gates.data_ptr(), cell.data_ptr(), dgx.data_ptr(), T, B, D, Hh, st))

    # ---- LFCC frontend pieces
    from audio_deepfake_adversarial_attacks_amd import frontend_ops, frontends
    lf = frontends.LFCC().to(dev)
    tables = lf._tables()
    Tn, NF, F, M, K = 64_600, 404, 257, 128, 80
    wav = torch.rand(B, Tn, device=dev)
    frames = torch.empty(B, NF, 512, device=dev)
    win = lf._window_nfft()
    timeit("stft_frames", lambda: lib.advstep_stft_frames_f32(wav.data_ptr(), win.data_ptr(), frames.data_ptr(), B, Tn, NF,
                                                               160, 512, st), bytes_moved=4.0 * (wav.numel() + frames.numel()))
    spec = torch.fft.rfft(frames, dim=-1)
    sr = torch.view_as_real(spec)
    timeit("rocFFT r2c (torch.fft.rfft)", lambda: torch.fft.rfft(frames, dim=-1))
    band = torch.empty(B, NF, M, device=dev)
    nblk = lib.advstep_lfcc_block_count(B, M, NF)
    bmax = torch.empty(nblk, device=dev)
    stats = torch.empty(4, device=dev)
    outl = torch.empty(B, NF, K, device=dev)
    timeit("lfcc_bands", lambda: lib.advstep_lfcc_bands_f32(sr.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(),
                                                           tables.span, band.data_ptr(), bmax.data_ptr(), B, F, M, NF, st),
           bytes_moved=4.0 * (sr.numel() + band.numel()))
    timeit("lfcc_reduce_max", lambda: lib.advstep_lfcc_reduce_max_f32(bmax.data_ptr(), nblk, stats.data_ptr(), st))
    from audio_deepfake_adversarial_attacks_amd.frontend_ops import dct_fragments
    frag = dct_fragments(lf.dct_mat)
    nblk_stft = lib.advstep_stft_bands_block_count(B, NF)     # what the fused STFT kernel leaves: 26 maxima per utterance
    timeit("lfcc_project", lambda: lib.advstep_lfcc_max_project_f32(band.data_ptr(), lf.dct_mat.data_ptr(), frag.data_ptr(),
                                                                   bmax.data_ptr(), nblk_stft, stats.data_ptr(), 80.0,
                                                                   outl.data_ptr(), B, M, NF, K, st),
           bytes_moved=4.0 * (band.numel() + outl.numel()), flops=2.0 * B * NF * M * K)
    dband = torch.empty_like(band)
    timeit("lfcc_project_backward", lambda: lib.advstep_lfcc_project_backward_zero_f32(
        outl.data_ptr(), lf.dct_mat.data_ptr(), frag.data_ptr(), band.data_ptr(), stats.data_ptr(), 80.0, dband.data_ptr(), B, M,
        NF, K, 0, 0, st),
        bytes_moved=4.0 * (2 * band.numel() + outl.numel()), flops=2.0 * B * NF * M * K)
    dspec = torch.empty_like(sr)
    timeit("lfcc_bands_backward", lambda: lib.advstep_lfcc_bands_backward_f32(
        dband.data_ptr(), sr.data_ptr(), tables.fbt_start.data_ptr(), tables.fbt_w.data_ptr(), tables.span_t,
        dspec.data_ptr(), B, F, M, NF, 1, st), bytes_moved=4.0 * (2 * sr.numel() + dband.numel()))
    cs = torch.view_as_complex(dspec)
    timeit("rocFFT c2r (torch.fft.irfft)", lambda: torch.fft.irfft(cs, n=512, dim=-1, norm="forward"))
    dfr = torch.empty(B, NF, 512, device=dev)
    dxx = torch.empty(B, Tn, device=dev)
    timeit("stft_overlap_add", lambda: lib.advstep_stft_overlap_add_f32(dfr.data_ptr(), win.data_ptr(), dxx.data_ptr(), B, Tn,
                                                                         NF, 160, 512, st),
           bytes_moved=4.0 * (dfr.numel() + dxx.numel()))
    # the STFT end fused around an in-LDS FFT: replaces stft_frames + r2c + lfcc_bands / lfcc_bands_backward + c2r + overlap-add
    nblk2 = lib.advstep_stft_bands_block_count(B, NF)
    bmax2 = torch.empty(nblk2, device=dev)
    timeit("stft_bands (framing+FFT+fbank+dB)", lambda: lib.advstep_stft_bands_f32(
        wav.data_ptr(), win.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(), tables.span, band.data_ptr(),
        bmax2.data_ptr(), B, Tn, NF, 160, 512, M, st), bytes_moved=4.0 * (wav.numel() + band.numel()))
    timeit("stft_bands_backward (fbank^T+FFT+iFFT+OLA)", lambda: lib.advstep_stft_bands_backward_f32(
        wav.data_ptr(), win.data_ptr(), dband.data_ptr(), tables.fbt_start.data_ptr(), tables.fbt_w.data_ptr(), tables.span_t,
        dxx.data_ptr(), B, Tn, NF, 160, 512, M, st), bytes_moved=4.0 * (2 * wav.numel() + dband.numel()))
    # mel-spec frontend of SpecRNet: STFT -> complex mel projection -> magnitude / phase, and back
    mel = frontends.MelSpecFrontend().to(dev)
    mt, mwin = mel._fused_state(torch.device(dev))
    Mm = mel.mel_scale.fb.shape[1]
    mout = torch.empty(B, 2, Mm, NF, device=dev)
    dmel = torch.randn(B, 2, Mm, NF, device=dev)
    timeit("stft_mel (framing+FFT+mel+abs/angle)", lambda: lib.advstep_stft_mel_f32(
        wav.data_ptr(), mwin.data_ptr(), mt.fb_start.data_ptr(), mt.fb_w.data_ptr(), mt.span, mout.data_ptr(), B, Tn, NF, 160, 512,
        Mm, st), bytes_moved=4.0 * (wav.numel() + mout.numel()))
    timeit("stft_mel_backward", lambda: lib.advstep_stft_mel_backward_f32(
        wav.data_ptr(), mwin.data_ptr(), dmel.data_ptr(), mt.fb_start.data_ptr(), mt.fb_w.data_ptr(), mt.span, mt.fbt_start.data_ptr(),
        mt.fbt_w.data_ptr(), mt.span_t, dxx.data_ptr(), B, Tn, NF, 160, 512, Mm, st),
        bytes_moved=4.0 * (2 * wav.numel() + dmel.numel()))
    timeit("stft_mel_backward_from_output", lambda: lib.advstep_stft_mel_backward_from_output_f32(
        mwin.data_ptr(), dmel.data_ptr(), mout.data_ptr(), mt.fbt_start.data_ptr(), mt.fbt_w.data_ptr(), mt.span_t, dxx.data_ptr(), B, Tn,
        NF, 160, 512, Mm, st), bytes_moved=4.0 * (wav.numel() + 2 * dmel.numel()))
    if a.json:
        Path(a.json).write_text(json.dumps({"batch": B, "kernels": res}, indent=1))


if __name__ == "__main__":
    main()
