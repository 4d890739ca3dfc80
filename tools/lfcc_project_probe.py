#!/usr/bin/env python
"""Timing of the LFCC projection pair (csrc/lfcc.hip) through the C ABI: plain and fused entry points, hot (one buffer set) and
cold (buffer sets rotated past the Infinity Cache).    python tools/lfcc_project_probe.py [--batch 128] [--launches 50]"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd import _lib, frontends  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--launches", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, T, M, K = a.batch, 64_600, 128, 80
    NF = 1 + T // 160
    st = torch.cuda.current_stream().cuda_stream
    lf = frontends.LFCC().to(dev)
    dct = lf.dct_mat
    sets = 12
    band = [torch.randn(B, NF, M, device=dev) * 10 for _ in range(sets)]
    out = [torch.randn(B, NF, K, device=dev) for _ in range(sets)]
    dband = [torch.empty(B, NF, M, device=dev) for _ in range(sets)]
    dx = [torch.empty(B, T, device=dev) for _ in range(sets)]
    nblk = lib.advstep_stft_bands_block_count(B, NF)
    bmax = torch.randn(nblk, device=dev)
    stats = torch.zeros(4, device=dev)
    lib.advstep_lfcc_reduce_max_f32(bmax.data_ptr(), nblk, stats.data_ptr(), st)

    def timeit(name, fn):
        for mode, nsets in (("hot", 1), ("cold", sets)):
            for i in range(3):
                fn(i % nsets)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.launches):
                fn(i % nsets)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:40s} {mode:5s} {e0.elapsed_time(e1) * 1e3 / a.launches:8.1f} us")

    from audio_deepfake_adversarial_attacks_amd.frontend_ops import dct_fragments
    frag = dct_fragments(dct)
    timeit("lfcc_project (vector ALU)", lambda i: lib.advstep_lfcc_project_f32(band[i].data_ptr(), dct.data_ptr(), stats.data_ptr(),
                                                                               80.0, out[i].data_ptr(), B, M, NF, K, st))
    timeit("lfcc_max_project", lambda i: lib.advstep_lfcc_max_project_f32(band[i].data_ptr(), dct.data_ptr(), frag.data_ptr(),
                                                                          bmax.data_ptr(), nblk, stats.data_ptr(), 80.0,
                                                                          out[i].data_ptr(), B, M, NF, K, st))
    timeit("lfcc_project_backward (vector ALU)", lambda i: lib.advstep_lfcc_project_backward_f32(
        out[i].data_ptr(), dct.data_ptr(), band[i].data_ptr(), stats.data_ptr(), 80.0, dband[i].data_ptr(), B, M, NF, K, st))
    timeit("lfcc_project_backward_zero (no fill)", lambda i: lib.advstep_lfcc_project_backward_zero_f32(
        out[i].data_ptr(), dct.data_ptr(), frag.data_ptr(), band[i].data_ptr(), stats.data_ptr(), 80.0, dband[i].data_ptr(), B, M,
        NF, K, 0, 0, st))
    timeit("lfcc_project_backward_zero", lambda i: lib.advstep_lfcc_project_backward_zero_f32(
        out[i].data_ptr(), dct.data_ptr(), frag.data_ptr(), band[i].data_ptr(), stats.data_ptr(), 80.0, dband[i].data_ptr(), B, M,
        NF, K, dx[i].data_ptr(), dx[i].numel(), st))
    timeit("memset dx (hipMemsetAsync)", lambda i: dx[i].zero_())


if __name__ == "__main__":
    main()
