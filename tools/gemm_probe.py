#!/usr/bin/env python
"""The recurrent layers' projection GEMMs (LCNN: (T B = 3200, 160) x (160, 640) + bias forward, (3200, 640) x (640, 160) backward) through
torch's BLAS back ends: rocBLAS (default) vs hipBLASLt, HIP events over bursts of launches.   python tools/gemm_probe.py"""
import torch


def main():
    dev = torch.device("cuda:0")
    x = torch.randn(3200, 160, device=dev)
    w = torch.randn(640, 160, device=dev)
    b = torch.randn(640, device=dev)
    g = torch.randn(3200, 640, device=dev)

    def timeit(name, fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:44s} {e0.elapsed_time(e1) * 1e3 / n:7.1f} us")

    for lib in ("default", "hipblaslt", "cublas"):
        try:
            if lib != "default":
                torch.backends.cuda.preferred_blas_library(lib)
        except Exception as exc:  # noqa: BLE001
            print(lib, "unavailable:", exc)
            continue
        print("== preferred_blas_library:", torch.backends.cuda.preferred_blas_library())
        timeit("addmm (3200,160)x(160,640)+b  [forward]", lambda: torch.addmm(b, x, w.t()))
        timeit("mm    (3200,640)x(640,160)    [backward]", lambda: torch.mm(g, w))
        wt = w.t().contiguous()
        timeit("addmm with W^T materialised", lambda: torch.addmm(b, x, wt))
        timeit("mm with W^T materialised (g x (W^T)^T)", lambda: torch.mm(g, wt.t()))


if __name__ == "__main__":
    main()
