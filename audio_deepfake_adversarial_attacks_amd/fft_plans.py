"""Batched 1-D real FFTs through hipFFT directly (the library PyTorch-ROCm itself loads), for the fused LFCC frontend.

`torch.fft.rfft` / `irfft` on ROCm clone their input first (hipFFT may overwrite it; the c2r transform always does),
which costs two 106 MB device copies per PGD iteration at B = 128.  The frontend's FFT inputs are scratch tensors it
owns, so it can hand them to hipFFT as they are.  Plans are cached per (device, batch, n_fft, direction) and always
execute on torch's current stream.  If the hipFFT handle cannot be created the callers fall back to torch.fft — the
same library underneath, only with the defensive copies."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

_HIPFFT_R2C, _HIPFFT_C2R = 0x2A, 0x2C
_lib: Optional[ctypes.CDLL] = None
_lib_failed = False
_plans: Dict[Tuple[int, int, int, int], ctypes.c_void_p] = {}


def _load() -> Optional[ctypes.CDLL]:
    global _lib, _lib_failed
    if _lib is not None or _lib_failed:
        return _lib
    candidates = [os.path.join(os.path.dirname(torch.__file__), "lib", "libhipfft.so"), "libhipfft.so"]
    for path in candidates:
        try:
            lib = ctypes.CDLL(path)
            lib.hipfftPlanMany.restype = ctypes.c_int
            lib.hipfftPlanMany.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                           ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int]
            lib.hipfftSetStream.restype = ctypes.c_int
            lib.hipfftSetStream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            for name in ("hipfftExecR2C", "hipfftExecC2R"):
                fn = getattr(lib, name)
                fn.restype = ctypes.c_int
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            _lib = lib
            return lib
        except (OSError, AttributeError):
            continue
    _lib_failed = True
    return None


def _plan(device: torch.device, batch: int, nfft: int, kind: int) -> Optional[ctypes.c_void_p]:
    lib = _load()
    if lib is None:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), batch, nfft, kind)
    plan = _plans.get(key)
    if plan is None:
        handle = ctypes.c_void_p()
        n = (ctypes.c_int * 1)(nfft)
        with torch.cuda.device(device):
            rc = lib.hipfftPlanMany(ctypes.byref(handle), 1, n, None, 1, 0, None, 1, 0, kind, batch)
        if rc != 0 or not handle.value:
            return None
        plan = _plans[key] = handle
    return plan


def rfft_into(frames: torch.Tensor, spec_real: torch.Tensor) -> bool:
    """frames (batch, nfft) f32 contiguous -> spec_real (batch, nfft//2+1, 2) f32 (interleaved complex), unnormalised.
    `frames` may be overwritten.  Returns False when hipFFT is unavailable (caller falls back to torch.fft)."""
    batch, nfft = frames.shape
    plan = _plan(frames.device, batch, nfft, _HIPFFT_R2C)
    if plan is None:
        return False
    with torch.cuda.device(frames.device):
        stream = torch.cuda.current_stream(frames.device).cuda_stream
        ok = _lib.hipfftSetStream(plan, stream) == 0 and _lib.hipfftExecR2C(plan, frames.data_ptr(), spec_real.data_ptr()) == 0
    return ok


def irfft_into(spec_real: torch.Tensor, frames: torch.Tensor) -> bool:
    """spec_real (batch, nfft//2+1, 2) -> frames (batch, nfft), unnormalised c2r; `spec_real` IS overwritten."""
    batch, nfft = frames.shape
    plan = _plan(frames.device, batch, nfft, _HIPFFT_C2R)
    if plan is None:
        return False
    with torch.cuda.device(frames.device):
        stream = torch.cuda.current_stream(frames.device).cuda_stream
        ok = _lib.hipfftSetStream(plan, stream) == 0 and _lib.hipfftExecC2R(plan, spec_real.data_ptr(), frames.data_ptr()) == 0
    return ok
