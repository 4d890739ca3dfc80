"""Parameterised sinc filterbank encoder — first layer of RawNet3.

The reference imports `Encoder(ParamSincFB(...))` from ``asteroid-filterbanks==0.4.0`` (requirements.txt:41,
src/models/rawnet3.py:7-8,27-33), which is not part of the reference tree.  This file restates that package's
published algorithm (SincNet, Ravanelli & Bengio 2018, with the odd "sin" filters of Pariente et al. 2020):
learnable low cut-offs and bandwidths initialised on a mel grid, Hamming-windowed even (cos) and odd (sin)
band-pass pairs, applied as a strided Conv1d without padding.

PARITY UNPINNED (no golden vector exists for it in the reference; see DESIGN.md section 5 — the even filters are
cross-checked against scipy.signal.firwin and the pair against the one-sided-spectrum property in tests/test_models.py).  Parameter / buffer names
follow asteroid's (`filterbank.low_hz_`, `filterbank.band_hz_`, `filterbank.window_`, `filterbank.n_`)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class ParamSincFB(nn.Module):
    def __init__(self, n_filters, kernel_size, stride=None, sample_rate=16000.0, min_low_hz=50, min_band_hz=50):
        super().__init__()
        if n_filters % 2 != 0:
            raise ValueError(f"ParamSincFB needs an even number of filters, got {n_filters}")
        if kernel_size % 2 == 0:
            kernel_size += 1  # symmetric filters need an odd length
        self.n_filters, self.kernel_size = n_filters, kernel_size
        self.stride = stride if stride else kernel_size // 2
        self.sample_rate = float(sample_rate)
        self.min_low_hz, self.min_band_hz = min_low_hz, min_band_hz
        self.half_kernel = kernel_size // 2
        self.cutoff = n_filters // 2

        # mel-spaced initial band edges between 30 Hz and Nyquist - (min_low + min_band)
        to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
        to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
        high_hz = self.sample_rate / 2 - (min_low_hz + min_band_hz)
        hz = to_hz(np.linspace(to_mel(30), to_mel(high_hz), n_filters // 2 + 1, dtype="float32"))
        self.low_hz_ = nn.Parameter(torch.from_numpy(hz[:-1]).float().view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.from_numpy(np.diff(hz)).float().view(-1, 1))

        window = np.hamming(kernel_size)[: self.half_kernel]
        self.register_buffer("window_", torch.from_numpy(window).float())
        self.register_buffer("n_", 2 * np.pi * (torch.arange(-self.half_kernel, 0.0).view(1, -1) / self.sample_rate))

    def _band_pass(self, low, high, odd: bool):
        band = (high - low)[:, 0]
        ft_low, ft_high = torch.matmul(low, self.n_), torch.matmul(high, self.n_)
        if not odd:
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (self.n_ / 2)) * self.window_
            center = 2 * band.view(-1, 1)
            right = torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (self.n_ / 2)) * self.window_
            center = torch.zeros_like(band.view(-1, 1))
            right = -torch.flip(left, dims=[1])
        bp = torch.cat([left, center, right], dim=1) / (2 * band[:, None])
        return bp.view(self.n_filters // 2, 1, self.kernel_size)

    def filters(self):
        low = self.min_low_hz + torch.abs(self.low_hz_)
        high = torch.clamp(low + self.min_band_hz + torch.abs(self.band_hz_), self.min_low_hz, self.sample_rate / 2)
        return torch.cat([self._band_pass(low, high, odd=False), self._band_pass(low, high, odd=True)], dim=0)


def _gemm_encoder_enabled() -> bool:
    """ADVSTEP_RAWNET3_SINC_GEMM=0 keeps F.conv1d for the encoder on HIP devices (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_SINC_GEMM", "1") != "0"


class _StridedCorrelationFrozen(torch.autograd.Function):
    """conv1d(x (B, 1, T), w (F, 1, K), stride) as ONE batched GEMM over an explicit (B, K, frames) patch matrix, and its input
    gradient as the transposed GEMM + `fold`.  Input gradient only (the filterbank is frozen while an attack runs).

    Why: MIOpen runs this one-input-channel, 251-tap, stride-10 convolution sample by sample — im2col + a 256 x 251 x 6435 GEMM
    per utterance forward, GEMM + col2im per utterance backward: 256 launches and 5.4 ms of a 72 ms RawNet3 iteration at B = 64.
    Batched, the same arithmetic is a 413 MB patch copy + one 53-GFLOP GEMM each way."""

    @staticmethod
    def forward(ctx, x, w, stride):
        B, _, T = x.shape
        nf, _, K = w.shape
        cols = x[:, 0].unfold(-1, K, stride).transpose(1, 2).contiguous()          # (B, K, frames)
        ctx.save_for_backward(w)
        ctx.meta = (T, K, stride)
        return torch.matmul(w[:, 0], cols)                                        # (F, K) @ (B, K, frames) -> (B, F, frames)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        T, K, stride = ctx.meta
        gcols = torch.matmul(w[:, 0].t(), g)                                      # (B, K, frames)
        frames = g.shape[-1]
        span = (frames - 1) * stride + K                                          # samples the frames cover (<= T)
        gx = F.fold(gcols, output_size=(1, span), kernel_size=(1, K), stride=(1, stride)).view(g.shape[0], 1, span)
        if span < T:
            gx = F.pad(gx, (0, T - span))
        return gx, None, None


class Encoder(nn.Module):
    """(B, T) or (B, 1, T) -> (B, n_filters, frames): strided correlation with the filterbank, no padding."""

    def __init__(self, filterbank: ParamSincFB):
        super().__init__()
        self.filterbank = filterbank

    def _frozen_filters(self):
        """The filterbank evaluated once per parameter version (≈ 40 small kernels per call otherwise)."""
        fb = self.filterbank
        tensors = [fb.low_hz_, fb.band_hz_, fb.window_, fb.n_]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if getattr(self, "_filters_key", None) != key:
            with torch.no_grad():
                self._filters_key, self._filters_val = key, fb.filters().contiguous()
        return self._filters_val

    def forward(self, waveform):
        if waveform.ndim == 2:
            waveform = waveform.unsqueeze(1)
        fb = self.filterbank
        frozen = not (torch.is_grad_enabled() and (fb.low_hz_.requires_grad or fb.band_hz_.requires_grad))
        if (frozen and waveform.is_cuda and waveform.dtype == torch.float32 and waveform.shape[1] == 1
                and waveform.shape[-1] >= fb.kernel_size and _gemm_encoder_enabled()):
            return _StridedCorrelationFrozen.apply(waveform, self._frozen_filters(), fb.stride)
        return F.conv1d(waveform, fb.filters(), stride=fb.stride, padding=0)
