"""Spectral frontends of the detectors (reference: src/frontends.py:1-79), as differentiable torch modules.

The reference delegates the arithmetic to ``torchaudio==0.10.0`` (``transforms.LFCC / MFCC / MelScale``), which is
not part of the reference tree.  The classes below restate torchaudio 0.10's published algorithms (function
names cited per method); they run forward + backward under PyTorch-ROCm (hipFFT + rocBLAS GEMMs) exactly where
the reference runs them — inside ``model(adv)`` on every attack step.

PARITY UNPINNED: the reference holds no test or golden vector for the frontends and torchaudio is not
installable here, so these are checked by analytic properties only (tests/test_frontends.py): filterbank
partition of unity, DCT orthonormality, STFT against a direct DFT, shape pins from the reference's smoke
blocks (src/models/lcnn.py:252 -> (B, 1, 80, 404)).

Differences from the reference that are deliberate
  * the reference keeps ONE module-global LFCC/MFCC/MelScale instance pinned to a global device
    (frontends.py:11-38) shared by every model; here each model owns its frontend module, so `.to(device)`
    and one-process-per-GPU placement work without a global;
  * buffer names follow torchaudio's (`filter_mat`, `dct_mat`, `Spectrogram.window`, ...) so a reference
    checkpoint — whose state_dict contains `frontend.*` buffers because the singleton is registered as a
    submodule (src/models/lcnn.py:230) — loads with strict=True.
"""
from __future__ import annotations

import math
from typing import Callable, List, Union

import torch
from torch import nn

def _fused_stft_enabled() -> bool:
    """ADVSTEP_FUSED_STFT=0 keeps torch.stft (pad + strided frames + clone) in front of the fused LFCC tail."""
    import os
    return os.environ.get("ADVSTEP_FUSED_STFT", "1") != "0"


def _fused_mel_enabled() -> bool:
    """ADVSTEP_FUSED_MEL=0 keeps the plain torch op chain of the mel-spec frontend on the GPU; default on."""
    import os
    return os.environ.get("ADVSTEP_FUSED_MEL", "1") != "0"


def _fused_lfcc_enabled() -> bool:
    """ADVSTEP_FUSED_LFCC=0 keeps the plain torch op chain on the GPU (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_FUSED_LFCC", "1") != "0"


# values from FakeAVCeleb paper (frontends.py:6-9)
SAMPLING_RATE = 16_000
win_length = 400  # int((25 / 1_000) * SAMPLING_RATE)
hop_length = 160  # int((10 / 1_000) * SAMPLING_RATE)
N_FFT = 512


# ---------------------------------------------------------------------------------------------------------
# torchaudio.functional restatements
# ---------------------------------------------------------------------------------------------------------

def _triangular_filterbank(all_freqs: torch.Tensor, f_pts: torch.Tensor) -> torch.Tensor:
    """torchaudio.functional._create_triangular_filterbank -> (n_freqs, n_filter)."""
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def linear_fbanks(n_freqs: int, f_min: float, f_max: float, n_filter: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.linear_fbanks."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    f_pts = torch.linspace(f_min, f_max, n_filter + 2)
    return _triangular_filterbank(all_freqs, f_pts)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk')."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    return _triangular_filterbank(all_freqs, f_pts)


def create_dct(n_mfcc: int, n_mels: int, norm: str = "ortho") -> torch.Tensor:
    """torchaudio.functional.create_dct (DCT-II) -> (n_mels, n_mfcc)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct *= 2.0
    else:
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


def amplitude_to_db_power(x: torch.Tensor, top_db: float = 80.0) -> torch.Tensor:
    """torchaudio.functional.amplitude_to_DB(x, multiplier=10, amin=1e-10, db_multiplier=0, top_db).

    torchaudio packs a 3-D (B, n_filter, time) input as ONE (1, B, n_filter, time) item, so the floor is
    (max over the whole batch) - top_db: outputs depend on batch composition."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
    shape = x_db.size()
    packed = shape[-3] if x_db.dim() > 2 else 1
    x_db = x_db.reshape(-1, packed, shape[-2], shape[-1])
    floor = (x_db.amax(dim=(-3, -2, -1)) - top_db).view(-1, 1, 1, 1)
    return torch.max(x_db, floor).reshape(shape)


# ---------------------------------------------------------------------------------------------------------
# torchaudio.transforms restatements
# ---------------------------------------------------------------------------------------------------------

class Spectrogram(nn.Module):
    """torchaudio.transforms.Spectrogram(power=2, center=True, pad_mode='reflect', onesided=True) with a
    periodic Hann window of `win_length`."""

    def __init__(self, n_fft: int = N_FFT, win_length: int = win_length, hop_length: int = hop_length):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.register_buffer("window", torch.hann_window(win_length))

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        spec = torch.stft(waveform, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                          window=self.window, center=True, pad_mode="reflect", normalized=False, onesided=True,
                          return_complex=True)
        return spec.abs().pow(2.0)


def _cached_tables(owner, fb: torch.Tensor):
    """Sparse filterbank tables for the fused kernels, cached on `owner` until `fb` moves or changes."""
    key = (fb.data_ptr(), fb._version, str(fb.device))
    if getattr(owner, "_tables_key", None) != key:
        from . import frontend_ops
        owner._tables_key, owner._tables_val = key, frontend_ops.filterbank_tables(fb)
    return owner._tables_val


def _cached_window_nfft(owner, spectrogram) -> torch.Tensor:
    """The analysis window zero-padded (centred) to n_fft, as torch.stft applies it; cached on `owner`."""
    w = spectrogram.window
    key = (w.data_ptr(), w._version, str(w.device))
    if getattr(owner, "_wpad_key", None) != key:
        n_fft = spectrogram.n_fft
        left = (n_fft - w.numel()) // 2
        padded = torch.zeros(n_fft, dtype=w.dtype, device=w.device)
        padded[left:left + w.numel()] = w.detach()
        owner._wpad_key, owner._wpad_val = key, padded
    return owner._wpad_val


def _fused_cepstrum(owner, waveform, spectrogram, fb, dct_mat, top_db):
    """filterbank -> dB -> DCT of the power STFT through the fused kernels, or None when they do not apply."""
    if not (waveform.is_cuda and waveform.dim() == 2 and waveform.dtype == torch.float32 and _fused_lfcc_enabled()
            and dct_mat.shape[0] <= 128 and dct_mat.shape[1] in (20, 40, 80)):
        return None
    from . import frontend_ops
    sg = spectrogram
    tables = _cached_tables(owner, fb)
    if waveform.shape[1] > sg.n_fft // 2 and sg.n_fft % 4 == 0 and _fused_stft_enabled():
        # framing, FFT, filterbank, dB in one kernel; floor + DCT in another (and two more on the way back)
        return frontend_ops.lfcc_from_waveform(waveform, _cached_window_nfft(owner, sg), sg.hop_length, tables, dct_mat, top_db)
    spec = torch.stft(waveform, n_fft=sg.n_fft, hop_length=sg.hop_length, win_length=sg.win_length, window=sg.window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return frontend_ops.lfcc_tail(spec, tables, dct_mat, top_db)


class LFCC(nn.Module):
    """torchaudio.transforms.LFCC(sample_rate, n_filter=128, n_lfcc, dct_type=2, norm='ortho', log_lf=False)."""

    def __init__(self, sample_rate: int = SAMPLING_RATE, n_filter: int = 128, n_lfcc: int = 80,
                 n_fft: int = N_FFT, win_length: int = win_length, hop_length: int = hop_length):
        super().__init__()
        self.top_db = 80.0
        self.Spectrogram = Spectrogram(n_fft, win_length, hop_length)
        self.register_buffer("filter_mat", linear_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_filter,
                                                         sample_rate))
        self.register_buffer("dct_mat", create_dct(n_lfcc, n_filter, "ortho"))

    def _tables(self):
        return _cached_tables(self, self.filter_mat)

    def _window_nfft(self):
        return _cached_window_nfft(self, self.Spectrogram)

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        fused = _fused_cepstrum(self, waveform, self.Spectrogram, self.filter_mat, self.dct_mat, self.top_db)
        if fused is not None:     # SURVEY.md section 8-f2; written frame-major for LCNN's first block
            return fused
        spec = self.Spectrogram(waveform)                                               # (B, 257, frames)
        bands = torch.matmul(spec.transpose(-1, -2), self.filter_mat).transpose(-1, -2)   # (B, 128, frames)
        bands = amplitude_to_db_power(bands, self.top_db)
        return torch.matmul(bands.transpose(-1, -2), self.dct_mat).transpose(-1, -2)     # (B, 80, frames)


class MelScale(nn.Module):
    """torchaudio.transforms.MelScale(n_mels, sample_rate, f_min=0, f_max=sr//2, n_stft, norm=None, 'htk')."""

    def __init__(self, n_mels: int = 80, sample_rate: int = SAMPLING_RATE, n_stft: int = N_FFT // 2 + 1,
                 persistent: bool = True):
        super().__init__()
        self.register_buffer("fb", melscale_fbanks(n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate),
                             persistent=persistent)

    def forward(self, specgram: torch.Tensor) -> torch.Tensor:
        return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram(n_mels=128) — only as the first stage of MFCC."""

    def __init__(self, sample_rate: int, n_fft: int, win_length: int, hop_length: int, n_mels: int = 128):
        super().__init__()
        self.spectrogram = Spectrogram(n_fft, win_length, hop_length)
        self.mel_scale = MelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, waveform):
        return self.mel_scale(self.spectrogram(waveform))


class MFCC(nn.Module):
    """torchaudio.transforms.MFCC(sample_rate, n_mfcc, dct_type=2, norm='ortho', log_mels=False, melkwargs)."""

    def __init__(self, sample_rate: int = SAMPLING_RATE, n_mfcc: int = 80, n_fft: int = N_FFT,
                 win_length: int = win_length, hop_length: int = hop_length):
        super().__init__()
        self.top_db = 80.0
        self.MelSpectrogram = MelSpectrogram(sample_rate, n_fft, win_length, hop_length)
        self.register_buffer("dct_mat", create_dct(n_mfcc, 128, "ortho"))

    def forward(self, waveform):
        # same structure as LFCC (power STFT -> filterbank -> dB -> DCT) with a mel filterbank: same fused kernels
        fused = _fused_cepstrum(self, waveform, self.MelSpectrogram.spectrogram, self.MelSpectrogram.mel_scale.fb,
                                self.dct_mat, self.top_db)
        if fused is not None:
            return fused
        mel = amplitude_to_db_power(self.MelSpectrogram(waveform), self.top_db)
        return torch.matmul(mel.transpose(-1, -2), self.dct_mat).transpose(-1, -2)


class MelSpecFrontend(nn.Module):
    """`prepare_mel_scale_vector` / `prepare_stft_features` (frontends.py:53-79): rectangular-window STFT
    (no window is passed to torch.stft at :62-68), MelScale applied to the real and imaginary parts
    separately (:71-72), then magnitude and phase stacked on a channel axis -> (B, 2, 80, frames)."""

    def __init__(self, win_length: int = win_length, hop_length: int = hop_length):
        super().__init__()
        self.win_length, self.hop_length = win_length, hop_length
        # non-persistent: in the reference this frontend is a plain function (no state_dict entries)
        self.mel_scale = MelScale(80, SAMPLING_RATE, N_FFT // 2 + 1, persistent=False)

    def _fused_state(self, device):
        """Sparse mel tables + the rectangular window zero-padded (centred) to n_fft, cached per device."""
        key = (self.mel_scale.fb.data_ptr(), self.mel_scale.fb._version, str(device))
        if getattr(self, "_fused_key", None) != key:
            from . import frontend_ops
            left = (N_FFT - self.win_length) // 2
            window = torch.zeros(N_FFT, dtype=torch.float32, device=device)
            window[left:left + self.win_length] = 1.0
            self._fused_key, self._fused_val = key, (frontend_ops.filterbank_tables(self.mel_scale.fb), window)
        return self._fused_val

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        if audio.is_cuda and audio.dim() == 2 and audio.dtype == torch.float32 and _fused_mel_enabled():
            from . import frontend_ops
            tables, window = self._fused_state(audio.device)
            if frontend_ops.mel_spec_supported(N_FFT, self.hop_length, audio.shape[1], self.mel_scale.fb.shape[1], tables.span,
                                               tables.span_t):
                # framing + FFT + complex mel projection + magnitude / phase in one kernel each way
                return frontend_ops.mel_spec_from_waveform(audio, window, self.hop_length, tables)
        stft = torch.stft(audio, n_fft=N_FFT, return_complex=True, hop_length=self.hop_length,
                          win_length=self.win_length,
                          window=torch.ones(self.win_length, dtype=audio.dtype, device=audio.device))
        mel = torch.complex(self.mel_scale(stft.real), self.mel_scale(stft.imag))
        return torch.stack([mel.abs(), mel.angle()], dim=1)


def get_frontend(frontends: List[str]) -> Union[nn.Module, Callable]:
    """frontends.py:41-50 — same precedence (mfcc, then lfcc, then mel_spec) and the same error."""
    if "mfcc" in frontends:
        return MFCC()
    elif "lfcc" in frontends:
        return LFCC()
    elif "mel_spec" in frontends:
        return MelSpecFrontend()
    raise ValueError(f"{frontends} frontend is not supported!")
