"""The LFCC frontend after the STFT as one differentiable op backed by the HIP kernels of include/advstep_frontend.h
(SURVEY.md section 8-f2): power -> sparse linear filterbank -> dB with the batch-wide floor -> DCT, forward and
backward, written frame-major so LCNN's first block reads it without a transpose copy.

`lfcc_tail(spec, tables, dct, top_db)` takes torch.stft's complex output (B, F, NF) and returns (B, K, NF) — a view of
a contiguous (B, NF, K) buffer.  Numerically it follows frontends.LFCC (this repository's restatement of torchaudio's
LFCC) to float rounding, including torchaudio's gradient path through `amax` (floored gradients flow to the batch
maximum); tests/test_gpu_frontend_ops.py.  HIP tensors only."""
from __future__ import annotations

import os
from typing import NamedTuple, Optional

import torch

from . import _lib, fft_plans
from .hip_ops import _Launch, _stream
from .lcnn_ops import _IdKeyed


def _direct_fft_enabled() -> bool:
    """ADVSTEP_DIRECT_FFT=0 routes the frontend's FFTs through torch.fft (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_DIRECT_FFT", "1") != "0"


class FilterbankTables(NamedTuple):
    """Sparse form of a (F, M) triangular filterbank: per band the first bin + `span` weights, and the transpose."""
    fb_start: torch.Tensor   # (M,) int32
    fb_w: torch.Tensor       # (M, span) float32
    span: int
    fbt_start: torch.Tensor  # (F,) int32
    fbt_w: torch.Tensor      # (F, span_t) float32
    span_t: int


def filterbank_tables(filter_mat: torch.Tensor) -> FilterbankTables:
    fb = filter_mat.detach().float().cpu()
    F, M = fb.shape

    def pack(mat):  # rows of `mat` are the outputs; gather each row's non-zero run
        n_out = mat.shape[0]
        starts, spans = [], []
        for r in range(n_out):
            nz = torch.nonzero(mat[r]).flatten()
            starts.append(int(nz.min()) if nz.numel() else 0)
            spans.append(int(nz.max()) - int(nz.min()) + 1 if nz.numel() else 1)
        span = max(spans)
        w = torch.zeros(n_out, span)
        for r in range(n_out):
            hi = min(starts[r] + span, mat.shape[1])
            w[r, : hi - starts[r]] = mat[r, starts[r]:hi]
        return torch.tensor(starts, dtype=torch.int32), w, span

    fb_start, fb_w, span = pack(fb.t().contiguous())      # per band over bins
    fbt_start, fbt_w, span_t = pack(fb)                    # per bin over bands
    dev = filter_mat.device
    return FilterbankTables(fb_start.to(dev), fb_w.to(dev).contiguous(), span, fbt_start.to(dev),
                            fbt_w.to(dev).contiguous(), span_t)


_FRAGMENTS = _IdKeyed()      # weak, keyed by tensor identity


def dct_fragments(dct: torch.Tensor) -> Optional[torch.Tensor]:
    """The DCT matrix in the operand order of the matrix-core projection kernels (advstep_lfcc_project_prepare_f32), cached
    per tensor object and version (the entry dies with the tensor: a recycled address cannot hit it); None for sizes without
    such a path."""
    lib = _lib.load()
    M, K = dct.shape
    n = lib.advstep_lfcc_project_fragment_floats(M, K)
    if n == 0 or not dct.is_contiguous():
        return None
    key = (dct.data_ptr(), dct._version, str(dct.device))
    hit = _FRAGMENTS.get(dct)
    if hit is not None and hit[0] == key:
        return hit[1]
    if torch.cuda.is_current_stream_capturing():
        # a table built inside a capture would be filled only when the graph replays, and this cache would hand it to eager
        # callers before that: take the two-launch path for this call (graphed.py warms up eagerly, so this does not happen
        # on the shipped path)
        return None
    frag = torch.empty(n, dtype=torch.float32, device=dct.device)
    _lib.check(lib.advstep_lfcc_project_prepare_f32(dct.data_ptr(), M, K, frag.data_ptr(), _stream(dct.device)),
               "advstep_lfcc_project_prepare_f32")
    # once per weight version: the table is shared by every stream that calls later, so it must be complete before it is
    # published (a second stream's first call is not ordered behind this launch)
    torch.cuda.current_stream(dct.device).synchronize()
    _FRAGMENTS[dct] = (key, frag)
    return frag


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _rearm(ctx, stats: torch.Tensor) -> None:
    """A further backward pass through the same forward (retain_graph): the floored-gradient sum the previous one accumulated
    in `stats[2]` starts again from 0 - and so does the tie counter `stats[1]` where the backward pass is the one that fills it
    (the matrix-core path: stats[3] == 1; the two-launch forward counts the ties itself).  `stats` is this node's own scratch,
    kept as a plain attribute of ctx - NOT among the saved tensors, whose version check would refuse the third pass after
    this in-place reset - and so is the fragment table the forward pass chose: who counts the ties was decided
    there, the backward pass must not look the table up again (a cache miss would hand it the other kernel)."""
    if ctx.backward_calls:
        if ctx.ties_in_backward:
            stats[1:3].zero_()
        else:
            stats[2:3].zero_()
    ctx.backward_calls += 1


class _LfccTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, tables: FilterbankTables, dct, top_db: float):
        if not spec.is_cuda or spec.dtype != torch.complex64 or spec.dim() != 3:
            raise _lib.AdvstepError("lfcc_tail needs a complex64 (B, F, NF) STFT on a HIP device (no CPU fallback)")
        B, F, NF = spec.shape
        sn = spec.transpose(1, 2)
        if not sn.is_contiguous():       # torch.stft's native layout is already (B, NF, F)
            sn = sn.contiguous()
        sr = torch.view_as_real(sn)      # (B, NF, F, 2) float32
        M, K = dct.shape
        dev = spec.device
        lib = _lib.load()
        band_db = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        nblk = lib.advstep_lfcc_block_count(B, M, NF)
        block_max = torch.empty(max(nblk, 1), dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        out = torch.empty((B, NF, K), dtype=torch.float32, device=dev)
        frag = dct_fragments(dct)
        with _Launch("lfcc_forward", dev):
            st = lib.advstep_lfcc_bands_f32(sr.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(), tables.span,
                                            band_db.data_ptr(), block_max.data_ptr(), B, F, M, NF, _stream(dev))
            _lib.check(st, "advstep_lfcc_bands_f32")
            st = lib.advstep_lfcc_max_project_f32(band_db.data_ptr(), dct.data_ptr(), _ptr(frag), block_max.data_ptr(), nblk,
                                                  stats.data_ptr(), top_db, out.data_ptr(), B, M, NF, K, _stream(dev))
            _lib.check(st, "advstep_lfcc_max_project_f32")
        ctx.save_for_backward(sr, band_db, dct, tables.fbt_start, tables.fbt_w)
        ctx.stats, ctx.frag = stats, frag
        ctx.meta = (B, F, NF, M, K, tables.span_t, float(top_db))
        ctx.backward_calls = 0
        ctx.ties_in_backward = frag is not None and M == 128 and K == 80
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, gout):
        sr, band_db, dct, fbt_start, fbt_w = ctx.saved_tensors
        stats, frag = ctx.stats, ctx.frag
        B, F, NF, M, K, span_t, top_db = ctx.meta
        dev = gout.device
        go = gout.transpose(1, 2).contiguous()     # (B, NF, K); a no-op when the consumer is frame-major
        lib = _lib.load()
        dband = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        dspec = torch.empty((B, NF, F, 2), dtype=torch.float32, device=dev)
        _rearm(ctx, stats)
        with _Launch("lfcc_backward", dev):
            st = lib.advstep_lfcc_project_backward_zero_f32(go.data_ptr(), dct.data_ptr(), _ptr(frag),
                                                            band_db.data_ptr(), stats.data_ptr(), top_db, dband.data_ptr(), B, M, NF,
                                                            K, 0, 0, _stream(dev))
            _lib.check(st, "advstep_lfcc_project_backward_zero_f32")
            st = lib.advstep_lfcc_floor_fixup_f32(band_db.data_ptr(), stats.data_ptr(), dband.data_ptr(), band_db.numel(),
                                                  _stream(dev))
            _lib.check(st, "advstep_lfcc_floor_fixup_f32")
            st = lib.advstep_lfcc_bands_backward_f32(dband.data_ptr(), sr.data_ptr(), fbt_start.data_ptr(),
                                                     fbt_w.data_ptr(), span_t, dspec.data_ptr(), B, F, M, NF, 0,
                                                     _stream(dev))
            _lib.check(st, "advstep_lfcc_bands_backward_f32")
        return torch.view_as_complex(dspec).transpose(1, 2), None, None, None


class _LfccFromWaveform(torch.autograd.Function):
    """The whole LFCC frontend: framing kernel -> rocFFT r2c -> tail kernels; backward: tail kernels -> rocFFT c2r ->
    overlap-add kernel.  No padded copy, no strided-frame clone, no index_add."""

    @staticmethod
    def forward(ctx, x, window, hop, tables: FilterbankTables, dct, top_db: float):
        B, T = x.shape
        nfft = window.numel()
        NF = 1 + T // hop
        F = nfft // 2 + 1
        M, K = dct.shape
        dev = x.device
        lib = _lib.load()
        frames = torch.empty((B, NF, nfft), dtype=torch.float32, device=dev)
        with _Launch("stft_frames", dev):
            st = lib.advstep_stft_frames_f32(x.data_ptr(), window.data_ptr(), frames.data_ptr(), B, T, NF, hop, nfft,
                                             _stream(dev))
        _lib.check(st, "advstep_stft_frames_f32")
        sr = torch.empty((B, NF, F, 2), dtype=torch.float32, device=dev)      # interleaved complex spectrum
        if not (_direct_fft_enabled() and fft_plans.rfft_into(frames.view(B * NF, nfft), sr.view(B * NF, F, 2))):
            sr = torch.view_as_real(torch.fft.rfft(frames, dim=-1))           # same library, plus a defensive input clone
        del frames
        band_db = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        nblk = lib.advstep_lfcc_block_count(B, M, NF)
        block_max = torch.empty(max(nblk, 1), dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        out = torch.empty((B, NF, K), dtype=torch.float32, device=dev)
        frag = dct_fragments(dct)
        with _Launch("lfcc_forward", dev):
            st = lib.advstep_lfcc_bands_f32(sr.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(), tables.span,
                                            band_db.data_ptr(), block_max.data_ptr(), B, F, M, NF, _stream(dev))
            _lib.check(st, "advstep_lfcc_bands_f32")
            st = lib.advstep_lfcc_max_project_f32(band_db.data_ptr(), dct.data_ptr(), _ptr(frag), block_max.data_ptr(), nblk,
                                                  stats.data_ptr(), top_db, out.data_ptr(), B, M, NF, K, _stream(dev))
            _lib.check(st, "advstep_lfcc_max_project_f32")
        ctx.save_for_backward(sr, band_db, dct, tables.fbt_start, tables.fbt_w, window)
        ctx.stats, ctx.frag = stats, frag
        ctx.meta = (B, T, F, NF, M, K, tables.span_t, float(top_db), hop, nfft)
        ctx.backward_calls = 0
        ctx.ties_in_backward = frag is not None and M == 128 and K == 80
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, gout):
        sr, band_db, dct, fbt_start, fbt_w, window = ctx.saved_tensors
        stats, frag = ctx.stats, ctx.frag
        B, T, F, NF, M, K, span_t, top_db, hop, nfft = ctx.meta
        dev = gout.device
        go = gout.transpose(1, 2).contiguous()
        lib = _lib.load()
        dband = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        dspec = torch.empty((B, NF, F, 2), dtype=torch.float32, device=dev)
        _rearm(ctx, stats)
        with _Launch("lfcc_backward", dev):
            st = lib.advstep_lfcc_project_backward_zero_f32(go.data_ptr(), dct.data_ptr(), _ptr(frag),
                                                            band_db.data_ptr(), stats.data_ptr(), top_db, dband.data_ptr(), B, M, NF,
                                                            K, 0, 0, _stream(dev))
            _lib.check(st, "advstep_lfcc_project_backward_zero_f32")
            st = lib.advstep_lfcc_floor_fixup_f32(band_db.data_ptr(), stats.data_ptr(), dband.data_ptr(), band_db.numel(),
                                                  _stream(dev))
            _lib.check(st, "advstep_lfcc_floor_fixup_f32")
            st = lib.advstep_lfcc_bands_backward_f32(dband.data_ptr(), sr.data_ptr(), fbt_start.data_ptr(),
                                                     fbt_w.data_ptr(), span_t, dspec.data_ptr(), B, F, M, NF, 1,
                                                     _stream(dev))
            _lib.check(st, "advstep_lfcc_bands_backward_f32")
        # gradient of the one-sided real FFT = unnormalised c2r inverse of the pre-scaled half spectrum
        dframes = torch.empty((B, NF, nfft), dtype=torch.float32, device=dev)
        if not (_direct_fft_enabled() and fft_plans.irfft_into(dspec.view(B * NF, F, 2), dframes.view(B * NF, nfft))):
            dframes = torch.fft.irfft(torch.view_as_complex(dspec), n=nfft, dim=-1, norm="forward").contiguous()
        dx = torch.empty((B, T), dtype=torch.float32, device=dev)
        with _Launch("stft_overlap_add", dev):
            st = lib.advstep_stft_overlap_add_f32(dframes.data_ptr(), window.data_ptr(), dx.data_ptr(), B, T, NF, hop, nfft,
                                                  _stream(dev))
        _lib.check(st, "advstep_stft_overlap_add_f32")
        return dx, None, None, None, None, None


def _inlds_fft_enabled() -> bool:
    """ADVSTEP_INLDS_FFT=0 keeps framing kernel + hipFFT + filterbank kernel (A/B measurements); default: the fused
    in-LDS FFT kernels of csrc/lfcc_stft.hip."""
    return os.environ.get("ADVSTEP_INLDS_FFT", "1") != "0"


class _LfccFromWaveformFused(torch.autograd.Function):
    """The whole LFCC frontend with the STFT inside the kernels, two launches each way: [framing + FFT + power + filterbank + dB]
    -> [batch max + floor + DCT]; backward: [DCT^T + floor + zero fill of dx] -> [floor fix-up + filterbank^T + spectrum
    recomputed + inverse FFT + window + overlap-add]."""

    @staticmethod
    def forward(ctx, x, window, hop, tables: FilterbankTables, dct, top_db: float):
        B, T = x.shape
        nfft = window.numel()
        NF = 1 + T // hop
        M, K = dct.shape
        dev = x.device
        lib = _lib.load()
        band_db = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        nblk = lib.advstep_stft_bands_block_count(B, NF)
        block_max = torch.empty(max(nblk, 1), dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        out = torch.empty((B, NF, K), dtype=torch.float32, device=dev)
        frag = dct_fragments(dct)
        with _Launch("lfcc_forward", dev, tensors=(x, band_db, band_db, out)):     # waveform in, band rows out + back in, cepstra out
            st = lib.advstep_stft_bands_f32(x.data_ptr(), window.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(),
                                            tables.span, band_db.data_ptr(), block_max.data_ptr(), B, T, NF, hop, nfft, M,
                                            _stream(dev))
            _lib.check(st, "advstep_stft_bands_f32")
            # batch maximum + floor + DCT: one launch (every workgroup reduces the block maxima itself)
            st = lib.advstep_lfcc_max_project_f32(band_db.data_ptr(), dct.data_ptr(), _ptr(frag), block_max.data_ptr(), nblk,
                                                  stats.data_ptr(), top_db, out.data_ptr(), B, M, NF, K, _stream(dev))
            _lib.check(st, "advstep_lfcc_max_project_f32")
        ctx.save_for_backward(x, band_db, dct, tables.fbt_start, tables.fbt_w, window)
        ctx.stats, ctx.frag = stats, frag
        ctx.meta = (B, T, NF, M, K, tables.span_t, float(top_db), hop, nfft)
        ctx.backward_calls = 0
        ctx.ties_in_backward = frag is not None and M == 128 and K == 80
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, gout):
        x, band_db, dct, fbt_start, fbt_w, window = ctx.saved_tensors
        stats, frag = ctx.stats, ctx.frag
        B, T, NF, M, K, span_t, top_db, hop, nfft = ctx.meta
        dev = gout.device
        go = gout.transpose(1, 2).contiguous()
        lib = _lib.load()
        dband = torch.empty((B, NF, M), dtype=torch.float32, device=dev)
        dx = torch.empty((B, T), dtype=torch.float32, device=dev)
        _rearm(ctx, stats)
        with _Launch("lfcc_backward", dev, tensors=(go, band_db, dband, dband, x, dx, dx)):
            # two launches: [DCT^T + floor mask + dB' + tie count; zero-fills dx] -> [floor fix-up folded into the band-gradient
            # load + filterbank^T + FFT pair + overlap-add]
            st = lib.advstep_lfcc_project_backward_zero_f32(go.data_ptr(), dct.data_ptr(), _ptr(frag),
                                                            band_db.data_ptr(), stats.data_ptr(), top_db, dband.data_ptr(), B, M, NF,
                                                            K, dx.data_ptr(), dx.numel(), _stream(dev))
            _lib.check(st, "advstep_lfcc_project_backward_zero_f32")
            st = lib.advstep_stft_bands_backward_fixup_f32(x.data_ptr(), window.data_ptr(), dband.data_ptr(), band_db.data_ptr(),
                                                           stats.data_ptr(), fbt_start.data_ptr(), fbt_w.data_ptr(), span_t,
                                                           dx.data_ptr(), 1, B, T, NF, hop, nfft, M, _stream(dev))
            _lib.check(st, "advstep_stft_bands_backward_fixup_f32")
        return dx, None, None, None, None, None


def lfcc_from_waveform(x: torch.Tensor, window_nfft: torch.Tensor, hop: int, tables: FilterbankTables, dct: torch.Tensor,
                       top_db: float = 80.0) -> torch.Tensor:
    """Waveform (B, T) -> LFCC (B, K, 1 + T // hop); `window_nfft` is the analysis window zero-padded (centred) to n_fft."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise _lib.AdvstepError("lfcc_from_waveform needs a float32 (B, T) waveform on a HIP device (no CPU fallback)")
    if _inlds_fft_enabled() and _lib.load().advstep_stft_bands_supported(window_nfft.numel(), hop, x.shape[1]):
        return _LfccFromWaveformFused.apply(x.contiguous(), window_nfft, hop, tables, dct, top_db)
    return _LfccFromWaveform.apply(x.contiguous(), window_nfft, hop, tables, dct, top_db)


class _MelSpecFromWaveform(torch.autograd.Function):
    """The mel-spec frontend (src/frontends.py:53-79) with the STFT inside the kernels: (B, T) -> (B, 2, M, NF)."""

    @staticmethod
    def forward(ctx, x, window, hop, tables: FilterbankTables):
        B, T = x.shape
        nfft = window.numel()
        NF = 1 + T // hop
        M = tables.fb_start.numel()
        dev = x.device
        out = torch.empty((B, 2, M, NF), dtype=torch.float32, device=dev)
        with _Launch("stft_mel", dev, tensors=(x, out)):
            st = _lib.load().advstep_stft_mel_f32(x.data_ptr(), window.data_ptr(), tables.fb_start.data_ptr(),
                                                  tables.fb_w.data_ptr(), tables.span, out.data_ptr(), B, T, NF, hop, nfft, M,
                                                  _stream(dev))
        _lib.check(st, "advstep_stft_mel_f32")
        ctx.from_output = os.environ.get("ADVSTEP_MEL_BWD_FROM_OUTPUT", "1") != "0"
        ctx.save_for_backward(out if ctx.from_output else x, window, tables.fb_start, tables.fb_w, tables.fbt_start, tables.fbt_w)
        ctx.meta = (B, T, NF, M, tables.span, tables.span_t, hop, nfft)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, window, fb_start, fb_w, fbt_start, fbt_w = ctx.saved_tensors
        B, T, NF, M, span, span_t, hop, nfft = ctx.meta
        dev = gout.device
        go = gout.contiguous()
        dx = torch.empty((B, T), dtype=torch.float32, device=dev)
        if ctx.from_output:
            # d x from d out and out alone (Y = |Y| e^{i phase} is all the gradient needs): no spectrum is recomputed
            with _Launch("stft_mel_backward", dev, tensors=(go, x, dx, dx)):
                st = _lib.load().advstep_stft_mel_backward_from_output_f32(window.data_ptr(), go.data_ptr(), x.data_ptr(),
                                                                           fbt_start.data_ptr(), fbt_w.data_ptr(), span_t,
                                                                           dx.data_ptr(), B, T, NF, hop, nfft, M, _stream(dev))
            _lib.check(st, "advstep_stft_mel_backward_from_output_f32")
            return dx, None, None, None
        with _Launch("stft_mel_backward", dev, tensors=(go, x, dx, dx)):
            st = _lib.load().advstep_stft_mel_backward_f32(x.data_ptr(), window.data_ptr(), go.data_ptr(), fb_start.data_ptr(),
                                                           fb_w.data_ptr(), span, fbt_start.data_ptr(), fbt_w.data_ptr(), span_t,
                                                           dx.data_ptr(), B, T, NF, hop, nfft, M, _stream(dev))
        _lib.check(st, "advstep_stft_mel_backward_f32")
        return dx, None, None, None


def mel_spec_supported(nfft: int, hop: int, T: int, n_mels: int, span: int = 1, span_t: int = 1) -> bool:
    return bool(_lib.load().advstep_stft_bands_supported(nfft, hop, T)) and n_mels <= 80 and span <= 48 and span_t <= 8


def mel_spec_from_waveform(x: torch.Tensor, window_nfft: torch.Tensor, hop: int, tables: FilterbankTables) -> torch.Tensor:
    """Waveform (B, T) -> (B, 2, n_mels, 1 + T // hop): magnitude and phase of the mel-projected complex STFT."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise _lib.AdvstepError("mel_spec_from_waveform needs a float32 (B, T) waveform on a HIP device (no CPU fallback)")
    return _MelSpecFromWaveform.apply(x.contiguous(), window_nfft, hop, tables)


def lfcc_tail(spec: torch.Tensor, tables: FilterbankTables, dct: torch.Tensor, top_db: float = 80.0) -> torch.Tensor:
    """Complex STFT (B, F, NF) -> LFCC (B, K, NF) (view of a frame-major buffer)."""
    return _LfccTail.apply(spec, tables, dct, top_db)
