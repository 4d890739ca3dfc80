"""Device side of the waveform batches (include/advstep_dataset.h).

`pad_tile` is `PadDataset.apply_pad` (src/datasets/base_dataset.py:344-355) + first-channel selection (:104-105) +
PCM16 normalisation (`torchaudio.load(normalize=True)`, :165) over a ragged batch; `RaggedWaveBatch` is how such a
batch travels from the DataLoader workers (raw payloads, concatenated, pinned) to the device — one H2D copy of the file
payloads instead of per-item padded float tensors.  `qual_select` / `gather_rows` serve `AttackAnalyser`.
HIP only: CPU tensors raise (hip_ops._require)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from ..hip_ops import _Launch, _require, _stream

WAVE_F32, WAVE_PCM16 = 0, 1
_KINDS = {torch.float32: WAVE_F32, torch.int16: WAVE_PCM16}
_MAX_ROWS = 65535  # grid.y


def pad_tile(src: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor, cut: int,
             channels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, cut) float32 from the concatenated payloads `src` (1-D float32 or int16, frames interleaved by channel):
    row b = channel 0 of frames offsets[b] + (t mod lengths[b]) * channels[b], cut or tiled to `cut` samples."""
    if src.dtype not in _KINDS:
        raise TypeError(f"src: expected float32 or int16 payloads, got {src.dtype}")
    _require(src, "src", src.dtype), _require(offsets, "offsets", torch.int64), _require(lengths, "lengths", torch.int64)
    if channels is not None:
        _require(channels, "channels", torch.int32)
    B = lengths.numel()
    if src.dim() != 1 or offsets.numel() != B or (channels is not None and channels.numel() != B) or cut < 0:
        raise ValueError(f"src {tuple(src.shape)}, offsets {tuple(offsets.shape)}, lengths {tuple(lengths.shape)}, "
                         f"cut {cut} do not describe a ragged batch")
    dst = torch.empty(B, cut, device=src.device)
    lib = _lib.load()
    for lo in range(0, B, _MAX_ROWS):
        n = min(_MAX_ROWS, B - lo)
        with _Launch("wave_pad_tile", src.device):
            st = lib.advstep_wave_pad_tile_f32(src.data_ptr(), _KINDS[src.dtype], offsets[lo:].data_ptr(),
                                               lengths[lo:].data_ptr(),
                                               None if channels is None else channels[lo:].data_ptr(),
                                               dst[lo:].data_ptr(), n, cut, _stream(src.device))
        _lib.check(st, "advstep_wave_pad_tile_f32")
    return dst


def apply_pad_batch(batch: torch.Tensor, cut: int, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`apply_pad` of every row of a (B, T) float32 device batch; `lengths` (B) int64 marks how much of each row is
    valid (default: all T)."""
    _require(batch, "batch")
    if batch.dim() != 2:
        raise ValueError(f"batch: expected (B, T), got {tuple(batch.shape)}")
    B, T = batch.shape
    if T == 0 and B:
        raise ZeroDivisionError("apply_pad of an empty waveform (base_dataset.py:352 divides by its length)")
    offsets = torch.arange(B, device=batch.device, dtype=torch.int64) * T
    if lengths is None:
        lengths = torch.full((B,), T, device=batch.device, dtype=torch.int64)
    return pad_tile(batch.reshape(-1), offsets, lengths, cut)


@dataclass
class RaggedWaveBatch:
    """B utterances as the decoder left them: one 1-D payload tensor (float32 or int16, channel-interleaved frames),
    per-utterance sample offsets, frame counts and channel counts (host tensors; pinned when a HIP device exists)."""
    payload: torch.Tensor
    offsets: torch.Tensor   # (B) int64, in samples of payload
    lengths: torch.Tensor   # (B) int64, frames
    channels: torch.Tensor  # (B) int32

    def __len__(self) -> int:
        return self.lengths.numel()

    def size(self, dim: int = 0) -> int:  # the evaluation loop asks batch_x.size(0)
        if dim != 0:
            raise IndexError("a ragged batch only has a batch dimension")
        return len(self)

    @classmethod
    def from_arrays(cls, waves: Sequence[np.ndarray]) -> "RaggedWaveBatch":
        """waves[b]: (frames,) or (frames, channels), all int16 or all float32 (a mixed batch is widened to float32
        with torchaudio's normalisation)."""
        from .audio_io import to_float32
        waves = [w[:, None] if w.ndim == 1 else w for w in waves]
        if any(w.shape[0] == 0 for w in waves):
            raise ZeroDivisionError("apply_pad of an empty waveform (base_dataset.py:352 divides by its length)")
        if not all(w.dtype == np.int16 for w in waves):
            waves = [to_float32(w) for w in waves]
        dtype = waves[0].dtype if waves else np.float32
        sizes = np.array([w.size for w in waves], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(waves) else sizes
        payload = torch.empty(int(sizes.sum()), dtype=torch.from_numpy(np.empty(0, dtype)).dtype)
        flat = payload.numpy()
        for w, o in zip(waves, offsets):
            flat[o:o + w.size] = w.reshape(-1)
        out = cls(payload, torch.from_numpy(offsets), torch.tensor([w.shape[0] for w in waves], dtype=torch.int64),
                  torch.tensor([w.shape[1] for w in waves], dtype=torch.int32))
        return out

    def pin_memory(self) -> "RaggedWaveBatch":  # DataLoader(pin_memory=True) calls this on custom batch types
        return RaggedWaveBatch(self.payload.pin_memory(), self.offsets.pin_memory(), self.lengths.pin_memory(),
                               self.channels.pin_memory())

    def to_padded(self, device, cut: int) -> torch.Tensor:
        """Upload the payloads and decode + mono + pad/tile them on the device: (B, cut) float32."""
        dev = torch.device(device)
        mono = bool((self.channels == 1).all())
        return pad_tile(self.payload.to(dev, non_blocking=True), self.offsets.to(dev, non_blocking=True),
                        self.lengths.to(dev, non_blocking=True), cut,
                        None if mono else self.channels.to(dev, non_blocking=True))


def qual_select(y: torch.Tensor, pred_noattack_label: torch.Tensor,
                pred_label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """attacks_analysis.py:78-84,100-106 on the device: (rows (B) int32 — false positives then false negatives, each
    ascending —, counts (2) int32 = [n_fp, n_fn])."""
    _require(y, "y", torch.int64), _require(pred_noattack_label, "pred_noattack_label", torch.int32)
    _require(pred_label, "pred_label", torch.int32)
    B = y.numel()
    if pred_noattack_label.numel() != B or pred_label.numel() != B:
        raise ValueError("y, pred_noattack_label and pred_label must have one entry per utterance")
    rows = torch.empty(max(B, 1), dtype=torch.int32, device=y.device)
    counts = torch.empty(2, dtype=torch.int32, device=y.device)
    with _Launch("qual_select", y.device):
        st = _lib.load().advstep_qual_select(y.data_ptr(), pred_noattack_label.data_ptr(), pred_label.data_ptr(), B,
                                             rows.data_ptr(), counts.data_ptr(), _stream(y.device))
    _lib.check(st, "advstep_qual_select")
    return rows, counts


def gather_rows(src: torch.Tensor, rows: torch.Tensor, n: Optional[int] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dst[i] = src[rows[i]] for the first n entries of the device index list `rows` (int32).  `out` may be a pinned
    host tensor's device twin; by default a new (n, T) device tensor."""
    _require(src, "src"), _require(rows, "rows", torch.int32)
    if src.dim() != 2:
        raise ValueError(f"src: expected (B, T), got {tuple(src.shape)}")
    n = rows.numel() if n is None else n
    if n > rows.numel():
        raise ValueError(f"n = {n} exceeds the {rows.numel()} indices given")
    T = src.shape[1]
    dst = torch.empty(n, T, device=src.device) if out is None else _require(out, "out")
    if dst.shape != (n, T) or dst.device != src.device:
        raise ValueError(f"out {tuple(dst.shape)} does not match ({n}, {T})")
    lib = _lib.load()
    for lo in range(0, n, _MAX_ROWS):
        m = min(_MAX_ROWS, n - lo)
        with _Launch("wave_gather_rows", src.device):
            st = lib.advstep_wave_gather_rows_f32(src.data_ptr(), rows[lo:].data_ptr(), dst[lo:].data_ptr(), m, T,
                                                  _stream(src.device))
        _lib.check(st, "advstep_wave_gather_rows_f32")
    return dst
