"""Audio file formats either side of the attack loop.

Reading — the reference calls `torchaudio.load(path, normalize=True)` (src/datasets/base_dataset.py:165), which is not
installed here.  RIFF/WAVE (WaveFake, LJSpeech, JSUT) is parsed by this module; `load` returns what torchaudio returns,
a float32 tensor `(channels, frames)` in [-1, 1) and the sample rate.  `read_wav_raw` keeps the file's own sample type
so that PCM16 payloads can be uploaded as they are and decoded on the device (`wave_ops.pad_tile`).  FLAC (ASVspoof)
and MP3 (FakeAVCeleb) need a codec: `register_decoder(".flac", fn)` plugs one in (e.g. `soundfile.read`); without
one those files raise `AudioDecodeError` — there is no silent substitute.

Writing — `AttackAnalyser` (src/aa/qualitative/attacks_analysis.py:127-137) saves float32 waveforms with
`scipy.io.wavfile.write(rate=16_000)`: an IEEE-float WAVE file (format tag 3, 18-byte fmt chunk, a `fact` chunk, the
`data` chunk).  `write_wav_f32` produces the same bytes straight from a (pinned) host buffer;
tests/test_datasets.py compares it with scipy's own output.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Callable, Dict, Tuple, Union

import numpy as np
import torch

PathLike = Union[str, Path]

WAVE_FORMAT_PCM, WAVE_FORMAT_IEEE_FLOAT, WAVE_FORMAT_EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


class AudioDecodeError(RuntimeError):
    """The file cannot be decoded by this build (unknown container/codec and no registered decoder)."""


# extension (lower case, with dot) -> fn(path) -> (ndarray (frames, channels) or (frames,), sample_rate)
_decoders: Dict[str, Callable[[PathLike], Tuple[np.ndarray, int]]] = {}


def register_decoder(extension: str, fn: Callable[[PathLike], Tuple[np.ndarray, int]]) -> None:
    """Plug in a codec for a file extension (".flac", ".mp3").  `fn(path)` returns (samples, sample_rate) with samples
    shaped (frames,) or (frames, channels), integer PCM or float in [-1, 1]."""
    _decoders[extension.lower()] = fn


def _chunks(buf: bytes, path: PathLike):
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise AudioDecodeError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    while pos + 8 <= len(buf):
        tag, size = buf[pos:pos + 4], struct.unpack_from("<I", buf, pos + 4)[0]
        yield tag, pos + 8, min(size, len(buf) - pos - 8)  # a truncated last chunk is read as far as it goes
        pos += 8 + size + (size & 1)


def read_wav_raw(path: PathLike) -> Tuple[np.ndarray, int]:
    """The data chunk as stored: ndarray (frames, channels) of uint8 / int16 / int32 / float32 / float64 (24-bit PCM is
    widened to int32, left-justified like libsndfile does) and the sample rate."""
    buf = Path(path).read_bytes()
    fmt = None
    for tag, start, size in _chunks(buf, path):
        if tag == b"fmt ":
            if size < 16:
                raise AudioDecodeError(f"{path}: fmt chunk of {size} bytes")
            code, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", buf, start)
            if code == WAVE_FORMAT_EXTENSIBLE and size >= 26:
                code = struct.unpack_from("<H", buf, start + 24)[0]  # first two bytes of the sub-format GUID
            fmt = (code, channels, rate, block_align, bits)
        elif tag == b"data":
            if fmt is None:
                raise AudioDecodeError(f"{path}: data chunk before fmt chunk")
            code, channels, rate, block_align, bits = fmt
            if channels < 1:
                raise AudioDecodeError(f"{path}: {channels} channels")
            width = bits // 8
            frames = size // (width * channels)
            raw = np.frombuffer(buf, dtype=np.uint8, count=frames * channels * width, offset=start)
            if code == WAVE_FORMAT_PCM and bits == 8:
                data = raw.copy()
            elif code == WAVE_FORMAT_PCM and bits == 16:
                data = raw.view("<i2").copy()
            elif code == WAVE_FORMAT_PCM and bits == 24:
                b3 = raw.reshape(-1, 3).astype(np.uint32)
                data = ((b3[:, 0] << 8) | (b3[:, 1] << 16) | (b3[:, 2] << 24)).view(np.int32)
            elif code == WAVE_FORMAT_PCM and bits == 32:
                data = raw.view("<i4").copy()
            elif code == WAVE_FORMAT_IEEE_FLOAT and bits == 32:
                data = raw.view("<f4").copy()
            elif code == WAVE_FORMAT_IEEE_FLOAT and bits == 64:
                data = raw.view("<f8").copy()
            else:
                raise AudioDecodeError(f"{path}: unsupported WAVE coding (format tag {code:#x}, {bits} bits)")
            return data.reshape(frames, channels), rate
    raise AudioDecodeError(f"{path}: no data chunk")


def decode_raw(path: PathLike) -> Tuple[np.ndarray, int]:
    """(frames, channels) samples in the file's own type + sample rate, through the WAVE parser or a registered codec."""
    ext = Path(path).suffix.lower()
    if ext in _decoders:
        data, rate = _decoders[ext](path)
        data = np.asarray(data)
        return (data[:, None] if data.ndim == 1 else data), int(rate)
    if ext in (".wav", ".wave"):
        return read_wav_raw(path)
    raise AudioDecodeError(
        f"{path}: no decoder for '{ext}' files in this build (WAVE is built in; FLAC/MP3 need a codec library — "
        f"call datasets.audio_io.register_decoder('{ext}', fn), e.g. with soundfile.read)")


def to_float32(data: np.ndarray) -> np.ndarray:
    """torchaudio.load(normalize=True): integer PCM scaled to [-1, 1) by its full-scale power of two."""
    if data.dtype == np.uint8:
        return (data.astype(np.float32) - 128.0) / 128.0
    if data.dtype == np.int16:
        return data.astype(np.float32) / 32768.0
    if data.dtype == np.int32:
        return (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    return data.astype(np.float32, copy=False)


def load(path: PathLike, normalize: bool = True) -> Tuple[torch.Tensor, int]:
    """Stand-in for `torchaudio.load` as the reference uses it (base_dataset.py:165,262): tensor (channels, frames)."""
    data, rate = decode_raw(path)
    if normalize or data.dtype.kind == "f":
        data = to_float32(data)
    return torch.from_numpy(np.ascontiguousarray(data.T)), rate


def wav_f32_header(n_frames: int, rate: int, channels: int = 1) -> bytes:
    """The bytes scipy.io.wavfile.write puts in front of float32 data (scipy/io/wavfile.py `write`, non-PCM branch)."""
    block = 4 * channels
    fmt = struct.pack("<HHIIHH", WAVE_FORMAT_IEEE_FLOAT, channels, rate, rate * block, block, 32) + b"\x00\x00"
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    body += b"fact" + struct.pack("<II", 4, n_frames)
    body += b"data" + struct.pack("<I", n_frames * block)
    return b"RIFF" + struct.pack("<I", len(body) + n_frames * block) + body


def write_wav_f32(path: PathLike, rate: int, data) -> None:
    """`scipy.io.wavfile.write(path, rate, data)` for float32 data (frames,) or (frames, channels), same bytes."""
    if isinstance(data, torch.Tensor):
        data = data.detach().cpu().numpy()
    data = np.ascontiguousarray(data, dtype="<f4")
    if data.ndim not in (1, 2):
        raise ValueError(f"expected (frames,) or (frames, channels), got {data.shape}")
    channels = 1 if data.ndim == 1 else data.shape[1]
    nbytes = data.size * 4
    with open(path, "wb") as f:
        f.write(wav_f32_header(data.shape[0], rate, channels))
        f.write(memoryview(data).cast("B"))
        if nbytes & 1:  # RIFF chunks are word aligned (never the case for float32)
            f.write(b"\x00")
