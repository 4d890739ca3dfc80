"""Real-corpus dataset base classes (reference: src/datasets/base_dataset.py).

Same item contract as the reference — `[waveform (64600,) f32, sample_rate, label (1 = bonafide), (attack_type, path,
subset, seconds)]` — and the same deterministic train/test/val partition (`split_samples`), so a corpus on disk yields
the same utterances in the same order.  What differs is where the sample-level work runs:

  * `wavefake_preprocessing_on_batch` (reference :122-148: batch.cpu() -> Python loop over rows -> stack -> .to(device))
    keeps a device batch on the device: first channel + cut/tile is one `wave_ops.pad_tile` launch.
  * `device_pad=True` datasets hand the DataLoader the decoded payload as it is on disk (PCM16 stays int16);
    `ragged_collate` packs a batch into one `RaggedWaveBatch`, and the evaluation loop uploads it once and decodes +
    pads on the device (`RaggedWaveBatch.to_padded`).  The default (`device_pad=False`) is the reference's item
    contract, computed in the DataLoader worker exactly like the reference does (`PadDataset.apply_pad`).
  * SoX effects (`apply_trim`, `resample_wave`, `process_phone_call`, reference :278-337) go through
    `torchaudio.sox_effects`, a binary dependency that is absent here and has no bit-level specification to restate.
    They dispatch to a backend registered with `register_sox_backend`; without one they raise `SoxUnavailableError`
    (no approximation is substituted).  `WAVE_FAKE_TRIM` keeps the reference's default (True): pass
    `wave_fake_trim=False` / `--no_trim` to run without SoX.
"""
from __future__ import annotations

import logging
import math
import random
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from . import audio_io
from .wave_ops import RaggedWaveBatch, apply_pad_batch

LOGGER = logging.getLogger(__name__)

WAVE_FAKE_INTERFACE = True
WAVE_FAKE_SR = 16_000
WAVE_FAKE_TRIM = True
WAVE_FAKE_NORMALIZE = True
WAVE_FAKE_CELL_PHONE = False
WAVE_FAKE_PAD = True
WAVE_FAKE_CUT = 64_600

# reference :29-33 — drop every stretch of silence longer than 0.2 s (below 1 % of the file's peak volume)
SOX_SILENCE = [["silence", "1", "0.2", "1%", "-1", "0.2", "1%"]]
SOX_PHONE_CALL = [["lowpass", "4000"],
                  ["compand", "0.02,0.05", "-60,-60,-30,-10,-20,-8,-5,-8,-2,-8", "-8", "-7", "0.05"],
                  ["rate", "8000"]]


class SoxUnavailableError(RuntimeError):
    """A SoX effect chain was requested and no backend is registered."""


# fn(waveform (C, L) f32 CPU tensor, sample_rate, effects) -> (waveform, sample_rate): torchaudio.sox_effects.apply_effects_tensor
_sox_backend: Optional[Callable] = None
_codec_backend: Optional[Callable] = None  # torchaudio.functional.apply_codec


def register_sox_backend(apply_effects_tensor: Optional[Callable], apply_codec: Optional[Callable] = None) -> None:
    """Plug in `torchaudio.sox_effects.apply_effects_tensor` (and optionally `torchaudio.functional.apply_codec`)."""
    global _sox_backend, _codec_backend
    _sox_backend, _codec_backend = apply_effects_tensor, apply_codec


def _sox(waveform: torch.Tensor, sample_rate: int, effects: List[List[str]]):
    if _sox_backend is None:
        raise SoxUnavailableError(
            f"SoX effect chain {effects} requested, but SoX/torchaudio is not part of this build and its algorithms are "
            "not restated. Register one with datasets.base_dataset.register_sox_backend("
            "torchaudio.sox_effects.apply_effects_tensor), or disable the step (wave_fake_trim=False / --no_trim; "
            "audio already at 16 kHz needs no resampling).")
    return _sox_backend(waveform, int(sample_rate), effects)


class SimpleAudioFakeDataset(Dataset):
    def __init__(self, subset, transform=None, return_label: bool = True, return_meta: bool = False,
                 return_raw: bool = False, device_pad: bool = False, wave_fake_trim: Optional[bool] = None):
        self.transform = transform
        self.samples = pd.DataFrame()
        self.subset = subset
        self.allowed_attacks = None
        self.partition_ratio = None
        self.seed = None
        self.return_label = return_label
        self.return_meta = return_meta
        self.return_raw = return_raw
        self.device_pad = device_pad          # build-specific: items carry the undecoded payload (see module docstring)
        self.wave_fake_trim = wave_fake_trim  # build-specific: None = the reference's WAVE_FAKE_TRIM default

    # ------------------------------------------------------------------------------------------------
    # partitioning (reference :55-70)
    # ------------------------------------------------------------------------------------------------
    def split_samples(self, samples_list):
        """Seeded shuffle, then [0, p) -> train, [p, p+s) -> test, the rest -> val."""
        if isinstance(samples_list, pd.DataFrame):
            ordered = samples_list.sort_values(by=list(samples_list.columns))
            ordered = ordered.sample(frac=1, random_state=self.seed)
            take = lambda lo, hi: ordered.iloc[lo:hi]  # noqa: E731
        else:
            ordered = sorted(samples_list)
            random.seed(self.seed)
            random.shuffle(ordered)
            take = lambda lo, hi: ordered[lo:hi]  # noqa: E731
        n = len(ordered)
        p, s = self.partition_ratio
        first, second = int(p * n), int((p + s) * n)
        bounds = {"train": (0, first), "test": (first, second), "val": (second, n)}
        return take(*bounds[self.subset])

    def df2tuples(self):
        self.samples = [(str(row["path"]), row["label"], row["attack_type"]) for _, row in self.samples.iterrows()]
        return self.samples

    # ------------------------------------------------------------------------------------------------
    # sample-level preprocessing (reference :83-148)
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def wavefake_preprocessing(waveform, sample_rate, wave_fake_sr: Optional[int] = None,
                               wave_fake_trim: Optional[bool] = None, wave_fake_cell_phone: Optional[bool] = None,
                               wave_fake_pad: Optional[bool] = None, wave_fake_cut: Optional[int] = None):
        """One utterance, on the host (DataLoader worker): resample -> first channel -> trim -> phone call -> pad."""
        wave_fake_sr = WAVE_FAKE_SR if wave_fake_sr is None else wave_fake_sr
        wave_fake_trim = WAVE_FAKE_TRIM if wave_fake_trim is None else wave_fake_trim
        wave_fake_cell_phone = WAVE_FAKE_CELL_PHONE if wave_fake_cell_phone is None else wave_fake_cell_phone
        wave_fake_pad = WAVE_FAKE_PAD if wave_fake_pad is None else wave_fake_pad
        wave_fake_cut = WAVE_FAKE_CUT if wave_fake_cut is None else wave_fake_cut

        if sample_rate != wave_fake_sr and wave_fake_sr != -1:
            waveform, sample_rate = AudioDataset.resample_wave(waveform, sample_rate, wave_fake_sr)
        if waveform.dim() > 1 and waveform.shape[0] > 1:
            waveform = waveform[:1, ...]
        if wave_fake_trim:
            waveform, sample_rate = AudioDataset.apply_trim(waveform, sample_rate)
        if wave_fake_cell_phone:
            waveform, sample_rate = AudioDataset.process_phone_call(waveform, sample_rate)
        if wave_fake_pad:
            waveform = PadDataset.apply_pad(waveform, wave_fake_cut)
        return waveform, sample_rate

    @staticmethod
    def wavefake_preprocessing_on_batch(batch_waveform, batch_sample_rate, *args, **kwargs):
        """A (B, T) batch through `wavefake_preprocessing`, row by row in the reference.  Here a device batch stays on
        the device when no SoX step is involved (all rows at the target rate, trim and phone-call off): rows are
        already single-channel, so what is left is `apply_pad`, one kernel launch.  With a SoX step the batch takes the
        reference's host round trip through the registered backend."""
        names = ("wave_fake_sr", "wave_fake_trim", "wave_fake_cell_phone", "wave_fake_pad", "wave_fake_cut")
        opts = dict(zip(names, args))
        opts.update(kwargs)
        sr = WAVE_FAKE_SR if opts.get("wave_fake_sr") is None else opts["wave_fake_sr"]
        trim = WAVE_FAKE_TRIM if opts.get("wave_fake_trim") is None else opts["wave_fake_trim"]
        phone = WAVE_FAKE_CELL_PHONE if opts.get("wave_fake_cell_phone") is None else opts["wave_fake_cell_phone"]
        pad = WAVE_FAKE_PAD if opts.get("wave_fake_pad") is None else opts["wave_fake_pad"]
        cut = WAVE_FAKE_CUT if opts.get("wave_fake_cut") is None else opts["wave_fake_cut"]

        rates = batch_sample_rate.cpu()
        needs_sox = trim or phone or (sr != -1 and bool((rates != sr).any()))
        if batch_waveform.is_cuda and not needs_sox:
            out = apply_pad_batch(batch_waveform.contiguous(), cut) if pad else batch_waveform.unsqueeze(1)
            return out, batch_sample_rate.clone()

        device_waveform, device_rate = batch_waveform.device, batch_sample_rate.device
        waveforms, sample_rates = [], []
        for waveform, sample_rate in zip(batch_waveform.cpu(), rates):
            waveform, sample_rate = SimpleAudioFakeDataset.wavefake_preprocessing(
                waveform.unsqueeze(0), sample_rate, **opts)
            waveforms.append(waveform)
            sample_rates.append(torch.tensor([sample_rate]))
        return torch.stack(waveforms, dim=0).to(device_waveform), torch.cat(sample_rates, dim=0).to(device_rate)

    # ------------------------------------------------------------------------------------------------
    # items (reference :150-205)
    # ------------------------------------------------------------------------------------------------
    def _row(self, index):
        if isinstance(self.samples, pd.DataFrame):
            sample = self.samples.iloc[index]
            attack_type = sample["attack_type"] if "attack_type" in sample else float("nan")
            if not isinstance(attack_type, str) and (attack_type is None or math.isnan(attack_type)):
                attack_type = "N/A"
            return str(sample["path"]), sample["label"], attack_type
        return self.samples[index]

    def __getitem__(self, index):
        path, label, attack_type = self._row(index)
        trim = self.wave_fake_trim
        if self.device_pad:
            data, sample_rate = audio_io.decode_raw(path)
            seconds = data.shape[0] / sample_rate
            if sample_rate != WAVE_FAKE_SR or (not self.return_raw and (WAVE_FAKE_TRIM if trim is None else trim)):
                # a SoX step is needed: do it on the host, ship the float result (still unpadded)
                waveform = torch.from_numpy(np.ascontiguousarray(audio_io.to_float32(data).T))
                waveform, sample_rate = self.wavefake_preprocessing(
                    waveform, sample_rate, wave_fake_trim=False if self.return_raw else trim,
                    wave_fake_cell_phone=False if self.return_raw else None, wave_fake_pad=False)
                data = waveform.reshape(-1, 1).numpy()
            elif data.dtype != np.int16:
                data = audio_io.to_float32(data)
            item = [RawWave(data), sample_rate]
        else:
            waveform, sample_rate = audio_io.load(path, normalize=WAVE_FAKE_NORMALIZE)
            seconds = len(waveform[0]) / sample_rate
            if self.return_raw:
                waveform, sample_rate = self.wavefake_preprocessing(waveform, sample_rate, wave_fake_trim=False,
                                                                    wave_fake_cell_phone=False)
            else:
                waveform, sample_rate = self.wavefake_preprocessing(waveform, sample_rate, wave_fake_trim=trim)
            item = [waveform, sample_rate]
        if self.return_label:
            item.append(1 if label == "bonafide" else 0)
        if self.return_meta:
            item.append((attack_type, path, self.subset, seconds))
        return item

    def __len__(self):
        return len(self.samples)


class RawWave:
    """One decoded, unpadded utterance on its way through the DataLoader: (frames, channels) int16 or float32."""
    __slots__ = ("data",)

    def __init__(self, data: np.ndarray):
        self.data = data


def ragged_collate(items: Sequence[Sequence]):
    """collate_fn for `device_pad=True` datasets: RawWave items -> one RaggedWaveBatch, the rest like default_collate."""
    from torch.utils.data import default_collate
    columns = list(zip(*items))
    batch = [RaggedWaveBatch.from_arrays([w.data for w in columns[0]])]
    batch.extend(default_collate(list(col)) for col in columns[1:])
    return batch


class AudioDataset(Dataset):
    """Every wav file under a directory (or an explicit list), loaded, resampled and trimmed (reference :208-337)."""

    def __init__(self, directory_or_path_list: Union[str, Path, List[Union[str, Path]]], sample_rate: int = 16_000,
                 amount: Optional[int] = None, normalize: bool = True, trim: bool = True, phone_call: bool = False):
        super().__init__()
        self.trim, self.sample_rate, self.normalize, self.phone_call = trim, sample_rate, normalize, phone_call
        if isinstance(directory_or_path_list, list):
            paths = directory_or_path_list
        elif isinstance(directory_or_path_list, (str, Path)):
            directory = Path(directory_or_path_list)
            if not directory.exists():
                raise IOError(f"Directory does not exists: {directory}")
            from ..utils import find_wav_files
            paths = find_wav_files(directory)
            if paths is None:
                raise IOError(f"Directory did not contain wav files: {directory}")
        else:
            raise TypeError(f"Supplied unsupported type for argument directory_or_path_list "
                            f"{type(directory_or_path_list)}!")
        self._paths = paths if amount is None else paths[:amount]

    def __getitem__(self, index: int) -> Tuple[torch.Tensor, int]:
        path = self._paths[index]
        waveform, sample_rate = audio_io.load(path, normalize=self.normalize)
        if sample_rate != self.sample_rate:
            waveform, sample_rate = self.resample(path, self.sample_rate, self.normalize)
        if self.trim:
            waveform, sample_rate = self.apply_trim(waveform, sample_rate)
        if self.phone_call:
            waveform, sample_rate = self.process_phone_call(waveform, sample_rate)
        return waveform, sample_rate

    def __len__(self) -> int:
        return len(self._paths)

    @staticmethod
    def apply_trim(waveform, sample_rate):
        trimmed, trimmed_rate = _sox(waveform, sample_rate, SOX_SILENCE)
        if trimmed.size()[1] > 0:  # an all-silent file is kept as it is
            return trimmed, trimmed_rate
        return waveform, sample_rate

    @staticmethod
    def resample_wave(waveform, sample_rate, target_sample_rate):
        return _sox(waveform, sample_rate, [["rate", f"{target_sample_rate}"]])

    @staticmethod
    def resample(path, target_sample_rate, normalize=True):
        waveform, sample_rate = audio_io.load(path, normalize=normalize)
        return _sox(waveform, sample_rate, [["rate", f"{target_sample_rate}"]])

    @staticmethod
    def process_phone_call(waveform, sample_rate):
        waveform, sample_rate = _sox(waveform, sample_rate, SOX_PHONE_CALL)
        if _codec_backend is None:
            raise SoxUnavailableError("the GSM codec step needs torchaudio.functional.apply_codec "
                                      "(register_sox_backend(..., apply_codec=...))")
        return _codec_backend(waveform, sample_rate, format="gsm"), sample_rate


class PadDataset(Dataset):
    def __init__(self, dataset: Dataset, cut: int = 64600, label=None):
        self.dataset, self.cut, self.label = dataset, cut, label  # cut: 4 s at 16 kHz, the ASVspoof default

    def __getitem__(self, index):
        waveform, sample_rate = self.dataset[index]
        waveform = self.apply_pad(waveform, self.cut)
        return (waveform, sample_rate) if self.label is None else (waveform, sample_rate, self.label)

    def __len__(self):
        return len(self.dataset)

    @staticmethod
    def apply_pad(waveform, cut):
        """Host form (one utterance, in a DataLoader worker): the first `cut` samples, a short waveform repeated until
        it fills them (reference :344-355).  The batched device form is `wave_ops.pad_tile`."""
        waveform = waveform.squeeze(0)
        n = waveform.shape[0]
        if n >= cut:
            return waveform[:cut]
        repeats = int(cut / n) + 1
        return waveform.repeat(repeats)[:cut]
