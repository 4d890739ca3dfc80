"""Synthetic stand-in for `DetectionDataset` (reference item contract: src/datasets/base_dataset.py:180-194).

The real corpora (ASVspoof2021-DF, WaveFake, FakeAVCeleb) and their SoX/torchaudio loading are out of the
hot-path scope; benchmark and parity runs use seeded noise of the shape the reference feeds the attacks:
float32 waveforms of 64 600 samples (the repo's "4 s @ 16 kHz" cut, base_dataset.py:22,27), label 1 =
bonafide / 0 = spoof, and the same 4-tuple per item so the default collate yields
(batch_x (B, 64600) f32, batch_sr (B) i64, batch_y (B) i64, metadata = 4 lists)."""
import torch
from torch.utils.data import Dataset

SAMPLING_RATE = 16_000
NUM_SAMPLES = 64_600  # base_dataset.py:27  (APPLY_PADDING cut)


def synthetic_waveforms(n: int, num_samples: int = NUM_SAMPLES, seed: int = 1234):
    """x ~ N(0, 0.05^2) clipped to [-1, 1], labels ~ randint(0, 2), CPU generator so the
    data are identical on every box and every rank."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, num_samples, generator=g) * 0.05).clamp_(-1.0, 1.0)
    y = torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
    return x, y


class SyntheticDetectionDataset(Dataset):
    def __init__(self, n: int, num_samples: int = NUM_SAMPLES, seed: int = 1234, return_meta: bool = True):
        """return_meta as in base_dataset.py:42,185: the evaluation loop asks for the metadata tuple (4 items per
        utterance), the trainers do not (3 items)."""
        self.x, self.y = synthetic_waveforms(n, num_samples, seed)
        self.seconds = num_samples / SAMPLING_RATE
        self.return_meta = return_meta

    def __len__(self):
        return len(self.y)

    def __getitem__(self, index):
        label = int(self.y[index])
        if not self.return_meta:
            return [self.x[index], SAMPLING_RATE, label]
        meta = ("-" if label == 1 else "synthetic", f"synthetic/{index:08d}.wav", "val", self.seconds)
        return [self.x[index], SAMPLING_RATE, label, meta]
