"""WaveFake (reference: src/datasets/wavefake_dataset.py).

Layout on disk: vocoder output in `<root>/generated_audio/<corpus>_<vocoder>/*.wav` (the attack type is the folder name
after its first underscore), genuine speech in `<root>/real_audio/jsut_ver1.1/basic5000/wav` and
`<root>/real_audio/LJSpeech-1.1/wavs`.  Generated and genuine files are partitioned separately (70 / 15 / 15 %, seed 45),
generated files listed first."""
from pathlib import Path

import pandas as pd

from .base_dataset import SimpleAudioFakeDataset

_VOCODERS = ["multi_band_melgan", "melgan_large", "parallel_wavegan", "waveglow", "full_band_melgan", "melgan", "hifiGAN"]
WAVEFAKE_SPLIT = {"train": list(_VOCODERS), "test": list(_VOCODERS), "val": list(_VOCODERS),
                  "partition_ratio": [0.7, 0.15], "seed": 45}

_COLUMNS = ("user_id", "sample_name", "attack_type", "label", "path")


class WaveFakeDataset(SimpleAudioFakeDataset):
    fake_data_path = "generated_audio"
    jsut_real_data_path = "real_audio/jsut_ver1.1/basic5000/wav"
    ljspeech_real_data_path = "real_audio/LJSpeech-1.1/wavs"

    def __init__(self, path, subset="train", transform=None):
        super().__init__(subset, transform)
        self.path = Path(path)
        self.fold_subset = subset
        self.allowed_attacks = WAVEFAKE_SPLIT[subset]
        self.partition_ratio = WAVEFAKE_SPLIT["partition_ratio"]
        self.seed = WAVEFAKE_SPLIT["seed"]
        self.samples = pd.concat([self.get_fake_samples(), self.get_real_samples()], ignore_index=True)

    @staticmethod
    def get_attack_from_path(path):
        return path.parent.name.split("_", maxsplit=1)[-1]

    def filter_samples_by_attack(self, samples_list):
        return [s for s in samples_list if self.get_attack_from_path(s) in self.allowed_attacks]

    def get_fake_samples(self):
        files = self.filter_samples_by_attack(list((self.path / self.fake_data_path).glob("*/*.wav")))
        rows = [(None, "_".join(f.stem.split("_")[:-1]), self.get_attack_from_path(f), "spoof", f)
                for f in self.split_samples(files)]
        return pd.DataFrame(rows, columns=_COLUMNS)

    def get_real_samples(self):
        files = list((self.path / self.jsut_real_data_path).glob("*.wav"))
        files += list((self.path / self.ljspeech_real_data_path).glob("*.wav"))
        rows = [(None, f.stem, "-", "bonafide", f) for f in self.split_samples(files)]
        return pd.DataFrame(rows, columns=_COLUMNS)
