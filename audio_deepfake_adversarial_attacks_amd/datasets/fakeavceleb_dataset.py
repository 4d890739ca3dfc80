"""FakeAVCeleb audio track (reference: src/datasets/fakeavceleb_dataset.py).

`<root>/FakeAVCeleb-audio/meta_data.csv` lists every clip (columns source, method, type, filename, path, ...); the audio
sits in `<root>/FakeAVCeleb-audio/<path minus its first component>/<filename>.mp3`.  Fake clips (`type` ending in
`FakeAudio`) are partitioned per generation method, genuine clips (`method == real`, `RealAudio`) on their own —
70 / 15 / 15 %, seed 45.  MP3 decoding needs a registered codec (datasets/audio_io.py)."""
from pathlib import Path

import pandas as pd

from .base_dataset import SimpleAudioFakeDataset

_METHODS = ["faceswap-wav2lip", "fsgan-wav2lip", "wav2lip", "rtvc"]
FAKEAVCELEB_SPLIT = {"train": list(_METHODS), "test": list(_METHODS), "val": list(_METHODS),
                     "partition_ratio": [0.7, 0.15], "seed": 45}

_COLUMNS = ("user_id", "sample_name", "attack_type", "label", "path")


class FakeAVCelebDataset(SimpleAudioFakeDataset):
    audio_folder = "FakeAVCeleb-audio"
    audio_extension = ".mp3"
    metadata_file = Path(audio_folder) / "meta_data.csv"
    subsets = ("train", "dev", "eval")

    def __init__(self, path, subset="train", transform=None):
        super().__init__(subset, transform)
        self.path = path
        self.subset = subset
        self.allowed_attacks = FAKEAVCELEB_SPLIT[subset]
        self.partition_ratio = FAKEAVCELEB_SPLIT["partition_ratio"]
        self.seed = FAKEAVCELEB_SPLIT["seed"]
        self.metadata = self.get_metadata()
        self.samples = pd.concat([self.get_fake_samples(), self.get_real_samples()], ignore_index=True)

    def get_metadata(self):
        md = pd.read_csv(Path(self.path) / self.metadata_file)
        md["audio_type"] = md["type"].apply(lambda x: x.split("-")[-1])
        return md

    def get_file_path(self, sample):
        below_root = "/".join([self.audio_folder, *sample["path"].split("/")[1:]])
        return Path(self.path) / below_root / Path(sample["filename"]).with_suffix(self.audio_extension)

    def _rows(self, clips, attack_type=None, label="spoof"):
        return [(clip["source"], Path(clip["filename"]).stem, clip["method"] if attack_type is None else attack_type,
                 label, self.get_file_path(clip)) for _, clip in clips]

    def get_fake_samples(self):
        rows = []
        for method in self.allowed_attacks:
            clips = self.metadata[(self.metadata["method"] == method) & (self.metadata["audio_type"] == "FakeAudio")]
            # the reference shuffles the (index, row) pairs of iterrows() as a list: order by index, then seeded shuffle
            rows += self._rows(self.split_samples(list(clips.iterrows())))
        return pd.DataFrame(rows, columns=_COLUMNS)

    def get_real_samples(self):
        clips = self.metadata[(self.metadata["method"] == "real") & (self.metadata["audio_type"] == "RealAudio")]
        return pd.DataFrame(self._rows(self.split_samples(clips).iterrows(), attack_type="-", label="bonafide"),
                            columns=_COLUMNS)
