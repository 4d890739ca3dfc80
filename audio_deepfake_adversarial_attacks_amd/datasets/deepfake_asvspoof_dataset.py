"""ASVspoof 2021 DF evaluation set (reference: src/datasets/deepfake_asvspoof_dataset.py).

Layout on disk: `<root>/keys/CM/trial_metadata.txt` (space-separated, utterance id in column 1, `bonafide`/`spoof` in
column 5) and the audio in `<root>/ASVspoof2021_DF_eval_part0{0..3}/ASVspoof2021_DF_eval/flac/<id>.flac`.
Spoofed and genuine trials are partitioned separately (70 / 15 / 15 %, seed 45), spoofed trials listed first.
FLAC decoding needs a registered codec (datasets/audio_io.py)."""
from pathlib import Path

import pandas as pd

from .base_dataset import SimpleAudioFakeDataset

DF_ASVSPOOF_SPLIT = {"partition_ratio": [0.7, 0.15], "seed": 45}


class DeepFakeASVSpoofDataset(SimpleAudioFakeDataset):
    protocol_file_name = "keys/CM/trial_metadata.txt"
    subset_dir_prefix = "ASVspoof2021_DF_eval"
    subset_parts = ("part00", "part01", "part02", "part03")

    def __init__(self, path, subset="train", transform=None):
        super().__init__(subset, transform)
        self.path = path
        self.partition_ratio = DF_ASVSPOOF_SPLIT["partition_ratio"]
        self.seed = DF_ASVSPOOF_SPLIT["seed"]
        self.flac_paths = self.get_file_references()
        self.samples = self.read_protocol()

    def get_file_references(self):
        """utterance id -> file, over the four archive parts."""
        found = {}
        for part in self.subset_parts:
            folder = Path(self.path) / f"{self.subset_dir_prefix}_{part}" / self.subset_dir_prefix / "flac"
            found.update((p.stem, p) for p in folder.glob("*.flac"))
        return found

    def read_protocol(self):
        by_label = {"bonafide": [], "spoof": []}
        with open(Path(self.path) / self.protocol_file_name, "r") as protocol:
            for line in protocol:
                label = line.strip().split(" ")[5]
                if label in by_label:
                    by_label[label].append(line)
        samples = {"sample_name": [], "label": [], "path": []}
        for label in ("spoof", "bonafide"):
            for line in self.split_samples(by_label[label]):
                self.add_line_to_samples(samples, line)
        return pd.DataFrame(samples)

    def add_line_to_samples(self, samples, line):
        _, sample_name, _, _, _, label, _, _ = line.strip().split(" ")
        sample_path = self.flac_paths[sample_name]
        assert sample_path.exists()
        samples["sample_name"].append(sample_name)
        samples["label"].append(label)
        samples["path"].append(sample_path)
        return samples
