"""The evaluation/training corpus: ASVspoof2021-DF + WaveFake + FakeAVCeleb, class-balanced
(reference: src/datasets/detection_dataset.py)."""
import logging
from typing import List, Optional

import pandas as pd

from .base_dataset import SimpleAudioFakeDataset
from .deepfake_asvspoof_dataset import DeepFakeASVSpoofDataset
from .fakeavceleb_dataset import FakeAVCelebDataset
from .wavefake_dataset import WaveFakeDataset

LOGGER = logging.getLogger()


class DetectionDataset(SimpleAudioFakeDataset):
    def __init__(self, asvspoof_path=None, wavefake_path=None, fakeavceleb_path=None, subset: str = "val",
                 transform=None, oversample: bool = True, undersample: bool = False, return_label: bool = True,
                 reduced_number: Optional[int] = None, return_meta: bool = False, return_raw: bool = False,
                 device_pad: bool = False, wave_fake_trim: Optional[bool] = None):
        super().__init__(subset=subset, transform=transform, return_label=return_label, return_meta=return_meta,
                         return_raw=return_raw, device_pad=device_pad, wave_fake_trim=wave_fake_trim)
        parts = self._init_datasets(asvspoof_path=asvspoof_path, wavefake_path=wavefake_path,
                                    fakeavceleb_path=fakeavceleb_path, subset=subset)
        self.samples = pd.concat([ds.samples for ds in parts], ignore_index=True)
        if oversample:
            self.oversample_dataset()
        elif undersample:
            self.undersample_dataset()
        if reduced_number:
            LOGGER.info(f"Using reduced number of samples - {reduced_number}!")
            self.samples = self.samples.sample(min(len(self.samples), reduced_number), random_state=42)

    def _init_datasets(self, asvspoof_path: Optional[str], wavefake_path: Optional[str],
                       fakeavceleb_path: Optional[str], subset: str) -> List[SimpleAudioFakeDataset]:
        corpora = ((asvspoof_path, DeepFakeASVSpoofDataset), (wavefake_path, WaveFakeDataset),
                   (fakeavceleb_path, FakeAVCelebDataset))
        return [cls(path, subset=subset) for path, cls in corpora if path is not None]

    def _of_label(self, label):
        # same rows, in the same order, as the reference's groupby(["label"]).get_group(label)
        return self.samples[self.samples["label"] == label]

    def oversample_dataset(self):
        """Draw (with replacement, from numpy's global RNG like the reference) as many extra bonafide rows as there
        are more spoof than bonafide rows."""
        bonafide, spoof = self._of_label("bonafide"), self._of_label("spoof")
        if len(bonafide) == 0 or len(spoof) == 0:
            raise KeyError("oversampling needs both bonafide and spoof samples")
        extra = len(spoof) - len(bonafide)
        if extra < 0:
            raise NotImplementedError
        if extra > 0:
            self.samples = pd.concat([self.samples, bonafide.sample(extra, replace=True)], ignore_index=True)

    def undersample_dataset(self):
        bonafide, spoof = self._of_label("bonafide"), self._of_label("spoof")
        if len(bonafide) == 0 or len(spoof) == 0:
            raise KeyError("undersampling needs both bonafide and spoof samples")
        if len(spoof) < len(bonafide):
            raise NotImplementedError
        if len(spoof) > len(bonafide):
            self.samples = pd.concat([bonafide, spoof.sample(len(bonafide), replace=True)], ignore_index=True)

    def get_bonafide_only(self):
        self.samples = self._of_label("bonafide")
        return self.samples

    def get_spoof_only(self):
        self.samples = self._of_label("spoof")
        return self.samples
