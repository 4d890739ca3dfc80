"""Adversarial-training strategy registry: CLI name -> trainer class (reference: src/aa/aa_trainer_types.py:12-17)."""
from enum import Enum

from .. import trainer


class AdversarialGDTrainerEnum(Enum):
    ONLY_ADV = trainer.OnlyOneAdversarialGDTrainer
    RANDOM = trainer.AdversarialGDTrainer
    ADAPTIVE = trainer.AdaptiveAdversarialGDTrainer
    ADAPTIVE_V2 = trainer.AdaptiveV2AdversarialGDTrainer
    EQUAL = trainer.EqualAdversarialGDTrainer
