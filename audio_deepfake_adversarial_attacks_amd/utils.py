"""Harness helpers of the evaluation loop (reference: src/utils.py:33-71)."""
import logging
import os
import random

import numpy as np
import torch
from torch import nn

from .models import models

LOGGER = logging.getLogger(__name__)


def set_seed(seed: int):
    """Fix the Python / numpy / torch PRNG seeds (src/utils.py:33-44)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    os.environ["PYTHONHASHSEED"] = str(seed)


def load_model(model_config, device: str = "cuda"):
    """Build the model named by a YAML config and optionally load its checkpoint (src/utils.py:47-71).

    Checkpoints saved from an `nn.DataParallel` wrapper carry a `module.` prefix; like the reference, a failed
    strict load is retried through a DataParallel wrapper and then unwrapped."""
    model_name, model_parameters = model_config["model"]["name"], model_config["model"]["parameters"]
    model_path = model_config["checkpoint"].get("path", "")

    model = models.get_model(model_name=model_name, config=model_parameters, device=device)
    if model_path:
        state = torch.load(model_path, map_location="cpu")
        try:
            model.load_state_dict(state)
        except RuntimeError:
            wrapped = nn.DataParallel(model)
            wrapped.load_state_dict(state)
            model = wrapped.module
        LOGGER.info("Loaded weigths on '%s' model, path: %s", model_name, model_path)
    model = model.to(device)
    model.weights_path = model_path
    return model


def find_wav_files(path_to_dir):
    """All *.wav files below a directory, sorted; None when there are none (src/utils.py:18-30)."""
    from pathlib import Path
    paths = sorted(Path(path_to_dir).glob("**/*.wav"))
    return paths or None
