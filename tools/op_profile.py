#!/usr/bin/env python
"""torch.profiler view of ONE PGD iteration's model work (LCNN + LFCC fwd + input-bwd, B = 128, fused paths on):
which ATen ops are still launched, how often, and their device time."""
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from audio_deepfake_adversarial_attacks_amd.models.models import get_model  # noqa: E402
from audio_deepfake_adversarial_attacks_amd.utils import set_seed  # noqa: E402

dev = torch.device("cuda:0")
set_seed(42)
MODEL = sys.argv[1] if len(sys.argv) > 1 else "lcnn"
CONFIGS = {"lcnn": ("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}),
           "specrnet": ("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2}),
           "rawnet3": ("rawnet3", {})}
model = get_model(*CONFIGS[MODEL], "cuda:0").to(dev)
model.train()
for m in model.modules():
    if "BatchNorm" in m.__class__.__name__ or "Dropout" in m.__class__.__name__:
        m.eval()
for p in model.parameters():
    p.requires_grad_(False)
x = torch.rand(64 if MODEL == "rawnet3" else 128, 64_600, device=dev)


def one_iter():
    adv = x.clone().requires_grad_(True)
    z = model(adv)
    (g,) = torch.autograd.grad(z, adv, grad_outputs=torch.ones_like(z))
    return g


for _ in range(3):
    one_iter()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(5):
        one_iter()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50,
                                                          max_shapes_column_width=60))
