"""Shared test helpers: the surrogate detector of the golden fixtures and small generators."""
import numpy as np
import torch


class Surrogate(torch.nn.Module):
    """Same tiny (B, T) -> (B, 1) detector tests/golden/generate_golden.py attacked with the reference."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv1d(1, 4, kernel_size=9, stride=4)
        self.fc = torch.nn.Linear(4, 1)

    def forward(self, x):
        h = torch.tanh(self.conv(x.unsqueeze(1)) * 8.0)
        return self.fc(h.mean(dim=2)) * 4.0


def surrogate_from(fixture: dict) -> Surrogate:
    m = Surrogate()
    m.load_state_dict({k[len("model_"):]: torch.from_numpy(v) for k, v in fixture.items() if k.startswith("model_")})
    return m.eval()


def rand01(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.random(shape, dtype=np.float32)


def randn(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)
