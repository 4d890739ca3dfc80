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


def fab_projection_inputs(T: int, seed: int):
    """The seeded (t, w, b) rows tests/golden/generate_golden.py fed to the reference's FAB projections (the fixture
    stores only outputs + input checksums): exact 0s / 1s in the points, exact zeros in the normals, hyperplanes from
    very close to out of the box's reach."""
    g = torch.Generator().manual_seed(seed)
    R = 6
    t = torch.rand(R, T, generator=g)
    t[:, ::7] = 0.0
    t[:, 3::11] = 1.0
    w = torch.randn(R, T, generator=g) * 0.01
    w[:, ::13] = 0.0
    dot = (w * t).sum(1)
    b = dot + torch.tensor([1e-4, -1e-3, 0.05, -0.2, 0.45, 5.0]) * w.abs().sum(1)
    return t, w, b


class TinyDetectionSet(torch.utils.data.Dataset):
    """The in-memory (waveform, sample_rate, label) set tests/golden/generate_golden.py trained the reference's
    trainers on: waveforms N(0, 0.05^2) clipped to [-1, 1] from torch.Generator(seed), labels from seed + 1."""

    def __init__(self, n: int, T: int, seed: int):
        g = torch.Generator().manual_seed(seed)
        self.x = (torch.randn(n, T, generator=g) * 0.05).clamp_(-1.0, 1.0)
        self.y = torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(seed + 1))

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], 16_000, int(self.y[i])
