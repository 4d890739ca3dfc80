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


class FixedEncoder(torch.nn.Module):
    """Stand-in for RawNet3's sinc encoder (`conv1`, third-party and parity-unpinned): returns the tensor it was
    given, so everything the reference computes AFTER conv1 can be pinned."""

    def __init__(self, *_, **__):
        super().__init__()
        self.h = None

    def forward(self, x):
        return self.h


RAWNET3_SEED = 81


def rawnet3_fixture_weights(make_model):
    """The full-size RawNet3 (15.5 M parameters: too large to store) of tests/golden/rawnet3_body.npz, rebuilt from a
    seed: default initialisation under torch.manual_seed, then non-trivial BatchNorm statistics / affine terms and AFMS
    alphas drawn from a dedicated generator in sorted key order.  The fixture stores a SHA-256 per tensor (taken from the
    REFERENCE class built with this recipe), which the tests check before comparing outputs."""
    torch.manual_seed(RAWNET3_SEED)
    model = make_model()
    g = torch.Generator().manual_seed(RAWNET3_SEED + 1)
    with torch.no_grad():
        for key, value in sorted(model.state_dict().items()):
            if key.startswith("conv1."):
                continue
            if key.endswith("running_mean"):
                value.copy_(torch.empty(value.shape).uniform_(-0.2, 0.2, generator=g))
            elif key.endswith("running_var"):
                value.copy_(torch.empty(value.shape).uniform_(0.5, 1.5, generator=g))
            elif ".bn" in key or key.startswith("bn") or key.startswith("attention.2."):
                if key.endswith("weight"):
                    value.copy_(torch.empty(value.shape).uniform_(0.8, 1.2, generator=g))
                elif key.endswith("bias"):
                    value.copy_(torch.empty(value.shape).uniform_(-0.1, 0.1, generator=g))
            elif key.endswith("afms.alpha"):
                value.copy_(torch.empty(value.shape).uniform_(0.7, 1.3, generator=g))
    return model


def tensor_digests(state_dict, skip=("conv1.",)):
    import hashlib
    return {k: hashlib.sha256(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes()).hexdigest()
            for k, v in state_dict.items() if not k.startswith(skip)}


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


# ---------------------------------------------------------------------------------------------------------
# a miniature copy of the three corpora's directory layouts
# ---------------------------------------------------------------------------------------------------------

def _write_wav(path, data, rate):
    """Minimal PCM16 / float32 WAVE writer for the test corpora (independent of the product's reader/writer)."""
    import struct
    data = np.ascontiguousarray(data)
    channels = 1 if data.ndim == 1 else data.shape[1]
    code, width = (1, 2) if data.dtype == np.int16 else (3, 4)
    fmt = struct.pack("<HHIIHH", code, channels, rate, rate * width * channels, width * channels, 8 * width)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", data.nbytes) + data.tobytes()
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def corpus_waveform(index: int):
    """Deterministic content of the index-th WaveFake wav file: (samples, rate); every third file is stereo, every
    fourth float32, lengths 150..2600 frames."""
    rng = np.random.default_rng(1000 + index)
    frames = 150 + (index * 397) % 2451
    shape = (frames, 2) if index % 3 == 2 else (frames,)
    if index % 4 == 3:
        return (rng.standard_normal(shape) * 0.1).astype(np.float32), 16_000
    return rng.integers(-20000, 20000, shape).astype(np.int16), 16_000


def build_corpus_trees(root):
    """Create <root>/ASVspoof2021/DF, <root>/WaveFake, <root>/FakeAVCeleb_v1.2 with the layouts the dataset classes
    parse.  FLAC / MP3 files are empty placeholders (listing needs existence only); WaveFake wav files are real."""
    from pathlib import Path
    root = Path(root)
    asv = root / "ASVspoof2021" / "DF"
    (asv / "keys" / "CM").mkdir(parents=True)
    lines = []
    for i in range(46):
        name = f"DF_E_{2000011 + 37 * i}"
        label = "bonafide" if i % 4 == 1 else "spoof"
        part = f"part0{i % 4}"
        folder = asv / f"ASVspoof2021_DF_eval_{part}" / "ASVspoof2021_DF_eval" / "flac"
        folder.mkdir(parents=True, exist_ok=True)
        (folder / f"{name}.flac").touch()
        attack = "-" if label == "bonafide" else f"A{7 + i % 13:02d}"
        lines.append(f"LA_{i:04d} {name} nocodec asvspoof {attack} {label} notrim eval")
    (asv / "keys" / "CM" / "trial_metadata.txt").write_text("\n".join(lines) + "\n")

    wf = root / "WaveFake"
    index = 0
    folders = ["ljspeech_melgan", "ljspeech_hifiGAN", "jsut_multi_band_melgan", "ljspeech_full_band_melgan",
               "ljspeech_unknownvoc", "common_voices_prompts_from_conformer_fastspeech2_pwg_ljspeech"]
    for folder in folders:
        d = wf / "generated_audio" / folder
        d.mkdir(parents=True)
        for k in range(9):
            stem = f"BASIC5000_{k + 1:04d}" if folder.startswith("jsut") else f"LJ{1 + k // 5:03d}-{1 + k % 5:04d}"
            _write_wav(d / f"{stem}_gen.wav", *corpus_waveform(index))
            index += 1
    for folder, stems in (("real_audio/jsut_ver1.1/basic5000/wav", [f"BASIC5000_{k + 1:04d}" for k in range(11)]),
                          ("real_audio/LJSpeech-1.1/wavs", [f"LJ{1 + k // 5:03d}-{1 + k % 5:04d}" for k in range(13)])):
        d = wf / folder
        d.mkdir(parents=True)
        for stem in stems:
            _write_wav(d / f"{stem}.wav", *corpus_waveform(index))
            index += 1

    celeb = root / "FakeAVCeleb_v1.2"
    audio = celeb / "FakeAVCeleb-audio"
    audio.mkdir(parents=True)
    rows = ["source,target1,target2,method,category,type,race,gender,filename,path"]
    methods = ["real", "rtvc", "wav2lip", "faceswap-wav2lip", "fsgan-wav2lip", "faceswap"]
    for i in range(60):
        method = methods[i % len(methods)]
        kind = "RealVideo-RealAudio" if method == "real" else (
            "RealVideo-FakeAudio" if method == "rtvc" else ("FakeVideo-FakeAudio" if i % 2 else "FakeVideo-RealAudio"))
        source = f"id{100 + i // 3:05d}"
        rel = f"FakeAVCeleb/{kind}/African/men/{source}"
        filename = f"{i:05d}_{method}.mp4"
        rows.append(f"{source},-,-,{method},A,{kind},African,men,{filename},{rel}")
        d = audio / kind / "African" / "men" / source
        d.mkdir(parents=True, exist_ok=True)
        (d / filename).with_suffix(".mp3").touch()
    (audio / "meta_data.csv").write_text("\n".join(rows) + "\n")
    return {"asvspoof_path": asv, "wavefake_path": wf, "fakeavceleb_path": celeb}


def listing_of(samples, root):
    """A dataset's `samples` frame as JSON-able rows with paths relative to the corpus root."""
    from pathlib import Path
    rows = []
    for _, r in samples.iterrows():
        attack = r["attack_type"] if "attack_type" in r else None
        rows.append([str(Path(r["path"]).relative_to(root)), r["label"],
                     attack if isinstance(attack, str) else None, r["sample_name"]])
    return rows
