"""SURVEY.md section 8-f4 on the MI355X: the waveform-batch kernels (include/advstep_dataset.h) through the C ABI against
oracle/dataset.py and the reference-generated fixtures — bit-exact (index work + a power-of-two scale)."""
import hashlib
import json

import numpy as np
import pytest
import torch

from audio_deepfake_adversarial_attacks_amd import _lib
from audio_deepfake_adversarial_attacks_amd.aa.qualitative.attacks_analysis import AttackAnalyser
from audio_deepfake_adversarial_attacks_amd.datasets import base_dataset, wave_ops
from audio_deepfake_adversarial_attacks_amd.datasets.detection_dataset import DetectionDataset
from oracle import dataset as OD
from tests import helpers
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ragged(lengths, channels, dtype, seed):
    rng = np.random.default_rng(seed)
    waves = []
    for n, ch in zip(lengths, channels):
        shape = (n, ch)
        waves.append(rng.integers(-32768, 32767, shape, endpoint=True).astype(np.int16) if dtype == np.int16
                     else (rng.standard_normal(shape) * 0.2).astype(np.float32))
    return waves


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
@pytest.mark.parametrize("cut", [64, 4099, 64_600])
def test_pad_tile_ragged_bit_exact(cuda, dtype, cut):
    lengths = [1, 2, 7, 63, 64, 65, 1000, 4099, 32_300, 64_599, 64_600, 70_001]
    channels = [1, 2, 1, 3, 1, 1, 2, 1, 1, 2, 1, 1]
    ragged = wave_ops.RaggedWaveBatch.from_arrays(_ragged(lengths, channels, dtype, 11))
    got = ragged.to_padded(cuda, cut)
    want = OD.pad_tile_batch(ragged.payload.numpy(), ragged.offsets.tolist(), ragged.lengths.tolist(),
                             ragged.channels.tolist(), cut)
    assert got.shape == (len(lengths), cut) and got.dtype == torch.float32
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_pad_tile_mono_fast_path_and_reference_fixture(cuda, golden):
    g = golden("datasets")
    for k, (n, cut) in enumerate(g["pad_cases"]):
        x = torch.from_numpy(g[f"pad_in_{k}"]).to(cuda)
        got = wave_ops.apply_pad_batch(x, int(cut))
        np.testing.assert_array_equal(got[0].cpu().numpy(), g[f"pad_out_{k}"], err_msg=f"length {n}, cut {cut}")


def test_pad_tile_zero_length_row_and_empty_batch(cuda):
    payload = torch.arange(10, dtype=torch.float32, device=cuda)
    offsets = torch.tensor([0, 3], device=cuda)
    lengths = torch.tensor([3, 0], device=cuda)
    got = wave_ops.pad_tile(payload, offsets, lengths, 8)
    np.testing.assert_array_equal(got.cpu().numpy(), np.array([[0, 1, 2, 0, 1, 2, 0, 1], [0] * 8], np.float32))
    empty = wave_ops.pad_tile(payload, offsets[:0], lengths[:0], 8)
    assert empty.shape == (0, 8)
    with pytest.raises(ZeroDivisionError):
        wave_ops.apply_pad_batch(torch.empty(2, 0, device=cuda), 8)


def test_pad_tile_full_size_properties(cuda):
    """BASELINE size (B = 128 utterances cut to 64 600): periodicity with each row's own period, identity for rows that
    are long enough."""
    rng = np.random.default_rng(5)
    lengths = rng.integers(8_000, 90_000, 128)
    ragged = wave_ops.RaggedWaveBatch.from_arrays(_ragged(lengths, [1] * 128, np.int16, 6))
    out = ragged.to_padded(cuda, 64_600)
    payload = ragged.payload.to(cuda).float() / 32768.0
    for b in (0, 17, 127):
        n, off = int(lengths[b]), int(ragged.offsets[b])
        head = min(n, 64_600)
        assert torch.equal(out[b, :head], payload[off:off + head])
        if n < 64_600:
            assert torch.equal(out[b, n:], out[b, :64_600 - n])
    assert float(out.abs().max()) <= 1.0


def test_preprocessing_on_batch_stays_on_device_and_matches_reference(cuda, golden):
    g = golden("datasets")
    x = torch.from_numpy(g["onbatch_in"]).to(cuda)
    rates = torch.full((x.shape[0],), 16_000, dtype=torch.int64)
    for name, cut in (("short", 2600), ("long", 640)):
        got, got_rates = base_dataset.SimpleAudioFakeDataset.wavefake_preprocessing_on_batch(
            x, rates, wave_fake_trim=False, wave_fake_cut=cut)
        assert got.is_cuda and not got_rates.is_cuda
        np.testing.assert_array_equal(got.cpu().numpy(), g[f"onbatch_{name}_out"])
        np.testing.assert_array_equal(got_rates.numpy(), g[f"onbatch_{name}_rates"])
    with pytest.raises(base_dataset.SoxUnavailableError):
        base_dataset.SimpleAudioFakeDataset.wavefake_preprocessing_on_batch(x, rates)  # default: SoX trim


def test_device_pad_dataset_equals_reference_item_contract(cuda, tmp_path):
    roots = helpers.build_corpus_trees(tmp_path)
    kw = dict(wavefake_path=str(roots["wavefake_path"]), subset="train", wave_fake_trim=False)
    plain = DetectionDataset(**kw)
    raw = DetectionDataset(device_pad=True, **kw)
    idx = list(range(0, len(plain), 3))[:12]
    ragged, rates, labels = base_dataset.ragged_collate([raw[i] for i in idx])
    got = ragged.pin_memory().to_padded(cuda, base_dataset.WAVE_FAKE_CUT)
    want = torch.stack([plain[i][0] for i in idx])
    assert torch.equal(got.cpu(), want)
    assert labels.tolist() == [plain[i][2] for i in idx]


@pytest.mark.parametrize("B", [0, 1, 63, 64, 65, 128, 1000])
def test_qual_select_bit_exact(cuda, B):
    rng = np.random.default_rng(B)
    y = rng.integers(0, 2, B)
    clean = rng.integers(0, 2, B).astype(np.int32)
    attacked = rng.integers(0, 2, B).astype(np.int32)
    rows, counts = wave_ops.qual_select(torch.from_numpy(y).to(cuda), torch.from_numpy(clean).to(cuda),
                                        torch.from_numpy(attacked).to(cuda))
    fp, fn = OD.qual_select(y, clean, attacked)
    n_fp, n_fn = counts.cpu().tolist()
    assert (n_fp, n_fn) == (len(fp), len(fn))
    rows = rows.cpu().numpy()
    np.testing.assert_array_equal(rows[:n_fp], fp)
    np.testing.assert_array_equal(rows[n_fp:n_fp + n_fn], fn)


@pytest.mark.parametrize("T", [800, 4099, 64_600])
def test_gather_rows_bit_exact(cuda, T):
    x = torch.randn(37, T, device=cuda)
    rows = torch.tensor([36, 0, 5, 5, 17], dtype=torch.int32, device=cuda)
    assert torch.equal(wave_ops.gather_rows(x, rows), x[rows.long()])
    assert torch.equal(wave_ops.gather_rows(x, rows, 2), x[rows[:2].long()])
    assert wave_ops.gather_rows(x, rows, 0).shape == (0, T)


def test_attack_analyser_writes_the_reference_files(cuda, golden, tmp_path, capsys):
    g = golden("datasets")
    listing = json.loads((GOLDEN / "datasets_listing.json").read_text())
    B = len(g["qual_y"])
    metadata = [["melgan"] * B, listing["qual_paths"], ["val"] * B, torch.from_numpy(g["qual_seconds"])]
    to = lambda a: torch.from_numpy(a).to(cuda)  # noqa: E731
    AttackAnalyser(tmp_path / "out").analyse(
        batch_x=to(g["qual_x"]), batch_x_attacked=to(g["qual_xa"]), batch_y=to(g["qual_y"]),
        batch_preds_label=to(g["qual_attacked"]), batch_preds=torch.rand(B, device=cuda),
        batch_preds_noattack_label=to(g["qual_clean"]), batch_preds_noattack=torch.rand(B, device=cuda),
        batch_metadata=metadata)
    files = {p.name: p.read_bytes() for p in (tmp_path / "out").iterdir()}
    assert sorted(files) == sorted(listing["qual_files"])
    for name, blob in files.items():
        assert hashlib.sha256(blob).hexdigest() == listing["qual_files"][name], name
    assert len(capsys.readouterr().out.strip().splitlines()) == B  # one diff line per utterance


def test_ops_refuse_cpu_tensors():
    with pytest.raises(_lib.AdvstepError, match="no CPU fallback"):
        wave_ops.apply_pad_batch(torch.zeros(2, 8), 16)
    with pytest.raises(_lib.AdvstepError, match="no CPU fallback"):
        wave_ops.qual_select(torch.zeros(2, dtype=torch.int64), torch.zeros(2, dtype=torch.int32),
                             torch.zeros(2, dtype=torch.int32))


def test_real_wav_corpus_end_to_end(cuda, tmp_path, capsys):
    """generate_attacks over a WaveFake-layout corpus on disk: the device-pad loader (payload upload + one kernel) and the
    reference-style loader (padded float items) give the same report; --qual writes a WAV pair per flipped utterance;
    --raw_from_dataset re-runs the (SoX-free) preprocessing on the device after the attack."""
    import yaml
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    from tests.conftest import ROOT
    roots = helpers.build_corpus_trees(tmp_path / "data")
    cfg = yaml.safe_load((ROOT / "configs" / "aa_evaluation" / "lcnn.yaml").read_text())
    paths = [None, str(roots["wavefake_path"]), None]
    cls, params = AttackEnum.FGSM_eps001.value
    reports = {}
    for name, kw in (("device_pad", dict(device_pad=True, num_workers=2)), ("host_pad", dict(device_pad=False)),
                     ("raw", dict(device_pad=True, raw_sample_from_dataset=True))):
        set_seed(42)
        np.random.seed(42)
        analyser = AttackAnalyser(tmp_path / f"qual_{name}")
        reports[name] = generate_attacks(paths, cfg, str(cuda), attack_model_config=cfg, attack_method=cls,
                                         attack_params=params, batch_size=4, share_weights=True,
                                         wave_fake_trim=False, on_attack_end_callback=analyser.analyse, **kw)
    capsys.readouterr()
    assert reports["device_pad"]["num_total"] == 12  # the validation part, class-balanced: three full batches of 4
    for key, value in reports["device_pad"].items():
        assert reports["host_pad"][key] == value and reports["raw"][key] == value, key
    written = sorted(p.name for p in (tmp_path / "qual_device_pad").iterdir())
    assert written == sorted(p.name for p in (tmp_path / "qual_host_pad").iterdir())
    assert len(written) % 2 == 0 and all(n.endswith(("_original.wav", "_attacked.wav")) for n in written)
    for n in written:  # same bytes from both loaders
        assert (tmp_path / "qual_device_pad" / n).read_bytes() == (tmp_path / "qual_host_pad" / n).read_bytes()
        wave, rate = audio_io_load(tmp_path / "qual_device_pad" / n)
        assert rate == 16_000 and wave.shape == (1, 64_600)


def audio_io_load(path):
    from audio_deepfake_adversarial_attacks_amd.datasets import audio_io
    return audio_io.load(path)


def test_training_cli_on_a_wav_corpus(cuda, tmp_path):
    """The second caller of the attack API fed from audio files: train / test parts of a WaveFake-layout corpus, payloads
    decoded in DataLoader workers, padded on the device."""
    import yaml

    import train_models_on_adversarial_attacks as cli
    roots = helpers.build_corpus_trees(tmp_path / "data")
    with open("configs/aa_training/finetune/lcnn_fgsm.yaml") as f:
        cfg = yaml.safe_load(f)
    cfg["data"]["adversarial_attacks"] = ["FGSM_eps001"]
    cfg["checkpoint"]["path"] = ""
    (tmp_path / "one.yaml").write_text(yaml.dump(cfg))
    args = cli.parse_args(["--config", str(tmp_path / "one.yaml"), "--wavefake_path", str(roots["wavefake_path"]),
                           "--no_trim", "--batch_size", "8", "--epochs", "1", "--ckpt", str(tmp_path / "ckpt"),
                           "--config_save_path", str(tmp_path), "--adv_training_strategy", "EQUAL"])
    model = cli.main(args)
    saved = list((tmp_path / "ckpt").glob("aad__lcnn_*/ckpt*.pth"))
    assert len(saved) == 2
    state = torch.load(saved[0], map_location="cpu")
    assert all(torch.isfinite(v).all() for v in state.values() if v.dtype.is_floating_point)
