"""`-m gpu`: the fused LFCC tail (include/advstep_frontend.h) against this repository's torch restatement of
torchaudio's LFCC (frontends.LFCC with the kernels switched off), values and waveform gradients, on the GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def lfcc(cuda):
    from audio_deepfake_adversarial_attacks_amd import frontends
    return frontends.LFCC().to(cuda)


def run(lfcc, x, gy, fused, monkeypatch):
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "1" if fused else "0")
    a = x.clone().requires_grad_(True)
    y = lfcc(a)
    (g,) = torch.autograd.grad(y, a, gy)
    return y.detach(), g


@pytest.mark.parametrize("inlds_fft", ["1", "0"])
@pytest.mark.parametrize("B,T", [(3, 64_600), (2, 8_000), (1, 1_000), (5, 16_160)])
def test_fused_lfcc_matches_torch_chain(lfcc, cuda, monkeypatch, B, T, inlds_fft):
    """inlds_fft = 1: framing + FFT + filterbank (and their backward) inside one kernel each (csrc/lfcc_stft.hip);
    0: framing kernel + hipFFT + filterbank kernel."""
    monkeypatch.setenv("ADVSTEP_INLDS_FFT", inlds_fft)
    gen = torch.Generator().manual_seed(B * 7 + T)
    x = torch.rand(B, T, generator=gen).to(cuda)
    y0, _ = run(lfcc, x, None, False, monkeypatch) if False else (None, None)
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "0")
    ref = lfcc(x)
    gy = torch.randn(ref.shape, generator=gen).to(cuda)
    y_ref, g_ref = run(lfcc, x, gy, False, monkeypatch)
    y, g = run(lfcc, x, gy, True, monkeypatch)
    assert y.shape == y_ref.shape == (B, 80, T // 160 + 1)
    # dB-scale values of magnitude ~1e2: 1e-3 absolute is 1e-5 relative (float rounding of log10 / summation order)
    assert (y - y_ref).abs().max().item() <= 2e-3
    rel = (g - g_ref).norm().item() / g_ref.norm().item()
    assert rel <= 1e-4, rel
    # the fused output is frame-major: LCNN's permute(0, 1, 3, 2) of it is contiguous
    assert y.unsqueeze(1).permute(0, 1, 3, 2).is_contiguous()
    # fixed summation order everywhere (the overlap-add's border atomics have two operands): bit-reproducible
    y2, g2 = run(lfcc, x, gy, True, monkeypatch)
    assert torch.equal(y, y2) and torch.equal(g, g2)


def test_fused_lfcc_floor_and_amax_gradient_path(lfcc, cuda, monkeypatch):
    """A batch with a silent utterance: its bands are floored at (batch max - 80 dB); torchaudio routes the floored
    gradients to the batch maximum through amax.  Fused and plain paths must agree on values and gradients."""
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(3, 16_000, generator=gen)
    x[1] = x[1] * 1e-7            # ~ -140 dB relative: fully floored
    x[2, 4_000:12_000] = 0.0      # partly silent
    x = x.to(cuda)
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "0")
    gy = torch.randn(lfcc(x).shape, generator=gen).to(cuda)
    y_ref, g_ref = run(lfcc, x, gy, False, monkeypatch)
    y, g = run(lfcc, x, gy, True, monkeypatch)
    assert (y - y_ref).abs().max().item() <= 2e-3
    assert (g - g_ref).norm().item() / g_ref.norm().item() <= 1e-3
    # the silent utterance sits on the floor (constant bands -> only the 0-th cepstral coefficient is non-zero)
    assert y[1, 1:].abs().max().item() <= 1e-2 and y[1, 0].std().item() <= 1e-3


def test_two_launch_lfcc_equals_the_separate_entry_points(lfcc, cuda, monkeypatch):
    """Round 4: forward = stft_bands -> max_project (the block-maximum reduction inside the projection), backward =
    project_backward_zero (tie count + zero fill of dx) -> stft_bands_backward_fixup (floor fix-up inside the band-gradient
    load).  Against the separate C-ABI calls they replace (reduce_max, project, project_backward, floor_fixup, memset +
    stft_bands_backward) on a batch whose floor is active: same cepstra and waveform gradient up to the summation order of the
    DCT (the plain calls are the vector-ALU kernels); a second backward pass through the same forward (retain_graph) gives the
    same gradient again."""
    from audio_deepfake_adversarial_attacks_amd import _lib, frontend_ops
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "1")
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(4, 32_000, generator=gen)
    x[1] = x[1] * 1e-7
    x[3, 9_000:20_000] = 0.0
    x = x.to(cuda)
    a = x.clone().requires_grad_(True)
    y = lfcc(a)
    gy = torch.randn(y.shape, generator=gen).to(cuda)
    (g1,) = torch.autograd.grad(y, a, gy, retain_graph=True)
    (g2,) = torch.autograd.grad(y, a, gy, retain_graph=True)
    (g3,) = torch.autograd.grad(y, a, gy, retain_graph=True)       # a THIRD pass (the counters are reset in place)
    (g4,) = torch.autograd.grad(y, a, gy)
    # (equal up to the rounding of ONE scalar: the floored gradients' sum is accumulated with float atomics over workgroups)
    for g in (g2, g3, g4):
        assert (g1 - g).norm().item() <= 1e-6 * g1.norm().item()
    # the fragment table the forward pass used is the one the backward pass gets, whatever the cache says meanwhile
    y2 = lfcc(a)
    frontend_ops._FRAGMENTS._rows.clear()                           # a cache miss between the two passes
    (g5,) = torch.autograd.grad(y2, a, gy)
    assert (g1 - g5).norm().item() <= 1e-6 * g1.norm().item()

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B, T = x.shape
    hop, nfft, NF = 160, 512, 1 + T // 160
    tables, dct, window = lfcc._tables(), lfcc.dct_mat, lfcc._window_nfft()
    M, K = dct.shape
    band = torch.empty(B, NF, M, device=cuda)
    nblk = lib.advstep_stft_bands_block_count(B, NF)
    bmax = torch.full((nblk,), float("nan"), device=cuda)          # garbage the band kernel must overwrite or fill
    stats = torch.full((4,), 7.0, device=cuda)
    out = torch.empty(B, NF, K, device=cuda)
    _lib.check(lib.advstep_stft_bands_f32(x.data_ptr(), window.data_ptr(), tables.fb_start.data_ptr(), tables.fb_w.data_ptr(),
                                          tables.span, band.data_ptr(), bmax.data_ptr(), B, T, NF, hop, nfft, M, st), "bands")
    _lib.check(lib.advstep_lfcc_reduce_max_f32(bmax.data_ptr(), nblk, stats.data_ptr(), st), "max")
    _lib.check(lib.advstep_lfcc_project_f32(band.data_ptr(), dct.data_ptr(), stats.data_ptr(), 80.0, out.data_ptr(), B, M, NF, K,
                                            st), "project")
    # (the plain entry points are the vector-ALU kernels since round 4: another summation order than the matrix-core pair)
    assert (out.transpose(1, 2) - y.detach()).abs().max().item() <= 2e-4
    go = gy.transpose(1, 2).contiguous()
    dband = torch.empty(B, NF, M, device=cuda)
    dx = torch.full((B, T), float("nan"), device=cuda)
    _lib.check(lib.advstep_lfcc_project_backward_f32(go.data_ptr(), dct.data_ptr(), band.data_ptr(), stats.data_ptr(), 80.0,
                                                     dband.data_ptr(), B, M, NF, K, st), "project_backward")
    assert stats[2].item() != 0.0 and stats[1].item() >= 1.0       # the floor is active in this batch
    _lib.check(lib.advstep_lfcc_floor_fixup_f32(band.data_ptr(), stats.data_ptr(), dband.data_ptr(), band.numel(), st), "fixup")
    _lib.check(lib.advstep_stft_bands_backward_f32(x.data_ptr(), window.data_ptr(), dband.data_ptr(), tables.fbt_start.data_ptr(),
                                                   tables.fbt_w.data_ptr(), tables.span_t, dx.data_ptr(), B, T, NF, hop, nfft, M,
                                                   st), "bands_backward")
    rel = (dx - g1).norm().item() / g1.norm().item()
    assert rel <= 2e-6, rel


def test_lcnn_forward_uses_fused_frontend_without_copy(cuda, monkeypatch):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda).eval()
    x = torch.rand(4, 64_600, device=cuda)
    with torch.no_grad():
        monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "0")
        z0 = model(x)
        monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "1")
        z1 = model(x)
    assert (z0 - z1).abs().max().item() <= 1e-4 * max(z0.abs().max().item(), 1.0)


# ---- mel-spec frontend (SpecRNet's input): fused kernels vs the torch op chain ---------------------------------------------

@pytest.mark.parametrize("B,T", [(3, 64_600), (2, 8_000), (1, 1_000)])
def test_fused_mel_spec_matches_torch_chain(cuda, monkeypatch, B, T):
    """(B, T) -> (B, 2, 80, frames): magnitude to 1e-5 of its scale; phase compared as a unit vector (it wraps at +-pi)
    where the magnitude is not at rounding level; waveform gradient to 1e-4 relative; bit-reproducible."""
    from audio_deepfake_adversarial_attacks_amd.frontends import MelSpecFrontend
    fe = MelSpecFrontend().to(cuda)
    gen = torch.Generator().manual_seed(B * 11 + T)
    x = (torch.rand(B, T, generator=gen) - 0.5).to(cuda)

    def run(fused, gy=None):
        monkeypatch.setenv("ADVSTEP_FUSED_MEL", "1" if fused else "0")
        a = x.clone().requires_grad_(True)
        y = fe(a)
        if gy is None:
            return y.detach(), None
        (g,) = torch.autograd.grad(y, a, gy)
        return y.detach(), g

    y_ref, _ = run(False)
    assert y_ref.shape == (B, 2, 80, T // 160 + 1)
    # a cotangent that stays away from the phase's singular points: weight the phase plane by the squared magnitude
    gy = torch.randn(y_ref.shape, generator=gen).to(cuda)
    gy[:, 1] *= (y_ref[:, 0] ** 2).clamp(max=1.0)
    y_ref, g_ref = run(False, gy)
    y, g = run(True, gy)
    scale = y_ref[:, 0].abs().max().item()
    assert (y[:, 0] - y_ref[:, 0]).abs().max().item() <= 1e-5 * scale
    big = y_ref[:, 0] > 1e-3 * scale
    dphi = torch.remainder(y[:, 1] - y_ref[:, 1] + torch.pi, 2 * torch.pi) - torch.pi
    assert dphi[big].abs().max().item() <= 1e-3
    assert (g - g_ref).norm().item() / g_ref.norm().item() <= 1e-4
    y2, g2 = run(True, gy)
    assert torch.equal(y, y2) and torch.equal(g, g2)


def test_mel_backward_from_output_equals_backward_from_waveform(cuda, monkeypatch):
    """The shipped mel-spec gradient reconstructs d Y from the forward OUTPUT (|Y|, phase) and skips the spectrum; the first
    version recomputed framing + FFT + band projection from the waveform.  With the radix-4 kernels on both sides (the
    recomputation runs the forward's own transform, bit for bit) the two agree to 1e-5 relative.  The shipped pair (round 4:
    register-resident transform) is held against a float64 evaluation of the torch chain instead: its error is the same
    4e-5 the radix-4 pair has — the float32 phase written by the forward enters through g_phase / |Y| either way — and exact
    zeros come out where |Y| = 0 (silent utterance)."""
    from audio_deepfake_adversarial_attacks_amd.frontends import MelSpecFrontend
    fe = MelSpecFrontend().to(cuda)
    gen = torch.Generator().manual_seed(21)
    x = (torch.rand(3, 64_600, generator=gen) - 0.5).to(cuda)
    x[2] = 0.0
    gy = torch.randn(3, 2, 80, 404, generator=gen).to(cuda)

    def run(from_output, reg):
        monkeypatch.setenv("ADVSTEP_MEL_BWD_FROM_OUTPUT", "1" if from_output else "0")
        monkeypatch.setenv("ADVSTEP_STFT_REG", reg)
        a = x.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(fe(a), a, gy)
        return g

    g0, g1 = run(False, "0"), run(True, "0")
    assert (g0[:2] - g1[:2]).norm().item() <= 1e-5 * g0[:2].norm().item()
    assert not g0[2].any() and not g1[2].any()
    shipped = run(True, "1")
    assert not shipped[2].any() and torch.equal(run(True, "1"), shipped)
    monkeypatch.setenv("ADVSTEP_FUSED_MEL", "0")
    a = x.double().clone().requires_grad_(True)
    (g64,) = torch.autograd.grad(MelSpecFrontend().to(cuda).double()(a), a, gy.double())
    err = lambda g: ((g.double() - g64)[:2].norm() / g64[:2].norm()).item()
    assert err(shipped) <= 1.5e-4 and err(shipped) <= 1.25 * err(g1), (err(shipped), err(g1))


def test_fused_mel_spec_with_wide_bands(cuda, monkeypatch):
    """32 mel bands: the widest band covers more than 16 bins, which takes the kernels' second tap-register size."""
    from audio_deepfake_adversarial_attacks_amd import frontend_ops
    from audio_deepfake_adversarial_attacks_amd.frontends import MelScale, MelSpecFrontend, N_FFT, SAMPLING_RATE
    fe = MelSpecFrontend()
    fe.mel_scale = MelScale(32, SAMPLING_RATE, N_FFT // 2 + 1, persistent=False)
    fe = fe.to(cuda)
    assert 16 < frontend_ops.filterbank_tables(fe.mel_scale.fb).span <= 48
    x = (torch.rand(2, 8_000, generator=torch.Generator().manual_seed(3)) - 0.5).to(cuda)

    def run(fused, gy):
        monkeypatch.setenv("ADVSTEP_FUSED_MEL", "1" if fused else "0")
        a = x.clone().requires_grad_(True)
        y = fe(a)
        (g,) = torch.autograd.grad(y, a, gy)
        return y.detach(), g

    monkeypatch.setenv("ADVSTEP_FUSED_MEL", "0")
    with torch.no_grad():
        y0 = fe(x)
    gy = torch.randn(y0.shape, generator=torch.Generator().manual_seed(4)).to(cuda)
    gy[:, 1] *= (y0[:, 0] ** 2).clamp(max=1.0)
    y_ref, g_ref = run(False, gy)
    y, g = run(True, gy)
    scale = y_ref[:, 0].abs().max().item()
    assert y.shape == (2, 2, 32, 51)
    assert (y[:, 0] - y_ref[:, 0]).abs().max().item() <= 1e-5 * scale
    assert (g - g_ref).norm().item() / g_ref.norm().item() <= 1e-4


def test_specrnet_uses_fused_mel_frontend(cuda, monkeypatch):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    model = get_model("specrnet", {"input_channels": 2, "frontend_algorithm": ["mel_spec"]}, str(cuda)).to(cuda).eval()
    x = (torch.rand(2, 64_600, device=cuda) - 0.5)
    with torch.no_grad():
        monkeypatch.setenv("ADVSTEP_FUSED_MEL", "0")
        z0 = model(x)
        monkeypatch.setenv("ADVSTEP_FUSED_MEL", "1")
        z1 = model(x)
    assert z0.shape == (2, 1) and (z0 - z1).abs().max().item() <= 1e-3 * max(z0.abs().max().item(), 1.0)


def test_fused_mfcc_matches_torch_chain(cuda, monkeypatch):
    """MFCC = the LFCC structure with a 128-band mel filterbank (up to 6 bands per bin): same fused kernels."""
    from audio_deepfake_adversarial_attacks_amd.frontends import MFCC
    fe = MFCC().to(cuda)
    gen = torch.Generator().manual_seed(21)
    x = torch.rand(3, 16_000, generator=gen).to(cuda)
    monkeypatch.setenv("ADVSTEP_FUSED_LFCC", "0")
    gy = torch.randn(fe(x).shape, generator=gen).to(cuda)
    y_ref, g_ref = run(fe, x, gy, False, monkeypatch)
    y, g = run(fe, x, gy, True, monkeypatch)
    assert y.shape == y_ref.shape == (3, 80, 101)
    assert (y - y_ref).abs().max().item() <= 2e-3
    assert (g - g_ref).norm().item() / g_ref.norm().item() <= 1e-4


# ---- fused kernels vs INDEPENDENT third-party arithmetic (transformers.audio_utils + scipy.fft) -------------------------------
# tests/golden/frontends_xcheck.npz; not the reference (torchaudio 0.10 is absent: parity stays unpinned), but no longer
# only this repository's own reading of the algorithm.

@pytest.mark.parametrize("tag", ["full", "short", "loud"])
def test_fused_lfcc_matches_independent_implementation(cuda, golden, parity_record, tag):
    from audio_deepfake_adversarial_attacks_amd import frontends
    from tests.test_frontends import lfcc_error
    err = lfcc_error(frontends.LFCC().to(cuda), golden("frontends_xcheck"), tag, cuda)
    parity_record[f"fused_lfcc_vs_third_party_{tag}_max_over_scale"] = err
    assert err <= 2e-5, err


@pytest.mark.parametrize("tag", ["full", "short", "loud"])
def test_fused_mel_spec_matches_independent_implementation(cuda, golden, parity_record, tag):
    from audio_deepfake_adversarial_attacks_amd import frontends
    from tests.test_frontends import mel_error
    err = mel_error(frontends.MelSpecFrontend().to(cuda), golden("frontends_xcheck"), tag, cuda)
    parity_record[f"fused_mel_spec_vs_third_party_{tag}_max_over_scale"] = err
    assert err <= 2e-5, err


def test_fused_lfcc_db_floor_is_taken_over_the_whole_batch(cuda, golden, parity_record):
    """tests/test_frontends.py::test_lfcc_db_floor_is_taken_over_the_whole_batch for the fused kernels (the floor is a
    batch-wide maximum found in a separate launch, `advstep_lfcc_reduce_max_f32`): the batch result equals the third-party
    chain with the batch-wide floor applied by hand, and is NOT what per-utterance floors give."""
    from audio_deepfake_adversarial_attacks_amd import frontends
    from tests.test_frontends import batch_floor_errors
    vs_batch, vs_each, alone_vs_each = batch_floor_errors(frontends.LFCC().to(cuda), golden("frontends_batch_floor"), cuda)
    parity_record["fused_lfcc_batch_floor"] = {"vs_batch_wide_floor_max_over_scale": vs_batch,
                                               "vs_per_utterance_floors_max_over_scale": vs_each,
                                               "one_at_a_time_vs_per_utterance_floors": alone_vs_each}
    assert vs_batch <= 2e-5 and alone_vs_each <= 2e-5, (vs_batch, alone_vs_each)
    assert vs_each >= 1e-2, vs_each


_TWO_STREAM_SCRIPT = r"""
import os, sys
import torch
sys.path.insert(0, os.environ["ADVSTEP_REPO"])
from audio_deepfake_adversarial_attacks_amd import frontends
dev = torch.device("cuda:0")
lfcc = frontends.LFCC().to(dev)
mel = frontends.MelSpecFrontend().to(dev)
gen = torch.Generator().manual_seed(11)
xs = [torch.rand(4, 64_600, generator=gen).to(dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
torch.cuda.synchronize()
out = [None, None]
# the process's FIRST launches of the STFT kernels, one per stream, back to back, nothing ordering the streams
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        a = xs[i].clone().requires_grad_(True)
        y = lfcc(a)
        (g,) = torch.autograd.grad(y, a, torch.ones_like(y))
        m = mel(xs[i])
        out[i] = (y.detach(), g, m)
torch.cuda.synchronize()
ok = True
for i in range(2):        # the same work again, on the default stream, long after any start-up effect
    a = xs[i].clone().requires_grad_(True)
    y = lfcc(a)
    (g,) = torch.autograd.grad(y, a, torch.ones_like(y))
    m = mel(xs[i])
    ok &= bool(torch.equal(y.detach(), out[i][0]) and torch.equal(g, out[i][1]) and torch.equal(m, out[i][2]))
    ok &= bool(torch.isfinite(y).all() and torch.isfinite(g).all())
print("RESULT", "equal" if ok else "DIFFERENT")
"""


def test_first_calls_on_two_streams_need_no_shared_setup(cuda, tmp_path):
    """include/advstep.h promises "no global state"; until round 3 the STFT kernels filled a device-global
    twiddle table on the first caller's stream, so a second stream's first call could race it.  The tables are compile-time
    constants now (csrc/stft_tables.inc): in a fresh process, first calls issued on two streams with nothing ordering them
    must give the bits a later call gives."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    script = tmp_path / "two_streams.py"
    script.write_text(_TWO_STREAM_SCRIPT)
    env = dict(os.environ, ADVSTEP_REPO=str(Path(__file__).resolve().parent.parent))
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert "RESULT equal" in proc.stdout, proc.stdout[-500:]
