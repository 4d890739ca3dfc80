"""Which kernel moves the product's input gradient away from the CPU oracle's?  (VERDICT r03 "what's weak" item 1.)

At one iterate (FGSM's x01 of configs[0]; PGD's iteration 0 of configs[1], teacher-forced so nothing compounds) the
product's forward + input-backward on LCNN + LFCC is run under every ADVSTEP_* switch setting of interest — everything
fused (what ships), everything off (plain PyTorch-ROCm: MIOpen + rocFFT), each switch off on its own, each switch on on
its own — and compared with the CPU oracle (oracle/attacks.py::_cost_and_grad on a CPU copy of the same weights):

    sign flips of the gradient, its relative L2 error, the logits' max-abs error, and
    PER LAYER the number of max-feature-map / max-pool winners that differ from the CPU run's (a "re-route")

The winners of a fused block come from the selection bytes it saved for its backward (expanded with the library's own
backward kernels, so no encoding is restated here); those of an unfused block from the convolution's output.  A float64
CPU run of the same iterate gives the scale: how many winners the ORACLE's own float32 rounding re-routes.

Reference: src/models/lcnn.py:76-95 (max-feature-map), :120-157 (trunk), fgsm.py:59-60, pgd.py:74-76.
Test infrastructure (imports oracle/): used by tests/test_gpu_parity_attribution.py and `python -m tests.parity_attribution`."""
from __future__ import annotations

import copy
import json
import os
import sys
from contextlib import contextmanager
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import attacks as OA                     # noqa: E402
from tests import e2e_parity as E                     # noqa: E402

# every switch the LCNN + LFCC path reads (models/lcnn.py, frontends.py, frontend_ops.py), default "1"
SWITCHES = ("ADVSTEP_FUSED_STFT", "ADVSTEP_FUSED_LFCC", "ADVSTEP_LCNN_CONV0", "ADVSTEP_LCNN_CONV1X1",
            "ADVSTEP_LCNN_CONV3X3", "ADVSTEP_LCNN_LSTM", "ADVSTEP_LCNN_BN", "ADVSTEP_LCNN_FUSED")


@contextmanager
def switches(**values):
    """Set ADVSTEP_* switches for the duration of the block (they are read at call time)."""
    before = {k: os.environ.get(k) for k in values}
    os.environ.update({k: str(v) for k, v in values.items()})
    try:
        yield
    finally:
        for k, v in before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def settings():
    """name -> {switch: "0" | "1"} for every switch."""
    rows = {"all_fused": {s: "1" for s in SWITCHES}, "plain_pytorch_rocm": {s: "0" for s in SWITCHES}}
    for s in SWITCHES:
        short = s.replace("ADVSTEP_", "").lower()
        rows[f"only_{short}_off"] = {k: ("0" if k == s else "1") for k in SWITCHES}
    for s in SWITCHES:
        if s == "ADVSTEP_LCNN_FUSED":
            continue
        short = s.replace("ADVSTEP_", "").lower()
        on = {k: "0" for k in SWITCHES}
        on[s] = "1"
        on["ADVSTEP_LCNN_FUSED"] = "1"        # the block switches sit below it (models/lcnn.py::_transform)
        if s == "ADVSTEP_FUSED_STFT":
            on["ADVSTEP_FUSED_LFCC"] = "1"    # the in-LDS FFT kernel feeds the fused tail
        rows[f"only_{short}_on"] = on
    return rows


# ---- winners -----------------------------------------------------------------------------------------------------------------

def _codes_from_mask(mask: torch.Tensor, pooled: bool) -> torch.Tensor:
    """(N, 2C, H, W) 0/1 mask of the elements a block's output selects -> one small integer per block output."""
    N, C2, H, W = mask.shape
    C = C2 // 2
    m = mask.reshape(N, 2, C, H, W)
    if not pooled:
        return m[:, 1].to(torch.uint8)
    H2, W2 = H // 2, W // 2
    m = m[:, :, :, :2 * H2, :2 * W2].reshape(N, 2, C, H2, 2, W2, 2).permute(0, 2, 3, 5, 1, 4, 6).reshape(N, C, H2, W2, 8)
    return m.to(torch.float32).argmax(dim=-1).to(torch.uint8)


def _codes_from_conv_output(h: torch.Tensor, pooled: bool) -> torch.Tensor:
    """The winners ATen's max-feature-map (+ MaxPool2d(2, 2)) picks on a convolution output (bias included)."""
    with torch.enable_grad():
        h = h.detach().clone().requires_grad_(True)
        N, C2, H, W = h.shape
        y = h.view(N, 2, C2 // 2, H, W).max(1)[0]
        if pooled:
            y = F.max_pool2d(y, 2, 2)
        (g,) = torch.autograd.grad(y.sum(), h)
    return _codes_from_mask(g != 0, pooled)


def _ones_like_output(N, C, H, W, pooled, device):
    return torch.ones((N, C, H // 2, W // 2) if pooled else (N, C, H, W), dtype=torch.float32, device=device)


def _codes_from_selection(kind: str, saved, shape, pooled: bool) -> torch.Tensor:
    """The winners a fused block recorded, through the library's own expansion kernels."""
    from audio_deepfake_adversarial_attacks_amd import _lib
    from audio_deepfake_adversarial_attacks_amd.hip_ops import _stream
    lib = _lib.load()
    N, C, H, W = shape
    dev = saved.device
    if kind == "bits":                     # conv1x1 (round-6 layout, include/advstep_lcnn.h): [n][p / 32][p % 32][h] lane masks
        P = H * W
        PW = (P + 31) // 32
        wide = C > 32
        raw = saved[: N * PW * 64 * (4 if wide else 2)].view(torch.int32 if wide else torch.int16).view(N, PW, 32, 2)
        raw = raw.to(torch.int64) & (0xFFFFFFFF if wide else 0xFFFF)
        c = torch.arange(C, device=dev)
        half, bit = (c >> 2) & 1, 16 * (c >> 5) + (c & 3) + 4 * ((c & 31) >> 3)
        bits = (raw[:, :, :, half] >> bit) & 1                       # (N, PW, 32, C)
        return bits.permute(0, 3, 1, 2).reshape(N, C, PW * 32)[:, :, :P].reshape(N, C, H, W).to(torch.uint8)
    gx = torch.empty((N, 2 * C, H, W), dtype=torch.float32, device=dev)
    ones = _ones_like_output(N, C, H, W, pooled, dev)
    if kind == "pool":
        st = lib.advstep_mfm_pool2_backward_f32(ones.data_ptr(), saved.data_ptr(), None, gx.data_ptr(), N, C, H, W, _stream(dev))
    elif kind == "wino_mfm":
        st = lib.advstep_conv3x3_mfm_backward_f32(ones.data_ptr(), saved.data_ptr(), None, gx.data_ptr(), N, C, H, W, _stream(dev))
    else:
        st = lib.advstep_mfm_backward_f32(ones.data_ptr(), saved.data_ptr(), None, gx.data_ptr(), N, C, H * W, _stream(dev))
    _lib.check(st, "selection expansion")
    return _codes_from_mask(gx != 0, pooled)


class WinnerTap:
    """Collects, in trunk order, the winner codes of every conv -> max-feature-map [-> pool] block of one forward pass."""

    def __init__(self):
        self.codes = []
        self.inputs = []      # what every block's convolution read (block 0: the frontend's output), float64 on the CPU

    def _keep_input(self, x):
        self.inputs.append(x.detach().double().cpu())

    # -- unfused blocks (CPU model; GPU model with ADVSTEP_LCNN_FUSED=0): hooks on the Sequential's convolutions
    def hook_modules(self, base):
        from audio_deepfake_adversarial_attacks_amd.models.lcnn import MaxFeatureMap2D
        mods = list(base.m_transform)
        handles = []
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], MaxFeatureMap2D):
                pooled = i + 2 < len(mods) and isinstance(mods[i + 2], nn.MaxPool2d)

                def hook(mod, inp, out, pooled=pooled):
                    self._keep_input(inp[0])
                    self.codes.append(_codes_from_conv_output(out, pooled).cpu())
                handles.append(m.register_forward_hook(hook))
        return handles

    # -- fused blocks: wrap the lcnn_ops entry points the model calls
    @contextmanager
    def patch_ops(self):
        from audio_deepfake_adversarial_attacks_amd import lcnn_ops
        names = {"conv5_mfm_pool2": ("pool", True), "conv3x3_mfm_pool2": ("pool", True), "mfm_pool2": ("pool", True),
                 "conv3x3_mfm": ("wino_mfm", False), "mfm": ("mfm", False), "conv1x1_mfm": ("bits", False)}
        originals = {n: getattr(lcnn_ops, n) for n in names}

        conv2d = F.conv2d
        fused_trunk = os.environ.get("ADVSTEP_LCNN_FUSED", "1") != "0"

        def traced_conv2d(x, *args, **kwargs):      # _transform's own F.conv2d in front of lcnn_ops.mfm / mfm_pool2
            if x.dim() == 4 and x.is_cuda:
                self._keep_input(x)
            return conv2d(x, *args, **kwargs)

        def wrap(name, kind, pooled):
            def call(x, *args, **kwargs):
                if name not in ("mfm", "mfm_pool2"):
                    self._keep_input(x)
                y = originals[name](x, *args, **kwargs)
                saved = y.grad_fn.saved_tensors[0]
                N, C = y.shape[0], y.shape[1]
                H, W = x.shape[2], x.shape[3]
                self.codes.append(_codes_from_selection(kind, saved, (N, C, H, W), pooled).cpu())
                return y
            return call

        for n, (kind, pooled) in names.items():
            setattr(lcnn_ops, n, wrap(n, kind, pooled))
        if fused_trunk:         # with the trunk unfused the Conv2d modules' hooks see the same calls
            F.conv2d = traced_conv2d
        try:
            yield
        finally:
            F.conv2d = conv2d
            for n, f in originals.items():
                setattr(lcnn_ops, n, f)


def _reroutes(a, b):
    assert len(a) == len(b), (len(a), len(b))
    return [int((x != y).sum()) for x, y in zip(a, b)]


# ---- one iterate under one setting -------------------------------------------------------------------------------------------

def product_gradient(model, ops, adv, y):
    """The shipped path under the current switches: grad, logits, winner codes and convolution inputs per block."""
    tap = WinnerTap()
    handles = tap.hook_modules(model)
    probe = E.armed(E.GradProbe(model), ops)
    try:
        with tap.patch_ops():
            grad, cost, z = probe(adv, y)
    finally:
        for h in handles:
            h.remove()
    return {"grad": grad.cpu(), "z": z.cpu().reshape(-1), "codes": tap.codes, "inputs": tap.inputs}


def oracle_gradient(model_cpu, adv, y, dtype=torch.float32):
    m = model_cpu if dtype == torch.float32 else copy.deepcopy(model_cpu).to(dtype)
    tap = WinnerTap()
    handles = tap.hook_modules(m)
    try:
        with OA.attack_mode(m):
            grad, cost, z = OA._cost_and_grad(m, adv.clone().detach().to(dtype), y, with_cost=True)
    finally:
        for h in handles:
            h.remove()
    return {"grad": grad, "z": z.reshape(-1), "codes": tap.codes, "inputs": tap.inputs}


def _row(run, ref, truth):
    """`run` against `ref` (flips, gradient, logits, winners) and against the float64 `truth` (the error of what each
    block's convolution reads, relative RMS: the quantity a near-tie winner is sensitive to)."""
    g, g_ref = run["grad"].double(), ref["grad"].double()
    same = g.sign() == g_ref.sign()
    rel = g_ref.abs() / g_ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    rr = _reroutes(run["codes"], ref["codes"])
    # utterances none of whose winners differ: there the two gradients are the same function evaluated in two roundings
    rows_hit = sum((a != b).flatten(1).sum(1) for a, b in zip(run["codes"], ref["codes"])) > 0
    row_rel = (g - g_ref).norm(dim=1) / g_ref.norm(dim=1).clamp_min(1e-300)
    clean = ~rows_hit
    return {"utterances_without_reroute": int(clean.sum()),
            "grad_rel_l2_without_reroute_worst": row_rel[clean].max().item() if clean.any() else 0.0,
            "sign_flips_without_reroute": int((~same)[clean].sum()),
            "sign_flips": int((~same).sum()), "samples": same.numel(),
            "flip_rel_worst": rel[~same].max().item() if (~same).any() else 0.0,
            "grad_rel_l2": ((g - g_ref).norm() / g_ref.norm()).item(),
            "logit_max_abs": (run["z"].double() - ref["z"].double()).abs().max().item(),
            "reroutes_per_block": rr, "reroutes": sum(rr),
            "reroutes_vs_f64_per_block": _reroutes(run["codes"], truth["codes"]),
            "input_rel_rms_error_vs_f64_per_block": [((a - b).norm() / b.norm()).item()
                                                      for a, b in zip(run["inputs"], truth["inputs"])],
            "grad_rel_l2_vs_f64": ((g - truth["grad"].double()).norm() / truth["grad"].double().norm()).item()}


def batches(batch, seeds):
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    for seed in seeds:
        x, y = synthetic_waveforms(batch, seed=seed)
        y[: batch // 2], y[batch // 2:] = 0, 1
        x01, _, _ = OA.to_minmax(x)
        eps = 0.003
        noise = torch.empty_like(x).uniform_(-eps, eps, generator=torch.Generator().manual_seed(77 + seed))
        yield f"seed{seed}_configs0_fgsm_x01", x01, y
        yield f"seed{seed}_configs1_pgd_iteration0", torch.clamp(x01 + noise, min=0, max=1), y


def attribute(model, ops, device, batch=8, seeds=(1234,), threads=16, which=None):
    """{iterate: {"oracle_f32": row of the float32 oracle against float64, setting: row of the product against the
    float32 oracle, ...}} plus "total": the same rows summed over the iterates."""
    model_cpu = copy.deepcopy(model).cpu()
    rows = settings()
    if which:
        rows = {k: v for k, v in rows.items() if k in which}
    table = {}
    for name, adv, y in batches(batch, seeds):
        with E.cpu_threads(threads):
            o32 = oracle_gradient(model_cpu, adv, y)
            o64 = oracle_gradient(model_cpu, adv, y, torch.float64)
        out = {"winners_per_block": [int(k.numel()) for k in o32["codes"]], "oracle_f32": _row(o32, o64, o64)}
        for setting, env in rows.items():
            with switches(**env):
                run = product_gradient(model, ops, adv.to(device), y.to(device))
            out[setting] = _row(run, o32, o64)
        table[name] = out
    table["total"] = _total(table)
    return table


def _total(table):
    names = [k for k in next(iter(table.values())) if k != "winners_per_block"]
    tot = {"winners_per_block": [sum(v) for v in zip(*(t["winners_per_block"] for t in table.values()))]}
    for k in names:
        rows = [t[k] for t in table.values()]
        tot[k] = {"sign_flips": sum(r["sign_flips"] for r in rows), "samples": sum(r["samples"] for r in rows),
                  "flip_rel_worst": max(r["flip_rel_worst"] for r in rows),
                  "grad_rel_l2": max(r["grad_rel_l2"] for r in rows),
                  "grad_rel_l2_best": min(r["grad_rel_l2"] for r in rows),
                  "utterances_without_reroute": sum(r["utterances_without_reroute"] for r in rows),
                  "grad_rel_l2_without_reroute_worst": max(r["grad_rel_l2_without_reroute_worst"] for r in rows),
                  "sign_flips_without_reroute": sum(r["sign_flips_without_reroute"] for r in rows),
                  "grad_rel_l2_vs_f64": max(r["grad_rel_l2_vs_f64"] for r in rows),
                  "logit_max_abs": max(r["logit_max_abs"] for r in rows),
                  "reroutes": sum(r["reroutes"] for r in rows),
                  "reroutes_per_block": [sum(v) for v in zip(*(r["reroutes_per_block"] for r in rows))],
                  "reroutes_vs_f64_per_block": [sum(v) for v in zip(*(r["reroutes_vs_f64_per_block"] for r in rows))],
                  "input_rel_rms_error_vs_f64_per_block":
                      [max(v) for v in zip(*(r["input_rel_rms_error_vs_f64_per_block"] for r in rows))]}
    return tot


def render(table) -> str:
    lines = []
    for name, rows in table.items():
        lines.append(f"== {name}: rows = product (MI355X) vs the float32 CPU oracle; `oracle_f32` = that oracle vs float64")
        lines.append(f"   winners per block: {rows['winners_per_block']}")
        lines.append(f"{'setting':30s} {'flips':>6s} {'grad relL2':>10s} {'clean rows: n / flips / relL2':>30s} {'logit':>8s} {'reroutes':>8s}  per block"
                     f"{'':22s}| vs f64 per block{'':14s}| conv-input rel. RMS error vs f64 per block (x 1e-7)")
        for k, r in rows.items():
            if k == "winners_per_block":
                continue
            err = " ".join(f"{v * 1e7:5.1f}" for v in r["input_rel_rms_error_vs_f64_per_block"])
            clean = f"{r['utterances_without_reroute']:4d} / {r['sign_flips_without_reroute']:3d} / {r['grad_rel_l2_without_reroute_worst']:.2e}"
            lines.append(f"{k:30s} {r['sign_flips']:6d} {r['grad_rel_l2']:10.2e} {clean:>30s} {r['logit_max_abs']:8.1e} {r['reroutes']:8d}  "
                         f"{str(r['reroutes_per_block']):30s} | {str(r['reroutes_vs_f64_per_block']):30s} | {err}")
        lines.append("")
    return "\n".join(lines)


def main():
    import argparse
    from audio_deepfake_adversarial_attacks_amd import hip_ops
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1234, 99])
    ap.add_argument("--out", default="parity_attribution")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(device)).to(device).eval()
    table = attribute(model, hip_ops, device, batch=args.batch, seeds=tuple(args.seeds))
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / f"{args.out}.json").write_text(json.dumps(E.slim(table), indent=1))
    text = render(table)
    (out / f"{args.out}.txt").write_text(text)
    print(text)


if __name__ == "__main__":
    main()
