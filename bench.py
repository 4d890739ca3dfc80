#!/usr/bin/env python
"""Benchmark of the adversarial-evaluation hot loop on MI355X.  Default = the headline: adversarial utterances / second,
PGD-40 (L-inf, eps = 0.003, alpha = 2/255) on LCNN + LFCC over synthetic 4 s @ 16 kHz utterances (T = 64 600), 128
utterances per GPU (BASELINE.json configs[1]; configs[4] is the same workload on 8 GPUs).

A "step" is one pass of the hot-loop body of evaluate_models_on_adversarial_attacks.py:211-265 over one batch that
is already resident in HBM:  to_minmax -> attack (N x [model fwd + input-bwd, fused HIP update step]) ->
revert_minmax -> target-model forward -> sigmoid / threshold.  After the K timed steps the per-utterance scores are
aggregated once (RCCL all-reduce + all-gather when N > 1) inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3}]

With N > 1 and no launcher in the environment (WORLD_SIZE unset) the script starts its own N ranks — it re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` — so both
`python bench.py --gpus 8` and the explicit launcher line work; rank 0 prints ONE JSON line either way.

--config selects the BASELINE.json workload (the line's `metric` / `config.workload` name it):
    1  configs[1]  LCNN + LFCC, PGD-40 L-inf eps 0.003, B = 128        roofline: advstep_pgd_linf_step_f32, 16 B/sample
    2  configs[2]  SpecRNet + mel-spec, PGDL2-40 eps 0.1, B = 128      roofline: advstep_pgd_l2_step_f32,   16 B/sample
    3  configs[3]  RawNet3 -> LCNN + LFCC transfer, FGSM + CW-100, B = 64 (a step attacks the batch with both)
                                                                       roofline: advstep_cw_adam_step_f32,  32 B/sample

Round 6: the timed region runs the SHIPPED launch path — the attack iteration's model part replayed from its hipGraph, the
update step launched (and bracketed) between the replays — with `--in-flight` batches in flight on streams of their own
(default 2 for the graph-replayed attacks of configs 1 and 2: batch i on stream i % 2; evaluation._Lanes).  Every step is
still one batch through the whole body; K steps are timed between the same barriers.  Untimed priming steps in front of the
W warm-up steps let every stream capture its graph (a graph is captured at the second sight of a workload).

`roofline` prices the dominant hand-written kernel of the workload with HIP events recorded on its launch stream
inside the timed region; `cpu_baseline` times the CPU oracle ("port": oracle/attacks.py, torch CPU ops in the
reference's order) on a bounded sample of the same workload on this box's host cores (rank 0, N = 1 only)."""
import argparse
import json
import os
import socket
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

T = 64_600
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: FP32 (matrix) 157.3 TFLOP/s spec (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), dense
LCNN = ("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1})
SPECRNET = ("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2})
RAWNET3 = ("rawnet3", {})

# attacks are named by their AttackEnum member (aa/aa_types.py) so bench, CLI and tests run the same hyper-parameters;
# `kernel` = (hip_ops entry point, algorithmic bytes per waveform sample per launch — SURVEY.md section 8(d))
WORKLOADS = {
    1: dict(tag="configs[1]", target=LCNN, attacked=LCNN, white_box=True, attacks=("PGD40_eps003",), batch=128,
            metric="adversarial utterances/sec, PGD-40 LCNN+LFCC 4s@16kHz",
            what="LCNN+LFCC, PGD-40 Linf eps=0.003 alpha=2/255 random start",
            kernel=("pgd_linf_step", 16, "advstep_pgd_linf_step_f32 (flat_vec_kernel<3, PgdLinfOp>)")),
    2: dict(tag="configs[2]", target=SPECRNET, attacked=SPECRNET, white_box=True, attacks=("PGDL2_40",), batch=128,
            metric="adversarial utterances/sec, PGDL2-40 SpecRNet+mel-spec 4s@16kHz",
            what="SpecRNet+mel-spec (2 channels), PGDL2-40 eps=0.1 alpha=0.2 random start",
            kernel=("pgd_l2_step", 16, "advstep_pgd_l2_step_f32 (all kernels of one call)")),
    3: dict(tag="configs[3]", target=LCNN, attacked=RAWNET3, white_box=False, attacks=("FGSM", "CW"), batch=64,
            metric="adversarial utterances/sec, FGSM + CW-100 transfer pair RawNet3->LCNN+LFCC 4s@16kHz",
            what="attack model RawNet3 (raw waveform), target LCNN+LFCC; every batch is attacked with FGSM eps=0.0005 "
                 "and with CW c=1 kappa=0 steps=100 lr=0.01, each adversarial batch scored by the target",
            kernel=("cw_adam_step", 32, "advstep_cw_adam_step_f32 (cw_adam_vec_kernel)")),
}
# matrix-core kernels of the attacked model (lcnn_ops / detector_ops launch names); their launches carry the flop they must
# put through the matrix cores (Winograd F(2x2, 3x3): a direct 3x3 convolution's count / 2.25, unpadded)
MODEL_MATRIX_KERNELS = {
    1: ("conv3x3_mfm_pool2_forward", "conv3x3_mfm_pool2_backward", "conv3x3_mfm_forward", "conv3x3_backward_data"),
    2: ("resconv_forward", "resconv_pool2_forward", "resconv_pooled_grad"),
    3: (),        # RawNet3: library GEMMs (rocBLAS / hipBLASLt), no hand-written matrix-core kernel to price
}
# `roofline_step`: every hand-written launch of one step, grouped by what bounds it (entry-point prefixes of hip_ops /
# lcnn_ops / frontend_ops / detector_ops launch names).  mfma: flop through the fp32 matrix cores; valu: flop on the vector
# ALUs (packed fp32: the same 256 flop / clock / CU); hbm: the bytes of the operands each launch reads or writes once.
VALU_F32_PEAK_TFLOPS = 157.3   # 256 CUs x 256 flop / clock (v_pk_fma_f32) x 2.4 GHz: the guide's FP32 vector figure
STEP_FAMILIES = (
    ("3x3 convolutions, Winograd F(2x2,3x3) on the matrix cores", "mfma",
     ("conv3x3_mfm_pool2_forward", "conv3x3_mfm_pool2_backward", "conv3x3_mfm_forward", "conv3x3_backward_data",
      "resconv_forward", "resconv_pool2_forward", "resconv_pooled_grad")),
    ("first block 5x5 / few-channel 3x3 convolutions (vector ALU)", "valu",
     ("conv5_mfm_pool2_forward", "conv5_mfm_pool2_backward", "conv3x3_fewin_forward", "conv3x3_fewout_grad")),
    ("1x1 convolution + max-feature-map blocks", "hbm", ("conv1x1_mfm_forward", "conv1x1_mfm_backward")),
    ("frontend: STFT (in-LDS FFT) + filterbank + dB + DCT, both directions", "hbm",
     ("lfcc_forward", "lfcc_backward", "stft_frames", "stft_overlap_add", "stft_mel", "stft_mel_backward")),
    ("library GEMMs: the recurrent layers' input projections (rocBLAS)", "mfma", ("rnn_projection_gemm",)),
    ("library matrix products and convolutions outside the package's launches (rocBLAS / hipBLASLt / MIOpen: RawNet3's 1x1 and "
     "dilated convolutions as GEMMs, its sinc encoder, the attention's projections)", "mfma", ("library_matrix_op",)),
    ("other ATen kernels (elementwise, reductions, copies, cat) launched by the model and the attack loop", "hbm", ("aten_other",)),
    ("recurrent layers (one workgroup per utterance and direction) + tail", "hbm",
     ("lstm_forward", "lstm_backward", "gru_forward", "gru_backward", "lcnn_tail_pack", "lcnn_tail_forward",
      "lcnn_tail_unpack_add")),
    ("elementwise / pooling / selection kernels of the detector", "hbm",
     ("mfm_forward", "mfm_backward", "mfm_pool2_forward", "mfm_pool2_backward", "conv3x3_mfm_backward", "affine_act_forward",
      "affine_act_backward", "add_maxpool2_forward", "maxpool2_backward", "add_maxpool1d_forward", "maxpool1d_backward",
      "gate_maxpool2_forward", "gate_maxpool2_backward", "weighted_stats_forward", "weighted_stats_backward",
      "log_meannorm_forward", "log_meannorm_backward", "tail_pool1d_forward", "tail_pool1d_backward", "attend_pool_forward",
      "attend_pool_backward", "attend_gate_fc", "afms_row", "res2net_link_forward", "res2net_link_backward")),
    ("attack step, random start, min-max, loss gradient", "hbm",
     ("pgd_linf_step", "pgd_linf_init", "pgd_l2_step", "pgd_l2_init", "fgsm_step", "cw_adam_step", "cw_tanh_sqdist",
      "cw_best_update", "cw_init_w", "minmax_normalize", "minmax_revert", "ce2_loss_grad")),
)
PROFILED = ("pgd_linf_step", "pgd_linf_init", "pgd_l2_step", "pgd_l2_init", "fgsm_step", "cw_adam_step",
            "cw_tanh_sqdist", "cw_best_update", "minmax_normalize", "minmax_revert", "ce2_loss_grad")


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed steps (default 4; 2 for --config 3)")
    p.add_argument("--warmup", type=int, default=None, help="untimed steps (default 2; 1 for --config 3)")
    p.add_argument("--config", type=int, default=1, choices=sorted(WORKLOADS), help="BASELINE.json configs[N]")
    p.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default: the config's)")
    p.add_argument("--in-flight", type=int, default=None,
                   help="batches in flight, each on its own stream (default: 2 for --config 1 and 2, 1 for --config 3 whose CW "
                        "loop is driven from the host)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the legs after the timed region (graph-replay timing, model-side roofline, cold-regime and "
                        "live PMC traffic of the priced kernel); N = 1 only anyway")
    p.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = swept default)")
    p.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                   help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for single-GPU rehearsals)")
    p.add_argument("--share-device", action="store_true",
                   help="rehearsal on a box with fewer GPUs than ranks: every rank uses cuda:0 (needs --backend gloo)")
    a = p.parse_args(argv)
    if a.steps is None:
        a.steps = 2 if a.config == 3 else 4
    if a.warmup is None:
        a.warmup = 1 if a.config == 3 else 2
    if a.in_flight is None:
        a.in_flight = 1 if a.config == 3 else 2
    return a


def launcher_command(argv, gpus, port=None):
    """The command `python bench.py --gpus N ...` replaces itself with when no launcher set WORLD_SIZE."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def build_models(spec, device):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    set_seed(42)
    target = get_model(spec["target"][0], dict(spec["target"][1]), device).to(device)
    attacked = get_model(spec["attacked"][0], dict(spec["attacked"][1]), device).to(device)
    if spec["white_box"]:
        attacked.load_state_dict(target.state_dict())  # same weights
    return target.eval(), attacked.eval()


def cpu_baseline(config: int, threads: int, iterations: float = 0.0):
    """Oracle ("port") on the host cores: the same per-batch body on a bounded sample of the same workload, run in
    a separate process (oracle/cpu_baseline.py) after the GPU measurement."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--config", str(config), "--threads", str(threads),
           "--iterations", str(iterations)]
    fail = {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port"}
    try:
        proc = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
        if proc.returncode == 0 and lines:
            return json.loads(lines[-1])
        return dict(fail, sample=f"failed rc={proc.returncode}: {proc.stderr[-300:]}")
    except subprocess.TimeoutExpired:
        return dict(fail, sample="timed out (900 s)")


def library_op_brackets(everything: bool = False):
    """A dispatch mode that brackets every LIBRARY matrix product / convolution (aten mm / addmm / bmm / baddbmm / convolution /
    convolution_backward: rocBLAS, hipBLASLt, MIOpen underneath) with HIP events on the launch stream and records its flop
    (2 x multiply-adds of the direct product / convolution) - RawNet3's iteration is 77 % such calls and `roofline_step` could not
    see them (What's weak 7).  Calls inside one of the package's own bracketed launches (the recurrent layers'
    projections, lcnn_ops._gemm) are left to that bracket.  The autograd engine carries the mode to its device thread.
    everything: also bracket every OTHER aten call that launches something (elementwise, reductions, copies, cat: name "aten",
    priced by the bytes of its tensor arguments and results) - for host-driven workloads (configs[3]) whose step is otherwise
    a quarter unpriced."""
    import math

    import torch
    from torch.utils._python_dispatch import TorchDispatchMode

    from audio_deepfake_adversarial_attacks_amd import hip_ops

    aten = torch.ops.aten

    def conv_flop(out_numel, weight, groups):
        return 2.0 * out_numel * (weight.shape[1]) * math.prod(weight.shape[2:])        # weight (Cout, Cin / groups, *k)

    # metadata-only ops: no kernel behind them (bracketing one reads an empty event pair, which roofline_step subtracts anyway,
    # but every bracket costs host time in a loop the host drives)
    VIEWS = {"view", "_unsafe_view", "reshape", "_reshape_alias", "expand", "permute", "transpose", "t", "slice", "select",
             "unsqueeze", "squeeze", "detach", "alias", "as_strided", "split", "split_with_sizes", "unbind", "chunk", "narrow",
             "unfold", "view_as", "expand_as", "flatten", "unflatten", "movedim", "swapaxes", "diagonal", "lift_fresh", "empty",
             "empty_like", "empty_strided", "new_empty", "new_empty_strided", "is_same_size", "sym_size", "sym_stride",
             "sym_numel", "size", "stride", "numel", "dim", "_local_scalar_dense", "item", "result_type", "set_", "resize_"}

    def tensor_bytes(*trees):
        total = 0
        for tree in trees:
            for t in (tree if isinstance(tree, (list, tuple)) else (tree,)):
                if isinstance(t, torch.Tensor):
                    total += t.numel() * t.element_size()
                elif isinstance(t, (list, tuple)):
                    total += tensor_bytes(*t)
        return total

    class Mode(TorchDispatchMode):
        def __init__(self, everything: bool = False):
            super().__init__()
            self.everything = everything
            self.records = []                      # (name, event0, event1, flop or bytes)

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            packet = getattr(func, "overloadpacket", None)
            name = flop = None
            first = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if hip_ops._launch_depth == 0 and first is not None and first.is_cuda:
                if packet in (aten.mm, aten.bmm):
                    a, b = args[0], args[1]
                    name, flop = "gemm", 2.0 * a.numel() * b.shape[-1]
                elif packet in (aten.addmm, aten.baddbmm):
                    a, b = args[1], args[2]
                    name, flop = "gemm", 2.0 * a.numel() * b.shape[-1]
                elif packet is aten.convolution:
                    name = "convolution"
                elif packet is aten.convolution_backward:
                    name = "convolution_backward"
                elif self.everything and getattr(packet, "__name__", "") not in VIEWS and not torch.cuda.is_current_stream_capturing():
                    name = "aten"
            if name is None:
                return func(*args, **kwargs)
            stream = torch.cuda.current_stream(first.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            out = func(*args, **kwargs)
            e1.record(stream)
            if name == "convolution":
                flop = conv_flop(out.numel(), args[1], args[8])
            elif name == "convolution_backward":
                mask = args[10]
                flop = conv_flop(args[0].numel(), args[2], args[9]) * (int(mask[0]) + int(mask[1]))
            elif name == "aten":
                flop = float(tensor_bytes(args, out))           # bytes of the tensors the call reads and writes once
            self.records.append((name, e0, e1, flop))
            return out

    return Mode(everything)


def through_loop(config: int, spec, B: int, device, warm: int = 6, timed: int = 16, workers: int = 3, in_flight=None):
    """Steady-state rate of the SHIPPED loop (evaluation.generate_attacks: DataLoader with `workers` worker processes,
    pinned staging + side-stream upload one batch ahead, hipGraph replay of the attack iteration, scores kept on the device)
    for the workload's first attack: utterances / second between a HIP event recorded when batch `warm` has been queued and
    one when the last batch has, read after the loop's single end-of-run synchronisation
    (reference: evaluate_models_on_adversarial_attacks.py:197-265)."""
    import torch
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
    from audio_deepfake_adversarial_attacks_amd.evaluation import generate_attacks
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed

    def cfg(model):
        return {"data": {"seed": 42}, "checkpoint": {"path": ""}, "model": {"name": model[0], "parameters": dict(model[1])}}

    member = spec["attacks"][0]
    cls, params = AttackEnum[member].value
    marks = {}

    tail = 2          # batches queued AFTER the closing mark: the interval ends while every stream still has work (a stream that
                      # runs its last batch alone finishes it faster than in steady state)

    def queued(i):
        # called under the stream batch i was queued on.  `timed` is even, so with two batches in flight both marks sit on the
        # same stream and the interval is timed / 2 batches of that stream, during which the other completes as many
        if i in (warm - 1, warm + timed - 1):
            marks[i] = torch.cuda.Event(enable_timing=True)
            marks[i].record(torch.cuda.current_stream(device))

    set_seed(42)
    rep = generate_attacks([None, None, None], cfg(spec["target"]), str(device), attack_model_config=cfg(spec["attacked"]),
                           attack_method=cls, attack_params=params, batch_size=B,
                           dataset=SyntheticDetectionDataset(B * (warm + timed + tail)), share_weights=spec["white_box"], shuffle=False,
                           num_workers=workers, on_batch_queued=queued, in_flight=in_flight)
    torch.cuda.synchronize()
    ms = marks[warm - 1].elapsed_time(marks[warm + timed - 1])
    return {"value": B * timed / (ms * 1e-3), "unit": "utterances/s", "ms_per_batch": ms / timed, "batches_timed": timed,
            "batches_warm": warm, "dataloader_workers": workers, "attack": member, "accuracy": rep["adv_eval/accuracy"],
            "batches_in_flight": in_flight if in_flight is not None else "default (2 for PGD / PGDL2)",
            "what": "evaluation.generate_attacks end to end (collate in worker processes, H2D through pinned buffers on a side "
                    "stream, hipGraph replay, one host sync at the end); HIP-event time from the end of batch `batches_warm` to the end of "
                    "batch `batches_warm + batches_timed`, two more batches queued behind it (steady state on every stream)"}


def live_traffic(entry_point: str, B: int):
    """HBM bytes per launch of the priced kernel MEASURED IN THIS RUN: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE:
    they do not fit one pass; --kernel-trace only, no other trace domain) over tools/traffic_probe.py — the same entry
    point, same batch, operands rotated past the Infinity Cache — reduced with the guide's gfx950 corrections
    (tools/hbm_traffic.py).  Returns (bytes or None, provenance)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, str(ROOT / "tools"))
    import hbm_traffic
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", f"{tmp}/{counter}", "--",
                       sys.executable, str(ROOT / "tools" / "traffic_probe.py"), "--entry", entry_point, "--batch", str(B)]
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=150)
            paths = glob.glob(f"{tmp}/*/*/*counter_collection.csv") + glob.glob(f"{tmp}/*/*counter_collection.csv")
            row = hbm_traffic.reduce(paths, batch_override=B).get(entry_point)
            # the same passes' kernel traces: the priced kernel's own duration by rocprofv3's clock (cold regime)
            import csv
            import re
            pat, durs = hbm_traffic.ENTRY_POINTS[entry_point][0][0], []
            for kt in glob.glob(f"{tmp}/*/*/*kernel_trace.csv") + glob.glob(f"{tmp}/*/*kernel_trace.csv"):
                with open(kt, newline="") as f:
                    durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(f)
                             if re.search(pat, r["Kernel_Name"])]
            if row and durs:
                row["rocprof_avg_launch_ms_cold"] = sum(durs) / len(durs)
    except (subprocess.TimeoutExpired, OSError, SystemExit, KeyError, ValueError) as exc:
        return None, f"live PMC pass failed: {exc!r}"
    if not row:
        return None, "live PMC pass produced no counter rows for this kernel"
    live_traffic.rocprof_ms_cold = row.get("rocprof_avg_launch_ms_cold")
    return row["hbm_bytes_per_launch"], ("measured in this run: " + hbm_traffic.SOURCE + f"; {row['launches_averaged']} launches "
                                         f"of tools/traffic_probe.py --entry {entry_point} --batch {B}")


def measured_traffic(entry_point: str, B: int):
    """Fallback when the live pass is unavailable: HBM bytes per launch from the committed rocprofv3 --pmc passes
    (profiles/hbm_traffic.json, written by tools/hbm_traffic.py); null when no pass exists for this batch size."""
    path = ROOT / "profiles" / "hbm_traffic.json"
    if not path.exists():
        return None, "no PMC pass committed"
    table = json.loads(path.read_text())
    row = table.get("entry_points", {}).get(entry_point)
    if not row or row.get("batch") != B:
        return None, f"no PMC pass for {entry_point} at B={B}"
    return row["hbm_bytes_per_launch"], f"profiles/hbm_traffic.json: {table['source']}"


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.execv(sys.executable, launcher_command(sys.argv[1:], args.gpus))      # never returns

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the attack kernels have no CPU fallback)")
    if args.share_device:
        if args.backend != "gloo":
            raise SystemExit("--share-device needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node exposes {torch.cuda.device_count()} HIP device(s)")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")

    from audio_deepfake_adversarial_attacks_amd import hip_ops, metrics
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import (_Lanes, aggregate_across_ranks, attack_batch, score_batch)

    spec = WORKLOADS[args.config]
    target, attacked = build_models(spec, device)
    attacks = []
    for member in spec["attacks"]:
        cls, params = AttackEnum[member].value
        atk = cls(attacked, **params)
        atk.set_training_mode(model_training=True, batchnorm_training=False)   # evaluate_...:170
        attacks.append((member, atk))
    torch.manual_seed(42 + rank)  # per-rank random starts

    B = args.batch or spec["batch"]
    n_batches = args.warmup + args.steps
    x_all, y_all = synthetic_waveforms(B * n_batches, T, seed=1234 + rank)
    x_all, y_all = x_all.to(device), y_all.to(device)   # inputs resident in HBM before the clock starts

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lanes = _Lanes(device, args.in_flight)          # batch i on stream i % in_flight (1: the current stream, as in rounds 1-5)
    one_lane = _Lanes(device, 1)
    counter = [0]                                    # steps queued so far: consecutive steps alternate streams across calls

    def timed_loop(first, last, lanes=lanes):
        """The loop body for batches [first, last): per attack the scores, labels and a per-step event mark (recorded on the
        stream the step ran on); returns after every stream's work has been ordered in front of the caller's stream."""
        out = {m: ([], [], []) for m, _ in attacks}
        correct = {m: [torch.zeros((), dtype=torch.int64, device=device) for _ in range(lanes.n)] for m, _ in attacks}
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        for i in range(first, last):
            bx, by = x_all[i * B:(i + 1) * B], y_all[i * B:(i + 1) * B]
            with lanes.batch(counter[0], (B, T)) as lane:
                counter[0] += 1
                for m, atk in attacks:
                    adv = attack_batch(atk, bx, by)
                    p, l = score_batch(target, adv)
                    out[m][0].append(p), out[m][1].append(l), out[m][2].append(by)
                    correct[m][lane] += (l == by.int()).sum()
                    lanes.keep(p, l)
                    marks.append(torch.cuda.Event(enable_timing=True))
                    marks[-1].record()
        for m in correct:
            lanes.keep(*correct[m])
        lanes.join()
        return out, {m: torch.stack(v).sum() for m, v in correct.items()}, marks

    def aggregate(out, correct, n_steps):
        total = torch.tensor(B * n_steps, dtype=torch.int64, device=device)
        return {m: aggregate_across_ranks(torch.cat(out[m][0]), torch.cat(out[m][1]), torch.cat(out[m][2]),
                                          correct[m], total) for m, _ in attacks}

    # Priming (untimed, in front of the W warm-up steps): a stream captures its graph at the second sight of the workload, so
    # two steps per stream in flight bring every stream to steady-state replay whatever W is.
    # Warm-up = the timed loop's body, bookkeeping kernels and launch profiling included: the first launch of any kernel
    # loads its code object (tens of milliseconds in the first process on a fresh box), which must not land in a timed
    # step ... and the end-of-run aggregate (RCCL communicator set-up for world > 1, the D2H copy kernels otherwise).
    graph_replayed = all(getattr(atk, "replays_from_graph", False) for _, atk in attacks)
    prime = 2 * lanes.n if graph_replayed else 0
    hip_ops.start_profile(*PROFILED, graph_ok=True)
    for j in range(prime):
        timed_loop(j % n_batches, j % n_batches + 1)
    if args.warmup > 0:
        w_out, w_correct, _ = timed_loop(0, args.warmup)
        aggregate(w_out, w_correct, args.warmup)
        del w_out, w_correct
    elif world > 1:  # communicator warm-up outside the timed region
        z = torch.zeros(B, device=device)
        aggregate_across_ranks(z, z, z, torch.zeros((), device=device), torch.zeros((), device=device))
    hip_ops.stop_profile()
    sync_all()

    # graph_ok: the brackets sit on launches OUTSIDE the captured model part (update step, random start, min-max), so the
    # iteration keeps replaying from its graph — the timed region runs the shipped launch path
    hip_ops.start_profile(*PROFILED, graph_ok=True)
    t0 = time.perf_counter()
    out, correct, marks = timed_loop(args.warmup, n_batches)
    gathered = aggregate(out, correct, args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = hip_ops.stop_profile()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        utterances = B * args.steps * world
        entry, bytes_per_sample, kernel_name = spec["kernel"]
        step_ms = kernel_ms[entry]
        avg_ms = sum(step_ms) / len(step_ms)
        launch_bytes = bytes_per_sample * B * T
        achieved = launch_bytes / (avg_ms * 1e-3) / 1e9
        extras = world == 1 and not args.no_extras
        traffic, provenance = live_traffic(entry, B) if extras else (None, "")
        if traffic is None:
            t2, p2 = measured_traffic(entry, B)
            traffic, provenance = t2, (p2 + (f" (live pass: {provenance})" if provenance else ""))
        # what an event pair measures with NOTHING between its two records (the command processor's own gap): reported beside
        # `avg_launch_ms`, not subtracted from it — it is most of the difference to the kernel trace's average duration
        stream = torch.cuda.current_stream(device)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for a, b in pairs:
            a.record(stream)
            b.record(stream)
        torch.cuda.synchronize()
        empty_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
        # completion time of every (step, attack) after the loop's start mark, differenced in step order (one batch in flight:
        # the step's own duration; two: the interval at which steps complete — consecutive steps run on different streams)
        ends = [marks[0].elapsed_time(m) for m in marks[1:]]
        each = [e - p for p, e in zip([0.0] + ends[:-1], ends)]
        k = len(attacks)
        line = {
            "metric": spec["metric"],
            "value": utterances / elapsed,
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{spec['what']}, batch={B}/GPU, T={T} (BASELINE.json {spec['tag']}); "
                            f"step = minmax -> attack -> revert -> target fwd -> score",
                "global_batch": B * world,
                "batches_in_flight": lanes.n,
                "launch_path": ("hipGraph replay of the attack iteration's model part, update step launched between the replays"
                                if graph_replayed else "eager launches (host-driven attack loop)"),
                "sharding": f"{world} independent contiguous shards, no per-step collective; one "
                            f"{'RCCL' if args.backend == 'nccl' else args.backend} all-reduce + all-gather of the "
                            f"scores after the last step",
                "weights": "seeded random init (set_seed(42))" + (", target == attacked (white-box)"
                                                                    if spec["white_box"] else ", transfer (two models)"),
            },
            "roofline": {
                "kernel": kernel_name,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_provenance": provenance,
                "algorithmic_bytes_per_launch": launch_bytes,
                "avg_launch_ms": avg_ms,
                "launches_timed": len(step_ms),
                "empty_event_pair_ms": empty_ms,
            },
            "attack_kernel_ms_per_step": {n: sum(v) / args.steps for n, v in kernel_ms.items() if v},
            "ms_each_step": [round(sum(each[i * k:(i + 1) * k]), 2) for i in range(args.steps)],
            "adv_eval": {},
        }
        for j, (m, _) in enumerate(attacks):
            all_pred, all_label, all_y, _, _ = gathered[m]
            rep = metrics.adversarial_report(all_y, all_pred, all_label)
            line["adv_eval"][m] = {key.split("/")[1]: round(v, 4) for key, v in rep.items()}
            if k > 1:   # the pair of configs[3], separately (HIP-event time of each attack's share of the steps)
                ms = sum(each[j::k]) / args.steps
                line.setdefault("per_attack", {})[m] = {"ms_per_batch": round(ms, 2),
                                                        "utterances_per_s": round(B / (ms * 1e-3), 1)}
        if k == 1:
            line["adv_eval"] = line["adv_eval"][attacks[0][0]]
        if extras:
            # (1) the same K steps with launch profiling OFF (no event brackets around the update step): what the brackets of the
            #     timed region cost.  Same launch path, same streams, graphs already captured.
            from audio_deepfake_adversarial_attacks_amd.torchattacks import graphed
            sync_all()
            t1 = time.perf_counter()
            o2, c2, _ = timed_loop(args.warmup, n_batches)
            aggregate(o2, c2, args.steps)
            sync_all()
            dt = time.perf_counter() - t1
            line["shipped_path"] = {"launch_path": line["config"]["launch_path"] + ", no event brackets",
                                    "graphs_captured": len(graphed._GRAPHS), "batches_in_flight": lanes.n, "steps": args.steps,
                                    "ms_per_step": 1e3 * dt / args.steps, "value": utterances / dt, "unit": "utterances/s"}
            del o2, c2
            # (1b) ONE batch in flight (rounds 1-5's schedule), same launch path and brackets: the timed region's counterpart
            #      for the in-flight comparison, and the priced kernel's in-loop bracket with the chip to itself
            if lanes.n > 1:
                for j in range(2 if graph_replayed else 0):           # the caller's stream captures its own graph
                    timed_loop(j % n_batches, j % n_batches + 1, one_lane)
                sync_all()
                hip_ops.start_profile(*PROFILED, graph_ok=True)
                t1 = time.perf_counter()
                o2, c2, _ = timed_loop(args.warmup, n_batches, one_lane)
                aggregate(o2, c2, args.steps)
                sync_all()
                dt = time.perf_counter() - t1
                k1 = hip_ops.stop_profile().get(entry) or [float("nan")]
                a1 = sum(k1) / len(k1)
                line["one_in_flight"] = {"ms_per_step": 1e3 * dt / args.steps, "value": utterances / dt, "unit": "utterances/s",
                                         "priced_kernel_avg_launch_ms": a1,
                                         "priced_kernel_frac": launch_bytes / (a1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "what": "the same K steps, same graphs-plus-bracketed-steps launch path, ONE batch in flight "
                                                 "(the schedule of rounds 1-5); with two in flight the priced kernel shares the chip "
                                                 "with the other batch's convolutions, so `roofline.frac` (timed region) reads lower "
                                                 "than this leg's `priced_kernel_frac` for the same kernel"}
                line["roofline"]["avg_launch_ms_one_in_flight"] = a1
                line["roofline"]["frac_one_in_flight"] = line["one_in_flight"]["priced_kernel_frac"]
                del o2, c2
            # (2) model-side roofline: ONE step with HIP events around every launch of the attacked model's hand-written
            #     matrix-core kernels (outside the timed region: ~400 event pairs per step would cost it ~2 %)
            names = MODEL_MATRIX_KERNELS[args.config]
            if names:
                hip_ops.start_profile(*names)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                timed_loop(args.warmup, args.warmup + 1, one_lane)
                e1.record()
                ms, work = hip_ops.stop_profile(with_work=True)
                step_ms_profiled = e0.elapsed_time(e1)
                tot_ms = sum(sum(v) for v in ms.values())
                tot_flop = sum(sum(v) for v in work.values())
                if tot_ms > 0:
                    tflops = tot_flop / (tot_ms * 1e-3) / 1e12
                    line["roofline_model"] = {
                        "kernel": "wino3x3_kernel family (csrc/lcnn_wino.hip: Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32) via "
                                  + ", ".join(n for n in names if ms.get(n)),
                        "bound": "mfma", "dtype": "f32", "achieved": tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / MFMA_F32_PEAK_TFLOPS,
                        "flop_convention": "products the Winograd algorithm must put through the matrix cores (a direct 3x3 "
                                           "convolution's count / 2.25, unpadded channels); direct-convolution-equivalent = "
                                           "achieved x 2.25",
                        "direct_conv_equivalent_tflops": 2.25 * tflops,
                        "kernel_ms_per_step": tot_ms, "share_of_step": tot_ms / step_ms_profiled, "launches_timed":
                        sum(len(v) for v in ms.values()),
                        "per_entry_point": {n: {"launches": len(ms[n]), "ms_per_step": round(sum(ms[n]), 3),
                                                "tflops": round(sum(work[n]) / (sum(ms[n]) * 1e-3) / 1e12, 2)}
                                            for n in names if ms.get(n)},
                    }
            # (2b) the WHOLE step by kernel family: one step with HIP events around every hand-written launch; what is not
            #      bracketed (ATen / rocBLAS / hipFFT launches, gaps between launches) is the last row, by subtraction
            hip_ops.start_profile("*")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            # host-driven workloads (configs[3]) only: the dispatch mode costs host time per ATen call, which an eager, launch-bound
            # step of the graph-replayed workloads would book into the package's own brackets (their library products - the
            # recurrent layers' projections - already sit inside lcnn_ops._gemm's bracket)
            import contextlib
            libmode = library_op_brackets(everything=True) if not graph_replayed else None
            with (libmode if libmode is not None else contextlib.nullcontext()):
                timed_loop(args.warmup, args.warmup + 1, one_lane)
            e1.record()
            ms_all, work_all, bytes_all = hip_ops.stop_profile(with_work="bytes")
            if libmode is not None and libmode.records:          # library products / convolutions OUTSIDE the package's own brackets, as one more family
                lib = [r for r in libmode.records if r[0] != "aten"]
                ms_all["library_matrix_op"] = [a.elapsed_time(b) for _, a, b, _ in lib]
                work_all["library_matrix_op"] = [f for _, _, _, f in lib]
                bytes_all["library_matrix_op"] = [0.0] * len(lib)
                other = [r for r in libmode.records if r[0] == "aten"]
                if other:
                    ms_all["aten_other"] = [a.elapsed_time(b) for _, a, b, _ in other]
                    work_all["aten_other"] = [0.0] * len(other)
                    bytes_all["aten_other"] = [f for _, _, _, f in other]
            step_ms_all = e0.elapsed_time(e1)
            # shares are of a step with ONE batch in flight (the brackets time kernels that have the chip to themselves; with two
            # batches in flight a step completes faster than the sum of its kernels' solo durations)
            base_ms = line.get("one_in_flight", {}).get("ms_per_step") or line["ms_per_step"]
            rows, covered, seen = [], 0.0, set()
            for family, bound, names in STEP_FAMILIES:
                n_launch = sum(len(ms_all.get(n, ())) for n in names)
                if not n_launch:
                    continue
                seen.update(names)
                raw = sum(sum(ms_all.get(n, ())) for n in names)
                # an event pair with nothing between its records reads `empty_ms`: most of what a bracket adds to its kernel
                net = sum(max(v - empty_ms, 0.0) for n in names for v in ms_all.get(n, ()))
                amount = sum(sum((work_all if bound != "hbm" else bytes_all).get(n, ())) for n in names)
                peak = {"mfma": MFMA_F32_PEAK_TFLOPS, "valu": VALU_F32_PEAK_TFLOPS, "hbm": HBM_PEAK_GBS}[bound]
                achieved = amount / (net * 1e-3) / (1e9 if bound == "hbm" else 1e12) if net > 0 else 0.0
                rows.append({"family": family, "bound": bound, "ms_per_step": round(net, 3), "ms_per_step_bracketed": round(raw, 3),
                             "share_of_step": round(net / base_ms, 4), "launches": n_launch, "achieved": round(achieved, 2),
                             "peak": peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(achieved / peak, 4)})
                covered += net
            stray = {n: round(sum(v), 3) for n, v in ms_all.items() if n not in seen and v}
            rows.append({"family": "not bracketed: ATen / hipFFT launches, other hand-written launches, gaps between launches",
                         "bound": None, "ms_per_step": round(base_ms - covered, 3),
                         "share_of_step": round(1.0 - covered / base_ms, 4), "other_hand_written_ms": stray})
            line["roofline_step"] = {
                "step_ms_profiled": round(step_ms_all, 3), "launches_bracketed": sum(len(v) for v in ms_all.values()),
                "share_priced": round(covered / base_ms, 4), "share_base_ms_per_step": round(base_ms, 3),
                "clock": "HIP events on the launch stream around every hand-written launch (and the recurrent layers' library "
                         "GEMMs) of ONE step after the timed region; a row's ms_per_step = sum over its launches of (bracket - "
                         "empty_event_pair_ms), i.e. kernel time; shares are of `share_base_ms_per_step` = a step with ONE batch in flight "
                         "(`one_in_flight.ms_per_step`; the timed region's own ms_per_step when it runs one in flight; the bracketed "
                         "step itself runs `step_ms_profiled`: ~1 200 event pairs cost it ~10 %)",
                "conventions": "mfma: Winograd products through the matrix cores (direct count / 2.25); valu: 2 flop per "
                               "multiply-add of the direct convolution (the first block's input gradient: of the 25 taps per pooled "
                               "output and channel it needs, an eighth of a dense transposed convolution; that kernel is bound by its "
                               "LDS reads, DESIGN.md 4k); hbm: bytes of the operands a launch reads or writes once",
                "rows": rows}
            # (2c) the loop the user runs: generate_attacks over a synthetic dataset with its DataLoader workers, staging stream and
            #      hipGraph replay — HIP events after batch 4 and after the last batch, ONE host synchronisation at the end
            line["through_loop"] = through_loop(args.config, spec, B, device)
            if graph_replayed:
                line["through_loop_one_in_flight"] = through_loop(args.config, spec, B, device, in_flight=1)
            # (3) the priced kernel with its operands rotated past the 256 MiB Infinity Cache (the hot figure above runs on a
            #     132 MB working set that the cache holds): same clock, one event pair per launch
            sys.path.insert(0, str(ROOT / "tools"))
            import traffic_probe
            cold_ms = traffic_probe.cold_launch_ms(entry, B, device)
            line["roofline"]["avg_launch_ms_cold"] = cold_ms
            line["roofline"]["achieved_cold"] = launch_bytes / (cold_ms * 1e-3) / 1e9
            line["roofline"]["frac_cold"] = line["roofline"]["achieved_cold"] / HBM_PEAK_GBS
            rp = getattr(live_traffic, "rocprof_ms_cold", None)
            if rp:
                line["roofline"]["rocprof_avg_launch_ms_cold"] = rp
                line["roofline"]["frac_rocprof_cold"] = launch_bytes / (rp * 1e-3) / 1e9 / HBM_PEAK_GBS
            line["roofline"]["which_is_hbm_honest"] = (
                "frac_cold / frac_rocprof_cold: operands rotated past the 256 MiB Infinity Cache (HIP-event bracket / rocprofv3 "
                "kernel-trace duration of the same launches) - the kernel's HBM figure.  `frac` is the event bracket around the "
                "kernel's launches INSIDE the timed region, as the bench contract asks: with two batches in flight the bracket "
                "opens when the stream's previous graph retires and closes when the kernel has found compute units next to the "
                "other batch's convolutions and finished - it measures the schedule, not the kernel.  `frac_one_in_flight` is the "
                "same bracket in the same loop with one batch in flight (the figure of rounds 1-5: the kernel alone on the chip, "
                "its 132 MB working set partly served by the Infinity Cache, so it is not an HBM figure either)")
        # CW stops early on its own cost (cw.py:107-110): report what it executed, per iteration, and price the CPU leg at the
        # iterations the GPU run executed
        cw_iters = len(kernel_ms.get("cw_adam_step", ())) / args.steps if args.config == 3 else 0.0
        if cw_iters:
            # the metric names what was executed: the enum member is CW with steps = 100, the attack stops on its own cost
            line["metric"] = line["metric"].replace("CW-100", f"CW (steps=100, stops on its own cost after {cw_iters:g})")
            cw = line["per_attack"]["CW"]
            cw["iterations_per_batch"] = cw_iters
            cw["ms_per_iteration"] = round(cw["ms_per_batch"] / cw_iters, 3)
            cw["note"] = (f"cw.py:107-110 stops on its own cost after {cw_iters:g} of 100 iterations on this input; "
                          "ms_per_iteration is the comparable figure")
            if extras:
                # the same attack with the early stop switched off (CW.set_early_stop(False), additive): all 100 iterations
                atk_cw = dict(attacks)["CW"]
                atk_cw.set_early_stop(False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                adv_full = attack_batch(atk_cw, x_all[:B], y_all[:B])
                score_batch(target, adv_full)
                e1.record()
                torch.cuda.synchronize()
                atk_cw.set_early_stop(True)
                full_ms = e0.elapsed_time(e1)
                cw["without_early_stop"] = {"iterations": atk_cw.steps, "ms_per_batch": round(full_ms, 1),
                                            "ms_per_iteration": round(full_ms / atk_cw.steps, 3),
                                            "utterances_per_s": round(B / (full_ms * 1e-3), 2)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_threads, cw_iters)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
