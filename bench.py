#!/usr/bin/env python
"""Headline benchmark: adversarial utterances / second, PGD-40 (L-inf, eps = 0.003, alpha = 2/255) on
LCNN + LFCC over synthetic 4 s @ 16 kHz utterances (T = 64 600), 128 utterances per GPU (BASELINE.json configs[1];
configs[4] is the same workload on 8 GPUs).

A "step" is one pass of the hot-loop body of evaluate_models_on_adversarial_attacks.py:211-265 over one batch that
is already resident in HBM:  to_minmax -> PGD-40 (40 x [LCNN fwd + input-bwd under PyTorch-ROCm, fused HIP
sign/project/clamp step]) -> revert_minmax -> target-model forward -> sigmoid / threshold.  After the K timed steps
the per-utterance scores are aggregated once (RCCL all-reduce + all-gather when N > 1) inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant hand-written kernel (advstep_pgd_linf_step_f32,
16 algorithmic bytes per waveform sample) with HIP events recorded on its launch stream inside the timed region;
`cpu_baseline` times the CPU oracle ("port": oracle/attacks.py, torch CPU ops in the reference's order) on a bounded
sample of the same workload on this box's host cores (rank 0, N = 1 only)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

T = 64_600
PER_GPU_BATCH = 128
EPS, ALPHA, PGD_STEPS = 0.003, 2 / 255, 40
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEP_BYTES_PER_SAMPLE = 16     # adv + grad + orig read, adv written, f32
LCNN_CONFIG = {"frontend_algorithm": ["lfcc"], "input_channels": 1}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="utterances per GPU per step")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=8, help="utterances in the CPU-baseline sample (one batch)")
    p.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = min(cores, 64))")
    return p.parse_args()


def build_models(device):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    set_seed(42)
    target = get_model("lcnn", dict(LCNN_CONFIG), device).to(device)
    attacked = get_model("lcnn", dict(LCNN_CONFIG), device).to(device)
    attacked.load_state_dict(target.state_dict())  # white-box: same weights
    return target.eval(), attacked.eval()


def cpu_baseline(n_utt: int, threads: int):
    """Oracle ("port") on the host cores: the same per-batch body on a bounded sample of the same workload, run in
    a separate process (oracle/cpu_baseline.py) after the GPU measurement."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--utterances", str(n_utt), "--threads", str(threads)]
    try:
        proc = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
        if proc.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port",
                "sample": f"failed rc={proc.returncode}: {proc.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port", "sample": "timed out (600 s)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the attack kernels have no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=device)  # RCCL over xGMI

    from audio_deepfake_adversarial_attacks_amd import hip_ops, torchattacks
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import (aggregate_across_ranks, attack_batch, score_batch)
    from audio_deepfake_adversarial_attacks_amd import metrics

    target, attacked = build_models(device)
    atk = torchattacks.PGD(attacked, eps=EPS, alpha=ALPHA, steps=PGD_STEPS, random_start=True)
    atk.set_training_mode(model_training=True, batchnorm_training=False)
    torch.manual_seed(42 + rank)  # per-rank random starts

    B = args.batch
    n_batches = args.warmup + args.steps
    x_all, y_all = synthetic_waveforms(B * n_batches, T, seed=1234 + rank)
    x_all, y_all = x_all.to(device), y_all.to(device)   # inputs resident in HBM before the clock starts

    def one_step(i):
        bx, by = x_all[i * B:(i + 1) * B], y_all[i * B:(i + 1) * B]
        adv = attack_batch(atk, bx, by)
        preds, labels = score_batch(target, adv)
        return preds, labels, by

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up = the timed loop's body, bookkeeping kernels and launch profiling included: the first launch of any kernel
    # loads its code object (tens of milliseconds in the first process on a fresh box), which must not land in a timed step.
    profiled = ("pgd_linf_step", "pgd_linf_init", "minmax_normalize", "minmax_revert", "ce2_loss_grad")

    def timed_loop(first, last):
        preds, labels, ys = [], [], []
        correct = torch.zeros((), dtype=torch.int64, device=device)
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        for i in range(first, last):
            p, l, by = one_step(i)
            preds.append(p), labels.append(l), ys.append(by)
            correct += (l == by.int()).sum()
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        return preds, labels, ys, correct, marks

    if args.warmup > 0:
        hip_ops.start_profile(*profiled)
        w = timed_loop(0, args.warmup)
        # ... and the end-of-run aggregate (RCCL communicator set-up for world > 1, the D2H copy kernels otherwise)
        aggregate_across_ranks(torch.cat(w[0]), torch.cat(w[1]), torch.cat(w[2]), w[3],
                               torch.tensor(B * args.warmup, dtype=torch.int64, device=device))
        hip_ops.stop_profile()
        del w
    elif world > 1:  # RCCL communicator warm-up outside the timed region
        aggregate_across_ranks(torch.zeros(B, device=device), torch.zeros(B, device=device),
                               torch.zeros(B, device=device), torch.zeros((), device=device),
                               torch.zeros((), device=device))
    sync_all()

    hip_ops.start_profile(*profiled)
    t0 = time.perf_counter()
    preds, labels, ys, correct, marks = timed_loop(args.warmup, n_batches)
    total = torch.tensor(B * args.steps, dtype=torch.int64, device=device)
    all_pred, all_label, all_y, n_correct, n_total = aggregate_across_ranks(
        torch.cat(preds), torch.cat(labels), torch.cat(ys), correct, total)
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = hip_ops.stop_profile()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        report = metrics.adversarial_report(all_y, all_pred, all_label)
        utterances = B * args.steps * world
        step_ms = kernel_ms["pgd_linf_step"]
        avg_ms = sum(step_ms) / len(step_ms)
        launch_bytes = STEP_BYTES_PER_SAMPLE * B * T
        achieved = launch_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pgd_linf_step_pmc.json"  # per-launch HBM bytes from the rocprofv3 --pmc passes
        if pmc.exists() and B == PER_GPU_BATCH:
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        line = {
            "metric": "adversarial utterances/sec, PGD-40 LCNN+LFCC 4s@16kHz",
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
                "workload": f"LCNN+LFCC, PGD-40 Linf eps=0.003 alpha=2/255 random start, batch={B}/GPU, T={T} "
                            f"(BASELINE.json configs[1]); step = minmax -> attack -> revert -> target fwd -> score",
                "global_batch": B * world,
                "sharding": f"{world} independent contiguous shards, no per-step collective; one RCCL "
                            f"all-reduce + all-gather of the scores after the last step",
                "weights": "seeded random init (set_seed(42)), target == attacked (white-box)",
            },
            "roofline": {
                "kernel": "advstep_pgd_linf_step_f32 (flat_vec_kernel<3, PgdLinfOp>)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": launch_bytes,
                "avg_launch_ms": avg_ms,
                "launches_timed": len(step_ms),
            },
            "attack_kernel_ms_per_step": {k: (sum(v) / args.steps if v else 0.0) for k, v in kernel_ms.items()},
            "ms_each_step": [round(a.elapsed_time(b), 2) for a, b in zip(marks[:-1], marks[1:])],
            "adv_eval": {k.split("/")[1]: round(v, 4) for k, v in report.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)  # fastest of 8..128 on the 256-thread EPYC box (profiles/)
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, threads)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
