"""TEST INFRASTRUCTURE — the CPU leg of bench.py: the oracle ("port" of the reference's CPU path, oracle/attacks.py: torch
CPU ops in the reference's order) timed on a bounded sample of a BASELINE.json workload, in its own process so the
host-thread pool is configured before any torch work and a CPU-side problem cannot take the GPU measurement down.

    python -m oracle.cpu_baseline --config 1 --threads 32          # what bench.py runs
    python -m oracle.cpu_baseline --config 1 --sweep 8,16,32,64    # thread sweep at the workload's batch size

SURVEY.md section 8(d) asks for two CPU figures next to the headline: the reference's own CPU-runnable case
(BASELINE.json configs[0]: LCNN + LFCC, FGSM eps = 0.001, B = 8, N = 64 — run in full) and a short configs[1]
(B = 128, PGD-40).  A full PGD-40 batch of 128 is minutes of CPU work, so the sample is ONE batch of the
workload's size run at a reduced iteration count; every iteration is the same work (model forward + input
backward + update) and is timed individually (start of one forward pass of the attacked model to the next), so
t(K) = t_outside_iterations + K * t_iteration is evaluated at the workload's K.  The line says so in `sample`."""
import argparse
import json
import os
import platform
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

T = 64_600
LCNN = ("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1})
SPECRNET = ("specrnet", {"frontend_algorithm": ["mel_spec"], "input_channels": 2})
RAWNET3 = ("rawnet3", {})
# (target, attacked, white_box, batch, [(oracle attack, params, name of the iteration-count parameter or None, iterations sampled)])
WORKLOADS = {
    1: (LCNN, LCNN, True, 128, [("PGD", {"eps": 0.003, "alpha": 2 / 255, "steps": 40, "random_start": True}, "steps", 6)]),
    2: (SPECRNET, SPECRNET, True, 128, [("PGDL2", {"eps": 0.1, "alpha": 0.2, "steps": 40, "random_start": True}, "steps", 6)]),
    3: (LCNN, RAWNET3, False, 64, [("FGSM", {"eps": 0.0005}, None, None),
                                    ("CW", {"c": 1.0, "kappa": 0, "steps": 100, "lr": 0.01}, "steps", 40)]),
}


def cpu_model_name() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def default_threads() -> int:
    """Fastest of the sweep at B = 128 on the GPU box's 2 x EPYC 9575F (profiles/r02_cpu_baseline_threads.jsonl)."""
    return min(os.cpu_count() or 1, 32)


def build(target_spec, attacked_spec, white_box):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    from audio_deepfake_adversarial_attacks_amd.utils import set_seed
    set_seed(42)
    target = get_model(target_spec[0], dict(target_spec[1]), "cpu").eval()
    attacked = get_model(attacked_spec[0], dict(attacked_spec[1]), "cpu").eval()
    if white_box:
        attacked.load_state_dict(target.state_dict())
    return target, attacked


def time_body(target, attacked, name, params, x, y):
    """(seconds, start times of the attacked model's forward passes) of one pass of the loop body.  One forward pass
    starts each attack iteration, so successive start times are one iteration apart (model forward + input backward +
    update step), and their count is the number of iterations that actually ran (CW may stop early, cw.py:107-110)."""
    from oracle import attacks as oracle_attacks
    starts = []
    hook = attacked.register_forward_pre_hook(lambda *_: starts.append(time.perf_counter()))
    t0 = time.perf_counter()
    try:
        oracle_attacks.attack_and_score(target, attacked, name, params, x, y)
    finally:
        hook.remove()
    return time.perf_counter() - t0, starts


def run_workload(config: int, batch: int, iterations: float = 0.0):
    """Seconds of one full batch of the workload (all its attacks) + a description of what was actually run."""
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    target_spec, attacked_spec, white_box, default_batch, attacks = WORKLOADS[config]
    B = batch or default_batch
    target, attacked = build(target_spec, attacked_spec, white_box)
    x, y = synthetic_waveforms(B, T, seed=1234)
    total, notes, ran = 0.0, [], 0.0
    for name, params, count_key, k in attacks:
        warm = dict(params, **({count_key: 1} if count_key else {}))
        time_body(target, attacked, name, warm, x, y)   # untimed: thread pools, oneDNN primitives for these shapes
        if count_key is None:
            t, _ = time_body(target, attacked, name, params, x, y)
            total, ran = total + t, ran + t
            notes.append(f"{name} in full ({t:.1f} s)")
            continue
        # (CW looks at its cost every steps // 10 iterations and may return, cw.py:107-110: 40 steps = a look every 4)
        t, starts = time_body(target, attacked, name, dict(params, **{count_key: k}), x, y)
        gaps = [b - a for a, b in zip(starts[:-1], starts[1:])]
        if len(gaps) < 2:
            raise RuntimeError(f"{name}: only {len(starts)} iteration(s) ran, cannot time an iteration")
        per_iter = sum(gaps) / len(gaps)
        fixed = max(t - len(starts) * per_iter, 0.0)
        K = iterations or params[count_key]
        total += fixed + K * per_iter
        ran += t
        notes.append(f"{name}-{K:g} = {fixed:.2f} s outside the iterations + {K:g} x {per_iter:.3f} s/iteration, from a "
                     f"{name}-{len(starts)} run of the whole batch ({t:.1f} s; iterations are identical work, timed "
                     f"between successive forward passes: {min(gaps):.3f}-{max(gaps):.3f} s)")
    return B, total, ran, "; ".join(notes)


def run_configs0():
    """BASELINE.json configs[0] in full: LCNN + LFCC, FGSM eps = 0.001, batch 8, 64 utterances."""
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    target, attacked = build(LCNN, LCNN, True)
    x, y = synthetic_waveforms(64, T, seed=1234)
    time_body(target, attacked, "FGSM", {"eps": 0.001}, x[:8], y[:8])
    t0 = time.perf_counter()
    for b in range(8):
        time_body(target, attacked, "FGSM", {"eps": 0.001}, x[8 * b:8 * b + 8], y[8 * b:8 * b + 8])
    dt = time.perf_counter() - t0
    return {"value": 64 / dt, "unit": "utterances/s", "sample": f"configs[0] in full: FGSM eps=0.001, 8 batches of 8, {dt:.1f} s"}


def run_full_small(config: int, batch: int = 16):
    """The workload's iterative attack IN FULL (every iteration, nothing scaled) on a small batch — the un-extrapolated
    companion of the one-batch figure (whose batch is the workload's, but whose iteration count is reduced)."""
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    target_spec, attacked_spec, white_box, _, attacks = WORKLOADS[config]
    name, params, count_key, _ = next(a for a in attacks if a[2] is not None)
    target, attacked = build(target_spec, attacked_spec, white_box)
    x, y = synthetic_waveforms(batch, T, seed=1234)
    time_body(target, attacked, name, dict(params, **{count_key: 1}), x, y)
    t, starts = time_body(target, attacked, name, params, x, y)
    return {"value": batch / t, "unit": "utterances/s",
            "sample": f"{name} with {params} in full on {batch} utterances: {len(starts)} iterations + scoring, {t:.1f} s, "
                      "nothing extrapolated"}


def one(config: int, threads: int, batch: int, with_configs0: bool, iterations: float = 0.0):
    import torch
    torch.set_num_threads(threads)
    B, seconds, ran, note = run_workload(config, batch, iterations)
    line = {
        "value": B / seconds, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"one batch of {B} utterances (T={T}) of BASELINE.json configs[{config}] via oracle/attacks.py (torch CPU "
                  f"ops in the reference's order): {note}; {ran:.0f} s of CPU work on {threads} of {os.cpu_count()} host "
                  f"hardware threads",
        "cpu_model": cpu_model_name(),
    }
    if with_configs0:
        line["configs0"] = run_configs0()
    if config in (1, 2) and not batch:
        line["full_attack_small_batch"] = run_full_small(config)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=sorted(WORKLOADS))
    ap.add_argument("--threads", type=int, default=0, help="0 = the swept default")
    ap.add_argument("--batch", type=int, default=0, help="0 = the workload's batch size")
    ap.add_argument("--sweep", default="", help="comma-separated thread counts: print one line per count")
    ap.add_argument("--no-configs0", action="store_true")
    ap.add_argument("--iterations", type=float, default=0.0,
                    help="iterations to price the iterative attack at (0 = its nominal count); bench.py passes the average "
                         "number of CW iterations its GPU run executed per batch, since CW stops early on its own cost")
    a = ap.parse_args()
    os.environ["HIP_VISIBLE_DEVICES"] = ""  # this leg must not touch the GPU
    if a.sweep:
        for n in (int(s) for s in a.sweep.split(",")):
            print(json.dumps(one(a.config, n, a.batch, False)), flush=True)
        return
    threads = a.threads or default_threads()
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    print(json.dumps(one(a.config, threads, a.batch, a.config == 1 and not a.no_configs0, a.iterations)))


if __name__ == "__main__":
    main()
