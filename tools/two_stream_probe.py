#!/usr/bin/env python
"""Batches in flight: the loop body of `generate_attacks` for batch i on stream i % L, each stream
replaying its OWN captured graphs of the one attacked model (torchattacks/graphed.py keys captures by launch stream: a
capture bakes in its static buffers, so one capture cannot serve two batches at once), against the same batches one after
another on one stream.  (The first version of this probe, whose numbers opened the work, used two COPIES of the model.)

Batches are independent (evaluate_models_on_adversarial_attacks.py:211-265; each keeps its own batch-wide dB floor), so
the scores must not change: the adversarial batches of both schedules are compared bit for bit (explicit start noise).

    python tools/two_stream_probe.py [--config 1|2] [--batches 12] [--rounds 3] [--lanes 1,2,3]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--lanes", default="1,2,3")
    ap.add_argument("--offsets", default="0", help="with 2 in flight: stream 1 first runs this many attack iterations on a "
                                                   "throw-away batch, i.e. its batches start that far out of phase (comma list)")
    ap.add_argument("--delays-us", default="", help="with 2 in flight: stream 1 first idles this long (torch.cuda._sleep, "
                                                    "calibrated here), i.e. a phase shift INSIDE the 1.7 ms iteration (comma list)")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import synthetic_waveforms
    from audio_deepfake_adversarial_attacks_amd.evaluation import attack_batch, score_batch
    dev = torch.device("cuda:0")
    spec = bench.WORKLOADS[a.config]
    target, attacked = bench.build_models(spec, dev)
    cls, params = AttackEnum[spec["attacks"][0]].value
    atk = cls(attacked, **params)
    atk.set_training_mode(model_training=True, batchnorm_training=False)
    B = spec["batch"]
    nb = a.batches
    x, y = synthetic_waveforms(B * nb, bench.T, seed=1234)
    x, y = x.to(dev), y.to(dev)
    g = torch.Generator().manual_seed(7)
    if cls.__name__ == "PGDL2":      # pgdl2.py:57-62: a normal draw and a radius per row
        noise = [(torch.randn(B, bench.T, generator=g).to(dev), torch.rand(B, 1, generator=g).to(dev)) for _ in range(nb)]
    else:
        noise = [torch.rand(B, bench.T, generator=g).to(dev) for _ in range(nb)]    # explicit random starts: comparable runs
    from audio_deepfake_adversarial_attacks_amd.evaluation import _Lanes

    shifter = cls(attacked, **params)
    shifter.set_training_mode(model_training=True, batchnorm_training=False)

    # cycles of torch.cuda._sleep per microsecond, measured
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(10_000_000); e1.record()
    torch.cuda.synchronize()
    cycles_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
    print(f"torch.cuda._sleep: {cycles_per_us:.1f} cycles per microsecond", flush=True)

    def run(lanes, offset=0, delay_us=0):
        outs = [None] * nb
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if delay_us and lanes.n > 1:
            with torch.cuda.stream(lanes.streams[1]):
                torch.cuda._sleep(int(delay_us * cycles_per_us))
        if offset and lanes.n > 1:
            shifter.steps = offset
            with torch.cuda.stream(lanes.streams[1]):
                shifter.set_init_noise(noise[0])
                shifter(x[:B].clamp(0, 1), y[:B])
        for i in range(nb):
            with lanes.batch(i, (B, bench.T)):
                atk.set_init_noise(noise[i])
                adv = attack_batch(atk, x[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
                p, _ = score_batch(target, adv)
                lanes.keep(adv, p)
                outs[i] = (adv, p)
        lanes.join()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nb * 1e3, outs

    sets = {L: _Lanes(dev, L) for L in sorted(int(v) for v in a.lanes.split(","))}
    for L, lanes in sets.items():          # every stream captures its graph (second sight) and leaves the serialised phase
        run(lanes)
    ref = None
    offsets = [int(v) for v in a.offsets.split(",")]
    for r in range(a.rounds):
        for L, lanes in sets.items():
            delays = [int(v) for v in a.delays_us.split(",")] if a.delays_us else []
            for off, dly in ([(o, 0) for o in offsets] + [(0, d) for d in delays] if L == 2 else [(0, 0)]):
                ms, outs = run(lanes, off, dly)
                same = ""
                if ref is None:
                    ref = outs
                else:
                    eq = all(torch.equal(o[0], q[0]) and torch.equal(o[1], q[1]) for o, q in zip(outs, ref))
                    worst = max(float((o[0] - q[0]).abs().max()) for o, q in zip(outs, ref))
                    same = f"  adv, scores == first run: {eq} (max |adv - adv_first| {worst:.2e})"
                tag = f"{L} in flight" + (f", stream 1 shifted by {off} iterations (its cost is inside the time)" if off else "") \
                    + (f", stream 1 starts {dly} us late" if dly else "")
                print(f"round {r}  {tag}  {ms:8.2f} ms/batch   {B / ms * 1e3:8.1f} utt/s{same}", flush=True)


if __name__ == "__main__":
    main()
