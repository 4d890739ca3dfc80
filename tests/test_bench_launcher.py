"""bench.py's launcher logic (no GPU needed): `--gpus N` without a launcher re-executes under torch.distributed.run."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_launcher_command_is_the_drivers_line():
    import bench
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "5", "--warmup", "2"], 8, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7:] == [str(ROOT / "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_defaults_and_workload_table():
    import bench
    a = bench.parse([])
    assert (a.gpus, a.config, a.steps, a.warmup, a.backend) == (1, 1, 4, 2, "nccl")
    assert bench.parse(["--config", "3"]).steps == 2
    assert bench.WORKLOADS[1]["metric"] == "adversarial utterances/sec, PGD-40 LCNN+LFCC 4s@16kHz"   # BASELINE.json
    from audio_deepfake_adversarial_attacks_amd.aa.aa_types import AttackEnum
    for spec in bench.WORKLOADS.values():
        for member in spec["attacks"]:
            assert AttackEnum[member].value[0] is not None
    assert AttackEnum.PGD40_eps003.value[1] == {"eps": 0.003, "steps": 40}


@pytest.mark.timeout(300)
def test_gpus_2_without_launcher_starts_two_ranks(tmp_path):
    """On this GPU-less box both ranks must come up (so the re-exec happened) and stop loudly: no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered with a device by tests/test_gpu_distributed.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=280)
    assert proc.returncode != 0
    assert proc.stderr.count("bench.py needs a HIP device") >= 2, proc.stderr[-1500:]
