"""MI355X-native adversarial-attack evaluation path for audio-deepfake detectors.

Drop-in for ONE hot path of piotrkawa/audio-deepfake-adversarial-attacks: the per-batch loop of
``evaluate_models_on_adversarial_attacks.py`` (min-max -> FGSM / PGD / PGDL2 / CW -> revert -> score).

Layout (mirrors the reference's module names for this path)
  csrc/advstep.hip        hand-written gfx950 kernels + the C ABI of include/advstep.h  -> libadvstep.so
  _lib.py / hip_ops.py    ctypes binding of the C ABI; torch tensors in, raw device pointers + HIP stream out
  torchattacks/           Attack base class + FGSM, PGD, PGDL2, CW   (reference: adversarial_attacks/torchattacks)
  aa/                     to_minmax / revert_minmax, AttackEnum        (reference: src/aa)
  frontends.py, models/   LFCC / mel-spec frontends, LCNN / SpecRNet / RawNet3 (PyTorch-ROCm fwd+bwd)
  metrics.py              EER / accuracy / P / R / F1 / AUC            (reference: src/metrics.py + sklearn calls)
  evaluation.py           generate_attacks() loop, synthetic dataset, per-rank sharding + RCCL aggregate

The waveform arithmetic runs ONLY through libadvstep.so on a HIP device: there is no CPU fallback in this
package (the CPU restatement lives in the test-only ``oracle/`` tree at the repository root).
"""

__version__ = "0.1.0"
