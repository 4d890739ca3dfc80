"""Per-utterance min-max normalisation around an attack (reference: src/aa/utils.py:4-14).

Same names, arguments and return values as the reference; the arithmetic runs in the HIP kernels
(`advstep_minmax_normalize_f32` / `advstep_minmax_revert_f32`)."""
from .. import hip_ops


def to_minmax(batch_x):
    """(B, T) -> ((x - mn) / (mx - mn), mn (B,1), mx (B,1)); a constant row yields NaN, as in the reference."""
    return hip_ops.to_minmax(batch_x.contiguous())


def revert_minmax(batch_x, mn, mx):
    """(x * (mx - mn)) + mn."""
    return hip_ops.revert_minmax(batch_x.contiguous(), mn.contiguous(), mx.contiguous())
