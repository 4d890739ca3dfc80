"""ctypes binding of libadvstep.so (include/advstep.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes
from pathlib import Path

import os

# ADVSTEP_LIB: another build of the library (A/B measurements of experimental kernels); the default is the in-tree build,
# which must match the present sources (build key) to load
_LIB_PATH = Path(os.environ.get("ADVSTEP_LIB") or Path(__file__).resolve().parent / "libadvstep.so")
_lib = None

OK, EINVAL, EWORKSPACE, ELAUNCH, ENODEVICE = 0, 1, 2, 3, 4
ABI_VERSION = 3

_p, _i64, _f32, _f64, _u64, _sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                   ctypes.c_uint64, ctypes.c_size_t)

# name -> (restype, argtypes); one entry per symbol declared in include/advstep.h, advstep_lcnn.h, advstep_frontend.h, advstep_fab.h, advstep_dataset.h and advstep_detector.h
SIGNATURES = {
    "advstep_abi_version": (ctypes.c_int, []),
    "advstep_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "advstep_device_count": (ctypes.c_int, []),
    "advstep_row_workspace_bytes": (_sz, [_i64, _i64]),
    "advstep_minmax_normalize_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _sz, _p]),
    "advstep_minmax_revert_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "advstep_fgsm_step_f32": (ctypes.c_int, [_p, _p, _p, _i64, _f32, _f32, _f32, _p]),
    "advstep_pgd_linf_init_noise_f32": (ctypes.c_int, [_p, _p, _p, _i64, _f32, _f32, _p]),
    "advstep_pgd_linf_init_philox_f32": (ctypes.c_int, [_p, _p, _i64, _f32, _f32, _f32, _u64, _u64, _p]),
    "advstep_pgd_linf_step_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _p]),
    "advstep_pgd_l2_init_noise_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _f32, _f32, _f32, _p, _sz, _p]),
    "advstep_pgd_l2_init_philox_f32": (ctypes.c_int, [_p, _p, _i64, _i64, _f32, _f32, _f32, _u64, _u64, _p, _sz, _p]),
    "advstep_pgd_l2_step_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _p, _p,
                                               _p, _sz, _p]),
    "advstep_pgd_l2_repaired_rows": (ctypes.c_int, [_p, _sz, _i64, _i64, _p, _p]),
    "advstep_cw_init_w_f32": (ctypes.c_int, [_p, _p, _i64, _p]),
    "advstep_cw_tanh_sqdist_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _sz, _p]),
    "advstep_cw_adam_step_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _f64, _f64, _f64, _f64, _p]),
    "advstep_cw_best_update_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _p]),
    "advstep_ce2_loss_grad_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _f32, _p]),
    # include/advstep_lcnn.h
    "advstep_mfm_sel_bytes": (_sz, [_i64, _i64, _i64]),
    "advstep_mfm_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "advstep_mfm_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "advstep_mfm_pool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_mfm_pool2_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv5_mfm_pool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv5_mfm_pool2_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv1x1_mfm_supported": (ctypes.c_int, [_i64]),
    "advstep_conv1x1_mfm_sel_bytes": (_sz, [_i64, _i64, _i64]),
    "advstep_conv1x1_mfm_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv1x1_mfm_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lstm_supported": (ctypes.c_int, [_i64]),
    "advstep_lstm_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lstm_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lstm_backward_bcast_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lstm_backward_outer_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lcnn_tail_pack_f32": (ctypes.c_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lcnn_tail_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "advstep_lcnn_tail_unpack_add_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lcnn_tail_unpack_add_outer_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gru_supported": (ctypes.c_int, [_i64]),
    "advstep_gru_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gru_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_supported": (ctypes.c_int, [_i64, _i64]),
    "advstep_conv3x3_prepared_floats": (_sz, [_i64, _i64, ctypes.c_int]),
    "advstep_conv3x3_prepare_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, ctypes.c_int, _p]),
    "advstep_conv3x3_mfm_pool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_mfm_pool2_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_mfm_sel_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "advstep_conv3x3_mfm_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_mfm_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_backward_data_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_afms_row_f32": (ctypes.c_int, [ctypes.c_int, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "advstep_gate_maxpool2_backward_gate_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gate_maxpool2_backward_input_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_weighted_stats_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "advstep_weighted_stats_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "advstep_log_meannorm_max_length": (ctypes.c_int, []),
    "advstep_log_meannorm_forward_f32": (ctypes.c_int, [_p, ctypes.c_float, _p, _i64, _i64, _p]),
    "advstep_log_meannorm_backward_f32": (ctypes.c_int, [_p, _p, ctypes.c_float, _p, _i64, _i64, _p]),
    "advstep_tail_pool1d_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_tail_pool1d_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_res2net_link_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p]),
    "advstep_res2net_link_backward_f32": (ctypes.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "advstep_resconv_supported": (ctypes.c_int, [_i64, _i64, _i64]),
    "advstep_resconv_prepared_floats": (_sz, [_i64, _i64, _i64]),
    "advstep_resconv_prepare_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, ctypes.c_int, _p]),
    "advstep_resconv_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, ctypes.c_float, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_resconv_pooled_grad_f32": (ctypes.c_int, [_p, _p, _p, _p, ctypes.c_float, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_resconv_forward_act_f32": (ctypes.c_int, [_p, _p, _p, _p, ctypes.c_float, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_resconv_pooled_grad_act_f32": (ctypes.c_int, [_p, _p, _p, _p, ctypes.c_float, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_fewin_forward_act_f32": (ctypes.c_int, [_p, _p, _p, ctypes.c_float, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_fewin_supported": (ctypes.c_int, [_i64]),
    "advstep_conv3x3_fewin_forward_f32": (ctypes.c_int, [_p, _p, _p, ctypes.c_float, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_conv3x3_fewout_grad_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_resconv_pool2_forward_few_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_resconv_pool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    # include/advstep_frontend.h
    "advstep_lfcc_block_count": (_sz, [_i64, _i64, _i64]),
    "advstep_lfcc_bands_f32": (ctypes.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lfcc_reduce_max_f32": (ctypes.c_int, [_p, _i64, _p, _p]),
    "advstep_lfcc_project_f32": (ctypes.c_int, [_p, _p, _p, _f32, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lfcc_project_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _f32, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lfcc_floor_fixup_f32": (ctypes.c_int, [_p, _p, _p, _i64, _p]),
    "advstep_lfcc_project_fragment_floats": (ctypes.c_size_t, [_i64, _i64]),
    "advstep_lfcc_project_prepare_f32": (ctypes.c_int, [_p, _i64, _i64, _p, _p]),
    "advstep_lfcc_max_project_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_lfcc_project_backward_zero_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _f32, _p, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "advstep_lfcc_bands_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, ctypes.c_int, _p]),
    "advstep_stft_frames_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_overlap_add_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_bands_block_count": (_sz, [_i64, _i64]),
    "advstep_stft_bands_supported": (ctypes.c_int, [_i64, _i64, _i64]),
    "advstep_stft_bands_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_bands_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_bands_backward_fixup_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _p, ctypes.c_int, _i64, _i64, _i64, _i64,
                                                             _i64, _i64, _p]),
    "advstep_stft_mel_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_mel_backward_from_output_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "advstep_stft_mel_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64,
                                                    _i64, _p]),
    # include/advstep_fab.h
    "advstep_fab_hyperplane_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, ctypes.c_int, _p]),
    "advstep_fab_projection_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, ctypes.c_int, _p]),
    "advstep_fab_combine_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _f32, _p]),
    "advstep_fab_backward_step_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _f32, ctypes.c_int, _p]),
    # include/advstep_dataset.h
    "advstep_wave_pad_tile_f32": (ctypes.c_int, [_p, ctypes.c_int, _p, _p, _p, _p, _i64, _i64, _p]),
    "advstep_qual_select": (ctypes.c_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "advstep_wave_gather_rows_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _p]),
    # include/advstep_detector.h
    "advstep_affine_act_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, ctypes.c_int, _f32, _p]),
    "advstep_affine_act_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, ctypes.c_int, _f32, _p]),
    "advstep_add_maxpool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_maxpool2_backward_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_add_maxpool1d_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_maxpool1d_backward_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gate_fc_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "advstep_gate_fc_backward_f32": (ctypes.c_int, [_p, _i64, _p, _p, _f32, _p, _i64, _i64, _p]),
    "advstep_gate_maxpool2_blocks": (_sz, [_i64, _i64]),
    "advstep_gate_maxpool2_forward_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gate_maxpool2_forward_xw_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gate_maxpool2_backward_gate_pooled_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "advstep_gate_maxpool2_backward_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p]),
}


class AdvstepError(RuntimeError):
    """A libadvstep.so entry point returned a non-zero status."""


def library_path() -> Path:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load libadvstep.so and bind every symbol of the header.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise AdvstepError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`python -m audio_deepfake_adversarial_attacks_amd.build`) from the repository root. "
            "There is no CPU fallback for the attack kernels.")
    from . import build as _build
    if _LIB_PATH == _build.LIB and not _build.is_current():
        raise AdvstepError(
            f"{_LIB_PATH} was not built from the present csrc/*.hip, include/*.h and compiler flags (its build key "
            f"{_build.STAMP.name} is missing or differs): rebuild with `python -m audio_deepfake_adversarial_attacks_amd.build`. "
            "A stale kernel library is never loaded silently.")
    # One HIP runtime per process: PyTorch-ROCm wheels carry their own libamdhip64.so.  If this library were loaded first it
    # would bind /opt/rocm's copy, and kernels launched through that runtime on torch's streams and allocations fail
    # (seen as "HIP kernel launch failed" on the first call).  Importing torch first makes its copy the one both use.
    import torch  # noqa: F401
    lib = ctypes.CDLL(str(_LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.advstep_abi_version()
    if got != ABI_VERSION:
        raise AdvstepError(f"libadvstep.so ABI version {got} != binding version {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != OK:
        msg = load().advstep_status_string(status).decode()
        raise AdvstepError(f"{what}: {msg} (status {status})")
