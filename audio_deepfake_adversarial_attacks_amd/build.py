"""Compile csrc/*.hip into libadvstep.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SOURCES = [PKG / "csrc" / "advstep.hip", PKG / "csrc" / "lcnn_mfm.hip", PKG / "csrc" / "lcnn_conv0.hip",
           PKG / "csrc" / "lcnn_conv1x1.hip", PKG / "csrc" / "lcnn_lstm.hip", PKG / "csrc" / "lcnn_wino.hip",
           PKG / "csrc" / "lfcc.hip", PKG / "csrc" / "lfcc_stft.hip", PKG / "csrc" / "fab.hip",
           PKG / "csrc" / "wave_prep.hip", PKG / "csrc" / "specrnet_gru.hip", PKG / "csrc" / "detector_elem.hip",
           PKG / "csrc" / "detector_conv.hip"]
HEADERS = [PKG / "csrc" / "stft_tables.inc",       # generated twiddle constants (tools/gen_stft_tables.py), #included by lfcc_stft.hip
           ROOT / "include" / "advstep.h", ROOT / "include" / "advstep_lcnn.h", ROOT / "include" / "advstep_frontend.h",
           ROOT / "include" / "advstep_fab.h", ROOT / "include" / "advstep_dataset.h", ROOT / "include" / "advstep_detector.h"]
LIB = PKG / "libadvstep.so"
STAMP = PKG / "libadvstep.so.buildkey"   # git-ignored like the library; travels with it to the GPU box

# -ffp-contract=off: the kernels must round exactly like the reference's one-ATen-op-per-expression chains
# (bit-exactness rules); f32 division/sqrt stay IEEE (hipcc default).
# -fno-slp-vectorize: keeps the first-block convolution on v_fmac_f32 with scalar (SGPR) weight operands instead of
# v_pk_fma_f32 + per-operand v_mov broadcasts (3x the VALU instructions); the streaming kernels do not care.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
               "-shared"]


def _flags() -> list:
    """HIPCC_FLAGS + ADVSTEP_EXTRA_HIPCC_FLAGS (experiments: -DWINO_PREFETCH=2 ...); part of the build key."""
    import os
    return HIPCC_FLAGS + os.environ.get("ADVSTEP_EXTRA_HIPCC_FLAGS", "").split()


def build_key() -> str:
    """SHA-256 over the compiler flags and the bytes of every source and header: what the library was built from."""
    h = hashlib.sha256(" ".join(_flags()).encode())
    for p in SOURCES + HEADERS:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def is_current() -> bool:
    """True when libadvstep.so exists and was built from exactly the present sources, headers and flags (a library that
    is merely newer than its sources — a stale copy, a checkout that rewound a file — does not count)."""
    return LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == build_key()


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and is_current():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *_flags(), f"-I{ROOT / 'include'}", *map(str, SOURCES), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    STAMP.unlink(missing_ok=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    STAMP.write_text(build_key() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
