"""Compile csrc/*.hip into libadvstep.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SOURCES = [PKG / "csrc" / "advstep.hip", PKG / "csrc" / "lcnn_mfm.hip", PKG / "csrc" / "lcnn_conv0.hip",
           PKG / "csrc" / "lcnn_conv1x1.hip", PKG / "csrc" / "lcnn_lstm.hip", PKG / "csrc" / "lcnn_wino.hip",
           PKG / "csrc" / "lfcc.hip", PKG / "csrc" / "lfcc_stft.hip", PKG / "csrc" / "fab.hip",
           PKG / "csrc" / "wave_prep.hip", PKG / "csrc" / "specrnet_gru.hip"]
HEADERS = [ROOT / "include" / "advstep.h", ROOT / "include" / "advstep_lcnn.h", ROOT / "include" / "advstep_frontend.h",
           ROOT / "include" / "advstep_fab.h", ROOT / "include" / "advstep_dataset.h"]
LIB = PKG / "libadvstep.so"

# -ffp-contract=off: the kernels must round exactly like the reference's one-ATen-op-per-expression chains
# (bit-exactness rules); f32 division/sqrt stay IEEE (hipcc default).
# -fno-slp-vectorize: keeps the first-block convolution on v_fmac_f32 with scalar (SGPR) weight operands instead of
# v_pk_fma_f32 + per-operand v_mov broadcasts (3x the VALU instructions); the streaming kernels do not care.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
               "-shared"]


def build(force: bool = False, verbose: bool = False) -> Path:
    newest_src = max(p.stat().st_mtime for p in SOURCES + HEADERS)
    if not force and LIB.exists() and LIB.stat().st_mtime >= newest_src:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, f"-I{ROOT / 'include'}", *map(str, SOURCES), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
