"""Build libvisrep_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.  In-tree output so the
built library travels with the repo snapshot to the GPU box.  `python -m law_of_vision_representation_in_mllms_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvisrep_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_bf16_v2.hip", "gemm_bf16_v3.hip", "attention.hip", "rowops.hip", "convnet.hip", "ascore.hip", "cscore.hip", "visrep_abi.hip"]
HEADERS = ["common.h", "gemm_epilogue.h", "visrep_internal.h", os.path.join("..", "..", "include", "visrep.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_ablation_lib() -> str:
    """Separate library with the GEMM-v2 timing ablation knobs compiled in (tools/gemm_ablate.py only)."""
    out = os.path.join(PKG, "libvisrep_hip_ablate.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-DVISREP_GEMM_ABLATE"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def build_attn_ablation_lib(mask: int) -> str:
    """Library with -DVISREP_ATTN_ABLATE=mask (tools/attn_ablate.py only; results are wrong for mask != 0)."""
    out = os.path.join(PKG, f"libvisrep_hip_attn{mask}.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + [f"-DVISREP_ATTN_ABLATE={mask}"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvisrep_hip.so")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvisrep_hip.so")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
