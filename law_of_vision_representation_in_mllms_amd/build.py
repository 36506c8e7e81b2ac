"""Build libvisrep_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.  In-tree output so the
built library travels with the repo snapshot to the GPU box.  `python -m law_of_vision_representation_in_mllms_amd.build`.

One object per translation unit (compiled in parallel, cached under csrc/.obj by a content hash of the source, the headers and
the flags), then one link.  Staleness of the library is decided by the same content hash (libvisrep_hip.srchash beside the
library), not by mtimes - a snapshot copy of the tree (gpurun) does not have to preserve them.
Concurrent callers - every rank of a `torch.distributed.run` launch reaches `_lib.load()` at the same time - are serialised
with an exclusive file lock, every temporary carries the pid, and the library is moved into place with one atomic rename: a
rank either loads the previous complete library or the new complete one, never a half-written file.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(PKG, "libvisrep_hip.so")
# the product library: GEMM v1 (128x128: tails, split-K, N % 256 != 0, implicit convolution), v2 (K % 64 != 0), v5 (default), attn_fwd
SOURCES = ["gemm_bf16.hip", "gemm_bf16_v2.hip", "gemm_bf16_v5.hip", "gemm_bf16_duo.hip", "attention.hip", "rowops.hip", "convnet.hip", "conv_halo.hip", "conv_in.hip", "ascore.hip", "ascore_ref.hip",
           "cscore.hip", "f32ops.hip", "jpeg_decode.hip", "host_twins.hip", "visrep_abi.hip"]
# measured dead ends (GEMM v3 / v4, attn_fwd_ab): since round 5 they live as a patch (tools/experiments/dead_end_kernels_r2_r3.patch adds the three
# files back); with it applied, build_experiments_lib() compiles them into the tools-only libvisrep_hip_exp.so (-DVISREP_EXPERIMENTS) - never
# into what ships
EXPERIMENT_SOURCES = ["gemm_bf16_v3.hip", "gemm_bf16_v4.hip", "attention_ab.hip"]


def _need_experiment_sources():
    missing = [f for f in EXPERIMENT_SOURCES if not os.path.exists(os.path.join(CSRC, f))]
    if missing:
        raise RuntimeError(f"experiment sources {missing} are not in the tree: `git apply tools/experiments/dead_end_kernels_r2_r3.patch` first")
HEADERS = ["common.h", "gemm_epilogue.h", "visrep_internal.h", os.path.join("..", "..", "include", "visrep.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
# Per-file flags.  attention_ab.hip: -O3's SLP vectoriser packs the softmax's adjacent f32 multiplies / adds into v_pk_*_f32, which
# cost more than the two plain VALU ops they replace when they sit beside MFMAs (MI355X guide, per-instruction constants).
FILE_FLAGS = {"attention_ab.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvisrep_hip.so")
    return hipcc


def _hipcc_or_none():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return hipcc if os.path.exists(hipcc) else None


def _digest(files, extra=()) -> str:
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(list(CFLAGS) + list(extra) + [f"{k}:{' '.join(v)}" for k, v in sorted(FILE_FLAGS.items()) if k in files]).encode())
    return h.hexdigest()


def have_sources() -> bool:
    return all(os.path.exists(os.path.join(CSRC, f)) for f in SOURCES + HEADERS)


def source_hash(defines=(), sources=None) -> str:
    return _digest(list(sources or SOURCES) + HEADERS, defines)


def _hash_file(out: str) -> str:
    return os.path.splitext(out)[0] + ".srchash"


def _stale(out: str = LIB, defines=(), sources=None) -> bool:
    if not os.path.exists(out) or not os.path.exists(_hash_file(out)):
        return True
    with open(_hash_file(out)) as fh:
        return fh.read().strip() != source_hash(defines, sources)


def _compile(src: str, objdir: str, defines, verbose: bool) -> str:
    """src -> objdir/src.<hash>.o unless that exact (source, headers, flags) combination was compiled before."""
    path = os.path.join(CSRC, src)
    defines = list(defines) + FILE_FLAGS.get(src, [])
    key = _digest([src] + HEADERS, defines)[:16]
    obj = os.path.join(objdir, f"{src}.{key}.o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(objdir):                         # drop this unit's superseded objects
        if old.startswith(src + ".") and old.endswith(".o"):
            os.remove(os.path.join(objdir, old))
    tmp = f"{obj}.{os.getpid()}.tmp"
    cmd = [_hipcc()] + CFLAGS + list(defines) + ["-c", path, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    os.replace(tmp, obj)
    return obj


def _build(out: str, defines=(), tag: str = "", verbose: bool = False, only=None, sources=None) -> str:
    """only: the translation units the extra defines apply to (diagnostic builds); every other unit is the default build's object.
    sources: the translation units to link (default: the product library's)."""
    sources = list(sources or SOURCES)
    objdir = os.path.join(OBJ, tag) if tag else OBJ
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)                  # one builder at a time (per object directory), across processes
        try:
            def one(src):
                if only is not None and src not in only:
                    return _compile(src, OBJ, (), verbose)
                return _compile(src, objdir, defines, verbose)
            with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(one, sources))
            if not _stale(out, defines, sources):
                return out                                # another process linked it while this one waited for the lock
            tmp = f"{out}.{os.getpid()}.tmp"
            cmd = [_hipcc()] + LDFLAGS + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed linking {os.path.basename(out)}")
            os.replace(tmp, out)
            with open(f"{_hash_file(out)}.{os.getpid()}.tmp", "w") as fh:
                fh.write(source_hash(defines, sources))
            os.replace(f"{_hash_file(out)}.{os.getpid()}.tmp", _hash_file(out))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def build_ablation_lib() -> str:
    """Separate library with the GEMM-v2 timing ablation knobs compiled in (tools/gemm_ablate.py only)."""
    return _build(os.path.join(PKG, "libvisrep_hip_ablate.so"), ["-DVISREP_GEMM_ABLATE"], "ablate")


def build_attn_ablation_lib(mask: int) -> str:
    """Library with -DVISREP_ATTN_ABLATE=mask (tools/attn_ablate.py only; results are wrong for mask != 0)."""
    return _build(os.path.join(PKG, f"libvisrep_hip_attn{mask}.so"), [f"-DVISREP_ATTN_ABLATE={mask}"], f"attn{mask}")


def build_variant_lib(name: str, defines, only=None, experiments: bool = False) -> str:
    """Diagnostic build with extra -D flags (tools/ only): libvisrep_hip_<name>.so, loaded through VISREP_LIB."""
    if experiments:
        _need_experiment_sources()
        return _build(os.path.join(PKG, f"libvisrep_hip_{name}.so"), ["-DVISREP_EXPERIMENTS"] + list(defines), name, sources=SOURCES + EXPERIMENT_SOURCES)
    return _build(os.path.join(PKG, f"libvisrep_hip_{name}.so"), list(defines), name, only=only)


def build_experiments_lib() -> str:
    """libvisrep_hip_exp.so = the product sources + GEMM v3 / v4 + attn_fwd_ab with -DVISREP_EXPERIMENTS (tools/ only, loaded through
    VISREP_LIB): the variants the round-2 / round-3 profiles measured and rejected stay reproducible without shipping."""
    _need_experiment_sources()
    return _build(os.path.join(PKG, "libvisrep_hip_exp.so"), ["-DVISREP_EXPERIMENTS"], "exp", sources=SOURCES + EXPERIMENT_SOURCES)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    if force:
        for f in (os.listdir(OBJ) if os.path.isdir(OBJ) else []):
            if f.endswith(".o"):
                os.remove(os.path.join(OBJ, f))
        if os.path.exists(_hash_file(LIB)):
            os.remove(_hash_file(LIB))
    return _build(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
