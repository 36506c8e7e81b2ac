"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol include/visrep.h declares."""
import ctypes
import os
import re

import pytest

from law_of_vision_representation_in_mllms_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "visrep.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(visrep_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 14
    assert sorted(_lib.SIGNATURES) == syms


def test_library_builds_loads_and_exports_everything():
    build.build_lib()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert _lib.load().visrep_version() == 200


def test_error_reporting_without_gpu_work():
    lib = _lib.load()
    # argument validation happens before any device work: a null pointer must produce an error code + message
    rc = lib.visrep_gemm_bf16(None, 0, None, 0, None, None, 0, 1, 128, 64, 0, 0, None, None, None)
    assert rc == -1
    assert "null" in _lib.last_error()
    assert lib.visrep_ascore_workspace_bytes(2, 10, 7) == 4 * (20 + 14 + 80)        # row scales of both operands + [n, Nt, 4 * ceil(Nr / 192)] row maxima


def test_compute_paths_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from law_of_vision_representation_in_mllms_amd import ascore_ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ascore_ops.max_cos_mean(torch.zeros(1, 4, 16), torch.zeros(1, 4, 16))


def test_graft_entry_build_runs_without_gpu():
    """The driver's "does it build" hook: compiles (or finds up to date) the gfx950 library, loads it, checks the ABI version."""
    import __graft_entry__ as g
    g.build()
