"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/multimesh_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from multimesh_amd import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "multimesh_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_and_loader_agree():
    declared = _declared_functions()
    assert sorted(helpers.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    lib = helpers.load_lib()
    assert os.path.basename(lib._filename).startswith("multi_mesh")
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} missing from {lib._filename}"
    assert helpers.load_lib() is lib  # cached handle, like the reference loader


def test_no_silent_cpu_fallback_without_gpu():
    lib = helpers.load_lib()
    if lib.mm_device_count() > 0:
        pytest.skip("a GPU is present")
    from multimesh_amd.device import Context

    with pytest.raises(helpers.MultiMeshHipError):
        Context(0)
    # the legacy symbol reports failure through its return code, never computes on the CPU
    nn = np.zeros((4, 2), np.int64)
    conn = np.arange(8, dtype=np.int64)[None, :].copy()
    enc = np.zeros((4, 8), np.int64)
    w = np.zeros((4, 8))
    nodes = np.random.default_rng(0).uniform(size=(8, 3))
    pts = np.full((4, 3), 0.5)
    rc = lib.triLinearInterpolator(2, 4, nn, conn, enc, nodes, w, pts)
    assert rc < 0 and lib.mm_last_status() < 0
    assert not w.any() and not enc.any()
    with pytest.raises(helpers.MultiMeshHipError):
        helpers.check(rc, "triLinearInterpolator")


def test_argument_validation_needs_no_gpu():
    lib = helpers.load_lib()
    h = C.c_void_p()
    assert lib.mm_context_create(0, None, None) < 0      # null out pointer
    assert lib.mm_gather(None, None, 0, 0, None, None, 0, 8, None, 1) < 0   # null ctx
    assert b"null" in lib.mm_last_error()
