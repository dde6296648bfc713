"""CPU-side checks of the drop-in boundary: libgpcore.so loads and exports every symbol that
include/gpcore.h declares (no compute calls here -- those need a GPU)."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry
from gp_algos_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "gpcore.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in gpcore.h but not exported: %s" % missing


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_integer_entry_points(lib):
    assert b"gfx950" in lib.gp_version()
    theta = (ctypes.c_double * 5)(1.0, 5.0, 2.0, 3.0, 0.0)  # KernelRequisitesTest.scala:26-34
    out = ctypes.c_double()
    got = []
    for pos in range(1, 6):
        assert lib.gp_hp_get_at_position(theta, 3, pos, ctypes.byref(out)) == _lib.GP_OK
        got.append(out.value)
    assert got == [1.0, 5.0, 2.0, 3.0, 0.0]
    assert lib.gp_hp_get_at_position(theta, 3, 6, ctypes.byref(out)) == _lib.GP_ERANGE
    assert lib.gp_hp_get_at_position(theta, 3, 0, ctypes.byref(out)) == _lib.GP_ERANGE


def test_no_gpu_means_loud_failure(lib):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.gp_ctx_create(0, None, ctypes.byref(h)) != _lib.GP_OK
    from gp_algos_amd.core import Context
    with pytest.raises(_lib.GpCoreError):
        Context(0)
