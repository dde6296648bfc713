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


def test_single_launch_cholesky_plan_is_a_valid_schedule():
    """gp_chol_plan_info (host only): the task list of the single-launch factorisation for several shapes -- every block factored once,
    every row block solved once after its updates, every tile updated by every panel in order, every dependency pointing backwards --
    and the makespan of its schedule under the cost model (the figure DESIGN.md section 7 compares the measured fit with)."""
    import ctypes as C
    from gp_algos_amd import _lib as L
    lib = L.load()
    nt, us = C.c_int(), C.c_double()
    seen = {}
    for n, extra, wg in ((2048, 128, 256), (2560, 0, 256), (4096, 128, 256), (8192, 128, 256), (8192, 0, 64), (3200, 128, 7)):
        assert lib.gp_chol_plan_info(n, extra, wg, C.byref(nt), C.byref(us)) == L.GP_OK, (n, extra, wg)
        nb, nrow = n // 128, (n + extra) // 128
        assert nt.value >= nb + sum(nrow - k - 1 for k in range(nb)) and us.value > 0.0
        seen[(n, extra, wg)] = (nt.value, us.value)
    # every shape the library may plan: all row-block counts from one outer panel to past the C2 size's, ragged last panels, with and
    # without the extra row block, few and many workgroups (the replay inside gp_chol_plan_info is the check)
    for nb in list(range(1, 42)) + [47, 53, 64, 65, 79, 96, 112]:
        for extra in (0, 128):
            for wg in (256, 3) if nb in (5, 17, 41, 64) else (256,):
                assert lib.gp_chol_plan_info(128 * nb, extra, wg, C.byref(nt), C.byref(us)) == L.GP_OK, (nb, extra, wg)
                assert nt.value > 0 and us.value > 0.0
    assert seen[(8192, 0, 64)][1] > seen[(8192, 128, 256)][1]                    # a quarter of the workgroups: a longer schedule
    assert 3000.0 < seen[(8192, 128, 256)][1] < 5000.0                           # the model's n = 8192: about 4 ms on 256 CUs
    for bad in ((100, 0, 256), (8192, 64, 256), (8192, 256, 256), (8192, 0, 0)):
        assert lib.gp_chol_plan_info(*bad, C.byref(nt), C.byref(us)) == L.GP_EINVAL
    assert lib.gp_chol_plan_info(8192, 128, 256, None, None) == L.GP_EINVAL
