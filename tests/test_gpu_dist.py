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
