"""world_size-2 gloo tests of the N > 1 driver logic (gp_algos_amd/dist.py): sharding of independent units, the
max-over-ranks timing reduce and the one all_gather that assembles batched LML/gradient results.  The per-setting
evaluator is injected; here it is the CPU oracle on a tiny problem (on the GPU box it is Context.lml_grad_batched)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_are_disjoint():
    from gp_algos_amd.dist import owner_of, shard_range
    for total in (0, 1, 7, 64, 65, 1000000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                assert 0 <= lo <= hi <= total
                seen += list(range(lo, hi)) if total <= 1000 else []
                if total and hi > lo:
                    assert owner_of(lo, total, world) == r and owner_of(hi - 1, total, world) == r
            if total <= 1000:
                assert seen == list(range(total))
    # BASELINE config C3: 64 settings over 8 GPUs -> 8 per GPU, index b -> rank b // 8
    assert [shard_range(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from gp_algos_amd import dist, synth
    from oracle import gp_oracle as orc
    dist.init("gloo")
    p = synth.regression(24, 2, 0, 5, 6, 0, synth.ard_theta(2, 1.0, 1.0, 0.2))
    thetas = np.stack([synth.ard_theta(2, sf, s, 0.2) for sf in (0.8, 1.3) for s in (0.7, 1.0, 1.6)][:5])   # B = 5: ragged split 3 + 2
    calls = []

    def evaluate(th):
        calls.append(len(th))
        res = [orc.lml_grad(p["X"], p["y"], t) for t in th]
        return np.array([r[0] for r in res]), np.stack([r[1] for r in res])

    lml, grad = dist.lml_grad_sharded(evaluate, thetas)
    tmax = dist.max_over_ranks(10.0 + rank)
    # one posterior request of 7 test points split 4 + 3 (config C5's sharding): every rank fits the same model
    Lf, alpha = orc.fit(p["X"], p["y"], thetas[0])
    Xs = synth.regression(24, 2, 7, 5, 6, 9, thetas[0])["Xs"]
    pcalls = []

    def predict(xs):
        pcalls.append(len(xs))
        mean, var, _, _ = orc.predict(p["X"], thetas[0], Lf, alpha, np.asfortranarray(xs))
        return mean, var

    pmean, pvar = dist.predict_sharded(predict, Xs)
    dist.barrier()
    out.put((rank, calls, lml, grad, tmax, pcalls, pmean, pvar))
    import torch.distributed as td
    td.destroy_process_group()


def test_lml_grad_sharded_two_ranks_gloo():
    import torch.multiprocessing as mp
    from gp_algos_amd import synth
    from oracle import gp_oracle as orc
    orc.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = synth.regression(24, 2, 0, 5, 6, 0, synth.ard_theta(2, 1.0, 1.0, 0.2))
    thetas = np.stack([synth.ard_theta(2, sf, s, 0.2) for sf in (0.8, 1.3) for s in (0.7, 1.0, 1.6)][:5])
    ref = [orc.lml_grad(p["X"], p["y"], t) for t in thetas]
    assert res[0][1] == [3] and res[1][1] == [2]            # each rank evaluated only its own settings
    Lf, alpha = orc.fit(p["X"], p["y"], thetas[0])
    Xs = synth.regression(24, 2, 7, 5, 6, 9, thetas[0])["Xs"]
    rmean, rvar, _, _ = orc.predict(p["X"], thetas[0], Lf, alpha, Xs)
    assert res[0][5] == [4] and res[1][5] == [3]            # the 7 test points were split 4 + 3, each rank predicted only its slice
    for rank, calls, lml, grad, tmax, pcalls, pmean, pvar in res:
        assert tmax == 11.0                                   # max over ranks
        np.testing.assert_array_equal(lml, [r[0] for r in ref])    # assembled in setting order on every rank
        np.testing.assert_array_equal(grad, np.stack([r[1] for r in ref]))
        np.testing.assert_array_equal(pmean, rmean)          # assembled in test-point order on every rank
        np.testing.assert_array_equal(pvar, rvar)


def _failing_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from gp_algos_amd import dist
    from gp_algos_amd import _lib
    from gp_algos_amd._lib import GpCoreError
    dist.init("gloo")
    thetas = np.arange(15.0).reshape(5, 3)

    def evaluate(th):
        if rank == 1:
            raise GpCoreError(3, "injected: hipMalloc failed on this rank")       # GP_ENOMEM on ONE rank only
        return th[:, 0], th

    got = None
    try:
        dist.lml_grad_sharded(evaluate, thetas)
    except _lib.PeerFailure as e:             # ONE exception class for a failed peer, whichever exchange reported it (dist.PeerFailure
        assert isinstance(e, dist.PeerFailure) and e.status == _lib.GP_EPEER   # is a subclass of the C-ABI's _lib.PeerFailure)
        got = ("peer", e.bad_rank, e.peer_status)
    except GpCoreError as e:
        got = ("own", rank, e.status)
    # the group is still usable afterwards: nobody is stuck inside a collective, nothing is half-exchanged
    lml, grad = dist.lml_grad_sharded(lambda th: (th[:, 0], th), thetas)
    dist.barrier()
    out.put((rank, got, lml.tolist()))
    import torch.distributed as td
    td.destroy_process_group()


def test_failing_rank_does_not_strand_its_peer_gloo():
    """a rank whose local evaluation fails must not leave the others blocked inside the all_gather.
    dist.agree exchanges one status word per rank first (the torch.distributed form of gpcore_dist.hip dist_agree)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert res[0][1] == ("peer", 1, 3)        # rank 0 was fine and is told who failed and how
    assert res[1][1] == ("own", 1, 3)         # rank 1 sees its own error
    assert res[0][2] == res[1][2] == [0.0, 3.0, 6.0, 9.0, 12.0]


def test_status_scan_rule_c_abi_and_python_agree():
    """gp_dist_status_scan (host only: no device, no communicator) and dist.status_scan implement the same agreement rule."""
    import ctypes as C
    import __graft_entry__ as entry
    from gp_algos_amd import _lib, dist
    entry.build()
    lib = _lib.load()
    cases = [([0, 0, 0, 0], 2, (0, -1)), ([0, 3, 0, 4], 0, (_lib.GP_EPEER, 1)), ([0, 3, 0, 4], 1, (3, 1)),
             ([0, 3, 0, 4], 3, (4, 3)), ([2], 0, (2, 0)), ([0], 0, (0, -1))]
    for status, rank, want in cases:
        arr = (C.c_double * len(status))(*[float(s) for s in status])
        bad = C.c_int(-7)
        st = lib.gp_dist_status_scan(arr, len(status), rank, C.byref(bad))
        assert (st, bad.value) == want, (status, rank)
        assert dist.status_scan(status, rank) == want
    assert lib.gp_dist_status_scan(None, 2, 0, None) == _lib.GP_EINVAL
    arr = (C.c_double * 2)(0.0, 0.0)
    assert lib.gp_dist_status_scan(arr, 2, 2, None) == _lib.GP_EINVAL
    assert lib.gp_dist_status_scan(arr, 2, 1, None) == _lib.GP_OK      # bad_rank may be NULL
