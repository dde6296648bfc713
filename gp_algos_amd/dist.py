"""Multi-GPU driver logic: one process per GPU, the path shards over INDEPENDENT units (test points, hyper-parameter
settings), no data-path collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) is used for rendezvous, barriers, a max-reduce of timings and one small all_gather of results.

Partitioning follows SURVEY.md 8(e): setting index b -> rank b // ceil(B / G); test points sliced contiguously."""
import os

import numpy as np

from . import _lib


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one chunk at the tail."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def owner_of(index, total, world):
    return index // (-(-total // world))


def init(backend, device=None):
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class PeerFailure(_lib.PeerFailure):
    """Another rank's local evaluation failed; nothing was exchanged (the torch.distributed form of GP_EPEER).  A subclass of the
    exception the C-ABI path raises for GP_EPEER (`_lib.PeerFailure`, itself a `GpCoreError`), so `except _lib.PeerFailure` catches
    a peer failure whichever of the two exchanges reported it."""

    def __init__(self, bad_rank, status):
        _lib.PeerFailure.__init__(self, _lib.GP_EPEER, "rank %d of the group failed with status %d; no result was exchanged" % (bad_rank, status))
        self.bad_rank, self.peer_status = bad_rank, status


def status_scan(status, rank):
    """gp_dist_status_scan's rule on a vector of per-rank status words: (0, -1) if every rank is fine, (own status, rank)
    if this rank failed, else (GP_EPEER = 7, first failing rank)."""
    status = [int(s) for s in status]
    bad = next((r for r, s in enumerate(status) if s != 0), -1)
    if bad < 0:
        return 0, -1
    if status[rank] != 0:
        return status[rank], rank
    return 7, bad


def agree(local_exc, device="cpu"):
    """Status exchange BEFORE a result collective (mirrors gpcore_dist.hip dist_agree): every rank contributes one word --
    0, or the gpcore status / 1 of the exception its local part raised -- so a rank that failed locally never leaves its peers
    blocked in the result all_gather.  The failing rank re-raises its own exception, the others raise PeerFailure."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if local_exc is not None:
            raise local_exc
        return
    code = 0 if local_exc is None else int(getattr(local_exc, "status", 1) or 1)
    mine = torch.tensor([float(code)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    st, bad = status_scan([float(p.item()) for p in parts], dist.get_rank())
    if st == 0:
        return
    if local_exc is not None:
        raise local_exc
    raise PeerFailure(bad, int(parts[bad].item()))


def all_gather_rows(local, total_rows, device="cpu"):
    """Assemble per-rank result rows (each rank holds the rows of shard_range(total_rows, rank, world)) on every rank.
    Message size is tiny (B x (1+P) doubles), one collective."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(np.atleast_2d(local), dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    per = -(-total_rows // world)
    pad = np.full((per, local.shape[1]), np.nan)
    pad[:local.shape[0]] = local
    mine = torch.from_numpy(pad).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = np.concatenate([p.cpu().numpy() for p in parts], axis=0)
    return full[:total_rows]


def lml_grad_sharded(evaluate, thetas, device="cpu"):
    """Batched LML + gradient over B settings, sharded setting-wise.  `evaluate(thetas_slice) -> (lml[b], grad[b, P])`
    runs on this rank's GPU (Context.lml_grad_batched); returns the assembled (lml[B], grad[B, P]) on every rank."""
    rank, _, world = env_rank_world()
    thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
    B = thetas.shape[0]
    lo, hi = shard_range(B, rank, world)
    local, failed = np.zeros((0, 1 + thetas.shape[1])), None
    if hi > lo:
        try:
            lml, grad = evaluate(thetas[lo:hi])
            local = np.concatenate([np.asarray(lml).reshape(-1, 1), np.atleast_2d(grad)], axis=1)
        except Exception as e:        # noqa: BLE001 -- whatever failed here, the peers must hear of it before they gather
            failed = e
    agree(failed, device=device)
    full = all_gather_rows(local, B, device=device)
    return full[:, 0].copy(), full[:, 1:].copy()


def predict_sharded(predict, Xs, device="cpu"):
    """ONE posterior request of m test points split over the ranks: rank r takes the contiguous rows
    shard_range(m, r, world) of Xs, `predict(Xs_slice) -> (mean, var)` runs on its GPU (RegressionModel.predict /
    gp_predict_dev against the factor that rank holds), and one all_gather of 2 m/G doubles per rank assembles
    (mean[m], var[m]) on every rank.  No other collective: test points are independent units."""
    rank, _, world = env_rank_world()
    Xs = np.asarray(Xs)
    m = Xs.shape[0]
    lo, hi = shard_range(m, rank, world)
    local, failed = np.zeros((0, 2)), None
    if hi > lo:
        try:
            mean, var = predict(Xs[lo:hi])
            local = np.stack([np.asarray(mean, dtype=np.float64), np.asarray(var, dtype=np.float64)], axis=1)
        except Exception as e:        # noqa: BLE001
            failed = e
    agree(failed, device=device)
    full = all_gather_rows(local, m, device=device)
    return full[:, 0].copy(), full[:, 1].copy()
