"""Try a two-rank gp_dist group on ONE GPU (expected to be refused by RCCL: two ranks on one device); id shipped through a file.
usage: python tools/dist_two_ranks.py <rank> <world> <idfile>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_algos_amd import synth
from gp_algos_amd.core import Context, DistGroup
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ctx = Context(0)

def exchange(ident):
    if ident is not None:
        open(idfile + ".tmp", "wb").write(ident); os.replace(idfile + ".tmp", idfile)
        return ident
    for _ in range(600):
        if os.path.exists(idfile):
            return open(idfile, "rb").read()
        time.sleep(0.05)
    raise RuntimeError("no id file")

grp = DistGroup(ctx, rank, world, exchange)
p = synth.regression(200, 2, 0, 5, 6, 0, synth.ard_theta(2, 1.0, 1.0, 0.1))
thetas = p["theta"][None, :] * np.linspace(0.8, 1.4, 5)[:, None]
lml, grad, info = grp.lml_grad_batched(p["X"], p["y"], thetas)
ref, _, _ = ctx.lml_grad_batched(p["X"], p["y"], thetas)
print("rank", rank, "shard", grp.shard(5), "match", np.array_equal(lml, ref), lml)
grp.close()
