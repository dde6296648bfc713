"""Throughput of gp_ep_lml_rbf_batched (EP LML over a grid of settings), C4-sized problem: n=4096, d=8.
usage: python tools/ep_mesh_perf.py [B] [sweeps] [n] [grad]
env: GPCORE_EP_LOCKSTEP=0 one setting at a time (the serial aggregate), default / 1 the lockstep batch (GPCORE_EP_GROUP slots);
     GPCORE_EP_WORKERS concurrency of the serial path; grad = 1 also times gp_ep_lml_grad_rbf_batched."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import core, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
p = synth.config_c4(n, 8)
ctx = core.Context(0)
th = p["theta"]
thetas = np.stack([th * np.concatenate(([1.0 + 0.05 * b], np.ones(th.size - 2) * (1.0 + 0.03 * b), [1.0])) for b in range(B)])
ctx.ep_lml_rbf_batched(p["X"], p["y"], thetas[:3], stop_eps=-1.0, max_sweeps=2)     # warm-up: contexts, workspaces
t0 = time.perf_counter()
lml, sw, info = ctx.ep_lml_rbf_batched(p["X"], p["y"], thetas, stop_eps=-1.0, max_sweeps=sweeps)
dt = time.perf_counter() - t0
print("n=%d lockstep=%s group=%s workers=%s B=%d sweeps=%d: %.1f ms -> %.2f settings/s, %.1f sweeps/s aggregate; lml[0]=%.9f lml[-1]=%.9f info=%s" %
      (n, os.environ.get("GPCORE_EP_LOCKSTEP", "default"), os.environ.get("GPCORE_EP_GROUP", "default"), os.environ.get("GPCORE_EP_WORKERS", "default"),
       B, sweeps, dt * 1e3, B / dt, B * sweeps / dt, lml[0], lml[-1], sorted(set(info.tolist()))))
if len(sys.argv) > 4 and sys.argv[4] == "1":
    t0 = time.perf_counter()
    lml, grad, sw, info = ctx.ep_lml_grad_rbf_batched(p["X"], p["y"], thetas, stop_eps=-1.0, max_sweeps=sweeps, strict=False)
    dt = time.perf_counter() - t0
    print("   with the gradient (Alg. 5.2): %.1f ms -> %.2f settings/s, %.1f sweeps/s aggregate; |grad[0]|=%.6e" % (dt * 1e3, B / dt, B * sweeps / dt,
                                                                                                               float(np.linalg.norm(grad[0]))))
