"""A few EP sweeps on the C4-sized problem (n=4096, d=8) for profiling: python tools/ep_sweeps.py [n] [sweeps]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import core, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = synth.config_c4(n, 8)
ctx = core.Context(0)
K = ctx.gram_rbf(p["X"], p["theta"], full=True)
ep = core.EpClassifierState(ctx, K, p["y"])
ep.sweep(1)
ctx.sync()
t0 = time.perf_counter()
ep.sweep(sweeps)
ctx.sync()
dt = time.perf_counter() - t0
print("n=%d: %d sweeps in %.2f ms -> %.2f ms/sweep, %.1f sweeps/s" % (n, sweeps, dt * 1e3, dt * 1e3 / sweeps, sweeps / dt))
