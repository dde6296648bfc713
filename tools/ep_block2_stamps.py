"""Phase stamps of ep_block2_kernel (the fused EP chain kernel) for block 5 of the last sweep: needs a lab library built with -DEP_STAMPS,
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEP_STAMPS [-DEP_LIB_ERF] -o tools/lab/libgpcore_stamps.so gp_algos_amd/csrc/*.hip
and run as  GPCORE_LIB_PATH=tools/lab/libgpcore_stamps.so python tools/ep_block2_stamps.py [n] [quiet].
quiet = 1: end-of-sweep refactorisation (GPCORE_EP_PIPELINE=0, GPCORE_EP_FUSED=1): the chain kernel without the other streams' GEMMs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2 and sys.argv[2] == "1":
    os.environ["GPCORE_EP_PIPELINE"] = "0"
os.environ.setdefault("GPCORE_EP_FUSED", "1")
import numpy as np  # noqa: E402
from gp_algos_amd import core, synth  # noqa: E402

p = synth.config_c4(n, 8)
ctx = core.Context(0)
K = ctx.gram_rbf(p["X"], p["theta"], full=True)
ep = core.EpClassifierState(ctx, K, p["y"])
ep.sweep(4)
ctx.sync()
st = (C.c_ulonglong * 512)()
ctx._lib.gp_debug_ep_stamps.restype = C.c_int
assert ctx._lib.gp_debug_ep_stamps(st) == 0
a = np.array(list(st)[:38], dtype=np.float64).reshape(19, 2)
names = ["start", "loads landed (DMA strip, D tiles, L fragments)", "solve done (this wave)", "barrier after solve", "X/X2 stores issued + mean",
         "D update done (this wave)", "own stores acknowledged", "barrier: strip dead", "A written, chain starts"] + \
        ["chunk %d done" % c for c in range(8)] + ["site loop done", "outputs + tile inverses done"]
t0, c0 = a[0]
prev = a[0]
for k, nm in enumerate(names):
    rt, cy = a[k]
    print("%-52s %8.2f us  (+%6.2f us, +%7.0f cycles, %.2f GHz)" % (nm, (rt - t0) / 100.0, (rt - prev[0]) / 100.0, cy - prev[1],
                                                                   (cy - prev[1]) / max(1.0, (rt - prev[0]) * 10.0)))
    prev = a[k]
