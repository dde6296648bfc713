"""EP state identities (EpParameterEstimator.scala:56-61) after a few sweeps, for the streamed and the end-of-sweep refactorisation:
python tools/ep_identity_check.py [n] [sweeps]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gp_algos_amd import _lib as L, core, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = synth.config_c4(n, 8)
ctx = core.Context(0)
K = ctx.gram_rbf(p["X"], p["theta"])
K[np.diag_indices_from(K)] += 1e-6
V = np.random.default_rng(2).standard_normal((n, 3))
KV = K @ V
res = {}
for mode in ("1", "0"):
    for per_call in (sweeps, 1):
        os.environ["GPCORE_EP_PIPELINE"] = mode
        ep = core.EpClassifierState(ctx, K, p["y"])
        for _ in range(sweeps // per_call):
            tau, nu = ep.sweep(per_call)
        Sig, mu, Lf = ep.get(L.GP_EP_GET_SIGMA), ep.get(L.GP_EP_GET_MU), ep.get(L.GP_EP_GET_L)
        st = np.sqrt(tau)
        BV = V + st[:, None] * (K @ (st[:, None] * V))
        print("pipeline=%s sweeps per call %d: |Sig(I+tau K)V - KV| %.2e  |Sig nu - mu| %.2e  |L L^T V - B V| %.2e  tau[0..2] %s" % (
            mode, per_call, np.linalg.norm(Sig @ (V + tau[:, None] * KV) - KV) / np.linalg.norm(KV),
            np.max(np.abs(Sig @ nu - mu)) / np.max(np.abs(mu)), np.linalg.norm(Lf @ (Lf.T @ V) - BV) / np.linalg.norm(BV), tau[:3]))
        res[(mode, per_call)] = tau
        ep.close()
for a in res:
    for b in res:
        if a < b:
            print(a, b, "max rel tau diff %.2e" % (np.max(np.abs(res[a] - res[b])) / np.max(np.abs(res[b]))))
