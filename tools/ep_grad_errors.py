"""Achieved error of the EP log-marginal-likelihood gradient against the oracle (MarginalLikelihoodEvaluator.scala:46-66), per problem:
what the tolerances of tests/test_gpu_parity.py / test_gpu_ep_optimize.py are set from.  python tools/ep_grad_errors.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_algos_amd import synth
from gp_algos_amd.core import Context, EpClassifierState
from oracle import gp_oracle as orc

ctx = Context(0)


def problem(n, seed, d=3):
    p = synth.regression(n, d, 0, seed, seed + 1, 0, synth.ard_theta(d, 1.1, 1.0, 0.0))
    K = orc.gram_sym(p["X"], p["theta"])
    y = np.where(p["y"] >= np.median(p["y"]), 1, -1).astype(np.int32)
    return p, K, y


for n, seed, sweeps in ((150, 11, 3), (80, 21, 3), (300, 5, 4), (520, 7, 3), (1100, 47, 2)):
    p, K, y = problem(n, seed)
    o = orc.ep_estimate(K, y, sweeps)
    ep = EpClassifierState(ctx, K, y)
    ep.sweep(sweeps)
    for strict in (True, False):
        g = ep.lml_grad_rbf(p["X"], p["theta"], strict=strict)
        og = orc.ep_lml_grad(p["X"], p["theta"], K, o["L"], o["tau"], o["nu"], strict=strict)
        print("n=%4d sweeps=%d strict=%d  max|g - g_oracle| / max|g_oracle| = %.2e   (|g| range %.2e .. %.2e)"
              % (n, sweeps, strict, np.max(np.abs(g - og)) / np.max(np.abs(og)), np.min(np.abs(og)), np.max(np.abs(og))))
    ep.close()
for n, seed in ((260, 51), (1100, 47)):
    p, K, y = problem(n, seed)
    rng = np.random.default_rng(3)
    thetas = p["theta"][None, :] * rng.uniform(0.7, 1.4, size=(5, 5))
    lml, grad, sweeps, info = ctx.ep_lml_grad_rbf_batched(p["X"], y, thetas, stop_eps=-1.0, max_sweeps=2, strict=False)
    for b in range(5):
        Kb = orc.gram_sym(p["X"], thetas[b])
        o = orc.ep_estimate(Kb, y, 2)
        og = orc.ep_lml_grad(p["X"], thetas[b], Kb, o["L"], o["tau"], o["nu"], strict=False)
        ol = orc.ep_lml(o, y, False)
        print("batched n=%4d setting %d: lml rel %.2e  grad rel %.2e" % (n, b, abs(lml[b] - ol) / abs(ol), np.max(np.abs(grad[b] - og)) / np.max(np.abs(og))))
ctx.close()
