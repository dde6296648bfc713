#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference for the data files).

1. boston.csv / cancer.csv / bostonPredResults_head.txt: DATA files the reference ships under src/main/resources
   (inputs of its GpPredictorTest and a stored posterior dump); copied verbatim -- data, not source.
2. oracle_vectors.json: small seeded problems with the CPU oracle's outputs (fit / predict / LML /
   gradient / EP).  The JVM reference cannot run here (no JDK), so these vectors are
   ORACLE-generated: they pin the oracle and the HIP path against regressions; parity against the JVM is
   pinned only by the reference's own known-answer literals (tests/test_oracle_kat.py)."""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/main/resources"


def main():
    if os.path.isdir(REF):
        shutil.copyfile(os.path.join(REF, "boston.csv"), os.path.join(HERE, "boston.csv"))
        shutil.copyfile(os.path.join(REF, "cancer.csv"), os.path.join(HERE, "cancer.csv"))
        with open(os.path.join(REF, "boston", "bostonPredResults.txt")) as f, \
                open(os.path.join(HERE, "bostonPredResults_head.txt"), "w") as g:
            for k, line in enumerate(f):
                if k >= 40:
                    break
                g.write(line)
    from gp_algos_amd import synth
    from oracle import gp_oracle as orc
    cases = []
    for (n, d, m, seed) in [(8, 1, 4, 3), (33, 2, 9, 4), (64, 3, 16, 5)]:
        p = synth.regression(n, d, m, seed, seed + 1, seed + 2, synth.ard_theta(d, 1.2, 0.9, 0.15))
        L, alpha = orc.fit(p["X"], p["y"], p["theta"])
        mean, var, cov, _ = orc.predict(p["X"], p["theta"], L, alpha, p["Xs"], full_cov=True)
        lml, grad = orc.lml_grad(p["X"], p["y"], p["theta"])
        f = p["X"].sum(axis=1) / np.sqrt(d) + 0.3 * synth.normal(seed + 9, np.arange(n))
        yc = np.where(f >= 0, 1, -1).astype(int)
        thc = np.concatenate(([1.5], 1.1 * np.ones(d), [0.0]))
        Kc = orc.gram_sym(p["X"], thc)
        ep = orc.ep_estimate(Kc, yc, 3)
        cases.append(dict(n=n, d=d, m=m, X=p["X"].tolist(), y=p["y"].tolist(), Xs=p["Xs"].tolist(), theta=p["theta"].tolist(),
                          alpha=alpha.tolist(), L_diag=np.diag(L).tolist(), lml=lml, grad=grad.tolist(), mean=mean.tolist(),
                          var=var.tolist(), cov_first_row=cov[0].tolist(), y_class=yc.tolist(), theta_class=thc.tolist(),
                          ep_tau=ep["tau"].tolist(), ep_nu=ep["nu"].tolist(), ep_lml_strict=orc.ep_lml(ep, yc, True),
                          ep_lml_corrected=orc.ep_lml(ep, yc, False)))
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(dict(note="oracle-generated (see make_fixtures.py); not JVM outputs", cases=cases), f)


if __name__ == "__main__":
    main()
