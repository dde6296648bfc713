"""CPU-only sanitizer pass (ASan + UBSan) over the C oracle: GPU AddressSanitizer is not available on the pool, so
memory-safety of the checker is established on the CPU build, through every entry point on small problems."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r'''
import ctypes, numpy as np, sys
sys.path.insert(0, %r)
from oracle import gp_oracle as orc
lib = ctypes.CDLL(%r)
for name in ("orc_dnorm", "orc_pnorm", "orc_rbf_kernel", "orc_lml", "orc_ep_lml", "orc_avg_between_site_params"):
    getattr(lib, name).restype = ctypes.c_double
lib.orc_dnorm.argtypes = [ctypes.c_double]; lib.orc_pnorm.argtypes = [ctypes.c_double]
orc._lib = lib
from gp_algos_amd import synth
p = synth.regression(40, 3, 7, 1, 2, 3, synth.ard_theta(3, 1.2, 1.0, 0.2))
L, a = orc.fit(p["X"], p["y"], p["theta"])
orc.predict(p["X"], p["theta"], L, a, p["Xs"], full_cov=True, want_v=True)
orc.lml_grad(p["X"], p["y"], p["theta"]); orc.inv_triangular(L, False); orc.back_solve(L, p["y"], trans=True)
th = np.array([1.5, 1, 1, 1, 0.0]); y = np.where(p["y"] > 0, 1, -1); K = orc.gram_sym(p["X"], th)
ep = orc.ep_estimate(K, y, 3); orc.ep_lml(ep, y, True); orc.ep_lml(ep, y, False)
orc.ep_classify(K, ep["L"], ep["tau"], ep["nu"], orc.gram_cross(p["Xs"], p["X"], th), np.full(7, 2.25))
orc.ep_lml_grad(p["X"], th, K, ep["L"], ep["tau"], ep["nu"], strict=True)
orc.ep_lml_grad(p["X"], th, K, ep["L"], ep["tau"], ep["nu"], strict=False)
print("SANITIZER_CLEAN")
'''


def test_oracle_is_asan_ubsan_clean():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "_build", "libgporacle_asan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, so)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SANITIZER_CLEAN" in r.stdout, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
