"""Build-time guard (runs without a GPU: hipcc cross-compiles gfx950): no hand-written kernel may spill to scratch.
A kernel that needs scratch memory makes every launch wait on the queue's scratch allocation and cost the lockstep LML batch
half its throughput once (a panel-solve variant that grew past 256 VGPRs), so it is checked, not assumed."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(ROOT, "gp_algos_amd", "csrc", "kernels_*.hip")) +
                                       glob.glob(os.path.join(ROOT, "gp_algos_amd", "csrc", "gpcore_ep.hip"))),
                         ids=os.path.basename)
def test_no_kernel_uses_scratch(src, tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
                          "-o", str(tmp_path / "x.o"), src], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert names and len(names) == len(scratch)
    spilled = {n: s for n, s in zip(names, scratch) if s}
    assert not spilled, "kernels spilling to scratch: %r" % spilled
