"""GPU tier: real kernel launches through the host-sanitized (UBSan) build of the kernel library.

Device code is the same as in the release library; what the sanitizers watch is the host half -- parameter
structs, tap tables, grid and workspace arithmetic, the deterministic-mode switches -- while convolution,
weight-gradient and statistics launches run and are checked against the PyTorch reference ops."""
import os
import subprocess
import sys

import pytest

from pytorch_distributed_training_amd.ops import build

pytestmark = pytest.mark.gpu


def test_kernel_launches_run_clean_under_host_ubsan():
    if not os.path.exists(build.UBSAN_LIB):          # (build stamps do not travel to the GPU box: existence is the check)
        pytest.fail("libpdt_hip_ubsan.so is missing: run __graft_entry__.build() (or "
                    "`python -m pytorch_distributed_training_amd.ops.build --preset ubsan`) before the GPU tier")
    env = build.ubsan_env()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "_asan_probe.py"), "gpu"], env=env,
                         capture_output=True, text=True, timeout=150)
    assert res.returncode == 0 and "ASAN_PROBE_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
