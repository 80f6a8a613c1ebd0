"""Build the gfx950 kernel library ``libpdt_hip.so`` in-tree with ``hipcc``.

The kernels have a plain C ABI and do not include any PyTorch header, so each source
compiles in seconds and the result is loaded with ``ctypes``.  ``hipcc`` cross-compiles
for gfx950 without a GPU present.  Usage::

    python -m pytorch_distributed_training_amd.ops.build [--force] [--verbose]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpdt_hip.so")
ARCH = "gfx950"
SOURCES = ["elementwise.hip", "reduce.hip", "conv_igemm.hip", "conv_wgrad.hip", "xgmi_allreduce.hip"]
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def _digest(paths: List[str]) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def sources_digest() -> str:
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h")]
    return _digest(files)


def is_up_to_date() -> bool:
    stamp = os.path.join(BUILD, "stamp")
    if not (os.path.exists(LIB) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == sources_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every ``.hip`` source for gfx950 and link the shared library; returns its path."""
    if not force and is_up_to_date():
        return LIB
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", tmp, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, LIB)
    with open(os.path.join(BUILD, "stamp"), "w") as f:
        f.write(sources_digest())
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(path)
