"""Build the gfx950 kernel library ``libpdt_hip.so`` in-tree with ``hipcc``.

The kernels have a plain C ABI and do not include any PyTorch header, so each source
compiles in seconds and the result is loaded with ``ctypes``.  ``hipcc`` cross-compiles
for gfx950 without a GPU present.  Usage::

    python -m pytorch_distributed_training_amd.ops.build [--force] [--verbose]
    python -m pytorch_distributed_training_amd.ops.build --resources      # per-kernel VGPR/LDS/occupancy table
    python -m pytorch_distributed_training_amd.ops.build --preset asan    # host-side ASan+UBSan debug library

Every compile runs with ``-Rpass-analysis=kernel-resource-usage``; the per-kernel register, LDS,
scratch and occupancy figures the compiler reports are kept in ``_build/resources.json``.  The
``asan`` preset instruments the HOST half of each source (launch wrappers, table builders, IPC
setup) with AddressSanitizer + UBSan and writes ``libpdt_hip_asan.so``; device code is unchanged.
Run a process that loads it with the ASan runtime first in the library order and
``PDT_HIP_LIB=<path>``.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
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
SOURCES = ["elementwise.hip", "reduce.hip", "conv_igemm.hip", "conv_dma.hip", "conv_wgrad.hip", "xgmi_allreduce.hip"]
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]
REMARKS = ["-Rpass-analysis=kernel-resource-usage"]
ASAN_LIB = os.path.join(HERE, "libpdt_hip_asan.so")
ASAN_FLAGS = ["-g", "-fno-omit-frame-pointer", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined"]
RESOURCES = os.path.join(BUILD, "resources.json")
_FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
           "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
           "LDS Size [bytes/block]": "lds"}


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
    """Every file under ``csrc/`` counts (kernels and all the headers they include)."""
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    return _digest(files)


def _stamp_text() -> str:
    """Digest of the sources plus the identity of the library file they were built into: a library
    left over from other sources (a checkout that restored the sources but not the binary) has a
    different size / modification time and is rebuilt."""
    st = os.stat(LIB)
    return f"{sources_digest()} {st.st_size} {st.st_mtime_ns}"


def is_up_to_date() -> bool:
    stamp = os.path.join(BUILD, "stamp")
    if not (os.path.exists(LIB) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _stamp_text()


def parse_resource_remarks(text: str) -> dict:
    """``{mangled kernel name: {vgprs, agprs, sgprs, lds, scratch, occupancy, ...}}`` from the
    compiler's kernel-resource-usage remarks."""
    out: dict = {}
    cur = None
    for m in re.finditer(r"remark:\s+(.*?)\s+\[-Rpass-analysis=kernel-resource-usage\]", text):
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
            continue
        key, _, val = body.rpartition(":")
        if cur is not None and key.strip() in _FIELDS and val.strip().lstrip("-").isdigit():
            cur[_FIELDS[key.strip()]] = int(val)
    return out


def demangle(names: List[str]) -> List[str]:
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt")
    if not filt or not names:
        return list(names)
    res = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    got = res.stdout.splitlines()
    return got if res.returncode == 0 and len(got) == len(names) else list(names)


def build(force: bool = False, verbose: bool = False, preset: str = "release") -> str:
    """Compile every ``.hip`` source for gfx950 and link the shared library; returns its path.

    ``preset="asan"`` builds the host-sanitized debug library next to the release one."""
    if preset not in ("release", "asan"):
        raise ValueError(f"unknown build preset {preset!r}")
    asan = preset == "asan"
    if not asan and not force and is_up_to_date():
        return LIB
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    flags = FLAGS + (ASAN_FLAGS if asan else REMARKS)
    lib = ASAN_LIB if asan else LIB

    def compile_one(src: str):
        obj = os.path.join(BUILD, src.replace(".hip", ".asan.o" if asan else ".o"))
        cmd = [hipcc, *flags, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        return obj, parse_resource_remarks(res.stderr)

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    tmp = lib + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", tmp, *objs]
    if asan:
        cmd += ["-fsanitize=address", "-fsanitize=undefined", "-fno-gpu-sanitize"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, lib)
    if asan:
        return lib
    with open(RESOURCES, "w") as f:
        json.dump({src: r for src, (_, r) in zip(SOURCES, results)}, f, indent=1, sort_keys=True)
    with open(os.path.join(BUILD, "stamp"), "w") as f:
        f.write(_stamp_text())
    return LIB


def kernel_resources(rebuild_if_missing: bool = True) -> dict:
    """Per-source, per-kernel resource figures of the library that is built in-tree."""
    if not os.path.exists(RESOURCES) and rebuild_if_missing:
        build(force=True)
    with open(RESOURCES) as f:
        return json.load(f)


def resources_table() -> str:
    lines = [f"{'kernel':<78} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'LDS B':>7} {'scratch':>7} {'waves/SIMD':>10}"]
    for src, kernels in kernel_resources().items():
        lines.append(f"-- {src}")
        names = sorted(kernels)
        for name, pretty in zip(names, demangle(names)):
            k = kernels[name]
            pretty = re.sub(r"\(.*$", "", pretty)[:78]
            lines.append(f"{pretty:<78} {k.get('vgprs', 0):>5} {k.get('agprs', 0):>5} {k.get('sgprs', 0):>5} "
                         f"{k.get('lds', 0):>7} {k.get('scratch', 0):>7} {k.get('occupancy', 0):>10}")
    return "\n".join(lines)


if __name__ == "__main__":
    if "--resources" in sys.argv:
        print(resources_table())
    else:
        preset = sys.argv[sys.argv.index("--preset") + 1] if "--preset" in sys.argv else "release"
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, preset=preset))
