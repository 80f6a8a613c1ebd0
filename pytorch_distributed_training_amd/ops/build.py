"""Build the gfx950 kernel library ``libpdt_hip.so`` in-tree with ``hipcc``.

The kernels have a plain C ABI and do not include any PyTorch header, so each source
compiles in seconds and the result is loaded with ``ctypes``.  ``hipcc`` cross-compiles
for gfx950 without a GPU present.  Usage::

    python -m pytorch_distributed_training_amd.ops.build [--force] [--verbose]
    python -m pytorch_distributed_training_amd.ops.build --resources      # per-kernel VGPR/LDS/occupancy table
    python -m pytorch_distributed_training_amd.ops.build --preset asan    # host-side ASan+UBSan debug library
    python -m pytorch_distributed_training_amd.ops.build --preset ubsan   # host-side UBSan library (can drive a GPU)

Every compile runs with ``-Rpass-analysis=kernel-resource-usage``; the per-kernel register, LDS,
scratch and occupancy figures the compiler reports are kept in ``_build/resources.json``.  The
``asan`` preset instruments the HOST half of each source (launch wrappers, table builders, IPC
setup) with AddressSanitizer + UBSan and writes ``libpdt_hip_asan.so``; device code is unchanged.
A process runs on it with the environment :func:`asan_env` returns (ASan runtime preloaded,
``PDT_HIP_LIB=<path>``): ``tests/test_build.py`` drives every wrapper path that returns before a GPU
call that way.  With a GPU the ASan runtime of this toolchain cannot be used (it takes over the HSA
memory-pool allocator for device-side checking and aborts without XNACK), so real launches run on the
``ubsan`` preset instead (``libpdt_hip_ubsan.so``, :func:`ubsan_env`; ``tests/test_sanitizer_gpu.py``).
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
SOURCES = ["elementwise.hip", "reduce.hip", "conv_igemm.hip", "conv_dma.hip", "conv_wgrad.hip", "bn_fold.hip",
           "xgmi_allreduce.hip"]
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]
REMARKS = ["-Rpass-analysis=kernel-resource-usage"]
ASAN_LIB = os.path.join(HERE, "libpdt_hip_asan.so")
ASAN_FLAGS = ["-g", "-fno-omit-frame-pointer", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined"]
UBSAN_LIB = os.path.join(HERE, "libpdt_hip_ubsan.so")
UBSAN_FLAGS = ["-g", "-fno-omit-frame-pointer", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
               "-fno-sanitize-recover=undefined"]
_PRESETS = {"release": (LIB, "stamp", ".o"), "asan": (ASAN_LIB, "stamp.asan", ".asan.o"),
            "ubsan": (UBSAN_LIB, "stamp.ubsan", ".ubsan.o")}
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


def _stamp_text(lib: str = LIB) -> str:
    """Digest of the sources plus the identity of the library file they were built into: a library
    left over from other sources (a checkout that restored the sources but not the binary) has a
    different size / modification time and is rebuilt."""
    st = os.stat(lib)
    return f"{sources_digest()} {st.st_size} {st.st_mtime_ns}"


def is_up_to_date(preset: str = "release") -> bool:
    lib, stamp_name, _ = _PRESETS[preset]
    stamp = os.path.join(BUILD, stamp_name)
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _stamp_text(lib)


def asan_runtime() -> str:
    """Path of the AddressSanitizer runtime that matches ``hipcc``'s clang (to be preloaded by a process that
    loads the ``asan`` preset's library into an uninstrumented python)."""
    res = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = res.stdout.strip()
    if res.returncode != 0 or not os.path.isabs(path) or not os.path.exists(path):
        import glob
        found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        if not found:
            raise RuntimeError("libclang_rt.asan-x86_64.so not found")
        path = found[-1]
    return path


def asan_env(base: dict = None) -> dict:
    """Environment for a child process that runs on the host-sanitized library: the ASan runtime is APPENDED to
    whatever is already preloaded (python itself is not instrumented, hence no link-order check; leak checking is
    off because the interpreter and the HIP runtime never free their start-up allocations)."""
    env = dict(os.environ if base is None else base)
    pre = [p for p in env.get("LD_PRELOAD", "").split(":") if p]
    env["LD_PRELOAD"] = ":".join(pre + [asan_runtime()])
    env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["PDT_HIP_LIB"] = ASAN_LIB
    return env


def ubsan_env(base: dict = None) -> dict:
    """Environment for a child process on the UBSan-only library (its runtime is linked in: nothing to preload).
    This is the sanitized build that can drive a GPU: the ASan runtime of this toolchain intercepts the HSA
    memory-pool allocator for device-side checking, which needs XNACK and aborts without it."""
    env = dict(os.environ if base is None else base)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["PDT_HIP_LIB"] = UBSAN_LIB
    return env


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

    ``preset="asan"`` / ``"ubsan"`` build the host-sanitized debug libraries next to the release one."""
    if preset not in _PRESETS:
        raise ValueError(f"unknown build preset {preset!r}")
    lib, stamp_name, obj_ext = _PRESETS[preset]
    if not force and is_up_to_date(preset):
        return lib
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    flags = FLAGS + {"release": REMARKS, "asan": ASAN_FLAGS, "ubsan": UBSAN_FLAGS}[preset]

    def compile_one(src: str):
        obj = os.path.join(BUILD, src.replace(".hip", obj_ext))
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
    if preset == "asan":
        cmd += ["-fsanitize=address", "-fsanitize=undefined", "-fno-gpu-sanitize"]
    elif preset == "ubsan":
        # a shared library gets no sanitizer runtime by default: link the shared UBSan runtime and record where it lives
        rt_dir = os.path.dirname(asan_runtime())
        cmd += ["-fsanitize=undefined", "-fno-gpu-sanitize", "-shared-libsan", f"-Wl,-rpath,{rt_dir}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, lib)
    if preset != "release":
        with open(os.path.join(BUILD, stamp_name), "w") as f:
            f.write(_stamp_text(lib))
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
