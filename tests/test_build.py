"""Compile-only tier: every HIP source cross-compiles for gfx950 without a GPU, and the
compiler's per-kernel resource report stays inside the budgets the kernels were designed for."""
import ctypes
import os

import pytest

from pytorch_distributed_training_amd.ops import build


def test_library_is_built_and_resources_are_reported():
    lib = build.build()
    assert os.path.exists(lib) and build.is_up_to_date()
    res = build.kernel_resources()
    assert set(res) == set(build.SOURCES)
    kernels = {k: v for per_src in res.values() for k, v in per_src.items()}
    assert len(kernels) >= 40
    for name, k in kernels.items():
        assert k["lds"] <= 160 * 1024, (name, k)                 # CDNA4: 160 KB of LDS per CU
        assert k["vgprs"] + k.get("agprs", 0) <= 512, (name, k)
        assert k["vgpr_spill"] == 0 and k["scratch"] == 0, (name, k)    # nothing spills to scratch memory
    gemm = {n: k for n, k in kernels.items() if "igemm_kernel" in n or "wgrad_kernel" in n}
    assert len(gemm) >= 12
    for name, k in gemm.items():
        assert k["occupancy"] >= 2, (name, k)                    # two workgroups per CU hide the tile prologue
    table = build.resources_table()
    assert "igemm_kernel<128, 128, 64, 2, 2, false>" in table or "igemm_kernel" in table


def test_exported_entry_points_resolve_without_a_gpu():
    lib = ctypes.CDLL(build.build())
    for fn in ("pdt_conv_igemm", "pdt_conv_igemm_bstats", "pdt_conv_wgrad", "pdt_bn_finalize_fused",
               "pdt_bn_bwd_sums_fused", "pdt_bn_apply", "pdt_bn_apply2", "pdt_bn_bwd_elemt", "pdt_sgd_step",
               "pdt_softmax_ce", "pdt_car_create", "pdt_car_allreduce", "pdt_transpose_w_batched"):
        assert hasattr(lib, fn), fn


def test_remark_parser():
    text = ("a.hip:4:1: remark: Function Name: _Z1kv [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:4:1: remark:     VGPRs: 12 [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:4:1: remark:     LDS Size [bytes/block]: 16392 [-Rpass-analysis=kernel-resource-usage]\n"
            "a.hip:4:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]\n")
    assert build.parse_resource_remarks(text) == {"_Z1kv": {"vgprs": 12, "lds": 16392, "occupancy": 8}}


def test_host_sanitizer_library_loads_and_its_host_code_runs_clean():
    """The ASan + UBSan build is loaded into a child python (ASan runtime preloaded) and its launch wrappers are
    called on paths that return before touching a GPU; a sanitizer report would abort the child.  The library
    stays in the tree: the GPU tier runs real launches through it (tests/test_sanitizer_gpu.py)."""
    import subprocess
    import sys
    lib = build.build(preset="asan")
    assert lib.endswith("libpdt_hip_asan.so") and os.path.getsize(lib) > 0 and build.is_up_to_date("asan")
    env = build.asan_env()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "_asan_probe.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and "ASAN_PROBE_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    with pytest.raises(ValueError):
        build.build(preset="tsan")


def test_prefetched_epilogue_registers_are_untouched_until_their_wait(tmp_path):
    """The convolution kernels that prefetch the next tile's epilogue operands request them with inline-assembly loads the
    compiler cannot see landing.  Audit of the generated gfx950 code: between each group of requests and the wait that
    names their destination registers, no instruction reads or writes one of them (a compiler-placed copy or spill
    there would move bytes that have not arrived).  bench/diag/audit_prefetch_regs.py does the walk."""
    import importlib.util
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from pytorch_distributed_training_amd.ops import build as kbuild
    asm = tmp_path / "conv_dma.s"
    flags = [f for f in kbuild.FLAGS if f != "-fPIC"]
    subprocess.run([hipcc, *flags, "-I", kbuild.CSRC, "-S", "--offload-device-only",
                    os.path.join(kbuild.CSRC, "conv_dma.hip"), "-o", str(asm)], check=True, capture_output=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_prefetch_regs", os.path.join(root, "bench", "diag", "audit_prefetch_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(str(asm)) == 0
