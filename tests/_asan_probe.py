"""Child process of the host-sanitizer tests: loads a sanitized build of the kernel library ($PDT_HIP_LIB; for
the ASan build the parent preloads the ASan runtime) and drives its HOST code -- argument checks, tap-table packing, launch
geometry, workspace handling -- with and without a GPU.  Any sanitizer report aborts the process."""
import ctypes
import sys

import torch


def host_only(hip) -> None:
    # rejected arguments return before any GPU call: exercises the wrappers' own loads / stores
    taps = hip._fprop_taps(3, 3, 1)
    rc = hip._real.pdt_conv_wgrad(None, None, None, 0, 8, 8, 64, 64, 8, 8, 1, 64, 64, 576, taps.T, taps.offy,
                                  taps.offx, taps.widx, 0, 0, 0, None, 0, None)
    assert rc == 0, rc                                  # M == 0: nothing to do
    rc = hip._real.pdt_conv_wgrad(None, None, None, 64, 8, 8, 64, 64, 8, 8, 1, 64, 64, 576, 0, taps.offy,
                                  taps.offx, taps.widx, 0, 0, 0, None, 0, None)
    assert rc == -3, rc                                 # no taps
    assert hip._real.pdt_set_deterministic(1) == 0 and hip._real.pdt_accumulator_rows() == 1024
    assert hip._real.pdt_set_deterministic(0) == 0 and hip._real.pdt_accumulator_rows() == 16


def on_gpu(hip) -> None:
    from pytorch_distributed_training_amd.ops import ref
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for det in (False, True):
        hip.set_deterministic(det)
        x = torch.randn(4, 14, 14, 128, device=dev).to(torch.bfloat16)
        w = (torch.randn(128, 3, 3, 128, device=dev) * 0.05).to(torch.bfloat16)
        y, stats = hip.conv_fprop(x, w, 3, 3, 1, 1, stat_shift=torch.zeros(128, device=dev))
        want, _ = ref.conv_fprop(x, w, 3, 3, 1, 1, stat_shift=None)
        assert ((y.float() - want.float()).norm() / want.float().norm()).item() < 2e-2
        dw = torch.zeros(128, 3, 3, 128, device=dev)
        hip.conv_wgrad(x, y, 3, 3, 1, 1, dw)
        dwr = torch.zeros_like(dw)
        ref.conv_wgrad(x, y, 3, 3, 1, 1, dwr)
        assert ((dw - dwr).norm() / dwr.norm()).item() < 5e-3
        m = hip.bn_stats(y)
        assert torch.allclose(m, ref.bn_stats(y), rtol=2e-3, atol=2e-3)
        # folded BatchNorm backward: prepare, data gradient with a second operand, scaled weight gradient + correction
        a = torch.relu(torch.randn(4, 14, 14, 64, device=dev)).to(torch.bfloat16)
        w1 = (torch.randn(256, 1, 1, 64, device=dev) * 0.1).to(torch.bfloat16)
        wt = torch.empty(64, 1, 1, 256, dtype=torch.bfloat16, device=dev)
        hip.transpose_weights(w1, wt)
        g = (torch.randn(4, 14, 14, 256, device=dev) * 0.1).to(torch.bfloat16)
        sums, vec = torch.randn(2, 256, device=dev), torch.rand(256, device=dev) + 0.5
        fold = hip.bn_fold_prepare(sums, vec, vec, vec, 1.0 / 784, w1, wt)
        dx = hip.conv_dgrad_folded(g, a, fold)
        dxr = ref.conv_dgrad_folded(g, a, fold)
        assert ((dx.float() - dxr.float()).norm() / dxr.float().norm()).item() < 2e-2
        dw1, dw1r = torch.zeros(256, 1, 1, 64, device=dev), torch.zeros(256, 1, 1, 64, device=dev)
        hip.conv_wgrad_folded(a, g, fold, w1, dw1)
        ref.conv_wgrad_folded(a, g, fold, w1, dw1r)
        assert ((dw1 - dw1r).norm() / dw1r.norm()).item() < 5e-3
    hip.set_deterministic(False)
    torch.cuda.synchronize()


def main() -> int:
    from pytorch_distributed_training_amd.ops import hip
    import os
    want = os.path.basename(os.environ["PDT_HIP_LIB"])
    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libpdt_hip" in ln]
    assert loaded and all(p.endswith(want) for p in loaded), loaded
    if "asan" in want:
        assert ctypes.CDLL(None).__asan_init is not None     # the runtime really is in the process
    host_only(hip)
    if len(sys.argv) > 1 and sys.argv[1] == "gpu":
        on_gpu(hip)
    print("ASAN_PROBE_OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
