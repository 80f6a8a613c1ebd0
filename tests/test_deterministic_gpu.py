"""GPU tier: ``runtime.deterministic`` -- two runs on the same inputs end with the same bits.

The fast path adds BatchNorm statistics, weight gradients and the loss with float atomics, whose order
depends on which workgroup finishes first; the deterministic path must (a) compute the same values within
rounding and (b) repeat itself exactly.
"""
import pytest
import torch

from pytorch_distributed_training_amd.ops import ref

pytestmark = pytest.mark.gpu


@pytest.fixture()
def det_hip(device):
    from pytorch_distributed_training_amd.ops import hip
    hip.set_deterministic(True)
    yield hip
    hip.set_deterministic(False)


def _cfg(deterministic: bool, name: str = "ResNet50"):
    from pytorch_distributed_training_amd.config import validate_cfg
    return validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 100, "image_size": 96, "input_format": "uint8_nhwc"},
        "training": {"optimizer": {"name": "SGD", "lr": 0.05, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "multi_step", "milestones": [100], "gamma": 0.1}, "train_iters": 10,
                     "batch_size": 32, "per_gpu_batch_size": 32, "sync_bn": False, "precision": "bf16"},
        "runtime": {"deterministic": deterministic},
        "model": {"name": name}})


def _run(device, deterministic: bool, steps: int = 6):
    from pytorch_distributed_training_amd.engine.engines import build_engine
    torch.manual_seed(0)
    eng = build_engine(_cfg(deterministic), device)
    gen = torch.Generator(device="cpu").manual_seed(7)
    losses = []
    for _ in range(steps):          # eager step, recorded step, then replays
        img = torch.randint(0, 256, (32, 96, 96, 3), generator=gen, dtype=torch.uint8).to(device)
        label = torch.randint(0, 100, (32,), generator=gen).to(device)
        losses.append(eng.train_step(img, label).item())
    torch.cuda.synchronize()
    m = eng.model
    out = (eng.flat_parameters().clone(), m.running_mean.clone(), m.running_var.clone(), losses,
           m.active_replay_mode())
    from pytorch_distributed_training_amd.ops import hip
    hip.set_deterministic(False)
    return out


def test_two_deterministic_runs_end_bit_identical(device):
    p1, rm1, rv1, l1, mode1 = _run(device, True)
    p2, rm2, rv2, l2, mode2 = _run(device, True)
    assert mode1 == mode2
    assert l1 == l2, (l1, l2)
    assert torch.equal(p1, p2), f"{(p1 != p2).sum().item()} of {p1.numel()} parameters differ"
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    assert all(l == l for l in l1)


def test_two_deterministic_runs_with_every_batchnorm_backward_folded(device, monkeypatch):
    """The folded BatchNorm backward (forced onto every bottleneck block, however small) keeps the guarantee: its
    products go through the slab weight gradient, its small kernels have one writer per element."""
    monkeypatch.setenv("PDT_BN_FOLD_MIN_MB", "0")
    monkeypatch.setenv("PDT_BN_FOLD_MAX_COUT", "0")
    p1, rm1, rv1, l1, _ = _run(device, True)
    p2, rm2, rv2, l2, _ = _run(device, True)
    assert l1 == l2, (l1, l2)
    assert torch.equal(p1, p2), f"{(p1 != p2).sum().item()} of {p1.numel()} parameters differ"
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    assert all(l == l for l in l1)


def test_deterministic_run_tracks_the_fast_path(device):
    """Same math, different summation order: the first loss agrees to bf16 noise and the gradients next to the
    loss point the same way.  (Deeper in the network nothing tight can be asked: statistics that differ in their
    last fp32 bit flip a few bf16 roundings in the first layers, those reach most activations a few layers on,
    and on random weights and random images two FAST runs agree with each other no better than with this one.)"""
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec
    from pytorch_distributed_training_amd.ops import hip
    torch.manual_seed(1)
    img = torch.randint(0, 256, (32, 96, 96, 3), dtype=torch.uint8, device=device)
    label = torch.randint(0, 100, (32,), device=device)
    out = {}
    try:
        for det in (True, False):
            hip.set_deterministic(det)
            torch.manual_seed(0)
            eng = NativeResNet(get_spec("ResNet50", 100), device, hip, torch.bfloat16, hip_graphs=False)
            eng.load_state_dict(get_model("ResNet50", 100).state_dict())
            loss = eng.forward_backward(img, label).item()
            torch.cuda.synchronize()
            out[det] = (loss, eng.params.grad("fc.weight").flatten().clone(), eng.params.grad("fc.bias").clone())
    finally:
        hip.set_deterministic(False)
    (ld, wd, bd), (lf, wf, bf) = out[True], out[False]
    cos = torch.nn.functional.cosine_similarity
    print(f"deterministic vs fast: loss {ld:.5f} / {lf:.5f}, fc.weight gradient cosine {cos(wd, wf, dim=0).item():.4f}")
    assert abs(ld - lf) < 2e-2 * max(1.0, abs(lf)), (ld, lf)
    assert cos(wd, wf, dim=0).item() > 0.95 and cos(bd, bf, dim=0).item() > 0.999


@pytest.mark.parametrize("Ci,Co,H,R,stride", [(64, 64, 56, 3, 1), (256, 128, 28, 1, 1), (128, 128, 28, 3, 2),
                                              (512, 2048, 7, 1, 1)])
def test_deterministic_wgrad_matches_oracle_and_repeats(det_hip, device, Ci, Co, H, R, stride):
    hip = det_hip
    B, pad = 16, R // 2
    torch.manual_seed(3)
    x = torch.randn(B, H, H, Ci, device=device).to(torch.bfloat16)
    Ho = (H + 2 * pad - R) // stride + 1
    dy = torch.randn(B, Ho, Ho, Co, device=device).to(torch.bfloat16)
    base = torch.randn(Co, R, R, Ci, device=device)              # accumulation on top of existing content
    outs = []
    for _ in range(2):
        dw = base.clone()
        hip.conv_wgrad(x, dy, R, R, stride, pad, dw)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    want = base.clone()
    ref.conv_wgrad(x.float(), dy.float(), R, R, stride, pad, want)
    err = ((outs[0] - want).norm() / want.norm()).item()
    assert err < 2e-3, err


def test_deterministic_wgrad_with_a_small_workspace(det_hip, device, monkeypatch):
    """The slice count is capped by the workspace; a workspace smaller than one slab is refused."""
    hip = det_hip
    torch.manual_seed(4)
    x = torch.randn(8, 14, 14, 256, device=device).to(torch.bfloat16)
    dy = torch.randn(8, 14, 14, 256, device=device).to(torch.bfloat16)
    want = torch.zeros(256, 3, 3, 256, device=device)
    ref.conv_wgrad(x.float(), dy.float(), 3, 3, 1, 1, want)
    slab = want.numel() * 4
    monkeypatch.setattr(hip, "WGRAD_WORKSPACE_BYTES", slab + 1024)       # room for exactly one slice
    hip._wgrad_ws.clear()
    dw = torch.zeros_like(want)
    hip.conv_wgrad(x, dy, 3, 3, 1, 1, dw)
    assert ((dw - want).norm() / want.norm()).item() < 2e-3
    monkeypatch.setattr(hip, "WGRAD_WORKSPACE_BYTES", slab // 2)
    hip._wgrad_ws.clear()
    with pytest.raises(hip.HipKernelError):
        hip.conv_wgrad(x, dy, 3, 3, 1, 1, torch.zeros_like(want))
    hip._wgrad_ws.clear()


@pytest.mark.parametrize("C,H,B", [(64, 112, 16), (256, 56, 16), (2048, 7, 32)])
def test_deterministic_statistics_repeat_and_match(det_hip, device, C, H, B):
    hip = det_hip
    torch.manual_seed(5)
    y = (torch.randn(B, H, H, C, device=device) * 2 + 0.5).to(torch.bfloat16)
    m1 = hip.bn_stats(y).clone()
    m2 = hip.bn_stats(y).clone()
    assert torch.equal(m1, m2)
    want = ref.bn_stats(y.float())
    assert torch.allclose(m1, want, rtol=1e-3, atol=1e-3)


def test_deterministic_loss_and_bias_gradient(det_hip, device):
    hip = det_hip
    torch.manual_seed(6)
    logits = torch.randn(512, 1000, device=device).to(torch.bfloat16)
    labels = torch.randint(0, 1000, (512,), device=device)
    l1, d1 = hip.softmax_ce(logits, labels)
    l2, d2 = hip.softmax_ce(logits, labels)
    assert l1.item() == l2.item() and torch.equal(d1, d2)
    want = torch.nn.functional.cross_entropy(logits.float(), labels)
    assert abs(l1.item() - want.item()) < 1e-4 * max(1.0, want.item())
