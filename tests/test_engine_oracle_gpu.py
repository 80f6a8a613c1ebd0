"""GPU tier: an end-to-end oracle for the HIP engine that can fail.

Upstream gets whole-network gradient correctness from autograd (reference ``train_distributed.py:274-277``); the native
engine is an explicit backward program, so a sign error in one fused epilogue of a deep layer must show up in a test.
Gradients of a RANDOMLY INITIALISED ResNet on a handful of random images do not qualify as a yardstick: one bf16 rounding of
an activation flips ReLU signs and moves the deep gradients by tens of per cent, so bf16 and fp32 runs of ONE correct
implementation disagree (cosine 0.15 on such inputs) and any comparison "within a factor of that" passes anything.

Here the network is CONDITIONED first: the last BatchNorm of every block starts small (residual branches are
perturbations), it is trained for a few steps by the fp32 REFERENCE engine (PyTorch ops) on a learnable task (class
prototypes plus noise), and the probe batch is large.  Then
  1. the bf16 reference engine's gradient must be close to the fp32 reference engine's, tensor by tensor -- the precondition
     that makes the comparison mean something (asserted, with the measured floor per model);
  2. the HIP engine's gradient must be as close to fp32 as the bf16 reference is, tensor by tensor (a margin for the
     different rounding points of the fused kernels), with the folded / fused block tails on AND off;
  3. loss within 1 %, running statistics within 1e-2.
"""
import pytest
import torch

from pytorch_distributed_training_amd.engine import NativeResNet
from pytorch_distributed_training_amd.models import get_model, get_spec
from pytorch_distributed_training_amd.ops import ref
from pytorch_distributed_training_amd.optim import FlatSGD

pytestmark = pytest.mark.gpu

N_PROTO = 16


def _batch(proto, B, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    label = torch.randint(0, N_PROTO, (B,), generator=g)
    img = proto[label] + 0.5 * torch.randn(B, *proto.shape[1:], generator=g)
    return img.to(device), label.to(device)


def conditioned_state(name, device, H, B, steps, gamma_last=0.25, lr=0.05):
    """state_dict of ``name`` after ``steps`` fp32 reference-engine SGD steps on the prototype task, and the prototypes."""
    spec = get_spec(name, 1000)
    torch.manual_seed(0)
    sd = get_model(name, 1000).state_dict()
    for blk in spec.blocks:
        key = blk.units[-1].bn + ".weight"
        sd[key] = sd[key] * gamma_last
    eng = NativeResNet(spec, device, ref, torch.float32)
    eng.load_state_dict(sd)
    proto = torch.randn(N_PROTO, 3, H, H, generator=torch.Generator().manual_seed(5))
    opt = FlatSGD(eng.params, ref, lr=lr, momentum=0.9, weight_decay=1e-4)
    for s in range(steps):
        img, label = _batch(proto, B, 100 + s, device)
        eng.forward_backward(img, label)
        opt.step(1.0)
        eng.refresh_derived_weights()
    return spec, {k: v.clone() for k, v in eng.state_dict().items()}, proto


def _per_tensor_cos(a, b):
    cos = torch.nn.functional.cosine_similarity
    return {k: cos(a.params.grad(k).flatten().float(), b.params.grad(k).flatten().float(), dim=0).item()
            for k in a.params.order}


# (model, probe batch, image size, conditioning steps, floor of the bf16 REFERENCE engine's per-tensor cosine to fp32,
#  floor of its whole-gradient cosine, per-tensor margin of the HIP engine against the bf16 reference): measured on this
#  construction, with margin.  ResNet-50's worst tensors (BatchNorm biases of the first stage) sit at 0.88-0.93 for BOTH
#  bf16 implementations and move by +-0.04 from run to run (the conditioning itself is not bit-reproducible), hence the
#  wider margin there; a sign error in one layer scores -0.9 (see the test below).
CASES = [("ResNet18", 64, 128, 12, 0.95, 0.985, 0.04), ("ResNet50", 64, 128, 14, 0.85, 0.93, 0.08)]


@pytest.mark.parametrize("name,B,H,steps,floor_tensor,floor_whole,margin", CASES)
def test_hip_engine_against_fp32_on_a_conditioned_network(device, monkeypatch, name, B, H, steps, floor_tensor, floor_whole, margin):
    from pytorch_distributed_training_amd.ops import hip
    spec, sd, proto = conditioned_state(name, device, H, B, steps)
    img, label = _batch(proto, B, 999, device)

    def run(ops, dtype):
        eng = NativeResNet(spec, device, ops, dtype)
        eng.load_state_dict(sd)
        loss = eng.forward_backward(img, label)
        torch.cuda.synchronize()
        return eng, float(loss)

    cos = torch.nn.functional.cosine_similarity
    f32, l32 = run(ref, torch.float32)
    r16, lr16 = run(ref, torch.bfloat16)
    pre = _per_tensor_cos(r16, f32)
    pre_whole = cos(r16.params.g.float(), f32.params.g.float(), dim=0).item()
    worst_pre = min(pre.items(), key=lambda kv: kv[1])
    print(f"{name}: loss fp32 {l32:.4f} bf16-ref {lr16:.4f}; bf16-ref vs fp32: whole {pre_whole:.4f}, worst tensor "
          f"{worst_pre[0]} {worst_pre[1]:.4f}")
    # 1. the yardstick is meaningful on this network
    assert l32 > 0.05, "the conditioning over-fitted: no gradient signal left"
    assert pre_whole >= floor_whole and worst_pre[1] >= floor_tensor, (pre_whole, worst_pre)

    def nerr(x, y):
        return ((x.float() - y.float()).norm() / (y.float().norm() + 1e-12)).item()

    for fold in ("1", "0"):
        # fold "1": every block tail folded / fused whatever its size (the shipped rule needs tensors of >= 150 MB);
        # "0": the plain passes
        monkeypatch.setenv("PDT_BN_FOLD", fold)
        monkeypatch.setenv("PDT_BN_FOLD_MIN_MB", "0")
        monkeypatch.setenv("PDT_BN_FOLD_MAX_COUT", "0")
        eng, lh = run(hip, torch.bfloat16)
        if fold == "1" and name != "ResNet18":
            assert eng._can_fold(eng.spec.blocks[0].units[-1], 1 << 30)
        got = _per_tensor_cos(eng, f32)
        whole = cos(eng.params.g.float(), f32.params.g.float(), dim=0).item()
        worst = min(got.items(), key=lambda kv: kv[1])
        print(f"  fold={fold}: loss {lh:.4f}; hip vs fp32: whole {whole:.4f}, worst tensor {worst[0]} {worst[1]:.4f}")
        # 2. as close to fp32 as a known-good bf16 implementation, tensor by tensor
        assert abs(lh - l32) <= 0.01 * max(abs(l32), 1e-3) + 2e-3, (lh, l32)
        assert whole >= pre_whole - 0.01, (whole, pre_whole)
        bad = {k: (v, pre[k]) for k, v in got.items() if v < min(pre[k], 0.995) - margin}
        assert not bad, f"fold={fold}: tensors further from fp32 than the bf16 reference: {dict(list(bad.items())[:6])}"
        # 3. running statistics (updated once by this step)
        assert nerr(eng.running_mean, f32.running_mean) < 1e-2 and nerr(eng.running_var, f32.running_var) < 1e-2


def test_a_wrong_sign_in_one_layer_fails_the_oracle(device):
    """The yardstick has teeth: flipping the sign of ONE deep layer's weight gradient (what a sign error in that layer's
    fused epilogue would do to it, and only to it) is far outside what the oracle accepts."""
    from pytorch_distributed_training_amd.ops import hip
    name, B, H, steps, floor_tensor, _, _ = CASES[0]
    spec, sd, proto = conditioned_state(name, device, H, B, steps)
    img, label = _batch(proto, B, 999, device)
    f32 = NativeResNet(spec, device, ref, torch.float32)
    f32.load_state_dict(sd)
    f32.forward_backward(img, label)
    eng = NativeResNet(spec, device, hip, torch.bfloat16)
    eng.load_state_dict(sd)
    eng.forward_backward(img, label)
    torch.cuda.synchronize()
    good = _per_tensor_cos(eng, f32)
    key = "layer3.1.conv2.weight"
    assert good[key] > floor_tensor - 0.04
    eng.params.grad(key).neg_()
    assert _per_tensor_cos(eng, f32)[key] < -0.9


def test_record_what_stock_autocast_scores_on_unconditioned_inputs(device, capsys):
    """Not an assertion about this framework: on a randomly initialised ResNet-50 and 8 random images, stock PyTorch
    autograd under bf16 autocast (a known-good implementation) is itself far from its own fp32 run -- which is why the
    oracle above conditions the network first.  The numbers are printed for the record (profiles/)."""
    torch.manual_seed(0)
    m32 = get_model("ResNet50", 1000).to(device).train()
    torch.manual_seed(1)
    img = torch.randn(8, 3, 128, 128, device=device)
    label = torch.randint(0, 1000, (8,), device=device)

    def grads(autocast):
        m32.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss = torch.nn.functional.cross_entropy(m32(img).float(), label)
        loss.backward()
        return torch.cat([p.grad.flatten().float() for p in m32.parameters()]), loss.item()

    g32, l32 = grads(False)
    g16, l16 = grads(True)
    c = torch.nn.functional.cosine_similarity(g16, g32, dim=0).item()
    print(f"stock autocast bf16 vs fp32, random-init ResNet-50, 8 random images at 128 px: loss {l16:.4f} vs {l32:.4f}, "
          f"gradient cosine {c:.4f}")
    assert l16 == l16 and l32 == l32
