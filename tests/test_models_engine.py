"""CPU tier: model schema, native engine (reference ops) against eager autograd, optimizer, checkpoints."""
import os

import pytest
import torch

from pytorch_distributed_training_amd import checkpoint as ckpt
from pytorch_distributed_training_amd.engine import FlatParams, NativeResNet, backward_param_order, make_buckets
from pytorch_distributed_training_amd.models import get_model, get_spec, model_names
from pytorch_distributed_training_amd.ops import get_ops, ref
from pytorch_distributed_training_amd.optim import FlatSGD, get_optimizer

PARAM_COUNTS = {"ResNet18": 11689512, "ResNet34": 21797672, "ResNet50": 25557032, "ResNet101": 44549160,
                "ResNet152": 60192808}


@pytest.mark.parametrize("name", model_names())
def test_parameter_counts_and_schema(name):
    spec = get_spec(name, 1000)
    assert spec.num_parameters() == PARAM_COUNTS[name]
    schema = spec.state_dict_schema()
    assert schema["conv1.weight"][0] == (64, 3, 7, 7)
    assert schema["bn1.num_batches_tracked"] == ((), torch.int64)
    exp = 1 if name in ("ResNet18", "ResNet34") else 4
    assert schema["fc.weight"][0] == (1000, 512 * exp) and schema["fc.bias"][0] == (1000,)
    assert not any(k.startswith("module.") for k in schema)


def test_resnet50_torchvision_key_schema():
    schema = get_spec("ResNet50", 1000).state_dict_schema()
    keys = list(schema)
    assert len(keys) == 320                                       # 161 parameters + 159 buffers
    assert keys[:7] == ["conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
                        "bn1.num_batches_tracked", "layer1.0.conv1.weight"]
    assert schema["layer1.0.conv1.weight"][0] == (64, 64, 1, 1)
    assert schema["layer1.0.conv3.weight"][0] == (256, 64, 1, 1)
    assert schema["layer1.0.downsample.0.weight"][0] == (256, 64, 1, 1)
    assert schema["layer2.0.conv2.weight"][0] == (128, 128, 3, 3)      # v1.5: stride on the 3x3
    assert schema["layer2.0.downsample.0.weight"][0] == (512, 256, 1, 1)
    assert "layer1.1.downsample.0.weight" not in schema
    assert schema["layer4.2.bn3.running_var"][0] == (2048,)
    r18 = get_spec("ResNet18", 1000).state_dict_schema()
    assert "layer1.0.downsample.0.weight" not in r18 and "layer2.0.downsample.0.weight" in r18
    assert "layer1.0.conv3.weight" not in r18


@pytest.mark.parametrize("name", ["ResNet18", "ResNet50"])
def test_eager_module_state_dict_equals_schema(name):
    m = get_model(name, 10)
    schema = get_spec(name, 10).state_dict_schema()
    sd = m.state_dict()
    assert list(sd.keys()) == list(schema.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == schema[k][0] and v.dtype == schema[k][1], k
    assert [k for k, _ in m.named_parameters()] == get_spec(name, 10).parameter_keys()
    # v1.5 stride placement
    if name == "ResNet50":
        assert m.layer2[0].conv2.stride == (2, 2) and m.layer2[0].conv1.stride == (1, 1)


def _randomize_bn(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.2, 0.2)


def test_native_engine_matches_eager_autograd_resnet18():
    torch.manual_seed(0)
    m = get_model("ResNet18", 10)
    _randomize_bn(m)
    eng = NativeResNet(get_spec("ResNet18", 10), torch.device("cpu"), ref, torch.float32)
    eng.load_state_dict(m.state_dict())
    x = torch.randn(4, 3, 64, 64)
    y = torch.randint(0, 10, (4,))
    m.train()
    loss = torch.nn.functional.cross_entropy(m(x), y)
    loss.backward()
    l2 = eng.forward_backward(x, y)
    assert l2.item() == pytest.approx(loss.item(), rel=1e-5)
    g = eng.params.export("g")
    for k, p in m.named_parameters():
        # Frobenius-relative: one ReLU whose pre-activation is ~1e-6 may flip between the two
        # implementations (different rounding), which moves single elements but not the norm.
        err = (g[k] - p.grad).norm().item() / (p.grad.norm().item() + 1e-12)
        assert err < 3e-2, (k, err)
    sd, sd_ref = eng.state_dict(), m.state_dict()
    for k in sd_ref:
        assert torch.allclose(sd[k].float(), sd_ref[k].float(), rtol=1e-4, atol=1e-5), k
    # eval path uses running statistics
    m.eval()
    with torch.no_grad():
        ref_logits = m(x)
    met = eng.eval_step(x, y)
    ce = torch.nn.functional.cross_entropy(ref_logits, y, reduction="sum")
    assert met[0].item() == pytest.approx(ce.item(), rel=1e-4) and met[3].item() == 4


def test_native_engine_bottleneck_vs_fp64():
    """ResNet-50 at random init is ill-conditioned in fp32 (eager fp32 itself is percents away from
    fp64), so the native engine is required to be as close to an fp64 oracle as eager fp32 is."""
    torch.manual_seed(0)
    m = get_model("ResNet50", 10)
    eng = NativeResNet(get_spec("ResNet50", 10), torch.device("cpu"), ref, torch.float32)
    eng.load_state_dict(m.state_dict())
    m64 = get_model("ResNet50", 10).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    x = torch.randn(4, 3, 64, 64)
    y = torch.randint(0, 10, (4,))
    for net, inp in ((m, x), (m64, x.double())):
        net.train()
        torch.nn.functional.cross_entropy(net(inp), y).backward()
    eng.forward_backward(x, y)
    g = eng.params.export("g")
    worst_eager = worst_native = 0.0
    for (k, p), (_, p64) in zip(m.named_parameters(), m64.named_parameters()):
        d = p64.grad.abs().max().item() + 1e-12
        worst_eager = max(worst_eager, (p.grad.double() - p64.grad).abs().max().item() / d)
        worst_native = max(worst_native, (g[k].double() - p64.grad).abs().max().item() / d)
    assert worst_native < 2.0 * worst_eager + 1e-3, (worst_native, worst_eager)


def test_flat_params_layout_and_buckets():
    spec = get_spec("ResNet50", 1000)
    order = backward_param_order(spec)
    assert order[:2] == ["fc.weight", "fc.bias"] and order[-1] == "conv1.weight"
    assert sorted(order) == sorted(spec.parameter_keys())
    schema = spec.state_dict_schema()
    P = FlatParams([(k, schema[k][0]) for k in order], torch.device("cpu"), torch.float32)
    assert P.num_parameters() == 25557032
    assert P.param("layer1.0.conv2.weight").shape == (64, 3, 3, 64)     # OHWI internally
    assert all(e.offset % 64 == 0 for e in P.entries.values())
    buckets = make_buckets(P, 25.0, 1.0)
    assert buckets[0][0] == 0 and buckets[-1][1] == P.total
    assert all(b[1] == buckets[i + 1][0] for i, b in enumerate(buckets[:-1]))     # contiguous cover
    sizes_mb = [(e - s) * 4 / 2 ** 20 for s, e in buckets]
    assert sizes_mb[0] >= 1.0 and max(sizes_mb) < 36.0 and len(buckets) >= 4
    # layout conversion round trip
    w = torch.randn(64, 64, 3, 3)
    P.load_from({"layer1.0.conv2.weight": w}, strict=False)
    assert torch.equal(P.export()["layer1.0.conv2.weight"], w)


@pytest.mark.parametrize("nesterov,wd", [(False, 1e-4), (True, 0.0)])
def test_flat_sgd_matches_torch_sgd(nesterov, wd):
    torch.manual_seed(0)
    spec = get_spec("ResNet18", 10)
    schema = spec.state_dict_schema()
    P = FlatParams([(k, schema[k][0]) for k in backward_param_order(spec)], torch.device("cpu"), torch.float32)
    m = get_model("ResNet18", 10)
    P.load_from(m.state_dict())
    opt = FlatSGD(P, ref, lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    topt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    for _ in range(5):
        grads = {k: torch.randn_like(p) for k, p in m.named_parameters()}
        for k, p in m.named_parameters():
            p.grad = grads[k].clone()
        P.load_from({k: 2.0 * v for k, v in grads.items()}, which="g")
        topt.step()
        opt.step(grad_scale=0.5)
        assert P.g.abs().sum() == 0
    out = P.export()
    for k, p in m.named_parameters():
        assert torch.allclose(out[k], p.data, rtol=1e-5, atol=1e-6), k
    assert get_optimizer({"name": "SGD"}) is FlatSGD
    with pytest.raises(ValueError):
        get_optimizer({"name": "Adam"})


def test_no_wd_bn_segments():
    spec = get_spec("ResNet18", 10)
    schema = spec.state_dict_schema()
    P = FlatParams([(k, schema[k][0]) for k in backward_param_order(spec)], torch.device("cpu"), torch.float32)
    P.p.fill_(1.0)
    opt = FlatSGD(P, ref, lr=1.0, momentum=0.0, weight_decay=0.5, no_wd_bn=True)
    opt.step()
    assert torch.all(P.param("bn1.weight") == 1.0) and torch.all(P.param("fc.bias") == 1.0)
    assert torch.all(P.param("conv1.weight") == 0.5)


def test_checkpoint_roundtrip_and_torchvision_compat(tmp_path):
    torch.manual_seed(0)
    eng = NativeResNet(get_spec("ResNet18", 10), torch.device("cpu"), ref, torch.float32)
    eng.load_state_dict(get_model("ResNet18", 10).state_dict())
    opt = FlatSGD(eng.params, ref, lr=0.1, momentum=0.9, weight_decay=1e-4)
    x, y = torch.randn(2, 3, 32, 32), torch.randint(0, 10, (2,))
    eng.forward_backward(x, y)
    opt.step()
    path = os.path.join(tmp_path, "ck", "iter_1.pth")
    ckpt.save_checkpoint(path, eng.state_dict(), opt.state_dict(), {"last_iter": 1, "base_lrs": [0.1]}, 1, {"a": 1})
    assert ckpt.latest_checkpoint(os.path.join(tmp_path, "ck")) == path
    ck = ckpt.load_checkpoint(path)                               # weights_only=True loader
    assert ck["iter"] == 1 and ck["config"] == {"a": 1}
    # the model part loads into a stock eager module with torchvision key names, strictly
    fresh = get_model("ResNet18", 10)
    fresh.load_state_dict(ck["model"], strict=True)
    assert int(fresh.bn1.num_batches_tracked) == 1
    assert all(v.dtype in (torch.float32, torch.int64) for v in ck["model"].values())
    assert ck["model"]["conv1.weight"].shape == (64, 3, 7, 7) and ck["model"]["conv1.weight"].is_contiguous()
    # and back into a new engine + optimizer, bit-exactly
    eng2 = NativeResNet(get_spec("ResNet18", 10), torch.device("cpu"), ref, torch.float32)
    eng2.load_state_dict(ck["model"])
    opt2 = FlatSGD(eng2.params, ref, lr=0.5)
    opt2.load_state_dict(ck["optimizer"])
    assert torch.equal(eng2.params.p, eng.params.p) and torch.equal(eng2.params.m, eng.params.m)
    assert torch.equal(eng2.running_var, eng.running_var) and opt2.param_groups[0]["lr"] == 0.1
    ckpt.restore_rng(ck)
    # bare state-dict files and DDP-style "module." prefixes are accepted too
    bare = os.path.join(tmp_path, "bare.pth")
    ckpt.save_model_only(bare, {"module." + k: v for k, v in eng.state_dict().items()})
    eng3 = NativeResNet(get_spec("ResNet18", 10), torch.device("cpu"), ref, torch.float32)
    eng3.load_state_dict(ckpt.load_checkpoint(bare)["model"])
    assert torch.equal(eng3.params.p, eng.params.p)
    with pytest.raises(KeyError):
        eng3.load_state_dict({"conv1.weight": torch.zeros(64, 3, 7, 7)})


def test_ops_backend_selection():
    assert get_ops(torch.device("cpu")) is ref
    with pytest.raises(RuntimeError):
        get_ops(torch.device("cpu"), "hip")
    with pytest.raises(ValueError):
        get_ops(torch.device("cpu"), "triton")


def test_reference_ops_against_autograd():
    """The oracle itself: hand-written BN / pooling / loss backward formulas vs autograd."""
    torch.manual_seed(0)
    y = torch.randn(3, 5, 5, 8, requires_grad=True)
    gamma, beta = torch.rand(8) + 0.5, torch.randn(8)
    rm, rv = torch.zeros(8), torch.ones(8)
    out = torch.relu(torch.nn.functional.batch_norm(y.permute(0, 3, 1, 2), None, None, gamma, beta, True, 0.1, 1e-5))
    g = torch.randn_like(out)
    out.backward(g)
    mom = ref.bn_stats(y.detach(), rm)
    scale, shift, mean, invstd = ref.bn_finalize(mom, 1.0, gamma, beta, rm, rv, 0.1, 1e-5, 75)
    gn = g.permute(0, 2, 3, 1).contiguous()
    sums = ref.bn_bwd_reduce(gn, y.detach(), mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE)
    dx = ref.bn_bwd_elemt(gn, y.detach(), mean, invstd, scale, shift, sums, 1.0 / 75, None, ref.MASK_RECOMPUTE)
    assert torch.allclose(dx, y.grad, rtol=1e-4, atol=1e-5)
    a = ref.bn_apply(y.detach(), scale, shift, None, True)
    assert torch.allclose(a, out.detach().permute(0, 2, 3, 1), rtol=1e-5, atol=1e-6)
    # maxpool with ties
    x = torch.relu(torch.randn(2, 9, 9, 4)).requires_grad_(True)
    p = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1)
    gp = torch.randn_like(p)
    p.backward(gp)
    o, idx = ref.maxpool_fwd(x.detach())
    dxp = ref.maxpool_bwd(gp.permute(0, 2, 3, 1).contiguous(), idx, tuple(x.shape))
    assert torch.allclose(o, p.detach().permute(0, 2, 3, 1)) and torch.allclose(dxp, x.grad, atol=1e-6)
    # softmax-CE
    logits = torch.randn(6, 11, requires_grad=True)
    lab = torch.randint(0, 11, (6,))
    l = torch.nn.functional.cross_entropy(logits, lab)
    l.backward()
    l2, d = ref.softmax_ce(logits.detach(), lab)
    assert l2.item() == pytest.approx(l.item(), rel=1e-6) and torch.allclose(d, logits.grad, atol=1e-6)
