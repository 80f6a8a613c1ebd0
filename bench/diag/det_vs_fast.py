"""Gradients of one step in deterministic and in fast mode against the fp32 reference engine (one MI355X)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from pytorch_distributed_training_amd.engine import NativeResNet  # noqa: E402
from pytorch_distributed_training_amd.models import get_model, get_spec  # noqa: E402
from pytorch_distributed_training_amd.ops import hip, ref  # noqa: E402


def build(ops, dtype):
    torch.manual_seed(0)
    m = get_model("ResNet50", 100)
    eng = NativeResNet(get_spec("ResNet50", 100), dev, ops, dtype, hip_graphs=False)
    eng.load_state_dict(m.state_dict())
    return eng


dev = torch.device("cuda", 0)
B, H = int(os.environ.get("B", 32)), int(os.environ.get("H", 96))
torch.manual_seed(1)
img = torch.randint(0, 256, (B, H, H, 3), dtype=torch.uint8, device=dev)
label = torch.randint(0, 100, (B,), device=dev)
c = build(ref, torch.float32)
lc = c.forward_backward(img, label).item()
res = {}
for mode in ("fast", "det", "fast2", "det2"):
    hip.set_deterministic(mode.startswith("det"))
    e = build(hip, torch.bfloat16)
    l = e.forward_backward(img, label).item()
    torch.cuda.synchronize()
    res[mode] = (l, e.params.g.clone(), e)
hip.set_deterministic(False)
cos = torch.nn.functional.cosine_similarity
print(f"fp32 loss {lc:.5f}")
for mode, (l, g, e) in res.items():
    print(f"{mode}: loss {l:.5f}  cos(g, fp32) {cos(g, c.params.g, dim=0).item():.4f}  |g| {g.norm().item():.4f} (fp32 {c.params.g.norm().item():.4f})")
print("cos(fast, det)", cos(res['fast'][1], res['det'][1], dim=0).item(), " cos(fast, fast2)",
      cos(res['fast'][1], res['fast2'][1], dim=0).item(), " equal(det, det2)", torch.equal(res['det'][1], res['det2'][1]))
e = res["det"][2]
for key in ("fc.weight", "fc.bias", "layer4.2.conv3.weight", "layer4.2.bn3.weight", "layer4.0.conv2.weight", "layer3.0.conv1.weight",
            "layer2.0.downsample.0.weight", "layer1.0.conv1.weight", "layer1.0.bn1.weight", "bn1.weight", "conv1.weight"):
    gd, gf, gc = res["det"][2].params.grad(key).flatten(), res["fast"][2].params.grad(key).flatten(), c.params.grad(key).flatten()
    print(f"{key:32s} cos(det,fp32) {cos(gd, gc, dim=0).item():.4f}  cos(fast,fp32) {cos(gf, gc, dim=0).item():.4f}  cos(det,fast) {cos(gd, gf, dim=0).item():.4f}")
