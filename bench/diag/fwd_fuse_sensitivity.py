import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["PDT_BN_FOLD_MIN_MB"] = "0"
os.environ["PDT_BN_FOLD_MAX_COUT"] = "0"
from pytorch_distributed_training_amd.engine.native_resnet import NativeResNet
from pytorch_distributed_training_amd.models import get_model, get_spec
from pytorch_distributed_training_amd.ops import hip
dev = torch.device("cuda", 0)
torch.manual_seed(5)
state = get_model("ResNet50", 10).state_dict()
for B, S in ((8, 64), (64, 96)):
    img = torch.randn(B, 3, S, S, device=dev); label = torch.randint(0, 10, (B,), device=dev)
    for fwd in ("0", "1"):
        os.environ["PDT_FWD_FUSE"] = fwd
        row = []
        for eps in (0.0, 4e-3, -4e-3, 8e-3):
            eng = NativeResNet(get_spec("ResNet50", 10), dev, hip, torch.bfloat16)
            eng.load_state_dict(state)
            row.append(round(float(eng.forward_backward(img * (1 + eps), label)), 4))
        print(f"batch {B} {S}px fwd_fuse={fwd}: loss at input scale 1, 1+4e-3, 1-4e-3, 1+8e-3:", row)
