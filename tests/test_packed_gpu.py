"""GPU tier: training from packed records end to end -- native gather threads into the pinned ring, asynchronous upload
with release events, crop / resize / flip in the HIP kernel behind it, the engine, validation, resume."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

from test_packed_cpu import _image_folder

from pytorch_distributed_training_amd.data import DevicePrefetcher, GpuAugment, build_loader, get_dataset, make_iter_dataloader
from pytorch_distributed_training_amd.data import native, pack_image_folder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_loader_feeds_the_gpu_what_the_dataloader_feeds(tmp_path, device):
    """Many batches through the pinned ring while the consumer never synchronises: every uploaded batch equals the
    DataLoader path's (a buffer rewritten before its copy had finished would show as a mismatch)."""
    src = _image_folder(tmp_path / "raw", n_per_class=24)
    out = tmp_path / "packed"
    pack_image_folder(str(src), "train", str(out), side=64, workers=2)
    ds = get_dataset("packed", str(out), "train", image_size=32, fmt="uint8_nhwc")
    fast = build_loader(ds, 4, 4, True, 1, 0, seed=9)
    slow = build_loader(ds, 4, 0, True, 1, 0, seed=9, native=False)
    assert isinstance(fast, native.NativeRecordLoader) and fast.pin
    a = DevicePrefetcher(make_iter_dataloader(fast), device)
    b = DevicePrefetcher(make_iter_dataloader(slow), device)
    got = [next(a) for _ in range(40)]                   # 40 batches through a ring of 5, no host synchronisation
    want = [next(b) for _ in range(40)]
    torch.cuda.synchronize()
    for (xa, ya), (xb, yb) in zip(got, want):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    fast.close()


def test_cli_trains_from_packed_records_on_the_gpu(tmp_path):
    src = _image_folder(tmp_path / "raw", n_per_class=16)
    out = tmp_path / "packed"
    for split in ("train", "val"):
        pack_image_folder(str(src), split, str(out), side=80, workers=2)
    cfg = {
        "dataset": {"name": "packed", "root": str(out), "n_classes": 2, "image_size": 64, "input_format": "uint8_nhwc"},
        "training": {"optimizer": {"name": "SGD", "lr": 0.02, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "multi_step", "milestones": [6], "gamma": 0.1}, "train_iters": 8,
                     "print_interval": 1, "val_interval": 4, "batch_size": 8, "num_workers": 3, "sync_bn": False,
                     "checkpoint_interval": 4},
        "validation": {"batch_size": 8, "num_workers": 0},
        "model": {"name": "ResNet18"},
    }
    path = tmp_path / "cfg.yml"
    path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    log_dir = tmp_path / "run"
    cmd = [sys.executable, os.path.join(ROOT, "train_distributed.py"), "--cfg-filepath", str(path), "--log-dir", str(log_dir),
           "--file-name-cfg", "t", "--seed", "3", "--dist-url", "tcp://127.0.0.1:29641"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(log_dir / "t.log").read()
    assert "Iter [7/8]" in log and log.count("Acc@1:") >= 1
    # resume from the checkpoint in the middle: the run continues at iteration 4 and ends like the first one did
    res = subprocess.run(cmd + ["--set", f"training.resume={log_dir / 'checkpoints' / 'iter_4.pth'}", "--set", "training.train_iters=8"],
                         env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
