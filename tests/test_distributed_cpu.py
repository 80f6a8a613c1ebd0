"""Distributed tier without GPUs: two gloo processes on localhost.

Covers the plumbing configuration "ResNet-18, world_size 2, gloo, synthetic data": data-parallel
equivalence of the bucketed reducer, SyncBatchNorm equivalence with single-process BatchNorm
on the concatenated batch, replica consistency after optimizer steps, rank-0-only writers,
validation schedule, checkpoint + resume and fail-fast behaviour of the launcher.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _engine(sync_bn, bucket_cap_mb=4.0, overlap=False, model="ResNet18"):
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec
    from pytorch_distributed_training_amd.ops import ref

    torch.manual_seed(7)
    eng = NativeResNet(get_spec(model, 10), torch.device("cpu"), ref, torch.float32, sync_bn=sync_bn,
                       bucket_cap_mb=bucket_cap_mb, first_bucket_mb=0.25, syncbn_overlap=overlap)
    eng.load_state_dict(get_model(model, 10).state_dict())
    return eng


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)


def _dp_worker(rank, world, port, sync_bn, out_dir, overlap=False):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from pytorch_distributed_training_amd.ops import ref
    from pytorch_distributed_training_amd.optim import FlatSGD
    from pytorch_distributed_training_amd.parallel.dist_env import check_replicas_in_sync

    eng = _engine(sync_bn, overlap=overlap)
    eng.broadcast_state(0)
    x, y = _data()
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    loss = eng.forward_backward(xs, ys)
    grads = eng.params.g.clone() * eng.reducer.grad_scale
    rm1, rv1 = eng.running_mean.clone(), eng.running_var.clone()      # after exactly one training forward
    opt = FlatSGD(eng.params, ref, lr=0.05, momentum=0.9, weight_decay=1e-4)
    opt.step(grad_scale=eng.reducer.grad_scale)
    eng.refresh_derived_weights()
    for _ in range(2):                                   # two more steps: replicas must stay identical
        eng.forward_backward(xs, ys)
        opt.step(grad_scale=eng.reducer.grad_scale)
        eng.refresh_derived_weights()
    in_sync = check_replicas_in_sync(eng.params.p)
    torch.save({"loss": loss, "grads": grads, "params": eng.params.p.clone(), "rm": rm1,
                "rv": rv1, "in_sync": in_sync, "launched": eng.reducer.stats["launched"],
                "n_buckets": len(eng.buckets)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn,overlap", [(True, False), (True, True), (False, False)])
def test_data_parallel_equivalence(tmp_path, sync_bn, overlap):
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, sync_bn, str(tmp_path), overlap), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # (a) every bucket was all-reduced each step, and replicas agree bit-for-bit
    assert r0["n_buckets"] >= 3 and r0["launched"] == 3 * r0["n_buckets"]
    assert r0["in_sync"] and r1["in_sync"]
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["params"], r1["params"])

    from pytorch_distributed_training_amd.ops import ref
    from pytorch_distributed_training_amd.optim import FlatSGD
    x, y = _data()
    torch.set_num_threads(4)
    if sync_bn:
        # (b)+(c) SyncBN over 2x4 samples == plain BN over the concatenated batch of 8
        single = _engine(False)
        loss = single.forward_backward(x, y)
        ref_grads = single.params.g.clone()
        assert torch.allclose(r0["rm"], single.running_mean, rtol=1e-4, atol=1e-6)
        assert torch.allclose(r0["rv"], single.running_var, rtol=1e-4, atol=1e-6)
        assert torch.equal(r0["rm"], r1["rm"]) and torch.equal(r0["rv"], r1["rv"])
        mean_loss = 0.5 * (r0["loss"] + r1["loss"])
        assert mean_loss.item() == pytest.approx(loss.item(), rel=1e-5)
    else:
        # plain DP: averaged gradient of two independent half-batch passes
        halves = []
        for r in range(2):
            e = _engine(False)
            e.forward_backward(x[r * 4:(r + 1) * 4], y[r * 4:(r + 1) * 4])
            halves.append(e.params.g.clone())
        ref_grads = 0.5 * (halves[0] + halves[1])
        assert not torch.equal(r0["rm"], r1["rm"])           # local statistics differ per rank
    err = (r0["grads"] - ref_grads).norm() / ref_grads.norm()
    assert err.item() < 2e-3, err.item()


def _fold_worker(rank, world, port, out_dir):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    x, y = _data()
    out = {}
    for tag, fold, fwd in (("F", "1", "1"), ("1", "1", "0"), ("0", "0", "0")):
        os.environ["PDT_BN_FOLD"], os.environ["PDT_FWD_FUSE"] = fold, fwd
        eng = _engine(True, model="ResNet50")
        assert eng._bn_fold == (fold == "1") and eng._bn_fold_product == (fold == "1") and eng._fwd_fuse == (fwd == "1")
        eng.broadcast_state(0)
        out["loss" + tag] = eng.forward_backward(x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4])
        out["grads" + tag] = eng.params.g.clone() * eng.reducer.grad_scale
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_syncbn_with_folded_batchnorm_backward(tmp_path):
    """Bottleneck network on two ranks: the BatchNorm backward of block tails and projection shortcuts is folded
    into their convolutions and takes its sums from the weight-gradient product; under SyncBN the local [s0, s1] of
    every folded unit are all-reduced.  Compared with the SAME two-rank run without the fold: the forward passes
    are identical, so what differs is backward rounding only.  (Against ONE process on the concatenated batch the
    gradients of this 53-BatchNorm network agree to about a per cent with or without the fold: at 8 samples its
    forward pass amplifies a 1e-7 change of the input into 1e-2 of the gradient.)  Third variant: block tails fused
    in forward as well (statistics from the Gram matrix, exchanged like any other) -- another rounding of the
    forward pass, hence compared by loss and direction."""
    port = _free_port()
    mp.spawn(_fold_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for tag in ("F", "1", "0"):
        assert torch.equal(r0["grads" + tag], r1["grads" + tag])            # replicas agree bit for bit
    assert r0["loss1"].item() == r0["loss0"].item()
    err = (r0["grads1"] - r0["grads0"]).norm() / r0["grads0"].norm()
    assert err.item() < 1e-4, err.item()
    assert r0["lossF"].item() == pytest.approx(r0["loss0"].item(), rel=1e-3)
    assert torch.nn.functional.cosine_similarity(r0["gradsF"], r0["grads0"], dim=0).item() > 0.99


def _trainer_cfg(tmp_path, extra=None):
    cfg = {
        "dataset": {"name": "synthetic", "n_classes": 10, "image_size": 32, "train_size": 64, "val_size": 24},
        "training": {
            "optimizer": {"name": "SGD", "lr": 0.02, "weight_decay": 1.0e-4, "momentum": 0.9},
            "lr_schedule": {"name": "multi_step", "milestones": [4], "gamma": 0.1},
            "train_iters": 6, "print_interval": 1, "val_interval": 3, "batch_size": 8, "num_workers": 0,
            "sync_bn": True, "precision": "fp32", "bucket_cap_mb": 8,
        },
        "validation": {"batch_size": 8, "num_workers": 0},
        "model": {"name": "ResNet18"},
    }
    if extra:
        cfg["training"].update(extra)
    import yaml
    path = os.path.join(tmp_path, "cfg.yml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def _run_cli(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "2"
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    if env_extra:
        env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "train_distributed.py"), *args], env=env,
                          capture_output=True, text=True, timeout=timeout)


def _scalars(log_dir, name="t"):
    path = os.path.join(log_dir, "tf-board-logs", name, "metrics.jsonl")
    with open(path) as f:
        return [json.loads(line) for line in f]


def test_cli_two_process_run_validation_checkpoint_resume(tmp_path):
    cfg = _trainer_cfg(str(tmp_path))
    log_dir = os.path.join(tmp_path, "run")
    common = ["--cfg-filepath", cfg, "--dist-backend", "gloo", "--multiprocessing", "--nproc-per-node", "2",
              "--num-nodes", "1", "--rank", "0", "--log-dir", log_dir, "--file-name-cfg", "t", "--seed", "3"]
    res = _run_cli([*common, "--dist-url", f"tcp://127.0.0.1:{_free_port()}"])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(os.path.join(log_dir, "t.log")).read()
    # upstream-format train lines from rank 0 only, LR drop at the milestone, validation at iters 2 and 5
    assert "Iter [0/6] Lr: [0.02] Loss:" in log and "Iter [5/6] Lr: [0.002" in log
    assert log.count("Start valuation") == 2 and log.count("Acc@1:") == 2
    assert "worker_rank_1] INFO: Iter" not in log
    sc = _scalars(log_dir)
    tags = {s["tag"] for s in sc}
    assert {"loss/train", "lr_group/0", "eval/Acc@1", "eval/Acc@5", "eval/loss"} <= tags
    assert sorted(s["step"] for s in sc if s["tag"] == "eval/Acc@1") == [2, 5]
    assert len([s for s in sc if s["tag"] == "loss/train"]) == 6
    ck_dir = os.path.join(log_dir, "checkpoints")
    assert sorted(os.listdir(ck_dir)) == ["iter_3.pth", "iter_6.pth"]

    # the checkpoint is a torchvision-schema state dict + optimizer/scheduler state
    from pytorch_distributed_training_amd import checkpoint as ckpt
    from pytorch_distributed_training_amd.models import get_model
    ck = ckpt.load_checkpoint(os.path.join(ck_dir, "iter_3.pth"))
    get_model("ResNet18", 10).load_state_dict(ck["model"], strict=True)
    assert ck["iter"] == 3 and ck["scheduler"]["last_iter"] == 3

    # resume from iteration 3 in a fresh log dir: iterations 3..5 only, and the loss curve continues
    log_dir2 = os.path.join(tmp_path, "run2")
    common2 = [a if a != log_dir else log_dir2 for a in common]
    res = _run_cli([*common2, "--dist-url", f"tcp://127.0.0.1:{_free_port()}", "--set",
                    f"training.resume={os.path.join(ck_dir, 'iter_3.pth')}"])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    sc2 = _scalars(log_dir2)
    steps2 = sorted(s["step"] for s in sc2 if s["tag"] == "loss/train")
    assert steps2 == [3, 4, 5]
    first = {s["step"]: s["value"] for s in sc if s["tag"] == "loss/train"}
    second = {s["step"]: s["value"] for s in sc2 if s["tag"] == "loss/train"}
    assert second[3] == pytest.approx(first[3], rel=2e-3)          # same weights + same data order
    lr2 = {s["step"]: s["value"] for s in sc2 if s["tag"] == "lr_group/0"}
    assert lr2[3] == pytest.approx(0.02) and lr2[4] == pytest.approx(0.002)


def test_eval_only_evaluates_a_checkpoint(tmp_path):
    """--eval-only: one validation pass over the weights in training.resume, no training iterations."""
    cfg = _trainer_cfg(str(tmp_path), {"train_iters": 2, "val_interval": 2, "sync_bn": False})
    log_dir = os.path.join(tmp_path, "run")
    base = ["--cfg-filepath", cfg, "--dist-backend", "gloo", "--file-name-cfg", "t"]
    res = _run_cli([*base, "--log-dir", log_dir])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    ck = os.path.join(log_dir, "checkpoints", "iter_2.pth")
    trained = [s for s in _scalars(log_dir) if s["tag"] == "eval/Acc@1"][-1]
    log_dir2 = os.path.join(tmp_path, "eval")
    res = _run_cli([*base, "--log-dir", log_dir2, "--eval-only", "--set", f"training.resume={ck}"])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(os.path.join(log_dir2, "t.log")).read()
    assert "Resumed from" in log and log.count("Acc@1:") == 1 and "Iter [" not in log
    sc = _scalars(log_dir2)
    assert {s["tag"] for s in sc} == {"eval/Acc@1", "eval/Acc@5", "eval/loss"}
    assert sc[0]["value"] == pytest.approx(trained["value"], abs=1e-6)       # same weights, same validation set
    assert not os.path.exists(os.path.join(log_dir2, "checkpoints"))


def test_fault_injection_fails_fast_and_keeps_logs(tmp_path):
    cfg = _trainer_cfg(str(tmp_path), {"train_iters": 50})
    log_dir = os.path.join(tmp_path, "run")
    res = _run_cli(["--cfg-filepath", cfg, "--dist-backend", "gloo", "--multiprocessing", "--nproc-per-node", "2",
                    "--num-nodes", "1", "--rank", "0", "--log-dir", log_dir, "--file-name-cfg", "t",
                    "--dist-url", f"tcp://127.0.0.1:{_free_port()}"],
                   env_extra={"FAULT_INJECT_ITER": "2", "FAULT_INJECT_RANK": "1"}, timeout=300)
    assert res.returncode != 0
    log = open(os.path.join(log_dir, "t.log")).read()          # log directory survives the crash
    assert "CRITICAL" in log and "injected fault at iteration 2" in log
    assert "Iter [1/50]" in log and "Iter [10/50]" not in log


def test_single_process_torch_engine_cli(tmp_path):
    """The stock-PyTorch engine goes through the same trainer (CPU, one process)."""
    cfg = _trainer_cfg(str(tmp_path), {"engine": "torch", "train_iters": 3, "val_interval": 2, "sync_bn": False})
    log_dir = os.path.join(tmp_path, "run")
    res = _run_cli(["--cfg-filepath", cfg, "--dist-backend", "gloo", "--log-dir", log_dir, "--file-name-cfg", "t"])
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(os.path.join(log_dir, "t.log")).read()
    assert "TorchEngine" in log and log.count("Acc@1:") == 2


def test_torchrun_env_launch(tmp_path):
    cfg = _trainer_cfg(str(tmp_path), {"train_iters": 2, "val_interval": 2})
    log_dir = os.path.join(tmp_path, "run")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "train_distributed.py"),
           "--cfg-filepath", cfg, "--dist-backend", "gloo", "--log-dir", log_dir, "--file-name-cfg", "t"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(os.path.join(log_dir, "t.log")).read()
    assert "world 2" in log and "Iter [1/2]" in log and "Acc@1:" in log
