"""GPU tier: the distributed code paths on ONE MI355X.

Real multi-GPU RCCL runs need a multi-GPU node; what can be checked on one GPU:
* two ranks sharing the GPU over gloo (GPU tensors): SyncBN + bucketed reducer with the side
  stream / events give the same gradients as one process on the concatenated batch;
* a one-rank RCCL group with PDT_FORCE_COLLECTIVES=1: every SyncBN all-reduce and every
  gradient bucket really goes through RCCL on the communication stream, and the result is
  unchanged.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _engine(sync_bn):
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec
    from pytorch_distributed_training_amd.ops import hip
    torch.manual_seed(7)
    dev = torch.device("cuda", 0)
    eng = NativeResNet(get_spec("ResNet18", 100), dev, hip, torch.bfloat16, sync_bn=sync_bn, bucket_cap_mb=4.0,
                       first_bucket_mb=0.25)
    eng.load_state_dict(get_model("ResNet18", 100).state_dict())
    return eng


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(16, 3, 64, 64, generator=g), torch.randint(0, 100, (16,), generator=g)


def _worker(rank, world, port, out_dir):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    eng = _engine(True)
    eng.broadcast_state(0)
    x, y = _data()
    xs = x[rank * 8:(rank + 1) * 8].cuda()
    ys = y[rank * 8:(rank + 1) * 8].cuda()
    loss = eng.forward_backward(xs, ys)
    torch.cuda.synchronize()
    torch.save({"loss": loss.cpu(), "grads": (eng.params.g * eng.reducer.grad_scale).cpu(),
                "rm": eng.running_mean.cpu(), "launched": eng.reducer.stats["launched"],
                "n_buckets": len(eng.buckets)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo_syncbn_equivalence(tmp_path):
    # Each rank is its own interpreter under a hard time limit: two processes contending for one
    # GPU through gloo's host-staged copies is an arrangement that can stall on some hosts, and a
    # stall must not take the whole GPU test run with it.
    port = _free_port()
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, "-c",
                               f"import test_dist_gpu as t; t._worker({r}, 2, {port}, {str(tmp_path)!r})"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=90)
            outs.append(out)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        pytest.skip("two gloo ranks sharing one GPU did not finish within 90 s on this host")
    if any(pr.returncode != 0 for pr in procs):
        joined = "\n".join(outs)
        if "gloo" in joined.lower() and ("cuda" in joined.lower() or "hip" in joined.lower()) and "support" in joined.lower():
            pytest.skip("gloo cannot reduce GPU tensors on this build")
        raise AssertionError(joined[-3000:])
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["launched"] == r0["n_buckets"] >= 3
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["rm"], r1["rm"])
    single = _engine(False)
    x, y = _data()
    loss = single.forward_backward(x.cuda(), y.cuda())
    ref_grads = single.params.g.cpu()
    cos = torch.nn.functional.cosine_similarity(r0["grads"], ref_grads, dim=0).item()
    assert cos > 0.97, cos            # bf16: different tile/summation order, same maths
    assert abs(0.5 * (r0["loss"] + r1["loss"]).item() - loss.item()) < 0.02 * abs(loss.item())
    assert torch.allclose(r0["rm"], single.running_mean.cpu(), rtol=5e-2, atol=5e-3)


def _bench(env_extra, launcher):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "32",
            "--model", "ResNet50"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_forced_collectives_over_rccl_single_rank():
    plain = _bench({}, launcher=False)
    forced = _bench({"PDT_FORCE_COLLECTIVES": "1"}, launcher=True)
    assert forced["n_gpus"] == 1 and forced["value"] > 0
    assert forced["config"]["sync_bn_active"] and forced["config"]["grad_buckets_reduced_per_step"] >= 4
    assert not plain["config"]["sync_bn_active"] and plain["config"]["grad_buckets_reduced_per_step"] == 0
    # (random labels at lr 0.1 are chaotic after a few steps, so losses are only checked for sanity)
    assert 0.0 < forced["config"]["final_loss"] < 1e3 and 0.0 < plain["config"]["final_loss"] < 1e3
    assert forced["value"] > 0.4 * plain["value"]
