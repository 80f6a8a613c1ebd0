"""GPU tier: a multi-rank step on ONE GPU with every kind of stream a real job has, on TWO hardware queues.

A rank submits rendezvous kernels (kernels that end only when the peer's matching kernel runs) on two streams: the
in-kernel SyncBN exchange on the compute stream and the bucket all-reduce on the communication side.  With fewer hardware
queues than streams, streams share queues; parallel/streams.py argues that no cycle of waits can form because all ranks
submit in the same order.  This test runs that situation for 50 steps: two ranks share the GPU, ``GPU_MAX_HW_QUEUES=2``,
weight-gradient stream, communication stream with the per-bucket optimizer, input copies from a prefetcher, SyncBN over
the exchange, and the gradient buckets through a second instance of the exchange on the communication stream -- RCCL
itself cannot take that part here, it refuses two ranks on one device ("Duplicate GPU detected"), so the stand-in is a
kernel with the same blocking behaviour.  A cycle would show as an exchange timeout (20 s, status word, CommFailure),
not as a hang.
"""
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

from test_dist_gpu import ROOT, _free_port

pytestmark = pytest.mark.gpu
STEPS = 50


def _worker(rank, port, out_dir):
    from pytorch_distributed_training_amd.config import validate_cfg
    from pytorch_distributed_training_amd.data.loader import DevicePrefetcher
    from pytorch_distributed_training_amd.engine.engines import build_engine
    from pytorch_distributed_training_amd.parallel import streams
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=2, rank=rank)
    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 100, "image_size": 64},
        "training": {"optimizer": {"name": "SGD", "lr": 0.02, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "constant"}, "train_iters": STEPS, "batch_size": 8, "sync_bn": True,
                     "syncbn_comm": "xgmi", "bucket_cap_mb": 4.0},
        "model": {"name": "ResNet18"},
        "runtime": {"step_replay": "list", "xgmi_timeout_s": 20},
    })
    torch.manual_seed(5)
    eng = build_engine(cfg, dev)
    calls0 = eng.model._bucket_xgmi.calls             # (the self-test's)
    g = torch.Generator().manual_seed(100 + rank)
    proto = torch.randn(10, 3, 64, 64, generator=torch.Generator().manual_seed(1))

    def batches():
        for _ in range(STEPS):
            label = torch.randint(0, 10, (8,), generator=g)
            yield (proto[label] + 0.3 * torch.randn(8, 3, 64, 64, generator=g)).pin_memory(), label.pin_memory()

    losses = [float(eng.train_step(img, label)) for img, label in DevicePrefetcher(batches(), dev)]
    torch.cuda.synchronize()
    m = eng.model
    ok = m.check_comm_health()
    torch.save({"losses": losses, "p": m.params.p.cpu(), "rm": m.running_mean.cpu(), "mode": m.active_replay_mode(),
                "comm": m.syncbn_comm, "launched": m.reducer.stats["launched"], "nb": len(m.buckets), "healthy": ok,
                "streams": sorted(streams.in_use(dev)), "n_streams": streams.count(dev),
                "bucket_calls": m._bucket_xgmi.calls - calls0, "in_backward": m.optimizer_in_backward,
                "wgrad_stream": m._wgrad_stream is not None}, os.path.join(out_dir, f"budget{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_two_hardware_queues_fifty_steps(tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env.update({"GPU_MAX_HW_QUEUES": "2", "PDT_XGMI_ALLOW_SHARED_GPU": "1", "PDT_BUCKETS_OVER_XGMI": "1"})
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", f"import test_stream_budget_gpu as t; t._worker({r}, {port}, {str(tmp_path)!r})"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=240)
            outs.append(out)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        raise AssertionError("the two ranks did not finish 50 steps within 240 s")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    r0, r1 = (torch.load(tmp_path / f"budget{r}.pt") for r in range(2))
    for r in (r0, r1):
        assert r["healthy"] and r["mode"] == "list" and r["comm"] == "xgmi" and r["wgrad_stream"] and r["in_backward"]
        assert r["launched"] == STEPS * r["nb"] == r["bucket_calls"] and r["nb"] >= 3
        # compute + wgrad + comm + copy (no RCCL stream in this arrangement): the budget, on half as many queues
        assert r["streams"] == ["comm", "compute", "copy", "wgrad"] and r["n_streams"] == 4
        assert all(l == l for l in r["losses"])
    # the replicas stayed bit-identical through 50 steps of exchanged statistics and summed buckets
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["rm"], r1["rm"])
    mean = lambda xs: sum(xs) / len(xs)
    assert mean(r0["losses"][-10:]) < mean(r0["losses"][:10])
