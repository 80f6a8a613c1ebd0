"""GPU tier: the one-kernel xGMI all-reduce used for SyncBN statistics.

A one-GPU box cannot show traffic over xGMI links, but it can run the whole protocol: IPC handle
exchange, peer mappings, flag/slot reuse and the bounded wait, with two processes that share the
GPU (their kernels run concurrently on separate hardware queues).  Ranks are subprocesses under a
hard time limit; a stall skips instead of hanging the run.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(extra=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["PDT_XGMI_ALLOW_SHARED_GPU"] = "1"       # the ranks of these tests share the one GPU on purpose
    env.update(extra or {})
    return env


def _one_rank_worker():
    import torch.distributed as dist
    from pytorch_distributed_training_amd.parallel.xgmi_allreduce import XgmiAllReduce
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    x = XgmiAllReduce.create(None, dev)
    out = {"created": x is not None}
    if x is not None:
        out["self_test"] = x.self_test()
        t = torch.arange(4096, dtype=torch.float32, device=dev)
        ref = t.clone()
        for _ in range(5):
            x.all_reduce(t)
        out["identity"] = bool(torch.equal(t, ref))
        out["status"] = x.status()
        out["healthy"] = x.healthy()
        # replayable from a HIP graph: the sequence number lives on the device
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode="thread_local")
            x.all_reduce(t)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(7):
            g.replay()
        torch.cuda.synchronize()
        out["graph_identity"] = bool(torch.equal(t, ref)) and x.status() == 0
        # a payload whose checksum does not match what was read is refused: NaN result, status 2 (not a timeout)
        clean = x.rereads()                              # nothing had to be read twice so far
        x._lib.pdt_car_corrupt_checksum(x._h, 0x10)
        bad = torch.ones(512, device=dev)
        x.all_reduce(bad)
        out["checksum_refused"] = bool(torch.isnan(bad).all()) and x.status() == 2 and x.peek() == 2
        out["rereads"] = [clean, x.rereads()]            # ... and the refused payload was re-read the bounded number of times
        x.close()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


def test_single_rank_protocol_and_graph_replay():
    res = subprocess.run([sys.executable, "-c", "import test_xgmi_gpu as t; t._one_rank_worker()"],
                         env=_env({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out == {"created": True, "self_test": True, "identity": True, "status": 0, "healthy": True,
                   "graph_identity": True, "checksum_refused": True, "rereads": [0, 65]}, out


def _two_rank_worker(rank, port, provoke_timeout):
    import torch.distributed as dist
    from pytorch_distributed_training_amd.parallel.xgmi_allreduce import XgmiAllReduce
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=2, rank=rank)
    x = XgmiAllReduce.create(None, dev)
    out = {"created": x is not None}
    if x is not None:
        out["self_test"] = x.self_test()
        mine = torch.full((2048,), float(rank + 1), device=dev)
        x.all_reduce(mine)
        out["sum_ok"] = bool(torch.equal(mine, torch.full((2048,), 3.0, device=dev)))
        out["status"] = x.status()
        dist.barrier()
        if provoke_timeout:
            # rank 1 stays away from one call: rank 0's wait must end by itself and be reported
            if rank == 0:
                x._lib.pdt_car_set_timeout_ms(x._h, 200)
                out["peek_before"] = x.peek()
                x.all_reduce(mine)
                out["status_after_missing_peer"] = x.status()          # synchronises
                out["peek_after"] = x.peek()                           # the pinned host word, no GPU call
                out["poisoned"] = bool(torch.isnan(mine).all())        # a defined result, never stale memory
            dist.barrier()
        x.close()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


def _run_pair(provoke_timeout):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c",
                               f"import test_xgmi_gpu as t; t._two_rank_worker({r}, {port}, {provoke_timeout})"],
                              env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    try:
        for pr in procs:
            o, _ = pr.communicate(timeout=120)
            outs.append(o)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        pytest.skip("two ranks sharing one GPU did not finish within 120 s on this host")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    return [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]) for o in outs]


def test_two_processes_share_one_gpu():
    r0, r1 = _run_pair(False)
    if not r0["created"]:
        pytest.skip("IPC peer mapping is not available between two processes on this host")
    for r in (r0, r1):
        assert r["self_test"] and r["sum_ok"] and r["status"] == 0, (r0, r1)


def test_missing_peer_times_out_instead_of_hanging():
    r0, r1 = _run_pair(True)
    if not r0["created"]:
        pytest.skip("IPC peer mapping is not available between two processes on this host")
    assert r0["self_test"] and r0["status"] == 0
    assert r0["status_after_missing_peer"] == 1, r0
    assert r0["peek_before"] == 0 and r0["peek_after"] == 1 and r0["poisoned"], r0


def _engine_timeout_worker(rank, port):
    """Two ranks train with SyncBN over the in-kernel exchange; rank 1 stalls before its third step
    for longer than the exchange's timeout.  Rank 0 must end with CommFailure, not train on."""
    import time
    import torch.distributed as dist
    from pytorch_distributed_training_amd.config import validate_cfg
    from pytorch_distributed_training_amd.engine.engines import build_engine
    from pytorch_distributed_training_amd.engine.native_resnet import CommFailure
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=2, rank=rank)
    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 100, "image_size": 64},
        "training": {"optimizer": {"name": "SGD", "lr": 0.05, "momentum": 0.9},
                     "lr_schedule": {"name": "constant"}, "train_iters": 8, "batch_size": 16, "sync_bn": True,
                     "syncbn_comm": "xgmi", "bucket_cap_mb": 4.0},
        "model": {"name": "ResNet18"},
        "runtime": {"xgmi_timeout_s": 1.0},
    })
    torch.manual_seed(5)
    eng = build_engine(cfg, dev)
    # no host-blocking gradient all-reduce between the ranks in this test (gloo would stage GPU tensors
    # through the host while the peer's exchange kernel spins): the statistics exchange is the subject
    eng.model.reducer.collective = False
    out = {"comm": eng.model.syncbn_comm, "raised": False, "steps": 0}
    if eng.model.syncbn_comm == "xgmi":
        g = torch.Generator().manual_seed(rank)
        x = torch.randn(8, 3, 64, 64, generator=g).cuda()
        y = torch.randint(0, 100, (8,), generator=g).cuda()
        try:
            for i in range(6):
                if rank == 1 and i == 2:
                    torch.cuda.synchronize()
                    time.sleep(4.0)
                eng.train_step(x, y)
                torch.cuda.synchronize()
                eng.model.raise_on_comm_failure()
                out["steps"] = i + 1
        except CommFailure as exc:
            out["raised"] = True
            out["message"] = str(exc)
        out["params_finite"] = bool(torch.isfinite(eng.model.params.p).all())
    print("RESULT " + json.dumps(out), flush=True)
    os._exit(0)          # no collective teardown: the peer may be gone, as after a real failure


def test_exchange_timeout_ends_training_loudly():
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", f"import test_xgmi_gpu as t; t._engine_timeout_worker({r}, {port})"],
                              env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    try:
        for pr in procs:
            o, _ = pr.communicate(timeout=240)
            outs.append(o)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        pytest.skip("two ranks sharing one GPU did not finish within 240 s on this host")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    r0, r1 = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]) for o in outs]
    if r0["comm"] != "xgmi":
        pytest.skip("IPC peer mapping is not available between two processes on this host")
    # the rank that waited in vain stops with an error within a step of the stall, and what it holds is
    # recognisably poisoned (NaN), not a plausible-looking wrong model
    assert r0["raised"] and r0["steps"] <= 3 and "checkpoint" in r0["message"], (r0, r1)
    assert not r0["params_finite"], r0
