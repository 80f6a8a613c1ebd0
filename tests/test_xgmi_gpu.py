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
                   "graph_identity": True}, out


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
                x.all_reduce(mine)
                out["status_after_missing_peer"] = x.status()
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
