"""Multi-GPU tier (``-m multigpu``): real RCCL / xGMI runs, one process per GPU.

Skipped unless at least two GPUs are visible.  World sizes 2, 4 and 8 are tried up to the number
of GPUs on the node.  Covered: the bucketed reducer against a plain ``dist.all_reduce`` of the
per-rank gradients, bit-identical replicas, SyncBatchNorm over the one-kernel xGMI exchange and
over RCCL, and ``bench.py`` under ``torch.distributed.run``.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = [pytest.mark.gpu, pytest.mark.multigpu]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PER_RANK = 8


def _n_gpus() -> int:
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


def _worlds():
    return [w for w in (2, 4, 8) if w <= _n_gpus()] or [2]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _engine(device, sync_bn, comm="auto"):
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec
    from pytorch_distributed_training_amd.ops import hip
    torch.manual_seed(7)
    eng = NativeResNet(get_spec("ResNet18", 100), device, hip, torch.bfloat16, sync_bn=sync_bn, bucket_cap_mb=4.0,
                       first_bucket_mb=0.25, syncbn_comm=comm)
    eng.load_state_dict(get_model("ResNet18", 100).state_dict())
    return eng


def _data(world):
    g = torch.Generator().manual_seed(11)
    n = PER_RANK * world
    return torch.randn(n, 3, 64, 64, generator=g), torch.randint(0, 100, (n,), generator=g)


def _worker(rank, world, port, out_dir, sync_bn, comm):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    x, y = _data(world)
    xs = x[rank * PER_RANK:(rank + 1) * PER_RANK].to(device)
    ys = y[rank * PER_RANK:(rank + 1) * PER_RANK].to(device)
    local = None
    if not sync_bn:
        # before the process group exists the engine has no reducer: purely local gradients
        solo = _engine(device, False)
        solo.forward_backward(xs, ys)
        local = solo.params.g.clone()
        del solo
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank,
                            device_id=device)
    eng = _engine(device, sync_bn, comm)
    eng.broadcast_state(0)
    loss = eng.forward_backward(xs, ys)
    torch.cuda.synchronize()
    out = {"loss": loss.float().cpu(), "grads": (eng.params.g * eng.reducer.grad_scale).cpu(),
           "rm": eng.running_mean.cpu(), "launched": eng.reducer.stats["launched"], "n_buckets": len(eng.buckets),
           "buckets": list(eng.buckets),
           "comm": eng.syncbn_comm if sync_bn else None}
    if local is not None:
        dist.all_reduce(local)
        out["oracle"] = (local / world).cpu()
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _spawn(world, tmp_path, sync_bn, comm):
    port = _free_port()
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, "-c",
                               f"import test_multigpu as t; "
                               f"t._worker({r}, {world}, {port}, {str(tmp_path)!r}, {sync_bn!r}, {comm!r})"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=300)
            outs.append(out)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        raise AssertionError(f"{world} ranks did not finish within 300 s")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    return [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs")


def _spawn_conditioned(world, tmp_path, sync_bn, comm, duplicate=False):
    import test_dist_gpu as one
    one.save_conditioned_case(tmp_path / "case.pt", world)
    port = _free_port()
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, "-c",
                               f"import test_dist_gpu as t; t.conditioned_worker({r}, {world}, {port}, {str(tmp_path)!r}, "
                               f"{sync_bn!r}, {comm!r}, 'nccl', {r}, {duplicate!r})"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=300)[0])
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        raise AssertionError(f"{world} ranks did not finish within 300 s")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    return [torch.load(os.path.join(tmp_path, f"cond{r}.pt")) for r in range(world)]


@needs_two
@pytest.mark.parametrize("world", _worlds())
def test_bucketed_reducer_matches_all_reduce(tmp_path, world):
    """Deterministic kernels, no SyncBN: the reduced gradient IS the sum of the ranks' local gradients (bit for bit with
    two ranks, to the fp32 rounding of RCCL's summation order with more)."""
    import test_dist_gpu as one
    res = _spawn_conditioned(world, tmp_path, False, "auto")
    one.check_exact_buckets(res, world)
    assert not torch.equal(res[0]["rm"], res[1]["rm"])            # local BatchNorm statistics per rank


def bucket_errors(grads, ref_grads, buckets):
    return [((grads[a:b] - ref_grads[a:b]).norm() / (ref_grads[a:b].norm() + 1e-12)).item() for a, b in buckets]


def _reducer_worker(rank, world, port, out_dir):
    """Integer-valued gradients: the bucketed, overlapped reducer must equal one plain all-reduce
    of the whole buffer bit for bit, whatever order RCCL sums the ranks in."""
    from pytorch_distributed_training_amd.parallel.reducer import GradReducer
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank,
                            device_id=device)
    n = 30_000_000
    idx = torch.arange(n, device=device)
    mine = ((idx * 7 + rank * 13) % 31 - 15).float()
    oracle = mine.clone()
    dist.all_reduce(oracle)
    cuts = [0, 262_144, 6_000_000, 12_500_032, 29_000_000, n]
    buckets = list(zip(cuts[:-1], cuts[1:]))
    out = {}
    for overlap in (True, False):
        flat = mine.clone()
        red = GradReducer(flat, buckets, overlap=overlap)
        red.begin_step()
        for end in cuts[1:-1]:
            red.mark_ready(end)
        red.finish()
        torch.cuda.synchronize()
        out[f"equal_{overlap}"] = bool(torch.equal(flat, oracle))
    torch.save(out, os.path.join(out_dir, f"red{rank}.pt"))
    dist.destroy_process_group()


@needs_two
@pytest.mark.parametrize("world", _worlds())
def test_reducer_is_exact_against_plain_all_reduce(tmp_path, world):
    port = _free_port()
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, "-c",
                               f"import test_multigpu as t; t._reducer_worker({r}, {world}, {port}, {str(tmp_path)!r})"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [pr.communicate(timeout=300)[0] for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"red{r}.pt"))
        assert res == {"equal_True": True, "equal_False": True}, (r, res)


@needs_two
@pytest.mark.parametrize("world", _worlds())
@pytest.mark.parametrize("comm", ["xgmi", "rccl"])
def test_syncbn_equals_single_process_batchnorm(tmp_path, world, comm):
    """On the conditioned network: SyncBN over the ranks against BatchNorm over the whole batch (bounded by what bf16 costs
    there and by what unreduced statistics would score), and -- identical shards on every rank -- bit for bit."""
    import test_dist_gpu as one
    res = _spawn_conditioned(world, tmp_path, True, comm)
    assert res[0]["comm"] == comm
    one.check_syncbn_against_single(res, torch.load(tmp_path / "case.pt"), world)
    res = _spawn_conditioned(world, tmp_path, True, comm, duplicate=True)
    one.check_syncbn_of_identical_shards(res, world)


@needs_two
@pytest.mark.parametrize("world", _worlds())
def test_bench_from_a_plain_shell(world):
    """``python bench.py --gpus N`` with no launcher: the script starts its ranks itself."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "4",
           "--batch", "64"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == world and cfg["rccl_world"] == world and cfg["backend"] == "nccl"
    assert cfg["sync_bn_active"] and cfg["comm_healthy"] and cfg["grad_buckets_reduced_per_step"] == cfg["grad_buckets"]
    assert cfg["exposed_comm_ms_per_step"] is not None and rec["value"] > 0


@needs_two
@pytest.mark.parametrize("world", _worlds())
def test_bench_under_torchrun(world):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "6", "--warmup", "4", "--batch", "64"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    rec = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == world and rec["config"]["global_batch"] == 64 * world
    assert rec["config"]["sync_bn_active"] and rec["config"]["syncbn_comm"] in ("xgmi", "rccl")
    assert rec["config"]["grad_buckets_reduced_per_step"] >= 4 and rec["value"] > 0
