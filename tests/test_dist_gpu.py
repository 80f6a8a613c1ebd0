"""GPU tier: the distributed code paths on ONE MI355X.

Real multi-GPU RCCL runs need a multi-GPU node; what can be checked on one GPU:
* two ranks sharing the GPU over gloo (GPU tensors), deterministic kernels, a conditioned network: the bucketed
  reducer gives the sum of the local gradients bit for bit; SyncBN over identical shards reproduces the one-process
  run bit for bit; SyncBN over different shards matches one process on the whole batch within what bf16 costs there
  and far closer than unreduced statistics;
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


def bucket_errors(grads, ref_grads, buckets):
    return [((grads[a:b] - ref_grads[a:b]).norm() / (ref_grads[a:b].norm() + 1e-12)).item() for a, b in buckets]


def _reducer_worker(rank, world, port, out_dir, backend):
    """The bucketed reducer alone, on integer-valued gradients (sums are exact in any order):
    overlapped and in-line launches against one plain all-reduce of the whole buffer."""
    from pytorch_distributed_training_amd.parallel.reducer import GradReducer
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    n = 3_000_000
    idx = torch.arange(n, device=dev)
    mine = ((idx * 7 + rank * 13) % 31 - 15).float()
    from pytorch_distributed_training_amd.parallel import collectives
    oracle = mine.clone()
    collectives.all_reduce(oracle)
    cuts = [0, 70_000, 1_000_064, 2_400_000, n]
    buckets = list(zip(cuts[:-1], cuts[1:]))
    out = {}
    for overlap in (True, False):
        flat = mine.clone()
        red = GradReducer(flat, buckets, overlap=overlap)
        hooked = []
        red.set_bucket_hook(lambda lo, hi: hooked.append((lo, hi)))
        red.begin_step()
        red.mark_ready(60_000)           # nothing complete yet
        red.mark_ready(1_000_064)        # two buckets
        assert red.completes_bucket(2_400_000) and not red.completes_bucket(2_000_000)
        red.finish()                     # the rest
        torch.cuda.synchronize()
        out[f"equal_{overlap}"] = bool(torch.equal(flat, oracle))
        out[f"launched_{overlap}"] = red.stats["launched"]
        out[f"hooked_{overlap}"] = hooked == buckets
    torch.save(out, os.path.join(out_dir, f"red{rank}.pt"))
    dist.destroy_process_group()


def _run_ranks(fn_call, world, tmp_path, timeout=120, extra_env=None):
    port = _free_port()
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, "-c", f"import test_dist_gpu as t; t.{fn_call.format(rank=r, port=port)}"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=timeout)
            outs.append(out)
    except subprocess.TimeoutExpired:
        for pr in procs:
            pr.kill()
        pytest.skip(f"{world} ranks sharing one GPU did not finish within {timeout} s on this host")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)[-4000:]
    return outs


# ------------------------------------------------------------------ exact and conditioned comparisons
def save_conditioned_case(path, world, per_rank=32, H=128):
    """ResNet-18 conditioned as in tests/test_engine_oracle_gpu.py (bf16 and fp32 agree tensor by tensor there), one probe
    batch of ``world * per_rank`` images: written once by the test process, read by every rank."""
    from test_engine_oracle_gpu import _batch, conditioned_state
    dev = torch.device("cuda", 0)
    spec, sd, proto = conditioned_state("ResNet18", dev, H, 64, 12)
    img, label = _batch(proto, world * per_rank, 999, torch.device("cpu"))
    torch.save({"sd": {k: v.cpu() for k, v in sd.items()}, "img": img, "label": label}, path)


def conditioned_engine(case, device, sync_bn, comm="auto"):
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_spec
    from pytorch_distributed_training_amd.ops import hip
    eng = NativeResNet(get_spec("ResNet18", 1000), device, hip, torch.bfloat16, sync_bn=sync_bn, bucket_cap_mb=4.0,
                       first_bucket_mb=0.25, syncbn_comm=comm)
    eng.load_state_dict(case["sd"])
    return eng


def conditioned_worker(rank, world, port, out_dir, sync_bn, comm, backend, device_index, duplicate=False):
    """One rank of a conditioned, DETERMINISTIC comparison: the local gradient of its shard (engine without a process
    group), then the same shard through the distributed engine.  ``duplicate``: every rank takes shard 0."""
    from pytorch_distributed_training_amd.ops import hip
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", device_index)
    torch.cuda.set_device(dev)
    hip.set_deterministic(True)
    case = torch.load(os.path.join(out_dir, "case.pt"))
    n = case["img"].shape[0] // world
    first = 0 if duplicate else rank * n
    xs, ys = case["img"][first:first + n].to(dev), case["label"][first:first + n].to(dev)
    solo = conditioned_engine(case, dev, False)
    solo.forward_backward(xs, ys)
    local, local_rm = solo.params.g.clone(), solo.running_mean.cpu()
    del solo
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank, **kw)
    eng = conditioned_engine(case, dev, sync_bn, comm)
    eng.broadcast_state(0)
    loss = eng.forward_backward(xs, ys)
    torch.cuda.synchronize()
    if backend == "nccl":
        on_dev = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(on_dev, local)
        gathered = [t.cpu() for t in on_dev]
    else:
        gathered = [torch.empty_like(local).cpu() for _ in range(world)]
        dist.all_gather(gathered, local.cpu())
    torch.save({"loss": loss.float().cpu(), "g_sum": eng.params.g.cpu(), "locals": gathered, "rm": eng.running_mean.cpu(),
                "local_rm": local_rm, "launched": eng.reducer.stats["launched"], "buckets": list(eng.buckets),
                "comm": eng.syncbn_comm if sync_bn else None}, os.path.join(out_dir, f"cond{rank}.pt"))
    dist.destroy_process_group()


def check_exact_buckets(res, world):
    """Without SyncBN the distributed engine does the local engine's work and sums: with deterministic kernels the result
    IS the sum of the local gradients -- bit for bit with two ranks (a + b has one order), to fp32 rounding of the
    summation order with more (a ring starts every chunk at another rank)."""
    r0 = res[0]
    assert r0["launched"] == len(r0["buckets"]) >= 3
    for r in res[1:]:
        assert torch.equal(r["g_sum"], r0["g_sum"])
    if world == 2:
        assert torch.equal(r0["g_sum"], r0["locals"][0] + r0["locals"][1])
    else:
        want = torch.stack(r0["locals"]).double().sum(0)
        errs = [((r0["g_sum"][a:b].double() - want[a:b]).norm() / (want[a:b].norm() + 1e-30)).item() for a, b in r0["buckets"]]
        assert max(errs) < 1e-6, errs


def check_syncbn_of_identical_shards(res, world):
    """Every rank holds the SAME shard: the statistics over the ranks are the shard's own (sums and counts both scale by
    the world size, a power of two), so SyncBN must reproduce the one-process run of that shard BIT FOR BIT -- forward
    statistics, running statistics, and every gradient (summed over the ranks: ``world`` times the local one).  A
    count taken per rank instead of globally, a sum applied twice or a payload read from the wrong slot shows here."""
    for r in res:
        assert torch.equal(r["g_sum"], r["locals"][0] * world), \
            bucket_errors(r["g_sum"] / world, r["locals"][0], r["buckets"])
        assert torch.equal(r["rm"], r["local_rm"])


def check_syncbn_against_single(res, case, world, tol=0.15):
    """SyncBN over ``world`` shards against plain BatchNorm over the whole batch in one process, on the conditioned
    network.  The two runs use statistics that differ in the last fp32 bit (another summation order), so bf16 roundings
    flip here and there: on this network that noise is what bf16 itself costs against fp32 (cosine 0.995 = 10 % in the
    norm, profiles/oracle_conditioning_r3.txt), which bounds how tight this can be: every bucket within ``tol`` AND at
    most 0.6 of what per-rank BatchNorm (statistics NOT reduced) scores on the same data (measured: 0.3-0.5 of it; the
    conditioning itself differs a little from run to run, worst bucket 0.08-0.12).
    The exact half of the claim is check_syncbn_of_identical_shards."""
    from pytorch_distributed_training_amd.ops import hip
    r0 = res[0]
    for r in res[1:]:
        assert torch.equal(r["g_sum"], r0["g_sum"]) and torch.equal(r["rm"], r0["rm"])
    dev = torch.device("cuda", 0)
    hip.set_deterministic(True)
    try:
        single = conditioned_engine(case, dev, False)
        loss = single.forward_backward(case["img"].to(dev), case["label"].to(dev))
        want = single.params.g.cpu()
    finally:
        hip.set_deterministic(False)
    errs = bucket_errors(r0["g_sum"] / world, want, r0["buckets"])
    unsynced = bucket_errors(torch.stack(r0["locals"]).sum(0) / world, want, r0["buckets"])
    print("per-bucket error against one process: SyncBN over", world, "ranks", [f"{e:.4f}" for e in errs],
          "-- per-rank BatchNorm", [f"{e:.4f}" for e in unsynced])
    assert max(errs) < tol, errs
    assert all(e < 0.6 * u for e, u in zip(errs, unsynced)), (errs, unsynced)
    mean_loss = sum(r["loss"].item() for r in res) / world
    assert abs(mean_loss - loss.item()) < 1e-2 * abs(loss.item())
    assert torch.allclose(r0["rm"], single.running_mean.cpu(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("mode", ["buckets", "syncbn", "syncbn_identical_shards"])
def test_two_ranks_one_gpu_conditioned_exact(tmp_path, mode):
    save_conditioned_case(tmp_path / "case.pt", 2)
    sync_bn, dup = mode != "buckets", mode == "syncbn_identical_shards"
    _run_ranks(f"conditioned_worker({{rank}}, 2, {{port}}, {str(tmp_path)!r}, {sync_bn}, 'auto', 'gloo', 0, {dup})", 2,
               tmp_path, timeout=200)
    res = [torch.load(tmp_path / f"cond{r}.pt") for r in range(2)]
    if mode == "buckets":
        check_exact_buckets(res, 2)
    elif dup:
        check_syncbn_of_identical_shards(res, 2)
    else:
        check_syncbn_against_single(res, torch.load(tmp_path / "case.pt"), 2)


def test_bucketed_reducer_is_exact_two_ranks_one_gpu(tmp_path):
    _run_ranks(f"_reducer_worker({{rank}}, 2, {{port}}, {str(tmp_path)!r}, 'gloo')", 2, tmp_path)
    for r in range(2):
        res = torch.load(tmp_path / f"red{r}.pt")
        assert res == {"equal_True": True, "launched_True": 4, "hooked_True": True,
                       "equal_False": True, "launched_False": 4, "hooked_False": True}, res


def test_bucketed_reducer_is_exact_over_rccl_one_rank(tmp_path):
    """Forced one-rank RCCL group: the communication stream, events and RCCL launches are real."""
    _run_ranks(f"_reducer_worker({{rank}}, 1, {{port}}, {str(tmp_path)!r}, 'nccl')", 1, tmp_path,
               extra_env={"PDT_FORCE_COLLECTIVES": "1"})
    res = torch.load(tmp_path / "red0.pt")
    assert res["equal_True"] and res["equal_False"] and res["launched_True"] == 4 and res["hooked_True"]


def _engine_steps_worker(port, out_dir, syncbn_overlap, replay, steps=5):
    """One rank, forced collectives over RCCL, the full engine + optimizer for a few steps."""
    from pytorch_distributed_training_amd.config import validate_cfg
    from pytorch_distributed_training_amd.engine.engines import build_engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 100, "image_size": 64},
        "training": {"optimizer": {"name": "SGD", "lr": 0.05, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "constant"}, "train_iters": steps, "batch_size": 16, "sync_bn": True,
                     "syncbn_comm": "rccl", "syncbn_overlap": bool(syncbn_overlap), "bucket_cap_mb": 4.0},
        "model": {"name": "ResNet18"},
        "runtime": {"step_replay": replay},
    })
    torch.manual_seed(5)
    eng = build_engine(cfg, dev)
    x, y = _data()
    x, y = x.cuda(), y.cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(steps)]
    torch.cuda.synchronize()
    m = eng.model
    torch.save({"losses": losses, "mode": m.active_replay_mode(), "sync_bn": m.sync_bn, "comm": m.syncbn_comm,
                "launched": m.reducer.stats["launched"], "nb": len(m.buckets), "opt_steps": eng.optimizer.steps,
                "in_backward": m.optimizer_in_backward, "finite": bool(torch.isfinite(m.params.p).all())},
               os.path.join(out_dir, f"steps_{int(syncbn_overlap)}_{replay}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("syncbn_overlap", [False, True])
def test_rccl_syncbn_under_list_replay(tmp_path, syncbn_overlap):
    """SyncBN over RCCL (with and without the asynchronous backward all-reduce, whose wait() returns
    True) recorded into a command list and replayed: the replayed steps must run and train."""
    _run_ranks(f"_engine_steps_worker({{port}}, {str(tmp_path)!r}, {syncbn_overlap}, 'list')", 1, tmp_path,
               timeout=300, extra_env={"PDT_FORCE_COLLECTIVES": "1"})
    res = torch.load(tmp_path / f"steps_{int(syncbn_overlap)}_list.pt")
    assert res["mode"] == "list" and res["sync_bn"] and res["comm"] == "rccl" and res["in_backward"]
    assert res["launched"] == 5 * res["nb"] and res["opt_steps"] == 5 and res["finite"]
    assert res["losses"][-1] < res["losses"][0]                  # same batch five times: the loss must fall


def _overlap_worker(port, out_dir, overlap_comm, in_backward, steps=4):
    """One rank, forced collectives over RCCL, deterministic kernels: the whole training step with the gradient
    buckets reduced on the communication stream (overlapped with backward) or in line."""
    from pytorch_distributed_training_amd.config import validate_cfg
    from pytorch_distributed_training_amd.engine.engines import build_engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 100, "image_size": 64},
        "training": {"optimizer": {"name": "SGD", "lr": 0.05, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "constant"}, "train_iters": steps, "batch_size": 16, "sync_bn": True,
                     "syncbn_comm": "rccl", "bucket_cap_mb": 4.0, "overlap_comm": bool(overlap_comm),
                     "optimizer_in_backward": bool(in_backward)},
        "model": {"name": "ResNet18"},
        "runtime": {"deterministic": True},
    })
    torch.manual_seed(5)
    eng = build_engine(cfg, dev)
    x, y = _data()
    x, y = x.cuda(), y.cuda()
    losses = [float(eng.train_step(x, y)) for _ in range(steps)]
    torch.cuda.synchronize()
    m = eng.model
    torch.save({"losses": losses, "p": m.params.p.cpu(), "rm": m.running_mean.cpu(), "launched": m.reducer.stats["launched"],
                "overlap": m.reducer.overlap, "nb": len(m.buckets)},
               os.path.join(out_dir, f"overlap_{int(overlap_comm)}_{int(in_backward)}.pt"))
    dist.destroy_process_group()


def test_overlapped_and_inline_gradient_reduction_train_bit_identically(tmp_path):
    """``training.overlap_comm`` on / off and the per-bucket optimizer on / off change WHEN buckets are reduced and
    updated, never what is computed: with the deterministic kernels the parameters after four steps are equal bit for
    bit (a bucket reduced before its gradients were final, or updated twice, would show here)."""
    res = {}
    for overlap, in_bwd in ((1, 1), (0, 1), (1, 0)):
        _run_ranks(f"_overlap_worker({{port}}, {str(tmp_path)!r}, {overlap}, {in_bwd})", 1, tmp_path, timeout=170,
                   extra_env={"PDT_FORCE_COLLECTIVES": "1"})
        res[(overlap, in_bwd)] = torch.load(tmp_path / f"overlap_{overlap}_{in_bwd}.pt")
    a, b, c = res[(1, 1)], res[(0, 1)], res[(1, 0)]
    assert a["overlap"] and not b["overlap"] and a["launched"] == b["launched"] == 4 * a["nb"]
    assert a["losses"] == b["losses"] == c["losses"], (a["losses"], b["losses"], c["losses"])
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["p"], c["p"])
    assert torch.equal(a["rm"], b["rm"]) and torch.equal(a["rm"], c["rm"])


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


def test_bench_self_launch_two_ranks_share_the_gpu():
    """``python bench.py --gpus 2`` from a plain shell (no launcher, no RANK / WORLD_SIZE): the script
    starts its two ranks itself; with one GPU visible they share it (gloo carries the gradient
    buckets, the SyncBN statistics go through the IPC exchange or gloo)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
           "--warmup", "3", "--batch", "16"]
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        pytest.skip("two ranks sharing one GPU did not finish within 420 s on this host")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and cfg["rccl_world"] == 2 and cfg["global_batch"] == 32 and rec["value"] > 0
    assert cfg["sync_bn_active"] and cfg["syncbn_comm"] in ("xgmi", "rccl") and cfg["comm_healthy"]
    assert cfg["grad_buckets_reduced_per_step"] == cfg["grad_buckets"] >= 4
    assert cfg["exposed_comm_ms_per_step"] is not None and cfg["optimizer_in_backward"]
