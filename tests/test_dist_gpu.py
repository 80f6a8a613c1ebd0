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
    # SyncBN over 2 x 8 samples against plain BatchNorm over the 16, every gradient bucket on its own.
    # The two forwards are the same arithmetic on statistics that differ in the last fp32 bit; a handful
    # of bf16 roundings flip in the first block and the flips multiply layer by layer (measured with
    # bench/diag/syncbn_diag2.py: 0 differing elements after the stem, 11 after layer1.0, 87 % of the
    # tensor by layer4.1, batch statistics 0.5 % apart there), so the gradients of this random-init,
    # batch-16 network agree to 4 % (classifier) .. 18 % (stem) and no better.  A dropped or mis-scaled
    # bucket is an error of 50-100 % in that bucket; exactness of the reducer itself is tested on
    # integer data below, SyncBN in fp32 to 2e-3 in tests/test_distributed_cpu.py.
    errs = bucket_errors(r0["grads"], ref_grads, single.buckets)
    assert max(errs) < BUCKET_TOL and errs[0] < 0.08, errs
    assert abs(0.5 * (r0["loss"] + r1["loss"]).item() - loss.item()) < 2e-3 * abs(loss.item())
    n_early = single._bn_off["layer1.1.bn2"][0]          # statistics of the layers before the flips spread
    assert torch.allclose(r0["rm"][:n_early], single.running_mean.cpu()[:n_early], rtol=1e-3, atol=1e-5)
    assert torch.allclose(r0["rm"], single.running_mean.cpu(), rtol=5e-2, atol=2e-3)


BUCKET_TOL = 0.3


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
