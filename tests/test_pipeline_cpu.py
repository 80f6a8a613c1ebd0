"""CPU tier: the pieces around the step that need no GPU.

* ``bench.py``'s own multi-rank launcher (environment of the ranks, fail-fast on a dead rank);
* the per-bucket optimizer update enqueued from the gradient reducer against the classic
  whole-model update after backward (bit-identical parameters), alone and over gloo world 2;
* bf16 gradient transport; start-up replica verification; checkpoint retention;
* per-(seed, rank, worker) augmentation streams; host actions of a recorded command list.
"""
import json
import os
import socket
import sys
import textwrap
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ----------------------------------------------------------------------------- bench launcher
def test_bench_self_launch_sets_rank_environment(tmp_path, monkeypatch):
    import bench
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                "HSA_ENABLE_IPC_MODE_LEGACY"]
        with open(os.path.join({str(tmp_path)!r}, "rank" + os.environ["RANK"] + ".json"), "w") as f:
            json.dump({{k: os.environ.get(k) for k in keys}} | {{"argv": sys.argv[1:]}}, f)
    """))
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch(3, ["--gpus", "3", "--steps", "2"], timeout_s=60, script=str(script)) == 0
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert all(r["WORLD_SIZE"] == "3" and r["MASTER_ADDR"] == "127.0.0.1" for r in recs)
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and recs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert recs[1]["argv"] == ["--gpus", "3", "--steps", "2"]


def test_bench_self_launch_fails_fast_when_a_rank_dies(tmp_path, monkeypatch):
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(120)\n")
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    t0 = time.monotonic()
    assert bench.self_launch(2, [], timeout_s=100, script=str(script)) == 1
    assert time.monotonic() - t0 < 30          # the surviving rank was terminated, not waited for


def test_bench_takes_the_trainer_yamls(monkeypatch):
    """``bench.py --cfg-filepath``: the upstream-schema files name the benchmark configurations; the per-node batch of
    the file is divided by the GPU count (reference train_distributed.py:191-195), explicit flags win."""
    import argparse
    import bench
    ns = argparse.Namespace(gpus=4, steps=3, warmup=1, model="ResNet18", batch=1024, engine="native", sync_bn=0,
                            bucket_cap_mb=25.0, image_size=224, input_format="uint8_nhwc")
    cfg = bench.cfg_from_file(os.path.join(ROOT, "config", "ResNet50.yml"), ns, explicit=set())
    assert (ns.model, ns.batch, ns.sync_bn) == ("ResNet50", 64, 1)
    assert cfg["dataset"]["name"] == "synthetic" and cfg["training"]["per_gpu_batch_size"] == 64
    assert cfg["training"]["batch_size"] == 256 and cfg["training"]["optimizer"]["lr"] == 0.1
    ns2 = argparse.Namespace(**{**vars(ns), "gpus": 1, "batch": 512, "model": "X"})
    bench.cfg_from_file(os.path.join(ROOT, "config", "test-sync.yml"), ns2, explicit={"batch"})
    assert (ns2.model, ns2.batch) == ("ResNet50", 512)
    big = [f for f in os.listdir(os.path.join(ROOT, "config")) if "large-batch" in f]
    if big:
        ns3 = argparse.Namespace(**{**vars(ns), "gpus": 8})
        bench.cfg_from_file(os.path.join(ROOT, "config", big[0]), ns3, explicit=set())
        assert ns3.batch >= 256                               # per_gpu_batch_size of the file, not batch_size / 8


def test_rccl_start_up_log_summary():
    import bench
    log = "\n".join([
        "h:1:1 [0] NCCL INFO RCCL version 2.22.3-HEAD",
        "h:1:1 [0] NCCL INFO NCCL_MIN_NCHANNELS set by environment to 8.",
        "h:1:1 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId 5000 commId 0xabc - Init START",
        "h:1:1 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7",
        "h:1:1 [0] NCCL INFO Channel 01/16 :    0   3   2   1   4   7   6   5",
        "h:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 3/-1/-1->0->-1",
        "h:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC",
        "h:1:1 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer",
        "h:1:1 [0] NCCL INFO AllReduce: 26214400 Bytes -> Algo 1 proto 2 time 123.4", ""])
    got = bench.parse_rccl_log(log)
    assert got["version"].startswith("2.22") and got["nranks"] == 8 and got["channels"] == 16
    assert got["coll_channels"] == 16 and got["p2p_channels_per_peer"] == 2
    assert got["rings"] == ["0 1 2 3 4 5 6 7", "0 3 2 1 4 7 6 5"] and got["transports"] == ["P2P/IPC"]
    assert got["env"] == {"NCCL_MIN_NCHANNELS": "8"}
    assert got["choices"] == [{"collective": "AllReduce", "bytes": 26214400, "algo_proto": "ring/simple"}]
    assert bench.parse_rccl_log("")["lines"] == 0             # nothing matched: no field but the size


def test_bucket_sweep_starts_its_own_ranks(tmp_path, monkeypatch):
    """bench/bucket_sweep.py --gpus N from a plain shell starts N interpreters of ITSELF with the rank environment."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench", "bucket_sweep.py"), "--gpus", "2", "--caps", "8",
                          "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    # no GPU here: every rank reports that and exits 2 -> the launcher names both ranks and fails
    assert res.returncode == 1, res.stdout + res.stderr
    assert res.stderr.count("no GPU visible") == 2 and "ranks failed" in res.stderr


# ----------------------------------------------------------------------------- optimizer in backward
def _engine(n_classes=10, sync_bn=False, **kw):
    from pytorch_distributed_training_amd.engine import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec
    from pytorch_distributed_training_amd.ops import ref
    torch.manual_seed(7)
    eng = NativeResNet(get_spec("ResNet18", n_classes), torch.device("cpu"), ref, torch.float32, sync_bn=sync_bn,
                       bucket_cap_mb=4.0, first_bucket_mb=0.25, **kw)
    eng.load_state_dict(get_model("ResNet18", n_classes).state_dict())
    return eng


def _data(n=8):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n, 3, 32, 32, generator=g), torch.randint(0, 10, (n,), generator=g)


def _train(eng, in_backward, steps=3, no_wd_bn=False, x=None, y=None):
    from pytorch_distributed_training_amd.ops import ref
    from pytorch_distributed_training_amd.optim import FlatSGD
    opt = FlatSGD(eng.params, ref, lr=0.05, momentum=0.9, weight_decay=1e-4, no_wd_bn=no_wd_bn)
    if in_backward:
        eng.attach_optimizer(opt)
    if x is None:
        x, y = _data()
    for _ in range(steps):
        eng.forward_backward(x, y)
        if in_backward:
            opt.finish_step()
        else:
            opt.step(grad_scale=eng.reducer.grad_scale)
        eng.refresh_derived_weights()
    assert opt.steps == steps
    return eng.params.p.clone(), eng.params.m.clone(), eng.params.g.clone()


@pytest.mark.parametrize("no_wd_bn", [False, True])
def test_per_bucket_optimizer_update_is_bit_identical(no_wd_bn):
    torch.set_num_threads(4)
    p0, m0, g0 = _train(_engine(), False, no_wd_bn=no_wd_bn)
    eng = _engine()
    p1, m1, g1 = _train(eng, True, no_wd_bn=no_wd_bn)
    assert len(eng.buckets) >= 3 and eng.reducer.enabled and not eng.reducer.collective
    assert torch.equal(p0, p1) and torch.equal(m0, m1)
    assert float(g1.abs().max()) == 0.0 and float(g0.abs().max()) == 0.0      # both leave the gradients zeroed


def _dp_worker(rank, world, port, out_dir, in_backward, comm_dtype, mismatch):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from pytorch_distributed_training_amd.engine.native_resnet import ReplicaMismatchError
    out = {}
    try:
        eng = _engine(n_classes=(12 if (mismatch and rank == 1) else 10), sync_bn=True,
                      grad_allreduce_dtype=comm_dtype)
        try:
            eng.broadcast_state(0)
            out["mismatch"] = False
        except ReplicaMismatchError as exc:
            out["mismatch"] = True
            out["message"] = str(exc)
        if not out["mismatch"]:
            x, y = _data()
            p, m, _ = _train(eng, in_backward, x=x[rank * 4:(rank + 1) * 4], y=y[rank * 4:(rank + 1) * 4])
            out.update(p=p, m=m, launched=eng.reducer.stats["launched"], nb=len(eng.buckets),
                       bytes=eng.reducer.stats["bytes"])
    finally:
        torch.save(out, os.path.join(out_dir, f"rank{rank}_{int(in_backward)}.pt"))
        dist.destroy_process_group()


def _spawn(tmp_path, in_backward, comm_dtype="fp32", mismatch=False):
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path), in_backward, comm_dtype, mismatch), nprocs=2, join=True)
    return [torch.load(tmp_path / f"rank{r}_{int(in_backward)}.pt") for r in range(2)]


def test_per_bucket_update_over_gloo_matches_classic_update(tmp_path):
    classic = _spawn(tmp_path, False)
    piped = _spawn(tmp_path, True)
    assert piped[0]["launched"] == 3 * piped[0]["nb"]
    assert torch.equal(piped[0]["p"], piped[1]["p"]) and torch.equal(piped[0]["m"], piped[1]["m"])
    assert torch.equal(classic[0]["p"], piped[0]["p"]) and torch.equal(classic[0]["m"], piped[0]["m"])


def test_bf16_gradient_transport(tmp_path):
    full = _spawn(tmp_path, True)
    half = _spawn(tmp_path, True, comm_dtype="bf16")
    assert half[0]["bytes"] * 2 == full[0]["bytes"]
    assert torch.equal(half[0]["p"], half[1]["p"])                      # replicas still agree bit for bit
    rel = (half[0]["p"] - full[0]["p"]).norm() / full[0]["p"].norm()
    assert 0.0 < rel.item() < 3e-2, rel.item()                          # rounded transport, same training


def test_replica_layout_mismatch_is_detected_on_every_rank(tmp_path):
    res = _spawn(tmp_path, False, mismatch=True)
    assert res[0]["mismatch"] and res[1]["mismatch"]
    assert "rank 0" in res[0]["message"] and "rank 1" in res[0]["message"]


# ----------------------------------------------------------------------------- checkpoints
def test_prune_checkpoints_keeps_newest_and_named(tmp_path):
    from pytorch_distributed_training_amd import checkpoint as ckpt
    for it in (2, 4, 6, 8, 10):
        (tmp_path / f"iter_{it}.pth").write_bytes(b"x")
    (tmp_path / "best.pth").write_bytes(b"x")
    removed = ckpt.prune_checkpoints(str(tmp_path), 2)
    assert sorted(os.path.basename(p) for p in removed) == ["iter_2.pth", "iter_4.pth", "iter_6.pth"]
    assert sorted(os.listdir(tmp_path)) == ["best.pth", "iter_10.pth", "iter_8.pth"]
    assert ckpt.prune_checkpoints(str(tmp_path), 0) == []
    assert ckpt.latest_checkpoint(str(tmp_path)).endswith("iter_10.pth")


def test_prune_leaves_another_runs_checkpoints_alone(tmp_path):
    """A fresh run in a reused directory: the stale files with higher iteration numbers are neither deleted nor allowed
    to push the run's own checkpoints out of the newest ``keep``."""
    from pytorch_distributed_training_amd import checkpoint as ckpt
    for it in (400000, 402000):
        (tmp_path / f"iter_{it}.pth").write_bytes(b"old")
    for it in (2000, 4000, 6000):
        (tmp_path / f"iter_{it}.pth").write_bytes(b"new")
        removed = ckpt.prune_checkpoints(str(tmp_path), 2, upto=it)
    assert [os.path.basename(p) for p in removed] == ["iter_2000.pth"]
    assert sorted(os.listdir(tmp_path)) == ["iter_4000.pth", "iter_400000.pth", "iter_402000.pth", "iter_6000.pth"]
    assert [os.path.basename(p) for p in ckpt.stale_checkpoints(str(tmp_path), 6000)] == ["iter_400000.pth", "iter_402000.pth"]
    assert ckpt.stale_checkpoints(str(tmp_path), 500000) == []


def test_rng_restore_reports_what_it_could_not_restore(caplog):
    import logging
    import random
    from pytorch_distributed_training_amd import checkpoint as ckpt
    state = ckpt._pack_rng() if hasattr(ckpt, "_pack_rng") else None
    assert state is not None
    random.random()
    assert ckpt.restore_rng({"rng": state}) is True
    broken = dict(state)
    broken["numpy"] = {"kind": "MT19937"}                     # keys / pos missing
    with caplog.at_level(logging.WARNING):
        assert ckpt.restore_rng({"rng": broken}, logging.getLogger("t")) is False
    assert any("numpy" in r.getMessage() and "NOT restored" in r.getMessage() for r in caplog.records)
    assert ckpt.restore_rng({"rng": None}) is True


# ----------------------------------------------------------------------------- augmentation streams
def test_augmentation_streams_differ_per_worker_and_follow_the_seed():
    from pytorch_distributed_training_amd.data.datasets import TrainTransform
    from pytorch_distributed_training_amd.data.loader import augmentation_seed
    PIL = pytest.importorskip("PIL.Image")
    import numpy as np
    rs = np.random.RandomState(0)
    img = PIL.fromarray(rs.randint(0, 255, (64, 80, 3), dtype=np.uint8))

    def draw(seed, rank, worker):
        t = TrainTransform(32)
        t.reseed(augmentation_seed(seed, rank, worker))
        return np.stack([t(img) for _ in range(4)])

    a = draw(5, 0, 0)
    assert np.array_equal(a, draw(5, 0, 0))                    # reproducible
    assert not np.array_equal(a, draw(5, 0, 1))                # other worker
    assert not np.array_equal(a, draw(5, 1, 0))                # other rank
    assert not np.array_equal(a, draw(6, 0, 0))                # other seed
    seeds = {augmentation_seed(s, r, w) for s in range(3) for r in range(8) for w in range(-1, 16)}
    assert len(seeds) == 3 * 8 * 17


def test_loader_workers_get_distinct_streams(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    import numpy as np
    from pytorch_distributed_training_amd.data import build_loader, get_dataset
    rs = np.random.RandomState(1)
    for cls in ("a", "b"):
        os.makedirs(tmp_path / "train" / cls)
        for i in range(4):
            PIL.fromarray(rs.randint(0, 255, (48, 64, 3), dtype=np.uint8)).save(tmp_path / "train" / cls / f"{i}.png")
    # every sample is the SAME picture: what differs between items is only the augmentation stream
    one = PIL.open(tmp_path / "train" / "a" / "0.png").copy()
    for cls in ("a", "b"):
        for i in range(4):
            one.save(tmp_path / "train" / cls / f"{i}.png")

    def epoch(seed):
        ds = get_dataset("imagenet", str(tmp_path), "train", n_classes=2, image_size=32, fmt="uint8_nhwc", seed=seed)
        loader = build_loader(ds, 1, 2, True, 1, 0, seed=seed, pin_memory=False)
        return [b[0].clone() for b in loader]

    first = epoch(3)
    assert len(first) == 8
    distinct = {bytes(t.numpy().tobytes()) for t in first}
    assert len(distinct) >= 7                                   # two workers in lock-step would give at most 4
    again = epoch(3)
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    other = epoch(4)
    assert not all(torch.equal(a, b) for a, b in zip(first, other))


# ----------------------------------------------------------------------------- command-list replay
class _FakeOps:
    def __init__(self):
        self.recording = None

    def begin_recording(self, items):
        self.recording = items

    def end_recording(self):
        self.recording = None


def test_list_program_host_actions(monkeypatch):
    from pytorch_distributed_training_amd.engine import graphs

    class _S:
        cuda_stream = 1

    monkeypatch.setattr(torch.cuda, "MemPool", lambda: object(), raising=False)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: _S())
    ops = _FakeOps()
    prog = graphs.ListProgram(torch.device("cpu"), ops)
    prog.begin()
    assert ops.recording is prog.items
    calls = []
    prog.items.append((lambda a: calls.append(("lib", a)) or 0, (5,)))
    seen = []

    def action():
        seen.append(ops.recording)          # library calls made by a host action must not be listed
        calls.append("host")
        return True                         # e.g. Work.wait(): truthy, and not a failure

    prog.host(action)
    assert seen == [None] and ops.recording is prog.items
    prog.finish()
    assert prog.num_host_actions == 1 and prog.num_calls == 1
    calls.clear()
    prog.replay()
    assert calls == [("lib", 5), "host"]
    prog.items.append((lambda: 3, ()))
    with pytest.raises(RuntimeError):
        prog.replay()


def test_stream_budget_roles(monkeypatch):
    """parallel/streams.py: one stream per role, the capture stream shares the weight-gradient slot, a fifth stream of
    our own is refused; RCCL's internal stream is counted but does not take one of the four roles away."""
    import torch.distributed as dist

    from pytorch_distributed_training_amd.parallel import streams

    class FakeStream:
        def __init__(self, device=None, priority=0):
            self.device, self.priority = device, priority

    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    dev = torch.device("cuda", 0)
    streams.reset()
    try:
        w = streams.get("wgrad", dev, priority=-1)
        assert w.priority == -1 and streams.get("wgrad", dev) is w and streams.get("capture", dev) is w
        c, cp = streams.get("comm", dev), streams.get("copy", dev)
        assert c is not cp and streams.count(dev) == 4 and sorted(streams.in_use(dev)) == ["comm", "compute", "copy", "wgrad"]
        with pytest.raises(RuntimeError, match="stream budget exceeded"):
            streams.get("another", dev)
        # multi-rank RCCL job: RCCL's stream comes on top of the four
        streams.reset()
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_backend", lambda *a: "nccl")
        monkeypatch.setattr(dist, "get_world_size", lambda *a: 8)
        cp = streams.get("copy", dev)
        assert streams.get("comm", dev) is not cp
        streams.get("wgrad", dev)
        assert streams.count(dev) == 5 and "rccl" in streams.in_use(dev)
        with pytest.raises(RuntimeError, match="stream budget exceeded"):
            streams.get("another", dev)
        # the measurement switch: the copy on the communication stream
        streams.reset()
        monkeypatch.setenv("PDT_COPY_STREAM", "comm")
        assert streams.get("copy", dev) is streams.get("comm", dev)
    finally:
        streams.reset()
