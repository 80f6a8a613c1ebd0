"""Diagnostic: first layer at which a 2-rank SyncBN forward (ranks sharing one GPU over gloo)
departs from the single-process forward over the concatenated batch."""
import os, sys, socket, subprocess
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as t


def fwd_record(eng, x):
    eng.forward(x, train=True)
    torch.cuda.synchronize()
    rec = {}
    for u in eng.units:
        c = eng._ctx[u.bn]
        rec[u.bn] = dict(y=c.y.cpu(), scale=c.scale.cpu(), shift=c.shift.cpu(), mean=c.mean.cpu(), invstd=c.invstd.cpu())
    return rec


def worker(rank, port, out):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=2, rank=rank)
    eng = t._engine(True)
    eng.broadcast_state(0)
    x, y = t._data()
    rec = fwd_record(eng, x[rank * 8:(rank + 1) * 8].cuda())
    torch.save(rec, os.path.join(out, f"fwd{rank}.pt"))
    dist.destroy_process_group()


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + os.path.join(ROOT, "bench", "diag")
    procs = [subprocess.Popen([sys.executable, "-c", f"import syncbn_diag2 as d; d.worker({r}, {port}, {out!r})"], env=env) for r in range(2)]
    for p in procs: p.wait(timeout=200)
    r0 = torch.load(os.path.join(out, "fwd0.pt")); r1 = torch.load(os.path.join(out, "fwd1.pt"))
    single = t._engine(False)
    x, y = t._data()
    rec = fwd_record(single, x.cuda())
    for bn in rec:
        a, b = rec[bn], r0[bn]
        y16 = a["y"]; y8 = torch.cat([r0[bn]["y"], r1[bn]["y"]])
        ydiff = (y16.float() - y8.float()).abs().max().item()
        nflip = int((y16 != y8).sum())
        print(f"{bn:28s} y maxdiff {ydiff:.5f} differing {nflip}/{y16.numel()}  mean {(a['mean']-b['mean']).abs().max():.2e} "
              f"invstd rel {((a['invstd']-b['invstd']).abs()/a['invstd']).max():.2e} scale rel {((a['scale']-b['scale']).abs()/a['scale'].abs().clamp_min(1e-6)).max():.2e}")


if __name__ == "__main__":
    main()
