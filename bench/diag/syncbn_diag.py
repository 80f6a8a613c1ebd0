"""Diagnostic: where do two SyncBN ranks (sharing one GPU over gloo) differ from one process on the
concatenated batch?  Prints loss, running statistics and per-bucket gradient errors, plus the
run-to-run noise of the single process for scale."""
import os, sys, socket, subprocess, json
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as t


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + os.path.join(ROOT, "tests")
    for comm in ("auto", "rccl"):
        env["PDT_SYNCBN_COMM"] = comm
        procs = [subprocess.Popen([sys.executable, "-c", f"import test_dist_gpu as t; t._worker({r}, 2, {port}, {out!r})"], env=env) for r in range(2)]
        for p in procs: p.wait(timeout=200)
        r0 = torch.load(os.path.join(out, "rank0.pt")); r1 = torch.load(os.path.join(out, "rank1.pt"))
        single = t._engine(False)
        x, y = t._data()
        loss = single.forward_backward(x.cuda(), y.cuda())
        g1 = single.params.g.cpu().clone()
        single.params.g.zero_()
        single2 = t._engine(False)
        loss2 = single2.forward_backward(x.cuda(), y.cuda())
        g2 = single2.params.g.cpu().clone()
        print(comm, "loss 2rank", 0.5 * (r0["loss"] + r1["loss"]).item(), "single", loss.item(), loss2.item())
        print(" rm err", (r0["rm"] - single.running_mean.cpu()).abs().max().item())
        print(" bucket errs 2rank vs single", [round(e, 5) for e in t.bucket_errors(r0["grads"], g1, single.buckets)])
        print(" bucket errs single vs single", [round(e, 5) for e in t.bucket_errors(g2, g1, single.buckets)])
        # per-parameter errors
        P = single.params
        worst = []
        for k in P.order:
            e = P.entries[k]
            a, b = r0["grads"][e.offset:e.offset + e.numel], g1[e.offset:e.offset + e.numel]
            worst.append((((a - b).norm() / (b.norm() + 1e-12)).item(), k))
        for err, k in worst[:12] + worst[-6:]:
            print(f"   {k:32s} {err:.5f}")


main()
