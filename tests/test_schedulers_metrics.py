import math

import pytest
import torch

from pytorch_distributed_training_amd.optim.schedulers import get_scheduler
from pytorch_distributed_training_amd.trainer import is_val_iter, per_process_batch, per_process_workers
from pytorch_distributed_training_amd.utils.metrics import AverageMeter, TopKCounter, accuracy, topk_correct


class _Opt:
    def __init__(self, lr):
        self.param_groups = [{"lr": lr}]


def test_multistep_values_at_reference_milestones():
    opt = _Opt(0.1)
    s = get_scheduler(opt, {"name": "multi_step", "milestones": [150000, 300000], "gamma": 0.1})
    assert s.lr_at(0)[0] == pytest.approx(0.1)
    assert s.lr_at(149999)[0] == pytest.approx(0.1)
    assert s.lr_at(150000)[0] == pytest.approx(0.01)
    assert s.lr_at(299999)[0] == pytest.approx(0.01)
    assert s.lr_at(300000)[0] == pytest.approx(0.001)
    # stepping once per iteration reproduces the closed form
    for _ in range(7):
        s.step()
    assert s.get_last_lr() == s.lr_at(7) and opt.param_groups[0]["lr"] == s.lr_at(7)[0]


def test_matches_torch_multistep():
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=0.1)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, milestones=[3, 6], gamma=0.1)
    s = get_scheduler(_Opt(0.1), {"name": "multi_step", "milestones": [3, 6], "gamma": 0.1})
    for _ in range(9):
        assert s.get_last_lr()[0] == pytest.approx(tsch.get_last_lr()[0])
        topt.step(); tsch.step(); s.step()


def test_warmup_linear_and_constant():
    s = get_scheduler(_Opt(0.3), {"name": "multi_step", "milestones": [1000], "gamma": 0.1, "warmup_iters": 300,
                                  "warmup_mode": "linear", "warmup_factor": 0.3333})
    assert s.lr_at(0)[0] == pytest.approx(0.3 * 0.3333)
    assert s.lr_at(150)[0] == pytest.approx(0.3 * (0.3333 * 0.5 + 0.5))
    assert s.lr_at(300)[0] == pytest.approx(0.3)
    c = get_scheduler(_Opt(0.3), {"name": "constant", "warmup_iters": 10, "warmup_mode": "constant",
                                  "warmup_factor": 0.1})
    assert c.lr_at(9)[0] == pytest.approx(0.03) and c.lr_at(10)[0] == pytest.approx(0.3)


def test_cosine_poly_and_state():
    s = get_scheduler(_Opt(1.0), {"name": "cosine"}, train_iters=100)
    assert s.lr_at(0)[0] == pytest.approx(1.0) and s.lr_at(50)[0] == pytest.approx(0.5)
    assert s.lr_at(100)[0] == pytest.approx(0.0, abs=1e-12)
    p = get_scheduler(_Opt(1.0), {"name": "poly", "power": 2.0}, train_iters=10)
    assert p.lr_at(5)[0] == pytest.approx(0.25)
    for _ in range(13):
        s.step()
    s2 = get_scheduler(_Opt(1.0), {"name": "cosine"}, train_iters=100)
    s2.load_state_dict(s.state_dict())
    assert s2.get_last_lr() == s.get_last_lr()
    with pytest.raises(ValueError):
        get_scheduler(_Opt(1.0), {"name": "nope"})


def test_is_val_schedule_matches_reference_formula():
    # reference train_distributed.py:255-259 with val_interval 20, train_iters 50
    vals = [i for i in range(50) if is_val_iter(i, 20, 50)]
    assert vals == [19, 39, 49]
    assert [i for i in range(5) if is_val_iter(i, 1, 5)] == [1, 2, 3, 4]     # iter 0 is excluded by p1
    assert is_val_iter(0, 2000, 1)                                            # last iteration always validates


def test_batch_and_worker_split():
    cfg = {"training": {"batch_size": 256, "num_workers": 16}}
    assert per_process_batch(cfg, 4, True) == 64 and per_process_batch(cfg, 4, False) == 256
    assert per_process_workers(16, 4, True) == 4 and per_process_workers(10, 4, True) == 3
    cfg["training"]["per_gpu_batch_size"] = 512
    assert per_process_batch(cfg, 8, True) == 512


def test_accuracy_vs_bruteforce():
    torch.manual_seed(0)
    pred = torch.randn(64, 20)
    label = torch.randint(0, 20, (64,))
    a1, a5 = accuracy(pred, label, (1, 5))
    top5 = pred.topk(5, dim=1).indices
    brute1 = (top5[:, 0] == label).float().mean() * 100
    brute5 = (top5 == label[:, None]).any(dim=1).float().mean() * 100
    assert a1.item() == pytest.approx(brute1.item()) and a5.item() == pytest.approx(brute5.item())
    c1, c5 = topk_correct(pred, label)
    assert c1.item() == pytest.approx(brute1.item() * 64 / 100)


def test_meters():
    m = AverageMeter()
    for v in (torch.tensor(1.0), torch.tensor(3.0), 5.0):
        m.update(v)
    assert m.value() == pytest.approx(3.0) and m.count == 3
    c = TopKCounter(torch.device("cpu"))
    c.update(torch.tensor(4.0), torch.tensor(1.0), torch.tensor(2.0), 4)
    c.add_tensor(torch.tensor([2.0, 1.0, 2.0, 4.0]))
    r = c.result()
    assert r["count"] == 8 and r["loss"] == pytest.approx(0.75) and r["acc1"] == pytest.approx(25.0)
    assert r["acc5"] == pytest.approx(50.0)
