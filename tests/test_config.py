import os

import pytest

from pytorch_distributed_training_amd.config import (ConfigError, apply_overrides, get_cfg, validate_cfg,
                                                     validation_loader_settings)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["ResNet50.yml", "test-sync.yml", "resnet18-cpu-gloo.yml", "ResNet50-synthetic.yml",
                                  "ResNet101-large-batch.yml", "ResNet152-syncbn.yml"])
def test_shipped_configs_load(name):
    cfg = get_cfg(os.path.join(ROOT, "config", name))
    assert cfg["model"]["name"].startswith("ResNet")
    assert cfg["training"]["optimizer"]["name"] == "SGD"


def test_upstream_schema_values():
    cfg = get_cfg(os.path.join(ROOT, "config", "ResNet50.yml"))
    tr = cfg["training"]
    assert tr["train_iters"] == 450000 and tr["print_interval"] == 50 and tr["val_interval"] == 2000
    assert tr["batch_size"] == 256 and tr["num_workers"] == 16 and tr["sync_bn"] is True
    assert tr["optimizer"] == {"name": "SGD", "lr": 0.1, "weight_decay": 1e-4, "momentum": 0.9,
                               "nesterov": False, "dampening": 0.0}
    assert tr["lr_schedule"]["milestones"] == [150000, 300000] and tr["lr_schedule"]["gamma"] == 0.1
    assert cfg["dataset"]["n_classes"] == 1000
    assert "~" not in cfg["dataset"]["root"]          # expanded
    # defaults of the extension keys preserve upstream behaviour
    assert tr["precision"] == "bf16" and tr["bucket_cap_mb"] == 25.0 and tr["engine"] == "native"
    sync = get_cfg(os.path.join(ROOT, "config", "test-sync.yml"))
    assert sync["training"]["train_iters"] == 15000 and sync["training"]["val_interval"] == 20
    assert sync["training"]["batch_size"] == 64 and sync["training"]["print_interval"] == 10


def _minimal():
    return {"dataset": {"name": "synthetic", "n_classes": 10},
            "training": {"optimizer": {"name": "SGD", "lr": 0.1},
                         "lr_schedule": {"name": "multi_step", "milestones": [5]},
                         "train_iters": 10, "batch_size": 4},
            "model": {"name": "ResNet18"}}


def test_unknown_keys_rejected():
    raw = _minimal()
    raw["training"]["batchsize"] = 3
    with pytest.raises(ConfigError, match="unknown configuration key"):
        validate_cfg(raw)
    raw = _minimal()
    raw["bogus"] = {}
    with pytest.raises(ConfigError):
        validate_cfg(raw)


def test_missing_and_bad_types():
    raw = _minimal()
    del raw["training"]["train_iters"]
    with pytest.raises(ConfigError, match="train_iters"):
        validate_cfg(raw)
    raw = _minimal()
    raw["training"]["batch_size"] = "many"
    with pytest.raises(ConfigError):
        validate_cfg(raw)
    raw = _minimal()
    raw["training"]["lr_schedule"] = {"name": "multi_step"}
    with pytest.raises(ConfigError, match="milestones"):
        validate_cfg(raw)


def test_overrides_and_validation_fallback():
    raw = apply_overrides(_minimal(), ["training.optimizer.lr=0.5", "training.sync_bn=true", "dataset.image_size=64"])
    cfg = validate_cfg(raw)
    assert cfg["training"]["optimizer"]["lr"] == 0.5 and cfg["training"]["sync_bn"] is True
    assert cfg["dataset"]["image_size"] == 64
    assert validation_loader_settings(cfg) == {"batch_size": 4, "num_workers": 0}
    raw = _minimal()
    raw["validation"] = {"batch_size": 32, "num_workers": 3}
    assert validation_loader_settings(validate_cfg(raw)) == {"batch_size": 32, "num_workers": 3}
    with pytest.raises(ConfigError):
        apply_overrides(_minimal(), ["nonsense"])


def test_scientific_notation_string_lr():
    raw = _minimal()
    raw["training"]["optimizer"]["weight_decay"] = "1e-4"     # YAML 1.1 reads "1e-4" as a string
    assert validate_cfg(raw)["training"]["optimizer"]["weight_decay"] == pytest.approx(1e-4)
