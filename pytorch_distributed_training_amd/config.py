"""YAML configuration: load, validate, default and override.

The upstream trainer reads a YAML file into a plain nested ``dict`` and indexes it with
string keys, without any validation (reference ``train_distributed.py:64`` and every
``self.global_cfg[...]`` access; schema in ``config/ResNet50.yml:1-31``).  This module
keeps that surface -- :func:`get_cfg` still returns a plain nested dict that the two
upstream YAML files load into unchanged -- and adds what upstream lacks: a schema with
unknown-key rejection, defaults for optional keys, ``~`` expansion of paths and
``a.b=c`` command-line overrides.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Iterable, Optional

import yaml

# Sentinel: key must be present in the file.
_REQUIRED = object()

# Schema: nested dict of key -> (type or tuple of types, default).  A nested dict value
# is a sub-section.  ``None`` default means "optional, absent".
_SCHEMA: Dict[str, Any] = {
    "dataset": {
        "name": (str, _REQUIRED),              # imagenet | packed | synthetic | synthetic_learnable (brightness encodes the label)
        "root": (str, ""),
        "n_classes": (int, _REQUIRED),
        # extensions
        "image_size": (int, 224),
        "train_size": (int, 1281167),          # synthetic only: virtual dataset length
        "val_size": (int, 50000),
        "input_format": (str, "float_nchw"),   # float_nchw (upstream contract) | uint8_nhwc
        "cpu_augment": (bool, False),          # packed only: crop / resize / flip in the loader workers, not on the GPU
    },
    "training": {
        "optimizer": {
            "name": (str, _REQUIRED),
            "lr": ((int, float), _REQUIRED),
            "weight_decay": ((int, float), 0.0),
            "momentum": ((int, float), 0.0),
            "nesterov": (bool, False),
            "dampening": ((int, float), 0.0),
        },
        "lr_schedule": {
            "name": (str, _REQUIRED),
            "milestones": (list, None),
            "gamma": ((int, float), 0.1),
            "warmup_iters": (int, 0),
            "warmup_mode": (str, "linear"),
            "warmup_factor": ((int, float), 1.0 / 3.0),
            "max_iters": (int, None),          # cosine / poly horizon; defaults to train_iters
            "min_lr": ((int, float), 0.0),
            "power": ((int, float), 0.9),
        },
        "train_iters": (int, _REQUIRED),
        "print_interval": (int, 50),
        "val_interval": (int, 2000),
        "batch_size": (int, _REQUIRED),
        "num_workers": (int, 0),
        "sync_bn": (bool, False),
        # extensions (defaults preserve upstream behaviour)
        "per_gpu_batch_size": (int, None),
        "precision": (str, "bf16"),            # bf16 on GPU; the CPU path always computes fp32
        "bucket_cap_mb": ((int, float), 25.0),
        "first_bucket_mb": ((int, float), 1.0),
        "overlap_comm": (bool, True),
        "checkpoint_interval": (int, None),
        "checkpoint_dir": (str, None),
        "resume": (str, None),
        "no_wd_bn": (bool, False),
        "syncbn_comm": (str, "auto"),          # auto | xgmi | rccl: transport of the SyncBN statistics
        "syncbn_overlap": (bool, False),       # async BN-backward all-reduce overlapped with the previous wgrad
        "engine": (str, "native"),             # native | torch
        "max_val_batches": (int, None),
        "keep_last_checkpoints": (int, 3),     # periodic checkpoints kept on disk (0 = keep all); named tags are never pruned
        "grad_allreduce_dtype": (str, "fp32"), # fp32 | bf16: wire format of the gradient buckets
        "optimizer_in_backward": (bool, True), # per-bucket fused update behind each bucket's all-reduce
    },
    "validation": {
        "batch_size": (int, None),
        "num_workers": (int, None),
    },
    "model": {
        "name": (str, _REQUIRED),
    },
    "runtime": {
        "profile": (bool, False),              # roctx ranges around step phases
        "debug_sync_check": (int, 0),          # every N iters compare param hashes across ranks
        "collective_timeout_s": (int, 1800),
        "xgmi_timeout_s": ((int, float), None),# bound of a wait inside the xGMI SyncBN exchange; default: min(60, collective_timeout_s)
        "deterministic": (bool, False),
        "hip_graphs": (bool, True),            # native engine on GPU: record the step once, then replay it
        "step_replay": (str, "list"),          # list (command list of stream launches) | graph (HIP graphs) | off
        "eval_only": (bool, False),            # load training.resume (if any), run one validation pass, exit
    },
}


class ConfigError(ValueError):
    """Raised for missing / unknown / ill-typed configuration keys."""


def _check_type(path: str, value: Any, typ: Any) -> Any:
    if value is None:
        return None
    if typ is bool or (isinstance(typ, tuple) and bool in typ):
        if isinstance(value, bool):
            return value
    if isinstance(value, bool) and typ is not bool:
        raise ConfigError(f"{path}: expected {typ}, got bool {value!r}")
    if not isinstance(value, typ):
        # YAML reads "1e-4" (no dot) as a string; accept numeric strings for numbers.
        if isinstance(value, str) and typ in ((int, float), float):
            try:
                return float(value)
            except ValueError:
                pass
        raise ConfigError(f"{path}: expected {typ}, got {type(value).__name__} {value!r}")
    return value


def _validate(section: Dict[str, Any], schema: Dict[str, Any], path: str) -> Dict[str, Any]:
    if section is None:
        section = {}
    if not isinstance(section, dict):
        raise ConfigError(f"{path or '<root>'}: expected a mapping, got {type(section).__name__}")
    unknown = set(section) - set(schema)
    if unknown:
        raise ConfigError(
            f"unknown configuration key(s) under '{path or '<root>'}': {sorted(unknown)}; "
            f"known keys: {sorted(schema)}"
        )
    out: Dict[str, Any] = {}
    for key, spec in schema.items():
        sub_path = f"{path}.{key}" if path else key
        if isinstance(spec, dict):
            out[key] = _validate(section.get(key), spec, sub_path)
            continue
        typ, default = spec
        if key in section:
            out[key] = _check_type(sub_path, section[key], typ)
        elif default is _REQUIRED:
            raise ConfigError(f"missing required configuration key '{sub_path}'")
        else:
            out[key] = copy.deepcopy(default)
    return out


def _expand_paths(cfg: Dict[str, Any]) -> None:
    root = cfg["dataset"].get("root")
    if root:
        cfg["dataset"]["root"] = os.path.expanduser(root)
    for key in ("checkpoint_dir", "resume"):
        val = cfg["training"].get(key)
        if val:
            cfg["training"][key] = os.path.expanduser(val)


def parse_override(text: str) -> Any:
    """Parse the value part of an ``a.b=c`` override with YAML scalar rules."""
    return yaml.safe_load(text)


def apply_overrides(raw: Dict[str, Any], overrides: Optional[Iterable[str]]) -> Dict[str, Any]:
    """Apply ``section.key=value`` strings onto a raw (unvalidated) config dict."""
    for item in overrides or ():
        if "=" not in item:
            raise ConfigError(f"override '{item}' is not of the form a.b=value")
        dotted, value = item.split("=", 1)
        node = raw
        parts = dotted.strip().split(".")
        for part in parts[:-1]:
            nxt = node.get(part)
            if nxt is None:
                nxt = {}
                node[part] = nxt
            if not isinstance(nxt, dict):
                raise ConfigError(f"override '{item}': '{part}' is not a section")
            node = nxt
        node[parts[-1]] = parse_override(value)
    return raw


def validate_cfg(raw: Dict[str, Any]) -> Dict[str, Any]:
    """Validate a raw nested dict against the schema and fill in defaults."""
    cfg = _validate(copy.deepcopy(raw), _SCHEMA, "")
    _expand_paths(cfg)
    sched = cfg["training"]["lr_schedule"]
    if sched["name"] == "multi_step" and not sched.get("milestones"):
        raise ConfigError("training.lr_schedule.milestones is required for multi_step")
    if cfg["training"]["precision"] not in ("bf16", "fp32"):
        raise ConfigError("training.precision must be 'bf16' or 'fp32'")
    if cfg["training"]["engine"] not in ("native", "torch"):
        raise ConfigError("training.engine must be 'native' or 'torch'")
    if cfg["dataset"]["input_format"] not in ("float_nchw", "uint8_nhwc"):
        raise ConfigError("dataset.input_format must be 'float_nchw' or 'uint8_nhwc'")
    return cfg


def get_cfg(filepath: str, overrides: Optional[Iterable[str]] = None) -> Dict[str, Any]:
    """Load a YAML config into a validated nested dict.

    Same call shape as upstream ``get_cfg(args.cfg_filepath)``
    (reference ``train_distributed.py:29,64``).
    """
    with open(os.path.expanduser(filepath), "r") as f:
        raw = yaml.safe_load(f)
    if not isinstance(raw, dict):
        raise ConfigError(f"{filepath}: top level of the YAML file must be a mapping")
    raw = apply_overrides(raw, overrides)
    return validate_cfg(raw)


def validation_loader_settings(cfg: Dict[str, Any]) -> Dict[str, int]:
    """Batch size / worker count for the validation loader.

    Upstream builds the validation loader with the *training* values and never reads the
    ``validation`` section (reference ``train_distributed.py:235-241``).  Here the section
    is honoured when present and falls back to the training values otherwise.
    """
    val = cfg.get("validation") or {}
    return {
        "batch_size": val.get("batch_size") or cfg["training"]["batch_size"],
        "num_workers": (
            val["num_workers"] if val.get("num_workers") is not None else cfg["training"]["num_workers"]
        ),
    }
