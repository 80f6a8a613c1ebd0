"""ResNet-18/34/50/101/152: architecture description and a plain ``nn.Module`` build.

Upstream gets its model from ``get_model(model_name, num_classes)`` with names such as
``ResNet50`` (reference ``train_distributed.py:183-186``, ``config/ResNet50.yml:30-31``) and
its README quotes the torchvision model-zoo errors for ResNet-18..152
(reference ``README.md:9-13``), i.e. the torchvision "v1.5" ResNets (stride 2 on the 3x3
convolution of a bottleneck).

Two things live here:

* :class:`ResNetSpec` -- a declarative description (units of conv+BN, blocks, stages)
  from which both the eager module below and the native MI355X engine
  (``engine/native_resnet.py``) are built, together with the torchvision ``state_dict``
  key schema (:func:`ResNetSpec.state_dict_schema`).
* :class:`ResNet` -- a standard NCHW fp32 ``nn.Module`` whose ``state_dict`` keys and
  shapes equal torchvision's.  It is the numerical oracle for the native engine, the
  CPU fall-back model and the weight initialiser (kaiming-normal fan-out for
  convolutions, ones/zeros for BN, ``nn.Linear`` defaults for the classifier).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Declarative architecture
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConvBNSpec:
    """One convolution followed by BatchNorm (bias-free conv)."""

    conv: str          # state_dict prefix of the conv, e.g. "layer1.0.conv1"
    bn: str            # state_dict prefix of the BN,   e.g. "layer1.0.bn1"
    cin: int
    cout: int
    ksize: int
    stride: int
    pad: int


@dataclass(frozen=True)
class BlockSpec:
    name: str                          # "layer2.0"
    kind: str                          # "basic" | "bottleneck"
    units: Tuple[ConvBNSpec, ...]      # 2 (basic) or 3 (bottleneck) conv+BN units
    downsample: Optional[ConvBNSpec]   # 1x1 conv + BN on the identity path, or None
    cin: int
    cout: int
    stride: int


@dataclass
class ResNetSpec:
    name: str
    kind: str
    layers: Tuple[int, int, int, int]
    num_classes: int
    stem: ConvBNSpec = field(init=False)
    blocks: List[BlockSpec] = field(init=False)
    fc_in: int = field(init=False)

    def __post_init__(self) -> None:
        expansion = 1 if self.kind == "basic" else 4
        self.stem = ConvBNSpec("conv1", "bn1", 3, 64, 7, 2, 3)
        self.blocks = []
        inplanes = 64
        for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), self.layers), start=1):
            for bi in range(nblocks):
                stride = 2 if (bi == 0 and li > 1) else 1
                name = f"layer{li}.{bi}"
                outplanes = planes * expansion
                if self.kind == "basic":
                    units = (
                        ConvBNSpec(f"{name}.conv1", f"{name}.bn1", inplanes, planes, 3, stride, 1),
                        ConvBNSpec(f"{name}.conv2", f"{name}.bn2", planes, planes, 3, 1, 1),
                    )
                else:
                    units = (
                        ConvBNSpec(f"{name}.conv1", f"{name}.bn1", inplanes, planes, 1, 1, 0),
                        ConvBNSpec(f"{name}.conv2", f"{name}.bn2", planes, planes, 3, stride, 1),
                        ConvBNSpec(f"{name}.conv3", f"{name}.bn3", planes, outplanes, 1, 1, 0),
                    )
                down = None
                if stride != 1 or inplanes != outplanes:
                    down = ConvBNSpec(f"{name}.downsample.0", f"{name}.downsample.1",
                                      inplanes, outplanes, 1, stride, 0)
                self.blocks.append(BlockSpec(name, self.kind, units, down, inplanes, outplanes, stride))
                inplanes = outplanes
        self.fc_in = inplanes

    # -- inventories -------------------------------------------------------------------
    def conv_bn_units(self) -> List[ConvBNSpec]:
        """All conv+BN units in forward order (downsample listed after its block's units)."""
        out = [self.stem]
        for b in self.blocks:
            out.extend(b.units)
            if b.downsample is not None:
                out.append(b.downsample)
        return out

    def state_dict_schema(self) -> Dict[str, Tuple[Tuple[int, ...], torch.dtype]]:
        """``key -> (shape, dtype)`` in torchvision ``state_dict`` order."""
        schema: Dict[str, Tuple[Tuple[int, ...], torch.dtype]] = {}

        def add_unit(u: ConvBNSpec) -> None:
            schema[f"{u.conv}.weight"] = ((u.cout, u.cin, u.ksize, u.ksize), torch.float32)
            schema[f"{u.bn}.weight"] = ((u.cout,), torch.float32)
            schema[f"{u.bn}.bias"] = ((u.cout,), torch.float32)
            schema[f"{u.bn}.running_mean"] = ((u.cout,), torch.float32)
            schema[f"{u.bn}.running_var"] = ((u.cout,), torch.float32)
            schema[f"{u.bn}.num_batches_tracked"] = ((), torch.int64)

        add_unit(self.stem)
        for b in self.blocks:
            for u in b.units:
                add_unit(u)
            if b.downsample is not None:
                add_unit(b.downsample)
        schema["fc.weight"] = ((self.num_classes, self.fc_in), torch.float32)
        schema["fc.bias"] = ((self.num_classes,), torch.float32)
        return schema

    def parameter_keys(self) -> List[str]:
        """Trainable-parameter keys in ``model.parameters()`` order."""
        return [k for k in self.state_dict_schema()
                if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]

    def num_parameters(self) -> int:
        schema = self.state_dict_schema()
        total = 0
        for k in self.parameter_keys():
            n = 1
            for d in schema[k][0]:
                n *= d
            total += n
        return total


_ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}


def model_names() -> List[str]:
    return ["ResNet18", "ResNet34", "ResNet50", "ResNet101", "ResNet152"]


def get_spec(model_name: str, num_classes: int = 1000) -> ResNetSpec:
    key = model_name.lower().replace("-", "").replace("_", "")
    if key not in _ARCHS:
        raise ValueError(f"unknown model '{model_name}'; available: {model_names()}")
    kind, layers = _ARCHS[key]
    return ResNetSpec(name=key, kind=kind, layers=layers, num_classes=int(num_classes))


# --------------------------------------------------------------------------------------
# Eager nn.Module (torchvision-compatible state_dict)
# --------------------------------------------------------------------------------------
def _conv(u: ConvBNSpec) -> nn.Conv2d:
    return nn.Conv2d(u.cin, u.cout, u.ksize, stride=u.stride, padding=u.pad, bias=False)


class _Block(nn.Module):
    def __init__(self, spec: BlockSpec) -> None:
        super().__init__()
        self.n_units = len(spec.units)
        for i, u in enumerate(spec.units, start=1):
            setattr(self, f"conv{i}", _conv(u))
            setattr(self, f"bn{i}", nn.BatchNorm2d(u.cout))
        self.downsample = None
        if spec.downsample is not None:
            self.downsample = nn.Sequential(_conv(spec.downsample), nn.BatchNorm2d(spec.downsample.cout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = x
        for i in range(1, self.n_units + 1):
            out = getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(out))
            if i < self.n_units:
                out = F.relu(out)
        return F.relu(out + identity)


class ResNet(nn.Module):
    """NCHW eager ResNet built from a :class:`ResNetSpec`."""

    def __init__(self, spec: ResNetSpec) -> None:
        super().__init__()
        self.spec = spec
        self.conv1 = _conv(spec.stem)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        stages: Dict[str, List[_Block]] = {}
        for b in spec.blocks:
            stages.setdefault(b.name.split(".")[0], []).append(_Block(b))
        for stage_name, blocks in stages.items():
            setattr(self, stage_name, nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(spec.fc_in, spec.num_classes)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def get_model(model_name: str, num_classes: int = 1000) -> ResNet:
    """Name -> eager model factory (same call shape as upstream's ``get_model``)."""
    return ResNet(get_spec(model_name, num_classes))
