from .native_resnet import NativeResNet, backward_param_order
from .params import FlatParams, make_buckets

__all__ = ["NativeResNet", "FlatParams", "make_buckets", "backward_param_order"]
