from .schedulers import get_scheduler, MultiStepLR, CosineLR, PolyLR, ConstantLR
from .sgd import FlatSGD, get_optimizer

__all__ = ["get_scheduler", "MultiStepLR", "CosineLR", "PolyLR", "ConstantLR", "FlatSGD", "get_optimizer"]
