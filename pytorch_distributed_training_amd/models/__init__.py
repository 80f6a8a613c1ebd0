from .resnet import ResNet, ResNetSpec, ConvBNSpec, BlockSpec, get_model, get_spec, model_names

__all__ = ["ResNet", "ResNetSpec", "ConvBNSpec", "BlockSpec", "get_model", "get_spec", "model_names"]
