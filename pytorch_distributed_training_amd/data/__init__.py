from .datasets import get_dataset, SyntheticImageNet, ImageFolderDataset, IMAGENET_MEAN, IMAGENET_STD
from .loader import make_iter_dataloader, build_loader, DevicePrefetcher
from .packed import GpuAugment, PackedImageDataset, pack_image_folder

__all__ = ["get_dataset", "SyntheticImageNet", "ImageFolderDataset", "IMAGENET_MEAN", "IMAGENET_STD",
           "make_iter_dataloader", "build_loader", "DevicePrefetcher", "GpuAugment", "PackedImageDataset",
           "pack_image_folder"]
