"""MI355X-native data-parallel image-classification trainer.

Sub-packages: ``models`` (ResNet descriptions + eager oracle), ``ops`` (HIP kernels for
gfx950 and their PyTorch reference), ``engine`` (flat parameter store and the explicit
forward/backward program), ``parallel`` (process groups, bucketed RCCL gradient reducer,
samplers), ``optim`` (fused flat SGD, iteration schedules), ``data`` (synthetic and ImageNet
pipelines), ``utils`` (metrics, seeding, profiling).
"""
__version__ = "0.1.0"
