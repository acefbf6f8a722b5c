"""mvp — MI355X-native kernels + host glue for the midvision-probe hot path.

Import layout:  mvp.lib (C-ABI binding) · mvp.ops (tensor wrappers) · mvp.vit (backbone
engine) · mvp.functional (autograd ops of the probe path) · mvp.optim (flat fused AdamW) ·
mvp.train (the train_depth / train_snorm loop bodies) · mvp.dist (RCCL data parallelism).
"""
__version__ = "0.1.0"
