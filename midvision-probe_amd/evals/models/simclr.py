"""evals.models.simclr.SIMCLR — drop-in for evals/models/simclr.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

SIMCLR = make_ssl_resnet50("SIMCLR", "simclr", ['_feature_blocks.'], ['simclr_resnet50'], "evals/models/simclr.py")
