"""evals.models.rotnet.ROTNET — drop-in for evals/models/rotnet.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

ROTNET = make_ssl_resnet50("ROTNET", "$rotnet$", ['_feature_blocks.'], ['rotnet_resnet50'], "evals/models/rotnet.py")
