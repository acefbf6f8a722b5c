"""evals.models.npid.NPID — drop-in for evals/models/npid.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

NPID = make_ssl_resnet50("NPID", "$npid$", ['_feature_blocks.'], ['npid_resnet50'], "evals/models/npid.py")
