"""evals.models.deepclusterv2.DEEPCLUSTERV2 — drop-in for evals/models/deepclusterv2.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

DEEPCLUSTERV2 = make_ssl_resnet50("DEEPCLUSTERV2", "$deepcluster_v2$", ['module.'], ['deepclusterv2_resnet50'], "evals/models/deepclusterv2.py")
