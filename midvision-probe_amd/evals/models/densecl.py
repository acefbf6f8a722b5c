"""evals.models.densecl.DENSECL — drop-in for evals/models/densecl.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

DENSECL = make_ssl_resnet50("DENSECL", "$densecl$", [], ['densecl_resnet50'], "evals/models/densecl.py")
