"""evals.models.byol.BYOL — drop-in for evals/models/byol.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

BYOL = make_ssl_resnet50("BYOL", "$byol$", ['module.'], ['byol_resnet50'], "evals/models/byol.py")
