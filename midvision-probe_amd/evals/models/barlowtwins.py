"""evals.models.barlowtwins.BARLOWTWINS — drop-in for evals/models/barlowtwins.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

BARLOWTWINS = make_ssl_resnet50("BARLOWTWINS", "$barlowtwins$", ['backbone.'], ['barlowtwins_resnet50'], "evals/models/barlowtwins.py")
