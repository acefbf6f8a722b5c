"""evals.models.selav2.SELAV2 — drop-in for evals/models/selav2.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

SELAV2 = make_ssl_resnet50("SELAV2", "$sela_v2$", ['module.'], ['selav2_resnet50'], "evals/models/selav2.py")
