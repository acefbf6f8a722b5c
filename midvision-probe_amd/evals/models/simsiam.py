"""evals.models.simsiam.SIMSIAM — drop-in for evals/models/simsiam.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

SIMSIAM = make_ssl_resnet50("SIMSIAM", "$simsiam$", ['backbone.'], ['simsiam_resnet50'], "evals/models/simsiam.py")
