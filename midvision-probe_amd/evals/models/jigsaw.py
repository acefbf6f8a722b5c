"""evals.models.jigsaw.JIGSAW — drop-in for evals/models/jigsaw.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

JIGSAW = make_ssl_resnet50("JIGSAW", "$jigsaw$", ['_feature_blocks.'], ['jigsaw_resnet50'], "evals/models/jigsaw.py")
