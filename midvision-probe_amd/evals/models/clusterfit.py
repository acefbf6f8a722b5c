"""evals.models.clusterfit.CLUSTERFIT — drop-in for evals/models/clusterfit.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

CLUSTERFIT = make_ssl_resnet50("CLUSTERFIT", "$clusterfit$", ['_feature_blocks.'], ['clusterfit_resnet50'], "evals/models/clusterfit.py")
