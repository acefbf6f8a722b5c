"""evals.models.pirl.PIRL — drop-in for evals/models/pirl.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

PIRL = make_ssl_resnet50("PIRL", "$pirl$", ['_feature_blocks.'], ['pirl_resnet50'], "evals/models/pirl.py")
