"""evals.models.swav.SWAV — drop-in for evals/models/swav.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

SWAV = make_ssl_resnet50("SWAV", "$swav$", ['module.'], ['swav_resnet50', 'swav_800ep_pretrain'], "evals/models/swav.py")
