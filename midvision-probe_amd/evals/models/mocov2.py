"""evals.models.mocov2.MOCOV2 — drop-in for evals/models/mocov2.py (ResNet-50 SSL backbone, shared template)."""
from mvp.resnet_backbone import make_ssl_resnet50

MOCOV2 = make_ssl_resnet50("MOCOV2", "$mocov2$", ['module.encoder_q.'], ['mocov2_resnet50', 'moco_v2_800ep_pretrain'], "evals/models/mocov2.py")
