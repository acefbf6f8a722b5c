"""Working package init (the reference's evals/models/__init__.py:1-7 re-exports names that
no longer exist and raises ImportError; hydra only needs the sub-modules to be importable)."""
from . import probes  # noqa: F401
from .dino import DINO  # noqa: F401
from .ibot import iBOT  # noqa: F401
from .mae import MAE  # noqa: F401
from .mocov3 import MoCoV3  # noqa: F401
from . import ssl_resnet50 as _ssl  # noqa: E402

_ssl.register(__name__)  # evals.models.{barlowtwins, byol, ..., swav}: one table, fourteen importable sub-modules
