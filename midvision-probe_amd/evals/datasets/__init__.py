"""Input side of the hot path: batch-dict contract + loader construction of the reference
(evals/datasets/builder.py:39-67, nyu.py:245-251).  Real dataset decoders (NYU .mat, Taskonomy, NAVI) are out of scope."""
from .builder import build_loader  # noqa: F401
from .synthetic import SyntheticNYU  # noqa: F401
