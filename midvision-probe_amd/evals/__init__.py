"""Drop-in mirror of the reference's ``evals`` package for the feature-extraction +
probe-training hot path (same module paths, class names, constructor kwargs and attributes
that hydra ``_target_`` strings and the trainers use), backed by the HIP kernels in ``mvp``."""
