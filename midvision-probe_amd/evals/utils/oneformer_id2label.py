"""Stuff / things split of the OneFormer ADE20K-150 panoptic ids used by the validation metrics
(reference: evals/utils/oneformer_id2label.py:154-303 — data, consumed by metrics.py:180-190).  22 ids are "stuff";
every other id in 0..149 is a "thing" except 11, 17, 40 and 68, which the reference lists in neither group."""
STUFF = [0, 1, 2, 3, 4, 5, 6, 9, 13, 16, 21, 26, 29, 46, 52, 60, 91, 94, 96, 106, 113, 128]
_NEITHER = (11, 17, 40, 68)
THINGS = [i for i in range(150) if i not in STUFF and i not in _NEITHER]
