"""evals.utils.optim — LR schedule of the trainers (evals/utils/optim.py:124-133). Host scalar math."""
import math


def cosine_decay_linear_warmup(current_step, max_step, warmup_step, min_factor=0.01):
    assert max_step > warmup_step
    range_factor = 1 - min_factor
    if current_step <= warmup_step:
        return range_factor * (current_step / warmup_step) + min_factor
    rel_step = (current_step - warmup_step) / (max_step - warmup_step)
    return range_factor * math.cos(0.5 * rel_step * math.pi) + min_factor
