"""Shape schedule of the reference's variable-batch sampler, as a host-side driver for benchmarks (SURVEY §8f "next" row 3).

``image_batch_pairs`` restates data/sampler/utils.py:13-67 (+ create_intervallic_integer_list :123-146): the (H, W, batch) tuples the
sampler draws from — `max_scales` sizes between the min and max crop, the base size added, each rounded to `check_scale_div_factor`,
batch scaled so that H*W*batch stays at the base budget.  ``vbs_sequence`` draws per-step tuples the way
VariableBatchSamplerDDP.__iter__ does (variable_batch_sampler.py:315-320: random.choice after random.seed(epoch),
base_sampler.py:233), identically on every rank.  tests/test_dropin_cpu.py checks both against the reference module itself."""
from __future__ import annotations

import random
from typing import List, Tuple


def _make_divisible(v, divisor=8, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _intervallic(base, lo, hi, num_scales, div) -> List[int]:
    if num_scales == 1:
        vals = {float(lo)}
    else:
        step = (hi - lo) / (num_scales - 1)
        vals = {lo + step * i for i in range(num_scales - 1)} | {float(hi)}  # numpy.linspace end-point semantics
    vals.add(base)
    return sorted(_make_divisible(v, div) for v in vals)


def image_batch_pairs(crop_size_w: int, crop_size_h: int, batch_size_gpu0: int, max_scales: int = 5, check_scale_div_factor: int = 32,
                      min_crop_size_w: int = 160, max_crop_size_w: int = 320, min_crop_size_h: int = 160, max_crop_size_h: int = 320
                      ) -> List[Tuple[int, int, int]]:
    width_dims = _intervallic(crop_size_w, min_crop_size_w, max_crop_size_w, max_scales, check_scale_div_factor)
    height_dims = _intervallic(crop_size_h, min_crop_size_h, max_crop_size_h, max_scales, check_scale_div_factor)
    n_elements = crop_size_w * crop_size_h * batch_size_gpu0
    out = set()
    for h, w in zip(height_dims, width_dims):
        out.add((h, w, max(1, int(round(n_elements / (h * w), 2)))))
    return sorted(out)


def vbs_sequence(pairs: List[Tuple[int, int, int]], n_steps: int, epoch: int = 0) -> List[Tuple[int, int, int]]:
    rng = random.Random(epoch)  # random.seed(epoch) then one random.choice per batch
    return [rng.choice(pairs) for _ in range(n_steps)]
