"""ORACLE helper — test infrastructure, NOT product code.

Deterministic, machine-independent parameter values keyed by state_dict name, so that the
reference (authoring container), the oracle restatement and the HIP product path (GPU box) can
all be loaded with *identical* weights without committing multi-MB checkpoints.  Uses
numpy's PCG64 streams (stable across platforms/versions) seeded by crc32(name).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def seeded_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith("running_mean"):
        a = 0.1 * rng.standard_normal(shape)
    elif name.endswith("running_var"):
        a = 1.0 + 0.1 * np.abs(rng.standard_normal(shape))
    elif len(shape) >= 2:  # conv / linear / embedding weights: unit-gain fan-in scaling
        fan_in = int(np.prod(shape[1:]))
        a = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
    elif name.endswith("weight"):  # norm scales
        a = 1.0 + 0.1 * rng.standard_normal(shape)
    else:  # biases
        a = 0.1 * rng.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def seeded_state_dict(shapes: Dict[str, Iterable[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def seeded_input(shape: Tuple[int, ...], seed: int = 0, name: str = "input") -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def seeded_labels(n: int, n_classes: int = 1000, seed: int = 0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(b"labels")]))
    return torch.from_numpy(rng.integers(0, n_classes, size=(n,)).astype(np.int64))


def seeded_caption_tokens(batch: int, ctx: int, vocab: int, seed: int) -> torch.Tensor:
    """synthetic captions for the CLIP text tower: ids in [1, vocab-2], one EOT (= vocab-1, the arg-max id read by
    text_encoders/transformer.py:413-421) at a random position >= 4, padding (0) after it."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tok = rng.integers(1, vocab - 1, size=(batch, ctx))
    for b in range(batch):
        e = int(rng.integers(4, ctx))
        tok[b, e] = vocab - 1
        tok[b, e + 1:] = 0
    return torch.from_numpy(tok.astype(np.int64))


def sample_indices(name: str, numel: int, n: int = 512) -> np.ndarray:
    """deterministic subset of a tensor's flat indices (large-batch fixtures store gradients only at these positions)"""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    rng = np.random.Generator(np.random.PCG64([7, zlib.crc32(name.encode())]))
    return np.sort(rng.choice(numel, size=n, replace=False)).astype(np.int64)
