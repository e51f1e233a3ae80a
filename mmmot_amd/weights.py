"""Deterministic, key-addressed weight generator.

The reference publishes its checkpoints on Google Drive only (reference
README.md:67,88) and there is no network here, so every parity test and the
benchmark run on random-init weights of the reference architecture.  The
generator is *counter based*: each tensor is produced from
``Philox(key=[crc32(name), seed])`` so the value of a tensor depends only on
its state_dict key, its shape and the seed - never on module construction
order.  The same tensors can therefore be regenerated bit-identically on the
GPU box, by the oracle, and by the golden-vector script that loads them into
the imported reference.

Distributions are chosen so that no normalisation layer is trivial
(gamma != 1, beta != 0, running stats != (0,1)) and the STN ``output`` layers
are non-zero (the reference zero-initialises them, modules/point_net.py:69-70,
which would hide transform-folding bugs).
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed]))


def gen_tensor(name, shape, seed=0):
    """Return the float32 numpy tensor for state_dict key ``name``."""
    g = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.asarray(1000, dtype=np.int64)
    if leaf == 'idt':  # STN identity buffer (modules/point_net.py:62)
        return np.eye(shape[0], dtype=np.float32)
    if leaf == 'running_mean':
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == 'running_var':
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1:
        if leaf == 'weight':  # norm gamma
            return g.uniform(0.5, 1.5, shape).astype(np.float32)
        # conv / linear / norm bias
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    # conv / linear weight: He-normal over fan_in keeps activations O(1)
    fan_in = int(np.prod(shape[1:]))
    std = np.sqrt(2.0 / fan_in)
    if '.output.' in name:  # STN regression head: small but non-zero
        std = 0.05 / np.sqrt(fan_in)
    return (std * g.standard_normal(shape)).astype(np.float32)


def generate_state_dict(spec, seed=0):
    """spec: mapping key -> shape (e.g. from ``module.state_dict()``)."""
    out = {}
    for k, v in spec.items():
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        out[k] = torch.from_numpy(np.ascontiguousarray(gen_tensor(k, shape, seed)))
    return out


def init_module(module, seed=0):
    """Fill ``module`` (ours or the imported reference) with generated weights."""
    sd = generate_state_dict(module.state_dict(), seed)
    missing = module.load_state_dict(sd, strict=True)
    return missing
