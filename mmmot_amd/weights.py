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


# ---- "trained-like" weight statistics (robustness suites: tests/test_robust_*.py) -------------------------------
# The He-normal / var in [0.5, 1.5] generator above gives every channel of a layer the same gain.  A trained,
# BatchNorm-folded VGG does not look like that: gamma / sqrt(running_var + eps) spans orders of magnitude across the
# channels of one layer and the weights are heavy-tailed.  Two profiles, both counter-based like gen_tensor:
#   'wild'       - BN running_var log-uniform 1e-2 .. 1e2, gamma log-uniform 1e-2 .. 3 (folded gain spread >= 1e4 in
#                  every layer), Student-t(3) conv weights; NOT calibrated: activation magnitudes drift from layer to
#                  layer, which is what drives values across the e4m3 (1792) and fp16 (65000) range limits;
#   'calibrated' - the same gamma / weight laws, but running_mean / running_var are set by ``calibrate_bn`` to the
#                  actual statistics of the pre-BN activations on sample crops (what training does), so activations
#                  stay O(gamma) while the folded per-channel gains still spread over orders of magnitude.
def gen_tensor_trained(name, shape, seed=0, profile='wild'):
    g = _rng(name + '#' + profile, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    is_bn = name.startswith('appearance.layers.') and len(shape) == 1
    if is_bn and leaf == 'running_var':
        return np.exp(g.uniform(np.log(1e-2), np.log(1e2), shape)).astype(np.float32)
    if is_bn and leaf == 'weight':
        return np.exp(g.uniform(np.log(1e-2), np.log(3.0), shape)).astype(np.float32)
    if is_bn and leaf == 'running_mean':
        return (0.3 * g.standard_normal(shape)).astype(np.float32)
    if name.startswith('appearance.layers.') and len(shape) == 4:
        fan_in = int(np.prod(shape[1:]))
        t = g.standard_t(3, shape) / np.sqrt(3.0)           # unit variance, heavy tails
        return (np.sqrt(2.0 / fan_in) * t).astype(np.float32)
    return gen_tensor(name, shape, seed)


def generate_state_dict_trained(spec, seed=0, profile='wild'):
    out = {}
    for k, v in spec.items():
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        out[k] = torch.from_numpy(np.ascontiguousarray(gen_tensor_trained(k, shape, seed, profile)))
    return out


def calibrate_bn(sd, crops, eps=1e-5):
    """Set every VGG BatchNorm's running_mean / running_var to the statistics of its input on ``crops`` (in place on
    the state_dict; plain torch on the CPU - weight preparation, not the product path)."""
    import torch.nn.functional as F
    from .pack import VGG_STAGES
    x = crops.double()
    for s, stage in enumerate(VGG_STAGES):
        for (idx, cin, cout, pool) in stage:
            p = 'appearance.layers.%d.' % s
            y = F.conv2d(x, sd[p + '%d.weight' % idx].double(), sd[p + '%d.bias' % idx].double(), padding=1)
            mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
            sd[p + '%d.running_mean' % (idx + 1)] = mean.float()
            sd[p + '%d.running_var' % (idx + 1)] = var.clamp_min(1e-12).float()
            g, b = sd[p + '%d.weight' % (idx + 1)].double(), sd[p + '%d.bias' % (idx + 1)].double()
            x = F.relu((y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps) * g.view(1, -1, 1, 1) +
                       b.view(1, -1, 1, 1))
            if pool:
                x = F.max_pool2d(x, 2, 2)
    return sd


def state_dict_for_profile(spec, weights='default:0', S=64):
    """State dict for the golden-fixture manifest's ``weights`` field: ``'default:<seed>'`` = generate_state_dict (the
    He-normal law of every round-1..4 fixture, seed 0) or ``'calibrated:<seed>'`` = the trained-like 'calibrated'
    profile (per-channel folded gains over > 1e4, heavy-tailed weights) with the VGG BatchNorm statistics calibrated on
    sixteen sample crops of side ``S`` (seed 4100 + <seed>) - the recipe of tests/test_robust_gpu.py."""
    kind, _, seed = weights.partition(':')
    seed = int(seed or 0)
    if kind == 'default':
        return generate_state_dict(spec, seed)
    if kind == 'calibrated':
        from .synth import make_pair
        sd = generate_state_dict_trained(spec, seed, 'calibrated')
        return calibrate_bn(sd, make_pair(8, 8, S, 4, seed=4100 + seed)[0])
    raise ValueError('unknown weights profile %r' % (weights,))
