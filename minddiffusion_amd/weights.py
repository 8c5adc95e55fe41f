"""Synthetic, seeded weights for the UNet (there is no network: the reference's checkpoints are download
links).  Keys and shapes follow the reference's parameter names so that a
real checkpoint converted to {name: array} drops into ``UNetModel.load_state_dict`` unchanged.

Fan-in-scaled normals keep activations O(1) through all ~100 layers; ``zero_init=True`` reproduces the
reference constructor's ``zero_module`` layers (openaimodel.py:162-165,524; attention.py:223-231), for
which a freshly built UNet outputs exactly 0.
"""
import math

import numpy as np
import torch

_ZERO_TAGS = ("out_layers_conv.conv.", "out.2.conv.", ".proj_out.")


def _is_zero_module(name):
    return any(t in name for t in _ZERO_TAGS) or name.startswith("out.2.")


def synthetic_unet_params_numpy(shapes, seed=0, zero_init=False):
    """numpy RandomState stream (bit-reproducible on any host)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes.items():
        if name.endswith(".gamma"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".beta"):
            v = 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.05 * rng.standard_normal(shape)
        else:
            v = rng.standard_normal(shape) / math.sqrt(int(np.prod(shape[1:])))
        if zero_init and _is_zero_module(name):
            v = np.zeros(shape)
        out[name] = v.astype(np.float32)
    return out


def synthetic_unet_params_device(shapes, seed=0, device="cuda:0", zero_init=False):
    """Same distribution, generated on the GPU (fast for the 866 M-parameter models; used by bench.py)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        if zero_init and _is_zero_module(name):
            out[name] = torch.zeros(shape, device=device)
            continue
        r = torch.randn(shape, device=device, generator=g)
        if name.endswith(".gamma"):
            v = 1.0 + 0.1 * r
        elif name.endswith(".beta"):
            v = 0.1 * r
        elif name.endswith(".bias"):
            v = 0.05 * r
        else:
            v = r / math.sqrt(int(np.prod(shape[1:])))
        out[name] = v
    return out


def check_state_dict(shapes, params, strict=True, who="load_state_dict", ignore_prefixes=()):
    """Shared key / shape validation of every ``load_state_dict`` here: `shapes` = the names and shapes this object owns
    (reference parameter names), `params` = what the caller passed.  Returns (missing, unexpected).
    strict=True raises on EITHER list -- a renamed key in a real checkpoint must not be dropped silently
    (``ms.load_param_into_net`` returns the not-loaded list; the reference's CLIs print it) -- and always on a shape
    mismatch.  Keys under `ignore_prefixes` belong to a sibling object sharing the same dict."""
    missing = [k for k in shapes if k not in params]
    unexpected = [k for k in params if k not in shapes and not any(k.startswith(p) for p in ignore_prefixes)]
    if strict and (missing or unexpected):
        parts = []
        if missing:
            parts.append(f"{len(missing)} missing, e.g. {missing[:3]}")
        if unexpected:
            parts.append(f"{len(unexpected)} unexpected, e.g. {unexpected[:3]}")
        raise KeyError(f"{who}: " + "; ".join(parts))
    for k, shp in shapes.items():
        if k in params and tuple(np.shape(params[k])) != tuple(shp):
            raise ValueError(f"{who}: {k} has shape {tuple(np.shape(params[k]))}, expected {tuple(shp)}")
    return missing, unexpected
