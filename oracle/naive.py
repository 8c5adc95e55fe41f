"""Naive numpy 'second opinion' for each primitive of the oracle (float64, explicit loops).

TEST INFRASTRUCTURE (see oracle/__init__.py).  These follow the mathematical definition
of each MindSpore primitive rather than any library kernel, and are
used only to cross-check oracle/ldm.py at tiny shapes.
"""
import math

import numpy as np


def conv2d(x, w, b, stride=1, padding=1):
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    ho = (h + 2 * padding - kh) // stride + 1
    wo = (wd + 2 * padding - kw) // stride + 1
    xp = np.zeros((n, cin, h + 2 * padding, wd + 2 * padding), dtype=np.float64)
    xp[:, :, padding:padding + h, padding:padding + wd] = x
    out = np.zeros((n, cout, ho, wo), dtype=np.float64)
    for i in range(ho):
        for j in range(wo):
            patch = xp[:, :, i * stride:i * stride + kh, j * stride:j * stride + kw]
            out[:, :, i, j] = np.tensordot(patch, w.astype(np.float64), axes=([1, 2, 3], [1, 2, 3]))
    return out + b.reshape(1, -1, 1, 1)


def group_norm(x, gamma, beta, eps, groups=32):
    n, c = x.shape[:2]
    out = np.empty_like(x, dtype=np.float64)
    cpg = c // groups
    for i in range(n):
        for g in range(groups):
            sl = x[i, g * cpg:(g + 1) * cpg].astype(np.float64)
            mu = sl.sum() / sl.size
            var = ((sl - mu) ** 2).sum() / sl.size
            out[i, g * cpg:(g + 1) * cpg] = (sl - mu) / math.sqrt(var + eps)
    shape = (1, c) + (1,) * (x.ndim - 2)
    return out * gamma.reshape(shape) + beta.reshape(shape)


def layer_norm(x, gamma, beta, eps):
    x = x.astype(np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def silu(x):
    return x / (1.0 + np.exp(-x))


def gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def attention(q, k, v, heads):
    """softmax(q k^T / sqrt(d)) v per head; q [b,n,h*d], k/v [b,m,h*d]."""
    b, n, c = q.shape
    d = c // heads
    out = np.zeros((b, n, c), dtype=np.float64)
    for bi in range(b):
        for hi in range(heads):
            qs = q[bi, :, hi * d:(hi + 1) * d].astype(np.float64)
            ks = k[bi, :, hi * d:(hi + 1) * d].astype(np.float64)
            vs = v[bi, :, hi * d:(hi + 1) * d].astype(np.float64)
            s = qs @ ks.T * d ** -0.5
            s = s - s.max(-1, keepdims=True)
            p = np.exp(s)
            p = p / p.sum(-1, keepdims=True)
            out[bi, :, hi * d:(hi + 1) * d] = p @ vs
    return out


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    out = np.zeros((len(t), dim), dtype=np.float64)
    for i, tv in enumerate(t):
        for k in range(half):
            f = math.exp(-math.log(max_period) * k / half)
            out[i, k] = math.cos(tv * f)
            out[i, half + k] = math.sin(tv * f)
    return out
