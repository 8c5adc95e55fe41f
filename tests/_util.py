"""Shared helpers for the GPU parity tests: error metrics + a JSONL parity log under gpurun_out/."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_log.jsonl")


def to_np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().float().cpu().numpy()
    return np.asarray(t, dtype=np.float32)


def metrics(got, ref):
    got, ref = to_np(got).astype(np.float64), to_np(ref).astype(np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    diff = got - ref
    denom = np.sqrt((ref ** 2).sum()) + 1e-30
    return {
        "max_abs": float(np.abs(diff).max()) if diff.size else 0.0,
        "rel_l2": float(np.sqrt((diff ** 2).sum()) / denom),
        "ref_max": float(np.abs(ref).max()) if ref.size else 0.0,
        "finite": bool(np.isfinite(got).all()),
    }


def check(name, got, ref, rel_l2=None, max_abs=None, max_rel=None, abs_q=None, **extra):
    """Log and assert.  Tolerances are stated at the call site (fp16 storage / fp32 accumulate vs fp32 oracle)."""
    m = metrics(got, ref)
    rec = dict(name=name, **m, tol_rel_l2=rel_l2, tol_max_abs=max_abs, tol_max_rel=max_rel, **extra)
    if abs_q is not None:  # (quantile, bound): robust to the few elements a clipped / chaotic trajectory flips
        rec["abs_quantile"] = float(np.quantile(np.abs(to_np(got).astype(np.float64) - to_np(ref).astype(np.float64)), abs_q[0]))
        rec["tol_abs_quantile"] = [float(abs_q[0]), float(abs_q[1])]
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    print("PARITY", json.dumps(rec))
    assert m["finite"], f"{name}: non-finite output"
    if rel_l2 is not None:
        assert m["rel_l2"] <= rel_l2, f"{name}: rel_l2 {m['rel_l2']:.3e} > {rel_l2:.1e} (max_abs {m['max_abs']:.3e})"
    if max_abs is not None:
        assert m["max_abs"] <= max_abs, f"{name}: max_abs {m['max_abs']:.3e} > {max_abs:.1e}"
    if max_rel is not None:  # max |d| relative to the largest reference magnitude
        assert m["max_abs"] <= max_rel * max(m["ref_max"], 1e-30), \
            f"{name}: max_abs {m['max_abs']:.3e} > {max_rel:.1e} * ref_max {m['ref_max']:.3e}"
    if abs_q is not None:
        qv = rec["abs_quantile"]
        assert qv <= abs_q[1], f"{name}: |d| quantile {abs_q[0]} = {qv:.3e} > {abs_q[1]:.1e}"
    return m


def h16(a):
    """Round an fp32 numpy array to fp16 precision (so oracle and kernel see identical inputs)."""
    return np.asarray(a, dtype=np.float16).astype(np.float32)
