"""Parity of the eight-wave 256 x 128 dense core (minddiffusion_amd/csrc/gemm8p.hip) -- the launch form of nn.Dense / 1x1 convs over
token rows (attention.py:44, 66, 108-112, 212, 231) for M >= 4096 -- through the C-ABI.

It issues the same `v_mfma_f32_32x32x16_f16` over the same k order as the four-wave kernel of gemm.hip and runs the same epilogue code,
so every output must be BIT-IDENTICAL to the un-split four-wave launch (which tests/test_kernels_gpu.py pins against the oracle): every
epilogue (bias / residual, GEGLU, LayerNorm fold producer + consumer, q|k + V^T split output, row / column statistics), ragged M and
N, and a race screen at the benchmarked shapes.
"""
import math

import numpy as np
import pytest
import torch

from _util import check, h16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from minddiffusion_amd import ops as _ops
    return _ops


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float16)


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float32)


def _both(ops, build, q256=True, exact=None):
    """build(**form) -> (descriptor, outputs, keepalive).  Runs the eight-wave 256 x 128 form, its 256 x 256 form (gemm8q_kernel:
    the same MFMA over the same k order, the same epilogue code -- `exact`: which outputs must be bit-identical, default all) and the
    un-split four-wave form."""
    res = []
    forms = [dict(tile_m=256, stages=8), dict(tile_m=128, splitk=1)] + ([dict(tile_m=256, tile_n=256, stages=8)] if q256 else [])
    for form in forms:
        d, outs, keep = build(**form)
        q = ops.gemm_query(d)
        assert (q[0] == 256) == ("stages" in form), (form, q)
        assert q[1] == form.get("tile_n", q[1]), (form, q)
        for o in outs:
            o.fill_(float("nan"))
        ops.gemm_run(d)
        torch.cuda.synchronize()
        res.append([o.clone() for o in outs])
    for other in res[1:]:
        for i, (a, b) in enumerate(zip(res[0], other)):
            if exact is None or i in exact:
                assert torch.equal(a, b), "eight-wave and four-wave launches differ"
    return res[-1] if q256 else res[0]


@pytest.mark.parametrize("M,N,K,bias,res", [(512, 128, 128, False, False), (4096, 640, 640, True, True), (1000, 192, 320, True, False),
                                             (256, 1280, 2560, True, True), (4096, 128, 64 * 3, False, True)])
def test_gemm8p_plain(ops, M, N, K, bias, res):
    rng = np.random.RandomState(M + N + K)
    a = h16(rng.standard_normal((M, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
    bv = rng.standard_normal(N).astype(np.float32)
    r = h16(rng.standard_normal((M, N)))
    ad, wp, bd, rd = dev16(a), ops.pack_gemm_weight(dev16(w)), dev32(bv), dev16(r)
    nrb = (M + 255) // 256

    def build(**form):
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        st = torch.full((M, N // 64, 2), float("nan"), dtype=torch.float32, device=DEV)
        d = ops.make_gemm_desc(ad, wp, N, 1, M, 1, K, out, N, bias=bd if bias else None, residual=rd if res else None,
                               residual_ld=N if res else 0, stats_out=st, **form)
        return d, (out, st), None
    out, st = _both(ops, build, q256=(N % 64 == 0 and K % 64 == 0 and K >= 128))
    ref = a @ w.T + (bv if bias else 0) + (r if res else 0)
    check(f"gemm8p_plain_M{M}_N{N}_K{K}", out, ref, rel_l2=1e-3)
    xr = out.float()
    check(f"gemm8p_rowstats_M{M}_N{N}", st[:, :, 0], xr.reshape(M, N // 64, 64).sum(-1), rel_l2=1e-5)
    assert nrb >= 1


def test_gemm8p_colstats(ops):
    rng = np.random.RandomState(4)
    B, T, K, N = 3, 512, 320, 320
    a = h16(rng.standard_normal((B * T, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
    ad, wp = dev16(a), ops.pack_gemm_weight(dev16(w))
    rd = dev16(h16(rng.standard_normal((B * T, N))))
    out = torch.empty((B * T, N), dtype=torch.float16, device=DEV)
    cs = torch.full((B * T // 256, N, 2), float("nan"), dtype=torch.float32, device=DEV)
    for tn in (0, 256):
        cs.fill_(float("nan"))
        d = ops.make_gemm_desc(ad, wp, N, B, T, 1, K, out, N, residual=rd, residual_ld=N, colstats_out=cs, tile_m=256, tile_n=tn, stages=8)
        q = ops.gemm_query(d)
        assert q[5] == 256 and (tn == 0 or q[1] == 256), q
        ops.gemm_run(d)
        torch.cuda.synchronize()
        o = out.float().reshape(B * T // 256, 256, N)
        check(f"gemm8p_colstats_sum_tn{tn}", cs[:, :, 0], o.sum(1), rel_l2=1e-5)
        check(f"gemm8p_colstats_sumsq_tn{tn}", cs[:, :, 1], (o * o).sum(1), rel_l2=1e-5)


@pytest.mark.parametrize("M,C,kind", [(4096, 640, "geglu"), (1024, 320, "qkv"), (768, 128, "plain"), (2048, 1280, "geglu")])
def test_gemm8p_layernorm_fold_consumers(ops, M, C, kind):
    """GEGLU / q|k + V^T / plain consumers of the LayerNorm fold on raw token rows with the producer's row statistics."""
    rng = np.random.RandomState(M + C)
    x = h16(rng.standard_normal((M, C)) + 0.7)
    g = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    be = (0.3 * rng.standard_normal(C)).astype(np.float32)
    nout = {"plain": C, "geglu": 8 * C, "qkv": 3 * C}[kind]
    w1 = h16(rng.standard_normal((nout, C)) / math.sqrt(C))
    b1 = rng.standard_normal(nout).astype(np.float32)
    xd = dev16(x)
    xr = xd.float()
    stats = torch.stack([xr.reshape(M, C // 64, 64).sum(-1), (xr ** 2).reshape(M, C // 64, 64).sum(-1)], -1).contiguous()
    wg, s, cb = ops.fold_layernorm(dev16(w1), dev32(g), dev32(be), dev32(b1))
    wgp = ops.pack_gemm_weight(wg)

    def build(**form):
        if kind == "plain":
            out = torch.empty((M, C), dtype=torch.float16, device=DEV)
            return ops.make_gemm_desc(xd, wgp, C, 1, M, 1, C, out, C, bias=cb, ln_stats=stats, ln_s=s, **form), (out,), None
        if kind == "geglu":
            out = torch.empty((M, 4 * C), dtype=torch.float16, device=DEV)
            return ops.make_gemm_desc(xd, wgp, 8 * C, 1, M, 1, C, out, 4 * C, bias=cb, epilogue=ops.EPI_GEGLU, ln_stats=stats,
                                      ln_s=s, **form), (out,), None
        B, T = 2, M // 2
        qk = torch.empty((B, T, 2 * C), dtype=torch.float16, device=DEV)
        vt = torch.zeros((B, C, T), dtype=torch.float16, device=DEV)
        return ops.make_gemm_desc(xd, wgp, 3 * C, B, T, 1, C, qk, 2 * C, bias=cb, out2=vt, out2_ld=T, n_split=2 * C,
                                  ln_stats=stats, ln_s=s, **form), (qk, vt), None
    outs = _both(ops, build, q256=(kind != "qkv" or (2 * C) % 256 == 0))
    assert all(torch.isfinite(o.float()).all() for o in outs)


@pytest.mark.parametrize("M,N,K,epi", [(16384, 5120, 640, 1), (16384, 640, 2560, 0), (4096, 10240, 1280, 1), (4608, 1280, 5120, 0)])
def test_gemm8p_benchmarked_shapes_race_screen(ops, M, N, K, epi):
    """The token-GEMM shapes of BASELINE configs 2 / 3 on this core (option gemm_dense8p), 12 launches each on a NaN-poisoned output with cold weight
    copies: bit-stable, and bit-identical to the four-wave launch."""
    rng = np.random.RandomState(M + N)
    a = dev16(h16(rng.standard_normal((M, K))))
    ws = [ops.pack_gemm_weight(dev16(h16(rng.standard_normal((N, K)) / math.sqrt(K)))) for _ in range(2)]
    bd = dev32(rng.standard_normal(N).astype(np.float32))
    ncols = N // 2 if epi else N
    outs = [torch.empty((M, ncols), dtype=torch.float16, device=DEV) for _ in range(2)]
    ops.set_option("gemm_dense8p", 1)       # the automatic route (off by default: equal-or-slower in situ)
    try:
        descs = [ops.make_gemm_desc(a, ws[i], N, 1, M, 1, K, outs[i], ncols, bias=bd, epilogue=epi) for i in range(2)]
        q = ops.gemm_query(descs[0])
        assert q[0] == 256 and q[1] == 128 and q[2] == 1, q
        _race(ops, descs, outs, a, ws, N, M, K, bd, epi)
    finally:
        ops.set_option("gemm_dense8p", 0)


def _race(ops, descs, outs, a, ws, N, M, K, bd, epi):
    firsts = []
    for i in range(2):
        outs[i].fill_(float("nan"))
        ops.gemm_run(descs[i])
        firsts.append(outs[i].clone())
    for rep in range(12):
        i = rep % 2
        outs[i].fill_(float("nan"))
        ops.gemm_run(descs[i])
        assert torch.equal(outs[i], firsts[i]), f"launch {rep}: output changed"
    old = ops.gemm(a, ws[0], N, 1, M, 1, K, bias=bd, epilogue=epi, tile_m=128, splitk=1)
    assert torch.equal(firsts[0], old)


@pytest.mark.parametrize("M,N,K,epi", [(16384, 5120, 640, 1), (16384, 1920, 640, 0), (4096, 3840, 1280, 0), (18432, 5120, 640, 1), (4608, 10240, 1280, 1)])
def test_gemm8q_benchmarked_shapes_race_screen(ops, M, N, K, epi):
    """The GEGLU ff1 / q|k|v shapes of BASELINE configs 2 / 3 on the 256 x 256 form (automatic route, option gemm_dense8q = 1 -- off by default), 12 launches
    each on a NaN-poisoned output with cold weight copies: bit-stable (the counted vmcnt / barrier schedule of the four-slab ring
    has no timing-dependent read), and bit-identical to the four-wave launch."""
    rng = np.random.RandomState(M + N + 1)
    a = dev16(h16(rng.standard_normal((M, K))))
    ws = [ops.pack_gemm_weight(dev16(h16(rng.standard_normal((N, K)) / math.sqrt(K)))) for _ in range(2)]
    bd = dev32(rng.standard_normal(N).astype(np.float32))
    ncols = N // 2 if epi else N
    outs = [torch.empty((M, ncols), dtype=torch.float16, device=DEV) for _ in range(2)]
    for var in (0, 32):      # DMA issue in the MFMA burst / in the read burst
        ops.set_option("gemm_dense8q", 1)
        ops.set_option("gemm_dense8q_var", var)
        try:
            descs = [ops.make_gemm_desc(a, ws[i], N, 1, M, 1, K, outs[i], ncols, bias=bd, epilogue=epi) for i in range(2)]
            q = ops.gemm_query(descs[0])
            assert q[0] == 256 and q[1] == 256 and q[2] == 1, q
            _race(ops, descs, outs, a, ws, N, M, K, bd, epi)
        finally:
            ops.set_option("gemm_dense8q", 0)
            ops.set_option("gemm_dense8q_var", 0)


def test_gemm8q_shape_gate(ops):
    """The automatic route takes 256 x 256 tiles only where they fit: not for N = 640 / 1280 outputs (padding / too few tiles), not
    below gemm_dense8p_min_m rows, not with the option off."""
    a = torch.zeros((16384, 640), dtype=torch.float16, device=DEV)

    def q(M, N, K, **kw):
        w = ops.pack_gemm_weight(torch.zeros((N, K), dtype=torch.float16, device=DEV))
        out = torch.empty((M, N), dtype=torch.float16, device=DEV)
        return ops.gemm_query(ops.make_gemm_desc(a[:M, :K].contiguous(), w, N, 1, M, 1, K, out, N, **kw))
    assert q(16384, 5120, 640)[1] != 256         # off by default (measured equal-or-slower inside an evaluation)
    ops.set_option("gemm_dense8q", 1)
    try:
        assert q(16384, 5120, 640)[:2] == (256, 256)
        assert q(16384, 640, 640)[1] != 256          # 3 N tiles of 256 for 640 columns: 17 % padding
        assert q(2048, 5120, 640)[1] != 256          # UNet batch 2
    finally:
        ops.set_option("gemm_dense8q", 0)
