"""GPU parity of the row-local fused SpatialTransformer tail (mdx_st_tail_f16, csrc/stchain.hip) through the C-ABI.

Three references for the same seeded inputs (fp16-rounded, so every side sees identical operands):
  * `chain_ref(round16=True)`: the cited reference lines (attention.py:41-70, 96-166, 176-185, 231, 256) restated in float64
    with a rounding to fp16 at exactly the tensors the kernel keeps in fp16 (t1, LN2, q2, P, o2, t2, LN3, GEGLU output, t3,
    out) -- stage by stage through the kernel's debug taps (tolerance: one fp16 rounding + fp32 accumulation order, 1e-3);
  * `chain_ref(round16=False)`: the same lines in float64 with no rounding at all = the all-fp32 oracle (tolerance 3e-3, the
    single-UNet-call bar of SURVEY 8(c));
  * the UNFUSED chain of the existing kernels (mdx_gemm_f16 with LayerNorm fold + mdx_attention_f16), i.e. what the UNet ran
    before: two fp16 paths with different rounding points, 2e-3.
"""
import numpy as np
import pytest
import torch

from _util import check, h16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def make_case(seed, B, tokens, C, heads, ctx_len, ctx_dim, ctx_cap=80, wscale=1.0):
    rng = np.random.RandomState(seed)
    M = B * tokens
    w = {}
    for name, shape in (("o1", (C, C)), ("q2", (C, C)), ("k2", (C, ctx_dim)), ("v2", (C, ctx_dim)), ("o2", (C, C)),
                        ("ff1", (8 * C, C)), ("ff2", (C, 4 * C)), ("po", (C, C))):
        w[name] = h16(rng.standard_normal(shape) * wscale / np.sqrt(shape[1]))
    for name, n in (("bo1", C), ("bo2", C), ("b1", 8 * C), ("b2", C), ("bpo", C), ("be2", C), ("be3", C)):
        w[name] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    for name in ("g2", "g3"):
        w[name] = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    x = dict(attn_o=h16(rng.standard_normal((M, C))), tok=h16(rng.standard_normal((M, C))), x_in=h16(rng.standard_normal((M, C))),
             ctx=h16(rng.standard_normal((B, ctx_len, ctx_dim))))
    # the cached context projections, as the unfused path's context GEMMs store them (fp16)
    k = np.zeros((B, ctx_cap, C), np.float32)
    v = np.zeros((B, ctx_cap, C), np.float32)
    k[:, :ctx_len] = h16(x["ctx"].astype(np.float64) @ w["k2"].astype(np.float64).T)
    v[:, :ctx_len] = h16(x["ctx"].astype(np.float64) @ w["v2"].astype(np.float64).T)
    x["k"], x["vt"] = k, np.ascontiguousarray(v.transpose(0, 2, 1))
    return w, x


def chain_ref(w, x, B, tokens, C, heads, ctx_len, round16, eps=1e-5):
    r = (lambda a: h16(a).astype(np.float64)) if round16 else (lambda a: a)
    f = lambda a: np.asarray(a, np.float64)
    d = C // heads
    st = {}
    t1 = r(f(x["attn_o"]) @ f(w["o1"]).T + f(w["bo1"]) + f(x["tok"]))
    st[1] = t1
    ln2 = r(layer_norm(t1, f(w["g2"]), f(w["be2"]), eps))
    st[2] = ln2
    q2 = r(ln2 @ f(w["q2"]).T)
    st[3] = q2
    q = q2.reshape(B, tokens, heads, d).transpose(0, 2, 1, 3)
    k = f(x["k"])[:, :ctx_len].reshape(B, ctx_len, heads, d).transpose(0, 2, 1, 3)
    v = f(x["vt"]).transpose(0, 2, 1)[:, :ctx_len].reshape(B, ctx_len, heads, d).transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2) * d ** -0.5
    s = s - s.max(-1, keepdims=True)
    pexp = np.exp(s)
    # the kernel rounds the UNNORMALISED weights to fp16 (they feed the PV MFMA) and divides the fp32 sum afterwards
    o2 = (r(pexp) @ v) / pexp.sum(-1, keepdims=True)
    o2 = r(o2.transpose(0, 2, 1, 3).reshape(B * tokens, C))
    st[4] = o2
    t2 = r(o2 @ f(w["o2"]).T + f(w["bo2"]) + t1)
    st[5] = t2
    ln3 = r(layer_norm(t2, f(w["g3"]), f(w["be3"]), eps))
    st[6] = ln3
    y = ln3 @ f(w["ff1"]).T + f(w["b1"])
    hdn = r(y[:, :4 * C] * gelu_tanh(y[:, 4 * C:]))
    t3 = r(hdn @ f(w["ff2"]).T + f(w["b2"]) + t2)
    st[7] = t3
    st[0] = r(t3 @ f(w["po"]).T + f(w["bpo"]) + f(x["x_in"]))
    return st


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float16)


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float32)


def run_fused(ops, w, x, B, tokens, C, heads, ctx_len, tile_rows, stage=0, colstats=False, ctx_cap=80):
    stream, vec = ops.pack_st_tail(*(dev16(w[n]) for n in ("o1", "q2", "o2", "ff1", "ff2", "po")),
                                   *(dev32(w[n]) for n in ("bo1", "g2", "be2", "bo2", "g3", "be3", "b1", "b2", "bpo")))
    M = B * tokens
    t = {n: dev16(x[n]) for n in ("attn_o", "tok", "x_in", "k", "vt")}
    out = torch.full((M, C), float("nan"), dtype=torch.float16, device=DEV)
    dbg = torch.full((M, C), float("nan"), dtype=torch.float16, device=DEV) if stage else None
    cs = torch.zeros((M // tile_rows, C, 2), dtype=torch.float32, device=DEV) if colstats else None
    d = ops.make_st_tail_desc(t["attn_o"], t["tok"], t["x_in"], out, t["k"], t["vt"], stream, vec, B, tokens, C, heads, C // heads,
                              ctx_len, ctx_cap, tile_rows=tile_rows, colstats_out=cs, debug_out=dbg, debug_stage=stage)
    ops.st_tail_run(d)
    torch.cuda.synchronize()
    return (dbg if stage else out), cs


@pytest.fixture(scope="module")
def ops():
    from minddiffusion_amd import ops as _ops
    return _ops


STAGE_NAMES = {1: "t1", 2: "ln2", 3: "q2", 4: "xattn", 5: "t2", 6: "ln3", 7: "t3", 0: "out"}


@pytest.mark.parametrize("heads", [5, 8])
@pytest.mark.parametrize("tile_rows", [64, 32])
def test_st_tail_stages_vs_reference(ops, heads, tile_rows):
    """Every stage tap of the fused kernel against the fp16-storage restatement (1e-3) and the final output against the
    all-fp32 restatement (3e-3); heads = 5 (SDv2, d = 64) and 8 (Wukong-Huahua, d = 40)."""
    B, tokens, C, ctx_len = 2, 128, 320, 77
    w, x = make_case(7 + heads, B, tokens, C, heads, ctx_len, 1024 if heads == 5 else 768)
    ref16 = chain_ref(w, x, B, tokens, C, heads, ctx_len, True)
    ref32 = chain_ref(w, x, B, tokens, C, heads, ctx_len, False)
    for stage in (1, 2, 3, 4, 5, 6, 7, 0):
        got, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, tile_rows, stage)
        check(f"st_tail_h{heads}_r{tile_rows}_{STAGE_NAMES[stage]}_vs_fp16ref", got, ref16[stage], rel_l2=1e-3, max_rel=6e-3)
    got, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, tile_rows, 0)
    check(f"st_tail_h{heads}_r{tile_rows}_out_vs_fp32ref", got, ref32[0], rel_l2=3e-3)


@pytest.mark.parametrize("ctx_len", [1, 33, 80])
def test_st_tail_context_lengths(ops, ctx_len):
    """Key masking: one key (softmax == 1), a length that ends inside a 32-key tile, and the full capacity."""
    B, tokens, C, heads = 1, 64, 320, 5
    w, x = make_case(3, B, tokens, C, heads, ctx_len, 1024)
    ref16 = chain_ref(w, x, B, tokens, C, heads, ctx_len, True)
    got, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, 64, 4)
    check(f"st_tail_ctx{ctx_len}_xattn", got, ref16[4], rel_l2=1e-3, max_rel=6e-3)
    got, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, 64, 0)
    check(f"st_tail_ctx{ctx_len}_out", got, ref16[0], rel_l2=1e-3, max_rel=6e-3)


def test_st_tail_attention_outlier_keys(ops):
    """A context key that dominates the softmax by a wide margin and rows with large scores: the row maximum is exact (all
    <= 96 keys of a head are in registers at once), so there is no running-maximum regime to go stale."""
    B, tokens, C, heads, ctx_len = 1, 64, 320, 5, 77
    w, x = make_case(11, B, tokens, C, heads, ctx_len, 1024)
    x["k"][:, 50] *= 12.0
    x["k"][:, 3] *= -9.0
    x["k"] = h16(x["k"])
    ref16 = chain_ref(w, x, B, tokens, C, heads, ctx_len, True)
    got, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, 64, 4)
    check("st_tail_outlier_xattn", got, ref16[4], rel_l2=1e-3, max_rel=6e-3)


def unfused_chain(ops, w, x, B, tokens, C, heads, ctx_len, ctx_cap=80):
    """The launches the UNet plan emitted for the same block before the fused kernel existed (openaimodel.py transformer():
    to_out + residual + row statistics, LayerNorm-folded to_q, attention over the cached keys, ..., GEGLU epilogue)."""
    M = B * tokens
    d = C // heads
    t = {n: dev16(x[n]) for n in ("attn_o", "tok", "x_in", "k", "vt")}
    st = torch.zeros((M, C // 64, 2), dtype=torch.float32, device=DEV)

    def fold(name, norm, bias=None):
        wt, s, cb = ops.fold_layernorm(dev16(w[name]), dev32(w["g" + norm]), dev32(w["be" + norm]), bias)
        return ops.pack_gemm_weight(wt), s, cb
    pk = lambda n: ops.pack_gemm_weight(dev16(w[n]))
    t1 = ops.gemm(t["attn_o"], pk("o1"), C, B, tokens, 1, C, bias=dev32(w["bo1"]), residual=t["tok"], residual_ld=C, stats_out=st)
    wq, sq, cbq = fold("q2", "2")
    q2 = ops.gemm(t1, wq, C, B, tokens, 1, C, bias=cbq, ln_stats=st, ln_s=sq)
    o2 = torch.empty_like(q2)
    ops.attention(q2.data_ptr(), t["k"].data_ptr(), t["vt"].data_ptr(), o2.data_ptr(), B, heads, d, tokens, ctx_len, d ** -0.5,
                  tokens * C, C, ctx_cap * C, C, C * ctx_cap, ctx_cap, tokens * C, C)
    t2 = ops.gemm(o2, pk("o2"), C, B, tokens, 1, C, bias=dev32(w["bo2"]), residual=t1, residual_ld=C, stats_out=st)
    half = 4 * C
    nt = half // 64
    gw, gb = dev16(w["ff1"]), dev32(w["b1"])
    b1i = torch.stack([gb[:half].reshape(nt, 64), gb[half:].reshape(nt, 64)], 1).reshape(-1).contiguous()
    w1i = torch.stack([gw[:half].reshape(nt, 64, C), gw[half:].reshape(nt, 64, C)], 1).reshape(2 * half, C)
    wt, s1, cb1 = ops.fold_layernorm(w1i, dev32(w["g3"]), dev32(w["be3"]), b1i)
    g = ops.gemm(t2, ops.pack_gemm_weight(wt), 8 * C, B, tokens, 1, C, bias=cb1, ln_stats=st, ln_s=s1, epilogue=ops.EPI_GEGLU)
    t3 = ops.gemm(g, pk("ff2"), C, B, tokens, 1, 4 * C, bias=dev32(w["b2"]), residual=t2, residual_ld=C)
    out = ops.gemm(t3, pk("po"), C, B, tokens, 1, C, bias=dev32(w["bpo"]), residual=t["x_in"], residual_ld=C)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("heads", [5, 8])
def test_st_tail_vs_unfused_chain(ops, heads):
    """Fused launch vs the unfused launches it replaces, at a size with many row blocks (B = 2, 32 x 32 tokens), plus the
    column statistics it hands to the next GroupNorm against the sums of the fp16 tensor it stored."""
    B, tokens, C, ctx_len = 2, 1024, 320, 77
    w, x = make_case(21 + heads, B, tokens, C, heads, ctx_len, 1024 if heads == 5 else 768)
    ref = unfused_chain(ops, w, x, B, tokens, C, heads, ctx_len)
    ref32 = chain_ref(w, x, B, tokens, C, heads, ctx_len, False)[0]
    for tile_rows in (64, 32):
        got, cs = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, tile_rows, 0, colstats=True)
        check(f"st_tail_h{heads}_r{tile_rows}_vs_unfused", got, ref, rel_l2=2e-3)
        check(f"st_tail_h{heads}_r{tile_rows}_vs_fp32ref_big", got, ref32, rel_l2=3e-3)
        blk = got.float().reshape(-1, tile_rows, C)
        check(f"st_tail_h{heads}_r{tile_rows}_colstats_sum", cs[:, :, 0], blk.sum(1), rel_l2=1e-5)
        check(f"st_tail_h{heads}_r{tile_rows}_colstats_sumsq", cs[:, :, 1], (blk * blk).sum(1), rel_l2=1e-5)
    check(f"unfused_h{heads}_vs_fp32ref_big", ref, ref32, rel_l2=3e-3)


def test_st_tail_rejects_bad_arguments(ops):
    from minddiffusion_amd._lib import MdxError
    B, tokens, C, heads, ctx_len = 1, 64, 320, 5, 77
    w, x = make_case(1, B, tokens, C, heads, ctx_len, 1024)
    with pytest.raises(MdxError):
        run_fused(ops, w, x, B, tokens, C, heads, ctx_len, 48)          # tile_rows
    with pytest.raises(MdxError):
        run_fused(ops, w, x, B, tokens, C, heads, 81, 64)               # ctx_len > capacity
    assert not ops.st_tail_supported(640, 10, 64, 1024, 64)
    assert ops.st_tail_supported(320, 8, 40, 4096, 64)


# --------------------------------------------------------------------------- fused head (mdx_st_head_f16)
def group_norm_rows(x, B, tokens, C, g, b, eps, groups=32):
    """GroupNorm(32) of NHWC rows [B * tokens, C] (util.py:87-108; biased variance)."""
    xr = x.reshape(B, tokens, groups, C // groups)
    mu = xr.mean(axis=(1, 3), keepdims=True)
    var = ((xr - mu) ** 2).mean(axis=(1, 3), keepdims=True)
    return ((xr - mu) / np.sqrt(var + eps)).reshape(B * tokens, C) * g + b


def make_head_case(seed, B, tokens, C):
    rng = np.random.RandomState(seed)
    w = {n: h16(rng.standard_normal((C, C)) / np.sqrt(C)) for n in ("pi", "q", "k", "v")}
    for n in ("gn_b", "bpi", "be1"):
        w[n] = (0.1 * rng.standard_normal(C)).astype(np.float32)
    for n in ("gn_g", "g1"):
        w[n] = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    # per-channel offsets and scales so that the GroupNorm statistics matter
    x = h16(rng.standard_normal((B * tokens, C)) * (0.5 + rng.rand(C)) + rng.standard_normal(C))
    return w, x


def head_ref(w, x, B, tokens, C, round16):
    r = (lambda a: h16(a).astype(np.float64)) if round16 else (lambda a: a)
    f = lambda a: np.asarray(a, np.float64)
    st = {}
    st[1] = r(group_norm_rows(f(x), B, tokens, C, f(w["gn_g"]), f(w["gn_b"]), 1e-6))
    st[2] = r(st[1] @ f(w["pi"]).T + f(w["bpi"]))
    st[3] = r(layer_norm(st[2], f(w["g1"]), f(w["be1"]), 1e-5))
    st["q"], st["k"], st["v"] = (r(st[3] @ f(w[n]).T) for n in ("q", "k", "v"))
    return st


def run_head(ops, w, x, B, tokens, C, tile_rows, rows_per_stat_block, stage=0):
    M = B * tokens
    xd = dev16(x)
    stream, vec = ops.pack_st_head(*(dev16(w[n]) for n in ("pi", "q", "k", "v")),
                                   *(dev32(w[n]) for n in ("gn_g", "gn_b", "bpi", "g1", "be1")))
    nrb = tokens // rows_per_stat_block
    blk = xd.float().reshape(B * nrb, rows_per_stat_block, C)          # what the producer's epilogue would have emitted
    cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
    tok = torch.full((M, C), float("nan"), dtype=torch.float16, device=DEV)
    qk = torch.full((M, 2 * C), float("nan"), dtype=torch.float16, device=DEV)
    vt = torch.full((B, C, tokens), float("nan"), dtype=torch.float16, device=DEV)
    dbg = torch.full((M, C), float("nan"), dtype=torch.float16, device=DEV) if stage else None
    d = ops.make_st_head_desc(xd, cs, nrb, stream, vec, tok, qk, vt, tokens, B, tokens, C, tile_rows=tile_rows, debug_out=dbg,
                              debug_stage=stage)
    ops.st_head_run(d)
    torch.cuda.synchronize()
    return dbg if stage else (tok, qk, vt)


@pytest.mark.parametrize("tile_rows", [32, 64])
def test_st_head_vs_reference(ops, tile_rows):
    """GroupNorm (statistics from per-row-block column partials) -> proj_in -> LayerNorm -> q | k | V^T against the fp16-storage
    restatement stage by stage (1e-3) and the all-fp32 restatement (3e-3); B = 2 so that the per-sample statistics and the V^T
    addressing are exercised, 128-row statistic blocks as a HALO conv producer emits them."""
    B, tokens, C = 2, 1024, 320
    w, x = make_head_case(5, B, tokens, C)
    ref16, ref32 = head_ref(w, x, B, tokens, C, True), head_ref(w, x, B, tokens, C, False)
    for stage, name in ((1, "groupnorm"), (2, "tok"), (3, "ln1")):
        got = run_head(ops, w, x, B, tokens, C, tile_rows, 128, stage)
        check(f"st_head_r{tile_rows}_{name}_vs_fp16ref", got, ref16[stage], rel_l2=1e-3, max_rel=6e-3)
    tok, qk, vt = run_head(ops, w, x, B, tokens, C, tile_rows, 128)
    check(f"st_head_r{tile_rows}_tok", tok, ref16[2], rel_l2=1e-3, max_rel=6e-3)
    check(f"st_head_r{tile_rows}_q", qk[:, :C], ref16["q"], rel_l2=1e-3, max_rel=6e-3)
    check(f"st_head_r{tile_rows}_k", qk[:, C:], ref16["k"], rel_l2=1e-3, max_rel=6e-3)
    vref = ref16["v"].reshape(B, tokens, C).transpose(0, 2, 1)
    check(f"st_head_r{tile_rows}_vt", vt, vref, rel_l2=1e-3, max_rel=6e-3)
    check(f"st_head_r{tile_rows}_q_vs_fp32ref", qk[:, :C], ref32["q"], rel_l2=3e-3)
    check(f"st_head_r{tile_rows}_vt_vs_fp32ref", vt, ref32["v"].reshape(B, tokens, C).transpose(0, 2, 1), rel_l2=3e-3)


def test_st_head_statistic_block_sizes(ops):
    """The fold must not depend on how the producer cut the rows: 32-row blocks (a fused tail upstream), 64, and one block
    per sample give the same normalised rows."""
    B, tokens, C = 2, 256, 320
    w, x = make_head_case(9, B, tokens, C)
    ref16 = head_ref(w, x, B, tokens, C, True)
    for rows in (32, 64, 256):
        got = run_head(ops, w, x, B, tokens, C, 32, rows, 1)
        check(f"st_head_stats_rows{rows}", got, ref16[1], rel_l2=1e-3, max_rel=6e-3)


@pytest.mark.parametrize("tile_rows", [32, 64])
def test_warmer_schedules_match_the_compute_waves(ops, tile_rows):
    """The L2 warmer wave of the fused kernels walks a HAND-MIRRORED schedule of the compute waves' block barriers
    (csrc/stchain.hip make_sched / make_head_sched): one barrier more or less on either side deadlocks the product launch.
    debug_stage = MDX_ST_DEBUG_COUNT_BARRIERS (100) runs the whole chain without the warmer and reports how many barriers the
    compute waves executed; it must equal the length of the schedule the warmer would walk."""
    from minddiffusion_amd import _lib
    lib = _lib.load()
    B, tokens, C, heads, ctx_len = 1, 128, 320, 5, 77
    w, x = make_case(5, B, tokens, C, heads, ctx_len, 1024)
    dbg, _ = run_fused(ops, w, x, B, tokens, C, heads, ctx_len, tile_rows, stage=100)
    n_tail = int(dbg.view(torch.int32).reshape(-1)[0])
    assert n_tail == lib.mdx_st_tail_sched_barriers(C, tile_rows), (n_tail, lib.mdx_st_tail_sched_barriers(C, tile_rows))
    wh, xh = make_head_case(6, B, tokens, C)
    dbg = run_head(ops, wh, xh, B, tokens, C, tile_rows, 32, stage=100)
    n_head = int(dbg.view(torch.int32).reshape(-1)[0])
    assert n_head == lib.mdx_st_head_sched_barriers(C, tile_rows), (n_head, lib.mdx_st_head_sched_barriers(C, tile_rows))
