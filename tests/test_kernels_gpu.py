"""GPU parity tests, one kernel at a time, through the C-ABI (ctypes -> libmdx.so) against the fp32 CPU oracle.

Inputs are rounded to fp16 first so both sides see identical operands; the kernels accumulate in fp32
and store fp16, so the stated tolerances are fp16 output rounding (2^-11 relative) plus accumulation-order
noise: rel-L2 <= 1e-3 for single kernels unless noted.
"""
import math

import numpy as np
import pytest
import torch

from _util import check, h16
from oracle import ldm as O

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


def nhwc(x):  # [B,C,H,W] -> [B,HW,C]
    b, c, h, w = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(b, h * w, c))


def from_nhwc(y, b, h, w):  # [B*HW, C] or [B,HW,C] -> [B,C,H,W]
    y = y.reshape(b, h, w, -1)
    return y.transpose(0, 3, 1, 2)


def pack_conv(w):  # [Cout,Cin,kh,kw] numpy -> packed device weight (the only format mdx_gemm_f16 reads)
    from minddiffusion_amd import ops as _ops
    return _ops.pack_conv_weight(torch.from_numpy(np.ascontiguousarray(w)).to(DEV))


def pack_dense(w):  # [N,K] numpy -> packed device weight
    from minddiffusion_amd import ops as _ops
    return _ops.pack_gemm_weight(dev16(w))


# --------------------------------------------------------------------------- hardware layout pin
def test_mfma_32x32x16_layout(ops):
    """Pins the fragment maps every MFMA kernel here assumes (cdna guide section 3):
    A lane l: A[l&31][8*(l>>5)+j]; B lane l: B[8*(l>>5)+j][l&31]; C lane l reg r: C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]."""
    rng = np.random.RandomState(0)
    A = h16(rng.standard_normal((32, 16)))
    B = h16(rng.standard_normal((16, 32)))  # asymmetric on purpose
    af = np.zeros((64, 8), np.float32)
    bf = np.zeros((64, 8), np.float32)
    for l in range(64):
        for j in range(8):
            af[l, j] = A[l & 31, 8 * (l >> 5) + j]
            bf[l, j] = B[8 * (l >> 5) + j, l & 31]
    c = ops.probe_mfma(dev16(af), dev16(bf)).cpu().numpy()
    C = np.zeros((32, 32), np.float32)
    for l in range(64):
        for r in range(16):
            C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = c[l, r]
    check("mfma_32x32x16_layout", C, A @ B, rel_l2=1e-6)


# --------------------------------------------------------------------------- layout boundary
def test_nchw_nhwc_roundtrip(ops):
    x = np.random.RandomState(1).standard_normal((3, 4, 6, 10)).astype(np.float32)
    y = ops.nchw_to_nhwc(dev32(x), 8)
    assert y.shape == (3, 60, 8)
    yn = y.float().cpu().numpy()
    np.testing.assert_array_equal(yn[:, :, 4:], 0)
    np.testing.assert_allclose(yn[:, :, :4], nhwc(h16(x)), atol=0)
    back = ops.nhwc_to_nchw(y, 4, 6, 10).cpu().numpy()
    np.testing.assert_array_equal(back, h16(x))


# --------------------------------------------------------------------------- GroupNorm / LayerNorm
@pytest.mark.parametrize("B,H,W,C1,C2,silu,eps", [
    (2, 8, 8, 64, 0, True, 1e-5),
    (1, 16, 16, 320, 0, True, 1e-5),
    (2, 8, 8, 1280, 1280, True, 1e-5),      # two-source concat, C = 2560 > 256 chunk columns
    (2, 16, 16, 640, 320, True, 1e-5),
    (1, 5, 7, 320, 0, False, 1e-6),          # ragged pixel count, SpatialTransformer norm (no SiLU, eps 1e-6)
    (1, 64, 64, 320, 0, True, 1e-5),
    (2, 8, 8, 1280, 640, True, 1e-5),        # C = 1920: 60 channels per group (15-chunk group period)
    (1, 4, 4, 192, 0, True, 1e-5),           # GLIDE width: 6 channels per group
    (3, 2, 2, 1280, 0, False, 1e-6),         # 4 pixels per sample
])
def test_groupnorm(ops, B, H, W, C1, C2, silu, eps):
    rng = np.random.RandomState(C1 + C2 + H)
    C = C1 + C2
    x = h16(rng.standard_normal((B, C, H, W)) * 1.5 + 0.3)
    g = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    ref = O.group_norm(torch.tensor(x), torch.tensor(g), torch.tensor(b), eps)
    if silu:
        ref = O.silu(ref)
    x1 = dev16(nhwc(x[:, :C1]))
    x2 = dev16(nhwc(x[:, C1:])) if C2 else None
    y = ops.groupnorm(x1, x2, dev32(g), dev32(b), eps, silu)
    got = from_nhwc(y.float().cpu().numpy(), B, H, W)
    check(f"groupnorm_B{B}_{H}x{W}_C{C1}+{C2}_silu{int(silu)}", got, ref, rel_l2=1e-3, max_abs=2e-2)


@pytest.mark.parametrize("rows,C", [(7, 64), (130, 320), (64, 640), (33, 1280)])
def test_layernorm(ops, rows, C):
    rng = np.random.RandomState(rows)
    x = h16(rng.standard_normal((rows, C)) * 2 + 0.5)
    g = rng.standard_normal(C).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    ref = O.layer_norm(torch.tensor(x), torch.tensor(g), torch.tensor(b), 1e-5)
    y = ops.layernorm(dev16(x), dev32(g), dev32(b), 1e-5)
    check(f"layernorm_{rows}x{C}", y, ref, rel_l2=1e-3, max_abs=2e-2)


# --------------------------------------------------------------------------- implicit GEMM: Dense
@pytest.mark.parametrize("M,N,K,bias,res,splitk", [
    (128, 128, 64, False, False, 1),
    (256, 320, 320, True, True, 1),       # BN = 64 path (320 = 5 x 64)
    (200, 136, 192, True, False, 1),      # ragged M and N tails
    (512, 1280, 1280, True, True, 1),
    (128, 1280, 2560, True, True, 4),     # explicit split-K
    (64, 640, 5120, True, False, 0),      # auto split-K (small M, deep K)
    (4096, 320, 1280, True, True, 1),
])
def test_gemm_dense(ops, M, N, K, bias, res, splitk):
    rng = np.random.RandomState(M + N + K)
    a = h16(rng.standard_normal((M, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
    bv = rng.standard_normal(N).astype(np.float32) if bias else None
    r = h16(rng.standard_normal((M, N))) if res else None
    ref = a @ w.T
    if bias:
        ref = ref + bv
    if res:
        ref = ref + r
    out = ops.gemm(dev16(a), pack_dense(w), N, 1, M, 1, K, bias=dev32(bv) if bias else None,
                   residual=dev16(r) if res else None, residual_ld=N if res else 0, splitk=splitk)
    check(f"gemm_dense_M{M}_N{N}_K{K}_b{int(bias)}_r{int(res)}_s{splitk}", out, ref, rel_l2=1e-3)


@pytest.mark.parametrize("M,N,K,tile_n,splitk,epi", [
    (384, 640, 1280, 128, 1, "none"),       # ragged M tail, 128 x 128 tiles
    (512, 320, 640, 64, 1, "none"),         # 128 x 64 tiles
    (256, 1280, 2560, 128, 3, "none"),      # split-K, reduced by the last arriver
    (256, 1280, 5120, 64, 8, "none"),       # split-K through the reduce kernel
    (256, 2560, 320, 128, 1, "geglu"),
])
def test_gemm_launch_forms_are_bit_identical(ops, M, N, K, tile_n, splitk, epi):
    """mdx_gemm_desc.stages: ring depth 2 .. 4 with four waves per block and 10 | 11 = depth 2 | 3 with eight (128-row tiles),
    depth 2 .. 6 on 64-row tiles.  The launch form changes which wave computes an output and how far ahead the DMAs run, never
    the order in which an output's products are added: all must agree bit for bit, and with the fp32 reference to fp16
    accuracy."""
    rng = np.random.RandomState(M + N + K + splitk)
    a = h16(rng.standard_normal((M, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
    bv = rng.standard_normal(N).astype(np.float32)
    if epi == "geglu":
        full = torch.tensor(a).float() @ torch.tensor(w).float().T + torch.tensor(bv)
        ref = full[:, :N // 2] * O.gelu_tanh(full[:, N // 2:])
        half = N // 2
        nt = half // 64
        wp = np.stack([w[:half].reshape(nt, 64, K), w[half:].reshape(nt, 64, K)], 1).reshape(N, K)
        bp = np.stack([bv[:half].reshape(nt, 64), bv[half:].reshape(nt, 64)], 1).reshape(-1)
        kw = dict(epilogue=ops.EPI_GEGLU)
    else:
        ref = torch.tensor(a).float() @ torch.tensor(w).float().T + torch.tensor(bv)
        wp, bp, kw = w, bv, {}
    ad, wd, bd = dev16(a), pack_dense(wp), dev32(np.ascontiguousarray(bp))
    outs = {}
    forms = [(128, st) for st in (2, 3, 4, 10, 11)] + [(64, st) for st in (2, 3, 4, 5, 6)]
    for tm, st in forms:
        outs[(tm, st)] = ops.gemm(ad, wd, N, 1, M, 1, K, bias=bd, splitk=splitk, tile_m=tm, tile_n=tile_n, stages=st, **kw).clone()
    for f in forms[1:]:
        assert torch.equal(outs[forms[0]], outs[f]), f"tile_m, stages = {f} differs from {forms[0]}"
    check(f"gemm_launch_forms_M{M}_N{N}_K{K}_s{splitk}_{epi}", outs[(64, 6)], ref, rel_l2=1e-3)


def test_gemm_splitk_workspace_reuse(ops):
    """ONE split-K workspace serves different problems launched back to back and repeatedly (the planned executor and
    hipGraph replay rely on this), and the slab reduction order is fixed, so results are bit-identical run to run."""
    rng = np.random.RandomState(77)
    probs = []
    for (M, N, K, sk) in [(256, 320, 1280, 5), (512, 128, 2560, 8), (100, 72, 640, 3)]:
        a = h16(rng.standard_normal((M, K)))
        w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
        bv = rng.standard_normal(N).astype(np.float32)
        ref = torch.tensor(a).float() @ torch.tensor(w).float().T + torch.tensor(bv)
        probs.append((dev16(a), pack_dense(w), dev32(bv), M, N, K, sk, ref))
    need = 0
    descs = []
    for (a, w, bv, M, N, K, sk, ref) in probs:
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        d = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, bias=bv, splitk=sk)
        need = max(need, ops.gemm_workspace_bytes(d), sk * M * N * 4)
        descs.append((d, out))
    ws = torch.full((need // 4,), float("nan"), dtype=torch.float32, device=DEV)   # stale garbage must not leak; the arrival
    # counters are library-owned (include/mdx.h), so the WHOLE workspace may hold garbage
    for d, _ in descs:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    first = None
    for rep in range(3):
        for d, _ in descs:
            ops.gemm_run(d)
        torch.cuda.synchronize()
        outs = [o.float().cpu() for _, o in descs]
        if first is None:
            first = outs
        else:
            for o, f in zip(outs, first):
                assert torch.equal(o, f), "split-K result changed between launches (must be deterministic)"
    for o, pr in zip(first, probs):
        check(f"gemm_splitk_reuse_M{pr[3]}", o, pr[7], rel_l2=1e-3)


def test_gemm_splitk_ticket_stress(ops):
    """In-kernel split-K reduce (csrc/gemm.hip splitk_last_block_reduce: write-through partials, one relaxed agent-scope
    ticket per tile, the last arriver sums in split order) under stress: three problems that ALL take the in-kernel form
    (2, 3 and 4 splits; many tiles; row statistics / GEGLU epilogues on top) rotate through ONE workspace whose partial area
    is NaN-poisoned before every round -- head included, the arrival counters are library-owned, and the workspace comes from
    torch's caching allocator, i.e. at an address earlier tests have used -- 400 eager rounds + 300 replays of a two-round hipGraph (2 800 launches of the in-kernel form in all).  A
    visibility bug -- a partial read before its writer's stores landed, a stale line, a ticket seen early -- would surface as a
    NaN from the poisoned area or as a changed bit: every round must be bit-identical to the first, and the first must agree
    with the same problems run through the separate reduce kernel (gemm_splitk_fixup_max = 0; same split order, but its
    epilogue rounds once instead of staging through fp16, so: to 2 fp16 ulp, not to the bit)."""
    rng = np.random.RandomState(99)
    specs = [(512, 1280, 1280, 3, "plain"), (2048, 640, 2560, 2, "stats"), (512, 1280, 5120, 4, "plain"),
             (256, 2560, 640, 2, "geglu")]
    probs = []
    for (M, N, K, sk, kind) in specs:
        a = dev16(h16(rng.standard_normal((M, K))))
        w = pack_dense(h16(rng.standard_normal((N, K)) / math.sqrt(K)))
        bv = dev32(rng.standard_normal(N).astype(np.float32))
        res = dev16(h16(rng.standard_normal((M, N)))) if kind != "geglu" else None
        probs.append((a, w, bv, res, M, N, K, sk, kind))

    def build(ws):
        descs = []
        for (a, w, bv, res, M, N, K, sk, kind) in probs:
            cols = N // 2 if kind == "geglu" else N
            out = torch.empty(M, cols, dtype=torch.float16, device=DEV)
            st = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=DEV) if kind == "stats" else None
            d = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, cols, bias=bv, splitk=sk, residual=res,
                                   residual_ld=N if res is not None else 0, stats_out=st,
                                   epilogue=ops.EPI_GEGLU if kind == "geglu" else ops.EPI_NONE)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            descs.append((d, out, st))
        return descs

    need = 0
    for (a, w, bv, res, M, N, K, sk, kind) in probs:
        need = max(need, 16384 + sk * ((M + 127) // 128 * 128) * ((N + 127) // 128 * 128) * 4)
    # reference: the reduce-kernel form
    ops.set_option("gemm_splitk_fixup_max", 0)
    try:
        ws0 = ops.new_gemm_workspace(need, DEV)
        ref_descs = build(ws0)
        for d, _, _ in ref_descs:
            assert ops.gemm_query(d)[6] == 0
            ops.gemm_run(d)
        torch.cuda.synchronize()
        refs = [(o.clone(), None if st is None else st.clone()) for _, o, st in ref_descs]
    finally:
        ops.set_option("gemm_splitk_fixup_max", 4)
    ws = torch.empty(need // 4 + 1, dtype=torch.float32, device=DEV)      # NOT zeroed: the library owns the arrival counters
    ws.fill_(float("nan"))
    descs = build(ws)
    for d, _, _ in descs:
        q = ops.gemm_query(d)
        assert q[6] == 1 and q[2] >= 2, f"expected the in-kernel reduce, got {q}"

    def one_round():
        for d, _, _ in descs:
            ops.gemm_run(d)

    one_round()
    torch.cuda.synchronize()
    first = [(o.clone(), None if st is None else st.clone()) for _, o, st in descs]
    for (o, st), (ro, rst), pr in zip(first, refs, probs):
        check(f"splitk_ticket_vs_reduce_kernel_M{pr[4]}_N{pr[5]}_K{pr[6]}", o, ro, rel_l2=5e-4, max_rel=4e-3)

    def same(tag):
        for (d, o, st), (fo, fst), pr in zip(descs, first, probs):
            assert torch.equal(o, fo), f"{tag}: output of M={pr[4]} N={pr[5]} K={pr[6]} changed between launches"
            if st is not None:
                assert torch.equal(st, fst), f"{tag}: row statistics changed between launches"
    for rep in range(400):
        ws.fill_(float("nan"))              # stale partials of the previous round must never be read
        one_round()
        if rep % 50 == 49:
            torch.cuda.synchronize()
            same(f"eager round {rep}")
    g = ops.capture_graph([one_round, one_round])      # (holds the cyclic GC off during the capture)
    for rep in range(300):
        g.replay()
        if rep % 100 == 99:
            torch.cuda.synchronize()
            same(f"graph replay {rep}")
    torch.cuda.synchronize()
    same("end")
    # the library never writes the reserved head of the workspace any more
    assert bool(torch.isnan(ws[:4096]).all()), "the reserved workspace head was written"


def test_gemm_release_counters(ops):
    """include/mdx.h mdx_gemm_release_counters: the library-owned arrival counters can be dropped (nothing in flight) and are
    allocated again by the next in-kernel split-K launch -- same bits before and after, on an uninitialised workspace."""
    from minddiffusion_amd import _lib
    rng = np.random.RandomState(17)
    M, N, K = 256, 640, 2560
    a = dev16(h16(rng.standard_normal((M, K))))
    w = pack_dense(h16(rng.standard_normal((N, K)) / math.sqrt(K)))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ws = torch.full((16384 // 4 + 3 * 256 * 640 + 64,), float("nan"), dtype=torch.float32, device=DEV)
    d = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, splitk=3, workspace=ws)
    assert ops.gemm_query(d)[6] == 1
    ops.gemm_run(d)
    torch.cuda.synchronize()
    first = out.clone()
    _lib.check(_lib.load().mdx_gemm_release_counters(), "mdx_gemm_release_counters")
    for _ in range(3):
        out.zero_()
        ops.gemm_run(d)
        torch.cuda.synchronize()
        assert torch.equal(out, first)
    ref = torch.tensor(a.float().cpu().numpy() @ ops.unpack_gemm_weight(w, N, K).float().cpu().numpy().T)
    check("gemm_release_counters_M256_N640_K2560_s3", out.float().cpu(), ref, rel_l2=1e-3)


def test_gemm_release_workspace_reuses_the_counters_under_a_recaptured_graph(ops):
    """include/mdx.h mdx_gemm_release_workspace: a workspace that is dropped hands its arrival counters back, the next workspace
    (another address) takes the SAME set, and a graph captured on it replays bit-stably -- the case of a server that plans at many
    resolutions (round-3 review: every distinct workspace address used to pin 16 KiB for good)."""
    from minddiffusion_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(23)
    M, N, K = 256, 640, 2560
    a = dev16(h16(rng.standard_normal((M, K))))
    w = pack_dense(h16(rng.standard_normal((N, K)) / math.sqrt(K)))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    n_ws = 16384 // 4 + 3 * 256 * 640 + 64
    ws1 = torch.full((n_ws,), float("nan"), dtype=torch.float32, device=DEV)
    d1 = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, splitk=3, workspace=ws1)
    assert ops.gemm_query(d1)[6] == 1
    ops.gemm_run(d1)
    torch.cuda.synchronize()
    first = out.clone()
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws1)) == 1
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws1)) == 0          # already released
    ws2 = torch.full((n_ws + 1024,), float("nan"), dtype=torch.float32, device=DEV)      # a different buffer
    assert ws2.data_ptr() != ws1.data_ptr()
    d2 = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, splitk=3, workspace=ws2)
    ops.gemm_run(d2)                                    # first use outside the capture: takes the released set
    torch.cuda.synchronize()
    g = ops.capture_graph([lambda: ops.gemm_run(d2), lambda: ops.gemm_run(d2)])
    for _ in range(20):
        out.zero_()
        ws2.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first)
    del g
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws2)) == 1


def _lean_cases():
    """(name, M, N, K, kwargs of make_gemm_desc incl. tile hints, feature) for the lean dense kernel of csrc/dense.hip."""
    cases = []
    for (tm, tn) in [(64, 64), (64, 128), (128, 64), (128, 128)]:
        for st in ([2, 3, 4, 5, 6] if tm == 64 else ([2, 3, 4] if tn == 64 else [2, 3])):
            cases.append((f"t{tm}x{tn}_s{st}_bias_res", 512, 640, 1280, dict(tile_m=tm, tile_n=tn, stages=st, splitk=1), "res"))
    cases += [
        ("plain_small", 128, 1280, 1280, dict(tile_m=64, tile_n=64, splitk=1), "bias"),
        ("nobias", 512, 1280, 640, dict(tile_m=64, tile_n=64, splitk=1), "none"),
        ("m_tail", 520, 1280, 1280, dict(tile_m=64, tile_n=64, splitk=1), "res"),                 # rows past M in the last tile
        ("n_tail_128", 512, 320, 1280, dict(tile_m=64, tile_n=128, splitk=1), "res"),             # a half-empty 128-column tile
        ("big_grid_no_pre", 8192, 640, 640, dict(tile_m=64, tile_n=64, splitk=1), "res"),         # > 512 blocks: the light form
        ("big_grid_128", 8192, 1280, 640, dict(tile_m=128, tile_n=128, splitk=1), "res"),
        ("stats", 512, 1280, 1280, dict(tile_m=64, tile_n=64, splitk=1), "stats"),
        ("colstats", 512, 640, 1280, dict(tile_m=64, tile_n=64, splitk=1), "colstats"),
        ("ln", 512, 1280, 1280, dict(tile_m=64, tile_n=64, splitk=1), "ln"),                     # 20 partials per row: held in registers
        ("ln_128x64", 512, 1280, 640, dict(tile_m=128, tile_n=64, splitk=1), "ln"),
        ("ln_long", 256, 640, 2560, dict(tile_m=64, tile_n=64, splitk=1), "ln"),                 # 40 partials: the epilogue folds
        ("ln_big_grid", 8192, 960, 320, dict(tile_m=64, tile_n=64, splitk=1), "ln"),
        ("geglu_ln", 512, 2560, 640, dict(tile_m=64, splitk=1), "geglu_ln"),
        ("geglu", 2048, 5120, 640, dict(tile_m=64, splitk=1), "geglu"),
        ("qkv_split", 512, 3 * 640, 640, dict(tile_m=64, tile_n=128, splitk=1), "nsplit_ln"),
        ("splitk3", 256, 640, 2560, dict(tile_m=64, tile_n=64, splitk=3), "res"),
        ("splitk4_ln", 128, 1280, 1280, dict(tile_m=64, tile_n=64, splitk=4), "ln"),
        ("splitk2_geglu", 512, 2560, 1280, dict(tile_m=64, splitk=2), "geglu"),
        ("auto_tiles", 2048, 640, 2560, dict(), "res"),                                 # the tile table's own choice
        ("auto_tiles_spread", 64, 1280, 1280, dict(), "res"),                           # one row of M tiles: the 2-D SPREAD grid
    ]
    return cases


@pytest.mark.parametrize("level", [1, 2, 3, 4])      # option gemm_lean_dense: prefetch levels 1 (product), 0, 2, 3 (csrc/dense.hip)
@pytest.mark.parametrize("case", _lean_cases(), ids=lambda c: c[0])
def test_lean_dense_kernel_is_bit_identical_to_the_generic_kernel(ops, case, level):
    """csrc/dense.hip (round 6): dense row-major launches run a lean instantiation of the tile program -- division-free prologue,
    the epilogue's global reads (bias, residual rows, LayerNorm partials, S[n]) requested before the K loop.  Same LDS image, same K
    order, same epilogue arithmetic: every output (and every statistic it emits) must carry the generic kernel's BITS, for every
    tile / ring instantiation, both prefetch forms (small / large grids), every epilogue feature, tails, and the in-kernel split-K
    reduce; the result is also checked against numpy so that "identical" does not mean "identically wrong"."""
    name, M, N, K, kw, feat = case
    r = np.random.RandomState(sum(map(ord, name)))
    a = h16(r.standard_normal((M, K)))
    geglu = feat.startswith("geglu")
    w = h16(r.standard_normal((N, K)) / math.sqrt(K))
    bv = r.standard_normal(N).astype(np.float32)
    No = N // 2 if geglu else N
    res = h16(r.standard_normal((M, No)))
    da = dev16(a)
    args = dict(kw)
    extra = {}
    if feat in ("res",):
        args.update(bias=dev32(bv), residual=dev16(res), residual_ld=No)
    elif feat == "bias":
        args.update(bias=dev32(bv))
    elif feat == "stats":
        extra["stats"] = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=DEV)
        args.update(bias=dev32(bv), residual=dev16(res), residual_ld=No, stats_out=extra["stats"])
    elif feat == "colstats":
        extra["cs"] = torch.zeros((M // 64, N, 2), dtype=torch.float32, device=DEV)
        args.update(bias=dev32(bv), colstats_out=extra["cs"])
    elif geglu:
        args.update(bias=dev32(bv), epilogue=ops.EPI_GEGLU)
    ln = feat.endswith("ln")
    wp = None
    if ln:
        # producer statistics of the rows (what the GEMM in front would have written) + folded weights
        xs = a.astype(np.float32).reshape(M, K // 64, 64)
        st = np.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).astype(np.float32)
        g = (1 + 0.3 * r.standard_normal(K)).astype(np.float32)
        be = (0.3 * r.standard_normal(K)).astype(np.float32)
        wg, s_, cb = ops.fold_layernorm(dev16(w), dev32(g), dev32(be), None if feat == "nsplit_ln" else dev32(bv))
        args.update(ln_stats=dev32(st), ln_s=s_)
        if feat == "nsplit_ln":
            args.pop("bias", None)
            cb_np = cb.cpu().numpy()
        else:
            args["bias"] = cb
        if not geglu:
            wp = ops.pack_gemm_weight(wg)
    if geglu:
        # GEGLU packing: every 128-wide tile = 64 'a' columns | 64 gate columns (include/mdx.h)
        src = wg.float().cpu().numpy() if ln else w.astype(np.float32)
        half = N // 2
        wi = np.empty_like(src)
        bsrc = args["bias"].cpu().numpy()
        bi = np.empty_like(bsrc)
        for t in range(N // 128):
            wi[t * 128:t * 128 + 64] = src[t * 64:t * 64 + 64]
            wi[t * 128 + 64:t * 128 + 128] = src[half + t * 64:half + t * 64 + 64]
            bi[t * 128:t * 128 + 64] = bsrc[t * 64:t * 64 + 64]
            bi[t * 128 + 64:t * 128 + 128] = bsrc[half + t * 64:half + t * 64 + 64]
        wp = ops.pack_gemm_weight(dev16(h16(wi)))
        args["bias"] = dev32(bi)
        if ln:
            si = np.empty(N, np.float32)
            sn = s_.cpu().numpy()
            for t in range(N // 128):
                si[t * 128:t * 128 + 64] = sn[t * 64:t * 64 + 64]
                si[t * 128 + 64:t * 128 + 128] = sn[half + t * 64:half + t * 64 + 64]
            args["ln_s"] = dev32(si)
    elif not ln:
        wp = pack_dense(w)
    vt = None
    if feat == "nsplit_ln":
        C = N // 3
        vt = torch.zeros((1, C, M), dtype=torch.float16, device=DEV)
        args.update(out2=vt, out2_ld=M, n_split=2 * C, bias=cb)      # (the fold's W beta term is the projection's bias)

    def run(lean):
        ops.set_option("gemm_lean_dense", lean)
        out = torch.zeros((M, (2 * N // 3) if feat == "nsplit_ln" else No), dtype=torch.float16, device=DEV)
        if vt is not None:
            vt.zero_()
        for t_ in extra.values():
            t_.zero_()
        d = ops.make_gemm_desc(da, wp, N, 1, M, 1, K, out, out.shape[1], **args)
        need = ops.gemm_workspace_bytes(d)
        ws = ops.new_gemm_workspace(max(need, 16), DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        form = ops.gemm_query(d)[3]
        ops.gemm_run(d)
        torch.cuda.synchronize()
        return form, [out.clone()] + ([vt.clone()] if vt is not None else []) + [t_.clone() for t_ in extra.values()]

    keep = ops.get_option("gemm_lean_dense")
    try:
        f1, lean = run(level)
        f0, gen = run(0)
    finally:
        ops.set_option("gemm_lean_dense", keep)
    assert f1 == 2 and f0 == 0, f"{name}: launch forms {f1} / {f0} (2 = lean dense kernel, 0 = generic)"
    for i, (x, y) in enumerate(zip(lean, gen)):
        assert torch.equal(x, y), (f"{name}: output {i} differs between the lean and the generic kernel: "
                                   f"max |d| = {float((x.float() - y.float()).abs().max())}")
    # and against numpy (fp16 inputs, fp32 accumulate)
    af, wf = a.astype(np.float32), w.astype(np.float32)
    if ln:
        mu = af.mean(1, keepdims=True)
        var = af.var(1, keepdims=True)
        af = (af - mu) / np.sqrt(var + 1e-5) * g + be
    y = af @ wf.T + (0 if feat in ("none", "nsplit_ln") else bv)
    if geglu:
        half = N // 2
        gate = y[:, half:]
        y = y[:, :half] * (0.5 * gate * (1 + np.tanh(0.7978845608028654 * (gate + 0.044715 * gate ** 3))))
    if feat in ("res", "stats"):
        y = y + res.astype(np.float32)
    if feat == "nsplit_ln":
        C = N // 3
        check(f"lean_dense_{name}_qk", lean[0].float().cpu(), torch.tensor(y[:, :2 * C]), rel_l2=2e-3)
        check(f"lean_dense_{name}_vt", lean[1][0].float().cpu(), torch.tensor(y[:, 2 * C:].T.copy()), rel_l2=2e-3)
    else:
        check(f"lean_dense_{name}", lean[0].float().cpu(), torch.tensor(y), rel_l2=2e-3 if ln else 1e-3)


def test_gemm_caller_owned_arrival_counters(ops):
    """include/mdx.h mdx_gemm_bind_counters: a caller that owns every byte binds MDX_GEMM_WS_HEAD zeroed bytes to its workspace; the
    in-kernel split-K reduce takes its tickets there and leaves them zero (a poisoned counter would prove the use, but it can reach
    the kernel's corrupted-counter trap, which takes the process down), results equal the library-owned form bit for bit, release
    only unbinds."""
    from minddiffusion_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(29)
    M, N, K = 256, 640, 2560
    a = dev16(h16(rng.standard_normal((M, K))))
    w = pack_dense(h16(rng.standard_normal((N, K)) / math.sqrt(K)))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    n_ws = 16384 // 4 + 3 * 256 * 640 + 64
    ws_lib = torch.full((n_ws,), float("nan"), dtype=torch.float32, device=DEV)          # library-owned counters
    d0 = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, splitk=3, workspace=ws_lib)
    assert ops.gemm_query(d0)[6] == 1
    ops.gemm_run(d0)
    torch.cuda.synchronize()
    ref_bits = out.clone()
    ws = torch.full((n_ws,), float("nan"), dtype=torch.float32, device=DEV)
    counters = torch.zeros(16384 // 4, dtype=torch.int32, device=DEV)
    _lib.check(lib.mdx_gemm_bind_counters(ctypes_ptr(ws), ctypes_ptr(counters)), "bind")
    _lib.check(lib.mdx_gemm_bind_counters(ctypes_ptr(ws), ctypes_ptr(counters)), "bind again (no-op)")
    d1 = ops.make_gemm_desc(a, w, N, M, 1, 1, K, out, N, splitk=3, workspace=ws)
    for _ in range(3):
        out.zero_()
        ops.gemm_run(d1)
        torch.cuda.synchronize()
        assert torch.equal(out, ref_bits)
        assert int(counters.abs().sum()) == 0            # every launch leaves them zero
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws)) == 1
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws)) == 0
    assert lib.mdx_gemm_release_workspace(ctypes_ptr(ws_lib)) == 1


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def test_gemm_two_source_1x1(ops):
    """ResBlock skip_connection on the (virtual) concat of h and the UNet skip tensor (openaimodel.py:174,568)."""
    rng = np.random.RandomState(5)
    B, H, W, C1, C2, N = 2, 8, 8, 128, 64, 192
    x = h16(rng.standard_normal((B, C1 + C2, H, W)))
    w = h16(rng.standard_normal((N, C1 + C2, 1, 1)) / 14)
    bv = rng.standard_normal(N).astype(np.float32)
    ref = O.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bv), padding=0)
    out = ops.gemm(dev16(nhwc(x[:, :C1])), pack_conv(w), N, B, H, W, C1, a2=dev16(nhwc(x[:, C1:])), c2=C2,
                   bias=dev32(bv))
    check("gemm_two_source_1x1", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)


# --------------------------------------------------------------------------- implicit GEMM: conv3x3
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up,splitk", [
    (2, 8, 8, 64, 64, 1, 0, 1),
    (1, 16, 16, 64, 128, 1, 0, 1),
    (2, 16, 16, 128, 128, 2, 0, 1),     # Downsample (openaimodel.py:81)
    (2, 8, 8, 128, 128, 1, 1, 1),       # Upsample: nearest-2x folded into the gather (openaimodel.py:57)
    (1, 6, 10, 64, 72, 1, 0, 1),        # ragged spatial extent / N tail
    (2, 8, 8, 320, 320, 1, 0, 0),       # auto split-K
    (1, 32, 32, 320, 640, 1, 0, 1),
    (1, 8, 8, 8, 64, 1, 0, 1),          # conv_in: Cin padded 4 -> 8, generic (non 64-aligned) K path
    (1, 8, 8, 64, 8, 1, 0, 1),          # conv_out: Cout padded 4 -> 8
    # 8x16-patch HALO kernel (H % 8 == 0, W % 16 == 0, Cin % 64 == 0, stride 1):
    (2, 16, 32, 192, 320, 1, 0, 1),     # 64-wide N tiles, several chunks, patches across two images
    (2, 16, 16, 320, 320, 1, 0, 0),     # auto split-K (chunk-aligned)
    (1, 16, 16, 640, 128, 1, 0, 3),     # 10 chunks over 3 splits (4 + 4 + 2)
    (3, 8, 16, 64, 72, 1, 0, 1),        # one patch per image, N tail
    (1, 24, 48, 128, 8, 1, 0, 1),       # conv_out shape, non-power-of-two patch grid
    (1, 64, 64, 64, 64, 1, 0, 1),       # 32 patches: image borders on all sides + interior
])
def test_gemm_conv3x3(ops, B, H, W, Cin, Cout, stride, up, splitk):
    rng = np.random.RandomState(Cin + Cout + H + stride + up)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    xin = torch.tensor(x)
    if up:
        xin = O.upsample_nearest2x(xin)
    ref = O.conv2d(xin, torch.tensor(w), torch.tensor(bv), stride=stride, padding=1)
    out = ops.gemm(dev16(nhwc(x)), pack_conv(w), Cout, B, H, W, Cin, bias=dev32(bv), ksize=3, stride=stride,
                   upsample=up, splitk=splitk)
    Ho, Wo = ref.shape[2], ref.shape[3]
    check(f"conv3x3_B{B}_{H}x{W}_{Cin}to{Cout}_s{stride}_u{up}_k{splitk}",
          from_nhwc(out.float().cpu().numpy(), B, Ho, Wo), ref, rel_l2=1e-3)


def test_gemm_conv_rowbias_residual(ops):
    """conv1 of a ResBlock: + bias + per-sample time-embedding row (openaimodel.py:188-200); conv2: + skip."""
    rng = np.random.RandomState(9)
    B, H, W, C = 3, 8, 8, 64
    x = h16(rng.standard_normal((B, C, H, W)))
    w = h16(rng.standard_normal((C, C, 3, 3)) / 24)
    bv = rng.standard_normal(C).astype(np.float32)
    emb = rng.standard_normal((B, 200)).astype(np.float32)  # emb_all with this block at columns 72..136
    res = h16(rng.standard_normal((B, C, H, W)))
    ref = O.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bv)) + torch.tensor(emb[:, 72:136])[:, :, None, None] \
        + torch.tensor(res)
    embd = dev32(emb)
    out = ops.gemm(dev16(nhwc(x)), pack_conv(w), C, B, H, W, C, bias=dev32(bv), ksize=3,
                   rowbias=embd[:, 72:136], rowbias_ld=200, residual=dev16(nhwc(res)), residual_ld=C)
    check("conv3x3_rowbias_residual", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)


@pytest.mark.parametrize("splitk", [1, 2])
def test_gemm_conv_halo_rowbias_residual(ops, splitk):
    """Same fused epilogue through the 8x16-patch HALO kernel (row -> pixel map differs from the linear one)."""
    rng = np.random.RandomState(19)
    B, H, W, C, N = 3, 16, 32, 128, 192
    x = h16(rng.standard_normal((B, C, H, W)))
    w = h16(rng.standard_normal((N, C, 3, 3)) / 34)
    bv = rng.standard_normal(N).astype(np.float32)
    emb = rng.standard_normal((B, 400)).astype(np.float32)
    res = h16(rng.standard_normal((B, N, H, W)))
    ref = O.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bv)) + torch.tensor(emb[:, 8:200])[:, :, None, None] \
        + torch.tensor(res)
    embd = dev32(emb)
    out = ops.gemm(dev16(nhwc(x)), pack_conv(w), N, B, H, W, C, bias=dev32(bv), ksize=3, splitk=splitk,
                   rowbias=embd[:, 8:200], rowbias_ld=400, residual=dev16(nhwc(res)), residual_ld=N)
    check(f"conv3x3_halo_rowbias_residual_k{splitk}", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)


@pytest.mark.parametrize("B,Cin,Cout,splitk", [(2, 64, 64, 1), (3, 128, 192, 1), (5, 320, 72, 0), (2, 1280, 1280, 0),
                                                (4, 640, 320, 5), (1, 192, 128, 3)])
def test_gemm_conv_halo8(ops, B, Cin, Cout, splitk):
    """The two-samples-per-tile HALO variant for 8 x 8 images (the deepest level of a 64 x 64-latent UNet): odd batches leave
    the last tile half empty, borders on all four sides of every sample, chunk-aligned split-K, N tails, and the fused
    time-embedding row + residual epilogue; the query must report the HALO kernel."""
    from minddiffusion_amd import ops as _ops
    rng = np.random.RandomState(B * 1000 + Cin + Cout)
    H = W = 8
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    emb = rng.standard_normal((B, Cout + 24)).astype(np.float32)
    res = h16(rng.standard_normal((B, Cout, H, W)))
    ref = O.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bv)) + torch.tensor(emb[:, 16:16 + Cout])[:, :, None, None] \
        + torch.tensor(res)
    embd = dev32(emb)
    a, wp = dev16(nhwc(x)), pack_conv(w)
    out = ops.gemm(a, wp, Cout, B, H, W, Cin, bias=dev32(bv), ksize=3, splitk=splitk, rowbias=embd[:, 16:16 + Cout],
                   rowbias_ld=Cout + 24, residual=dev16(nhwc(res)), residual_ld=Cout)
    check(f"conv3x3_halo8_B{B}_{Cin}to{Cout}_k{splitk}", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)
    d = _ops.make_gemm_desc(a, wp, Cout, B, H, W, Cin, torch.empty((B * 64, Cout), dtype=torch.float16, device=DEV), Cout,
                            ksize=3, splitk=1)
    assert _ops.gemm_query(d)[3] == 1, "8 x 8 stride-1 convs with Cin % 64 == 0 must take the HALO kernel"


@pytest.mark.parametrize("M,C,splitk", [(256, 64, 1), (100, 320, 1), (64, 320, 2)])
def test_gemm_geglu(ops, M, C, splitk):
    """GEGLU (attention.py:41-51): x, gate = split(proj(x)); x * gelu_tanh(gate), fused in the epilogue."""
    rng = np.random.RandomState(M + C)
    a = h16(rng.standard_normal((M, C)))
    w = h16(rng.standard_normal((8 * C, C)) / math.sqrt(C))
    bv = rng.standard_normal(8 * C).astype(np.float32)
    y = torch.tensor(a) @ torch.tensor(w).T + torch.tensor(bv)
    xa, gate = y.chunk(2, dim=-1)
    ref = xa * O.gelu_tanh(gate)
    half = 4 * C
    nt = half // 64
    wp = np.stack([w[:half].reshape(nt, 64, C), w[half:].reshape(nt, 64, C)], 1).reshape(8 * C, C)
    bp = np.stack([bv[:half].reshape(nt, 64), bv[half:].reshape(nt, 64)], 1).reshape(-1)
    out = ops.gemm(dev16(a), pack_dense(wp), 8 * C, 1, M, 1, C, bias=dev32(bp), epilogue=ops.EPI_GEGLU, splitk=splitk)
    assert out.shape == (M, 4 * C)
    check(f"gemm_geglu_M{M}_C{C}_s{splitk}", out, ref, rel_l2=2e-3)


@pytest.mark.parametrize("B,T,K,N,pad,splitk", [(2, 64, 128, 128, 0, 1), (2, 256, 320, 320, 0, 1), (2, 80, 64, 128, 0, 1),
                                              (1, 64, 640, 64, 0, 2)])
def test_gemm_transposed_store(ops, B, T, K, N, pad, splitk):
    """V^T store for the attention kernel: out[b][n][tok]."""
    rng = np.random.RandomState(T + N)
    a = h16(rng.standard_normal((B * T, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
    ref = (a @ w.T).reshape(B, T, N).transpose(0, 2, 1)
    out = ops.gemm(dev16(a), pack_dense(w), N, B, T, 1, K, out_mode=ops.OUT_TRANSPOSED, splitk=splitk)
    assert out.shape == (B, N, T)
    check(f"gemm_transposed_B{B}_T{T}_K{K}_N{N}_s{splitk}", out, ref, rel_l2=1e-3)


@pytest.mark.parametrize("B,T,C,splitk", [(2, 64, 64, 1), (2, 256, 320, 1), (1, 1024, 640, 1), (2, 64, 320, 3), (2, 128, 128, 2)])
def test_gemm_split_rowmajor_transposed_output(ops, B, T, C, splitk):
    """mdx_gemm_desc.n_split: [q | k] columns row-major and the v columns transposed (V^T for the attention kernel) from
    ONE launch over the concatenated [q; k; v] weight (attention.py:108-112), direct and split-K paths."""
    rng = np.random.RandomState(T + C + splitk)
    a = h16(rng.standard_normal((B * T, C)))
    w = h16(rng.standard_normal((3 * C, C)) / math.sqrt(C))
    bv = rng.standard_normal(3 * C).astype(np.float32)
    full = torch.tensor(a).float() @ torch.tensor(w).float().T + torch.tensor(bv)
    qk = torch.empty((B, T, 2 * C), dtype=torch.float16, device=DEV)
    vt = torch.zeros((B, C, T), dtype=torch.float16, device=DEV)
    d = ops.make_gemm_desc(dev16(a), pack_dense(w), 3 * C, B, T, 1, C, qk, 2 * C, bias=dev32(bv), splitk=splitk,
                           out2=vt, out2_ld=T, n_split=2 * C)
    keep = d.a, d.w, d.bias
    need = ops.gemm_workspace_bytes(d)
    ws = ops.new_gemm_workspace(need, DEV)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    # keep the operand tensors alive across the launch
    a_d, w_d, b_d = dev16(a), pack_dense(w), dev32(bv)
    d.a, d.w, d.bias = a_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr()
    ops.gemm_run(d)
    torch.cuda.synchronize()
    check(f"gemm_split_qk_B{B}_T{T}_C{C}_s{splitk}", qk.reshape(B * T, 2 * C), full[:, : 2 * C], rel_l2=1e-3)
    check(f"gemm_split_vt_B{B}_T{T}_C{C}_s{splitk}", vt, full[:, 2 * C:].reshape(B, T, C).permute(0, 2, 1), rel_l2=1e-3)


@pytest.mark.parametrize("M,C,kind,sp,sc", [
    (256, 320, "plain", 1, 1), (200, 64, "plain", 1, 1), (512, 640, "geglu", 1, 1), (256, 320, "qkv", 1, 1),
    (128, 1280, "plain", 4, 4), (128, 640, "geglu", 2, 2), (128, 640, "qkv", 1, 3), (4096, 320, "geglu", 1, 1),
    (64, 1280, "plain", 0, 0),
])
def test_gemm_layernorm_fold(ops, M, C, kind, sp, sc):
    """nn.LayerNorm folded into the GEMMs around it (mdx_gemm_desc.stats_out / ln_stats; BasicTransformerBlock
    attention.py:176-185): the producer GEMM (bias + residual) emits per-row {sum, sumsq}, the consumer multiplies the RAW
    rows by gamma (.) W and corrects the accumulators.  Checked against explicit LN on the producer's fp16 output, for the
    plain / GEGLU / merged q|k|v^T consumers, direct and split-K on either side.  The un-normalised rows carry a mean of
    about the same size as their spread, so the cancellation is exercised."""
    rng = np.random.RandomState(M + C + sp)
    K0 = 256
    a0 = h16(rng.standard_normal((M, K0)))
    w0 = h16(rng.standard_normal((C, K0)) / math.sqrt(K0))
    b0 = (rng.standard_normal(C) + 1.0).astype(np.float32)
    r0 = h16(rng.standard_normal((M, C)) * 2)
    g = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    be = (0.3 * rng.standard_normal(C)).astype(np.float32)
    nout = {"plain": C, "geglu": 8 * C, "qkv": 3 * C}[kind]
    w1 = h16(rng.standard_normal((nout, C)) / math.sqrt(C))
    b1 = rng.standard_normal(nout).astype(np.float32)

    def run(desc, keep):
        need = ops.gemm_workspace_bytes(desc)
        ws = ops.new_gemm_workspace(need, DEV)
        desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        ops.gemm_run(desc)
        torch.cuda.synchronize()

    # producer
    x = torch.empty((M, C), dtype=torch.float16, device=DEV)
    stats = torch.full((M, C // 64, 2), float("nan"), dtype=torch.float32, device=DEV)
    pa, pw, pb, pr = dev16(a0), pack_dense(w0), dev32(b0), dev16(r0)
    run(ops.make_gemm_desc(pa, pw, C, 1, M, 1, K0, x, C, bias=pb, residual=pr, residual_ld=C, splitk=sp, stats_out=stats), None)
    xr = x.float().cpu()
    check(f"lnfold_producer_M{M}_C{C}_s{sp}", x, torch.tensor(a0) @ torch.tensor(w0).T + torch.tensor(b0) + torch.tensor(r0),
          rel_l2=1e-3)
    st_ref = torch.stack([xr.reshape(M, C // 64, 64).sum(-1), (xr ** 2).reshape(M, C // 64, 64).sum(-1)], -1)
    check(f"lnfold_stats_M{M}_C{C}_s{sp}", stats, st_ref, rel_l2=1e-5)

    # consumer
    ln = O.layer_norm(xr, torch.tensor(g), torch.tensor(be), 1e-5)
    full = ln @ torch.tensor(w1).T + torch.tensor(b1)
    if kind == "geglu":
        half = 4 * C
        nt = half // 64
        w1p = np.stack([w1[:half].reshape(nt, 64, C), w1[half:].reshape(nt, 64, C)], 1).reshape(8 * C, C)
        b1p = np.stack([b1[:half].reshape(nt, 64), b1[half:].reshape(nt, 64)], 1).reshape(-1)
    else:
        w1p, b1p = w1, b1
    wg, s, cb = ops.fold_layernorm(dev16(w1p), dev32(g), dev32(be), dev32(b1p))
    wgp = ops.pack_gemm_weight(wg)
    if kind == "plain":
        out = torch.empty((M, C), dtype=torch.float16, device=DEV)
        run(ops.make_gemm_desc(x, wgp, C, 1, M, 1, C, out, C, bias=cb, splitk=sc, ln_stats=stats, ln_s=s), None)
        check(f"lnfold_plain_M{M}_C{C}_s{sc}", out, full, rel_l2=2e-3)
    elif kind == "geglu":
        out = torch.empty((M, 4 * C), dtype=torch.float16, device=DEV)
        run(ops.make_gemm_desc(x, wgp, 8 * C, 1, M, 1, C, out, 4 * C, bias=cb, splitk=sc, epilogue=ops.EPI_GEGLU,
                               ln_stats=stats, ln_s=s), None)
        xa, gate = full.chunk(2, dim=-1)
        check(f"lnfold_geglu_M{M}_C{C}_s{sc}", out, xa * O.gelu_tanh(gate), rel_l2=3e-3)
    else:
        B, T = 2, M // 2
        qk = torch.empty((B, T, 2 * C), dtype=torch.float16, device=DEV)
        vt = torch.zeros((B, C, T), dtype=torch.float16, device=DEV)
        run(ops.make_gemm_desc(x, wgp, 3 * C, B, T, 1, C, qk, 2 * C, bias=cb, splitk=sc, out2=vt, out2_ld=T, n_split=2 * C,
                               ln_stats=stats, ln_s=s), None)
        check(f"lnfold_qk_M{M}_C{C}_s{sc}", qk.reshape(M, 2 * C), full[:, : 2 * C], rel_l2=2e-3)
        check(f"lnfold_vt_M{M}_C{C}_s{sc}", vt, full[:, 2 * C:].reshape(B, T, C).permute(0, 2, 1), rel_l2=2e-3)


# --------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, heads):
    b, n, c = q.shape
    d = c // heads
    qt, kt, vt = [torch.tensor(t).reshape(b, -1, heads, d).permute(0, 2, 1, 3) for t in (q, k, v)]
    s = torch.matmul(qt, kt.transpose(2, 3)) * d ** -0.5
    o = torch.matmul(torch.softmax(s, -1), vt)
    return o.permute(0, 2, 1, 3).reshape(b, n, c)


@pytest.mark.parametrize("B,heads,Nq,Nk,spike,D", [
    (1, 1, 64, 64, False, 64),
    (2, 2, 256, 256, False, 64),
    (1, 5, 1024, 1024, False, 64),
    (2, 2, 100, 77, False, 64),      # cross-attention: 77 text tokens, ragged queries
    (1, 1, 128, 200, False, 64),     # key tail inside a 64-key tile
    (1, 2, 256, 320, True, 64),      # a late outlier key forces the online-softmax rescale branch (guide rule 26)
    (2, 8, 256, 256, False, 40),     # Wukong-Huahua: num_heads=8 => d = 320/8
    (1, 8, 100, 77, False, 40),
    (2, 8, 256, 256, True, 80),      # d = 640/8
    (1, 8, 128, 77, False, 80),
    (1, 8, 256, 256, True, 160),     # d = 1280/8
    (2, 8, 64, 77, False, 160),
])
def test_attention(ops, B, heads, Nq, Nk, spike, D):
    C = heads * D
    rng = np.random.RandomState(Nq + Nk + heads + D)
    q = h16(rng.standard_normal((B, Nq, C)))
    k = h16(rng.standard_normal((B, Nk, C)))
    v = h16(rng.standard_normal((B, Nk, C)))
    if spike:
        k[:, Nk - 30] = h16(q[:, 3] * 3.0)   # a key in the last tile dominates query row 3
        k[:, 10] = h16(-q[:, 7] * 2.0)
    ref = _attn_ref(q, k, v, heads)
    ld = (Nk + 7) // 8 * 8
    vt = np.zeros((B, C, ld), np.float32)
    vt[:, :, :Nk] = v.transpose(0, 2, 1)
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    out = torch.empty((B, Nq, C), dtype=torch.float16, device=DEV)
    ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                  Nq * C, C, Nk * C, C, C * ld, ld, Nq * C, C)
    # P is rounded to fp16 before PV and O is stored fp16: 2e-3 relative
    check(f"attention_B{B}_h{heads}_q{Nq}_k{Nk}_d{D}_spike{int(spike)}", out, ref, rel_l2=2e-3, max_abs=2e-2)


RAMPS = {
    # per-64-key-tile offsets of every query's scores, in the log2 units the kernel's exponentials are taken in
    "stale3": lambda t: 3.0 * t,                       # reference maximum stale by 3, 6 -> rescale at 9 (> 2^8), ...
    "stale6.5": lambda t: 6.5 * t,                     # stale by 6.5, then 13 -> rescale every second tile
    "edge7.9": lambda t: 7.9 * t,                      # just under the threshold: P up to 2^7.9 = 239 in fp16
    "always12": lambda t: 12.0 * t,                    # every tile outgrows the reference by more than 2^8
    "sawtooth": lambda t: [0.0, 7.0, 14.0, -6.0, 1.9, 9.8, 17.0, 40.0][t % 8],   # down-steps + one huge late tile
}


@pytest.mark.parametrize("ramp", sorted(RAMPS))
@pytest.mark.parametrize("D,heads", [(40, 2), (64, 2), (80, 1), (160, 1)])
@pytest.mark.parametrize("mode", ["self", "cross", "causal"])
def test_attention_stale_maximum_regime(ops, ramp, D, heads, mode):
    """The lazily moved softmax reference (csrc/attention.hip: the running maximum is only replaced when a query of the wave
    outgrows it by more than 2^8): key tiles whose scores climb by 3 / 6.5 / 7.9 / 12 log2 units per tile, and a sawtooth with
    down-steps and one +40 tile, so that P fragments between 1 and 2^8 are rounded to fp16 under a stale reference, the
    ballot-uniform rescale branch is taken on some tiles and skipped on others, for all four head dims, ragged key counts
    (cross) and the masked causal variant.  Reference: exact float64 softmax."""
    B = 1
    Nq, Nk = {"self": (256, 512), "cross": (200, 330), "causal": (384, 384)}[mode]
    C = heads * D
    rng = np.random.RandomState(sum(map(ord, ramp)) + D + Nk)
    q = h16(0.5 * rng.standard_normal((B, Nq, C)))
    k = h16(0.5 * rng.standard_normal((B, Nk, C)))
    v = h16(rng.standard_normal((B, Nk, C)))
    # one dedicated channel per head carries the ramp: q_last = 8, k_last(tile) = offset / (8 * scale * log2 e)
    scale_log2 = D ** -0.5 * 1.4426950408889634
    for hh in range(heads):
        q[:, :, hh * D + D - 1] = 8.0
        for j in range(Nk):
            k[:, j, hh * D + D - 1] = RAMPS[ramp](j // 64) / (8.0 * scale_log2)
    q, k = h16(q), h16(k)
    qt, kt, vt_ = [torch.tensor(t, dtype=torch.float64).reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
    sc = torch.matmul(qt, kt.transpose(2, 3)) * D ** -0.5
    if mode == "causal":
        sc = sc + torch.triu(torch.full((Nq, Nk), float("-inf"), dtype=torch.float64), 1)
    ref = torch.matmul(torch.softmax(sc, -1), vt_).permute(0, 2, 1, 3).reshape(B, Nq, C)
    ld = (Nk + 7) // 8 * 8
    vt = np.zeros((B, C, ld), np.float32)
    vt[:, :, :Nk] = v.transpose(0, 2, 1)
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    out = torch.empty((B, Nq, C), dtype=torch.float16, device=DEV)
    ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                  Nq * C, C, Nk * C, C, C * ld, ld, Nq * C, C, causal=(mode == "causal"))
    check(f"attention_{mode}_{ramp}_d{D}", out, ref, rel_l2=2e-3, max_abs=2e-2)


@pytest.mark.parametrize("ramp", [None] + sorted(RAMPS))
@pytest.mark.parametrize("D,heads,Nq,Nk", [(64, 2, 256, 256), (64, 5, 1024, 1024), (40, 8, 512, 512), (80, 2, 256, 640), (64, 1, 512, 64),
                                           (64, 3, 256, 4224)])
def test_attention_eight_wave_form(ops, ramp, D, heads, Nq, Nk):
    """attn8_kernel (csrc/attention.hip; opt-in, measured slower than the four-wave kernel): the eight-wave form -- two groups of four waves half a tile apart, PV of a tile deferred
    into the next matrix phase -- forced (option attn8 = 2) on shapes from one 256-query block up, with i.i.d. scores and with the
    stale-maximum ramps of the test above (the lazily moved reference maximum and the deferred PV interact: P(i) is produced under
    the reference of phase V(i) and consumed one phase later, after which O may be rescaled again).  Against the float64 softmax and
    against the four-wave kernel (same arithmetic per tile: equal to fp16 rounding of O)."""
    B = 2
    C = heads * D
    rng = np.random.RandomState(Nq + Nk + D + (sum(map(ord, ramp)) if ramp else 0))
    q = h16(0.5 * rng.standard_normal((B, Nq, C)))
    k = h16(0.5 * rng.standard_normal((B, Nk, C)))
    v = h16(rng.standard_normal((B, Nk, C)))
    if ramp:
        scale_log2 = D ** -0.5 * 1.4426950408889634
        for hh in range(heads):
            q[:, :, hh * D + D - 1] = 8.0
            for j in range(Nk):
                k[:, j, hh * D + D - 1] = RAMPS[ramp](j // 64) / (8.0 * scale_log2)
        q, k = h16(q), h16(k)
    qt, kt, vt_ = [torch.tensor(t, dtype=torch.float64).reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
    ref = torch.matmul(torch.softmax(torch.matmul(qt, kt.transpose(2, 3)) * D ** -0.5, -1), vt_).permute(0, 2, 1, 3).reshape(B, Nq, C)
    vt = np.ascontiguousarray(v.transpose(0, 2, 1))
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    outs = {}
    for form in (2, 0):
        ops.set_option("attn8", form)
        try:
            out = torch.full((B, Nq, C), float("nan"), dtype=torch.float16, device=DEV)
            ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                          Nq * C, C, Nk * C, C, C * Nk, Nk, Nq * C, C)
            torch.cuda.synchronize()
            outs[form] = out
        finally:
            ops.set_option("attn8", 0)
    check(f"attention8_{ramp}_d{D}_h{heads}_q{Nq}_k{Nk}", outs[2], ref, rel_l2=2e-3, max_abs=2e-2)
    check(f"attention8_vs_four_wave_{ramp}_d{D}_q{Nq}_k{Nk}", outs[2], outs[0], rel_l2=1e-3)


@pytest.mark.parametrize("ramp", [None] + sorted(RAMPS))
@pytest.mark.parametrize("D,heads,Nq,Nk,splits", [(64, 2, 256, 256, -1), (64, 5, 1024, 1024, -1), (40, 8, 512, 512, -1), (80, 2, 256, 640, -1),
                                                  (64, 1, 512, 128, -1), (64, 3, 200, 4224, -1), (40, 2, 100, 192, -1), (80, 1, 128, 2048, -1),
                                                  (64, 2, 256, 512, 2), (64, 1, 128, 4096, 8), (40, 8, 256, 640, 3), (80, 2, 128, 768, 3),
                                                  (64, 3, 384, 3200, 0)])
def test_attention_software_pipelined_form(ops, ramp, D, heads, Nq, Nk, splits):
    """attn_pipe_kernel (csrc/attention.hip, round 6; option attn_pipe): QK^T of key tile t + 1, softmax and PV of tile t interleaved MFMA
    by MFMA inside every wave, K staged one tile ahead of V, the next tile's row maximum taken at the end of the iteration.  Two,
    three and many key tiles (the odd / even tails of the two-tile unrolled loop), ragged query counts, i.i.d. scores and the
    stale-maximum ramps (the lazily moved reference is decided one iteration after its maximum was taken); unsplit (splits = -1) and
    split-KV launches (forced counts, 0 = the library's choice; a split's tile range starts at an even tile).  Against the float64
    softmax, and against attn_kernel BIT FOR BIT: the same operations on the same values in the same order per output element."""
    B = 2
    C = heads * D
    rng = np.random.RandomState(Nq + Nk + D + (sum(map(ord, ramp)) if ramp else 0))
    q = h16(0.5 * rng.standard_normal((B, Nq, C)))
    k = h16(0.5 * rng.standard_normal((B, Nk, C)))
    v = h16(rng.standard_normal((B, Nk, C)))
    if ramp:
        scale_log2 = D ** -0.5 * 1.4426950408889634
        for hh in range(heads):
            q[:, :, hh * D + D - 1] = 8.0
            for j in range(Nk):
                k[:, j, hh * D + D - 1] = RAMPS[ramp](j // 64) / (8.0 * scale_log2)
        q, k = h16(q), h16(k)
    qt, kt, vt_ = [torch.tensor(t, dtype=torch.float64).reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
    ref = torch.matmul(torch.softmax(torch.matmul(qt, kt.transpose(2, 3)) * D ** -0.5, -1), vt_).permute(0, 2, 1, 3).reshape(B, Nq, C)
    vt = np.ascontiguousarray(v.transpose(0, 2, 1))
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    outs = {}
    old = ops.get_option("attn_pipe")
    items = (Nq + 127) // 128 * heads * B
    ws = ops.attention_workspace(65536 + items * 8 * (128 * D * 2 + 1024), DEV) if splits >= 0 else None
    for form in (1, 0):
        ops.set_option("attn_pipe", form)
        try:
            out = torch.full((B, Nq, C), float("nan"), dtype=torch.float16, device=DEV)
            ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                          Nq * C, C, Nk * C, C, C * Nk, Nk, Nq * C, C, ws=ws, kv_splits=max(splits, 0))
            torch.cuda.synchronize()
            if ws is not None:
                assert int(ws[: items * 4].view(torch.int32).abs().sum()) == 0, "arrival counters not back at zero"
            outs[form] = out
        finally:
            ops.set_option("attn_pipe", old)
    check(f"attention_pipe_{ramp}_d{D}_h{heads}_q{Nq}_k{Nk}_s{splits}", outs[1], ref, rel_l2=2e-3, max_abs=2e-2)
    assert torch.equal(outs[1], outs[0]), f"pipelined attention differs from attn_kernel: max |d| = {(outs[1].float() - outs[0].float()).abs().max().item():.3e}"


@pytest.mark.parametrize("B,T,C,L,tile_m,lnfold", [
    (2, 1024, 640, 77, 0, True),      # SDv2 32 x 32 level at UNet batch 2: 10 heads, the LayerNorm-fold consumer form the planner emits
    (2, 256, 1280, 77, 0, True),      # 16 x 16 level, 20 heads
    (2, 64, 1280, 77, 64, True),      # 8 x 8 level: 64 tokens per sample, 64-row tiles
    (2, 576, 640, 77, 64, True),      # 24 x 24 level of a 768-pixel run: tokens per sample % 128 != 0 -> 64-row tiles
    (3, 128, 320, 77, 128, False),    # plain projection (explicit LayerNorm in front), 128-row tiles, odd batch
    (2, 256, 128, 64, 64, False),     # exactly one full key tile: no masked tile
    (1, 128, 192, 128, 0, False),     # two full key tiles
    (2, 128, 64, 5, 0, False),        # a handful of keys, one head
])
def test_dense_with_cross_attention_epilogue(ops, B, T, C, L, tile_m, lnfold):
    """mdx_gemm_desc.xattn_k (round 6): BasicTransformerBlock.attn2 -- q = LN(x) Wq^T, softmax(q K^T / sqrt(64)) V over the cached context
    keys -- with the attention as the EPILOGUE of the query projection (one 64-column tile = one head).  Against the oracle's fp32
    arithmetic, and against the two launches it replaces (projection, mdx_attention_f16) BIT FOR BIT."""
    heads = C // 64
    cap = (L + 7) // 8 * 8
    rng = np.random.RandomState(B * T + C + L)
    x = h16(rng.standard_normal((B * T, C)))
    wq = h16(rng.standard_normal((C, C)) / math.sqrt(C))
    k = h16(0.7 * rng.standard_normal((B, L, C)))
    v = h16(rng.standard_normal((B, L, C)))
    kd = torch.zeros((B, cap, C), dtype=torch.float16, device=DEV)
    kd[:, :L] = dev16(k)
    vtd = torch.zeros((B, C, cap), dtype=torch.float16, device=DEV)
    vtd[:, :, :L] = dev16(np.ascontiguousarray(v.transpose(0, 2, 1)))
    xd = dev16(x)
    scale = 64 ** -0.5
    kw = {}
    if lnfold:      # the planner's form: the projection runs on the raw rows with gamma-scaled weights + the producer's row statistics
        g = (1.0 + 0.1 * rng.standard_normal(C)).astype(np.float32)
        bt = (0.1 * rng.standard_normal(C)).astype(np.float32)
        xt = torch.tensor(x).float()
        xn = (xt - xt.mean(1, keepdim=True)) / torch.sqrt(xt.var(1, unbiased=False, keepdim=True) + 1e-5) * torch.tensor(g) + torch.tensor(bt)
        qref = xn @ torch.tensor(wq).float().T
        wg, sv, cb = ops.fold_layernorm(torch.tensor(wq).to(DEV), torch.tensor(g).to(DEV), torch.tensor(bt).to(DEV))
        wd = ops.pack_gemm_weight(wg)
        nt = C // 64
        st = torch.zeros((B * T, nt, 2), dtype=torch.float32, device=DEV)
        xs = xd.float().reshape(B * T, nt, 64)
        st[:, :, 0], st[:, :, 1] = xs.sum(2), (xs * xs).sum(2)
        kw = dict(ln_stats=st, ln_s=sv, bias=cb)      # (the fold's W beta term is the projection's bias)
    else:
        qref = torch.tensor(x).float() @ torch.tensor(wq).float().T
        wd = pack_dense(wq)
    qh = qref.reshape(B, T, heads, 64).permute(0, 2, 1, 3)
    kh = torch.tensor(k).float().reshape(B, L, heads, 64).permute(0, 2, 1, 3)
    vh = torch.tensor(v).float().reshape(B, L, heads, 64).permute(0, 2, 1, 3)
    ref = torch.matmul(torch.softmax(torch.matmul(qh, kh.transpose(2, 3)) * scale, -1), vh).permute(0, 2, 1, 3).reshape(B * T, C)
    # the two launches
    q = torch.empty((B * T, C), dtype=torch.float16, device=DEV)
    d0 = ops.make_gemm_desc(xd, wd, C, B, T, 1, C, q, C, tile_n=64, splitk=1, tile_m=tile_m, **kw)
    ops.gemm_run(d0)
    two = torch.empty((B * T, C), dtype=torch.float16, device=DEV)
    ops.attention(q.data_ptr(), kd.data_ptr(), vtd.data_ptr(), two.data_ptr(), B, heads, 64, T, L, scale,
                  T * C, C, cap * C, C, C * cap, cap, T * C, C)
    # the fused launch
    one = torch.full((B * T, C), float("nan"), dtype=torch.float16, device=DEV)
    d1 = ops.make_gemm_desc(xd, wd, C, B, T, 1, C, one, C, tile_n=64, splitk=1, tile_m=tile_m, xattn_k=kd, xattn_vt=vtd, xattn_len=L,
                            xattn_cap=cap, xattn_scale=scale, **kw)
    qq = ops.gemm_query(d1)
    assert qq[3] == 2 and qq[1] == 64 and qq[2] == 1, qq
    ops.gemm_run(d1)
    torch.cuda.synchronize()
    assert torch.equal(one, two), f"fused cross-attention differs from projection + mdx_attention_f16: max |d| = {(one.float() - two.float()).abs().max().item():.3e}"
    check(f"dense_xattn_B{B}_T{T}_C{C}_L{L}_tm{tile_m}_ln{int(lnfold)}", one, ref, rel_l2=3e-3, max_abs=3e-2)


@pytest.mark.parametrize("occ3", [1, 0])
@pytest.mark.parametrize("ramp", [None, "stale6.5", "always12", "sawtooth"])
@pytest.mark.parametrize("D,heads,Nq,Nk,splits", [
    (64, 2, 256, 512, 2), (64, 5, 200, 1024, 4), (64, 1, 128, 4096, 8), (64, 2, 256, 1000, 3),     # ragged last key tile in the last split
    (40, 8, 256, 512, 2), (80, 2, 128, 768, 3), (160, 1, 128, 512, 2), (64, 3, 384, 3200, 0),     # 0: the library's own choice (50 key tiles: two splits)
])
def test_attention_split_kv(ops, occ3, ramp, D, heads, Nq, Nk, splits):
    """mdx_attention_splitkv_f16 (csrc/attention.hip): the key tiles of every (batch, head, 128-query block) item dealt to
    2 ... 8 blocks, partials (normalised fp16 O, reference, row sum) through the workspace, the last arriver combines.  Against the
    float64 softmax (the tolerance of the unsplit kernel) and against the unsplit kernel (one more fp16 rounding of the partials:
    1e-3), with i.i.d. scores and with the stale-maximum ramps (each split starts its own reference; the combine weights
    l_s 2^(m_s - max m) span many octaves under a ramp), ragged queries / keys, both register-occupancy builds; the arrival counters
    at the head of the workspace must be zero again after every launch (the workspace is reused without clearing)."""
    B = 2
    C = heads * D
    rng = np.random.RandomState(Nq + Nk + D + splits + (sum(map(ord, ramp)) if ramp else 0))
    q = h16(0.5 * rng.standard_normal((B, Nq, C)))
    k = h16(0.5 * rng.standard_normal((B, Nk, C)))
    v = h16(rng.standard_normal((B, Nk, C)))
    if ramp:
        scale_log2 = D ** -0.5 * 1.4426950408889634
        for hh in range(heads):
            q[:, :, hh * D + D - 1] = 8.0
            for j in range(Nk):
                k[:, j, hh * D + D - 1] = RAMPS[ramp](j // 64) / (8.0 * scale_log2)
        q, k = h16(q), h16(k)
    qt, kt, vt_ = [torch.tensor(t, dtype=torch.float64).reshape(B, -1, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
    ref = torch.matmul(torch.softmax(torch.matmul(qt, kt.transpose(2, 3)) * D ** -0.5, -1), vt_).permute(0, 2, 1, 3).reshape(B, Nq, C)
    ld = (Nk + 7) // 8 * 8
    vt = np.zeros((B, C, ld), np.float32)
    vt[:, :, :Nk] = v.transpose(0, 2, 1)
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    items = (Nq + 127) // 128 * heads * B
    ws = ops.attention_workspace(65536 + items * 8 * (128 * D * 2 + 1024), DEV)
    ops.set_option("attn_occ3", occ3)
    try:
        if splits == 0:
            assert ops.attention_ws_bytes(B, heads, D, Nq, Nk) > 0, "the auto policy should split this under-filled launch"
        outs = []
        for rep in range(2):        # the second launch reuses the workspace as the first left it
            out = torch.full((B, Nq, C), float("nan"), dtype=torch.float16, device=DEV)
            ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                          Nq * C, C, Nk * C, C, C * ld, ld, Nq * C, C, ws=ws, kv_splits=splits)
            torch.cuda.synchronize()
            assert int(ws[: items * 4].view(torch.int32).abs().sum()) == 0, "arrival counters not back at zero"
            outs.append(out)
        plain = torch.empty((B, Nq, C), dtype=torch.float16, device=DEV)
        ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), plain.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                      Nq * C, C, Nk * C, C, C * ld, ld, Nq * C, C)
        torch.cuda.synchronize()
    finally:
        ops.set_option("attn_occ3", 1)
    assert torch.equal(outs[0], outs[1]), "split-KV attention is not reproducible launch to launch"
    name = f"attention_splitkv{splits}_occ{occ3}_{ramp}_d{D}_q{Nq}_k{Nk}"
    check(name, outs[0], ref, rel_l2=2e-3, max_abs=2e-2)
    check(name + "_vs_unsplit", outs[0], plain, rel_l2=1e-3)


def test_attention_split_kv_rejects_bad_arguments(ops):
    """Error behaviour of the split-KV entry point: too many splits for the key count, a workspace that is too small."""
    from minddiffusion_amd._lib import MdxError
    B, heads, D, N = 1, 1, 64, 256
    C = heads * D
    x = torch.zeros((B, N, C), dtype=torch.float16, device=DEV)
    vt = torch.zeros((B, C, N), dtype=torch.float16, device=DEV)
    o = torch.empty_like(x)
    args = (x.data_ptr(), x.data_ptr(), vt.data_ptr(), o.data_ptr(), B, heads, D, N, N, 0.125, N * C, C, N * C, C, C * N, N, N * C, C)
    with pytest.raises(MdxError, match="splits"):
        ops.attention(*args, ws=ops.attention_workspace(1 << 20, DEV), kv_splits=3)      # 4 key tiles: at most 2 splits
    with pytest.raises(MdxError, match="workspace"):
        ops.attention(*args, ws=ops.attention_workspace(256, DEV), kv_splits=2)


def test_attention_split_kv_shared_workspace(ops):
    """One workspace serves launches of DIFFERENT shapes in turn (a UNet plan's 64 x 64 and 32 x 32 self-attentions): the arrival
    counters sit in a fixed 64 KiB region at its head, so a small launch's partials cannot land on a bigger launch's counters
    (they did when the region was sized per launch: the next big launch trapped on a non-zero counter)."""
    shapes = [(2, 5, 512, 64, 2), (2, 10, 256, 64, 2), (1, 8, 384, 40, 3), (2, 5, 512, 64, 4)]
    need = max(65536 + ((n + 127) // 128 * h * b) * s_ * (128 * d * 2 + 1024) for b, h, n, d, s_ in shapes)
    ws = ops.attention_workspace(need, DEV)
    rng = np.random.RandomState(5)
    for rep in range(2):
        for b, h, n, d, s_ in shapes:
            c = h * d
            q, k, v = (h16(rng.standard_normal((b, n, c))) for _ in range(3))
            ref = _attn_ref(q, k, v, h)
            qd, kd, vtd = dev16(q), dev16(k), dev16(np.ascontiguousarray(v.transpose(0, 2, 1)))
            out = torch.empty((b, n, c), dtype=torch.float16, device=DEV)
            ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), b, h, d, n, n, d ** -0.5,
                          n * c, c, n * c, c, c * n, n, n * c, c, ws=ws, kv_splits=s_)
            torch.cuda.synchronize()
            assert int(ws[:65536].view(torch.int32).abs().sum()) == 0
            check(f"attention_splitkv_shared_ws_rep{rep}_B{b}_h{h}_N{n}_d{d}_s{s_}", out, ref, rel_l2=2e-3, max_abs=2e-2)


@pytest.mark.parametrize("B,heads,N,D", [(2, 2, 80, 64), (1, 3, 77, 64), (2, 1, 200, 64), (1, 2, 384, 64), (1, 8, 80, 40)])
def test_attention_causal(ops, B, heads, N, D):
    """mdx_attention_causal_f16: key j is visible to query i iff j <= i (text_encoder.py:136-139); N spans one tile,
    a ragged tail and several 128-query blocks (blocks skip the key tiles that lie wholly in their future)."""
    C = heads * D
    rng = np.random.RandomState(N + heads + D)
    q, k, v = (h16(rng.standard_normal((B, N, C))) for _ in range(3))
    qt, kt, vt_ = [torch.tensor(t).float().reshape(B, N, heads, D).permute(0, 2, 1, 3) for t in (q, k, v)]
    sc = torch.matmul(qt, kt.transpose(2, 3)) * D ** -0.5 + torch.triu(torch.full((N, N), float("-inf")), 1)
    ref = torch.matmul(torch.softmax(sc, -1), vt_).permute(0, 2, 1, 3).reshape(B, N, C)
    ld = (N + 7) // 8 * 8
    vt = np.zeros((B, C, ld), np.float32)
    vt[:, :, :N] = v.transpose(0, 2, 1)
    qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
    out = torch.empty((B, N, C), dtype=torch.float16, device=DEV)
    ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), out.data_ptr(), B, heads, D, N, N, D ** -0.5,
                  N * C, C, N * C, C, C * ld, ld, N * C, C, causal=True)
    check(f"attention_causal_B{B}_h{heads}_N{N}_d{D}", out, ref, rel_l2=2e-3, max_abs=2e-2)
    # row 0 attends to key 0 only
    assert float((out[:, 0].float().cpu() - torch.tensor(v[:, 0].astype(np.float32))).abs().max()) < 2e-3


def test_gemm_quick_gelu_epilogue(ops):
    """MDX_EPI_QUICKGELU: x * sigmoid(1.702 x) (Wukong text encoder MLP), direct and split-K paths."""
    rng = np.random.RandomState(5)
    M, N, K = 160, 256, 512
    a = h16(rng.standard_normal((M, K)))
    w = h16(rng.standard_normal((N, K)) / math.sqrt(K) * 2)
    bv = rng.standard_normal(N).astype(np.float32)
    x = torch.tensor(a).float() @ torch.tensor(w).float().T + torch.tensor(bv)
    ref = x * torch.sigmoid(1.702 * x)
    for sk in (1, 2):
        out = ops.gemm(dev16(a), pack_dense(w), N, 1, M, 1, K, bias=dev32(bv), epilogue=ops.EPI_QUICKGELU, splitk=sk)
        check(f"gemm_quick_gelu_s{sk}", out, ref, rel_l2=1e-3)


def test_attention_fused_qk_layout(ops):
    """q and k as column slices of one [M, 2C] projection buffer (how the UNet calls it)."""
    B, heads, N, D = 2, 2, 128, 64
    C = heads * D
    rng = np.random.RandomState(3)
    qk = h16(rng.standard_normal((B, N, 2 * C)))
    v = h16(rng.standard_normal((B, N, C)))
    ref = _attn_ref(qk[:, :, :C], qk[:, :, C:], v, heads)
    qkd = dev16(qk)
    vtd = dev16(v.transpose(0, 2, 1))
    out = torch.empty((B, N, C), dtype=torch.float16, device=DEV)
    ops.attention(qkd.data_ptr(), qkd.data_ptr() + C * 2, vtd.data_ptr(), out.data_ptr(), B, heads, D, N, N, D ** -0.5,
                  N * 2 * C, 2 * C, N * 2 * C, 2 * C, C * N, N, N * C, C)
    check("attention_fused_qk_layout", out, ref, rel_l2=2e-3, max_abs=2e-2)


# --------------------------------------------------------------------------- time embedding
def test_timestep_embedding_and_dense_small(ops):
    t = np.array([981.0, 1.0, 500.5], np.float32)
    e = ops.timestep_embedding(dev32(t), 320)
    ref = O.timestep_embedding(torch.tensor(t), 320)
    check("timestep_embedding", e, ref, max_abs=2e-4)   # fp32 sin/cos of arguments up to 1e3
    np.testing.assert_allclose(e[0, :3].cpu().numpy(), [0.6799572, -0.7984292, 0.578114], atol=2e-4)
    rng = np.random.RandomState(0)
    for M, N, K, ai, ao in ((3, 1280, 320, False, True), (2, 1280, 1280, False, False), (9, 2000, 1280, True, False)):
        x = rng.standard_normal((M, K)).astype(np.float32)
        w = h16(rng.standard_normal((N, K)) / math.sqrt(K))
        b = rng.standard_normal(N).astype(np.float32)
        xin = O.silu(torch.tensor(x)) if ai else torch.tensor(x)
        ref = xin @ torch.tensor(w).T + torch.tensor(b)
        if ao:
            ref = O.silu(ref)
        out = ops.dense_small(dev32(x), dev16(w), dev32(b), act_in=ai, act_out=ao)
        check(f"dense_small_M{M}_N{N}_K{K}", out, ref, rel_l2=2e-5)


# --------------------------------------------------------------------------- sampler step
@pytest.mark.parametrize("cfg,order,sigma", [(True, 0, 0.0), (False, 1, 0.0), (True, 3, 0.0), (True, 0, 0.3)])
def test_sampler_step(ops, cfg, order, sigma):
    rng = np.random.RandomState(order + int(cfg))
    B, C, H, W = 2, 4, 8, 8
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    eu = h16(rng.standard_normal((B, C, H, W)))
    ec = h16(rng.standard_normal((B, C, H, W)))
    olds = [rng.standard_normal((B, C, H, W)).astype(np.float32) for _ in range(order)]
    coef = {0: (1, 0, 0, 0), 1: (1.5, -0.5, 0, 0), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[order]
    noise = rng.standard_normal((B, C, H, W)).astype(np.float32)
    scale = 7.5
    a_t, a_prev = np.float32(0.3), np.float32(0.5)
    e_t = eu + scale * (ec - eu) if cfg else ec
    ep = coef[0] * e_t + sum(c * o for c, o in zip(coef[1:], olds))
    px0 = (x - np.sqrt(1 - a_t) * ep) / np.sqrt(a_t)
    xp = np.sqrt(a_prev) * px0 + np.sqrt(1 - a_prev - sigma ** 2) * ep + sigma * noise

    def eps_buf(e):  # NHWC fp16 with 8-channel stride
        buf = np.zeros((B, H * W, 8), np.float32)
        buf[:, :, :4] = nhwc(e)
        return dev16(buf)

    xd = dev32(x)
    e_out, x_out, p_out = torch.empty_like(xd), torch.empty_like(xd), torch.empty_like(xd)
    ops.sampler_step(xd, eps_buf(eu) if cfg else None, eps_buf(ec), 8, scale, [dev32(o) for o in olds], coef,
                     np.sqrt(a_t), np.sqrt(1 - a_t), np.sqrt(a_prev), np.sqrt(1 - a_prev - np.float32(sigma) ** 2),
                     sigma, dev32(noise) if sigma else None, e_out, x_out, p_out)
    check(f"sampler_step_cfg{int(cfg)}_o{order}_s{sigma}_x", x_out, xp, rel_l2=1e-5)
    check(f"sampler_step_cfg{int(cfg)}_o{order}_s{sigma}_p", p_out, px0, rel_l2=1e-5)
    check(f"sampler_step_cfg{int(cfg)}_o{order}_s{sigma}_e", e_out, e_t, rel_l2=1e-5)


# --------------------------------------------------------------------------- GroupNorm statistics from the producer
@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,splitk", [
    (2, 16, 16, 64, 128, 1, 1),      # generic kernel, single pass (row block = M tile)
    (2, 16, 32, 128, 192, 3, 1),     # HALO conv: row block = one 8x16 patch
    (2, 16, 16, 320, 320, 3, 3),     # split-K: the tiled reduce kernel (64-row blocks)
    (3, 32, 32, 64, 320, 1, 2),      # split-K dense, N tail of the 64-column reduce tiles
    (1, 64, 64, 64, 320, 3, 1),      # the 64x64-latent level
])
def test_gemm_colstats_and_groupnorm_from_them(ops, B, H, W, Cin, Cout, ks, splitk):
    """mdx_gemm_desc.colstats_out: per-row-block, per-column {sum, sumsq} of the fp16 values the launch stores (bias,
    time-embedding row and residual included), and mdx_groupnorm_colstats_f16 folding them -- against the two-launch GroupNorm
    on the same tensor (bit-identical input, statistics summed in a different order) and the oracle."""
    from minddiffusion_amd import ops as _ops
    rng = np.random.RandomState(B + H + Cin + Cout + ks)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, ks, ks)) / math.sqrt(ks * ks * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    emb = rng.standard_normal((B, Cout)).astype(np.float32)
    res = h16(rng.standard_normal((B, Cout, H, W)))
    a, wp = dev16(nhwc(x)), pack_conv(w)
    out = torch.empty((B * H * W, Cout), dtype=torch.float16, device=DEV)
    embd, resd, biasd = dev32(emb), dev16(nhwc(res)), dev32(bv)     # (descriptors hold raw pointers: keep the tensors alive)
    d = _ops.make_gemm_desc(a, wp, Cout, B, H, W, Cin, out, Cout, bias=biasd, ksize=ks, splitk=splitk, rowbias=embd,
                            rowbias_ld=Cout, residual=resd, residual_ld=Cout)
    need = _ops.gemm_workspace_bytes(d)
    wsb = _ops.new_gemm_workspace(need, DEV)
    d.workspace, d.workspace_bytes = wsb.data_ptr(), wsb.numel() * 4
    rows = _ops.gemm_query(d)[5]
    assert rows > 0 and (H * W) % rows == 0
    nrb = H * W // rows
    cs = torch.full((B * nrb, Cout, 2), float("nan"), dtype=torch.float32, device=DEV)
    d.colstats_out, d.colstats_cap = cs.data_ptr(), B * nrb - 1
    with pytest.raises(Exception):          # a buffer that is one row block short is refused, not overrun
        _ops.gemm_run(d)
    d.colstats_cap = B * nrb
    _ops.gemm_run(d)
    o = out.float().view(B, H * W, Cout)
    if rows == 128 and ks == 3:          # HALO patches: 8 rows x 16 columns of pixels
        img = o.view(B, H // 8, 8, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B * nrb, 128, Cout)
    else:
        img = o.reshape(B * nrb, rows, Cout)
    check(f"colstats_sum_{ks}x{ks}_k{splitk}_rows{rows}", cs[..., 0], img.sum(1), rel_l2=1e-5)
    check(f"colstats_sumsq_{ks}x{ks}_k{splitk}_rows{rows}", cs[..., 1], (img * img).sum(1), rel_l2=1e-5)
    # consumer: GroupNorm(+SiLU) of cat(out, other) with the second source's statistics from a second producer
    g = (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32)
    bt = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    o3 = out.view(B, H * W, Cout)
    y_ref = _ops.groupnorm(o3, None, dev32(g), dev32(bt), 1e-5, True)
    y = _ops.groupnorm_colstats(o3, cs, nrb, None, None, 0, dev32(g), dev32(bt), 1e-5, True)
    check(f"groupnorm_from_colstats_{ks}x{ks}_k{splitk}", y, y_ref, rel_l2=2e-4, max_abs=4e-3)
    ref = O.silu(O.group_norm(torch.tensor(from_nhwc(o.cpu().numpy().reshape(B, H * W, Cout), B, H, W)), torch.tensor(g),
                              torch.tensor(bt), 1e-5))
    check(f"groupnorm_from_colstats_vs_oracle_{ks}x{ks}_k{splitk}", from_nhwc(y.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)


def test_groupnorm_colstats_two_sources(ops):
    """UNet output blocks: GroupNorm over cat(h, skip) (openaimodel.py:568) whose sources come from two producers with
    different row-block sizes; groups straddle the source boundary (C = 192 + 64, 8 channels per group) and, for C = 960 =
    640 + 320, are not aligned to the 8-channel store granule (30 channels per group)."""
    from minddiffusion_amd import ops as _ops
    for (C1, C2, HW, r1, r2) in ((192, 64, 1024, 128, 64), (640, 320, 4096, 64, 128)):
        rng = np.random.RandomState(C1 + C2)
        B = 2
        x1 = h16(rng.standard_normal((B, HW, C1)) + 0.3)
        x2 = h16(rng.standard_normal((B, HW, C2)) * 2 - 0.5)
        st = lambda x, r: np.stack([x.reshape(B * (HW // r), r, -1).sum(1), (x.astype(np.float64) ** 2).reshape(B * (HW // r), r, -1).sum(1)], -1).astype(np.float32)
        g = (1 + 0.1 * rng.standard_normal(C1 + C2)).astype(np.float32)
        bt = (0.1 * rng.standard_normal(C1 + C2)).astype(np.float32)
        y = _ops.groupnorm_colstats(dev16(x1), dev32(st(x1, r1)), HW // r1, dev16(x2), dev32(st(x2, r2)), HW // r2,
                                    dev32(g), dev32(bt), 1e-6, False)
        xc = np.concatenate([x1, x2], 2).transpose(0, 2, 1).reshape(B, C1 + C2, HW, 1)
        ref = O.group_norm(torch.tensor(xc), torch.tensor(g), torch.tensor(bt), 1e-6)
        got = y.float().cpu().numpy().transpose(0, 2, 1).reshape(B, C1 + C2, HW, 1)
        check(f"groupnorm_colstats_two_sources_{C1}+{C2}", got, ref, rel_l2=1e-3)


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,splitk,silu", [
    (2, 16, 16, 320, 320, 3, 5, True),      # HALO conv split over chunks, 16x16 level
    (2, 8, 8, 1280, 1280, 3, 10, True),     # 8x8 level (two-sample HALO tiles), cpg = 40
    (3, 8, 8, 640, 1920, 1, 2, False),      # dense, cpg = 60 (15-chunk column blocks), odd batch
    (2, 16, 16, 128, 256, 3, 2, True),      # cpg = 8: one chunk per group
])
def test_groupnorm_fused_with_splitk_reduce(ops, B, H, W, Cin, Cout, ks, splitk, silu):
    """mdx_gemm_desc.defer_reduce + mdx_groupnorm_from_splitk_f16: the GroupNorm launch sums the producer's split-K slabs
    itself.  The conv output it stores and the normalised tensor are compared with the ordinary launch (in-kernel split-K
    reduce, whose epilogue rounds the accumulator to fp16 once before bias / residual are added): same sums in the same
    order, at most one fp16 rounding apart."""
    from minddiffusion_amd import ops as _ops
    rng = np.random.RandomState(B + H + Cin + Cout)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, ks, ks)) / math.sqrt(ks * ks * Cin))
    a, wp = dev16(nhwc(x)), pack_conv(w)
    bias, emb = dev32(rng.standard_normal(Cout).astype(np.float32)), dev32(rng.standard_normal((B, Cout)).astype(np.float32))
    res = dev16(nhwc(h16(rng.standard_normal((B, Cout, H, W)))))
    g, bt = dev32((1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32)), dev32((0.1 * rng.standard_normal(Cout)).astype(np.float32))

    def build(out):
        d = _ops.make_gemm_desc(a, wp, Cout, B, H, W, Cin, out, Cout, bias=bias, ksize=ks, splitk=splitk, rowbias=emb,
                                rowbias_ld=Cout, residual=res, residual_ld=Cout)
        return d
    out_a = torch.zeros((B, H * W, Cout), dtype=torch.float16, device=DEV)
    out_b = torch.zeros_like(out_a)
    da, db = build(out_a), build(out_b)
    need = _ops.gemm_workspace_bytes(da)
    assert need > 0
    ws = _ops.new_gemm_workspace(need, DEV)
    for d in (da, db):
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    _ops.gemm_run(da)
    y_a = _ops.groupnorm(out_a, None, g, bt, 1e-5, silu)
    db.defer_reduce = 1
    assert _ops.groupnorm_from_splitk_ok(db)
    _ops.gemm_run(db)
    assert float(out_b.abs().max()) == 0.0          # the deferred launch left the output to its consumer
    y_b = torch.empty_like(out_b)
    _ops.groupnorm_from_splitk(db, g, bt, 1e-5, silu, y_b)
    ulp = 2.0 ** -10
    assert float((out_a.float() - out_b.float()).abs().max()) <= 2 * ulp * float(out_a.float().abs().max())
    assert float((y_a.float() - y_b.float()).abs().max()) <= 8 * ulp * max(1.0, float(y_a.float().abs().max()))
    xr = O.conv2d(torch.tensor(x), torch.tensor(w), bias.cpu(), padding=ks // 2) + emb.cpu()[:, :, None, None] \
        + torch.tensor(from_nhwc(res.float().cpu().numpy(), B, H, W))
    ref = O.group_norm(xr, g.cpu(), bt.cpu(), 1e-5)
    check(f"groupnorm_from_splitk_{ks}x{ks}_{Cin}to{Cout}", from_nhwc(y_b.float().cpu().numpy(), B, H, W),
          O.silu(ref) if silu else ref, rel_l2=2e-3)
    # a producer that does not split cannot defer
    dn = _ops.make_gemm_desc(a, wp, Cout, B, H, W, Cin, out_b, Cout, bias=bias, ksize=ks, splitk=1)
    dn.defer_reduce = 1
    with pytest.raises(Exception):
        _ops.gemm_run(dn)


@pytest.mark.parametrize("B,H,W,Cin,Cout,splitk,tile_n", [
    (2, 8, 8, 128, 128, 1, 64),       # two-sample 8 x 8 tiles, one chunk pair, no split
    (2, 8, 8, 1280, 1280, 10, 64),    # the deepest UNet level: M = 128, K = 11520, ten chunk-aligned splits (reduce kernel)
    (2, 8, 8, 640, 320, 3, 64),       # in-kernel split-K reduce on top of the streamed weights
    (3, 8, 8, 256, 192, 2, 64),       # odd batch: the second sample of the last tile does not exist
    (2, 16, 16, 1280, 1280, 5, 64),   # 16 x 16 level: 8 x 16 patches (PW = 16), M = 512
    (1, 16, 32, 192, 128, 1, 128),    # 128-column tiles
    (2, 8, 8, 1280, 1280, 20, 128),   # 128-column tiles + slab split-K: the four waves side by side along N (W4)
    (2, 16, 16, 640, 256, 5, 128),    # W4 on 8 x 16 patches
    (3, 8, 8, 384, 128, 6, 128),      # W4, odd batch
])
def test_conv3x3_weight_streaming_form_is_bit_identical(ops, B, H, W, Cin, Cout, splitk, tile_n):
    """mdx_gemm_desc.w_frag: the HALO 3x3 conv reading fragment-major weights straight into registers (no weight tiles in LDS,
    one barrier per 64-channel chunk) multiplies and adds every output's products in the same order as the tile-major form --
    bit-identical, with bias + per-sample time-embedding row + residual on top -- and agrees with the fp32 reference to fp16
    accuracy."""
    rng = np.random.RandomState(B * H + Cin + Cout)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    wt = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    rb = rng.standard_normal((B, Cout)).astype(np.float32)
    res = h16(rng.standard_normal((B, Cout, H, W)))
    ref = O.conv2d(torch.tensor(x), torch.tensor(wt), torch.tensor(bv)) + torch.tensor(rb)[:, :, None, None] + torch.tensor(res)
    xd, resd = dev16(nhwc(x)), dev16(nhwc(res))
    outs = {}
    for frag in (0, 1):
        wd = ops.pack_conv_weight_frag(torch.tensor(wt).to(DEV)) if frag else pack_conv(wt)
        out = torch.empty((B, H * W, Cout), dtype=torch.float16, device=DEV)
        bd, rbd = dev32(bv), dev32(rb)      # (a descriptor holds raw pointers only)
        d = ops.make_gemm_desc(xd, wd, Cout, B, H, W, Cin, out, Cout, bias=bd, rowbias=rbd, rowbias_ld=Cout,
                               residual=resd, residual_ld=Cout, ksize=3, splitk=splitk, tile_m=128, tile_n=tile_n, w_frag=frag)
        need = ops.gemm_workspace_bytes(d)
        ws = ops.new_gemm_workspace(max(need, 1 << 20), DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        q = ops.gemm_query(d)
        assert q[3] == 1 and q[0] == 128, q
        ops.gemm_run(d)
        torch.cuda.synchronize()
        outs[frag] = out.clone()
    assert torch.equal(outs[0], outs[1]), "weight-streaming form differs from the tile-major form"
    got = from_nhwc(outs[1].float().cpu().numpy().reshape(B, H * W, Cout), B, H, W)
    check(f"conv3x3_stream_B{B}_{H}x{W}_{Cin}_{Cout}_s{splitk}", got, ref, rel_l2=1e-3)
    # the launch refuses fragment-major weights where the HALO kernel does not apply
    bad = ops.make_gemm_desc(xd, wd, Cout, B, H, W, Cin, out, Cout, ksize=3, stride=2, w_frag=1)
    from minddiffusion_amd._lib import MdxError
    with pytest.raises(MdxError):
        ops.gemm_run(bad)


@pytest.mark.parametrize("nrb,C,B", [(512, 192, 2), (72, 320, 3), (65, 640, 1)])
def test_groupnorm_colstats_two_level_fold(ops, nrb, C, B):
    """mdx_colstats_fold_f32 + mdx_groupnorm_colstats_f16: column partials with more than 64 row blocks per sample (GLIDE's
    256 x 256 level: 512 HALO patches; SDv2 at 96 x 96: 72) folded once to <= 64 blocks, then normalised -- against the
    fp32 GroupNorm of the tensor and against the fold done on the host."""
    rows = 16
    HW = nrb * rows
    rng = np.random.RandomState(nrb + C)
    x = h16(rng.standard_normal((B, HW, C)) * (0.5 + rng.rand(C)) + rng.standard_normal(C))
    g, b = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    xd = dev16(x)
    blk = xd.float().reshape(B * nrb, rows, C)
    cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
    f = ops.FoldedColStats(cs, nrb, B)
    folded, nrb2 = f.fold()
    torch.cuda.synchronize()
    assert nrb2 <= 64
    fac = (nrb + nrb2 - 1) // nrb2
    pad = torch.zeros((B, nrb2 * fac - nrb, C, 2), device=DEV)
    host = torch.cat([cs.reshape(B, nrb, C, 2), pad], 1).reshape(B, nrb2, fac, C, 2).sum(2).reshape(B * nrb2, C, 2)
    check(f"colstats_fold_nrb{nrb}", folded, host, rel_l2=1e-6)
    out = ops.groupnorm_colstats(xd, f, nrb, None, None, 0, dev32(g), dev32(b), 1e-5, True)
    xt = torch.tensor(x).permute(0, 2, 1).reshape(B, C, HW, 1)
    ref = O.silu(O.group_norm(xt, torch.tensor(g), torch.tensor(b), 1e-5)).reshape(B, C, HW).permute(0, 2, 1)
    check(f"groupnorm_folded_colstats_nrb{nrb}_C{C}", out, ref, rel_l2=1e-3)


@pytest.mark.parametrize("B,H,W,C,Cs1,Cs2,splitk", [
    (2, 16, 16, 128, 64, 0, 1),        # single-source skip, one chunk pair
    (2, 64, 64, 320, 320, 320, 1),     # the 64 x 64 up path: 640 -> 320 skip over the virtual concat, no split
    (2, 32, 32, 640, 640, 320, 3),     # in-kernel split-K: every split takes its share of the skip tiles
    (2, 16, 16, 1280, 1280, 1280, 5),  # slab split-K + reduce kernel
    (2, 8, 8, 1280, 1280, 1280, 10),   # 8 x 8 two-sample tiles
    (3, 8, 8, 128, 192, 64, 2),        # odd batch on the two-sample tiles
])
def test_conv3x3_with_fused_skip_connection(ops, B, H, W, C, Cs1, Cs2, splitk):
    """mdx_gemm_desc.skip_w: out = conv3x3(h) + conv1x1(cat(x, x2)) + biases + time-embedding row in ONE launch (ResBlock
    out_layers conv + skip_connection, openaimodel.py:174, 201-205) against the fp32 reference and against the two launches it
    replaces (conv1x1 -> fp16 -> residual of the conv3x3: one more fp16 rounding, 1e-3)."""
    rng = np.random.RandomState(B * H + C + Cs1 + Cs2)
    hmap = h16(rng.standard_normal((B, C, H, W)))
    x1 = h16(rng.standard_normal((B, Cs1, H, W)))
    x2 = h16(rng.standard_normal((B, Cs2, H, W))) if Cs2 else None
    w3 = h16(rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C))
    w1 = h16(rng.standard_normal((C, Cs1 + Cs2, 1, 1)) / math.sqrt(Cs1 + Cs2))
    b3, b1 = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    xcat = np.concatenate([x1, x2], 1) if Cs2 else x1
    ref = (O.conv2d(torch.tensor(hmap), torch.tensor(w3), torch.tensor(b3))
           + O.conv2d(torch.tensor(xcat), torch.tensor(w1), torch.tensor(b1), padding=0))
    hd, x1d = dev16(nhwc(hmap)), dev16(nhwc(x1))
    x2d = dev16(nhwc(x2)) if Cs2 else None
    w3p, w1p = pack_conv(w3), pack_conv(w1)
    out = torch.empty((B, H * W, C), dtype=torch.float16, device=DEV)
    bsum = dev32(b3 + b1)           # (a descriptor holds raw pointers only: keep every tensor it points at alive)
    d = ops.make_gemm_desc(hd, w3p, C, B, H, W, C, out, C, bias=bsum, ksize=3, splitk=splitk,
                           skip_a=x1d, skip_a2=x2d, skip_c1=Cs1, skip_c2=Cs2, skip_w=w1p)
    ws = ops.new_gemm_workspace(max(ops.gemm_workspace_bytes(d), 1 << 20), DEV)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert ops.gemm_query(d)[3] == 1
    ops.gemm_run(d)
    torch.cuda.synchronize()
    got = from_nhwc(out.float().cpu().numpy(), B, H, W)
    check(f"conv3x3_fused_skip_B{B}_{H}x{W}_{C}_{Cs1}+{Cs2}_s{splitk}", got, ref, rel_l2=1e-3)
    # the two launches it replaces
    skip = ops.gemm(x1d, w1p, C, B, H, W, Cs1, a2=x2d, c2=Cs2, bias=dev32(b1))
    two = ops.gemm(hd, w3p, C, B, H, W, C, bias=dev32(b3), ksize=3, residual=skip, residual_ld=C)
    check(f"conv3x3_fused_skip_vs_two_launches_{H}x{W}_{C}", out.reshape(-1, C), two, rel_l2=1e-3)
    # refused where the HALO kernel does not apply
    from minddiffusion_amd._lib import MdxError
    bad = ops.make_gemm_desc(hd, w3p, C, B, H, W, C, out, C, ksize=3, stride=2, skip_a=x1d, skip_c1=Cs1, skip_w=w1p)
    with pytest.raises(MdxError):
        ops.gemm_run(bad)


@pytest.mark.parametrize("B,H,W,Cin,Cout,splitk,rows,skip", [
    (2, 16, 16, 128, 128, 1, 64, False),
    (2, 64, 64, 320, 320, 1, 128, False),     # the 64 x 64 level (in_layers / out_layers convs)
    (2, 64, 64, 320, 320, 1, 32, False),      # statistics in 32-row blocks, as a fused SpatialTransformer tail emits them
    (2, 32, 32, 320, 320, 5, 128, False),     # slab split-K with single-chunk splits
    (2, 32, 32, 320, 640, 3, 128, False),
    (2, 32, 32, 640, 640, 3, 128, False),     # 32 x 32 level, in-kernel split-K
    (2, 32, 32, 640, 640, 3, 128, True),      # ... with the skip_connection riding on the same launch
    (3, 16, 32, 192, 64, 2, 128, False),      # 6 channels per group, odd batch
])
def test_conv3x3_with_fused_input_groupnorm(ops, B, H, W, Cin, Cout, splitk, rows, skip):
    """mdx_gemm_desc.gn_colstats: GroupNorm(32) -> SiLU -> Conv2d 3x3 (openaimodel.py:136-138, 159-163) in ONE launch on the RAW
    input, statistics folded from the producer's per-row-block column partials -- against the fp32 reference and against the
    GroupNorm launch + conv launch it replaces (same formulas, another summation order of the statistics: 1e-3)."""
    rng = np.random.RandomState(B * H + Cin + Cout + rows)
    x = h16(rng.standard_normal((B, Cin, H, W)) * (0.5 + rng.rand(Cin))[None, :, None, None] + rng.standard_normal(Cin)[None, :, None, None])
    g, bt = (1.0 + 0.2 * rng.standard_normal(Cin)).astype(np.float32), (0.1 * rng.standard_normal(Cin)).astype(np.float32)
    wt = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    rb = rng.standard_normal((B, Cout)).astype(np.float32)
    a = O.silu(O.group_norm(torch.tensor(x), torch.tensor(g), torch.tensor(bt), 1e-5))
    ref = O.conv2d(a, torch.tensor(wt), torch.tensor(bv)) + torch.tensor(rb)[:, :, None, None]
    kw = {}
    if skip:
        xs = h16(rng.standard_normal((B, 2 * Cout, H, W)))
        ws = h16(rng.standard_normal((Cout, 2 * Cout, 1, 1)) / math.sqrt(2 * Cout))
        ref = ref + O.conv2d(torch.tensor(xs), torch.tensor(ws), None, padding=0)
        xsd, wsd = dev16(nhwc(xs)), pack_conv(ws)
        kw = dict(skip_a=xsd, skip_c1=2 * Cout, skip_w=wsd)
    xd = dev16(nhwc(x))
    nrb = H * W // rows
    blk = xd.float().reshape(B * nrb, rows, Cin)
    cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
    out = torch.empty((B, H * W, Cout), dtype=torch.float16, device=DEV)
    wd, bd, rbd, gd, btd = pack_conv(wt), dev32(bv), dev32(rb), dev32(g), dev32(bt)     # (a descriptor holds raw pointers only)
    d = ops.make_gemm_desc(xd, wd, Cout, B, H, W, Cin, out, Cout, bias=bd, rowbias=rbd, rowbias_ld=Cout,
                           ksize=3, splitk=splitk, tile_n=64, gn_colstats=cs, gn_nrb=nrb, gn_gamma=gd, gn_beta=btd,
                           gn_eps=1e-5, gn_silu=1, **kw)
    ws_ = ops.new_gemm_workspace(max(ops.gemm_workspace_bytes(d), 1 << 20), DEV)
    d.workspace, d.workspace_bytes = ws_.data_ptr(), ws_.numel() * 4
    q = ops.gemm_query(d)
    assert q[3] == 1 and q[1] == 64, q
    ops.gemm_run(d)
    torch.cuda.synchronize()
    got = from_nhwc(out.float().cpu().numpy(), B, H, W)
    check(f"conv3x3_fused_gn_B{B}_{H}x{W}_{Cin}_{Cout}_s{splitk}_r{rows}_skip{int(skip)}", got, ref, rel_l2=1.5e-3)
    if not skip:
        an = ops.groupnorm(xd, None, gd, btd, 1e-5, True)
        two = ops.gemm(an, wd, Cout, B, H, W, Cin, bias=bd, rowbias=rbd, rowbias_ld=Cout, ksize=3)
        check(f"conv3x3_fused_gn_vs_two_launches_{H}x{W}_{Cin}_r{rows}", out.reshape(-1, Cout), two, rel_l2=1e-3)


@pytest.mark.parametrize("B,HW,Cin,N,rows,splitk,tile_m,tile_n,extra", [
    (2, 1024, 640, 640, 128, 0, 0, 0, "stats"),      # SDv2 32 x 32 level at UNet batch 2: proj_in with LayerNorm row statistics out
    (2, 256, 1280, 1280, 64, 3, 0, 0, "stats"),      # 16 x 16 level: in-kernel split-K, statistics from 64-row reduce blocks
    (16, 1024, 640, 640, 128, 0, 0, 0, ""),          # batch 16: 128-row tiles
    (3, 576, 320, 320, 64, 0, 64, 64, ""),           # 24 x 24 tokens, 10 channels per group, forced 64 x 64 tiles
    (2, 1024, 640, 640, 32, 0, 128, 128, ""),        # 32-row statistics blocks (a fused SpatialTransformer tail's), 128 x 128 tiles
    (2, 1024, 192, 576, 128, 0, 0, 0, "qkv"),        # GLIDE AttentionBlock: norm -> qkv (6 channels per group), no bias folding
])
def test_dense_with_fused_input_groupnorm(ops, B, HW, Cin, N, rows, splitk, tile_m, tile_n, extra):
    """mdx_gemm_desc.gn_colstats on a DENSE launch: nn.GroupNorm(32) without an activation -> Dense / 1x1 conv
    (SpatialTransformer.norm -> proj_in, attention.py:243-247; GLIDE AttentionBlock.norm -> qkv) in ONE launch on the RAW input,
    the normalisation applied to the A fragments as a packed fp16 fma.  Against the fp32 reference and against the GroupNorm
    launch + GEMM launch it replaces.  Tolerance 1.5e-3: the per-channel scale / shift are fp16 (2^-11 each) on top of the fp16
    rounding of the normalised value that the two-launch form also has."""
    rng = np.random.RandomState(B + HW + Cin + N + rows)
    x = h16(rng.standard_normal((B, HW, Cin)) * (0.5 + rng.rand(Cin))[None, None, :] + rng.standard_normal(Cin)[None, None, :])
    g, bt = (1.0 + 0.2 * rng.standard_normal(Cin)).astype(np.float32), (0.1 * rng.standard_normal(Cin)).astype(np.float32)
    wt = h16(rng.standard_normal((N, Cin)) / math.sqrt(Cin))
    bv = rng.standard_normal(N).astype(np.float32)
    xn = O.group_norm(torch.tensor(x).permute(0, 2, 1).reshape(B, Cin, HW, 1), torch.tensor(g), torch.tensor(bt), 1e-6)
    ref = xn.reshape(B, Cin, HW).permute(0, 2, 1).reshape(B * HW, Cin) @ torch.tensor(wt).T + torch.tensor(bv)
    xd = dev16(x)
    nrb = HW // rows
    blk = xd.float().reshape(B * nrb, rows, Cin)
    cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
    out = torch.empty((B * HW, N), dtype=torch.float16, device=DEV)
    wd, bd, gd, btd = pack_dense(wt), dev32(bv), dev32(g), dev32(bt)
    st = torch.zeros((B * HW, N // 64, 2), dtype=torch.float32, device=DEV) if extra == "stats" else None
    d = ops.make_gemm_desc(xd, wd, N, B, HW, 1, Cin, out, N, bias=bd, splitk=splitk, tile_m=tile_m, tile_n=tile_n,
                           stats_out=st, gn_colstats=cs, gn_nrb=nrb, gn_gamma=gd, gn_beta=btd, gn_eps=1e-6, gn_silu=0)
    ws_ = ops.new_gemm_workspace(max(ops.gemm_workspace_bytes(d), 1 << 20), DEV)
    d.workspace, d.workspace_bytes = ws_.data_ptr(), ws_.numel() * 4
    q = ops.gemm_query(d)
    assert q[3] == 0 and HW % q[0] == 0, q
    out.fill_(float("nan"))
    ops.gemm_run(d)
    torch.cuda.synchronize()
    name = f"dense_fused_gn_B{B}_HW{HW}_{Cin}_{N}_r{rows}_s{q[2]}_t{q[0]}x{q[1]}"
    check(name, out, ref, rel_l2=1.5e-3)
    an = ops.groupnorm(xd, None, gd, btd, 1e-6, False)
    two = ops.gemm(an, wd, N, B, HW, 1, Cin, bias=bd)
    check(name + "_vs_two_launches", out, two, rel_l2=1.5e-3)
    if st is not None:      # the LayerNorm row statistics the launch emits are those of the values it stored
        o32 = out.float().reshape(B * HW, N // 64, 64)
        check(name + "_rowstats", st[:, :, 0], o32.sum(-1), rel_l2=1e-5)
    # replay determinism, and refusal where the form does not apply (SiLU requested / an M tile that would straddle two samples)
    out2 = torch.empty_like(out)
    d.out = out2.data_ptr()
    ops.gemm_run(d)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    from minddiffusion_amd._lib import MdxError
    bad = ops.make_gemm_desc(xd, wd, N, B, HW, 1, Cin, out, N, gn_colstats=cs, gn_nrb=nrb, gn_gamma=gd, gn_beta=btd, gn_silu=1)
    with pytest.raises(MdxError):
        ops.gemm_run(bad)
    if HW % 128:
        bad = ops.make_gemm_desc(xd, wd, N, B, HW, 1, Cin, out, N, tile_m=128, gn_colstats=cs, gn_nrb=nrb, gn_gamma=gd,
                                 gn_beta=btd, gn_silu=0)
        with pytest.raises(MdxError):
            ops.gemm_run(bad)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(8, 8, 8, 1280, 1280), (8, 8, 8, 2560, 1280), (32, 4, 4, 1280, 1280)])
def test_conv3x3_shape_that_shares_M_with_a_256_row_tuned_row(ops, B, H, W, Cin, Cout):
    """Round-5 regression: the tile table is keyed by (M, N, K, ksize).  The 8 x 8 level at UNet batch 8 has M = 512, N = 1280,
    K = 11520 / 23040 -- the key of rows measured at batch 2 on the 16 x 16 level with 256-row tiles (16 x 16-patch HALO kernel).
    That kernel does not apply to 8 x 8 images; the row used to be taken anyway and the generic kernel ran 128-row tiles on a grid
    sized for 256-row ones: samples 4-7 were never written.  Every output row is checked against the fp32 conv."""
    rng = np.random.RandomState(B + Cin)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    wt = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    ref = O.conv2d(torch.tensor(x), torch.tensor(wt), torch.tensor(bv))
    xd, wd, bd = dev16(nhwc(x)), pack_conv(wt), dev32(bv)
    out = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device=DEV)
    d = ops.make_gemm_desc(xd, wd, Cout, B, H, W, Cin, out, Cout, bias=bd, ksize=3)
    ws_ = ops.new_gemm_workspace(max(ops.gemm_workspace_bytes(d), 1 << 20), DEV)
    d.workspace, d.workspace_bytes = ws_.data_ptr(), ws_.numel() * 4
    q = ops.gemm_query(d)
    assert q[0] in (64, 128), q      # never the 256-row form: these images do not tile into 16 x 16 patches
    ops.gemm_run(d)
    torch.cuda.synchronize()
    got = from_nhwc(out.float().cpu().numpy(), B, H, W)
    assert np.isfinite(got).all()
    check(f"conv3x3_M{B * H * W}_{H}x{W}_{Cin}_{Cout}_not_256_rows", got, ref, rel_l2=1e-3)
    for b in (0, B // 2, B - 1):      # the failure left whole samples stale: per-sample check
        check(f"conv3x3_M{B * H * W}_{H}x{W}_{Cin}_sample{b}", got[b:b + 1], ref[b:b + 1], rel_l2=1e-3)
    from minddiffusion_amd._lib import MdxError
    bad = ops.make_gemm_desc(xd, wd, Cout, B, H, W, Cin, out, Cout, bias=bd, ksize=3, tile_m=256)
    bad.workspace, bad.workspace_bytes = d.workspace, d.workspace_bytes
    qb = ops.gemm_query(bad)
    assert qb[0] != 256, qb          # a forced 256-row tile on a launch it does not apply to is not honoured


# --------------------------------------------------------------------------- round 5: the "latency diet" options move loads, not arithmetic
@pytest.mark.parametrize("opt", ["gn_prefetch", "gemm_dense_issue", "gemm_ln_prefetch", "attn_fast_stage"])
def test_round5_issue_order_options_do_not_change_a_bit(ops, opt):
    """gn_prefetch / gemm_dense_issue / gemm_ln_prefetch / attn_fast_stage (include/mdx.h) change WHEN a kernel requests its
    operands and parameters -- scalar-offset DMA issue, parameters ahead of the statistics, touches ahead of the K loop -- never
    the order of a sum: the outputs of the kernels they touch must be bit-identical with the option on and off."""
    rng = np.random.RandomState(11)

    def run_all():
        outs = []
        # GroupNorm: apply form with two sources, one-launch form (small tensor), FiLM-free
        for (B, H, W, C1, C2) in [(2, 16, 16, 640, 320), (2, 8, 8, 1280, 0), (1, 64, 64, 320, 0)]:
            C = C1 + C2
            x = h16(np.random.RandomState(C + H).standard_normal((B, C, H, W)) * 1.5 + 0.3)
            g = np.random.RandomState(1).standard_normal(C).astype(np.float32)
            b = np.random.RandomState(2).standard_normal(C).astype(np.float32)
            x1 = dev16(nhwc(x[:, :C1]))
            x2 = dev16(nhwc(x[:, C1:])) if C2 else None
            outs.append(ops.groupnorm(x1, x2, dev32(g), dev32(b), 1e-5, True).clone())
        # dense GEMMs (64 x 64 / 128 x 64 / 128 x 128 tiles through the tile rule), bias + residual, a split-K one
        for (M, N, K, sk) in [(512, 1280, 1280, 1), (2048, 640, 640, 1), (4096, 320, 1280, 1), (128, 1280, 2560, 4)]:
            r = np.random.RandomState(M + N)
            a = h16(r.standard_normal((M, K)))
            w = h16(r.standard_normal((N, K)) / math.sqrt(K))
            bv = r.standard_normal(N).astype(np.float32)
            res = h16(r.standard_normal((M, N)))
            outs.append(ops.gemm(dev16(a), pack_dense(w), N, 1, M, 1, K, bias=dev32(bv), residual=dev16(res), residual_ld=N,
                                 splitk=sk).clone())
        # LayerNorm fold: producer with row statistics -> plain consumer
        M, C, K0 = 512, 640, 256
        r = np.random.RandomState(5)
        a0, w0 = h16(r.standard_normal((M, K0))), h16(r.standard_normal((C, K0)) / math.sqrt(K0))
        g = (1 + 0.3 * r.standard_normal(C)).astype(np.float32)
        be = (0.3 * r.standard_normal(C)).astype(np.float32)
        w1, b1 = h16(r.standard_normal((C, C)) / math.sqrt(C)), r.standard_normal(C).astype(np.float32)
        x = torch.empty((M, C), dtype=torch.float16, device=DEV)
        stats = torch.zeros((M, C // 64, 2), dtype=torch.float32, device=DEV)
        d0 = ops.make_gemm_desc(dev16(a0), pack_dense(w0), C, 1, M, 1, K0, x, C, stats_out=stats)
        ws0 = ops.new_gemm_workspace(ops.gemm_workspace_bytes(d0), DEV)
        d0.workspace, d0.workspace_bytes = ws0.data_ptr(), ws0.numel() * 4
        ops.gemm_run(d0)
        wg, s, cb = ops.fold_layernorm(dev16(w1), dev32(g), dev32(be), dev32(b1))
        out = torch.empty((M, C), dtype=torch.float16, device=DEV)
        d1 = ops.make_gemm_desc(x, ops.pack_gemm_weight(wg), C, 1, M, 1, C, out, C, bias=cb, ln_stats=stats, ln_s=s)
        ws1 = ops.new_gemm_workspace(ops.gemm_workspace_bytes(d1), DEV)
        d1.workspace, d1.workspace_bytes = ws1.data_ptr(), ws1.numel() * 4
        ops.gemm_run(d1)
        outs.append(out.clone())
        # attention at the head dims that take the scalar-offset tile issue (40, 80) and one that does not (64); ragged key tail
        for (B, heads, Nq, Nk, D) in [(2, 8, 256, 256, 40), (1, 8, 128, 200, 80), (1, 2, 256, 320, 64)]:
            Cc = heads * D
            r = np.random.RandomState(Nq + D)
            q, k, v = (h16(r.standard_normal((B, n, Cc))) for n in (Nq, Nk, Nk))
            ld = (Nk + 7) // 8 * 8
            vt = np.zeros((B, Cc, ld), np.float32)
            vt[:, :, :Nk] = v.transpose(0, 2, 1)
            qd, kd, vtd = dev16(q), dev16(k), dev16(vt)
            o = torch.empty((B, Nq, Cc), dtype=torch.float16, device=DEV)
            ops.attention(qd.data_ptr(), kd.data_ptr(), vtd.data_ptr(), o.data_ptr(), B, heads, D, Nq, Nk, D ** -0.5,
                          Nq * Cc, Cc, Nk * Cc, Cc, Cc * ld, ld, Nq * Cc, Cc)
            outs.append(o.clone())
        torch.cuda.synchronize()
        return outs

    default = ops.get_option(opt)
    try:
        ops.set_option(opt, 1)
        on = run_all()
        ops.set_option(opt, 0)
        off = run_all()
    finally:
        ops.set_option(opt, default)
    assert len(on) == len(off)
    for i, (a, b) in enumerate(zip(on, off)):
        assert torch.equal(a, b), f"output {i} differs between {opt} = 1 and 0"
    assert all(torch.isfinite(a.float()).all() for a in on)
