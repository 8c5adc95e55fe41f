"""Parity of the eight-wave 256-pixel conv core (minddiffusion_amd/csrc/conv8p.hip) -- the launch form of nn.Conv2d 3x3 / stride 1 /
pad 1 (openaimodel.py:136-138, 159-163, 174, 201-205) for M >= 8192 output pixels -- through the C-ABI against the fp32 oracle and
against the 128-row HALO kernel it replaces (which tests/test_kernels_gpu.py pins against the oracle).

The core is a new SYNCHRONISATION structure (two wave groups half a phase apart, DMA batches in flight across barriers), so beside
the value checks there is a race screen: the benchmarked shapes run many times on a poisoned output and must reproduce bit for bit.
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


def nhwc(x):
    b, c, h, w = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(b, h * w, c))


def from_nhwc(y, b, h, w):
    return y.reshape(b, h, w, -1).transpose(0, 3, 1, 2)


def pack_conv(w):
    from minddiffusion_amd import ops as _ops
    return _ops.pack_conv_weight(torch.from_numpy(np.ascontiguousarray(w)).to(DEV))


def test_mfma_16x16x32_layout(ops):
    """Pins the fragment maps conv8p assumes for v_mfma_f32_16x16x32_f16 (cdna guide section 3): A lane l: A[l & 15][8 (l >> 4) + j];
    B lane l: B[8 (l >> 4) + j][l & 15]; C lane l register r: C[4 (l >> 4) + r][l & 15]."""
    rng = np.random.RandomState(0)
    A = h16(rng.standard_normal((16, 32)))
    B = h16(rng.standard_normal((32, 16)))      # asymmetric on purpose
    af = np.zeros((64, 8), np.float32)
    bf = np.zeros((64, 8), np.float32)
    for l in range(64):
        for j in range(8):
            af[l, j] = A[l & 15, 8 * (l >> 4) + j]
            bf[l, j] = B[8 * (l >> 4) + j, l & 15]
    c = ops.probe_mfma16(dev16(af), dev16(bf)).cpu().numpy()
    C = np.zeros((16, 16), np.float32)
    for l in range(64):
        for r in range(4):
            C[4 * (l >> 4) + r, l & 15] = c[l, r]
    check("mfma_16x16x32_layout", C, A @ B, rel_l2=1e-6)


@pytest.mark.parametrize("B,H,W,Cin,Cout,tile_n", [
    (1, 16, 16, 64, 160, 0),        # one patch, one chunk, one N tile: image border on every side
    (1, 16, 16, 64, 128, 0),
    (1, 16, 16, 128, 192, 0),
    (2, 32, 16, 128, 320, 0),       # 2 x 1 patch grid, two samples, two N tiles of 160
    (1, 32, 48, 192, 72, 0),        # 2 x 3 patch grid, N tail (72 of 128), three chunks (halo double buffer wraps)
    (1, 48, 48, 320, 320, 160),     # the 768-pixel level-1 geometry (3 x 3 patches), five chunks
    (2, 16, 32, 64, 640, 128),      # forced 128-column tiles, five N tiles
    (1, 64, 64, 64, 64, 0),         # 16 patches: interior + all borders; N = 64 (half an N tile)
    (1, 16, 32, 128, 192, 96),      # 96-column tiles
    (2, 16, 16, 64, 136, 64),       # 64-column tiles, N tail
])
def test_conv8p_vs_oracle(ops, B, H, W, Cin, Cout, tile_n):
    rng = np.random.RandomState(Cin + Cout + H + W)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    ref = O.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bv), stride=1, padding=1)
    xd, wp, bd = dev16(nhwc(x)), pack_conv(w), dev32(bv)
    out = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device=DEV)
    d = ops.make_gemm_desc(xd, wp, Cout, B, H, W, Cin, out, Cout, bias=bd, ksize=3, tile_m=256, stages=8, tile_n=tile_n)
    q = ops.gemm_query(d)
    assert q[0] == 256 and q[2] == 1 and q[3] == 1 and q[5] == 256 and q[1] in (64, 96, 128, 160, 192), q
    ops.gemm_run(d)
    torch.cuda.synchronize()
    check(f"conv8p_B{B}_{H}x{W}_{Cin}to{Cout}_bn{q[1]}", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)
    # the kernel it replaces (128-row HALO tiles; same k order per output up to the 16- vs 32-wide MFMA grouping)
    old = ops.gemm(xd, wp, Cout, B, H, W, Cin, bias=bd, ksize=3, tile_m=128)
    check(f"conv8p_vs_halo128_B{B}_{H}x{W}_{Cin}to{Cout}", out, old, rel_l2=5e-4)


def test_conv8p_two_source_rowbias_residual_colstats(ops):
    """conv1 of an up-path ResBlock reading the virtual concat (two sources, c1 = 128, c2 = 64), + bias + per-sample time-embedding
    row + residual, and the GroupNorm column statistics of each 256-pixel patch."""
    rng = np.random.RandomState(5)
    B, H, W, C1, C2, N = 3, 16, 32, 128, 64, 160
    x1 = h16(rng.standard_normal((B, C1, H, W)))
    x2 = h16(rng.standard_normal((B, C2, H, W)))
    w = h16(rng.standard_normal((N, C1 + C2, 3, 3)) / math.sqrt(9 * (C1 + C2)))
    bv = rng.standard_normal(N).astype(np.float32)
    emb = rng.standard_normal((B, 400)).astype(np.float32)
    res = h16(rng.standard_normal((B, N, H, W)))
    ref = O.conv2d(torch.tensor(np.concatenate([x1, x2], 1)), torch.tensor(w), torch.tensor(bv)) \
        + torch.tensor(emb[:, 8:8 + N])[:, :, None, None] + torch.tensor(res)
    embd = dev32(emb)
    nrb = B * (H // 16) * (W // 16)
    cs = torch.full((nrb, N, 2), float("nan"), dtype=torch.float32, device=DEV)
    out = torch.empty((B * H * W, N), dtype=torch.float16, device=DEV)
    keep = (dev16(nhwc(x1)), dev16(nhwc(x2)), pack_conv(w), dev32(bv), dev16(nhwc(res)))
    d = ops.make_gemm_desc(keep[0], keep[2], N, B, H, W, C1, out, N, a2=keep[1], c2=C2, bias=keep[3], ksize=3,
                           rowbias=embd[:, 8:8 + N], rowbias_ld=400, residual=keep[4], residual_ld=N, colstats_out=cs,
                           tile_m=256, stages=8)
    assert ops.gemm_query(d)[5] == 256
    ops.gemm_run(d)
    torch.cuda.synchronize()
    check("conv8p_two_source_rowbias_residual", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)
    # statistics of the fp16 values stored, per 16 x 16 patch (sample-major patch order) and column
    o = out.float().reshape(B, H // 16, 16, W // 16, 16, N).permute(0, 1, 3, 2, 4, 5).reshape(nrb, 256, N)
    check("conv8p_colstats_sum", cs[:, :, 0], o.sum(1), rel_l2=1e-5)
    check("conv8p_colstats_sumsq", cs[:, :, 1], (o * o).sum(1), rel_l2=1e-5)


@pytest.mark.parametrize("B,H,W,C,Cs1,Cs2", [(2, 16, 16, 128, 64, 0), (1, 32, 32, 320, 320, 320), (2, 16, 32, 160 * 2, 192, 64)])
def test_conv8p_fused_skip_connection(ops, B, H, W, C, Cs1, Cs2):
    """mdx_gemm_desc.skip_w on the eight-wave core: out = conv3x3(h) + conv1x1(cat(x, x2)) (+ biases) in one launch."""
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
    bsum = dev32(b3 + b1)
    d = ops.make_gemm_desc(hd, w3p, C, B, H, W, C, out, C, bias=bsum, ksize=3, skip_a=x1d, skip_a2=x2d, skip_c1=Cs1, skip_c2=Cs2,
                           skip_w=w1p, tile_m=256, stages=8)
    assert ops.gemm_query(d)[0] == 256
    ops.gemm_run(d)
    torch.cuda.synchronize()
    check(f"conv8p_fused_skip_B{B}_{H}x{W}_{C}_{Cs1}+{Cs2}", from_nhwc(out.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3)


@pytest.mark.parametrize("B,H,Cin,Cout", [(16, 64, 320, 320), (16, 32, 640, 640), (8, 96, 320, 320)])
def test_conv8p_benchmarked_shapes_race_screen(ops, B, H, Cin, Cout):
    """The shapes BASELINE configs 2 / 3 run through this core by DEFAULT (no override fields: M >= 8192), 24 launches each on a
    NaN-poisoned output with the weights rotating through cold copies: every launch must reproduce the first bit for bit (a read
    ahead of its DMA, or a refill ahead of a read, shows as a changed bit) and agree with the 128-row HALO kernel."""
    rng = np.random.RandomState(B + H + Cin)
    x = dev16(h16(rng.standard_normal((B, H * H, Cin))))
    ws = [pack_conv(h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))) for _ in range(3)]
    bd = dev32(rng.standard_normal(Cout).astype(np.float32))
    outs = [torch.empty((B * H * H, Cout), dtype=torch.float16, device=DEV) for _ in range(3)]
    descs = [ops.make_gemm_desc(x, ws[i], Cout, B, H, H, Cin, outs[i], Cout, bias=bd, ksize=3) for i in range(3)]
    q = ops.gemm_query(descs[0])
    assert q[0] == 256 and q[1] == 160 and q[2] == 1, q          # the default route at these sizes
    firsts = []
    for i in range(3):
        outs[i].fill_(float("nan"))
        ops.gemm_run(descs[i])
        firsts.append(outs[i].clone())
    for rep in range(24):
        i = rep % 3
        outs[i].fill_(float("nan"))
        ops.gemm_run(descs[i])
        assert torch.equal(outs[i], firsts[i]), f"launch {rep}: output changed"
    old = ops.gemm(x, ws[0], Cout, B, H, H, Cin, bias=bd, ksize=3, tile_m=128)
    check(f"conv8p_default_route_vs_halo128_B{B}_{H}x{H}_{Cin}to{Cout}", firsts[0], old, rel_l2=5e-4)


@pytest.mark.parametrize("B,H,W,Cin,Cout,tile_n,skip", [
    (9, 48, 48, 320, 320, 0, 0),       # 81 patches x 2 N tiles = 162 tiles: all of them in the tail, split 1 way (162 > 128) .. see plan
    (2, 64, 64, 320, 320, 160, 0),     # 32 x 2 = 64 tiles: no whole round, every tile split 4 ways (5 chunks -> 2 + 2 + 1 ... 3 splits)
    (8, 48, 48, 640, 640, 160, 0),     # 72 x 4 = 288 tiles (BASELINE config 3, level 1): 256 whole + 32 split 4 ways
    (8, 48, 48, 640, 640, 160, 320),   # ... with the fused skip tiles riding on the last split
    (1, 16, 16, 512, 160, 160, 0),     # ONE tile split 4 ways (8 chunks)
    (3, 16, 16, 128, 160, 160, 64),    # 3 tiles x 2 splits of one chunk each + skip
])
def test_conv8p_tail_split(ops, B, H, W, Cin, Cout, tile_n, skip):
    """Tile counts that do not fill whole rounds of 256 CUs: the tiles of the last round are split along the 64-channel chunks and
    reduced by the last-arriving block (fp32 partials in the workspace, library-owned tickets).  Against the 128-row HALO kernel
    and, run twice more on a NaN-poisoned workspace, bit-stable."""
    rng = np.random.RandomState(B + H + Cin + skip)
    x = dev16(h16(rng.standard_normal((B, H * W, Cin))))
    w = pack_conv(h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)))
    bd = dev32(rng.standard_normal(Cout).astype(np.float32))
    kw = {}
    if skip:
        xs = dev16(h16(rng.standard_normal((B, H * W, skip))))
        wsk = pack_conv(h16(rng.standard_normal((Cout, skip, 1, 1)) / math.sqrt(skip)))
        kw = dict(skip_a=xs, skip_c1=skip, skip_w=wsk)
    nrb = B * (H // 16) * (W // 16)
    cs = torch.zeros((nrb, Cout, 2), dtype=torch.float32, device=DEV)
    out = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device=DEV)
    d = ops.make_gemm_desc(x, w, Cout, B, H, W, Cin, out, Cout, bias=bd, ksize=3, tile_m=256, stages=8, tile_n=tile_n,
                           colstats_out=cs, **kw)
    need = ops.gemm_workspace_bytes(d)
    ws = ops.new_gemm_workspace(max(need, 1 << 16), DEV)
    ws.fill_(float("nan"))
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert ops.gemm_query(d)[0] == 256
    ops.gemm_run(d)
    torch.cuda.synchronize()
    first, cs_first = out.clone(), cs.clone()
    ops.set_option("gemm_conv8p", 0)
    try:
        old = ops.gemm(x, w, Cout, B, H, W, Cin, bias=bd, ksize=3, **kw)
    finally:
        ops.set_option("gemm_conv8p", 1)
    check(f"conv8p_tail_split_B{B}_{H}x{W}_{Cin}to{Cout}_skip{skip}_need{need >> 10}K", first, old, rel_l2=5e-4)
    o = first.float().reshape(B, H // 16, 16, W // 16, 16, Cout).permute(0, 1, 3, 2, 4, 5).reshape(nrb, 256, Cout)
    check(f"conv8p_tail_split_colstats_B{B}_{Cin}", cs_first[:, :, 0], o.sum(1), rel_l2=1e-5)
    for _ in range(3):
        ws.fill_(float("nan"))
        out.fill_(float("nan"))
        ops.gemm_run(d)
        assert torch.equal(out, first) and torch.equal(cs, cs_first)


def test_conv8p_phase_per_kstep_form_is_bit_identical(ops):
    """stages = 9: one phase per 32-deep k-step (the form the 192-column tile always uses) -- same products in the same order."""
    rng = np.random.RandomState(3)
    B, H, W, Cin, Cout = 2, 32, 32, 192, 320
    x = dev16(h16(rng.standard_normal((B, H * W, Cin))))
    w = pack_conv(h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)))
    a = ops.gemm(x, w, Cout, B, H, W, Cin, ksize=3, tile_m=256, stages=8, tile_n=160)
    b = ops.gemm(x, w, Cout, B, H, W, Cin, ksize=3, tile_m=256, stages=9, tile_n=160)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W,Cin,Cout,cs", [(1, 16, 16, 64, 64, False), (2, 32, 16, 128, 320, True), (2, 32, 32, 640, 640, True),
                                                (8, 48, 48, 128, 128, False), (2, 32, 32, 192, 192, True), (2, 16, 32, 128, 384, False),
                                                (1, 32, 32, 192, 576, False)])
def test_conv8p_subpixel_upsample_conv(ops, B, H, W, Cin, Cout, cs):
    """mdx_gemm_desc.w_sub: nearest-2x + conv3x3 (Upsample.construct, openaimodel.py:57-60) as four 2 x 2 convs of the low-resolution
    tensor with pre-summed taps (one fp16 rounding of the summed weights instead of separate products: within the usual 1e-3 of the
    fp32 oracle), pixel-shuffled by the epilogue; against the oracle and against the upsampling-gather form it replaces; column
    statistics of the interleaved row blocks."""
    rng = np.random.RandomState(B + H + Cin + Cout)
    x = h16(rng.standard_normal((B, Cin, H, W)))
    w = h16(rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin))
    bv = rng.standard_normal(Cout).astype(np.float32)
    ref = O.conv2d(O.upsample_nearest2x(torch.tensor(x)), torch.tensor(w), torch.tensor(bv), stride=1, padding=1)
    xd, wp, bd = dev16(nhwc(x)), pack_conv(w), dev32(bv)
    wsub = ops.pack_subpixel_conv_weight(torch.tensor(w).to(DEV))
    M = B * 4 * H * W
    out = torch.full((M, Cout), float("nan"), dtype=torch.float16, device=DEV)
    cst = torch.full((M // 256, Cout, 2), float("nan"), dtype=torch.float32, device=DEV) if cs else None
    d = ops.make_gemm_desc(xd, wp, Cout, B, H, W, Cin, out, Cout, bias=bd, ksize=3, upsample=1, w_sub=wsub, tile_m=256, stages=8,
                           colstats_out=cst)
    need = ops.gemm_workspace_bytes(d)
    ws = ops.new_gemm_workspace(max(need, 1 << 16), DEV)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    q = ops.gemm_query(d)
    assert q[0] == 256 and q[3] == 1, q
    ops.gemm_run(d)
    torch.cuda.synchronize()
    check(f"conv8p_subpixel_B{B}_{H}x{W}_{Cin}to{Cout}", from_nhwc(out.float().cpu().numpy(), B, 2 * H, 2 * W), ref, rel_l2=1e-3)
    old = ops.gemm(xd, wp, Cout, B, H, W, Cin, bias=bd, ksize=3, upsample=1)
    check(f"conv8p_subpixel_vs_gather_form_B{B}_{H}x{W}_{Cin}to{Cout}", out, old, rel_l2=1e-3)
    if cs:      # per sample the row blocks partition the output pixels: the per-sample column sums must match
        nb = M // 256 // B
        o = out.float().reshape(B, 4 * H * W, Cout)
        check("conv8p_subpixel_colstats_sum", cst[:, :, 0].reshape(B, nb, Cout).sum(1), o.sum(1), rel_l2=1e-5)
        check("conv8p_subpixel_colstats_sumsq", cst[:, :, 1].reshape(B, nb, Cout).sum(1), (o * o).sum(1), rel_l2=1e-5)
