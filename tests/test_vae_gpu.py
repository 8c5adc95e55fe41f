"""GPU parity of the VAE decode path: AutoencoderKL.decode on the HIP kernels vs the fp32 CPU
oracle (oracle/vae.py) on identical seeded weights and latents, plus its two helper kernels.

Tolerances (fp16 storage + fp32 accumulation vs all-fp32): helper kernels bit-exact / rel-L2 <= 1e-3; decoded image
rel-L2 <= 5e-3 (tiny) and <= 1e-2 (full 512x512 decoder: 30 layers deep, one materialised fp16 attention).
"""
import os

import numpy as np
import pytest
import torch

from _util import check
from oracle import vae as OV

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pack_b_operand_matches_weight_packing():
    """The device re-layout must produce bit for bit the image ops.pack_gemm_weight builds for weights."""
    from minddiffusion_amd import ops
    rng = np.random.RandomState(3)
    for rows, K, ld in ((64, 64, 64), (200, 136, 136), (512, 4096, 4096), (100, 72, 80)):
        src = torch.tensor(rng.standard_normal((rows, ld)), dtype=torch.float16, device=DEV)
        got = ops.pack_b_operand(src[:, :K])
        ref = ops.pack_gemm_weight(src[:, :K].contiguous())
        assert torch.equal(got, ref), (rows, K, ld)


@pytest.mark.parametrize("rows,cols,scale", [(7, 64, 1.0), (33, 4096, 512 ** -0.5), (5, 2056, 0.3), (3, 9216, 0.05)])
def test_softmax_rows(rows, cols, scale):
    from minddiffusion_amd import ops
    rng = np.random.RandomState(rows + cols)
    x = (rng.standard_normal((rows, cols)) * 20).astype(np.float16)
    x[0, 3] = 300.0   # a dominant score: exercises the max subtraction
    ref = torch.softmax(torch.tensor(x.astype(np.float32)) * scale, dim=1)
    got = ops.softmax_rows(torch.tensor(x, device=DEV), scale)
    check(f"softmax_rows_{rows}x{cols}", got, ref, rel_l2=1e-3, max_abs=1e-3)
    assert abs(float(got.float().sum(1).mean()) - 1.0) < 2e-3


def _build(dd, params, graph=True):
    from minddiffusion_amd.ldm.models.autoencoder import AutoencoderKL
    vae = AutoencoderKL(ddconfig=dd, embed_dim=4, device=DEV, use_graph=graph)
    vae.load_state_dict(params)
    return vae


@pytest.mark.parametrize("graph", [False, True])
def test_tiny_decoder(graph):
    from minddiffusion_amd.configs import TINY_VAE_DDCONFIG
    dd = dict(TINY_VAE_DDCONFIG)
    params = OV.init_params(dd, seed=5)
    vae = _build(dd, params, graph)
    for (B, h, w) in ((2, 16, 16), (1, 8, 24), (3, 16, 32)):
        z = np.random.RandomState(B + h).randn(B, 4, h, w).astype(np.float32)
        ref = OV.decode(params, z, dd)
        got = vae.decode(torch.tensor(z, device=DEV))
        assert tuple(got.shape) == (B, 3, 2 * h, 2 * w)
        check(f"tiny_vae_decode_graph{int(graph)}_B{B}_{h}x{w}", got, ref, rel_l2=5e-3, max_abs=5e-2)
        got2 = vae.decode(torch.tensor(z, device=DEV)).clone()
        assert torch.equal(got, got2)


def test_asymmetric_pad_stride2_conv(ops=None):
    """mdx_gemm_desc.asym_pad: nn.Pad((0,1),(0,1)) + valid 3x3 stride-2 conv (Encoder Downsample, model.py:55-78)."""
    import math
    from minddiffusion_amd import ops
    rng = np.random.RandomState(12)
    for (B, H, W, C, N) in ((2, 16, 16, 64, 64), (1, 32, 48, 128, 128), (1, 8, 8, 64, 72)):
        x = rng.standard_normal((B, C, H, W)).astype(np.float16)
        w = (rng.standard_normal((N, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float16)
        bv = rng.standard_normal(N).astype(np.float32)
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(torch.tensor(x).float(), (0, 1, 0, 1)),
                                         torch.tensor(w).float(), torch.tensor(bv), stride=2)
        xd = torch.tensor(x, device=DEV).permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous()
        out = torch.empty((B, (H // 2) * (W // 2), N), dtype=torch.float16, device=DEV)
        d = ops.make_gemm_desc(xd, ops.pack_conv_weight(torch.tensor(w, device=DEV)), N, B, H, W, C, out, N,
                               bias=torch.tensor(bv, device=DEV), ksize=3, stride=2, asym_pad=1, splitk=1)
        ops.gemm_run(d)
        got = out.float().cpu().reshape(B, H // 2, W // 2, N).permute(0, 3, 1, 2)
        check(f"conv3x3_s2_asympad_{H}x{W}_{C}to{N}", got, ref, rel_l2=1e-3)


@pytest.mark.parametrize("graph", [False, True])
def test_tiny_encoder(graph):
    from minddiffusion_amd.configs import TINY_VAE_DDCONFIG
    dd = dict(TINY_VAE_DDCONFIG)
    params = OV.init_params(dd, seed=8)
    vae = _build(dd, params, graph)
    for (B, H, W) in ((2, 32, 32), (1, 16, 48)):
        rng = np.random.RandomState(B + H)
        x = rng.randn(B, 3, H, W).astype(np.float32)
        noise = rng.randn(B, 4, H // 2, W // 2).astype(np.float32)
        ref = OV.encode(params, x, noise, dd)
        got = vae.encode(torch.tensor(x, device=DEV), noise=torch.tensor(noise, device=DEV))
        assert tuple(got.shape) == (B, 4, H // 2, W // 2)
        check(f"tiny_vae_encode_graph{int(graph)}_B{B}_{H}x{W}", got, ref, rel_l2=5e-3, max_abs=5e-2)
        mode = vae.encode(torch.tensor(x, device=DEV), sample=False)
        check(f"tiny_vae_encode_mode_graph{int(graph)}_B{B}_{H}x{W}", mode, OV.encode(params, x, None, dd), rel_l2=5e-3, max_abs=5e-2)


def test_full_sd_vae_encoder_256():
    """The shipped SD VAE encoder (34 M parameters) on a 256x256 image -> 32x32 latent moments (mode compared: the
    sampled form only adds exp(0.5 logvar) * noise)."""
    from minddiffusion_amd.configs import SD_VAE_DDCONFIG
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    dd = dict(SD_VAE_DDCONFIG)
    params = OV.init_params(dd, seed=9)
    vae = _build(dd, params)
    x = np.clip(np.random.RandomState(3).randn(1, 3, 256, 256) * 0.5, -1, 1).astype(np.float32)
    ref = OV.encode(params, x, None, dd)
    got = vae.encode(torch.tensor(x, device=DEV), sample=False)
    assert tuple(got.shape) == (1, 4, 32, 32)
    check("sd_vae_encode_256", got, ref, rel_l2=1e-2, max_abs=2e-1)


def test_decode_first_stage_scaling():
    """LatentDiffusion.decode_first_stage = decode(z / scale_factor) (ddpm.py:286-288)."""
    from minddiffusion_amd.configs import TINY_UNET, TINY_VAE_DDCONFIG, SD2_LDM
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    dd = dict(TINY_VAE_DDCONFIG)
    params = OV.init_params(dd, seed=6)
    model = LatentDiffusion(unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                                         "params": dict(TINY_UNET)}, **SD2_LDM)
    model.first_stage_model = _build(dd, params)
    z = np.random.RandomState(1).randn(2, 4, 16, 16).astype(np.float32)
    ref = OV.decode(params, z / SD2_LDM["scale_factor"], dd)
    got = model.decode_first_stage(torch.tensor(z, device=DEV))
    check("decode_first_stage", got, ref, rel_l2=5e-3, max_abs=1e-1)


def test_full_sd_vae_decoder_512():
    """The shipped SD VAE decoder (49.5 M parameters, ch 128 x (1,2,4,4)) on one 64x64 latent -> 512x512 image:
    1.24 TFLOP, single-head 4096 x 4096 x 512 attention in the middle."""
    from minddiffusion_amd.configs import SD_VAE_DDCONFIG
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    dd = dict(SD_VAE_DDCONFIG)
    params = OV.init_params(dd, seed=7)
    vae = _build(dd, params)
    z = np.random.RandomState(2).randn(1, 4, 64, 64).astype(np.float32)
    ref = OV.decode(params, z, dd)
    got = vae.decode(torch.tensor(z, device=DEV))
    assert tuple(got.shape) == (1, 3, 512, 512)
    check("sd_vae_decode_512", got, ref, rel_l2=1e-2, max_abs=2e-1)
