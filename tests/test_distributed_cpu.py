"""world_size-2 gloo tests (CPU) of the batch-sharding path: one broadcast before the loop, independent
shards, optional gather -- sharded results must equal the single-process results bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeSampler:
    """Stands in for PLMSSampler/DDIMSampler on the CPU: a deterministic per-sample function of (c, uc, x_T)."""

    def sample(self, S, conditioning, batch_size, shape, x_T=None, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, **kw):
        c = conditioning.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
        u = unconditional_conditioning.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
        return x_T * 0.5 + c + unconditional_guidance_scale * (c - u) + S, {}


class _FakeUnet:
    device = torch.device("cpu")
    context_dim = 8
    max_context_len = 6      # > T = 5: the payload is sized for the UNet's maximum, the real T travels in the header


class _FakeModel:
    unet = _FakeUnet()


def _inputs(B=4, T=5, Dc=8, h=4, w=6):
    rng = np.random.RandomState(0)
    c = torch.from_numpy(rng.randn(B, T, Dc).astype(np.float16))
    uc = torch.from_numpy(rng.randn(1, T, Dc).astype(np.float16))
    x_T = torch.from_numpy(rng.randn(B, 4, h, w).astype(np.float32))
    return c, uc, x_T


class _Dev:
    device = torch.device("cpu")


class _FakeGlideBase:
    """Stands in for GenerativePSampleDiffusionModel on the CPU: per-image deterministic in (x, prompt, uncond prompt, noise)."""
    model = _Dev()
    num_timesteps = 5
    shape = (4, 3, 4, 4)

    def __init__(self, pics):
        self.pics_generated = pics
        self.shape = (2 * pics, 3, 4, 4)

    def __call__(self, x, timesteps, token, mask, random_token=None, random_mask=None, noise=None):
        P = self.pics_generated
        t = float(timesteps.reshape(-1)[0])
        cond = token[:P].float().mean(1).reshape(-1, 1, 1, 1) * 1e-2
        unc = float(np.asarray(random_token, dtype=np.float64).mean()) * 1e-3
        s = 0.9 * x[:P] + cond - unc + (0.1 * noise[:P] if t > 0 else 0.0)
        return torch.cat([s, s], 0), None


class _FakeGlideUp:
    model = _Dev()
    num_timesteps = 3

    def __init__(self, pics):
        self.shape = (pics, 3, 8, 8)

    def __call__(self, x, timesteps, token, mask, samples):
        low = samples.float().mean(dim=(1, 2, 3)).reshape(-1, 1, 1, 1)
        return 0.8 * x + low + token.float().mean(1).reshape(-1, 1, 1, 1) * 1e-2, None


def _glide_prompts(P, T=6):
    rng = np.random.RandomState(3)
    return rng.randint(1, 100, (P, T)).astype(np.int32), np.ones((P, T), np.int32)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from minddiffusion_amd import distributed as D
    from minddiffusion_amd.pipeline import DiffusionPipeline
    r, n, _ = D.init_from_env(backend="gloo")
    assert (r, n) == (rank, world)
    c, uc, x_T = _inputs()
    # 1) the packed broadcast hands every rank its shard; the fp32 noise survives bit-exactly.  Count the collectives.
    calls = []
    real_bcast = dist.broadcast
    dist.broadcast = lambda *a, **k: (calls.append(1), real_bcast(*a, **k))[1]
    cpu = torch.device("cpu")
    r0 = lambda t: t if rank == 0 else None
    cs, ucs, xs = D.broadcast_conditioning(r0(c), r0(uc), r0(x_T), 4, (6, 8), (4, 4, 6), cpu)
    lo, hi = D.shard_bounds(4, rank, world)
    assert torch.equal(cs, c[lo:hi]) and torch.equal(xs, x_T[lo:hi])
    assert torch.equal(ucs, uc.expand(hi - lo, -1, -1))
    assert len(calls) == 1
    # uc = None and x_T = None are legal (no guidance; per-rank noise) and come back as None on every rank
    cs, ucs, xs = D.broadcast_conditioning(r0(c), None, None, 4, (6, 8), (4, 4, 6), cpu)
    assert torch.equal(cs, c[lo:hi]) and ucs is None and xs is None
    # per-sample negative prompts: B different rows are either sharded (per_sample_uc=True on every rank) or refused
    ucB = torch.from_numpy(np.random.RandomState(5).randn(4, 5, 8).astype(np.float16))
    cs, ucs, xs = D.broadcast_conditioning(r0(c), r0(ucB), r0(x_T), 4, (6, 8), (4, 4, 6), cpu, per_sample_uc=True)
    assert torch.equal(ucs, ucB[lo:hi])
    if rank == 0:
        with pytest.raises(ValueError, match="per_sample_uc"):
            D.broadcast_conditioning(c, ucB, x_T, 4, (6, 8), (4, 4, 6), cpu)
    # B identical rows (what txt2img.py:246-248 produces) travel as one; odd element counts keep the fp32 view aligned
    cs, ucs, xs = D.broadcast_conditioning(r0(c[:, :3, :]), r0(uc[:, :3].expand(4, -1, -1)), r0(x_T[:, :1, :1, :5]), 4, (3, 8),
                                           (1, 1, 5), cpu)
    assert torch.equal(ucs, uc[:, :3].expand(hi - lo, -1, -1)) and torch.equal(xs, x_T[lo:hi, :1, :1, :5])
    calls.clear()
    # 2) the pipeline: rank 0 owns the inputs, every rank samples its shard, rank 0 gathers
    pipe = DiffusionPipeline(_FakeModel(), sampler=_FakeSampler(), device="cpu")
    out = pipe(c=c if rank == 0 else None, uc=uc if rank == 0 else None, x_T=x_T if rank == 0 else None,
               H=32, W=48, steps=7, scale=3.0, gather=True, batch_size=4)
    assert len(calls) == 1, "the pipeline must issue exactly ONE broadcast before the loop"
    if rank == 0:
        torch.save(out, os.path.join(out_dir, "gathered.pt"))
    else:
        assert out is None
    # 3) Taichu-GLIDE sharding (BASELINE config 4): prompts + the per-step unconditional token ids + seed in ONE broadcast
    calls.clear()
    from minddiffusion_amd.glide.pipeline import GlidePipeline
    pipe = GlidePipeline(_FakeGlideBase(2), _FakeGlideUp(2), text_ctx=6, vocab_len=101)
    tok, msk = _glide_prompts(4)
    img = pipe(tokens=r0(tok), mask=r0(msk), seed=11, gather=True)
    assert len(calls) == 1
    torch.save(torch.from_numpy(pipe.last_uncond_tokens), os.path.join(out_dir, f"uncond_{rank}.pt"))
    if rank == 0:
        torch.save(img, os.path.join(out_dir, "glide.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_equals_single_process(tmp_path):
    from minddiffusion_amd.pipeline import DiffusionPipeline
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), "gathered.pt"))
    c, uc, x_T = _inputs()
    ref = DiffusionPipeline(_FakeModel(), sampler=_FakeSampler(), device="cpu")(
        c=c, uc=uc, x_T=x_T, H=32, W=48, steps=7, scale=3.0)
    assert got.shape == ref.shape == (4, 4, 4, 6)
    assert torch.equal(got, ref)
    # GLIDE: both ranks used the SAME per-step unconditional prompts, and every image equals the
    # single-process result bit for bit (per-image noise streams, no dependence on the sharding)
    from minddiffusion_amd.glide.pipeline import GlidePipeline
    u0 = torch.load(os.path.join(str(tmp_path), "uncond_0.pt"))
    u1 = torch.load(os.path.join(str(tmp_path), "uncond_1.pt"))
    assert torch.equal(u0, u1) and tuple(u0.shape) == (5, 6) and len(set(u0.reshape(-1).tolist())) > 5
    tok, msk = _glide_prompts(4)
    single = GlidePipeline(_FakeGlideBase(4), _FakeGlideUp(4), text_ctx=6, vocab_len=101)(tokens=tok, mask=msk, seed=11)
    got_g = torch.load(os.path.join(str(tmp_path), "glide.pt"))
    assert got_g.shape == single.shape == (4, 3, 8, 8) and torch.equal(got_g, single)


def test_shard_bounds():
    from minddiffusion_amd.distributed import shard_bounds
    assert [shard_bounds(32, r, 8) for r in (0, 3, 7)] == [(0, 4), (12, 16), (28, 32)]
    with pytest.raises(ValueError):
        shard_bounds(6, 0, 4)
