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


class _FakeModel:
    unet = _FakeUnet()


def _inputs(B=4, T=5, Dc=8, h=4, w=6):
    rng = np.random.RandomState(0)
    c = torch.from_numpy(rng.randn(B, T, Dc).astype(np.float16))
    uc = torch.from_numpy(rng.randn(1, T, Dc).astype(np.float16))
    x_T = torch.from_numpy(rng.randn(B, 4, h, w).astype(np.float32))
    return c, uc, x_T


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from minddiffusion_amd import distributed as D
    from minddiffusion_amd.pipeline import DiffusionPipeline
    r, n, _ = D.init_from_env(backend="gloo")
    assert (r, n) == (rank, world)
    c, uc, x_T = _inputs()
    # 1) the packed broadcast hands every rank its shard; the fp32 noise survives bit-exactly
    cs, ucs, xs = D.broadcast_conditioning(c if rank == 0 else None, uc if rank == 0 else None,
                                           x_T if rank == 0 else None, 4, (5, 8), (4, 4, 6), torch.device("cpu"))
    lo, hi = D.shard_bounds(4, rank, world)
    assert torch.equal(cs, c[lo:hi]) and torch.equal(xs, x_T[lo:hi])
    assert torch.equal(ucs, uc.expand(hi - lo, -1, -1))
    # 2) the pipeline: rank 0 owns the inputs, every rank samples its shard, rank 0 gathers
    pipe = DiffusionPipeline(_FakeModel(), sampler=_FakeSampler(), device="cpu")
    out = pipe(c=c if rank == 0 else None, uc=uc if rank == 0 else None, x_T=x_T if rank == 0 else None,
               H=32, W=48, steps=7, scale=3.0, gather=True)
    if rank == 0:
        torch.save(out, os.path.join(out_dir, "gathered.pt"))
    else:
        assert out is None
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


def test_shard_bounds():
    from minddiffusion_amd.distributed import shard_bounds
    assert [shard_bounds(32, r, 8) for r in (0, 3, 7)] == [(0, 4), (12, 16), (28, 32)]
    with pytest.raises(ValueError):
        shard_bounds(6, 0, 4)
