"""Batch-sharded sampling over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device inference at all.  The denoising trajectories of
different images are independent (GroupNorm / LayerNorm / attention are per-sample, the PLMS eps history is
per-sample), and the two CFG rows of an image stay on one GPU, so the only exchange is ONE broadcast of the
text embeddings (+ start noise) from rank 0 before the loop and an optional gather of the latents after it:
<= ~15 MB even for the 32-image 768x768 config -- no collective inside the 50-step loop, no all-reduce.

Every rank calls with the same global batch size (it is the length of the prompt list every rank's CLI holds); the
context length T, the shape of the unconditional conditioning and the presence of start noise travel in the payload's
header, so the whole exchange is exactly ONE collective.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

_HDR = 8      # int16 header slots at the front of the packed fp16 payload


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MDX_DIST_BACKEND=gloo: tests that put several ranks on ONE GPU (RCCL refuses that); payloads are then staged
            # through the host by _broadcast / gather_latents
            backend = os.environ.get("MDX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(global_batch, rank, world_size):
    """Contiguous chunks: rank r of n owns images [r*B/n, (r+1)*B/n); B must divide evenly."""
    if global_batch % world_size != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world_size}")
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


# What the collectives of this process did: bench.py --gpus N reports it so that a scaling record can PROVE which backend moved
# how many bytes over how many ranks.  calls / bytes / ms of the broadcasts since the last reset; ms is
# bracketed by HIP events on the current stream for device payloads, by the host clock for host payloads.
collective_stats = {"broadcasts": 0, "bytes": 0, "ms": 0.0, "last_ms": 0.0}
# Timing is OPT-IN (bench.py sets it): bracketing a device broadcast with HIP events and waiting on the second one makes the
# otherwise asynchronous RCCL broadcast a blocking call.  The call and byte counters are always kept.
time_collectives = False


def reset_collective_stats():
    collective_stats.update(broadcasts=0, bytes=0, ms=0.0, last_ms=0.0)


def _broadcast(payload, src):
    """The one collective.  RCCL moves device buffers directly; gloo (CPU tests, or several ranks sharing one GPU in the
    single-GPU parity test) stages a device payload through the host."""
    import time
    on_dev = payload.is_cuda
    timed = time_collectives
    if timed and on_dev:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    if dist.get_backend() == "gloo" and on_dev:
        host = payload.cpu()
        dist.broadcast(host, src=src)
        payload.copy_(host)
    else:
        dist.broadcast(payload, src=src)
    ms = 0.0
    if timed and on_dev:
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
    elif timed:
        ms = (time.perf_counter() - t0) * 1e3
    collective_stats["broadcasts"] += 1
    collective_stats["bytes"] += payload.numel() * payload.element_size()
    collective_stats["ms"] += ms
    collective_stats["last_ms"] = ms


def broadcast_conditioning(c, uc, x_T, global_batch, ctx_shape, latent_shape, device, src=0, dtype=torch.float16,
                           per_sample_uc=False):
    """ONE packed broadcast from `src`: conditioning c [B,T,D] (fp16), the unconditional conditioning and the start noise
    x_T [B,C,h,w] (fp32, bit-exact: it travels as int16 pairs inside the fp16 payload).

    ctx_shape = (T_max, D): every rank sizes the payload for T_max context tokens (the UNet's max_context_len); the actual
    T <= T_max is in the header.  uc may be None (no classifier-free guidance: samplers accept that), one row [1,T,D]
    (txt2img.py:246-248 encodes B copies of "" -- one row carries it) or, with per_sample_uc=True on every rank, B rows
    of per-sample negative prompts.  B identical rows are sent as one; B different rows without per_sample_uc raise
    instead of being silently replaced by row 0.  x_T may be None on `src` (each rank then draws its own).
    Ranks other than `src` pass None tensors.  Returns this rank's shard (c, uc or None, x_T or None)."""
    rank, n = world()
    T_max, D = int(ctx_shape[0]), int(ctx_shape[1])
    B = int(global_batch)
    uc_cap = B if per_sample_uc else 1
    n_c, n_uc = B * T_max * D, uc_cap * T_max * D
    n_lat = int(np.prod(latent_shape))
    off_x = _HDR + n_c + n_uc
    off_x += off_x % 2                     # the fp32 view of the noise needs a 4-byte aligned offset
    payload = torch.zeros(off_x + 2 * B * n_lat, dtype=dtype, device=device)
    assert payload.element_size() == 2
    hdr = payload[:_HDR].view(torch.int16)
    if rank == src:
        if c is None or tuple(c.shape[0:1]) != (B,) or c.shape[2] != D or c.shape[1] > T_max:
            raise ValueError(f"broadcast_conditioning: c must be [{B}, T<={T_max}, {D}], got "
                             f"{None if c is None else tuple(c.shape)}")
        T = int(c.shape[1])
        payload[_HDR:_HDR + n_c].view(B, T_max, D)[:, :T].copy_(c)
        uc_rows = 0
        if uc is not None:
            if uc.shape[1] != T or uc.shape[2] != D or uc.shape[0] not in (1, B):
                raise ValueError(f"broadcast_conditioning: uc must be [1 or {B}, {T}, {D}], got {tuple(uc.shape)}")
            if uc.shape[0] == B and B > 1 and not per_sample_uc:
                if not bool((uc == uc[:1]).all()):
                    raise ValueError("broadcast_conditioning: per-sample unconditional rows need per_sample_uc=True "
                                     "(on every rank); refusing to replace them by row 0")
                uc = uc[:1]
            uc_rows = int(uc.shape[0])
            payload[_HDR + n_c:_HDR + n_c + n_uc].view(uc_cap, T_max, D)[:uc_rows, :T].copy_(uc)
        hdr[0], hdr[1], hdr[2] = T, uc_rows, int(x_T is not None)
        if x_T is not None:
            payload[off_x:].view(torch.float32).copy_(x_T.reshape(-1).to(torch.float32))
    if n > 1:
        _broadcast(payload, src)
    T, uc_rows, has_x = (int(v) for v in hdr[:3].tolist())
    lo, hi = shard_bounds(B, rank, n)
    c_all = payload[_HDR:_HDR + n_c].view(B, T_max, D)[:, :T]
    uc_out = None
    if uc_rows:
        uc_all = payload[_HDR + n_c:_HDR + n_c + n_uc].view(uc_cap, T_max, D)[:uc_rows, :T]
        uc_out = (uc_all[lo:hi] if uc_rows == B and B > 1 else uc_all[:1].expand(hi - lo, T, D)).contiguous()
    x_out = None
    if has_x:
        x_out = payload[off_x:].view(torch.float32).view(B, *latent_shape)[lo:hi].contiguous()
    return c_all[lo:hi].contiguous(), uc_out, x_out


def broadcast_glide_inputs(tokens, mask, uncond_tokens, seed, global_pics, text_ctx, steps, device, src=0,
                           tokens_up=None, mask_up=None):
    """Taichu-GLIDE (BASELINE config 4) sharding, SURVEY 8(e): ONE packed int32 broadcast from `src` of the prompt token
    ids / masks [P,T] (base model and, optionally, the up-sampler's own tokenisation, src/txt2img.py:114,122), the
    per-step random unconditional token ids [steps,T] (main_funcs.py:37-38 redraws them every step: ALL ranks must use the
    same unconditional prompts, so rank `src` draws them once) and the base seed of the per-image noise streams.
    Returns (tokens, mask, tokens_up, mask_up) of this rank's shard, the [steps,T] unconditional ids (numpy int32), the
    seed, and (lo, hi)."""
    rank, n = world()
    P, T, S = int(global_pics), int(text_ctx), int(steps)
    n_tok = P * T
    payload = torch.zeros(4 + 4 * n_tok + S * T, dtype=torch.int32, device=device)
    if rank == src:
        as_i32 = lambda a: torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).to(device, torch.int32)
        tokens, mask = as_i32(tokens), as_i32(mask)
        if tuple(tokens.shape) != (P, T) or tuple(mask.shape) != (P, T):
            raise ValueError(f"broadcast_glide_inputs: tokens / mask must be [{P}, {T}]")
        tokens_up = tokens if tokens_up is None else as_i32(tokens_up)
        mask_up = mask if mask_up is None else as_i32(mask_up)
        unc = as_i32(uncond_tokens)
        if tuple(unc.shape) != (S, T):
            raise ValueError(f"broadcast_glide_inputs: uncond_tokens must be [{S}, {T}]")
        payload[0], payload[1] = int(seed) & 0x7FFFFFFF, 1
        for k, t in enumerate((tokens, mask, tokens_up, mask_up)):
            payload[4 + k * n_tok:4 + (k + 1) * n_tok].copy_(t.reshape(-1))
        payload[4 + 4 * n_tok:].copy_(unc.reshape(-1))
    if n > 1:
        _broadcast(payload, src)
    lo, hi = shard_bounds(P, rank, n)
    part = lambda k: payload[4 + k * n_tok:4 + (k + 1) * n_tok].view(P, T)[lo:hi].contiguous()
    unc = payload[4 + 4 * n_tok:].view(S, T).cpu().numpy().astype(np.int32)
    return part(0), part(1), part(2), part(3), unc, int(payload[0].item()), (lo, hi)


def gather_latents(local, dst=0):
    """Optional gather after the loop: [B/n,C,h,w] per rank -> [B,C,h,w] on `dst` (None elsewhere)."""
    rank, n = world()
    if n == 1:
        return local
    if dist.get_backend() == "gloo":
        host = local.cpu()
        out = [torch.empty_like(host) for _ in range(n)] if rank == dst else None
        dist.gather(host, out, dst=dst)
        return torch.cat(out, 0).to(local.device) if rank == dst else None
    out = [torch.empty_like(local) for _ in range(n)]
    dist.all_gather(out, local)     # RCCL: all_gather of <= 600 KB per rank; cheaper than emulating gather
    return torch.cat(out, 0) if rank == dst else None
