"""init_diffusion_model / init_super_res_model -- mirrors of the reference's diffusion_creator.py:27-61, returning
callables with the reference's call surface (main_funcs.py:40-41, 66):

    sample, pred_xstart = diffusion_model(x=, timesteps=, token=, mask=, random_token=, random_mask=)
    sample, pred_xstart = super_res_model(x=, timesteps=, token=, mask=, samples=)

Each call = one UNet evaluation (a replayed hipGraph, with the text transformer inside) + ONE fused sampler kernel
(mdx_glide_step_f32) replacing SamplingWithGuidance/Guider (guider.py:20-103), PMeanVariance
(gaussian_diffusion.py:229-313) and PSample / DDimSample (:65-142).

Batch convention: the reference runs the base model on 2P rows and discards rows P..2P
(src/txt2img.py:120).  Here only the P kept trajectories are computed; the UNet still sees 2P rows
([x ; x] with [prompt ; random-token] text) and the returned tensors have 2P rows with the second half a copy of the
first, so `[:pics_generated]` slicing by the caller is unchanged.
"""
import numpy as np
import torch

from .. import ops
from .gaussian_computation import alpha_calculator, get_named_beta_schedule, space_timesteps
from .model.text2im_model import SuperResText2ImUNet, Text2ImUNet

_MODEL_KEYS = ("text_ctx", "xf_width", "xf_layers", "xf_heads", "xf_final_ln", "n_vocab", "xf_padding",
               "num_res_blocks", "attention_resolutions", "dropout", "channel_mult", "use_fp16", "num_heads",
               "num_head_channels", "num_heads_upsample", "use_scale_shift_norm", "resblock_updown", "cache_text_emb")


def create_model(**options):
    """model_creator.py:20-75."""
    kw = {k: options[k] for k in _MODEL_KEYS}
    return Text2ImUNet(in_channels=3, model_channels=options["num_channels"], out_channels=6,
                       device=options.get("device", "cuda:0"), **kw)


def create_upsample_model(**options):
    """model_creator.py:78-135."""
    kw = {k: options[k] for k in _MODEL_KEYS}
    return SuperResText2ImUNet(image_size=options["image_size"], in_channels=6, model_channels=options["num_channels"],
                               out_channels=6, low_size=options.get("low_size", 64),
                               device=options.get("device", "cuda:0"), **kw)


def space_diffusion_from_base(use_timesteps, alphas_cumprod):
    """diffusion_creator.py:96-106."""
    timestep_map, new_betas, last = [], [], 1.0
    for i, a in enumerate(alphas_cumprod):
        if i in use_timesteps:
            new_betas.append(1 - a / last)
            last = a
            timestep_map.append(i)
    return timestep_map, np.array(new_betas)


class _Schedule:
    """The tables PMeanVariance.__init__ builds (gaussian_diffusion.py:196-226): betas cast to fp32 first."""

    def __init__(self, noise_schedule, diffusion_steps, timestep_respacing):
        base = get_named_beta_schedule(noise_schedule, diffusion_steps)
        use = space_timesteps(diffusion_steps, timestep_respacing or [diffusion_steps])
        tmap, nb = space_diffusion_from_base(use, alpha_calculator(base))
        self.timestep_map = np.asarray(tmap, dtype=np.int64)
        betas = np.array(nb, dtype=np.float32)
        self.num_timesteps = int(betas.shape[0])
        ac = np.cumprod(1.0 - betas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        f = lambda a: np.asarray(a, dtype=np.float32)
        self.alphas_cumprod_prev = f(ac_prev)
        self.log_betas = f(np.log(betas))
        self.post_logvar = f(np.log(np.append(pv[1], pv[1:])))
        self.sqrt_recip = f(np.sqrt(1.0 / ac))
        self.sqrt_recipm1 = f(np.sqrt(1.0 / ac - 1))
        self.coef1 = f(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.coef2 = f((1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac))

    def coef8(self, i):
        ab_prev = float(self.alphas_cumprod_prev[i])
        return (self.log_betas[i], self.post_logvar[i], self.sqrt_recip[i], self.sqrt_recipm1[i], self.coef1[i],
                self.coef2[i], np.sqrt(np.float32(ab_prev)), np.sqrt(np.float32(1.0) - np.float32(ab_prev)))


_LOOP_TABLES = __import__("os").environ.get("MDX_GLIDE_LOOP_TABLES", "1") != "0"     # 0: every step recomputes text + time embedding (A/B)


def _stamp(a):
    """Something that changes when the array's content may have.  Device tensors count their in-place writes (`_version`);
    everything on the host -- numpy arrays AND CPU tensors, whose `_version` does not move when the memory is edited through
    the numpy array `torch.from_numpy` shares it with, or through `.data` -- is hashed (128 ints per prompt: cheap)."""
    if isinstance(a, torch.Tensor):
        if a.is_cuda:
            return ("v", a._version, tuple(a.shape))
        a = a.detach().numpy()
    return ("h", hash(np.ascontiguousarray(a).tobytes()))


class GenerativePSampleDiffusionModel:
    """gaussian_diffusion.py:36-49: ancestral sampling step with classifier-free guidance."""

    def __init__(self, model, schedule, guidance_scale, shape):
        self.model = model
        self.schedule = schedule
        self.guidance_scale = float(guidance_scale)
        self.shape = tuple(shape)
        self.pics_generated = shape[0] // 2
        self.num_timesteps = schedule.num_timesteps
        self.generator = None
        self._loop = None

    # ---- whole-loop tables (Text2ImUNet.begin_loop): the loop (main_funcs.gaussian_p_sample_loop) announces every step's
    # unconditional prompt before its first step; a step whose arguments are the announced ones skips the text transformer, the
    # encoder_kv projections and the time-embedding chain.  Anything else falls back to the full per-step evaluation.
    def begin_loop(self, token, mask, uncond_tokens, uncond_masks=None):
        P = self.pics_generated
        S = self.num_timesteps
        unc = np.ascontiguousarray(np.asarray(uncond_tokens)[:S], dtype=np.int32)
        if unc.shape[0] != S:
            raise ValueError(f"begin_loop: {unc.shape[0]} unconditional prompts for {S} steps")
        um = None if uncond_masks is None else np.ascontiguousarray(np.asarray(uncond_masks)[:S], dtype=np.int32)
        order = list(range(S))[::-1]                       # main_funcs.py:36: i = S - 1 ... 0
        t_values = [float(self.schedule.timestep_map[i]) for i in order]
        tok = torch.as_tensor(token)[:P]
        msk = torch.as_tensor(mask)[:P]
        ctx = self.model.begin_loop(2 * P, self.shape[2], self.shape[3], t_values, tok, msk, unc, um)
        self._loop = dict(ctx=ctx, token=token, mask=mask, stamp=(_stamp(token), _stamp(mask)), unc=unc, um=um,
                          index={i: k for k, i in enumerate(order)})

    def end_loop(self):
        self._loop = None

    def _loop_step(self, i, token, mask, random_token, random_mask):
        """k if step i may take the loop tables: same prompt objects with unchanged content, and the announced unconditional
        prompt (compared by content: 128 ints)."""
        lp = self._loop
        if lp is None or i not in lp["index"] or token is not lp["token"] or mask is not lp["mask"]:
            return None
        k = lp["index"][i]
        rt = np.asarray(random_token.cpu() if isinstance(random_token, torch.Tensor) else random_token).reshape(-1)
        rm = np.asarray(random_mask.cpu() if isinstance(random_mask, torch.Tensor) else random_mask).reshape(-1)
        if not np.array_equal(rt, lp["unc"][k]) or not (np.all(rm == 1) if lp["um"] is None else np.array_equal(rm, lp["um"][k])):
            return None
        if lp["stamp"] != (_stamp(token), _stamp(mask)):
            return None
        return k

    def __call__(self, x, timesteps, token, mask, random_token=None, random_mask=None, is_train=False, noise=None):
        P = self.pics_generated
        dev = self.model.device
        i = int(torch.as_tensor(timesteps).reshape(-1)[0])
        xs = x[:P].contiguous()
        k = self._loop_step(i, token, mask, random_token, random_mask) if _LOOP_TABLES else None
        if k is not None and tuple(xs.shape) == (P,) + tuple(self.shape[1:]):
            out = self.model.loop_step(self._loop["ctx"], k, torch.cat([xs, xs], 0))
            if noise is None and i != 0:
                noise = torch.randn(xs.shape, device=dev, dtype=torch.float32, generator=self.generator)
            sample, pred = torch.empty_like(xs), torch.empty_like(xs)
            ops.glide_step(xs, out[:P], out[P:], out.shape[-1], self.guidance_scale, self.schedule.coef8(i), 0,
                           0.0 if i == 0 else 1.0, None if i == 0 else noise[:P].contiguous(), sample, pred)
            return torch.cat([sample, sample], 0), torch.cat([pred, pred], 0)
        tok = torch.as_tensor(token)[:P].to(dev, torch.int32)
        msk = torch.as_tensor(mask)[:P].to(dev, torch.int32)
        rt = torch.as_tensor(random_token).to(dev, torch.int32).reshape(1, -1).expand(P, -1)   # guider.py:46-47
        rm = torch.as_tensor(random_mask).to(dev, torch.int32).reshape(1, -1).expand(P, -1)
        t = torch.full((2 * P,), float(self.schedule.timestep_map[i]), device=dev)
        out = self.model.forward_nhwc(torch.cat([xs, xs], 0), t, torch.cat([tok, rt], 0), torch.cat([msk, rm], 0))
        if noise is None and i != 0:
            noise = torch.randn(xs.shape, device=dev, dtype=torch.float32, generator=self.generator)
        sample, pred = torch.empty_like(xs), torch.empty_like(xs)
        ops.glide_step(xs, out[:P], out[P:], out.shape[-1], self.guidance_scale, self.schedule.coef8(i), 0,
                       0.0 if i == 0 else 1.0, None if i == 0 else noise[:P].contiguous(), sample, pred)
        return torch.cat([sample, sample], 0), torch.cat([pred, pred], 0)


_TEXT_CACHE = __import__("os").environ.get("MDX_GLIDE_TEXT_CACHE", "1") != "0"      # 0: recompute the text transformer every step (A/B)
_TEXT_EPOCHS = __import__("itertools").count(1)      # process-wide, monotonically increasing text-prefix epochs


class DDimSampleDiffusionModel:
    """gaussian_diffusion.py:51-62: DDIM (eta = 0) step of the super-resolution model."""

    def __init__(self, model, schedule, shape):
        self.model = model
        self.schedule = schedule
        self.shape = tuple(shape)
        self.num_timesteps = schedule.num_timesteps
        self._text = None         # (token object, mask object, their versions / digests, device copies, epoch)
        self._epoch = 0
        self._loop = None

    _stamp = staticmethod(lambda a: _stamp(a))

    def begin_loop(self, token, mask):
        """main_funcs.ddim_sample_loop announces its loop: text transformer, encoder_kv projections and the time-embedding chain
        of all `num_timesteps` steps run once (Text2ImUNet.begin_loop)."""
        P = int(self.shape[0])
        S = self.num_timesteps
        order = list(range(S))[::-1]
        t_values = [float(self.schedule.timestep_map[i]) for i in order]
        ctx = self.model.begin_loop(P, self.shape[2], self.shape[3], t_values, torch.as_tensor(token)[:P], torch.as_tensor(mask)[:P])
        self._loop = dict(ctx=ctx, token=token, mask=mask, stamp=(_stamp(token), _stamp(mask)), P=P,
                          index={i: k for k, i in enumerate(order)})

    def end_loop(self):
        self._loop = None

    def __call__(self, x, timesteps, token, mask, samples, is_train=False):
        dev = self.model.device
        i = int(torch.as_tensor(timesteps).reshape(-1)[0])
        P = x.shape[0]
        lp = getattr(self, "_loop", None)
        if (_LOOP_TABLES and lp is not None and i in lp["index"] and token is lp["token"] and mask is lp["mask"] and P == lp["P"]
                and tuple(x.shape) == tuple(self.shape) and lp["stamp"] == (_stamp(token), _stamp(mask))):
            out = self.model.loop_step(lp["ctx"], lp["index"][i], x, low_res=samples[:P].to(dev, torch.float32).contiguous())
            sample, pred = torch.empty_like(x), torch.empty_like(x)
            ops.glide_step(x.contiguous(), out, None, out.shape[-1], 1.0, self.schedule.coef8(i), 1, 0.0, None, sample, pred)
            return sample, pred
        t = torch.full((P,), float(self.schedule.timestep_map[i]), device=dev)
        # the loop (main_funcs.py:47-69) hands the SAME token / mask objects to every one of its 27 steps: the model's text
        # transformer then runs once per loop instead of once per step (Text2ImUNet.forward_nhwc text_epoch)
        tx = self._text
        if (tx is None or tx[0] is not token or tx[1] is not mask or tx[2] != (self._stamp(token), self._stamp(mask), P)):
            # the epoch is drawn from ONE process-wide counter: (id(self), per-object counter) could repeat when a second sampler
            # object is allocated at a collected one's address and restarts its counter -- the model would then keep the previous
            # prompt's text prefix
            self._epoch = next(_TEXT_EPOCHS)
            tx = self._text = (token, mask, (self._stamp(token), self._stamp(mask), P),
                               torch.as_tensor(token)[:P].to(dev, torch.int32).contiguous(),
                               torch.as_tensor(mask)[:P].to(dev, torch.int32).contiguous(), ("ddim", self._epoch))
        out = self.model.forward_nhwc(x, t, tx[3], tx[4], low_res=samples[:P].to(dev, torch.float32).contiguous(),
                                      text_epoch=tx[5] if _TEXT_CACHE else None)
        sample, pred = torch.empty_like(x), torch.empty_like(x)
        ops.glide_step(x.contiguous(), out, None, out.shape[-1], 1.0, self.schedule.coef8(i), 1, 0.0, None, sample, pred)
        return sample, pred


# ---- checkpoint ingestion (src/txt2img.py:34-57)
BASE_NET_PREFIX = "p_mean_variance.guider_net.model."   # GenerativePSampleDiffusionModel.p_mean_variance (PMeanVariance)
#                                                          .guider_net (SamplingWithGuidance).model (Text2ImUNet)
SUPRES_NET_PREFIX = "p_mean_variance.guider_net."       # DDimSampleDiffusionModel.p_mean_variance.guider_net = the UNet


def rewrite_checkpoint_keys(param_dict, model_type="base"):
    """The key rewrite of src/txt2img.py:41-53, applied to {name: array}: drop every `diffusion_with_p_sample` path
    component (the training wrapper's attribute) and, for the base model, insert `model` after `guider_net` (at sampling
    time the UNet sits inside SamplingWithGuidance, diffusion_creator.py:33-44)."""
    out = {}
    for key, val in param_dict.items():
        parts = []
        for comp in key.split("."):
            if comp == "diffusion_with_p_sample":
                continue
            parts.append(comp)
            if comp == "guider_net" and model_type == "base":
                parts.append("model")
        out[".".join(parts)] = val
    return out


def load_ckpt(net, ckpt_file, model_type="base", strict=True):
    """src/txt2img.py:34-57 `load_ckpt(net, ckpt_file, model_type)`: read the MindSpore .ckpt, rewrite its keys, load
    them into `net` (what init_diffusion_model / init_super_res_model returned).  Returns the reference's
    `param_not_load` list (names of `net` parameters the file did not supply).  strict=True (default) raises if that
    list is not empty or if the file holds UNet keys the model does not own; strict=False loads what matches, as
    ms.load_param_into_net does -- but a model with missing weights cannot run here, so missing keys always raise."""
    if not ckpt_file:
        return []
    from ..ms_checkpoint import load_checkpoint
    sd = rewrite_checkpoint_keys(load_checkpoint(ckpt_file), model_type)
    prefix = BASE_NET_PREFIX if model_type == "base" else SUPRES_NET_PREFIX
    unet = net.model if hasattr(net, "model") else net
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if not own and model_type != "base":
        # diffusion_creator.py:49-50 loads a bare up-sampler checkpoint straight into the UNet (no wrapper prefix)
        own = dict(sd)
    shapes = unet.parameter_shapes()
    not_loaded = [prefix + k for k in shapes if k not in own]
    if not_loaded:
        raise KeyError(f"load_ckpt({ckpt_file!r}, model_type={model_type!r}): {len(not_loaded)} parameters of the model "
                       f"are not in the checkpoint, e.g. {not_loaded[:3]} (file keys look like {list(sd)[:2]})")
    unet.load_state_dict(own, strict=strict)
    return not_loaded


def init_diffusion_model(options, guidance_scale, shape, ckpt_path=None, params=None):
    """diffusion_creator.py:27-44.  Weights: `params` ({reference parameter name: array}) or `ckpt_path` (a MindSpore .ckpt
    in the layout src/txt2img.py:100 loads with load_ckpt(..., "base")); the reference accepts ckpt_path here and loads in
    the caller -- we load here so that the argument is never silently ignored."""
    model = create_model(**options)
    if params is not None and ckpt_path:
        raise ValueError("pass either params or ckpt_path")
    if params is not None:
        model.load_state_dict(params)
    sch = _Schedule(options["noise_schedule"], options["diffusion_steps"], options["timestep_respacing"])
    net = GenerativePSampleDiffusionModel(model, sch, guidance_scale, shape)
    if ckpt_path:
        load_ckpt(net, ckpt_path, "base")
    return net


def init_super_res_model(options, shape, ckpt_path=None, params=None):
    """diffusion_creator.py:46-61 (`load_checkpoint(ckpt_path, up_sample_model)` at :49-50: bare UNet names; the wrapped
    layout of src/txt2img.py:104 is accepted too)."""
    model = create_upsample_model(**options)
    if params is not None and ckpt_path:
        raise ValueError("pass either params or ckpt_path")
    if params is not None:
        model.load_state_dict(params)
    sch = _Schedule(options["noise_schedule"], options["diffusion_steps"], options["timestep_respacing"])
    net = DDimSampleDiffusionModel(model, sch, shape)
    if ckpt_path:
        load_ckpt(net, ckpt_path, "supres")
    return net
