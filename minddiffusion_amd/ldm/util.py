"""instantiate_from_config -- mirror of the reference's ldm/util.py:37-52 (importlib 'target' + 'params').

Reference YAMLs name ``ldm.…`` targets; they resolve to this package's mirrors of the hot-path classes."""
import importlib

_ALIASES = {
    "ldm.models.diffusion.ddpm.LatentDiffusion": "minddiffusion_amd.ldm.models.diffusion.ddpm.LatentDiffusion",
    "ldm.modules.diffusionmodules.openaimodel.UNetModel":
        "minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel.UNetModel",
}


def get_obj_from_str(string):
    string = _ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def exists(x):
    return x is not None


def default(val, d):
    return val if exists(val) else (d() if callable(d) else d)
