"""AutoencoderKL -- the decode half of the reference's ldm/models/autoencoder.py (:22-68): ``decode(z)`` =
``decoder(post_quant_conv(z))``.  ``encode`` (training / img2img only) is outside SURVEY 8(f) and raises.

Attach to ``LatentDiffusion.first_stage_model`` and ``decode_first_stage`` / ``DiffusionPipeline`` produce images:

    vae = AutoencoderKL(ddconfig=SD_VAE_DDCONFIG, embed_dim=4)
    vae.load_state_dict(params)            # reference names: post_quant_conv.*, decoder.*
    model.first_stage_model = vae
"""
from ..modules.diffusionmodules.model import Decoder


class AutoencoderKL:
    def __init__(self, ddconfig, embed_dim, ckpt_path=None, ignore_keys=(), image_key="image", colorize_nlabels=None,
                 monitor=None, use_fp16=False, device=None, use_graph=True):
        assert ddconfig["double_z"]
        if ckpt_path is not None:
            raise NotImplementedError("MindSpore .ckpt ingestion is SURVEY 8(f) item 4; pass arrays to load_state_dict")
        self.embed_dim = embed_dim
        self.ddconfig = dict(ddconfig)
        self.decoder = Decoder(device=device, use_graph=use_graph, **ddconfig)

    def parameter_shapes(self):
        zc = self.ddconfig["z_channels"]
        s = {"post_quant_conv.weight": (zc, self.embed_dim, 1, 1), "post_quant_conv.bias": (zc,)}
        s.update(self.decoder.parameter_shapes("decoder."))
        return s

    def load_state_dict(self, params, strict=True):
        self.decoder.load_state_dict(params, prefix="decoder.",
                                     post_quant=(params["post_quant_conv.weight"], params["post_quant_conv.bias"]),
                                     strict=strict)

    def decode(self, z):
        """autoencoder.py:65-68.  Returns a fresh tensor (the decoder's own output buffer is reused by the next call)."""
        return self.decoder(z).clone()

    def encode(self, x):
        raise NotImplementedError("AutoencoderKL.encode is not on the txt2img path (autoencoder.py:70-78)")
