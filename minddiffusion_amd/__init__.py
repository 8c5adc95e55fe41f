"""minddiffusion_amd -- MI355X-native (gfx950) implementation of minddiffusion's iterative UNet
denoising hot path.  The package mirrors the reference's module layout for that path:

    minddiffusion_amd.ldm.models.diffusion.plms.PLMSSampler      (reference: ldm/models/diffusion/plms.py)
    minddiffusion_amd.ldm.models.diffusion.ddim.DDIMSampler      (new)
    minddiffusion_amd.ldm.models.diffusion.ddpm.LatentDiffusion  (reference: ldm/models/diffusion/ddpm.py)
    minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel.UNetModel
    minddiffusion_amd.pipeline.DiffusionPipeline                 (reference: txt2img.py main loop)

All arithmetic runs in hand-written HIP kernels (minddiffusion_amd/csrc -> libmdx.so, C-ABI in
include/mdx.h).  There is no CPU fallback anywhere in the package.
"""
__version__ = "0.1.0"
