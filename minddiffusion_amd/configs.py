"""The reference's shipped UNet configurations (``unet_config.params`` of its YAMLs), verbatim keys."""

# vision/stablediffusionv2/configs/v2-inference.yaml:21-38
SD2_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
                use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, use_checkpoint=True,
                legacy=False, use_fp16=True)

# vision/wukong-huahua/configs/v1-inference-chinese.yaml:24-37
WUKONG_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                   transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False, use_fp16=True)

# vision/wukong-huahua/configs/wukong-huahua_inpaint_inference.yaml:20-36: the same UNet on 9 input channels
# (4 latent + 1 resized mask + 4 masked-image latent), LatentInpaintDiffusion / conditioning_key 'hybrid'
WUKONG_INPAINT_UNET = dict(WUKONG_UNET, in_channels=9)

# LatentDiffusion params common to both YAMLs (v2-inference.yaml:5-19)
SD2_LDM = dict(linear_start=0.00085, linear_end=0.0120, timesteps=1000, scale_factor=0.18215,
               conditioning_key="crossattn", image_size=64, channels=4, use_fp16=True)

# a small structurally identical UNet for tests (head dim 64, channels multiples of 64)
TINY_UNET = dict(image_size=8, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2, 1],
                 num_res_blocks=1, channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True,
                 use_linear_in_transformer=True, transformer_depth=1, context_dim=64, legacy=False)

# Wukong-style small UNet for tests: 8 heads => head dims 40 / 80 (the full model adds 160), conv proj_in/out
SMALL_WUKONG_UNET = dict(image_size=8, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[2, 1],
                         num_res_blocks=1, channel_mult=[1, 2], num_heads=8, use_spatial_transformer=True,
                         transformer_depth=1, context_dim=96, legacy=False)

# first_stage_config.params.ddconfig of both YAMLs (v2-inference.yaml:46-60): the SD VAE (embed_dim 4)
SD_VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                       ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)

# small structurally identical decoder for tests (channels multiples of 64 so the HALO conv kernel is exercised)
TINY_VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=64,
                         ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
