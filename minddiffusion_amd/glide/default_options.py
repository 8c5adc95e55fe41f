"""Model / diffusion hyper-parameters -- mirror of the reference's default_options.py:19-148 (same keys; `dtype` is
kept for signature compatibility, this implementation always runs fp16 storage / fp32 accumulate)."""


def model_and_diffusion_defaults(image_size=64, num_channels=192, num_res_blocks=3, channel_mult=(1, 2, 3, 4),
                                 num_heads=1, num_head_channels=64, num_heads_upsample=-1,
                                 attention_resolutions=(2, 4, 8), dropout=0.9, text_ctx=128, xf_width=512,
                                 xf_layers=16, xf_heads=8, xf_final_ln=True, n_vocab=50001, xf_padding=True,
                                 diffusion_steps=1000, noise_schedule="squaredcos_cap_v2", timestep_respacing="60",
                                 use_scale_shift_norm=True, resblock_updown=True, use_fp16=True, cache_text_emb=False,
                                 inpaint=False, super_res=False, chinese=True, sketch=False, class_balanced=False,
                                 sketch_classes=0, dtype=None):
    return dict(locals())


def model_and_diffusion_upsample(image_size=256, num_channels=192, num_res_blocks=2, channel_mult=(1, 1, 2, 2, 4, 4),
                                 num_heads=1, num_head_channels=64, num_heads_upsample=-1,
                                 attention_resolutions=(32, 16, 8), dropout=0.0, text_ctx=128, xf_width=512,
                                 xf_layers=16, xf_heads=8, xf_final_ln=True, n_vocab=50257, xf_padding=True,
                                 diffusion_steps=1000, noise_schedule="linear", timestep_respacing="fast27",
                                 use_scale_shift_norm=True, resblock_updown=True, use_fp16=True, cache_text_emb=False,
                                 inpaint=False, super_res=False, chinese=False, sketch=False, class_balanced=False,
                                 sketch_classes=0, dtype=None):
    return dict(locals())
