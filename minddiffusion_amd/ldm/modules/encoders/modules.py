"""FrozenCLIPEmbedder_ZH -- mirror of the reference's ldm/modules/encoders/modules.py:23-41: the frozen text encoder
LatentDiffusion.get_learned_conditioning calls (``cond_stage_model.encode(list[str])``).

The tokenizer (BPE for SDv2, WordPiece for Wukong: ldm/models/clip_zh/simple_tokenizer.py) is host-side preprocessing
outside SURVEY 8 and needs vocabulary files the repository does not ship; pass any callable ``tokenizer(list[str]) ->
int array [B, 77]`` to use ``encode``; ``construct(token_ids)`` needs none.
"""
from ...._lib import MdxError
from .text_encoder import TextEncoder


class FrozenCLIPEmbedder_ZH:
    def __init__(self, max_length=77, use_fp16=False, tokenizer=None, device=None, vocab_size=49408, width=1024,
                 layers=23, heads=16, act="gelu_tanh", ln_eps=1e-5):
        self.max_length = max_length
        self.tokenizer = tokenizer
        # modules.py:29: TextEncoder(context_length=77, vocab_size=49408, output_dim=1024, width=1024, layers=23, heads=16)
        self.transformer = TextEncoder(context_length=max_length, vocab_size=vocab_size, output_dim=width, width=width,
                                       layers=layers, heads=heads, act=act, device=device, ln_eps=ln_eps)

    @classmethod
    def wukong(cls, max_length=77, use_fp16=False, tokenizer=None, device=None):
        """The Wukong-Huahua embedder (wukong-huahua/ldm/modules/encoders/modules.py:24-30): width 768, 12 layers, 12 heads,
        QuickGELU = x * sigmoid(1.702 x) (its text_encoder.py:67-74), ln_1 / ln_2 with MindSpore's default epsilon 1e-7
        (its text_encoder.py:91,100; SDv2 passes 1e-5); WordPiece tokenizer to be supplied by the caller."""
        return cls(max_length=max_length, use_fp16=use_fp16, tokenizer=tokenizer, device=device, vocab_size=49408,
                   width=768, layers=12, heads=12, act="quick_gelu", ln_eps=1e-7)

    def parameter_shapes(self):
        return self.transformer.parameter_shapes("transformer.")

    def load_state_dict(self, params, strict=True):
        self.transformer.load_state_dict(params, prefix="transformer.", strict=strict)

    def tokenize(self, texts):
        if self.tokenizer is None:
            raise MdxError("FrozenCLIPEmbedder_ZH: no tokenizer attached (pass tokenizer=callable; the reference's BPE "
                           "vocabulary is not part of this repository)")
        return self.tokenizer(texts)

    def encode(self, text):
        return self.transformer(self.tokenize(text))          # modules.py:34-37

    def construct(self, c):
        return self.transformer(c)                            # modules.py:39-41

    __call__ = construct
