"""Shape configuration of the hot path (frozen LaViLa TimeSformer + object-query decoder).

Defaults are the reference's hard-coded hyper-parameters:
  vision tower   /root/reference/model/LaviLa.py:118-129  (TimeSformer-L, patch 14, 24x1024, 16 heads)
  text tower     /root/reference/model/LaviLa.py:151-162  (12x768, 12 heads, ctx 77, vocab 49408, embed 256)
  decoder        /root/reference/model/tfm_decoder.py:51-54,112-122 (d512, 8 heads, 6 layers, ffn 2048)
  run/train.py:447-457 builds ObjDecoder(num_queries=args.num_queries+1, num_classes=22047, feature_dim=1024)
"""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class HHConfig:
    img_size: int = 224
    patch_size: int = 14
    num_frames: int = 16
    embed_dim: int = 1024          # D, vision width
    depth: int = 24
    num_heads: int = 16
    mlp_ratio: int = 4
    text_width: int = 768
    text_layers: int = 12
    text_heads: int = 12
    context_length: int = 77
    vocab_size: int = 49408
    project_embed_dim: int = 256
    dec_dim: int = 512
    dec_heads: int = 8
    dec_layers: int = 6
    dec_ffn: int = 2048
    num_queries: int = 12          # nq of the reference CLI; decoder holds nq+1 (last = video summary)
    num_classes: int = 22047
    n_nouns: int = 582
    n_verbs: int = 118
    captions_per_clip: int = 5

    @property
    def patches_per_frame(self) -> int:
        return (self.img_size // self.patch_size) ** 2

    @property
    def tokens(self) -> int:
        return 1 + self.num_frames * self.patches_per_frame

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def dec_queries(self) -> int:
        return self.num_queries + 1

    def with_(self, **kw) -> "HHConfig":
        return replace(self, **kw)


# BASELINE.json configs
C1 = HHConfig(num_frames=4, num_queries=4)                       # run/train.py plumbing case (B=2)
C2 = HHConfig(num_frames=16, num_queries=12)                     # headline: 16-frame 224p nq=12
C4 = HHConfig(num_frames=32, img_size=336, num_queries=12)       # long clip / high-res stress
# reduced-width configs used by committed golden fixtures (same code path, d=64 heads)
TINY4 = HHConfig(num_frames=4, embed_dim=128, depth=2, num_heads=2, text_width=768, text_layers=2,
                 text_heads=12, num_queries=4, vocab_size=512)
TINY16 = HHConfig(num_frames=16, embed_dim=128, depth=2, num_heads=2, text_width=768, text_layers=2,
                  text_heads=12, num_queries=12, vocab_size=512)
