"""Text tower pieces of the reference's model/openai_model.py that sit on the hot-path call chain.

QuickGELU (openai_model.py:177-179) is the TimeSformer MLP activation -- on the GPU it is fused into the
fc1 GEMM epilogue (csrc/gemm.hip); the nn.Module here only carries the name for constructor compatibility.
The CLIP text Transformer (openai_model.py:182-232) is on the call path of `CLIP.forward` but is not part of
the north-star kernel set: it stays on stock PyTorch-ROCm ops.
"""
from collections import OrderedDict

import torch
from torch import nn


class QuickGELU(nn.Module):
    def forward(self, x: torch.Tensor):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    """openai_model.py:182-216 (pre-LN causal MHA + QuickGELU MLP)."""

    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.attn_mask = attn_mask

    def forward(self, x: torch.Tensor, use_checkpoint=False):
        mask = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    """openai_model.py:219-232."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x: torch.Tensor, use_checkpoint=False):
        return self.resblocks(x)
