"""Text tower pieces of the reference's model/openai_model.py that sit on the hot-path call chain.

QuickGELU (openai_model.py:177-179) is the TimeSformer MLP activation -- on the GPU it is fused into the
fc1 GEMM epilogue (csrc/gemm.hip); the nn.Module here only carries the name for constructor compatibility.
The CLIP text Transformer (openai_model.py:182-232) is on the call path of `CLIP.forward` (SURVEY.md section 8f
rank 1).  With frozen weights on the GPU (`Transformer.forward_frozen`) its Linears / LayerNorms run on the libhh
GEMM (bias / QuickGELU / residual epilogues) and LayerNorm kernels and the 77x77 causal attention core on
hh_text_attn_fwd (csrc/attn_text.hip; head dim 64, L <= 80 -- other shapes raise: there is no stock-op fallback).  The stock-module
`forward` is kept for trainable / CPU use.
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


# rows of the frozen text stream are padded to a multiple of this (the persistent GEMM's row tile); 0 = off (A/B: bench.py --no-text-pad)
ROW_PAD = 256


class Transformer(nn.Module):
    """openai_model.py:219-232."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])
        self._pack = None
        self._opad = {}

    def forward(self, x: torch.Tensor, use_checkpoint=False):
        return self.resblocks(x)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._pack = None
        self._opad = {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pack = None
        self._opad = {}
        return r

    def packed(self):
        if self._pack is None:
            from .. import ops
            f = lambda t: t.detach().float().contiguous()
            self._pack = [{"ln1": (f(b.ln_1.weight), f(b.ln_1.bias), b.ln_1.eps), "ln2": (f(b.ln_2.weight), f(b.ln_2.bias), b.ln_2.eps),
                           "win": ops.to_bf16(f(b.attn.in_proj_weight)), "bin": f(b.attn.in_proj_bias),
                           "wout": ops.to_bf16(f(b.attn.out_proj.weight)), "bout": f(b.attn.out_proj.bias),
                           "wfc": ops.to_bf16(f(b.mlp.c_fc.weight)), "bfc": f(b.mlp.c_fc.bias),
                           "wpr": ops.to_bf16(f(b.mlp.c_proj.weight)), "bpr": f(b.mlp.c_proj.bias)} for b in self.resblocks]
        return self._pack

    @torch.no_grad()
    def forward_frozen(self, x: torch.Tensor, final_ln=None, owns_x=False):
        """Inference path for frozen weights on the GPU: x fp32 [S, L, W] (batch-first) -> fp32 [S, L, W].
        Same maths as forward() (pre-LN causal MHA + QuickGELU MLP), bf16 GEMM operands, fp32 residual stream.
        final_ln = (weight, bias, eps): also apply that LayerNorm (CLIP.ln_final) -- fused with the last block's residual add, so the
        stream's last update and the normalisation are one pass; owns_x: x is a temporary of the caller and may be updated in place."""
        from .. import ops
        S, L, W = x.shape
        h, d = self.heads, W // self.heads
        if W % 64 or (3 * W) % 128 or (4 * W) % 128:
            raise NotImplementedError("Transformer.forward_frozen: width must be a multiple of 64 (GEMM tiling)")
        if d != 64 or L > 80:
            raise NotImplementedError("Transformer.forward_frozen: the causal attention core (hh_text_attn_fwd) takes head dim 64 and context <= 80 "
                                      "(CLIP's 12 x 64, 77); there is no stock-op fallback")
        xs = x.reshape(S * L, W).float().contiguous()
        if xs.data_ptr() == x.data_ptr() and not owns_x:      # the residual stream is updated in place: never the caller's own tensor
            xs = xs.clone()
        # Row padding (round 6): the stream is carried with its row count rounded up to the persistent GEMM's 256-row tile (32 captions x 77 =
        # 2464 -> 2560: + 3.9 % rows of a tower that runs hidden beside the vision tower), so that none of the 48 GEMMs leaves a row tail of
        # more than 64 rows to a separate `gemm_tail_kernel` launch (36 launches per step, each waiting for free CUs inside the pipelined
        # step).  Rows are independent in every kernel but the attention, which is handed the S * L real rows only; the padding rows start
        # as zeros and are never read back.
        R, Rp = S * L, ROW_PAD * ((S * L + ROW_PAD - 1) // ROW_PAD) if ROW_PAD else S * L
        o = None
        if Rp != R:
            xp = torch.zeros((Rp, W), dtype=torch.float32, device=xs.device)
            xp[:R].copy_(xs)
            xs = xp
            # attention output: only its real rows are ever written, so ONE zero-filled buffer per shape serves every call (calls of a
            # tower are serialised on its stream)
            key = (Rp, W, xs.device)
            o = self._opad.get(key)
            if o is None:                       # (kept per shape, never freed while the module lives: a buffer another stream may still read
                o = self._opad[key] = torch.zeros((Rp, W), dtype=torch.bfloat16, device=xs.device)       #  must not go back to the allocator)
        pending = None                        # bf16 branch output not yet added to the fp32 residual stream (as in SpaceTimeBlock.fused)
        for pk in self.packed():
            xn = ops.layernorm(xs, *pk["ln1"]) if pending is None else ops.add_layernorm(xs, pending, *pk["ln1"], write_x=True)
            qkv = ops.gemm(xn, pk["win"], pk["bin"], colscale=d ** -0.5, colscale_cols=W)                  # bf16 [Rp, 3W], q scaled
            o = ops.text_attention(qkv[:R], S, L, h, out=o)                                                # libhh causal attention
            a = ops.gemm(o, pk["wout"], pk["bout"])                                                        # bf16 branch
            hid = ops.gemm(ops.add_layernorm(xs, a, *pk["ln2"], write_x=True), pk["wfc"], pk["bfc"], act=ops.ACT_QUICKGELU)
            pending = ops.gemm(hid, pk["wpr"], pk["bpr"])
        if final_ln is not None:
            g, b, eps = final_ln
            y = ops.layernorm(xs, g, b, eps, out_dtype=torch.float32) if pending is None else \
                ops.add_layernorm(xs, pending, g, b, eps, write_x=False, out_dtype=torch.float32)
            return y[:R].view(S, L, W)
        if pending is not None:
            xs += pending.float()
        return xs[:R].view(S, L, W)
