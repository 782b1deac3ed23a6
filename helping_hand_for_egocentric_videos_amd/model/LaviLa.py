"""Frozen LaViLa dual encoder: TimeSformer vision tower on libhh kernels + CLIP wrapper.

Mirror of /root/reference/model/LaviLa.py for the hot path: `SpaceTimeTransformer` (:393-581), `SpaceTimeBlock`
(:305-390), `VarAttention` (:226-283), `Mlp` (:175-191), `VideoPatchEmbed` (:200-223), `CLIP` (:586-687).
Constructor arguments, forward signatures, returned values and state_dict keys are the reference's; factories
that download weights (:19-172) are out of scope (no network).

MI355X design (not a translation of the reference's op sequence):
  * the vision tower is inference-only (frozen, run under no_grad at run/train.py:109-110): weights are cached once
    as bf16 [N,K] operands, the residual stream stays fp32, every Linear runs on the MFMA GEMM with its bias /
    q-scale / QuickGELU / residual fused into the epilogue, LayerNorms are single-pass wave-per-row kernels,
  * one token-major activation layout [B*N, D] end to end -- no rearrange/cat/chunk copies (9 per attention call in
    the reference): the attention kernels read q|k|v straight out of the QKV GEMM output,
  * supports only what the hot path uses: attention_style 'frozen-in-time', no adapters / tanh gating / drop-path.
"""
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .openai_model import QuickGELU, Transformer


def _require_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the product path runs on libhh HIP kernels only (got a {x.device} tensor); "
                           "there is no CPU fallback -- use oracle/ for CPU reference numbers")


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        if not isinstance(self.act, QuickGELU):
            raise NotImplementedError("Mlp: only QuickGELU is fused into the GEMM epilogue (LaViLa uses QuickGELU)")
        if drop != 0.:
            raise NotImplementedError("Mlp: dropout is 0 on the frozen tower")

    def forward(self, x):
        """x [..., D] fp32/bf16 -> fc2(QuickGELU(fc1(x))) fp32 (unfused standalone form)."""
        _require_gpu(x, "Mlp")
        shp = x.shape
        a = ops.to_bf16(x.reshape(-1, shp[-1]).contiguous())
        h = ops.gemm(a, ops.to_bf16(self.fc1.weight.detach()), self.fc1.bias.detach(), act=ops.ACT_QUICKGELU)
        y = ops.gemm(h, ops.to_bf16(self.fc2.weight.detach()), self.fc2.bias.detach(), out_dtype=torch.float32)
        return y.view(shp)


class VideoPatchEmbed(nn.Module):
    """Video to patch embedding (LaviLa.py:200-223): Conv2d(k=P, s=P) == im2col + GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=8, ln_pre=False):
        super().__init__()
        img_size = (img_size, img_size) if not isinstance(img_size, tuple) else img_size
        patch_size = (patch_size, patch_size) if not isinstance(patch_size, tuple) else patch_size
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * num_frames
        self.num_frames, self.embed_dim = num_frames, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=not ln_pre)
        if in_chans != 3 or patch_size[0] != patch_size[1]:
            raise NotImplementedError("VideoPatchEmbed: 3-channel square patches only")

    def kpad(self):
        k = 3 * self.patch_size[0] * self.patch_size[1]
        return (k + 63) // 64 * 64

    def packed_weight(self):
        w = self.proj.weight.detach().reshape(self.embed_dim, -1)
        wp = torch.zeros((self.embed_dim, self.kpad()), dtype=torch.float32, device=w.device)
        wp[:, :w.shape[1]] = w
        return ops.to_bf16(wp)

    def forward(self, x):
        """x [B,F,C,H,W] -> [B*F, embed_dim, H/P, W/P] (reference layout, LaviLa.py:218-223)."""
        _require_gpu(x, "VideoPatchEmbed")
        B, Fr, C, H, W = x.shape
        P = self.patch_size[0]
        patches = ops.patch_im2col(x.float().contiguous(), P, self.kpad())
        tok = ops.gemm(patches, self.packed_weight(), self.proj.bias.detach() if self.proj.bias is not None else None,
                       out_dtype=torch.float32)
        return tok.view(B * Fr, H // P, W // P, self.embed_dim).permute(0, 3, 1, 2)


QKV_HEAD_MAJOR_PLANES = True       # False: the QKV projection writes nn.Linear's token-major [B*N, 3D] (A/B measurements)
# norm3 / norm1 / norm2 (LaviLa.py:353,372,388) folded into the GEMMs around them: the GEMMs that end a branch (time proj, space proj, fc2)
# add the fp32 residual stream in their epilogue, emit z = bf16(x + branch) with its row statistics and (space proj, fc2) write the fp32
# sum back; the time qkv / space qkv / fc1 GEMMs apply rstd / mean / gamma / beta algebraically (include/hh.h, hh_gemm_epilogue.ln_stats /
# z_out / z_update): no stand-alone add+LayerNorm pass is left inside the tower.  False: the fused add+LayerNorm kernels (A/B measurements).
# The flag is read when a block packs its weights; a block whose pack was made under the other setting re-packs on its next use
# (SpaceTimeBlock.packed), and `SpaceTimeTransformer.ln_fold_packed()` reports what the tower actually runs (bench.py records that).
LN_FOLD = True
TIME_PROJ_READS_Z3 = True          # False: the time projection's epilogue reads the fp32 residual rows (round 4; A/B measurements)
# every large kernel of a tower block walks its rows opposite to its predecessor (SpaceTimeBlock.fused); False: all first to last (rounds 1-4)
WALK_ALTERNATE = True
# the residual stream as a PAIR of bf16 tensors x = hi + lo (hi is the z every LayerNorm-fold consumer reads; hh_gemm_epilogue.z_resid_lo):
# the branch-ending GEMMs move 8 instead of 10 bytes per element.  False: fp32 stream + separate z (rounds 4-5)
STREAM_PAIR = True


class VarAttention(nn.Module):
    """Divided space-time attention (LaviLa.py:226-283)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., initialize='random'):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if initialize == 'zeros':
            self.qkv.weight.data.fill_(0)
            if self.qkv.bias is not None:
                self.qkv.bias.data.fill_(0)
            self.proj.weight.data.fill_(1)
            self.proj.bias.data.fill_(0)
        self.attn_drop = nn.Dropout(attn_drop)      # never applied by the reference either (LaviLa.py:243)
        self.proj_drop = nn.Dropout(proj_drop)
        if head_dim != 64:
            raise NotImplementedError("VarAttention: libhh attention kernels are specialised for head_dim 64")
        if proj_drop != 0.:
            raise NotImplementedError("VarAttention: proj_drop is 0 on the frozen tower")

    def packed(self):
        """bf16 [N,K] operands + fp32 biases for the fused path."""
        return {"wqkv": ops.to_bf16(self.qkv.weight.detach()), "bqkv": None if self.qkv.bias is None else self.qkv.bias.detach().float(),
                "wproj": ops.to_bf16(self.proj.weight.detach()), "bproj": self.proj.bias.detach().float()}

    @staticmethod
    def _mode(einops_to):
        return "space" if einops_to.replace(" ", "") == "(bf)nd" else "time"

    def core(self, xn, pk, B, T, n, mode, ln=None, rev_gemm=False, rev_attn=False):
        """xn bf16 [B*N, D] (already normalised) -> attention output bf16 [B*N, D] (before proj).
        ln=(stats, folded operands): xn holds UN-normalised rows z and the LayerNorm is applied inside the qkv GEMM."""
        D = xn.shape[1]
        # q *= d^-1/2 (LaviLa.py:252) in the GEMM epilogue; the space kernel takes base-2 logits (x log2 e, ops.attention_q_scale)
        qscale = self.scale * (ops.LOG2E if mode == "space" else 1.0)
        # head-major planes [3*heads, B*N, 64]: a head's rows are contiguous 128-byte lines for the attention kernels (ops.gemm col_blocked)
        if ln is None:
            qkv = ops.gemm(xn, pk["wqkv"], pk["bqkv"], colscale=qscale, colscale_cols=D, col_blocked=QKV_HEAD_MAJOR_PLANES, reverse=rev_gemm)
        else:
            stats, (wf, cs, bf) = ln
            qkv = ops.gemm(xn, wf, bf, colscale=qscale, colscale_cols=D, col_blocked=QKV_HEAD_MAJOR_PLANES, ln=(stats, cs), reverse=rev_gemm)
        return ops.divided_attention(qkv, B, T, n, self.num_heads, mode, reverse=rev_attn)

    def forward(self, x, einops_from, einops_to, einops_dims):
        """x [B, 1+T*n, D] -> proj(attention(x)) [B, N, D] fp32.  einops strings as the reference passes them
        (LaviLa.py:492-495): '(b f) n d' with f=T -> space, '(b n) f d' with n=patches -> time."""
        _require_gpu(x, "VarAttention")
        B, N, D = x.shape
        mode = self._mode(einops_to)
        if mode == "space":
            T = einops_dims["f"]
            n = (N - 1) // T
        else:
            n = einops_dims["n"]
            T = (N - 1) // n
        pk = self.packed()
        a = self.core(ops.to_bf16(x.reshape(B * N, D).contiguous()), pk, B, T, n, mode)
        return ops.gemm(a, pk["wproj"], pk["bproj"], out_dtype=torch.float32).view(B, N, D)


class SpaceTimeBlock(nn.Module):
    """LaviLa.py:305-390: t = timeattn(norm3(x)); s = attn(norm1(x+t)); y = x + s; out = y + mlp(norm2(y))."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, n_layer=0, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, time_init='zeros',
                 attention_style='frozen-in-time', is_tanh_gating=False, use_adapter=False):
        super().__init__()
        if attention_style != 'frozen-in-time' or is_tanh_gating or use_adapter or drop_path > 0.:
            raise NotImplementedError("SpaceTimeBlock: only the hot-path configuration of LaViLa is built "
                                      "(frozen-in-time, no gating / adapters / drop-path)")
        self.norm1 = norm_layer(dim)
        self.attn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.timeattn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                     proj_drop=drop, initialize=time_init)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.norm3 = norm_layer(dim)
        self.attention_style = attention_style
        self.use_adapter = False
        self._pack = None

    def packed(self, refresh=False):
        if self._pack is None or refresh or self._pack["ln_fold"] != bool(LN_FOLD):
            ln = lambda m: (m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous(), m.eps)
            self._pack = {"ln_fold": bool(LN_FOLD), "n1": ln(self.norm1), "n2": ln(self.norm2), "n3": ln(self.norm3), "time": self.timeattn.packed(),
                          "space": self.attn.packed(), "w1": ops.to_bf16(self.mlp.fc1.weight.detach()),
                          "b1": self.mlp.fc1.bias.detach().float(), "w2": ops.to_bf16(self.mlp.fc2.weight.detach()),
                          "b2": self.mlp.fc2.bias.detach().float()}
            if LN_FOLD:
                # norm3 -> time qkv, norm1 -> space qkv, norm2 -> fc1: (bf16(gamma o W), its row sums, beta W^T + b)
                self._pack["qkv_n3"] = ops.fold_layernorm_into_linear(self.timeattn.qkv.weight, self.timeattn.qkv.bias, self.norm3.weight, self.norm3.bias)
                self._pack["qkv_n1"] = ops.fold_layernorm_into_linear(self.attn.qkv.weight, self.attn.qkv.bias, self.norm1.weight, self.norm1.bias)
                self._pack["fc1_n2"] = ops.fold_layernorm_into_linear(self.mlp.fc1.weight, self.mlp.fc1.bias, self.norm2.weight, self.norm2.bias)
        return self._pack

    def fused(self, x, B, T, n, pending=None):
        """x fp32 [B*N, D] residual stream, updated IN PLACE.

        LN_FOLD (default): no stand-alone LayerNorm pass.  Every GEMM that ends a branch (time proj, space proj, fc2) reads the fp32
        residual rows in its epilogue, adds its result, and emits z = bf16(sum) with the row statistics for the LayerNorm that follows;
        space proj and fc2 also write the fp32 sum back (x <- x + s, x <- x + m: LaviLa.py:384,388 -- the time branch only feeds norm1,
        :372-384).  The GEMM that follows (time qkv, space qkv, fc1) applies norm3 / norm1 / norm2 algebraically (include/hh.h).
        `pending` = (z3, stats3) of THIS block's norm3 input, produced by the previous block's fc2 (None: first block, computed here);
        returns the pair for the next block.
        LN_FOLD False: the fused add+LayerNorm kernels; `pending` / the result are then the (space branch, MLP branch) bf16 outputs
        that the next add+LayerNorm adds to x (x = (x + space) + mlp written once per block)."""
        pk = self.packed()
        if "qkv_n1" in pk and isinstance(x, tuple):
            # bf16 pair stream (STREAM_PAIR): x = (hi, lo); hi is z3 on entry, z2 after the space projection, the next block's z3 after fc2
            xh, xl = x
            eps3, eps1, eps2 = pk["n3"][2], pk["n1"][2], pk["n2"][2]
            _, st3 = pending
            alt = WALK_ALTERNATE
            a = self.timeattn.core(xh, pk["time"], B, T, n, "time", ln=(st3, pk["qkv_n3"]), rev_gemm=False, rev_attn=alt)
            _, z1, st1 = ops.gemm(a, pk["time"]["wproj"], pk["time"]["bproj"], z=(xh, eps1, False))             # z1 = bf16(hi + t): feeds norm1 only
            a = self.attn.core(z1, pk["space"], B, T, n, "space", ln=(st1, pk["qkv_n1"]), rev_gemm=alt, rev_attn=False)
            _, _, st2 = ops.gemm(a, pk["space"]["wproj"], pk["space"]["bproj"], zpair=(xh, xl, eps2), reverse=alt)     # x <- x + s; hi = z2
            wf, cs, bf = pk["fc1_n2"]
            h = ops.gemm(xh, wf, bf, act=ops.ACT_QUICKGELU, ln=(st2, cs))
            _, _, st3n = ops.gemm(h, pk["w2"], pk["b2"], zpair=(xh, xl, eps3), reverse=alt)                      # x <- x + m; hi = next z3
            return xh, st3n
        if "qkv_n1" in pk:
            eps3, eps1, eps2 = pk["n3"][2], pk["n1"][2], pk["n2"][2]
            if pending is None:
                z3 = ops.to_bf16(x)
                st3 = ops.ln_rowstats(z3, eps3)
            else:
                z3, st3 = pending
            # Walk direction (round 5): every large kernel of the block walks its rows OPPOSITE to its predecessor, so that it starts on the
            # rows that were written last -- the part of its input still in the 256 MB Infinity Cache (a kernel that starts where its
            # predecessor started evicts exactly what it is about to need): time qkv up, time attention down, time proj up, space qkv
            # down, space attention up, space proj down, fc1 up, fc2 down -- eight kernels, so the pattern repeats block after block.
            # Same tiles, same arithmetic, bit-identical results (+0.6 % clips/s, profiles/r5_ab_walk.txt).
            alt = WALK_ALTERNATE
            a = self.timeattn.core(z3, pk["time"], B, T, n, "time", ln=(st3, pk["qkv_n3"]), rev_gemm=False, rev_attn=alt)   # qkv(LN3(x))
            # z1 = x + t feeds norm1 and nothing else (LaviLa.py:372-384: the space residual goes on x): the time projection adds its
            # result to z3 = bf16(x) -- the 2-byte rows this block's norm3 just consumed -- instead of re-reading the 4-byte fp32 stream
            _, z1, st1 = ops.gemm(a, pk["time"]["wproj"], pk["time"]["bproj"], z=(z3 if TIME_PROJ_READS_Z3 else x, eps1, False))
            a = self.attn.core(z1, pk["space"], B, T, n, "space", ln=(st1, pk["qkv_n1"]), rev_gemm=alt, rev_attn=False)     # qkv(LN1(x + t))
            _, z2, st2 = ops.gemm(a, pk["space"]["wproj"], pk["space"]["bproj"], z=(x, eps2, False, True), reverse=alt)   # x <- x + s; z2
            wf, cs, bf = pk["fc1_n2"]
            h = ops.gemm(z2, wf, bf, act=ops.ACT_QUICKGELU, ln=(st2, cs))                                        # fc1(LN2(x))
            _, z3n, st3n = ops.gemm(h, pk["w2"], pk["b2"], z=(x, eps3, False, True), reverse=alt)                 # x <- x + m; next z3
            return z3n, st3n
        if pending is None:
            xn = ops.layernorm(x, *pk["n3"])
        else:
            xn = ops.add_layernorm(x, pending[0], *pk["n3"], write_x=True, delta2=pending[1])          # x = (x + s_prev) + m_prev
        a = self.timeattn.core(xn, pk["time"], B, T, n, "time")
        t = ops.gemm(a, pk["time"]["wproj"], pk["time"]["bproj"])                                             # time branch
        a = self.attn.core(ops.add_layernorm(x, t, *pk["n1"], write_x=False), pk["space"], B, T, n, "space")   # LN1(x + t)
        sp = ops.gemm(a, pk["space"]["wproj"], pk["space"]["bproj"])
        h = ops.gemm(ops.add_layernorm(x, sp, *pk["n2"], write_x=False), pk["w1"], pk["b1"], act=ops.ACT_QUICKGELU)  # LN2(x + s)
        return sp, ops.gemm(h, pk["w2"], pk["b2"])

    def forward(self, x, einops_from_space, einops_to_space, einops_from_time, einops_to_time, time_n, space_f,
                use_checkpoint=False):
        _require_gpu(x, "SpaceTimeBlock")
        B, N, D = x.shape
        y = x.float().reshape(B * N, D).clone()
        r = self.fused(y, B, space_f, time_n)
        if "qkv_n1" in self.packed():
            return y.view(B, N, D)                       # (the fold updates the residual stream inside the GEMM epilogues)
        return ((y + r[0].float()) + r[1].float()).view(B, N, D)


class SpaceTimeTransformer(nn.Module):
    """TimeSformer vision tower (LaviLa.py:393-581).  forward(x [B,T,C,H,W]) -> (x_cls [B,D], x [B,1+T*n,D])."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 num_frames=8, time_init='rand', attention_style='frozen-in-time', ln_pre=False,
                 act_layer=nn.GELU, is_tanh_gating=False, use_adapter=False):
        super().__init__()
        if hybrid_backbone is not None or representation_size or drop_rate or drop_path_rate or not ln_pre:
            raise NotImplementedError("SpaceTimeTransformer: only the LaViLa configuration (ln_pre=True, no dropout, "
                                      "no representation layer) is built")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_frames = num_frames
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = VideoPatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                           embed_dim=embed_dim, num_frames=num_frames, ln_pre=ln_pre)
        self.patches_per_frame = self.patch_embed.num_patches // num_frames
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patches_per_frame + 1, embed_dim))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        self.ln_pre = nn.LayerNorm(embed_dim)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            SpaceTimeBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                           n_layer=i, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=0., norm_layer=norm_layer,
                           time_init=time_init, attention_style=attention_style, act_layer=act_layer,
                           is_tanh_gating=is_tanh_gating, use_adapter=use_adapter) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.einops_from_space, self.einops_to_space = 'b (f n) d', '(b f) n d'
        self.input_mean, self.input_std = ops.NORM_MEAN, ops.NORM_STD     # used only for uint8 inputs (run/train.py:442-445)
        self.einops_from_time, self.einops_to_time = 'b (f n) d', '(b n) f d'
        self._pack = None

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def refresh_weights(self):
        """Re-derive the cached bf16 operands (call after load_state_dict / parameter edits)."""
        self._pack = None
        for b in self.blocks:
            b._pack = None

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.refresh_weights()

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.refresh_weights()
        return r

    def ln_fold_packed(self):
        """True / False: every block's packed operands are the LayerNorm-fold / stand-alone set (None: mixed or nothing packed yet)."""
        modes = {b._pack["ln_fold"] if b._pack is not None else None for b in self.blocks}
        return modes.pop() if len(modes) == 1 else None

    def packed(self):
        if self._pack is None:
            f = lambda t: t.detach().float().contiguous()
            self._pack = {"wpatch": self.patch_embed.packed_weight(), "cls": f(self.cls_token).view(-1),
                          "pos": f(self.pos_embed[0]), "tmp": f(self.temporal_embed[0]),
                          "ln_pre": (f(self.ln_pre.weight), f(self.ln_pre.bias), self.ln_pre.eps),
                          "norm": (f(self.norm.weight), f(self.norm.bias), self.norm.eps)}
        return self._pack

    @torch.no_grad()
    def forward_features(self, x, use_checkpoint=False, cls_at_last=True, out_dtype=torch.float32, split_cls=False):
        """LaviLa.py:537-573.  The tower is frozen in this path (run/train.py:89,109-110): no autograd graph.
        split_cls (keyword-only in spirit; the reference has no such argument): return (x_cls [B, D], patch rows [B, T*n, D]) with the
        patch rows contiguous -- what the decoder takes (run/train.py:115: image_feature_map[:, 1:]) -- instead of the [B, 1+T*n, D] map;
        the final LayerNorm writes them there directly, no strided copy of the feature map."""
        with ops.prof_role(ops.PROF_ROLE_VISION):
            return self._forward_features(x, out_dtype, split_cls)

    def _forward_features(self, x, out_dtype, split_cls):
        _require_gpu(x, "SpaceTimeTransformer")
        B, T = x.shape[:2]
        if T > self.num_frames:
            raise ValueError(f"SpaceTimeTransformer: {T} frames > num_frames={self.num_frames}")
        n, D = self.patches_per_frame, self.embed_dim
        pk = self.packed()
        if x.dtype == torch.uint8:       # decoded frames: ToTensor + Normalize fused into the im2col kernel
            patches = ops.patch_im2col_u8(x.contiguous(), self.patch_embed.patch_size[0], self.patch_embed.kpad(),
                                          self.input_mean, self.input_std)
        else:
            patches = ops.patch_im2col(x.float().contiguous(), self.patch_embed.patch_size[0], self.patch_embed.kpad())
        tok = ops.gemm(patches, pk["wpatch"], out_dtype=torch.float32)
        fold = bool(LN_FOLD) and len(self.blocks) > 0
        pending = None
        if fold and STREAM_PAIR and "qkv_n1" in self.blocks[0].packed():
            # bf16 pair stream x = hi + lo (hh_gemm_epilogue.z_resid_lo): hi = bf16(x) is block 0's norm3 input and every later LayerNorm input, the
            # branch-ending GEMMs update both halves in place, the final norm reads their sum -- no fp32 copy of the stream exists
            xh, xl, st0 = ops.embed_ln_pre(tok, pk["cls"], pk["pos"], pk["tmp"], *pk["ln_pre"][:2], B, T, n, pk["ln_pre"][2], z_eps=self.blocks[0].norm3.eps, pair=True)
            del patches, tok
            pending = (xh, st0)
            for blk in self.blocks:
                pending = blk.fused((xh, xl), B, T, n, pending)
            if split_cls:
                cls, pat = ops.layernorm_split_cls(xh, *pk["norm"], clips=B, out_dtype=out_dtype, x_lo=xl)
                return self.pre_logits(cls), pat
            xs = xh.float().add_(xl.float())
            out = ops.layernorm(xs, *pk["norm"], out_dtype=out_dtype).view(B, 1 + T * n, D)
            return self.pre_logits(out[:, 0]), out
        if fold:      # z = bf16(x) and its row statistics for block 0's folded norm3 ride in the same pass (no cast + hh_ln_rowstats launches)
            xs, z0, st0 = ops.embed_ln_pre(tok, pk["cls"], pk["pos"], pk["tmp"], *pk["ln_pre"][:2], B, T, n, pk["ln_pre"][2], z_eps=self.blocks[0].norm3.eps)
            pending = (z0, st0)
        else:
            xs = ops.embed_ln_pre(tok, pk["cls"], pk["pos"], pk["tmp"], *pk["ln_pre"][:2], B, T, n, pk["ln_pre"][2])
        del patches, tok
        xs = xs.view(B * (1 + T * n), D)
        for blk in self.blocks:
            pending = blk.fused(xs, B, T, n, pending)
        if pending is None or "qkv_n1" in self.blocks[0].packed():       # (with the fold xs already holds the full residual stream)
            if split_cls:
                cls, pat = ops.layernorm_split_cls(xs, *pk["norm"], clips=B, out_dtype=out_dtype)
                return self.pre_logits(cls), pat
            out = ops.layernorm(xs, *pk["norm"], out_dtype=out_dtype)
        else:
            out = ops.add_layernorm(xs, pending[0], *pk["norm"], write_x=False, out_dtype=out_dtype, delta2=pending[1])
        out = out.view(B, 1 + T * n, D)                                                      # norm evaluated once (A5)
        if split_cls:
            return self.pre_logits(out[:, 0]), out[:, 1:].contiguous()
        return self.pre_logits(out[:, 0]), out

    def forward(self, x, use_checkpoint=False):
        x_cls, x = self.forward_features(x, use_checkpoint=use_checkpoint)
        return self.head(x_cls), x


class CLIP(nn.Module):
    """LaviLa.py:586-687: dual encoder wrapper; forward returns the reference's dict."""

    def __init__(self, embed_dim: int, vision_width: int, vision_model: nn.Module, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int, tempearture_init=0.07, **kwargs):
        super().__init__()
        self.context_length = context_length
        self.vision_width = vision_width
        self.visual = vision_model
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.image_projection = nn.Parameter(torch.empty(vision_width, embed_dim))
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / tempearture_init))
        self.initialize_parameters()

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, L = self.transformer.width, self.transformer.layers
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=w ** -0.5)
            nn.init.normal_(block.attn.out_proj.weight, std=(w ** -0.5) * ((2 * L) ** -0.5))
            nn.init.normal_(block.mlp.c_fc.weight, std=(2 * w) ** -0.5)
            nn.init.normal_(block.mlp.c_proj.weight, std=(w ** -0.5) * ((2 * L) ** -0.5))
        nn.init.normal_(self.image_projection, std=self.vision_width ** -0.5)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    def build_attention_mask(self):
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    def encode_image(self, image, use_checkpoint=False, apply_project=True):
        x_cls, x = self.visual(image, use_checkpoint=use_checkpoint)
        if not apply_project:
            return x_cls, x
        return x_cls @ self.image_projection, x

    def encode_text(self, text, use_checkpoint=False, apply_project=True, want_cls=True):
        """LaviLa.py:660-670 (apply_project=False, as encode_image has it, skips the EOT-token projection: the training step only
        uses the feature map).  The backbone is frozen on this path (run/train.py:89,109-116; run/test_EgoMCQ.py:60 under no_grad):
        Linears and LayerNorms of the 12 text blocks run on libhh kernels (`Transformer.forward_frozen`), the 77 x 77 causal core on
        hh_text_attn_fwd.  There is no stock-op branch: a text tower that wants gradients is not part of this path and raises."""
        _require_gpu(text, "CLIP.encode_text")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.transformer.parameters()):
            raise NotImplementedError("CLIP.encode_text: the text tower is frozen on the hot path (run/train.py:89 freezes the backbone; optim_policy, "
                                      "utils/train_utils.py:42-43, gives it no optimizer group); a trainable text tower has no libhh backward and there is "
                                      "no stock-op fallback -- call under torch.no_grad() or freeze the parameters")
        with torch.no_grad(), ops.prof_role(ops.PROF_ROLE_TEXT):
            x = self.token_embedding(text) + self.positional_embedding[:text.shape[1]]
            x = self.transformer.forward_frozen(x, final_ln=(self.ln_final.weight.detach().float(), self.ln_final.bias.detach().float(),
                                                             self.ln_final.eps), owns_x=True)
        if not want_cls and not apply_project:          # (the training step gathers the EOT rows itself, run/train.py:124: four launches saved)
            return None, x
        x_cls = x[torch.arange(x.shape[0], device=x.device), text.float().argmax(dim=-1)]      # (ids < 2^24: exact; the int64 ArgMax reduce is 20x slower)
        if not apply_project:
            return x_cls, x
        return ops.qgemm(x_cls.contiguous(), self.text_projection.detach().float().contiguous(), ops.NN), x

    def forward(self, image, text, use_checkpoint=False, norm_embed=True, return_feature_map=False):
        image_embed, image_fmap = self.encode_image(image, use_checkpoint=use_checkpoint)
        text_embed, text_fmap = self.encode_text(text, use_checkpoint=use_checkpoint)
        if norm_embed:
            image_embed = F.normalize(image_embed, dim=-1)
            text_embed = F.normalize(text_embed, dim=-1)
        out = {'image_embed': image_embed, 'text_embed': text_embed, 'logit_scale': self.logit_scale.exp()}
        if return_feature_map:
            out['image_feature_map'] = image_fmap
            out['text_feature_map'] = text_fmap
        return out


def CLIP_OPENAI_TIMESFORMER_LARGE(num_frames=4, timesformer_gated_xattn=False, drop_path_rate=0, timesformer_freeze_space=False,
                                  temperature_init=0.07, project_embed_dim=256, use_adapter=False, **kwargs):
    """The reference's factory name and arguments (LaviLa.py:114-172; called at run/train.py:425-431 and
    run/test_EgoMCQ.py:206-212 with pretrained=None, text_use_cls_token=False, project_embed_dim=256, num_frames=4,
    temperature_init=0.07), so those scripts import and call it unchanged.

    It builds the same CLIP(SpaceTimeTransformer-L/14, 12x768 text tower) on the CPU.  The one thing it does NOT do is the
    reference's download of OpenAI CLIP ViT-L/14 weights (LaviLa.py:130-133,163-171; there is no network here): parameters
    keep their constructor initialisation, and both call sites overwrite every one of them right afterwards with
    `backbone.load_state_dict(<LaViLa checkpoint>, strict=True)` (run/train.py:433-439).  Move the result to the GPU with
    `.to(device)` as the scripts do; the libhh kernels are used from there on."""
    if timesformer_freeze_space:
        raise NotImplementedError("CLIP_OPENAI_TIMESFORMER_LARGE: timesformer_freeze_space needs the CLIP key remap of the "
                                  "downloaded weights (LaviLa.py:134-146); the hot path freezes the whole backbone instead")
    vision_model = SpaceTimeTransformer(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=num_frames,
                                        time_init='zeros', attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU,
                                        is_tanh_gating=timesformer_gated_xattn, drop_path_rate=drop_path_rate,
                                        use_adapter=use_adapter)
    vision_model.head = nn.Identity()
    vision_model.pre_logits = nn.Identity()
    vision_model.fc = nn.Identity()
    return CLIP(embed_dim=project_embed_dim, vision_width=1024, vision_model=vision_model, context_length=77, vocab_size=49408,
                transformer_width=768, transformer_heads=12, transformer_layers=12, tempearture_init=temperature_init, **kwargs)


def build_backbone(cfg, state_dict=None, device="cuda"):
    """Construct CLIP(SpaceTimeTransformer) with the arguments of LaviLa.py:118-129,148-162, bypassing the
    network-bound factory; optionally load a reference-keyed state dict."""
    vis = SpaceTimeTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                               num_heads=cfg.num_heads, num_frames=cfg.num_frames, time_init='zeros',
                               attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU)
    vis.head = nn.Identity()
    vis.pre_logits = nn.Identity()
    vis.fc = nn.Identity()
    model = CLIP(embed_dim=cfg.project_embed_dim, vision_width=cfg.embed_dim, vision_model=vis,
                 context_length=cfg.context_length, vocab_size=cfg.vocab_size, transformer_width=cfg.text_width,
                 transformer_heads=cfg.text_heads, transformer_layers=cfg.text_layers, tempearture_init=0.07)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        if unexpected or any(not k.endswith("attn_mask") for k in missing):
            raise RuntimeError(f"backbone state dict mismatch: missing={missing} unexpected={unexpected}")
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad_(False)          # optim_policy freezes the backbone (utils/train_utils.py:42-43)
    return model
