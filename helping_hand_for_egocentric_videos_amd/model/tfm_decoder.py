"""Object-query decoder (trainable) on libhh kernels: mirror of /root/reference/model/tfm_decoder.py.

Classes / ctor arguments / forward signatures / state_dict keys follow the reference: `ObjDecoder` (:111-241),
`Cross_Attention` (:50-93), `TransformerDecoder` (:246-295), `TransformerDecoderLayer` (:358-479, the live path
is forward_pre with sa_first=True :430-461), `MLP` (:96-108).  Dead code of the reference (PositionEmbeddingSine,
TransformerEncoderLayer, forward_post -- unusable) is not built.

MI355X design:
  * memory side (99 % of decoder FLOPs): proj GEMM -> fused LayerNorm -> ONE batched K/V in-projection of the
    4096 memory tokens for all 6 layers ([B*M,512] x [512, 6*1024], memory is layer-invariant) on the MFMA GEMM;
    backward = two dgrad GEMMs (second one accumulates through the fp32 residual epilogue), split-K wgrad GEMMs
    and the LayerNorm backward kernel.  K/V and their gradients live in one [B*M, 6144] bf16 buffer each; the six
    cross-attention backward kernels write their column slices directly (no autograd slice/accumulate copies).
  * cross-attention core: hh_xattn_fwd / hh_xattn_bwd (13 x 4096 per clip and head), attention-dropout inside.
  * query side (13 rows per clip, < 1 % of FLOPs): model/qside.py -- ONE autograd node for the six layers on hh_qgemm_f32x3
    (fp32-grade GEMM on the bf16 matrix cores), hh_qself_attn_*, LayerNorm kernels; heads on LinearX3.
  * the frame-conditioning `cat[hs, frame_index] -> frame_proj` is evaluated in its exact decomposed form
    hs.W[:, :C]^T + (frame_index.W[:, C:]^T + b)[t]  (T x fewer FLOPs).
"""
import copy
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .. import ops
from .qside import LinearX3, QueryStack, linear_x3, stack_params


def _require_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the product path runs on libhh HIP kernels only (got a {x.device} tensor); "
                           "there is no CPU fallback -- use oracle/ for CPU reference numbers")


class _KVHolder:
    """What the memory side hands the six cross-attention layers, and what they hand back.

    kv_free (default, round 5): `mem` = pre_norm(proj(x)) and `mp` = mem + pos, bf16 [B, M, C] -- the un-projected rows every layer and
    head attends over (csrc/mattn.hip); the backward of the layers leaves Pd^T / dS^T (`pdT`, `dsT`: bf16 [B, L*128, M]) and the bf16
    mapped queries / pooled-row gradients (`qt16`, `dp16`: bf16 [B, L*128, C]) for ONE batched GEMM in _MemorySideKVFree.backward.
    Otherwise (the round-1..4 path, kept for A/B measurements and for decoders other than 512 / 8 heads): K/V of all layers
    [B*M, 2*L*C] bf16 (K columns first) and its gradient buffer."""

    def __init__(self):
        self.kv = None
        self.dkv = None
        self.mem = self.mp = None
        self.pdT = self.dsT = self.qt16 = self.dp16 = None
        self.B = self.M = self.C = self.L = 0
        self.Mv = 0                      # kv_free: the clip's real memory length T * n when M was rounded up to a multiple of 128 (else 0 = all M rows are keys)
        self.seed = 0
        self.keep, self.kept = False, None
        self.relu_masks = []             # keep only: per layer, the active units of the FFN hidden layer [B*Q, ffn] (tests/test_step_gpu.py)
        self.q_rows_sunk = False         # QueryStack.backward wrote the query rows of every cross-attention in-projection gradient into the arena

    def bwd_buffers(self):
        """Pd^T / dS^T / bf16 operand rows of all layers, allocated by the first layer whose backward runs (every layer writes its own 128
        rows -- all 16 query slots of all 8 heads, zeros beyond Q -- so nothing needs clearing)."""
        if self.pdT is None:
            dev, rows = self.mem.device, self.L * 128
            self.pdT = torch.empty((self.B, rows, self.M), dtype=torch.bfloat16, device=dev)
            self.dsT = torch.empty_like(self.pdT)
            self.qt16 = torch.empty((self.B, rows, self.C), dtype=torch.bfloat16, device=dev)
            self.dp16 = torch.empty_like(self.qt16)
        return self.pdT, self.dsT, self.qt16, self.dp16


class _MemorySide(torch.autograd.Function):
    """features -> proj -> pre_norm -> (+pos) -> K/V in-projection of every layer.  Returns a 1-element token that
    the cross-attention nodes consume, so that this node's backward runs after all of them have written holder.dkv."""

    @staticmethod
    def forward(ctx, feat_b, w_proj, g_pre, b_pre, pos, holder, eps, *in_wb):
        L = len(in_wb) // 2
        in_w, in_b = in_wb[:L], in_wb[L:]
        C = w_proj.shape[0]
        BM = feat_b.shape[0]
        M = pos.shape[0]
        B = BM // M
        mem0 = ops.gemm(feat_b, ops.to_bf16(w_proj.detach()), out_dtype=torch.float32)                   # [BM, C]
        # pre_norm and the key operand memory + pos in ONE pass (tfm_decoder.py:86-88,438-441: key = memory + pos, value = memory)
        memory, mem_pos, mean, rstd = ops.layernorm_pos(mem0, g_pre.detach().float(), b_pre.detach().float(), eps,
                                                        pos.detach().float().contiguous(), out_dtype=torch.bfloat16, save_stats=True)
        wk = torch.cat([w.detach()[C:2 * C] for w in in_w])                                               # [L*C, C]
        wv = torch.cat([w.detach()[2 * C:] for w in in_w])
        bk = torch.cat([b.detach()[C:2 * C] for b in in_b]).float()
        bv = torch.cat([b.detach()[2 * C:] for b in in_b]).float()
        kv = torch.empty((BM, 2 * L * C), dtype=torch.bfloat16, device=feat_b.device)
        ops.gemm(mem_pos, ops.to_bf16(wk), bk, out=kv[:, :L * C])
        ops.gemm(memory, ops.to_bf16(wv), bv, out=kv[:, L * C:])
        holder.kv, holder.dkv = kv, None
        holder.B, holder.M, holder.C, holder.L = B, M, C, L
        ctx.holder, ctx.L, ctx.C = holder, L, C
        ctx.param_objs = (w_proj, g_pre, b_pre) + tuple(in_wb)   # the Parameters themselves: their gradient sinks (parallel._GradSink), if any
        ctx.save_for_backward(feat_b, mem0, mean, rstd, memory, mem_pos, g_pre, wk, wv)
        return torch.zeros(1, dtype=torch.float32, device=feat_b.device)

    @staticmethod
    def backward(ctx, _gtoken):
        feat_b, mem0, mean, rstd, memory, mem_pos, g_pre, wk, wv = ctx.saved_tensors
        h, L, C = ctx.holder, ctx.L, ctx.C
        BM = feat_b.shape[0]
        if h.dkv is None:                      # no cross-attention gradient reached us
            h.dkv = torch.zeros_like(h.kv)
        dk_all, dv_all = h.dkv[:, :L * C], h.dkv[:, L * C:]
        # dgrad: d(mem_pos) = dK.Wk ; d(memory) = dV.Wv + d(mem_pos)
        dmem_pos = ops.gemm(dk_all, ops.transpose_bf16(wk), out_dtype=torch.float32)                      # [BM, C]
        dmem = ops.gemm(dv_all, ops.transpose_bf16(wv), resid=dmem_pos, out_dtype=torch.float32)
        dpos = dmem_pos.view(h.B, h.M, C).sum(0)
        # wgrad (contraction over the B*M tokens): dY and X are token-major as they sit in memory = the k-major operands of the
        # TN kernel (hh_gemm_tn_bf16, split-K over tokens) -- no transposed copies
        # (the bias gradients sum_tokens dK / dV are column sums of the TN kernel's A operand: a by-product instead of two more passes
        # over the 1.6 GB of dK / dV)
        dwk, dbk = ops.gemm_tn(dk_all, mem_pos, colsum=True)                                               # [L*C, C], [L*C]
        dwv, dbv = ops.gemm_tn(dv_all, memory, colsum=True)
        del dmem_pos
        dmem0, dg, db = ops.layernorm_bwd(mem0, g_pre.detach().float(), mean, rstd, dmem)
        dw_proj = ops.gemm_tn(ops.to_bf16(dmem0), feat_b)                                                  # [C, F]
        if h.keep:
            h.kept = (h.kv, h.dkv)                 # test hook (Cross_Attention.debug_keep_kv): K/V and dK/dV outlive the backward
        h.kv = h.dkv = None
        sinks = [getattr(t, "_hh_sink", None) for t in ctx.param_objs]
        if h.q_rows_sunk and all(s is not None and s.armed() for s in sinks):
            # gradient sinks (parallel._GradSink): QueryStack.backward has already written the query rows of every in-projection
            # gradient into the flat arena; the key / value rows, the memory projection and pre_norm follow in ONE fused multi-tensor
            # copy, the parameters are reported final and autograd gets None (no per-parameter `grad += new`, no zero-padded temporaries)
            dst = [sinks[0].view, sinks[1].view, sinks[2].view]
            src = [dw_proj, dg, db]
            for l in range(L):
                vw, vb = sinks[3 + l].view, sinks[3 + L + l].view
                dst += [vw[C:2 * C], vw[2 * C:], vb[C:2 * C], vb[2 * C:]]
                src += [dwk[l * C:(l + 1) * C], dwv[l * C:(l + 1) * C], dbk[l * C:(l + 1) * C], dbv[l * C:(l + 1) * C]]
            for s_ in sinks:
                s_.claim()
            torch._foreach_copy_(dst, src)
            for s_ in sinks:
                s_.done()
            return (None, None, None, None, dpos, None, None) + (None,) * (2 * L)
        if h.q_rows_sunk:                          # the query rows are in the arena, the rest goes through autograd after all: hand the
            for s_ in sinks[3:]:                   # in-projections back to their post-accumulate hooks (they add into the same slices)
                if s_ is not None:
                    s_.unclaim()
        gw, gb = [], []
        for l in range(L):
            w = torch.zeros((3 * C, C), dtype=torch.float32, device=feat_b.device)
            w[C:2 * C] = dwk[l * C:(l + 1) * C]
            w[2 * C:] = dwv[l * C:(l + 1) * C]
            b = torch.zeros((3 * C,), dtype=torch.float32, device=feat_b.device)
            b[C:2 * C] = dbk[l * C:(l + 1) * C]
            b[2 * C:] = dbv[l * C:(l + 1) * C]
            gw.append(w)
            gb.append(b)
        return (None, dw_proj, dg, db, dpos, None, None, *gw, *gb)


class _MemorySideKVFree(torch.autograd.Function):
    """features -> proj -> pre_norm -> (memory, memory + pos) and NOTHING else: no key / value projection of the M memory tokens
    (tfm_decoder.py:438-441 evaluated in memory space, csrc/mattn.hip).  Returns a 1-element token the query stack consumes, so that
    this node's backward runs after the layers' backward has left Pd^T / dS^T of every layer in the holder:
        d memory = sum_layers Pd^T dpooled + dS^T qt          ONE batched TN GEMM over [6 x 128 rows] per clip
        d pos    = sum_clips  dS^T qt                          one split-K TN GEMM over [B x 6 x 128 rows]
    then the LayerNorm backward and the weight gradient of the memory projection as before."""

    @staticmethod
    def forward(ctx, feat_b, w_proj, g_pre, b_pre, pos, holder, eps, L):
        C = w_proj.shape[0]
        BM = feat_b.shape[0]
        M = pos.shape[0]
        B = BM // M
        mem0 = ops.gemm(feat_b, ops.to_bf16(w_proj.detach()), out_dtype=torch.float32)                   # [BM, C]
        memory, mem_pos, mean, rstd = ops.layernorm_pos(mem0, g_pre.detach().float(), b_pre.detach().float(), eps,
                                                        pos.detach().float().contiguous(), out_dtype=torch.bfloat16, save_stats=True)
        Mp = (M + 127) // 128 * 128
        if Mp == M:
            holder.mem, holder.mp, holder.Mv = memory.view(B, M, C), mem_pos.view(B, M, C), 0
        else:
            # any memory length (round 6; e.g. patch 16 -> n = 196, M = 3136): the rows are rounded up to the 128-row tile of the batched
            # d-memory GEMM with ZERO rows that hh_mattn_* masks out of every softmax (keys_valid) -- one more copy of the rows, paid only
            # by shapes that are not multiples of 128; their Pd^T / dS^T columns are exact zeros, their d memory rows are dropped below
            holder.mem = torch.zeros((B, Mp, C), dtype=torch.bfloat16, device=feat_b.device)
            holder.mp = torch.zeros_like(holder.mem)
            holder.mem[:, :M].copy_(memory.view(B, M, C))
            holder.mp[:, :M].copy_(mem_pos.view(B, M, C))
            holder.Mv = M
        holder.B, holder.M, holder.C, holder.L = B, Mp, C, L
        ctx.M_valid = M
        ctx.holder = holder
        ctx.param_objs = (w_proj, g_pre, b_pre)
        ctx.save_for_backward(feat_b, mem0, mean, rstd, g_pre)
        return torch.zeros(1, dtype=torch.float32, device=feat_b.device)

    @staticmethod
    def backward(ctx, _gtoken):
        feat_b, mem0, mean, rstd, g_pre = ctx.saved_tensors
        h = ctx.holder
        B, Mp, C, L = h.B, h.M, h.C, h.L
        M = ctx.M_valid
        if h.pdT is None:                      # no cross-attention gradient reached us
            dmem = torch.zeros((B * M, C), dtype=torch.float32, device=feat_b.device)
            dpos = torch.zeros((M, C), dtype=torch.float32, device=feat_b.device)
        else:
            dmem = ops.gemm_tn_batched2(h.pdT, h.dp16, h.dsT, h.qt16)                                    # d(memory) incl. the key path  [B, Mp, C]
            dpos = ops.gemm_tn(h.dsT.view(B * L * 128, Mp), h.qt16.view(B * L * 128, C))                 # d(pos) = sum over clips of the key path
            if Mp != M:                        # (padding rows: exact zeros)
                dmem, dpos = dmem[:, :M].contiguous(), dpos[:M].contiguous()
            dmem = dmem.view(B * M, C)
        # gradient sinks (parallel._GradSink): the LayerNorm reductions accumulate into, and the projection's split-K partials are summed
        # into, the parameters' (zeroed) slices of the gradient arena -- no zero fills, no copies, no AccumulateGrad adds
        sinks = [getattr(t, "_hh_sink", None) for t in ctx.param_objs]
        sunk = all(s is not None and s.armed() for s in sinks)
        if sunk:
            for s_ in sinks:
                s_.claim()
        dmem0, dg, db = ops.layernorm_bwd(mem0, g_pre.detach().float(), mean, rstd, dmem, dg=sinks[1].view if sunk else None,
                                          db=sinks[2].view if sunk else None)
        dw_proj = ops.gemm_tn(ops.to_bf16(dmem0), feat_b, out=sinks[0].view if sunk else None)             # [C, F]
        if h.keep:
            h.kept = {"mem": h.mem[:, :M], "mp": h.mp[:, :M], "dmem": dmem, "dpos": dpos, "relu_masks": h.relu_masks}        # test hook (Cross_Attention.debug_keep_kv)
        h.mem = h.mp = h.pdT = h.dsT = h.qt16 = h.dp16 = None
        if sunk:
            for s_ in sinks:
                s_.done()
            return (None, None, None, None, dpos, None, None, None)
        return (None, dw_proj, dg, db, dpos, None, None, None)


class _LinearBF16(torch.autograd.Function):
    """y = x.W^T + b on hh_gemm_bf16 (bf16 operands, fp32 accumulation); backward = dgrad GEMM + TN weight-gradient GEMM.
    x [rows,K] fp32/bf16, w [N,K], b [N] or None.  Used by the reference-signature layer forward, where every layer projects
    its own K/V (the batched path projects all layers at once in _MemorySide)."""

    @staticmethod
    def forward(ctx, x, w, b, out_dtype):
        xb = ops.to_bf16(x.detach().contiguous())
        y = ops.gemm(xb, ops.to_bf16(w.detach().contiguous()), None if b is None else b.detach().float().contiguous(), out_dtype=out_dtype)
        ctx.save_for_backward(xb, w)
        ctx.has_bias, ctx.x_dtype = b is not None, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        dyb = ops.to_bf16(dy.contiguous())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dyb, ops.transpose_bf16(w.detach().contiguous()), out_dtype=torch.float32).to(ctx.x_dtype)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(dyb, xb, colsum=want_db)
            if want_db:
                dw, db = dw
        elif want_db:
            db = torch.sum(dyb, dim=0, dtype=torch.float32)
        return dx, dw, db, None


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dim on hh_layernorm_fwd / hh_layernorm_bwd; x fp32 [rows, C] -> fp32."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = x.detach().float().contiguous()
        y, mean, rstd = ops.layernorm(x, g.detach().float().contiguous(), b.detach().float().contiguous(), eps,
                                      out_dtype=torch.float32, save_stats=True)
        ctx.save_for_backward(x, g, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, g.detach().float().contiguous(), mean, rstd, dy.float().contiguous())
        return dx, dg, db, None


class _XAttnKV(torch.autograd.Function):
    """softmax(q.K^T).V on explicit K/V tensors (bf16 [B,M,C], contiguous); q fp32 [B,Q,C] pre-scaled."""

    @staticmethod
    def forward(ctx, q, k, v, heads, dropout_p, seed):
        q = q.detach().contiguous()
        out, lse = ops.xattn_fwd(q, k.detach(), v.detach(), heads, dropout_p, seed)
        ctx.heads, ctx.p, ctx.seed = heads, dropout_p, seed
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        dq = ops.xattn_bwd(q, k.detach(), v.detach(), out, lse, dout.contiguous(), dk, dv, ctx.heads, ctx.p, ctx.seed)
        return dq, dk, dv, None, None, None


def _draw_seed():
    """31-bit seed for the cross-attention dropout mask, drawn from torch's CPU generator (follows torch.manual_seed; no GPU sync)."""
    return int(torch.randint(0, 0x7FFFFFFF, (1,)).item())


class _PosEmbed3D(torch.autograd.Function):
    """construct_3d_pos_embed (tfm_decoder.py:138-143) as ONE node: pos[t * n + i] = pos_embed[0, 1 + i] + temporal_embed[0, t] -- one
    broadcast add forward; backward = two reductions written straight into the parameters' slices of the gradient arena
    (parallel._GradSink; the CLS row of pos_embed and the frames behind T get no gradient: their arena rows stay zero).  The stock-op
    chain it replaces (repeat, repeat_interleave, add; RepeatBackward / ExpandBackward sums, slice fills and copies, two
    AccumulateGrad adds) was 9 launches per step."""

    @staticmethod
    def forward(ctx, pos_embed, temporal_embed, T):
        n, C = pos_embed.shape[1] - 1, pos_embed.shape[2]
        ctx.param_objs, ctx.dims = (pos_embed, temporal_embed), (T, n, C)
        return (pos_embed.detach()[0, 1:].unsqueeze(0) + temporal_embed.detach()[0, :T].unsqueeze(1)).view(T * n, C)

    @staticmethod
    def backward(ctx, d):
        T, n, C = ctx.dims
        d3 = d.reshape(T, n, C)
        pe, te = ctx.param_objs
        sinks = [getattr(t, "_hh_sink", None) for t in ctx.param_objs]
        if all(s is not None and s.armed() for s in sinks):
            for s_ in sinks:
                s_.claim()
            torch.sum(d3, dim=0, out=sinks[0].view[0, 1:])
            torch.sum(d3, dim=1, out=sinks[1].view[0, :T])
            for s_ in sinks:
                s_.done()
            return None, None, None
        dpe, dte = torch.zeros_like(pe), torch.zeros_like(te)
        torch.sum(d3, dim=0, out=dpe[0, 1:])
        torch.sum(d3, dim=1, out=dte[0, :T])
        return dpe, dte, None


class _SplitRows(torch.autograd.Function):
    """y [R, N] -> (y[:n], y[n:]): two row blocks of one GEMM's output (step.py projects captions and nouns through txt_proj in one call);
    backward = one concatenation instead of two zero-filled SliceBackward temporaries and their sum."""

    @staticmethod
    def forward(ctx, y, n):
        ctx.n, ctx.shape = n, y.shape
        ctx.set_materialize_grads(False)
        yd = y.detach()
        return yd[:n], yd[n:]

    @staticmethod
    def backward(ctx, g1, g2):
        n, shape = ctx.n, ctx.shape
        if g1 is None and g2 is None:
            return None, None
        ref = g1 if g1 is not None else g2
        g1 = ref.new_zeros((n,) + tuple(shape[1:])) if g1 is None else g1
        g2 = ref.new_zeros((shape[0] - n,) + tuple(shape[1:])) if g2 is None else g2
        return torch.cat([g1, g2]), None


class _SplitLastQuery(torch.autograd.Function):
    """obj [B, Q, E] -> (obj[:, :Q-1], obj[:, Q-1]): the object queries' embeddings and the video query's (run/train.py:131-133);
    backward = one concatenation instead of SelectBackward / SliceBackward zero fills, copies and their sum."""

    @staticmethod
    def forward(ctx, obj):
        ctx.shape = obj.shape
        ctx.set_materialize_grads(False)
        od = obj.detach()
        return od[:, :-1], od[:, -1]

    @staticmethod
    def backward(ctx, g_rest, g_last):
        B, Q, E = ctx.shape
        if g_rest is None and g_last is None:
            return None
        ref = g_rest if g_rest is not None else g_last
        g_rest = ref.new_zeros((B, Q - 1, E)) if g_rest is None else g_rest
        g_last = ref.new_zeros((B, E)) if g_last is None else g_last
        return torch.cat([g_rest, g_last[:, None]], dim=1)


class _FirstRows(torch.autograd.Function):
    """w[:T] of a parameter whose only use in the step this is (frame_index.weight, tfm_decoder.py:196-203); backward copies the rows'
    gradient into the parameter's slice of the gradient arena (the other rows stay zero) instead of zero fill + copy + AccumulateGrad."""

    @staticmethod
    def forward(ctx, w, T):
        ctx.param_objs, ctx.T = (w,), T
        return w.detach()[:T]

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.param_objs
        sink = getattr(w, "_hh_sink", None)
        if sink is not None and sink.armed():
            sink.claim()
            sink.view[:ctx.T].copy_(g)
            sink.done()
            return None, None
        dw = torch.zeros_like(w)
        dw[:ctx.T] = g
        return dw, None


class _SplitCols(torch.autograd.Function):
    """w [N, 2C] -> (w[:, :C], w[:, C:]) for the decomposed frame conditioning (ObjDecoder.forward); backward writes the two column
    halves' gradients into the parameter's slice of the gradient arena (two copies instead of SliceBackward's zero-fill + copy per
    half, their sum and the AccumulateGrad add)."""

    @staticmethod
    def forward(ctx, w, C):
        ctx.param_objs, ctx.C = (w,), C
        wd = w.detach()
        return wd[:, :C], wd[:, C:]

    @staticmethod
    def backward(ctx, g1, g2):
        C = ctx.C
        (w,) = ctx.param_objs
        sink = getattr(w, "_hh_sink", None)
        if sink is not None and sink.armed() and g1 is not None and g2 is not None:
            sink.claim()
            sink.view[:, :C].copy_(g1)
            sink.view[:, C:].copy_(g2)
            sink.done()
            return None, None
        dw = torch.zeros_like(w)
        if g1 is not None:
            dw[:, :C] = g1
        if g2 is not None:
            dw[:, C:] = g2
        return dw, None


class MLP(nn.Module):
    """Simple multi-layer perceptron (tfm_decoder.py:96-108)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(LinearX3(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = linear_x3(x, layer.weight, layer.bias, relu=i < self.num_layers - 1,             # ReLU in the GEMM epilogue
                          sink_ok=getattr(self, "single_use", False))
        return x


class TransformerDecoderLayer(nn.Module):
    """tfm_decoder.py:358-479.  Only normalize_before=True / sa_first=True (the configuration run/train.py builds)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False,
                 sa_first=True):
        super().__init__()
        self.sa_first = sa_first
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = LinearX3(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = LinearX3(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        if activation != "relu":
            raise NotImplementedError("TransformerDecoderLayer: relu only (reference default)")
        self.activation = F.relu
        self.normalize_before = normalize_before
        self.nhead, self.p_attn = nhead, dropout
        if d_model // nhead != 64:
            raise NotImplementedError("TransformerDecoderLayer: libhh cross-attention is specialised for head_dim 64")

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return tensor if pos is None else tensor + pos

    def _self_attention(self, x, qpos):
        """nn.MultiheadAttention(q = k = x + qpos, v = x) on [B,Q,C] with stock ops (13 x 13 per clip)."""
        m = self.self_attn
        C, h = x.shape[-1], self.nhead
        wq, wk, wv = m.in_proj_weight.chunk(3, dim=0)
        bq, bk, bv = m.in_proj_bias.chunk(3, dim=0)
        xq = x + qpos
        B, Q = x.shape[:2]
        q = linear_x3(xq, wq, bq).view(B, Q, h, C // h).transpose(1, 2)
        k = linear_x3(xq, wk, bk).view(B, Q, h, C // h).transpose(1, 2)
        v = linear_x3(x, wv, bv).view(B, Q, h, C // h).transpose(1, 2)
        p = torch.softmax((q * (C // h) ** -0.5) @ k.transpose(-1, -2), dim=-1)
        p = F.dropout(p, self.p_attn, self.training)
        o = (p @ v).transpose(1, 2).reshape(B, Q, C)
        return linear_x3(o, m.out_proj.weight, m.out_proj.bias)

    def _attend(self, tgt, qpos, xattn):
        """forward_pre with sa_first (tfm_decoder.py:430-461) on batch-first fp32 [B,Q,C]; `xattn(q)` is the cross-attention core
        of this layer (q already projected and scaled)."""
        C, h = tgt.shape[-1], self.nhead
        a = self.norm1(tgt)
        tgt = tgt + self.dropout1(self._self_attention(a, qpos))
        c = self.norm2(tgt)
        m = self.multihead_attn
        q = linear_x3(c + qpos, m.in_proj_weight[:C], m.in_proj_bias[:C]) * ((C // h) ** -0.5)
        tgt = tgt + self.dropout2(linear_x3(xattn(q), m.out_proj.weight, m.out_proj.bias))
        e = self.norm3(tgt)
        return tgt + self.dropout3(self.linear2(self.dropout(linear_x3(e, self.linear1.weight, self.linear1.bias, relu=True))))

    def forward_pre(self, tgt, memory, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                    tgt_key_padding_mask: Optional[Tensor] = None, memory_key_padding_mask: Optional[Tensor] = None,
                    pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None, counter: Optional[Tensor] = None,
                    num_frames: Optional[int] = 4, seq_len: Optional[int] = 196):
        """Reference signature (tfm_decoder.py:420-461), sequence-first: tgt [Q,B,C], memory [M,B,C], pos [M,1|B,C],
        query_pos [Q,B,C] -> (tgt [Q,B,C], attn, self_attn).  This layer projects its own K/V of `memory` (the batched path,
        Cross_Attention.forward_tokens, projects all layers in one GEMM); the head-averaged attention maps the reference returns
        and discards are None."""
        _require_gpu(tgt, "TransformerDecoderLayer")
        if not self.sa_first:
            raise NotImplementedError("TransformerDecoderLayer: sa_first=True only (the reference default, tfm_decoder.py:54)")
        if tgt_mask is not None or memory_mask is not None or tgt_key_padding_mask is not None:
            raise NotImplementedError("TransformerDecoderLayer: attention masks are None on the hot path (tfm_decoder.py:270-276)")
        if memory_key_padding_mask is not None and bool(memory_key_padding_mask.any()):
            raise NotImplementedError("TransformerDecoderLayer: the key padding mask is all-False on the hot path (tfm_decoder.py:203)")
        Q, B, C = tgt.shape
        M = memory.shape[0]
        x = tgt.transpose(0, 1).float()
        qpos = torch.zeros_like(x) if query_pos is None else query_pos.transpose(0, 1).float()
        mem_b = memory.transpose(0, 1).reshape(B * M, C)
        kin = mem_b if pos is None else (memory + pos).transpose(0, 1).reshape(B * M, C)
        m = self.multihead_attn
        k = _LinearBF16.apply(kin, m.in_proj_weight[C:2 * C], m.in_proj_bias[C:2 * C], torch.bfloat16).view(B, M, C)
        v = _LinearBF16.apply(mem_b, m.in_proj_weight[2 * C:], m.in_proj_bias[2 * C:], torch.bfloat16).view(B, M, C)
        p = self.p_attn if self.training else 0.0
        seed = _draw_seed() if p > 0 else 0
        out = self._attend(x, qpos, lambda q: _XAttnKV.apply(q, k, v, self.nhead, p, seed))
        return out.transpose(0, 1), None, None

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, memory_key_padding_mask: Optional[Tensor] = None,
                pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None, counter: Optional[Tensor] = None,
                num_frames: Optional[int] = 4, seq_len: Optional[int] = 196):
        """tfm_decoder.py:463-479."""
        if not self.normalize_before:
            raise NotImplementedError("TransformerDecoderLayer: forward_post is unusable in the reference (it returns one value "
                                      "where TransformerDecoder.forward unpacks three, SURVEY A7)")
        return self.forward_pre(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos, query_pos,
                                counter, num_frames, seq_len)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerDecoder(nn.Module):
    """tfm_decoder.py:246-295."""

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None, memory_key_padding_mask: Optional[Tensor] = None,
                pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None, num_frames: Optional[int] = 4,
                seq_len: Optional[int] = 196):
        """Reference signature (tfm_decoder.py:255-295), sequence-first: -> (stack of normed intermediates [L,Q,B,C] or
        norm(output)[None], [], []).  Each layer projects its own K/V; ObjDecoder.forward uses the batched forward_tokens."""
        output = tgt
        intermediate = []
        for layer_i, layer in enumerate(self.layers):
            output, _, _ = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                                 tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask,
                                 pos=pos, query_pos=query_pos, counter=layer_i, num_frames=num_frames, seq_len=seq_len)
            if self.return_intermediate:
                intermediate.append(self.norm(output))
        if self.norm is not None:
            output = self.norm(output)
            if self.return_intermediate:
                intermediate.pop()
                intermediate.append(output)
        if self.return_intermediate:
            return torch.stack(intermediate), [], []
        return output.unsqueeze(0), [], []


class Cross_Attention(nn.Module):
    """tfm_decoder.py:50-93."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, hidden_dim=768, return_intermediate_dec=False,
                 sa_first=True):
        super().__init__()
        if not normalize_before or not sa_first:
            raise NotImplementedError("Cross_Attention: only normalize_before=True, sa_first=True is usable in the "
                                      "reference (forward_post returns 1 value where 3 are unpacked, SURVEY A7)")
        self.pre_norm = nn.LayerNorm(d_model)
        layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before, sa_first=sa_first)
        self.decoder = TransformerDecoder(layer, num_decoder_layers, nn.LayerNorm(d_model), return_intermediate=return_intermediate_dec)
        self._reset_parameters()
        self.d_model, self.nhead = d_model, nhead
        self.dec_layers, self.enc_layers = num_decoder_layers, num_encoder_layers
        # kv_free (default): the cross-attention runs in memory space (csrc/mattn.hip) -- no K/V projection of the memory tokens -- for EVERY
        # memory length (round 6: rows padded to a multiple of 128 and masked).  False: the round-1..4 path (one batched K/V in-projection for
        # all layers + hh_xattn_*), kept for same-session A/B measurements; it also serves decoders that are not the reference's 512 / 8 heads /
        # <= 16 queries, which hh_mattn_* is built for (tfm_decoder.py:51)
        self.kv_free = True
        self.debug_keep_kv, self.last_holder = False, None        # tests: keep the memory rows / K/V and their gradients after backward
        self._seed = None            # dropout-mask stream of the cross-attention kernels; see next_dropout_seed()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def next_dropout_seed(self):
        """Per-step seed of the attention-dropout masks (tfm_decoder.py:365, p = 0.1): an LCG stream that starts from
        torch.initial_seed() mixed with the data-parallel rank (so torch.manual_seed controls it and ranks draw different masks);
        TrainStep.state_dict() persists `_seed` so that a resumed run continues the stream."""
        if self._seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._seed = (torch.initial_seed() * 2654435761 + 40503 * rank + 1) & 0x7FFFFFFF
        self._seed = (self._seed * 1103515245 + 12345) & 0x7FFFFFFF
        return self._seed

    def forward_tokens(self, feat_b, w_proj, pos, query_embed, B):
        """feat_b bf16 [B*M, F] (no grad), w_proj [C,F], pos [M,C], query_embed [Q,C] -> hs [L,B,Q,C]."""
        holder = _KVHolder()
        holder.seed = self.next_dropout_seed()
        holder.keep = self.debug_keep_kv
        self.last_holder = holder if self.debug_keep_kv else None
        layers = self.decoder.layers
        kv_free = self.kv_free and self.d_model == ops.MATTN_C and self.nhead == ops.MATTN_H and query_embed.shape[0] <= 16
        if kv_free:
            token = _MemorySideKVFree.apply(feat_b, w_proj, self.pre_norm.weight, self.pre_norm.bias, pos, holder, self.pre_norm.eps, len(layers))
        else:
            in_w = [l.multihead_attn.in_proj_weight for l in layers]
            in_b = [l.multihead_attn.in_proj_bias for l in layers]
            token = _MemorySide.apply(feat_b, w_proj, self.pre_norm.weight, self.pre_norm.bias, pos, holder,
                                      self.pre_norm.eps, *in_w, *in_b)
        # the 13-row query side of all six layers + decoder.norm: one autograd node on libhh kernels (model/qside.py)
        p = layers[0].p_attn if self.training else 0.0
        norm = self.decoder.norm
        hs, last = QueryStack.apply(query_embed, token, norm.weight, norm.bias, holder, B, self.nhead, norm.eps, p, *stack_params(layers))
        if not self.decoder.return_intermediate:
            hs = last[None]
        hs._hh_last = last                        # == hs[-1] as a tensor of its own (QueryStack's second output): see ObjDecoder.forward / step.py
        if not torch.is_grad_enabled():
            holder.kv = holder.mem = holder.mp = None
        return hs

    def forward(self, src, mask, query_embed, pos_embed):
        """Reference layout (tfm_decoder.py:76-93): src [B,C,T,n] (already projected), mask [B,T,n] (all False),
        query_embed [Q,C], pos_embed [1,C,T,n] -> (hs [L,B,Q,C], memory [B,C,T,n], [], []).  Differentiable in src, pos_embed,
        query_embed and every parameter; ObjDecoder.forward takes the batched forward_tokens route instead."""
        _require_gpu(src, "Cross_Attention")
        bs, c, h, w = src.shape
        src = src.flatten(2).permute(2, 0, 1)                                  # [M,B,C]
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1)
        query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        mask = mask.flatten(1)
        tgt = torch.zeros_like(query_embed)
        memory = _LayerNormFn.apply(src.reshape(-1, c), self.pre_norm.weight, self.pre_norm.bias, self.pre_norm.eps).view(h * w, bs, c)
        hs, attn_rollout, self_attn = self.decoder(tgt, memory, memory_key_padding_mask=mask, pos=pos_embed, query_pos=query_embed,
                                                   num_frames=h, seq_len=w)
        return hs.transpose(1, 2), memory.permute(1, 2, 0).reshape(bs, c, h, w), attn_rollout, self_attn


class ObjDecoder(nn.Module):
    """tfm_decoder.py:111-241.  forward(features [B,T,n,feature_dim]) -> (out dict, hs [L,B,Q,C], [], [])."""

    def __init__(self, transformer, num_classes, num_queries, feature_dim=768, aux_loss=False, pred_traj=True, num_frames=4,
                 patches_per_frame=256, backbone='LaviLa', self_attn=False):
        super().__init__()
        if self_attn or num_queries == 1:
            raise NotImplementedError("ObjDecoder: self_attn=True / num_queries==1 branches are not on the hot path")
        self.backbone = backbone
        self.init_proj_layers()
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.hidden_dim = hidden_dim
        self.class_embed = LinearX3(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.bbox_embed.single_use = True             # applied once per step (forward below): weight gradients go straight into the arena
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.pred_traj = pred_traj
        self.n_decode = 1
        if self.pred_traj:
            self.frame_index = nn.Embedding(num_frames, hidden_dim)
            self.frame_proj = LinearX3(hidden_dim * 2, hidden_dim)
        self.aux_loss = aux_loss
        self.pos_embed = nn.Parameter(torch.zeros(1, patches_per_frame + 1, hidden_dim))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, hidden_dim))
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.temporal_embed, std=.02)
        self.patches_per_frame = patches_per_frame
        self.proj = nn.Linear(feature_dim, hidden_dim, bias=False)
        self.num_frames = num_frames
        self.init_obj_model()
        # The reference materialises class logits for all 6 layers expanded over frames ([6, B*T, Q, 22048], a copy;
        # tfm_decoder.py:208,216) although only argmax of the last layer feeds a no-grad metric (box_utils.py:151).
        # materialize_logits=False returns 'pred_logits_argmax' [B*T,Q] instead and skips unused aux boxes.
        self.materialize_logits = True

    def construct_3d_pos_embed(self, T):
        tile_pos_embed = self.pos_embed[:, 1:, :].repeat(1, T, 1)
        tile_temporal_embed = self.temporal_embed[:, :T].repeat_interleave(self.patches_per_frame, 1)
        return (tile_pos_embed + tile_temporal_embed).view(1, T, self.patches_per_frame, self.pos_embed.shape[-1])

    def init_proj_layers(self):
        self.txt_proj = nn.Sequential(nn.ReLU(), LinearX3(768, 256))
        self.vid_proj = nn.Sequential(LinearX3(768, 256))

    def init_obj_model(self):
        self.obj_proj = nn.Sequential(LinearX3(self.hidden_dim, self.hidden_dim), nn.ReLU(), LinearX3(self.hidden_dim, 256))

    def forward(self, features, use_checkpoint=False):
        _require_gpu(features, "ObjDecoder")
        B, T, n, Fd = features.shape
        C = self.hidden_dim
        feat_b = ops.to_bf16(features.detach().reshape(B * T * n, Fd).contiguous())
        pos = _PosEmbed3D.apply(self.pos_embed, self.temporal_embed, T)          # == construct_3d_pos_embed(T).view(T * n, C)
        hs = self.transformer.forward_tokens(feat_b, self.proj.weight, pos, self.query_embed.weight, B)     # [L,B,Q,C]
        L, _, Q, _ = hs.shape
        full = self.materialize_logits
        last = getattr(hs, "_hh_last", None)                      # the last layer's output as QueryStack's own second output (== hs[-1])
        last1 = hs[-1:] if last is None else last[None]
        if self.pred_traj and T == self.num_frames:
            w_hs, w_fr = _SplitCols.apply(self.frame_proj.weight, C)
            base = linear_x3(hs if full else last1, w_hs)                                                    # [l,B,Q,C]
            fr = linear_x3(_FirstRows.apply(self.frame_index.weight, T), w_fr, self.frame_proj.bias)       # [T,C]
            cond = (base[:, :, None] + fr[None, None, :, None, :]).flatten(1, 2)                             # [l,B*T,Q,C]
        else:
            cond = hs if full else last1
            if self.pred_traj and torch.is_grad_enabled():
                # the trajectory branch is skipped for this clip length (on every rank alike: the shape decides): tell the gradient
                # buckets now, or frame_index / frame_proj would hold back every all-reduce until the end of backward (parallel.py)
                for prm, nm in ((self.frame_index.weight, "frame_index.weight"), (self.frame_proj.weight, "frame_proj.weight"),
                                (self.frame_proj.bias, "frame_proj.bias")):
                    sink = getattr(prm, "_hh_sink", None)
                    if sink is not None:
                        sink.arena.declare_unused([sink.name])
        outputs_coord = self.bbox_embed(cond).sigmoid()
        # (one layer only on the fast path: a view, not a select -- SelectBackward would zero-fill a [1, ...] temporary and copy into it)
        out = {'pred_boxes': outputs_coord[-1] if outputs_coord.shape[0] != 1 else outputs_coord.view(outputs_coord.shape[1:])}
        expand_t = self.pred_traj and T == self.num_frames
        if full:
            outputs_class = self.class_embed(hs)
            if expand_t:
                outputs_class = outputs_class[:, :, None].expand(-1, -1, T, -1, -1).flatten(1, 2)
            out['pred_logits'] = outputs_class[-1]
            if self.aux_loss:
                out['aux_outputs'] = self._set_aux_loss(outputs_class, outputs_coord)
        else:
            with torch.no_grad():
                am = self.class_embed(last1[0]).argmax(-1)                                                   # [B,Q]
                out['pred_logits_argmax'] = am[:, None].expand(-1, T, -1).flatten(0, 1) if expand_t else am
            out['pred_logits'] = None
            out['num_classes'] = self.class_embed.out_features
            if self.aux_loss:
                out['aux_outputs'] = []
        return out, hs, [], []

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]


def build_decoder(cfg, state_dict=None, device="cuda", feature_dim=None):
    """ObjDecoder as run/train.py:447-457 builds it (num_queries = nq + 1, aux_loss, pred_traj, feature_dim)."""
    tfm = Cross_Attention(d_model=cfg.dec_dim, nhead=cfg.dec_heads, num_decoder_layers=cfg.dec_layers,
                          dim_feedforward=cfg.dec_ffn, normalize_before=True, return_intermediate_dec=True)
    model = ObjDecoder(transformer=tfm, num_classes=cfg.num_classes, num_queries=cfg.dec_queries, aux_loss=True, pred_traj=True,
                       feature_dim=feature_dim or cfg.embed_dim, num_frames=cfg.num_frames,
                       patches_per_frame=cfg.patches_per_frame, self_attn=False)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    return model.to(device)
