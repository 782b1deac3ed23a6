"""Query side of the object-query decoder on libhh kernels (csrc/qside.hip): the 13 query rows per clip of
/root/reference/model/tfm_decoder.py:430-461 (TransformerDecoderLayer.forward_pre, sa_first) for all six layers, the per-layer
`decoder.norm` of TransformerDecoder.forward (:281-282), and `nn.Linear` replacements for the heads / projections
(:170-180,208-233; run/train.py:124-125,187-189).

  * every GEMM is hh_qgemm_f32x3: fp32 operands, bf16 matrix cores, three MFMAs per product (fp32-grade accuracy -- the 1e-3 loss
    bound leaves no room for bf16 weights / gradients here, see csrc/qside.hip);
  * `QueryStack` is ONE autograd node for the six layers with an explicit backward: 11 launches per layer forward, 20 backward,
    no autograd bookkeeping kernels -- bias / ReLU / dropout / residual / LayerNorm-residual ride in GEMM epilogues and prologues;
    the weight gradient's bias column sums are a by-product of the TN GEMM;
  * dropout (p = 0.1 on three residual branches, the FFN hidden layer and both attention maps, tfm_decoder.py:365-380) is a
    counter-based hash of (seed, element index), regenerated in the backward instead of stored;
  * the cross-attention core between the two halves of a layer runs in MEMORY SPACE (round 5, csrc/mattn.hip): the projected query of
    every head is mapped through that head's key rows of in_proj_weight (`ops.head_map_in`, one head-batched launch), attends over the
    un-projected memory rows (`ops.mattn_fwd`), and the value rows / bias are applied to the 13 pooled rows (`ops.head_map_out`) -- the
    memory side projects nothing.  The round-1..4 form (hh_xattn_fwd / hh_xattn_bwd on K/V column slices that `_MemorySide` projected
    for all layers at once) is kept behind `Cross_Attention.kv_free = False`.
"""
import torch
from torch import nn

from .. import ops

_SITE = {"sa": 1, "d1": 2, "x": 3, "d2": 4, "ff": 5, "d3": 6}


def _seed(base, layer, site):
    return (int(base) + 7919 * layer + 104729 * _SITE[site]) & 0xFFFFFFFF


class _LinearX3(torch.autograd.Function):
    """y = [relu](x.W^T + b) on hh_qgemm_f32x3 (NT); backward = NN input-gradient GEMM + TN weight-gradient GEMM whose by-product
    is the bias gradient.  x [..., K] fp32, w [N, K] (a row-strided view is fine), b [N] or None."""

    @staticmethod
    def forward(ctx, x, w, b, relu, sink_ok=False):
        K = x.shape[-1]
        x2 = x.detach().reshape(-1, K)
        if x2.dtype != torch.float32 or x2.stride(-1) != 1 or x2.stride(0) % 4:
            x2 = x2.float().contiguous()
        wd = w.detach()
        if wd.stride(-1) != 1 or wd.stride(0) % 4 or wd.data_ptr() % 16:
            wd = wd.contiguous()
        bd = None if b is None else b.detach().contiguous()
        # hh_qgemm_f32x3 wants N % 4 == 0 and K % 4 == 0 (16-byte rows); nn.Linear in the reference takes any size (e.g. a class head
        # with num_classes + 1 not a multiple of 4): zero-pad the operands and slice the results
        N = wd.shape[0]
        Kp, Np = (K + 3) // 4 * 4, (N + 3) // 4 * 4
        if Kp != K or Np != N:
            x2 = torch.nn.functional.pad(x2, (0, Kp - K)) if Kp != K else x2
            wd = torch.nn.functional.pad(wd, (0, Kp - K, 0, Np - N))
            bd = None if bd is None else torch.nn.functional.pad(bd, (0, Np - N))
        y = ops.qgemm(x2, wd, ops.NT, bias=bd, relu=relu)
        ctx.save_for_backward(x2, wd, y if relu else None)
        ctx.has_bias, ctx.relu, ctx.xshape, ctx.N, ctx.K = b is not None, relu, x.shape, N, K
        # the Parameters themselves: their gradient sinks (parallel._GradSink) -- only where the caller vouches that this node is the
        # parameters' ONLY consumer in a step (sink_ok: the _GradSink precondition; txt_proj, e.g., runs twice per step)
        ctx.param_objs = (w, b) if sink_ok else (None, None)
        return (y if Np == N else y[:, :N]).reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        N = w.shape[0]                                    # padded sizes from here on; results are sliced back at the end
        dz = dy.reshape(-1, ctx.N)
        if N != ctx.N:
            dz = torch.nn.functional.pad(dz, (0, N - ctx.N))
        if ctx.relu:
            dz = torch.ops.aten.threshold_backward(dz.float(), y, 0.0)       # dz * (y > 0) in one launch
        dz = dz.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.qgemm(dz, w, ops.NN)
            dx = (dx if dx.shape[1] == ctx.K else dx[:, :ctx.K]).reshape(ctx.xshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            rows, K = x2.shape
            tiles = ((N + 63) // 64) * ((K + 63) // 64)
            split = max(2, min(32, rows // 256, 512 // tiles)) if rows >= 2048 and tiles < 256 else 1    # long contraction, few output tiles (box heads: 6656 rows): split-K, atomic accumulate
            # Gradient sinks (parallel._GradSink): inside a TrainStep a weight / bias that is an nn.Parameter with exactly this one consumer
            # owns a zeroed slice of the flat gradient arena -- the weight-gradient GEMM (and its bias by-product) write there, no zero
            # fill, no AccumulateGrad add per parameter.  Views of a parameter (frame_proj's column halves) have no sink and go through autograd.
            sw, sb = (getattr(t, "_hh_sink", None) for t in ctx.param_objs)
            sunk = (sw is not None and sw.armed() and ctx.needs_input_grad[1] and (N, K) == (ctx.N, ctx.K) and sw.view.shape == (N, K)
                    and sw.view.data_ptr() % 16 == 0
                    and (not ctx.has_bias or (sb is not None and sb.armed() and ctx.needs_input_grad[2] and sb.view.data_ptr() % 16 == 0)))
            if sunk:
                sw.claim()
                if ctx.has_bias:
                    sb.claim()
                ops.qgemm(dz, x2, ops.TN, colsum=sb.view if ctx.has_bias else None, out=sw.view, splitk=split)
                sw.done()
                if ctx.has_bias:
                    sb.done()
                return dx, None, None, None, None
            if split > 1:
                buf = torch.zeros(N * K + N, dtype=torch.float32, device=dz.device)
                dw, db = buf[:N * K].view(N, K), (buf[N * K:] if ctx.has_bias else None)
                ops.qgemm(dz, x2, ops.TN, colsum=db, out=dw, splitk=split)
            else:
                db = torch.empty(N, dtype=torch.float32, device=dz.device) if ctx.has_bias else None
                dw = ops.qgemm(dz, x2, ops.TN, colsum=db)
            if dw.shape != (ctx.N, ctx.K):
                dw = dw[:ctx.N, :ctx.K]
                db = None if db is None else db[:ctx.N]
        return dx, dw, db, None, None


def linear_x3(x, w, b=None, relu=False, sink_ok=False):
    return _LinearX3.apply(x, w, b, relu, sink_ok)


class LinearX3(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose GPU forward and backward run on hh_qgemm_f32x3.
    `single_use = True` (set by the owner, e.g. ObjDecoder for its box / object heads): the module is applied ONCE per training step, so
    its weight gradients may be written straight into the flat gradient arena (parallel._GradSink)."""

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("LinearX3: the product path runs on libhh HIP kernels only (got a %s tensor)" % x.device)
        return _LinearX3.apply(x, self.weight, self.bias, False, getattr(self, "single_use", False))


# A/B switch (bench.py --no-qgroup): False = every product of QueryStack is its own launch, as in rounds 2-5
GROUP_LAUNCHES = True


# parameters of one TransformerDecoderLayer in the order QueryStack receives them
LAYER_PARAMS = ("norm1.weight", "norm1.bias", "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                "self_attn.out_proj.bias", "norm2.weight", "norm2.bias", "multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias",
                "multihead_attn.out_proj.weight", "multihead_attn.out_proj.bias", "norm3.weight", "norm3.bias", "linear1.weight",
                "linear1.bias", "linear2.weight", "linear2.bias")
NP = len(LAYER_PARAMS)


class QueryStack(torch.autograd.Function):
    """hs [L, B, Q, C] = decoder.norm(layer_l(...)) for l = 0..L-1, starting from tgt = 0 (tfm_decoder.py:85,255-295).

    Inputs: query_embed [Q, C] (query_pos, broadcast over clips), `token` (the 1-element output of _MemorySide: orders that node's
    backward after this one, which writes the K/V gradients into holder.dkv), decoder.norm weight / bias, then the LAYER_PARAMS
    tensors of every layer.  Non-tensor arguments: holder (batched K/V of all layers), B, heads, eps, p (dropout probability, 0 in
    eval mode)."""

    @staticmethod
    def forward(ctx, query_embed, token, dnw, dnb, holder, B, heads, eps, p, *params):
        L = len(params) // NP
        Q, C = query_embed.shape
        R = B * Q
        dev = query_embed.device
        f = lambda t: t.detach().contiguous()
        qpos = f(query_embed).float()
        P = [[f(t) for t in params[l * NP:(l + 1) * NP]] for l in range(L)]
        M, Lk = holder.M, holder.L
        kv_free = holder.mem is not None                 # memory-space cross-attention (csrc/mattn.hip): no K/V projection of the memory
        kvv = None if kv_free else holder.kv.view(B, M, 2 * Lk * C)
        tgt_all = torch.empty((L + 1, R, C), dtype=torch.float32, device=dev)
        tgt_all[0].zero_()
        saved = []
        for l in range(L):
            n1w, n1b, wi_s, bi_s, wo_s, bo_s, n2w, n2b, wi_c, bi_c, wo_c, bo_c, n3w, n3b, w1, b1, w2, b2 = P[l]
            sd = lambda site: _seed(holder.seed, l, site)
            x = tgt_all[l]
            a, aq, mean1, rstd1 = ops.layernorm_pos(x, n1w, n1b, eps, qpos, save_stats=True)
            qkv = torch.empty((R, 3 * C), dtype=torch.float32, device=dev)
            pair = ops.QGemmGroup() if GROUP_LAUNCHES else None                                         # the two halves of the in-projection: one launch
            ops.qgemm(aq, wi_s[:2 * C], ops.NT, bias=bi_s[:2 * C], out=qkv[:, :2 * C], defer=pair)      # q = k = norm1(x) + query_pos
            ops.qgemm(a, wi_s[2 * C:], ops.NT, bias=bi_s[2 * C:], out=qkv[:, 2 * C:], defer=pair)       # v = norm1(x)
            if pair is not None:
                pair.launch()
            o = ops.qself_attn_fwd(qkv, B, Q, heads, p, sd("sa"))
            tgt1 = ops.qgemm(o, wo_s, ops.NT, bias=bo_s, drop_p=p, drop_seed=sd("d1"), resid=x)
            _, cq, mean2, rstd2 = ops.layernorm_pos(tgt1, n2w, n2b, eps, qpos, save_stats=True)
            q = ops.qgemm(cq, wi_c[:C], ops.NT, bias=bi_c[:C], scale=0.125)                             # (norm2 + query_pos).Wq, scaled d^-1/2
            if kv_free:
                # scores q_h.(mp Wk_h^T + bk_h) = (q_h Wk_h).mp + const: map the 13 rows of every head into memory space, attend over the
                # un-projected rows, apply Wv_h / bv_h to the 13 pooled rows (tfm_decoder.py:438-441; csrc/mattn.hip)
                qt = ops.head_map_in(q, wi_c[C:2 * C])                                                      # [R, heads * C]
                pooled, lse, rsum = ops.mattn_fwd(qt, holder.mp, holder.mem, Q, p, sd("x"), keys_valid=holder.Mv)
                ca = ops.head_map_out(pooled, wi_c[2 * C:], bias=bi_c[2 * C:], rowscale=rsum if p > 0 else None)
                xs = (qt, pooled, rsum)
            else:
                k, v = kvv[:, :, l * C:(l + 1) * C], kvv[:, :, (Lk + l) * C:(Lk + l + 1) * C]
                ca, lse = ops.xattn_fwd(q.view(B, Q, C), k, v, heads, p, sd("x"))
                xs = None
            tgt2 = ops.qgemm(ca.view(R, C), wo_c, ops.NT, bias=bo_c, drop_p=p, drop_seed=sd("d2"), resid=tgt1)
            e, mean3, rstd3 = ops.layernorm(tgt2, n3w, n3b, eps, out_dtype=torch.float32, save_stats=True)
            hid = ops.qgemm(e, w1, ops.NT, bias=b1, relu=True, drop_p=p, drop_seed=sd("ff"))
            ops.qgemm(hid, w2, ops.NT, bias=b2, drop_p=p, drop_seed=sd("d3"), resid=tgt2, out=tgt_all[l + 1])
            if holder.keep:                              # test hook: which FFN units are active (the branch of the ReLU this step differentiates)
                holder.relu_masks.append(hid > 0)
            saved.append((a, aq, mean1, rstd1, qkv, o, tgt1, cq, mean2, rstd2, q, ca, lse, tgt2, e, mean3, rstd3, hid, xs))
        hs, dmean, drstd = ops.layernorm(tgt_all[1:].reshape(L * R, C), f(dnw), f(dnb), eps, out_dtype=torch.float32, save_stats=True)
        ctx.holder, ctx.dims, ctx.eps, ctx.p = holder, (L, B, Q, C, heads), eps, p
        ctx.P, ctx.saved, ctx.tgt_all, ctx.dec = P, saved, tgt_all, (f(dnw), dmean, drstd)
        ctx.param_objs = (dnw, dnb) + tuple(params)             # the Parameters themselves: their gradient sinks (parallel._GradSink), if any
        ctx.query_obj = query_embed
        # Two outputs: all layers' normalised outputs [L,B,Q,C] and -- a separate tensor -- the LAST layer's, which is all the fast path
        # consumes (heads, obj_proj): its consumers' gradients then arrive as one [B,Q,C] tensor instead of two zero-filled [L,B,Q,C]
        # SelectBackward / SliceBackward temporaries and their sum, and the decoder.norm backward of the layers nobody read is skipped
        ctx.set_materialize_grads(False)
        hs = hs.view(L, B, Q, C)
        return hs, hs[L - 1].clone()

    @staticmethod
    def backward(ctx, dhs, dlast):
        L, B, Q, C, heads = ctx.dims
        R, p, h = B * Q, ctx.p, ctx.holder
        dev = ctx.tgt_all.device
        dhs = None if dhs is None else dhs.contiguous().view(L, R, C)
        dlast = None if dlast is None else dlast.contiguous().view(R, C)
        if dhs is None and dlast is None:
            dlast = torch.zeros((R, C), dtype=torch.float32, device=dev)
        dnw, dmean, drstd = ctx.dec
        F_ = ctx.P[0][14].shape[0]
        # Gradient sinks: inside a TrainStep every parameter of this node owns a zeroed slice of the flat gradient arena
        # (parallel._GradSink).  When all of them are armed the weight-gradient GEMMs and the LayerNorm reductions write there
        # directly and this node returns None for them -- no AccumulateGrad `grad += new` kernel per parameter, no temporaries.
        sinks = [getattr(t, "_hh_sink", None) for t in ctx.param_objs]
        sunk = all(s is not None and s.armed() for s in sinks)
        sv = (lambda l, i: sinks[2 + l * NP + i].view) if sunk else None        # arena view of LAYER_PARAMS[i] of layer l
        if sunk:
            lnbuf = d_wi_c = d_bi_c = None
            for s_ in sinks:
                s_.claim()
        else:
            # LayerNorm weight / bias gradients are accumulated with atomics: one zeroed slab for all of them
            lnbuf = torch.zeros((L * 3 + 1, 2, C), dtype=torch.float32, device=dev)
            d_wi_c = torch.zeros((L, 3 * C, C), dtype=torch.float32, device=dev)         # key / value rows belong to _MemorySide
            d_bi_c = torch.zeros((L, 3 * C), dtype=torch.float32, device=dev)
        # accumulation targets of the LayerNorm weight / bias gradients (zero on entry in both cases)
        ln_w = lambda l, k: sv(l, (0, 6, 12)[k]) if sunk else lnbuf[l * 3 + k, 0]
        ln_b = lambda l, k: sv(l, (1, 7, 13)[k]) if sunk else lnbuf[l * 3 + k, 1]
        dn_w, dn_b = (sinks[0].view, sinks[1].view) if sunk else (lnbuf[L * 3, 0], lnbuf[L * 3, 1])
        new = lambda l, i, *shape: sv(l, i) if sunk else torch.empty(shape, dtype=torch.float32, device=dev)
        kv_free = h.mem is not None
        M, Lk = h.M, h.L
        if kv_free:
            pdT, dsT, qt16, dp16 = h.bwd_buffers()
        else:
            if h.dkv is None:
                h.dkv = torch.empty_like(h.kv)
            kvv, dkvv = h.kv.view(B, M, 2 * Lk * C), h.dkv.view(B, M, 2 * Lk * C)
        keep = 1.0 / (1.0 - p) if p > 0 else 1.0
        grads = [None] * (L * NP)
        dqpos_parts = []
        g = None
        for l in reversed(range(L)):
            n1w, n1b, wi_s, bi_s, wo_s, bo_s, n2w, n2b, wi_c, bi_c, wo_c, bo_c, n3w, n3b, w1, b1, w2, b2 = ctx.P[l]
            a, aq, mean1, rstd1, qkv, o, tgt1, cq, mean2, rstd2, q, ca, lse, tgt2, e, mean3, rstd3, hid, xs = ctx.saved[l]
            sd = lambda site: _seed(h.seed, l, site)
            x, tgt3 = ctx.tgt_all[l], ctx.tgt_all[l + 1]
            # decoder.norm of this layer's output (+ the gradient arriving from layer l + 1)
            sl = slice(l * R, (l + 1) * R)
            dy = None if dhs is None else dhs[l]
            if l == L - 1 and dlast is not None:
                dy = dlast if dy is None else dy + dlast
            if dy is not None:
                g = ops.layernorm_bwd_add(tgt3, dnw, dmean[sl], drstd[sl], dy, g, dn_w, dn_b, out=g)
            elif g is None:                              # (nobody read the last layer: nothing flows into it)
                g = torch.zeros((R, C), dtype=torch.float32, device=dev)
            drop = lambda site: dict(a_drop_p=p, a_drop_seed=sd(site), a_drop_ld=C) if p > 0 else {}
            # the layer's nine weight gradients (TN products: they feed nothing inside this backward) are collected and go out as ONE grouped
            # launch at the end of the layer (hh_qgemm_f32x3_group; the reference's autograd: one cuBLAS call each)
            tn = ops.QGemmGroup() if GROUP_LAUNCHES else None
            # FFN:  tgt3 = tgt2 + drop3(hid.W2^T + b2),  hid = drop(relu(e.W1^T + b1)),  e = norm3(tgt2)
            db2 = new(l, 17, C)
            dw2 = ops.qgemm(g, hid, ops.TN, colsum=db2, out=sv(l, 16) if sunk else None, defer=tn, **drop("d3"))
            dz = ops.qgemm(g, w2, ops.NN, relu_mask=hid, mask_scale=keep, **drop("d3"))
            db1 = new(l, 15, F_)
            dw1 = ops.qgemm(dz, e, ops.TN, colsum=db1, out=sv(l, 14) if sunk else None, defer=tn)
            de = ops.qgemm(dz, w1, ops.NN)
            g2 = ops.layernorm_bwd_add(tgt2, n3w, mean3, rstd3, de, g, ln_w(l, 2), ln_b(l, 2))
            # cross-attention:  tgt2 = tgt1 + drop2(ca.Wo^T + bo),  ca = xattn(q, K_l, V_l),  q = (cq.Wq^T + bq) / 8,  cq = norm2(tgt1) + qpos
            dbo_c = new(l, 11, C)
            dwo_c = ops.qgemm(g2, ca.view(R, C), ops.TN, colsum=dbo_c, out=sv(l, 10) if sunk else None, defer=tn, **drop("d2"))
            dca = ops.qgemm(g2, wo_c, ops.NN, **drop("d2"))
            gw_c, gb_c = (sv(l, 8), sv(l, 9)) if sunk else (d_wi_c[l], d_bi_c[l])       # gradient of this layer's packed [q; k; v] in-projection
            if kv_free:
                qt, pooled, rsum = xs
                rs = rsum if p > 0 else None
                # head output O_h = pooled_h Wv_h^T + rsum_h bv_h:  d pooled, d Wv (value rows), d bv as the TN product's column sums
                dpooled = ops.head_map_in(dca, wi_c[2 * C:])
                ops.head_map_wgrad(dca, pooled, gw_c[2 * C:], colsum=gb_c[2 * C:], rowscale=rs, defer=tn)
                # attention core in memory space: d qt; Pd^T / dS^T of this layer for _MemorySideKVFree's batched d-memory GEMM
                dqt = ops.mattn_bwd(qt, dpooled, lse, dca, ca, bi_c[2 * C:], h.mp, h.mem, Q, pdT, dsT, qt16, dp16, l * 128, p, sd("x"), keys_valid=h.Mv)
                # qt_h = q_h Wk_h:  d q, d Wk (key rows); the key bias drops out of the softmax -- its gradient is exactly 0 (the rows stay zero)
                dq = ops.head_map_out(dqt, wi_c[C:2 * C])
                ops.head_map_wgrad(q, dqt, gw_c[C:2 * C], defer=tn)
            else:
                k, v = kvv[:, :, l * C:(l + 1) * C], kvv[:, :, (Lk + l) * C:(Lk + l + 1) * C]
                dk, dv = dkvv[:, :, l * C:(l + 1) * C], dkvv[:, :, (Lk + l) * C:(Lk + l + 1) * C]
                dq = ops.xattn_bwd(q.view(B, Q, C), k, v, ca, lse, dca.view(B, Q, C), dk, dv, heads, p, sd("x")).view(R, C)
            # query rows of the cross-attention in-projection (K/V path: its key / value rows are written by _MemorySide, which also
            # reports the parameter as final when both halves went straight into the arena)
            ops.qgemm(dq, cq, ops.TN, a_scale=0.125, colsum=gb_c[:C], out=gw_c[:C], defer=tn)
            dcq = ops.qgemm(dq, wi_c[:C], ops.NN, a_scale=0.125)
            g1 = ops.layernorm_bwd_add(tgt1, n2w, mean2, rstd2, dcq, g2, ln_w(l, 1), ln_b(l, 1))
            # self-attention:  tgt1 = x + drop1(o.Wo^T + bo),  o = attn(q = k = aq.W[:2C], v = a.W[2C:]),  a = norm1(x), aq = a + qpos
            dbo_s = new(l, 5, C)
            dwo_s = ops.qgemm(g1, o, ops.TN, colsum=dbo_s, out=sv(l, 4) if sunk else None, defer=tn, **drop("d1"))
            do = ops.qgemm(g1, wo_s, ops.NN, **drop("d1"))
            dqkv = ops.qself_attn_bwd(qkv, do, B, Q, heads, p, sd("sa"))
            dwi_s = new(l, 2, 3 * C, C)
            dbi_s = new(l, 3, 3 * C)
            ops.qgemm(dqkv[:, :2 * C], aq, ops.TN, colsum=dbi_s[:2 * C], out=dwi_s[:2 * C], defer=tn)
            ops.qgemm(dqkv[:, 2 * C:], a, ops.TN, colsum=dbi_s[2 * C:], out=dwi_s[2 * C:], defer=tn)
            if tn is not None:
                tn.launch()
            daq = ops.qgemm(dqkv[:, :2 * C], wi_s[:2 * C], ops.NN)
            da = ops.qgemm(dqkv[:, 2 * C:], wi_s[2 * C:], ops.NN, resid=daq)
            g = ops.layernorm_bwd_add(x, n1w, mean1, rstd1, da, g1, ln_w(l, 0), ln_b(l, 0))
            dqpos_parts += [dcq, daq]
            if not sunk:
                grads[l * NP:(l + 1) * NP] = [lnbuf[l * 3, 0], lnbuf[l * 3, 1], dwi_s, dbi_s, dwo_s, dbo_s, lnbuf[l * 3 + 1, 0], lnbuf[l * 3 + 1, 1],
                                              d_wi_c[l], d_bi_c[l], dwo_c, dbo_c, lnbuf[l * 3 + 2, 0], lnbuf[l * 3 + 2, 1], dw1, db1, dw2, db2]
        # d query_embed = sum over layers (two uses each) and clips: ONE reduction, written into the parameter's arena slice when it has an
        # armed sink (query_embed.weight enters only this node)
        qsink = getattr(ctx.query_obj, "_hh_sink", None)
        parts = torch.stack(dqpos_parts).view(-1, Q, C)
        if sunk and qsink is not None and qsink.armed() and qsink.view.shape == (Q, C):
            qsink.claim()
            torch.sum(parts, dim=0, out=qsink.view)
            qsink.done()
            dquery = None
        else:
            dquery = parts.sum(0)
        ctx.saved = ctx.tgt_all = None
        if sunk:
            # everything is in the arena: report the parameters final (bucket all-reduces may start), except the cross-attention
            # in-projections, whose key / value rows _MemorySide.backward still has to write
            for i, s_ in enumerate(sinks):
                if kv_free or i < 2 or (i - 2) % NP not in (8, 9):
                    s_.done()
            h.q_rows_sunk = not kv_free
            return (dquery, torch.zeros(1, dtype=torch.float32, device=dev), None, None, None, None, None, None, None, *grads)
        return (dquery, torch.zeros(1, dtype=torch.float32, device=dev), lnbuf[L * 3, 0], lnbuf[L * 3, 1], None, None, None, None, None, *grads)


def stack_params(layers):
    out = []
    for layer in layers:
        mods = dict(layer.named_parameters())
        out += [mods[n] for n in LAYER_PARAMS]
    return out
