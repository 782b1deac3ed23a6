"""Oracle: object-query decoder (fp32, CPU, autograd-capable).  Test infrastructure.

Restates /root/reference/model/tfm_decoder.py (ObjDecoder / Cross_Attention / TransformerDecoder /
TransformerDecoderLayer.forward_pre with sa_first=True) batch-first.  Dropout is p=0 here: parity is
defined in eval mode.  The literal `4` of tfm_decoder.py:216 is `T` here
(identical at T=4).
"""
import torch
import torch.nn.functional as F


def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def mha(q_in, k_in, v_in, sd, prefix, heads):
    """nn.MultiheadAttention forward (packed in_proj [q;k;v], scale d^-0.5 on q) -- the op behind
    tfm_decoder.py:433-441.  Inputs batch-first [B,L,C]; returns [B,Lq,C].  Head-averaged weights
    (need_weights=True default) are discarded by the caller and not computed."""
    C = q_in.shape[-1]
    d = C // heads
    Wq, Wk, Wv = sd[prefix + ".in_proj_weight"].chunk(3, dim=0)
    bq, bk, bv = sd[prefix + ".in_proj_bias"].chunk(3, dim=0)
    B, Lq, Lk = q_in.shape[0], q_in.shape[1], k_in.shape[1]
    q = F.linear(q_in, Wq, bq).view(B, Lq, heads, d).transpose(1, 2) * (d ** -0.5)
    k = F.linear(k_in, Wk, bk).view(B, Lk, heads, d).transpose(1, 2)
    v = F.linear(v_in, Wv, bv).view(B, Lk, heads, d).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, C)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def mha_given_kv(q_in, K, V, sd, prefix, heads):
    """Cross-attention of `mha` with the key/value projections ALREADY applied: K, V [B,M,C] = k_in.Wk^T+bk, v_in.Wv^T+bv.
    Lets a test feed the oracle's query side the very K/V tensors another implementation produced."""
    C = q_in.shape[-1]
    d = C // heads
    Wq, bq = sd[prefix + ".in_proj_weight"][:C], sd[prefix + ".in_proj_bias"][:C]
    B, Lq, Lk = q_in.shape[0], q_in.shape[1], K.shape[1]
    q = F.linear(q_in, Wq, bq).view(B, Lq, heads, d).transpose(1, 2) * (d ** -0.5)
    k = K.view(B, Lk, heads, d).transpose(1, 2)
    v = V.view(B, Lk, heads, d).transpose(1, 2)
    o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B, Lq, C)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def memory_rows(features, sd, cfg):
    """(memory, pos) of ObjDecoder / Cross_Attention (tfm_decoder.py:200-205,86-88): memory = pre_norm(proj(x)) [B,M,C], pos [M,C]."""
    B, T, n, _ = features.shape
    memory = _ln(F.linear(features, sd["proj.weight"]).reshape(B, T * n, cfg.dec_dim), sd, "transformer.pre_norm")
    return memory, pos_embed_3d(sd, T, n)


def memory_kv(features, sd, cfg):
    """Memory side of ObjDecoder/Cross_Attention (tfm_decoder.py:200-205,86-88,438-441): features [B,T,n,F] ->
    (K, V) [L,B,M,C] with K_l = (pre_norm(proj(x)) + pos).Wk_l^T + bk_l and V_l = pre_norm(proj(x)).Wv_l^T + bv_l."""
    B, T, n, _ = features.shape
    C, L = cfg.dec_dim, cfg.dec_layers
    memory = _ln(F.linear(features, sd["proj.weight"]).reshape(B, T * n, C), sd, "transformer.pre_norm")
    pos = pos_embed_3d(sd, T, n)[None]
    Ks, Vs = [], []
    for l in range(L):
        w, b = sd[f"transformer.decoder.layers.{l}.multihead_attn.in_proj_weight"], sd[f"transformer.decoder.layers.{l}.multihead_attn.in_proj_bias"]
        Ks.append(F.linear(memory + pos, w[C:2 * C], b[C:2 * C]))
        Vs.append(F.linear(memory, w[2 * C:], b[2 * C:]))
    return torch.stack(Ks), torch.stack(Vs)


def decoder_layer(tgt, memory, pos, qpos, sd, b, heads, kv=None, rows=None, relu_mask=None, relu_trace=None):
    """TransformerDecoderLayer.forward_pre, sa_first -- tfm_decoder.py:430-461 (all LN eps 1e-5).  kv = (K, V) of this layer
    replaces the in-layer key/value projection of (memory + pos, memory); rows = (memory, memory_plus_pos) replaces the two operands
    of that projection (lets a test feed the oracle the very memory rows another implementation attends over); relu_mask [B,Q,ffn]
    (bool) replaces the ReLU of the FFN by that fixed mask -- the branch of the piecewise-linear function another implementation took,
    so that units whose pre-activation sits within rounding distance of the kink do not make two fp32-grade gradients differ."""
    a = _ln(tgt, sd, b + "norm1")
    tgt = tgt + mha(a + qpos, a + qpos, a, sd, b + "self_attn", heads)
    c = _ln(tgt, sd, b + "norm2")
    if kv is not None:
        tgt = tgt + mha_given_kv(c + qpos, kv[0], kv[1], sd, b + "multihead_attn", heads)
    elif rows is not None:
        tgt = tgt + mha(c + qpos, rows[1], rows[0], sd, b + "multihead_attn", heads)
    else:
        tgt = tgt + mha(c + qpos, memory + pos, memory, sd, b + "multihead_attn", heads)
    e = _ln(tgt, sd, b + "norm3")
    h1 = F.linear(e, sd[b + "linear1.weight"], sd[b + "linear1.bias"])
    if relu_trace is not None:
        relu_trace.append((h1 > 0).detach())         # the branch THIS evaluation takes (tests count how many units another implementation flipped)
    ff = F.linear(F.relu(h1) if relu_mask is None else h1 * relu_mask.to(h1.dtype), sd[b + "linear2.weight"], sd[b + "linear2.bias"])
    return tgt + ff


def pos_embed_3d(sd, T, n):
    """ObjDecoder.construct_3d_pos_embed -- tfm_decoder.py:161-166 -> [T*n, C] (pos_embed[0] unused)."""
    return (sd["pos_embed"][0, 1:].repeat(T, 1) + sd["temporal_embed"][0, :T].repeat_interleave(n, dim=0))


def cross_attention_forward(src, mask, query_embed, pos_embed, sd, cfg, prefix="transformer."):
    """Cross_Attention.forward in the reference's own layout -- tfm_decoder.py:76-93 (+ TransformerDecoder.forward
    :255-295).  src [B,C,T,n] (already projected), mask [B,T,n] bool (all False on the hot path: a True entry would
    exclude that key), query_embed [Q,C], pos_embed [1,C,T,n] -> (hs [L,B,Q,C], memory [B,C,T,n])."""
    B, C, T, n = src.shape
    assert not bool(mask.any()), "oracle: key padding is all-False on the hot path (tfm_decoder.py:203)"
    mem = src.flatten(2).transpose(1, 2)                                       # [B,M,C] (batch-first view of [M,B,C])
    pos = pos_embed.flatten(2).transpose(1, 2)                                 # [1,M,C]
    memory = _ln(mem, sd, prefix + "pre_norm")
    qpos = query_embed[None].expand(B, -1, -1)
    tgt = torch.zeros_like(qpos)
    inter = []
    for l in range(cfg.dec_layers):
        tgt = decoder_layer(tgt, memory, pos, qpos, sd, f"{prefix}decoder.layers.{l}.", cfg.dec_heads)
        inter.append(_ln(tgt, sd, prefix + "decoder.norm"))
    return torch.stack(inter), memory.transpose(1, 2).reshape(B, C, T, n)


def objdecoder_forward(features, sd, cfg, compute_logits=True, kv=None, rows=None, relu_masks=None, relu_trace=None):
    """ObjDecoder.forward -- tfm_decoder.py:183-233 (+ Cross_Attention.forward :76-93,
    TransformerDecoder.forward :255-295).

    features [B,T,n,F] -> (out dict, hs [L,B,Q,C]).  out['pred_boxes'] [B*T,Q,4] (cx,cy,w,h),
    out['pred_logits'] [B*T,Q,classes+1], out['aux_outputs'] for layers 0..L-2.
    kv = (K, V) [L,B,M,C]: run the query side on these key/value projections (see mha_given_kv) instead of the memory side's.
    rows = (memory, memory + pos) [B,M,C] each: run every layer's key / value projection on these rows instead of the memory side's.
    relu_masks: per layer [B,Q,ffn] bool, see decoder_layer.  relu_trace: a list that receives every layer's own (pre-activation > 0) mask.
    """
    B, T, n, _ = features.shape
    C, heads, L = cfg.dec_dim, cfg.dec_heads, cfg.dec_layers
    mem = F.linear(features, sd["proj.weight"]).reshape(B, T * n, C)
    memory = _ln(mem, sd, "transformer.pre_norm")
    pos = pos_embed_3d(sd, T, n)[None]
    qpos = sd["query_embed.weight"][None].expand(B, -1, -1)
    tgt = torch.zeros_like(qpos)
    inter = []
    for l in range(L):
        tgt = decoder_layer(tgt, memory, pos, qpos, sd, f"transformer.decoder.layers.{l}.", heads,
                            None if kv is None else (kv[0][l], kv[1][l]), rows, None if relu_masks is None else relu_masks[l], relu_trace)
        inter.append(_ln(tgt, sd, "transformer.decoder.norm"))
    hs = torch.stack(inter)                                                   # [L,B,Q,C]
    Q = hs.shape[2]
    boxes = box_head(hs, sd, cfg, T)
    out = {"pred_boxes": boxes[-1]}
    if compute_logits:
        logits = F.linear(hs, sd["class_embed.weight"], sd["class_embed.bias"])            # [L,B,Q,K]
        logits = logits[:, :, None].expand(L, B, T, Q, logits.shape[-1]).flatten(1, 2)
        out["pred_logits"] = logits[-1]
        out["aux_outputs"] = [{"pred_logits": logits[l], "pred_boxes": boxes[l]} for l in range(L - 1)]
    else:
        out["aux_outputs"] = [{"pred_boxes": boxes[l]} for l in range(L - 1)]
    return out, hs


def box_head(hs, sd, cfg, T):
    """Frame-conditioned trajectory head of ObjDecoder.forward -- tfm_decoder.py:210-216,228: cat[hs, frame_index[t]] ->
    frame_proj -> bbox_embed (3-layer MLP) -> sigmoid.  hs [L,B,Q,C] -> boxes [L,B*T,Q,4] (cx,cy,w,h)."""
    L, B, Q, C = hs.shape
    fe = sd["frame_index.weight"][:T]
    cond = torch.cat([hs[:, :, None].expand(L, B, T, Q, C), fe[None, None, :, None, :].expand(L, B, T, Q, C)], -1)
    x = F.linear(cond, sd["frame_proj.weight"], sd["frame_proj.bias"]).flatten(1, 2)      # [L,B*T,Q,C]
    for i in range(3):
        x = F.linear(x, sd[f"bbox_embed.layers.{i}.weight"], sd[f"bbox_embed.layers.{i}.bias"])
        if i < 2:
            x = F.relu(x)
    return x.sigmoid()


def txt_proj(x, sd):
    """ObjDecoder.txt_proj = ReLU -> Linear(768,256) -- tfm_decoder.py:170-171 (ReLU first)."""
    return F.linear(F.relu(x), sd["txt_proj.1.weight"], sd["txt_proj.1.bias"])


def obj_proj(x, sd):
    """ObjDecoder.obj_proj = Linear -> ReLU -> Linear(256) -- tfm_decoder.py:175-180."""
    return F.linear(F.relu(F.linear(x, sd["obj_proj.0.weight"], sd["obj_proj.0.bias"])),
                    sd["obj_proj.2.weight"], sd["obj_proj.2.bias"])
