"""Oracle: frozen LaViLa TimeSformer vision tower + CLIP text tower (fp32, CPU).  Test infrastructure.

Functional restatement over a state dict `sd` with the reference's key names.
"""
import torch
import torch.nn.functional as F


def layer_norm(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def quick_gelu(x):
    # /root/reference/model/openai_model.py:177-179
    return x * torch.sigmoid(1.702 * x)


def divided_attention(x, sd, prefix, heads, T, n, mode):
    """VarAttention.forward -- /root/reference/model/LaviLa.py:246-283 (+ attn() :194-198).

    x [B, 1+T*n, D]; token order frame-major / patch-minor.  mode 'space': one problem per frame
    (n queries, CLS + n keys); mode 'time': one problem per patch location (T queries, CLS + T keys).
    The CLS query attends every key of the clip (:258).  q is scaled by d^-0.5 before the split (:252).
    """
    B, N, D = x.shape
    d = D // heads
    qkv = F.linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"])
    q, k, v = qkv.view(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)          # each [B,h,N,d]
    q = q * (d ** -0.5)
    # CLS query over all tokens
    cls_p = torch.softmax(q[:, :, :1] @ k.transpose(-1, -2), dim=-1)      # [B,h,1,N]
    cls_out = cls_p @ v                                                    # [B,h,1,d]
    ql, kl, vl = (t[:, :, 1:].reshape(B, heads, T, n, d) for t in (q, k, v))
    if mode == "time":
        ql, kl, vl = (t.transpose(2, 3) for t in (ql, kl, vl))             # [B,h,n,T,d]
    G = ql.shape[2]
    kc = k[:, :, None, :1].expand(B, heads, G, 1, d)
    vc = v[:, :, None, :1].expand(B, heads, G, 1, d)
    kl = torch.cat([kc, kl], dim=3)
    vl = torch.cat([vc, vl], dim=3)
    p = torch.softmax(ql @ kl.transpose(-1, -2), dim=-1)
    o = p @ vl                                                             # [B,h,G,L,d]
    if mode == "time":
        o = o.transpose(2, 3)
    o = o.reshape(B, heads, T * n, d)
    o = torch.cat([cls_out, o], dim=2)                                     # [B,h,N,d]
    o = o.permute(0, 2, 1, 3).reshape(B, N, D)
    return F.linear(o, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])


def block(x, sd, b, heads, T, n):
    """SpaceTimeBlock.forward -- /root/reference/model/LaviLa.py:345-390.  Block LNs use eps 1e-6
    (:439); the space residual is added to the block INPUT x, not to x+time (:384)."""
    t = divided_attention(layer_norm(x, sd, b + "norm3", 1e-6), sd, b + "timeattn", heads, T, n, "time")
    s = divided_attention(layer_norm(x + t, sd, b + "norm1", 1e-6), sd, b + "attn", heads, T, n, "space")
    y = x + s
    h = F.linear(layer_norm(y, sd, b + "norm2", 1e-6), sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"])
    return y + F.linear(quick_gelu(h), sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])


def embed_tokens(video, sd, cfg):
    """Patch embed + CLS + pos/temporal embed + ln_pre -- LaviLa.py:218-223,537-559 (ln_pre eps 1e-5)."""
    B, T, C, H, W = video.shape
    n = cfg.patches_per_frame
    x = F.conv2d(video.reshape(B * T, C, H, W), sd["visual.patch_embed.proj.weight"], None,
                 stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2).reshape(B, T * n, -1)
    x = torch.cat([sd["visual.cls_token"].expand(B, -1, -1), x], dim=1)
    pos = sd["visual.pos_embed"]
    tile_pos = pos[:, 1:].repeat(1, T, 1)
    tile_tmp = sd["visual.temporal_embed"][:, :T].repeat_interleave(n, dim=1)
    total = torch.cat([pos[:, :1], tile_pos + tile_tmp], dim=1)
    return layer_norm(x + total, sd, "visual.ln_pre", 1e-5)


def vision_forward(video, sd, cfg, return_blocks=False):
    """SpaceTimeTransformer.forward -- LaviLa.py:537-581 -> (x_cls [B,D], x [B,N,D]) after final norm."""
    T, n = video.shape[1], cfg.patches_per_frame
    x = embed_tokens(video, sd, cfg)
    inter = [x]
    for i in range(cfg.depth):
        x = block(x, sd, f"visual.blocks.{i}.", cfg.num_heads, T, n)
        if return_blocks:
            inter.append(x)
    x = layer_norm(x, sd, "visual.norm", 1e-6)
    if return_blocks:
        return x[:, 0], x, inter
    return x[:, 0], x


def encode_image(video, sd, cfg):
    """CLIP.encode_image -- LaviLa.py:650-658."""
    x_cls, x = vision_forward(video, sd, cfg)
    return x_cls @ sd["image_projection"], x


def encode_text(text, sd, cfg):
    """CLIP.encode_text -- LaviLa.py:660-670; blocks = openai_model.py:182-216 (causal MHA, QuickGELU)."""
    W, h = cfg.text_width, cfg.text_heads
    L = text.shape[1]
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"][:L]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    Bt = x.shape[0]
    for i in range(cfg.text_layers):
        b = f"transformer.resblocks.{i}."
        a = layer_norm(x, sd, b + "ln_1", 1e-5)
        qkv = F.linear(a, sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"])
        q, k, v = qkv.view(Bt, L, 3, h, W // h).permute(2, 0, 3, 1, 4)
        p = torch.softmax((q * (W // h) ** -0.5) @ k.transpose(-1, -2) + mask, dim=-1)
        o = (p @ v).permute(0, 2, 1, 3).reshape(Bt, L, W)
        x = x + F.linear(o, sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"])
        m = F.linear(layer_norm(x, sd, b + "ln_2", 1e-5), sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"])
        x = x + F.linear(quick_gelu(m), sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"])
    x = layer_norm(x, sd, "ln_final", 1e-5)
    x_cls = x[torch.arange(Bt), text.argmax(dim=-1)] @ sd["text_projection"]
    return x_cls, x


def clip_forward(video, text, sd, cfg):
    """CLIP.forward(return_feature_map=True, norm_embed=True) -- LaviLa.py:672-687."""
    ie, ifm = encode_image(video, sd, cfg)
    te, tfm = encode_text(text, sd, cfg)
    return {"image_embed": F.normalize(ie, dim=-1), "text_embed": F.normalize(te, dim=-1),
            "image_feature_map": ifm, "text_feature_map": tfm, "logit_scale": sd["logit_scale"].exp()}
