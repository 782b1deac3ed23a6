"""Deterministic synthetic weights and batches (no checkpoints / datasets are reachable offline).

Every tensor is drawn from its own torch CPU generator seeded by (seed, crc32(name)), so the GPU
box regenerates bit-identical tensors without the reference tree.  State-dict key names and shapes
follow the reference modules:
  backbone keys  /root/reference/model/LaviLa.py:449-469,234-235,180-182,615-620
  decoder keys   /root/reference/model/tfm_decoder.py:131-158,170-180,365-377
The batch contract follows run/train.py:50-76 and data_loader/EgoClip_EgoMCQ_dataset.py:283-293.
`time_init='zeros'` would make temporal attention exactly 0 (LaviLa.py:236-242); the synthetic
weights therefore randomise timeattn.{qkv,proj} like every other matrix.
"""
import zlib

import torch

from .config import HHConfig

SOT, EOT = 49406, 49407


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFF)
    return g


def _normal(seed, name, shape, std=0.02, mean=0.0, clip=2.0):
    x = torch.randn(shape, generator=_gen(seed, name), dtype=torch.float32)
    if clip is not None:
        x = x.clamp_(-clip, clip)
    return x * std + mean


def encoder_state(cfg: HHConfig, seed: int = 0, with_text: bool = True) -> dict:
    """fp32 state dict of the CLIP wrapper (keys as reference `CLIP.state_dict()`)."""
    D, n, T = cfg.embed_dim, cfg.patches_per_frame, cfg.num_frames
    H = D * cfg.mlp_ratio
    sd = {}

    def mat(name, shape, std=0.02):
        sd[name] = _normal(seed, name, shape, std)

    def ln(name, dim):
        sd[name + ".weight"] = _normal(seed, name + ".weight", (dim,), 0.1, 1.0)
        sd[name + ".bias"] = _normal(seed, name + ".bias", (dim,), 0.05)

    mat("visual.cls_token", (1, 1, D))
    mat("visual.pos_embed", (1, n + 1, D))
    mat("visual.temporal_embed", (1, T, D))
    mat("visual.patch_embed.proj.weight", (D, 3, cfg.patch_size, cfg.patch_size))
    ln("visual.ln_pre", D)
    ln("visual.norm", D)
    for i in range(cfg.depth):
        b = f"visual.blocks.{i}."
        for nm in ("norm1", "norm2", "norm3"):
            ln(b + nm, D)
        for at in ("attn", "timeattn"):
            mat(b + at + ".qkv.weight", (3 * D, D))
            mat(b + at + ".qkv.bias", (3 * D,))
            mat(b + at + ".proj.weight", (D, D))
            mat(b + at + ".proj.bias", (D,))
        mat(b + "mlp.fc1.weight", (H, D))
        mat(b + "mlp.fc1.bias", (H,))
        mat(b + "mlp.fc2.weight", (D, H))
        mat(b + "mlp.fc2.bias", (D,))
    mat("image_projection", (D, cfg.project_embed_dim), D ** -0.5)
    if with_text:
        W = cfg.text_width
        mat("token_embedding.weight", (cfg.vocab_size, W))
        mat("positional_embedding", (cfg.context_length, W), 0.01)
        for i in range(cfg.text_layers):
            b = f"transformer.resblocks.{i}."
            mat(b + "attn.in_proj_weight", (3 * W, W), W ** -0.5)
            mat(b + "attn.in_proj_bias", (3 * W,))
            mat(b + "attn.out_proj.weight", (W, W), (W ** -0.5) * ((2 * cfg.text_layers) ** -0.5))
            mat(b + "attn.out_proj.bias", (W,))
            ln(b + "ln_1", W)
            ln(b + "ln_2", W)
            mat(b + "mlp.c_fc.weight", (4 * W, W), (2 * W) ** -0.5)
            mat(b + "mlp.c_fc.bias", (4 * W,))
            mat(b + "mlp.c_proj.weight", (W, 4 * W), (W ** -0.5) * ((2 * cfg.text_layers) ** -0.5))
            mat(b + "mlp.c_proj.bias", (W,))
        ln("ln_final", W)
        mat("text_projection", (W, cfg.project_embed_dim), W ** -0.5)
        sd["logit_scale"] = torch.tensor(2.659260036932778)  # log(1/0.07), LaviLa.py:622
    return sd


def decoder_state(cfg: HHConfig, seed: int = 0, feature_dim: int = None, query_std: float = 1.0) -> dict:
    """fp32 state dict of ObjDecoder (keys as reference `ObjDecoder.state_dict()`).

    query_std=1.0 mirrors nn.Embedding's N(0,1) init so that queries do not collapse onto one box
    (keeps the Hungarian problems well conditioned).
    """
    C, Q, T, n = cfg.dec_dim, cfg.dec_queries, cfg.num_frames, cfg.patches_per_frame
    F = feature_dim or cfg.embed_dim
    s = seed + 7919
    sd = {}

    def mat(name, shape, std=None):
        if std is None:  # xavier-like scale
            fan = shape[-1] if len(shape) > 1 else shape[0]
            std = (1.0 / fan) ** 0.5
        sd[name] = _normal(s, name, shape, std)

    def ln(name, dim):
        sd[name + ".weight"] = _normal(s, name + ".weight", (dim,), 0.1, 1.0)
        sd[name + ".bias"] = _normal(s, name + ".bias", (dim,), 0.05)

    mat("pos_embed", (1, n + 1, C), 0.02)
    mat("temporal_embed", (1, T, C), 0.02)
    mat("txt_proj.1.weight", (256, 768))
    mat("txt_proj.1.bias", (256,), 0.02)
    mat("vid_proj.0.weight", (256, 768))
    mat("vid_proj.0.bias", (256,), 0.02)
    ln("transformer.pre_norm", C)
    for l in range(cfg.dec_layers):
        b = f"transformer.decoder.layers.{l}."
        for at in ("multihead_attn", "self_attn"):
            mat(b + at + ".in_proj_weight", (3 * C, C))
            mat(b + at + ".in_proj_bias", (3 * C,), 0.02)
            mat(b + at + ".out_proj.weight", (C, C))
            mat(b + at + ".out_proj.bias", (C,), 0.02)
        mat(b + "linear1.weight", (cfg.dec_ffn, C))
        mat(b + "linear1.bias", (cfg.dec_ffn,), 0.02)
        mat(b + "linear2.weight", (C, cfg.dec_ffn))
        mat(b + "linear2.bias", (C,), 0.02)
        for nm in ("norm1", "norm2", "norm3"):
            ln(b + nm, C)
    ln("transformer.decoder.norm", C)
    mat("class_embed.weight", (cfg.num_classes + 1, C))
    mat("class_embed.bias", (cfg.num_classes + 1,), 0.02)
    for i, (a, o) in enumerate(((C, C), (C, C), (C, 4))):
        mat(f"bbox_embed.layers.{i}.weight", (o, a))
        mat(f"bbox_embed.layers.{i}.bias", (o,), 0.02)
    mat("query_embed.weight", (Q, C), query_std)
    mat("frame_index.weight", (T, C), 1.0)
    mat("frame_proj.weight", (C, 2 * C))
    mat("frame_proj.bias", (C,), 0.02)
    mat("proj.weight", (C, F))
    mat("obj_proj.0.weight", (C, C))
    mat("obj_proj.0.bias", (C,), 0.02)
    mat("obj_proj.2.weight", (256, C))
    mat("obj_proj.2.bias", (256,), 0.02)
    return sd


def make_batch(cfg: HHConfig, batch: int, seed: int = 0) -> dict:
    """One synthetic training batch with the contract of run/train.py:50-76 (after prepare_data).

    video  fp32 [B,T,3,H,H]  N(0,1) clipped to [-1.6, 2.2] (normalised-pixel range)
    text   int64 [5B,77]     SOT, 4-12 ids, EOT, zero pad; rephrase slots 1-4 empty ([SOT,EOT]) w.p. 0.3
    boxes  fp32 [B,T,4,4]    per frame 2 hand + 2 object boxes, xyxy in 224-px units, zero = absent (p=0.2)
    noun_vec [B,582] / verb_vec [B,118] multi-hot; nouns int64 [B,4] (0 = pad); all_nouns [582,768]
    """
    B, T, R = batch, cfg.num_frames, cfg.captions_per_clip
    H = cfg.img_size
    g = lambda name: _gen(seed + 104729, name)
    video = torch.randn((B, T, 3, H, H), generator=g("video")).clamp_(-1.6, 2.2)
    text = torch.zeros((B * R, cfg.context_length), dtype=torch.int64)
    lens = torch.randint(4, 13, (B * R,), generator=g("text_len"))
    ids = torch.randint(1, min(cfg.vocab_size, SOT) - 1, (B * R, 12), generator=g("text_ids"))
    empty = torch.rand((B * R,), generator=g("text_empty")) < 0.3
    sot = SOT if cfg.vocab_size > SOT else cfg.vocab_size - 2
    eot = EOT if cfg.vocab_size > EOT else cfg.vocab_size - 1
    for r in range(B * R):
        text[r, 0] = sot
        if r % R != 0 and bool(empty[r]):
            text[r, 1] = eot
        else:
            L = int(lens[r])
            text[r, 1:1 + L] = ids[r, :L]
            text[r, 1 + L] = eot
    corner = torch.rand((B, T, 4, 2), generator=g("box_corner")) * 150.0
    size = torch.rand((B, T, 4, 2), generator=g("box_size")) * 60.0 + 5.0
    boxes = torch.cat([corner, corner + size], dim=-1)
    drop = torch.rand((B, T, 4), generator=g("box_drop")) < 0.2
    boxes = boxes * (~drop)[..., None]
    noun_vec = (torch.rand((B, cfg.n_nouns), generator=g("noun_vec")) < 0.01).float()
    verb_vec = (torch.rand((B, cfg.n_verbs), generator=g("verb_vec")) < 0.01).float()
    nouns = torch.randint(1, cfg.n_nouns, (B, 4), generator=g("nouns"))
    nlen = torch.randint(1, 5, (B,), generator=g("nouns_len"))
    nouns = nouns * (torch.arange(4)[None, :] < nlen[:, None])
    all_nouns = torch.randn((cfg.n_nouns, 768), generator=g("all_nouns"))
    image_size = torch.full((B, 2), float(H))
    return {"video": video, "text": text, "boxes": boxes, "noun_vec": noun_vec, "verb_vec": verb_vec,
            "nouns": nouns.to(torch.int64), "all_nouns": all_nouns, "image_size": image_size}


def make_mcq_item(cfg: HHConfig, items: int, seed: int = 0) -> dict:
    """q EgoMCQ items: 5 candidate clips + 1 query text each (run/test_EgoMCQ.py:54-60)."""
    b = make_batch(cfg, 5 * items, seed + 31)
    text = b["text"][:: cfg.captions_per_clip][:items].clone()
    answer = torch.randint(0, 5, (items,), generator=_gen(seed, "mcq_answer"))
    types = torch.randint(1, 3, (items,), generator=_gen(seed, "mcq_type"))
    return {"video": b["video"].view(items, 5, *b["video"].shape[1:]), "text": text,
            "answer": answer, "type": types}
