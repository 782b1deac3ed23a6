"""Helpers that drive the IMPORTED reference modules (dev container only) on synthetic state dicts.

This is the reference-side counterpart of oracle/step.py: it builds the reference nn.Modules
(bypassing the network-bound factory), loads the synthetic weights, and runs
the step glue of /root/reference/run/train.py:103-192 against them.  Used by
tests/test_oracle_vs_reference.py and tests/golden/make_golden.py.  Never runs on the GPU box.
"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _refload  # noqa: E402


def build_backbone(R, cfg, sd):
    with contextlib.redirect_stdout(io.StringIO()):
        vis = R.LaviLa.SpaceTimeTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
            num_heads=cfg.num_heads, num_frames=cfg.num_frames, time_init="zeros",
            attention_style="frozen-in-time", ln_pre=True, act_layer=R.openai_model.QuickGELU)
        vis.head = torch.nn.Identity()
        vis.pre_logits = torch.nn.Identity()
        vis.fc = torch.nn.Identity()
        clip = R.LaviLa.CLIP(embed_dim=cfg.project_embed_dim, vision_width=cfg.embed_dim, vision_model=vis,
                             context_length=cfg.context_length, vocab_size=cfg.vocab_size,
                             transformer_width=cfg.text_width, transformer_heads=cfg.text_heads,
                             transformer_layers=cfg.text_layers, tempearture_init=0.07)
    missing, unexpected = clip.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("attn_mask") for k in missing), missing
    return clip.eval()


def build_decoder(R, cfg, sd, feature_dim=None):
    tfm = R.tfm_decoder.Cross_Attention(normalize_before=True, return_intermediate_dec=True)
    dec = R.tfm_decoder.ObjDecoder(transformer=tfm, num_classes=cfg.num_classes, num_queries=cfg.dec_queries,
                                   aux_loss=True, pred_traj=True, feature_dim=feature_dim or cfg.embed_dim,
                                   num_frames=cfg.num_frames, patches_per_frame=cfg.patches_per_frame)
    dec.load_state_dict(sd, strict=True)
    return dec.eval()          # eval(): dropout off -- parity is defined at p=0


def build_criterion(R):
    matcher = R.box_utils.build_matcher(None)
    wd = {"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5, "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}
    return R.box_utils.SetCriterion(22047, matcher=matcher, weight_dict=wd, eos_coef=0.1,
                                    losses=["boxes", "cardinality"])


@contextlib.contextmanager
def no_cuda_calls():
    """prepare_targets() calls .cuda() (box_utils.py:255); make it the identity on this CPU-only box."""
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def reference_step(R, backbone, decoder, criterion, batch, cfg):
    """run/train.py:103-192 against the reference modules, world_size 1, fp32, T-generic harness."""
    video, text = batch["video"], batch["text"]
    B, T = video.shape[0], video.shape[1]
    with torch.no_grad():
        out = backbone(video, text, return_feature_map=True)
    fmap, tmap = out["image_feature_map"], out["text_feature_map"]
    grid = fmap[:, 1:].reshape(B, T, cfg.patches_per_frame, -1)
    det, hs, _, _ = decoder(grid)
    if T != 4:   # reference literal `4` (tfm_decoder.py:216): rebuild pred_logits with T so the matcher sees bs=B*T
        logits = decoder.class_embed(hs)
        det["pred_logits"] = logits[-1][:, None].expand(-1, T, -1, -1).flatten(0, 1)
    eot = text.argmax(dim=-1)
    text_embeds = decoder.txt_proj(tmap[torch.arange(text.shape[0]), eot])
    video_embeds = decoder.obj_proj(hs[-1])[:, -1]
    sim = R.metric.sim_matrix(text_embeds, video_embeds)
    noun_vec = batch["noun_vec"].clone()
    noun_vec[:, [102, 504, 364, 321, 556]] = 0
    sim_v = R.metric.sim_matrix(batch["verb_vec"], batch["verb_vec"])
    sim_n = R.metric.sim_matrix(noun_vec, noun_vec)
    pad = ((text != 0).sum(-1) != 2).float()[:, None].repeat(1, B)
    nce, _ = R.loss.EgoNCE()(sim, sim_v, sim_n, multi_pad_mask=pad, strict_mask=True)
    acc_vt, acc_tv = R.metric.compute_tv_accuracy(sim.view(B, -1, B)[:, 0, :], text_embeds, sim_v, sim_n, B, "cpu")
    hand = batch["boxes"][:, :, :2].flatten(0, 1).clone()
    objb = batch["boxes"][:, :, 2:].flatten(0, 1).clone()
    size = batch["image_size"][:, None, :].expand(-1, T, -1).flatten(0, 1)
    nq = cfg.num_queries
    with no_cuda_calls():
        lh, ih = R.box_utils.compute_box_loss("hand_boxes", criterion, det, hand, None, size, n_queries=nq)
        lo, io_ = R.box_utils.compute_box_loss("obj_boxes", criterion, det, objb, None, size, n_queries=nq)
        # the cardinality metric (box_utils.py:142-154) is computed by the criterion and dropped by compute_box_loss (:455-461):
        # call the criterion the way compute_box_loss does (:447-454) and keep it
        card = {}
        for bt, raw, (s0, s1) in (("hand_boxes", hand, (0, 2)), ("obj_boxes", objb, (2, nq))):
            tg = R.box_utils.prepare_targets(raw.clone(), None, size, center_crop=False)
            ld, _ = criterion(R.box_utils.split_detr_out(det, start=s0, end=s1), tg, bt, exclude_class=True)
            card[bt] = ld[f"cardinality_error_{bt}"]
    noun_embeds = decoder.txt_proj(batch["all_nouns"])
    word = R.loss.WordContrastiveLoss()(noun_embeds, decoder.obj_proj(hs[-1])[:, :-1], batch["nouns"])
    total = nce + lh + lo + 0.5 * word
    return {"total_loss": total, "nce_loss": nce, "box_loss_hand": lh, "box_loss_obj": lo, "word_loss": word,
            "acc_vt": acc_vt, "acc_tv": acc_tv, "idx_hand": ih, "idx_obj": io_, "pred_boxes": det["pred_boxes"],
            "hs": hs, "image_feature_map": fmap, "video_embeds": video_embeds, "text_embeds": text_embeds,
            "pred_logits": det["pred_logits"], "cardinality_error_hand_boxes": card["hand_boxes"],
            "cardinality_error_obj_boxes": card["obj_boxes"]}
