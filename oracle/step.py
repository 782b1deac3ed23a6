"""Oracle: the training step and the EgoMCQ forward restated from the reference scripts (fp32, CPU).

Test infrastructure.  Follows /root/reference/run/train.py:103-203 (step glue, loss weights 1/1/0.5,
projections, pad mask, noun-id zeroing :73) and /root/reference/run/test_EgoMCQ.py:56-83.
The hard-coded `4` frames of the scripts (train.py:115,163) is `T` here.
"""
import torch

from . import decoder as D
from . import encoder as E
from . import losses as L

ZEROED_NOUNS = [102, 504, 364, 321, 556]            # run/train.py:73


def no_decay(name: str) -> bool:
    """optim_policy -- /root/reference/utils/train_utils.py:30-46 (substring match on the param name)."""
    return any(s in name for s in (".ln_", ".bn", ".bias", ".logit_scale", ".entropy_scale"))


def step_losses(enc_sd, dec_sd, batch, cfg, world_size=1):
    """One forward of the training step; returns dict of scalars / indices.  dec_sd tensors may require grad."""
    video, text = batch["video"], batch["text"]
    B, T = video.shape[0], video.shape[1]
    n = cfg.patches_per_frame
    with torch.no_grad():
        out = E.clip_forward(video, text, enc_sd, cfg)
    fmap, tmap = out["image_feature_map"], out["text_feature_map"]
    grid = fmap[:, 1:].reshape(B, T, n, -1)
    det, hs = D.objdecoder_forward(grid, dec_sd, cfg)
    eot = text.argmax(dim=-1)
    text_embeds = D.txt_proj(tmap[torch.arange(text.shape[0]), eot], dec_sd)
    obj = D.obj_proj(hs[-1], dec_sd)                                   # [B,Q,256]
    video_embeds = obj[:, -1]
    sim = L.sim_matrix(text_embeds, video_embeds)                      # [5B,B]
    noun_vec = batch["noun_vec"].clone()
    noun_vec[:, ZEROED_NOUNS] = 0
    sim_v = L.sim_matrix(batch["verb_vec"], batch["verb_vec"])
    sim_n = L.sim_matrix(noun_vec, noun_vec)
    pad = ((text != 0).sum(-1) != 2).float()[:, None].repeat(1, B)
    nce, _ = L.egonce(sim, sim_v, sim_n, pad)
    R = text.shape[0] // B
    acc_vt, acc_tv = L.compute_tv_accuracy(sim.view(B, R, B)[:, 0], text_embeds, sim_v, sim_n, B, R)
    hand = batch["boxes"][:, :, :2].flatten(0, 1)
    objb = batch["boxes"][:, :, 2:].flatten(0, 1)
    nq = cfg.num_queries if cfg.num_queries != 0 else 10
    lh, ih, dh = L.compute_box_loss("hand_boxes", det["pred_boxes"], hand, nq, world_size)
    lo, io, do = L.compute_box_loss("obj_boxes", det["pred_boxes"], objb, nq, world_size)
    noun_embeds = D.txt_proj(batch["all_nouns"], dec_sd)
    word, wassign = L.word_contrastive(noun_embeds, obj[:, :-1], batch["nouns"], return_assign=True)
    total = nce + lh + lo + 0.5 * word
    # the no-grad cardinality metric of the criterion (box_utils.py:142-154; computed per box type on the query slice of
    # split_detr_out :433-442, then dropped by compute_box_loss :457-461)
    with torch.no_grad():
        card_h = L.cardinality_error(det["pred_logits"][:, 0:2], L.prepare_targets(hand))
        card_o = L.cardinality_error(det["pred_logits"][:, 2:nq], L.prepare_targets(objb))
    return {"total_loss": total, "pred_logits": det["pred_logits"], "cardinality_error_hand_boxes": card_h,
            "cardinality_error_obj_boxes": card_o, "nce_loss": nce, "box_loss_hand": lh, "box_loss_obj": lo, "word_loss": word,
            "acc_vt": acc_vt, "acc_tv": acc_tv, "idx_hand": ih, "idx_obj": io, "word_assign": wassign,
            "pred_boxes": det["pred_boxes"], "hs": hs, "image_feature_map": fmap,
            "video_embeds": video_embeds, "text_embeds": text_embeds,
            "loss_bbox_hand": dh["loss_bbox"], "loss_giou_hand": dh["loss_giou"],
            "loss_bbox_obj": do["loss_bbox"], "loss_giou_obj": do["loss_giou"]}


def adamw_update(params, grads, opt_state=None, lr=3e-5, wd=1e-5, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW (defaults) with the two param groups of optim_policy (train_utils.py:28-48):
    weight decay 0 for names matching no_decay(), 1e-5 otherwise.  Updates `params` in place."""
    if opt_state is None:
        opt_state = {"step": 0, "m": {}, "v": {}}
    opt_state["step"] += 1
    t = opt_state["step"]
    with torch.no_grad():
        for k, g in grads.items():
            p = params[k]
            m = opt_state["m"].setdefault(k, torch.zeros_like(p))
            v = opt_state["v"].setdefault(k, torch.zeros_like(p))
            decay = 0.0 if no_decay(k) else wd
            p.mul_(1 - lr * decay)
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
            p.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)
    return opt_state


def train_step(enc_sd, dec_sd, batch, cfg, opt_state=None, lr=3e-5, wd=1e-5):
    """Forward + backward + AdamW (run/train.py:199-203,519-520; no scaler in fp32).  dec_sd is updated
    in place.  Returns (loss dict, grads dict, opt_state)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in dec_sd.items()}
    res = step_losses(enc_sd, params, batch, cfg)
    res["total_loss"].backward()
    grads = {k: p.grad for k, p in params.items() if p.grad is not None}
    opt_state = adamw_update(dec_sd, grads, opt_state, lr, wd)
    return res, grads, opt_state


def mcq_forward(enc_sd, dec_sd, video, text, cfg):
    """EgoMCQ scoring -- run/test_EgoMCQ.py:56-83, batched over q items.
    video [q,5,T,3,H,W], text [q,77] -> scores [q,5] (cosine similarity), prediction = argmax."""
    q = video.shape[0]
    T, n = video.shape[2], cfg.patches_per_frame
    with torch.no_grad():
        out = E.clip_forward(video.flatten(0, 1), text, enc_sd, cfg)
        grid = out["image_feature_map"][:, 1:].reshape(q * 5, T, n, -1)
        _, hs = D.objdecoder_forward(grid, dec_sd, cfg, compute_logits=False)
        te = D.txt_proj(out["text_feature_map"][torch.arange(q), text.argmax(-1)], dec_sd)
        ve = D.obj_proj(hs[-1], dec_sd)[:, -1].view(q, 5, -1)
        return L.sim_matrix(te[:, None], ve)[:, 0]
