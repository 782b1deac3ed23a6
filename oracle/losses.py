"""Oracle: EgoNCE / word-contrastive / box losses and Hungarian matching (fp32, CPU).  Test infrastructure.

Restates /root/reference/model/loss.py, model/box_utils.py (live functions), utils/box_ops.py:9-61,
model/metric.py:363-392,209-225.  Matching indices are int64, rows ascending (scipy convention).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .lsap import linear_sum_assignment


# ---------------------------------------------------------------- box ops (utils/box_ops.py:9-61)
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_iou(a, b):
    """box_ops.py:24-37: IoU with `union + 1e-4` in the denominator; returns (iou, union)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b - inter
    return inter / (union + 0.0001), union


def generalized_box_iou(a, b):
    """box_ops.py:40-61: GIoU = IoU - (hull - union)/hull (no eps on the hull term)."""
    assert (a[:, 2:] >= a[:, :2]).all() and (b[:, 2:] >= b[:, :2]).all()
    iou, union = box_iou(a, b)
    lt = torch.min(a[:, None, :2], b[:, :2])
    rb = torch.max(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


# ---------------------------------------------------------------- metric.py:363-375
def sim_matrix(a, b, eps=1e-8):
    an = a / a.norm(dim=-1, keepdim=True).clamp_min(eps)
    bn = b / b.norm(dim=-1, keepdim=True).clamp_min(eps)
    return an @ bn.transpose(-1, -2)


# ---------------------------------------------------------------- loss.py:15-70 (multi-positive branch)
def egonce(x, mask_v, mask_n, multi_pad_mask, temperature=0.07):
    """EgoNCE.forward with multi_pad_mask given and both verb/noun masks (the call at run/train.py:146).

    x [R*Bg, Bg] text-major similarity, rows ordered clip-major / rephrase-minor; returns (loss, mask_bool).
    """
    pad = multi_pad_mask.bool()
    X = x.masked_fill(~pad, float("-inf"))
    Bg = x.shape[1]
    R = x.shape[0] // Bg
    pos = torch.eye(Bg, device=x.device).repeat_interleave(R, dim=0)
    mv = mask_v.repeat_interleave(R, dim=0)
    mn = mask_n.repeat_interleave(R, dim=0)
    mask = (mv * mn + pos) * multi_pad_mask
    keep = X.sum(-1) != float("-inf")
    mask, X = mask[keep], X[keep]
    mb = mask > 0
    li = (torch.log_softmax(X / temperature, dim=1) * mb).sum(1) / mb.sum(-1)
    lj = (torch.log_softmax(X.t() / temperature, dim=1) * mb.t()).sum(1) / mb.sum(0)
    return -li.sum() / len(li) - lj.sum() / len(lj), mb


# ---------------------------------------------------------------- loss.py:78-106
def word_contrastive(noun_embeds, pred_noun_embeds, noun_gt_inds, temperature=0.07, noun_threshold=0.6,
                     return_assign=False):
    gt = noun_embeds[noun_gt_inds.flatten()].view(*noun_gt_inds.shape, -1)
    cost = -sim_matrix(gt, pred_noun_embeds)                                  # [B,4,Q-1]
    sel, assign = [], []
    for b in range(noun_gt_inds.shape[0]):
        valid = noun_gt_inds[b] != 0
        if int(valid.sum()) == 0:
            assign.append(np.zeros(0, dtype=np.int64))
            continue
        _, cols = linear_sum_assignment(cost[b][valid].detach().numpy())
        assign.append(cols)
        sel.append(pred_noun_embeds[b][torch.as_tensor(cols)])
    sel = torch.cat(sel)
    gt_ids = noun_gt_inds[noun_gt_inds != 0]
    logits = sim_matrix(sel, noun_embeds)
    ns = sim_matrix(noun_embeds, noun_embeds).clone()
    ns.fill_diagonal_(0)
    logits = logits.masked_fill(ns[gt_ids] > noun_threshold, -1) / temperature
    loss = F.cross_entropy(logits, gt_ids)
    return (loss, assign) if return_assign else loss


# ---------------------------------------------------------------- box_utils.py:249-279
def prepare_targets(boxes, img=224.0):
    """prepare_targets(boxes, None, _, center_crop=False): boxes [F,k,4] xyxy px -> list of cxcywh [k_f,4]."""
    cls = 1.0 - (boxes.sum(-1) != 0).float()
    b = boxes.clamp(0, img) / img
    out = []
    for f in range(b.shape[0]):
        keep = (cls[f] != -1) & (b[f, :, 2] > b[f, :, 0]) & (b[f, :, 3] > b[f, :, 1])
        out.append(box_xyxy_to_cxcywh(b[f][keep]))
    return out


def matcher_cost(pred, tgt, w_bbox=5.0, w_giou=2.0):
    """HungarianMatcher cost with exclude_class=True -- box_utils.py:74-81 (cdist p=1 + GIoU)."""
    cb = torch.cdist(pred, tgt, p=1)
    cg = -generalized_box_iou(box_cxcywh_to_xyxy(pred), box_cxcywh_to_xyxy(tgt))
    return w_bbox * cb + w_giou * cg


def hungarian_match(pred_boxes, targets, w_bbox=5.0, w_giou=2.0, pred_logits=None, labels=None, w_class=1.0):
    """HungarianMatcher.forward -- box_utils.py:63-92; per-frame block-diagonal problems.
    pred_boxes [F,q,4]; targets list of [k_f,4] -> list of (int64 rows, int64 cols).
    pred_logits [F,q,classes] + labels (list of int64 [k_f]) = the exclude_class=False branch (:62,83-85):
    C += cost_class * (-softmax(logits)[:, tgt_ids])."""
    out = []
    for f, t in enumerate(targets):
        C = matcher_cost(pred_boxes[f].detach(), t, w_bbox, w_giou) if len(t) else torch.zeros((pred_boxes.shape[1], 0))
        if pred_logits is not None and len(t):
            C = C + w_class * (-pred_logits[f].detach().softmax(-1)[:, labels[f]])
        C = C.numpy()
        r, c = linear_sum_assignment(C)
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
    return out


def box_losses(pred_boxes, targets, indices, world_size=1, num_boxes_allreduce=None):
    """SetCriterion.loss_boxes -- box_utils.py:156-173, with num_boxes of :218-222."""
    num_boxes = float(sum(len(t) for t in targets))
    if num_boxes_allreduce is not None:
        num_boxes = float(num_boxes_allreduce)
    num_boxes = max(num_boxes / world_size, 1.0)
    src = torch.cat([pred_boxes[f][r] for f, (r, _) in enumerate(indices)])
    tgt = torch.cat([targets[f][c] for f, (_, c) in enumerate(indices)])
    l1 = (src - tgt).abs().sum() / num_boxes
    giou = (1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)))).sum() / num_boxes
    return l1, giou, num_boxes


def cardinality_error(pred_logits, targets):
    """SetCriterion.loss_cardinality -- box_utils.py:142-154 (no-grad metric)."""
    lens = torch.as_tensor([len(t) for t in targets], dtype=torch.float32)
    card = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1).float()
    return F.l1_loss(card, lens)


def compute_box_loss(box_type, pred_boxes, target_boxes, n_queries, world_size=1):
    """compute_box_loss -- box_utils.py:445-461: slice queries [0:2] / [2:n_queries], match, and
    return ((5*L1 + 2*GIoU)/(4/3), indices, dict).  aux losses are never computed (:437-441)."""
    targets = prepare_targets(target_boxes)
    sl = slice(0, 2) if box_type == "hand_boxes" else slice(2, n_queries)
    pb = pred_boxes[:, sl]
    idx = hungarian_match(pb, targets)
    l1, giou, nb = box_losses(pb, targets, idx, world_size)
    return (5.0 * l1 + 2.0 * giou) / (4.0 / 3.0), idx, {"loss_bbox": l1, "loss_giou": giou, "num_boxes": nb}


# ---------------------------------------------------------------- metric.py:378-392, 209-225
def compute_tv_accuracy(similarity, text_embeds, sim_v, sim_n, num_samples, R=5):
    tv_arg = similarity.argmax(dim=-1)
    vt_arg = similarity.argmax(dim=0)
    same = sim_matrix(text_embeds[::R], text_embeds[::R]) > 0.99
    same.fill_diagonal_(False)
    pos = (((sim_v * sim_n) + torch.eye(num_samples)) + same) > 0
    acc_vt = pos[vt_arg, torch.arange(num_samples)].float().mean()
    acc_tv = pos[torch.arange(num_samples), tv_arg].float().mean()
    return acc_vt, acc_tv


def egomcq_accuracy(preds, labels, types):
    out = {}
    for ty, name in zip(torch.unique(types).tolist(), ["Intra-video", "Inter-video"]):
        m = types == ty
        out[name] = float((preds[m].argmax(-1) == labels[m]).float().mean()) * 100
    return out
