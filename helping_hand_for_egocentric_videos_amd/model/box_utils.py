"""Hungarian matching and DETR-style box criterion: mirror of the LIVE functions of
/root/reference/model/box_utils.py (HungarianMatcher :20-92, build_matcher :95, SetCriterion :99-238,
prepare_targets :249-279, split_detr_out :433-442, compute_box_loss :445-461).

Training path (`compute_box_loss`): prepare_targets + cost + exact LSAP run in ONE kernel per box type
(hh_match_boxes, one thread per frame, fp64 shortest augmenting path = scipy semantics, bit-exact indices),
matched-pair L1/GIoU sums and their gradient in hh_box_loss_fwd/bwd; num_boxes stays a device scalar.  None of
the reference's `.cpu()` / `.item()` syncs remain.  The list-of-dicts API of the reference is kept for callers
that want it (it has to sync to build Python lists).
"""
import torch
import torch.distributed as dist
from torch import nn

from .. import ops
from ..utils import box_ops


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


class MatchResult:
    """Device-resident matching of F frames (k target slots each): pred_idx/tgt_idx int64 [F,k], n int32 [F],
    tgt cxcywh [F,k,4], count int32 [F].  Iterating / indexing converts to the reference's list of
    (LongTensor pred, LongTensor tgt) tuples (this is the only place that syncs)."""

    def __init__(self, m):
        self.m = m
        self._list = None

    def to_list(self):
        if self._list is None:
            n = self.m["n"].cpu().tolist()
            p, t = self.m["pred_idx"].cpu(), self.m["tgt_idx"].cpu()
            self._list = [(p[f, :n[f]].clone(), t[f, :n[f]].clone()) for f in range(len(n))]
        return self._list

    def __iter__(self):
        return iter(self.to_list())

    def __len__(self):
        return self.m["n"].shape[0]

    def __getitem__(self, i):
        return self.to_list()[i]


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def match_raw(self, pred_boxes, q0, q, raw_boxes, img=224.0, count_out=None):
        """Fused prepare_targets + cost + LSAP on raw xyxy pixel boxes [F,k,4] (training path).  count_out: int32 [F] row to receive the
        per-frame target counts (the step keeps both box types' counts in one buffer and reduces them in one launch)."""
        return ops.match_boxes(pred_boxes.detach().float().contiguous(), q0, q, raw_boxes.float().contiguous(), img,
                               self.cost_bbox, self.cost_giou, count_out=count_out)

    @torch.no_grad()
    def match_list(self, outputs, targets, exclude_class=False):
        """Device-side matching for the reference's list-of-dicts targets; returns (match dict, count int32 [F])."""
        pb = outputs["pred_boxes"].detach().float().contiguous()
        F_, q = pb.shape[:2]
        tg, cnt, labels = _dense_targets(targets, pb.device)
        cc, wc = None, 0.0
        if not exclude_class:
            # box_utils.py:62,83-85: cost_class = -softmax(pred_logits)[:, tgt_ids], C += self.cost_class * cost_class
            prob = outputs["pred_logits"].detach().float().softmax(-1)                          # [F,q,classes]
            cc = (-prob.gather(2, labels[:, None, :].expand(-1, q, -1))).contiguous()          # [F,q,k]
            wc = self.cost_class
        m = ops.match_boxes(pb, 0, q, tg, 224.0, self.cost_bbox, self.cost_giou, given_count=cnt, class_cost=cc, w_class=wc)
        return m, cnt

    @torch.no_grad()
    def forward(self, outputs, targets, exclude_class=False):
        """Reference API (box_utils.py:43-92): targets = list of {'labels','boxes' cxcywh}; returns list of tuples.
        exclude_class=False adds the class-probability cost (never used by run/train.py, which passes True at :456)."""
        m, _ = self.match_list(outputs, targets, exclude_class)
        return MatchResult(m).to_list()


def _dense_targets(targets, device):
    """list of {'labels' [k_f], 'boxes' [k_f,4]} -> (boxes fp32 [F,k,4] zero padded, count int32 [F], labels int64 [F,k])."""
    F_ = len(targets)
    k = max(1, max(len(t["boxes"]) for t in targets))
    tg = torch.zeros((F_, k, 4), dtype=torch.float32, device=device)
    labels = torch.zeros((F_, k), dtype=torch.int64, device=device)
    cnt = torch.tensor([len(t["boxes"]) for t in targets], dtype=torch.int32, device=device)
    for f, t in enumerate(targets):
        n = len(t["boxes"])
        if n:
            tg[f, :n] = t["boxes"]
            if "labels" in t and t["labels"] is not None:
                labels[f, :n] = t["labels"].to(torch.int64)
    return tg, cnt, labels


def build_matcher(args):
    return HungarianMatcher(cost_class=1, cost_bbox=5, cost_giou=2)


class _MatchedBoxLoss(torch.autograd.Function):
    """(sum |p - t|, sum (1 - GIoU)) over matched pairs; backward = hh_box_loss_bwd."""

    @staticmethod
    def forward(ctx, pred, q0, m):
        p = pred.detach().float().contiguous()
        sums = ops.box_loss_fwd(p, q0, m)
        ctx.q0, ctx.m = q0, m
        ctx.save_for_backward(p)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_l1, g_giou):
        (p,) = ctx.saved_tensors
        dpred = torch.zeros_like(p)
        ops.box_loss_bwd(p, ctx.q0, ctx.m, g_l1.reshape(1).float().contiguous(), g_giou.reshape(1).float().contiguous(), dpred)
        return dpred, None, None


class SetCriterion(nn.Module):
    """box_utils.py:99-238 with losses in {'boxes','cardinality'} (loss_labels is unreachable in the reference)."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.matcher, self.weight_dict, self.eos_coef, self.losses = matcher, weight_dict, eos_coef, losses
        empty_weight = torch.ones(num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer('empty_weight', empty_weight)
        for l in losses:
            assert l in ('boxes', 'cardinality'), f'do you really want to compute {l} loss?'

    @staticmethod
    def num_boxes(count):
        """clamp(all_reduce(sum k_f) / world, 1) as a device scalar (box_utils.py:218-222, no .item())."""
        nb = count.sum().float().reshape(1)
        if is_dist_avail_and_initialized():
            dist.all_reduce(nb)
        return torch.clamp(nb / get_world_size(), min=1)[0]

    def forward_raw(self, outputs, q0, q, raw_boxes, box_type, num_boxes=None, match=None):
        """Training path: outputs['pred_boxes'] [F,Qtot,4], queries [q0,q0+q), raw xyxy boxes [F,k,4].  num_boxes: the normaliser as
        a device scalar if the caller has already reduced it; match: a match_raw result for the same slice if already computed."""
        pred = outputs['pred_boxes']
        m = self.matcher.match_raw(pred, q0, q, raw_boxes) if match is None else match
        nb = self.num_boxes(m["count"]) if num_boxes is None else num_boxes
        losses = {}
        if 'boxes' in self.losses:
            s_l1, s_giou = _MatchedBoxLoss.apply(pred, q0, m)
            losses[f'loss_bbox_{box_type}'] = s_l1 / nb
            losses[f'loss_giou_{box_type}'] = s_giou / nb
        if 'cardinality' in self.losses:
            with torch.no_grad():
                if outputs.get('pred_logits') is not None:
                    lg = outputs['pred_logits'][:, q0:q0 + q]
                    card = (lg.argmax(-1) != lg.shape[-1] - 1).sum(1)
                elif outputs.get('pred_logits_argmax') is not None:
                    card = (outputs['pred_logits_argmax'][:, q0:q0 + q] != outputs['num_classes'] - 1).sum(1)
                else:
                    card = None
                if card is not None:
                    losses[f'cardinality_error_{box_type}'] = (card.float() - m["count"].float()).abs().mean()
        return losses, MatchResult(m)

    def forward(self, outputs, targets, box_type, exclude_class=False):
        """Reference API (box_utils.py:206-238): targets = list of dicts (already prepared); the aux_outputs loop is reproduced
        (run/train.py never reaches it: split_detr_out empties the list)."""
        m, cnt = self.matcher.match_list({k: v for k, v in outputs.items() if k != 'aux_outputs'}, targets, exclude_class)
        nb = self.num_boxes(m["count"])
        losses = self._list_losses(outputs, m, cnt, nb, box_type)
        for i, aux in enumerate(outputs.get('aux_outputs') or []):
            ma, _ = self.matcher.match_list(aux, targets, exclude_class)
            losses.update({k + f'_{i}': v for k, v in self._list_losses(aux, ma, cnt, nb, box_type).items()})
        return losses, MatchResult(m).to_list()

    def _list_losses(self, outputs, m, cnt, nb, box_type):
        losses = {}
        if 'boxes' in self.losses:
            s_l1, s_giou = _MatchedBoxLoss.apply(outputs['pred_boxes'], 0, m)
            losses[f'loss_bbox_{box_type}'] = s_l1 / nb
            losses[f'loss_giou_{box_type}'] = s_giou / nb
        if 'cardinality' in self.losses and outputs.get('pred_logits') is not None:
            with torch.no_grad():
                lg = outputs['pred_logits']
                card = (lg.argmax(-1) != lg.shape[-1] - 1).sum(1)
                losses[f'cardinality_error_{box_type}'] = (card.float() - cnt.float()).abs().mean()
        return losses


@torch.no_grad()
def prepare_targets(boxes, classes, image_size, center_crop=True):
    """box_utils.py:249-279 (center_crop=False branch is the one the step uses).  Returns the reference's list of
    {'labels','boxes'} (syncs to build variable-length entries); the training path fuses this into hh_match_boxes."""
    if center_crop:
        raise NotImplementedError("prepare_targets: run/train.py always calls center_crop=False (box_utils.py:448)")
    if classes is None:
        classes = 1 - (boxes.sum(-1) != 0).float()
    # tensor/tensor division: torch's GPU div-by-scalar multiplies by a reciprocal, which is not bit-identical to the
    # reference's CPU `.div(224)` (box_utils.py:270)
    b = torch.clip(boxes, min=0, max=224) / torch.full((), 224.0, dtype=boxes.dtype, device=boxes.device)
    out = []
    for idx in range(classes.shape[0]):
        c_, b_ = classes[idx], b[idx]
        keep = (c_ != -1) & (b_[:, 2] > b_[:, 0]) & (b_[:, 3] > b_[:, 1])
        out.append({'labels': c_[keep], 'boxes': box_ops.box_xyxy_to_cxcywh(b_[keep, :])})
    return out


def split_detr_out(detr_out, start=0, end=2):
    """box_utils.py:433-442 (aux_outputs are emptied before they could be used -- reproduced)."""
    o = dict(detr_out)
    o['pred_boxes'] = detr_out['pred_boxes'][:, start:end, :]
    if detr_out.get('pred_logits') is not None:
        o['pred_logits'] = detr_out['pred_logits'][:, start:end]
    o['aux_outputs'] = []
    return o


class _BoxTail(torch.autograd.Function):
    """Both box losses of the training step as one node: hh_box_loss_fwd per box type + ONE hh_box_tail_fwd launch for every scalar that
    follows (loss_bbox / loss_giou per type, the weighted totals of compute_box_loss, both cardinality errors), instead of ~35 scalar
    and [frames]-sized stock ops and their autograd; backward = two hh_box_loss_bwd_scaled launches into one zeroed buffer."""

    @staticmethod
    def forward(ctx, pred, mh, mo, nq, num_boxes, argmax, no_object, w_l1, w_giou, denom):
        p = pred.detach().float().contiguous()
        out, coef = ops.box_tail_fwd(p, mh, mo, 0, 2, 2, nq - 2, num_boxes, argmax, no_object, w_l1, w_giou, denom)
        ctx.mh, ctx.mo = mh, mo
        ctx.save_for_backward(p, coef)
        rest = out[2:]                                   # (loss_bbox_h, loss_giou_h, loss_bbox_o, loss_giou_o, card_h, card_o): logged, no gradient
        ctx.mark_non_differentiable(rest)
        return out[0], out[1], rest

    @staticmethod
    def backward(ctx, g_h, g_o, _):
        p, coef = ctx.saved_tensors
        f = lambda g: g.reshape(1).float().contiguous()
        return ops.box_tail_bwd(p, ctx.mh, ctx.mo, 0, 2, f(g_h), f(g_o), coef), None, None, None, None, None, None, None, None, None


def step_box_losses(criterion, detr_out, hand_boxes, obj_boxes, n_queries, num_boxes, match_hand, match_obj):
    """compute_box_loss('hand_boxes', ...) + compute_box_loss('obj_boxes', ...) of run/train.py:161-183 for the training step's fast path
    (raw xyxy targets, matching already done, `num_boxes` fp32 [>= 2] = the world-averaged, clamped normalisers of both box types,
    class head returning `pred_logits_argmax`): -> (loss_hand, loss_obj, MatchResult hand, MatchResult obj, dict of the criterion's
    unweighted terms).  Same values as the two compute_box_loss calls (tests/test_step_gpu.py)."""
    wd = criterion.weight_dict
    w_l1, w_giou = wd["loss_bbox_hand_boxes"], wd["loss_giou_hand_boxes"]
    assert wd["loss_bbox_obj_boxes"] == w_l1 and wd["loss_giou_obj_boxes"] == w_giou, "one weight pair for both box types"
    argmax = detr_out.get("pred_logits_argmax") if "cardinality" in criterion.losses else None
    if argmax is not None:
        argmax = argmax.contiguous()
    no_object = (detr_out.get("num_classes") or 0) - 1
    lh, lo, out = _BoxTail.apply(detr_out["pred_boxes"], match_hand, match_obj, n_queries, num_boxes.float().contiguous(), argmax, no_object,
                                 float(w_l1), float(w_giou), len(wd) / 3)
    terms = {"loss_bbox_hand_boxes": out[0], "loss_giou_hand_boxes": out[1], "loss_bbox_obj_boxes": out[2], "loss_giou_obj_boxes": out[3]}
    if argmax is not None:
        terms["cardinality_error_hand_boxes"], terms["cardinality_error_obj_boxes"] = out[4], out[5]
    return lh, lo, MatchResult(match_hand), MatchResult(match_obj), terms


def compute_box_loss(box_type, criterion, detr_out, target_boxes, target_classes, all_image_size, n_queries=10, *, num_boxes=None,
                     match=None, return_loss_dict=False):
    """box_utils.py:445-461: ((5*L1 + 2*GIoU)/num_boxes summed) / (len(weight_dict)/3), plus the matching.

    target_boxes: raw xyxy pixel boxes [F,k,4] (hand: k=2 / object: k=2) exactly as run/train.py:161-181 passes them.
    num_boxes: the already world-averaged, clamped normaliser of box_utils.py:218-222 as a device scalar (the step reduces all of
    its target counts in one collective, parallel.gather_contrastive); None = compute (and all-reduce) it here as the reference does.
    match: the result of criterion.matcher.match_raw on the same slice, if the caller already ran it.
    return_loss_dict=True appends the criterion's unweighted loss dict (incl. the no-grad `cardinality_error_<type>` the reference
    computes at :142-154 and drops at :457-461)."""
    if target_classes is not None:
        raise NotImplementedError("compute_box_loss: run/train.py passes target_classes=None")
    if box_type == 'hand_boxes':
        q0, q = 0, 2
    elif box_type == 'obj_boxes':
        q0, q = 2, n_queries - 2
    elif box_type == 'all_boxes':
        q0, q = 0, detr_out['pred_boxes'].shape[1]
    else:
        raise ValueError(box_type)
    loss_dict, matched = criterion.forward_raw(detr_out, q0, q, target_boxes, box_type, num_boxes=num_boxes, match=match)
    wd = criterion.weight_dict
    total = sum(v * wd[k] for k, v in loss_dict.items() if k in wd) / (len(wd) / 3)
    return (total, matched, loss_dict) if return_loss_dict else (total, matched)
