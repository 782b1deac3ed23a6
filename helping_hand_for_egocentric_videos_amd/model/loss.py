"""EgoNCE and word-level contrastive loss: mirror of /root/reference/model/loss.py (EgoNCE :8-70,
WordContrastiveLoss :72-106) without the reference's host round trips.

  * EgoNCE: the reference drops all-padding rows by boolean indexing (`x[mask]`, a device->host sync);
    here dropped rows are masked in place (identical value and gradient), so the step never syncs.
  * WordContrastiveLoss: scipy.optimize.linear_sum_assignment on `.cpu()` costs (loss.py:87-93) is replaced by the
    on-device exact LSAP kernel hh_lsap_rows (bit-identical assignment, csrc/match.hip).
  * Under data parallelism the word loss divides by the global valid-word count / W (the reference has no DP semantics): the
    returned value is this rank's share x W -- its MEAN over ranks is the global mean cross-entropy, which is what the mean-reduced
    gradients optimise (as `num_boxes` does for the box losses, box_utils.py:218-222).  The step passes the count it already
    gathered (parallel.gather_contrastive); stand-alone use all-reduces it here.
Small fp32 reductions (log-softmax over a [5B,B] matrix, CE over 582 nouns) stay on stock PyTorch-ROCm ops.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import ops
from .metric import sim_matrix


class EgoNCE(nn.Module):
    def __init__(self, temperature=0.07, noun=True, verb=True):
        super().__init__()
        self.noun, self.verb, self.temperature = noun, verb, temperature

    def forward(self, x, mask_v, mask_n, multi_pad_mask=None, strict_mask=False, vn_threshold=0, return_mask=True):
        """x [R*Bg, Bg] (or [Bg,Bg] when multi_pad_mask is None) -> (loss, mask_bool).

        return_mask=False skips the row compaction of mask_bool (needs a host sync) and returns the un-compacted mask."""
        T = self.temperature
        if multi_pad_mask is None:                                  # single positive sample (loss.py:16-24)
            diag = torch.eye(x.shape[0], device=x.device)
            if mask_v is not None and mask_n is not None:
                mask = mask_v * mask_n + diag
            elif mask_n is not None:
                mask = mask_n + diag
            elif mask_v is not None:
                mask = mask_v + diag
            else:
                mask = diag
            mb = mask > vn_threshold
            li = (torch.log_softmax(x / T, dim=1) * mb).sum(1) / mb.sum(-1)
            lj = (torch.log_softmax(x.t() / T, dim=1) * mb.t()).sum(1) / mb.sum(0)
            return -li.sum() / len(li) - lj.sum() / len(lj), mb
        pad = multi_pad_mask.bool()
        X = x.masked_fill(~pad, float('-inf'))
        Bg = x.shape[-1]
        R = multi_pad_mask.shape[0] // multi_pad_mask.shape[1]
        pos = torch.eye(Bg, device=x.device).repeat_interleave(R, dim=0)
        if mask_v is not None and mask_n is not None:
            extra = mask_v.repeat_interleave(R, dim=0) * mask_n.repeat_interleave(R, dim=0)
        elif mask_n is not None:
            extra = mask_n.repeat_interleave(5, dim=0)              # literal 5 of loss.py:47
        elif mask_v is not None:
            extra = mask_v.repeat_interleave(5, dim=0)
        else:
            extra = torch.zeros_like(pos)
        mask = (extra + pos) * multi_pad_mask
        keep = X.sum(-1) != float('-inf')                           # rows the reference removes (loss.py:42-56)
        mb = (mask > vn_threshold) & keep[:, None]
        zero = torch.zeros((), device=x.device, dtype=x.dtype)
        Xs = torch.where(keep[:, None], X, zero)                    # dropped rows -> finite dummy logits
        ls_i = torch.log_softmax(Xs / T, dim=1)
        li = torch.where(mb, ls_i, zero).sum(1) / mb.sum(-1).clamp(min=1)
        loss_i = torch.where(keep, li, zero).sum() / keep.sum()
        ls_j = torch.log_softmax(X.t() / T, dim=1)                  # dropped rows are -inf entries of every column
        lj = torch.where(mb.t(), ls_j, zero).sum(1) / mb.sum(0)
        loss_j = lj.sum() / len(lj)
        loss = -loss_i - loss_j
        return (loss, mb[keep]) if return_mask else (loss, mb)


class WordContrastiveLoss(nn.Module):
    def __init__(self, temperature=0.07, noun_threshold=0.6):
        super().__init__()
        self.temperature, self.noun_threshold = temperature, noun_threshold

    def forward(self, noun_embeds, pred_noun_embeds, noun_gt_inds, return_assignment=False, count=None):
        """noun_embeds [V,256], pred_noun_embeds [B,Q-1,256], noun_gt_inds int64 [B,W] (0 = pad) -> scalar.
        count: the normaliser (global valid-word count / world size) as a device scalar when the caller has already reduced it."""
        Bn, W = noun_gt_inds.shape
        gt = noun_embeds.index_select(0, noun_gt_inds.flatten()).view(Bn, W, -1)
        cost = -sim_matrix(gt, pred_noun_embeds)                                   # [B,W,Q-1]
        valid = noun_gt_inds != 0
        cols = ops.lsap_rows(cost.detach().float().contiguous(), valid)            # [B,W] int64, -1 on pad rows
        sel = torch.gather(pred_noun_embeds, 1, cols.clamp(min=0)[..., None].expand(-1, -1, pred_noun_embeds.shape[-1]))
        sim_all = sim_matrix(sel.reshape(Bn * W, -1), noun_embeds)                 # [B*W,V]
        noun_sim = sim_matrix(noun_embeds, noun_embeds).clone()
        noun_sim.fill_diagonal_(0)
        noun_mask = noun_sim.index_select(0, noun_gt_inds.flatten()) > self.noun_threshold
        ce = F.cross_entropy(sim_all.masked_fill(noun_mask, -1) / self.temperature, noun_gt_inds.flatten(), reduction='none')
        v = valid.flatten()
        if count is not None:
            count = count.clamp(min=1)                  # a rank (or the whole global batch) without valid words contributes 0, not NaN
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # data parallel: normalise by the mean word count over ranks (as SetCriterion.num_boxes does, box_utils.py:218-222), so
            # that mean-reduced parameter gradients equal those of one process on the concatenated batch
            count = v.sum().float().reshape(1)
            dist.all_reduce(count)
            count = (count[0] / dist.get_world_size()).clamp(min=1)
        else:
            count = v.sum().float()                     # single process: the reference's plain mean over valid words (loss.py:104)
        loss = torch.where(v, ce, torch.zeros((), device=ce.device)).sum() / count
        return (loss, cols) if return_assignment else loss
