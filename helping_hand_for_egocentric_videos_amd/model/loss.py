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
The step's call of EgoNCE (a per-row pad flag) and the 582-way masked cross-entropy run on the fused kernels of csrc/loss.hip
(hh_egonce_fwd, hh_masked_ce_fwd: value and gradient in one pass); the general forms stay on stock PyTorch-ROCm ops.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .. import ops
from .metric import sim_matrix


class _EgoNCERows(torch.autograd.Function):
    """EgoNCE for multi_pad_mask = row_pad[:, None].repeat(1, Bg) on hh_egonce_fwd: the loss and d loss / d x come out of the same
    three launches; the backward is one multiply by the incoming scalar."""

    @staticmethod
    def forward(ctx, x, mask_v, mask_n, row_pad, temperature, thr):
        xd = x.detach()
        if xd.stride(-1) != 1 or xd.dtype != torch.float32:
            xd = xd.float().contiguous()
        f = lambda t: None if t is None else t.detach().float().contiguous()
        R = xd.shape[0] // xd.shape[1]
        loss, grad = ops.egonce_fwd(xd, f(mask_v), f(mask_n), f(row_pad), R, temperature, thr)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


class _MaskedCE(torch.autograd.Function):
    """Per-row cross-entropy of the word loss (loss.py:95-104) on hh_masked_ce_fwd: ce [rows] (0 on invalid rows) and its Jacobian row
    d ce[r] / d sim[r, :] in one launch."""

    @staticmethod
    def forward(ctx, sim, noun_sim, gt, valid, temperature, threshold):
        sd = sim.detach()
        if sd.stride(-1) != 1:
            sd = sd.contiguous()
        ce, grad = ops.masked_ce_fwd(sd, noun_sim.detach().contiguous(), gt.contiguous(), valid.contiguous(), temperature, threshold)
        ctx.save_for_backward(grad)
        return ce

    @staticmethod
    def backward(ctx, dce):
        (grad,) = ctx.saved_tensors
        return grad * dce[:, None], None, None, None, None, None


class EgoNCE(nn.Module):
    def __init__(self, temperature=0.07, noun=True, verb=True):
        super().__init__()
        self.noun, self.verb, self.temperature = noun, verb, temperature

    def forward_rows(self, x, mask_v, mask_n, row_pad, vn_threshold=0):
        """forward(x, mask_v, mask_n, multi_pad_mask=row_pad[:, None].repeat(1, Bg), strict_mask=True) -- the call of
        run/train.py:144-149, where a caption is either present or absent for all clips -- as one fused node (hh_egonce_fwd).
        x [R*Bg, Bg], row_pad [R*Bg] -> loss."""
        if not x.is_cuda:
            return self.forward(x, mask_v, mask_n, multi_pad_mask=row_pad[:, None].repeat(1, x.shape[1]), strict_mask=True,
                                vn_threshold=vn_threshold, return_mask=False)[0]
        return _EgoNCERows.apply(x, mask_v, mask_n, row_pad, self.temperature, float(vn_threshold))

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
        v = valid.flatten()
        with torch.no_grad():                                                      # (a mask source: loss.py:95-97)
            noun_sim = sim_matrix(noun_embeds.detach(), noun_embeds.detach())
        # masked_fill(noun_sim[gt] > threshold with a zeroed diagonal, -1) / T -> cross-entropy, value and gradient in one launch
        ce = _MaskedCE.apply(sim_all, noun_sim, noun_gt_inds.flatten(), v, self.temperature, self.noun_threshold)
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
        loss = ce.sum() / count                                                    # (ce is 0 on the padding rows)
        return (loss, cols) if return_assignment else loss
