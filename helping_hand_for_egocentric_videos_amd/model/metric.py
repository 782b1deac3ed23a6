"""Metrics on the hot path: mirror of /root/reference/model/metric.py:363-392,209-225 (sim_matrix,
compute_tv_accuracy, egomcq_accuracy_metrics).  Legacy retrieval metrics of the reference are out of scope."""
import torch
import torch.nn.functional as F

from .. import ops


class _RowNorm(torch.autograd.Function):
    """x / max(||x||_2, eps) along the last dim on hh_rownorm_fwd / hh_rownorm_bwd (one launch each instead of norm + clamp + div
    and their five-op backward); fixed summation order, so a row-strided view gives the same bits as a dense copy."""

    @staticmethod
    def forward(ctx, x, eps):
        x2 = x.detach().reshape(-1, x.shape[-1]) if x.dim() != 2 else x.detach()
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y, nrm = ops.rownorm_fwd(x2, eps)
        ctx.save_for_backward(y, nrm)
        ctx.eps, ctx.shape = eps, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        return ops.rownorm_bwd(y, nrm, dy.reshape(y.shape), ctx.eps).view(ctx.shape), None


def _unit_rows(x, eps):
    if x.is_cuda and x.dtype == torch.float32:
        return _RowNorm.apply(x, eps)
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


_ZEROS = {}


def _pad_zero(x, pad_cols, pad_rows):
    """x [r, c] with pad_cols zero columns / pad_rows zero rows appended.  One padded side: a concatenation with a cached block of zeros (one
    launch; F.pad is a fill + a copy); both: F.pad."""
    if not pad_cols and not pad_rows:
        return x
    if pad_cols and pad_rows:
        return F.pad(x, (0, pad_cols, 0, pad_rows))
    shape = (x.shape[0], pad_cols) if pad_cols else (pad_rows, x.shape[1])
    key = (shape, x.dtype, x.device)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(shape, dtype=x.dtype, device=x.device)
    return torch.cat((x, _ZEROS[key]), dim=1 if pad_cols else 0)


def sim_matrix(a, b, eps=1e-8, norm=True):
    """Cosine similarity with eps-clamped norms (metric.py:363-375).  On the GPU the operands are normalised by hh_rownorm_fwd, the 2-D
    product runs on hh_qgemm_f32x3 (fp32-grade, differentiable; the contraction is zero-padded to a multiple of 4), the small batched
    form of the word loss as a broadcast multiply + sum -- no vendor BLAS on the step.  CPU tensors (tests, host-side use) take
    torch's ops."""
    same = a is b                                     # (sim_matrix(v, v): normalise and pad once)
    if norm:
        a = _unit_rows(a, eps)
        b = a if same else _unit_rows(b, eps)
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32:
        if a.dim() == 2 and b.dim() == 2:
            from .qside import linear_x3
            n = b.shape[0]
            pad_k, pad_n = (-a.shape[-1]) % 4, (-n) % 4              # the kernel wants its contiguous dimensions in multiples of 4
            if same and pad_k and not pad_n:
                a = b = _pad_zero(a, pad_k, 0)
            elif pad_k or pad_n:
                a, b = _pad_zero(a, pad_k, 0), _pad_zero(b, pad_k, pad_n)
            out = linear_x3(a, b)
            return out[:, :n] if pad_n else out
        if a.dim() == 3 and b.dim() == 3 and a.shape[0] * a.shape[1] * b.shape[1] * a.shape[2] <= (1 << 22):
            return (a[:, :, None, :] * b[:, None, :, :]).sum(-1)
    return a @ b.transpose(-1, -2)


def compute_tv_accuracy(similarity, text_embeds, sim_v, sim_n, num_samples, device=None):
    """metric.py:378-392: top-1 text<->video accuracy counting verb/noun-sharing and duplicate-text clips as hits
    (`[::5]` = first of the 5 rephrases).  Stays on device, no host sync."""
    if similarity.is_cuda and similarity.dtype == torch.float32:
        # one launch (hh_tv_accuracy) after the text-text cosine matrix: both argmaxes, the positives and the two means
        acc = ops.tv_accuracy(similarity, sim_matrix(text_embeds[::5], text_embeds[::5]).contiguous(), sim_v.float().contiguous(),
                              sim_n.float().contiguous())
        return acc[0], acc[1]
    tv_argmax = similarity.argmax(dim=-1)
    vt_argmax = similarity.argmax(dim=0)
    same = sim_matrix(text_embeds[::5], text_embeds[::5]) > 0.99
    ar = torch.arange(num_samples, device=similarity.device)
    same.fill_diagonal_(False)                       # (index_put_ with a Python scalar is an H2D copy + host sync)
    pos = (((sim_v * sim_n) + torch.eye(num_samples, device=similarity.device)) + same) > 0
    return pos[vt_argmax, ar].float().mean(), pos[ar, tv_argmax].float().mean()


def egomcq_accuracy_metrics(preds, labels, types):
    """metric.py:209-225: accuracy per question type (1 = Inter... group names as in the reference)."""
    metrics = {}
    for type_i, group_i in zip(torch.unique(types), ["Intra-video", "Inter-video"]):
        m = types == type_i
        metrics[group_i] = float((preds[m].argmax(-1) == labels[m]).float().mean()) * 100
    return metrics
