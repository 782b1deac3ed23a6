"""Data parallelism for the decoder-only training step: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI; "gloo" in CPU tests).

The reference has NO gradient synchronisation (run/train.py:475); the design here is the
build's own:
  * parameters and gradients of the trainable decoder live in two flat fp32 arenas (decay group first, no-decay
    group second -- utils/train_utils.py:28-48) so that AdamW is two fused kernel launches and all-reduce buckets
    are contiguous slices,
  * buckets are all-reduced asynchronously on a communication stream as soon as every parameter in the bucket has
    its final gradient (post-accumulate hooks) -> overlapped with the rest of backward; mean over ranks,
  * the contrastive batch is gathered with ONE packed all-gather per step (embeddings + pad flags + verb/noun
    vectors); its backward returns W x the local slice, which under mean-reduction of parameter gradients makes
    W ranks equivalent to one process on the concatenated batch.
xGMI note: 110 MB of fp32 gradients in ~8 buckets of ~14 MB; each bucket is one RCCL all-reduce, far below the
>= 90 ms of compute per step, so exposed communication is only the last bucket.
"""
import torch
import torch.distributed as dist


def no_decay(name: str) -> bool:
    """optim_policy of the reference (utils/train_utils.py:30-46): substring match on the parameter name."""
    return any(s in name for s in ('.ln_', '.bn', '.bias', '.logit_scale', '.entropy_scale'))


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class FlatArena:
    """Flat fp32 parameter / gradient arenas over the trainable parameters of `module`.

    Parameters that never receive a gradient in the step (class_embed.*, vid_proj.* -- SURVEY Appendix A10) are left
    out (torch.optim.AdamW skips grad-less parameters too)."""

    SKIP_PREFIXES = ("class_embed.", "vid_proj.")

    def __init__(self, module, bucket_bytes=16 << 20):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and not n.startswith(self.SKIP_PREFIXES)]
        decay = [(n, p) for n, p in named if not no_decay(n)]
        nodecay = [(n, p) for n, p in named if no_decay(n)]
        # reverse registration order ~ order in which backward finishes gradients (heads first, memory side last)
        self.entries = list(reversed(decay)) + list(reversed(nodecay))
        self.n_decay = sum(p.numel() for _, p in decay)
        dev = named[0][1].device
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in self.entries]          # keep every view 16-byte aligned
        self.n_decay_padded = sum(s for s, (n, _) in zip(sizes, self.entries) if not no_decay(n))
        total = sum(sizes)
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = {}
        off = 0
        for (n, p), s in zip(self.entries, sizes):
            self.params[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + p.numel()].view_as(p)
            p.grad = self.grads[off:off + p.numel()].view_as(p)
            self.offsets[n] = (off, p.numel())
            off += s
        self.total = total
        # buckets: contiguous [start,end) ranges of ~bucket_bytes with the parameter names they contain
        self.buckets = []
        cur_start, cur_names, cur_bytes = 0, [], 0
        off = 0
        for (n, p), s in zip(self.entries, sizes):
            cur_names.append(n)
            cur_bytes += s * 4
            off += s
            if cur_bytes >= bucket_bytes:
                self.buckets.append((cur_start, off, cur_names))
                cur_start, cur_names, cur_bytes = off, [], 0
        if cur_names:
            self.buckets.append((cur_start, off, cur_names))
        # parameters whose gradient was written in the current step (torch.optim.AdamW skips grad-less parameters entirely, no
        # weight decay either: e.g. frame_index / frame_proj when a clip length != num_frames skips the trajectory branch)
        self.touched = set()
        self.steps = {n: 0 for n, _ in self.entries}            # per-parameter AdamW step count (bias correction)
        self._sizes = dict(zip((n for n, _ in self.entries), sizes))
        self._plan_cache = {}
        self._touch_hooks = [p.register_post_accumulate_grad_hook(self._make_touch_hook(n)) for n, p in self.entries]

    def _make_touch_hook(self, name):
        def hook(param):
            self.touched.add(name)
        return hook

    def update_plan(self):
        """AdamW launch plan for the parameters touched in this step: list of (start, end, weight_decayed, step) over contiguous
        arena ranges.  Adjacent entries of the same decay group and the same step count share a range -- when every parameter has a
        gradient every step (the normal case) that is two ranges.  Advances the per-parameter step counts."""
        key = frozenset(self.touched)
        ranges = self._plan_cache.get(key)
        if ranges is None:                                      # [start, end, decayed, names] ignoring step counts
            ranges, off = [], 0
            for n, _ in self.entries:
                sz = self._sizes[n]
                if n in key:
                    dec = not no_decay(n)
                    if ranges and ranges[-1][1] == off and ranges[-1][2] == dec:
                        ranges[-1][1] = off + sz
                        ranges[-1][3].append(n)
                    else:
                        ranges.append([off, off + sz, dec, [n]])
                off += sz
            self._plan_cache[key] = ranges
        plan = []
        for a, b, dec, names in ranges:
            if len({self.steps[n] for n in names}) == 1:
                plan.append((a, b, dec, self.steps[names[0]] + 1))
            else:                                               # parameters that skipped earlier steps: one launch per parameter
                for n in names:
                    o, _ = self.offsets[n]
                    plan.append((o, o + self._sizes[n], dec, self.steps[n] + 1))
        for n in key:
            self.steps[n] += 1
        return plan

    def zero_grad(self):
        self.touched.clear()
        self.grads.zero_()
        for n, p in self.entries:                    # autograd may have replaced .grad; re-point it at the arena
            o, k = self.offsets[n]
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + o * 4:
                p.grad = self.grads[o:o + k].view_as(p)


class BucketedAllReduce:
    """Asynchronous mean all-reduce of arena buckets, launched from post-accumulate-grad hooks on a communication stream.

    The mean is taken by the collective itself (ReduceOp.AVG) on RCCL; gloo (CPU tests) has no AVG, there each bucket is summed
    and scaled by 1/W when it has landed.  `force=True` runs the collectives even in a 1-rank group (used to exercise the real
    RCCL code path -- comm stream, hooks, bucket order -- on a single GPU)."""

    def __init__(self, arena: FlatArena, group=None, force=False):
        self.arena, self.group = arena, group
        self.W, _ = world()
        self.enabled = self.W > 1 or (force and dist.is_available() and dist.is_initialized())
        self.pending = []
        self.comm_stream = torch.cuda.Stream() if (self.enabled and arena.params.is_cuda) else None
        self.avg = self.enabled and dist.get_backend(group) == "nccl"
        self.launched = 0                                        # collectives issued so far (tests / diagnostics)
        self._remaining = []
        self._bucket_of = {}
        for bi, (_, _, names) in enumerate(arena.buckets):
            for n in names:
                self._bucket_of[n] = bi
        self._hooks = []
        if self.enabled:
            for n, p in arena.entries:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))
        self.reset()

    def reset(self):
        self._remaining = [len(names) for _, _, names in self.arena.buckets]
        self.pending = []

    def _make_hook(self, name):
        def hook(param):
            o, k = self.arena.offsets[name]
            if param.grad is not None and param.grad.data_ptr() != self.arena.grads.data_ptr() + o * 4:
                self.arena.grads[o:o + k].copy_(param.grad.reshape(-1))      # autograd produced a fresh tensor
                param.grad = self.arena.grads[o:o + k].view_as(param)
            bi = self._bucket_of[name]
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        s, e, _ = self.arena.buckets[bi]
        buf = self.arena.grads[s:e]
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        self.launched += 1
        self.pending.append((work, buf))

    def finish(self):
        """Wait for every bucket (launching the ones whose hooks never fired, e.g. unused parameters); gradients are then the
        mean over ranks."""
        if not self.enabled:
            return
        for bi, rem in enumerate(self._remaining):
            if rem > 0:
                self._remaining[bi] = 0
                self._launch(bi)
        for w, buf in self.pending:
            w.wait()
            if not self.avg and self.W > 1:
                buf.mul_(1.0 / self.W)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


class _AllGatherScaled(torch.autograd.Function):
    """all_gather along dim 0 whose backward returns W x the local slice of the incoming gradient.

    Every rank computes the identical loss on the gathered batch, so sum_ranks(slice) is the true gradient of the
    local slice; parameter gradients are then MEAN-reduced, hence the factor W.  (The
    reference's AllGather_multi.backward, run/train.py:42-47, returns the bare slice and never syncs gradients.)"""

    @staticmethod
    def forward(ctx, x, group):
        W, r = world()
        ctx.W, ctx.r, ctx.b = W, r, x.shape[0]
        out = [torch.empty_like(x) for _ in range(W)]
        dist.all_gather(out, x.contiguous(), group=group)
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.b * ctx.r: ctx.b * (ctx.r + 1)] * ctx.W, None


def gather_contrastive(video_embeds, text_embeds, pad_flag, verb_vec, noun_vec, group=None, force=False):
    """ONE packed all-gather of everything EgoNCE needs across ranks (run/train.py:126-140 uses 5-6 collectives).

    video_embeds [b,E], text_embeds [R*b,E] (differentiable), pad_flag [R*b], verb_vec [b,V], noun_vec [b,Nn].
    Returns the global tensors in rank-major order.  force=True runs the collective in a 1-rank group too (RCCL smoke test)."""
    W, _ = world()
    if W == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return video_embeds, text_embeds, pad_flag, verb_vec, noun_vec
    b, E = video_embeds.shape
    Rb = text_embeds.shape[0]
    R = Rb // b
    # one row per clip: [video | R text embeds | R pad flags | verb | noun]
    row = torch.cat([video_embeds.float(), text_embeds.float().reshape(b, R * E), pad_flag.float().reshape(b, R),
                     verb_vec.float(), noun_vec.float()], dim=1)
    g = _AllGatherScaled.apply(row, group)                                   # [W*b, ...]
    o = 0
    ve = g[:, o:o + E]; o += E
    te = g[:, o:o + R * E].reshape(-1, E); o += R * E
    pf = g[:, o:o + R].reshape(-1).detach(); o += R
    vv = g[:, o:o + verb_vec.shape[1]].detach(); o += verb_vec.shape[1]
    nv = g[:, o:o + noun_vec.shape[1]].detach()
    return ve, te, pf, vv, nv
