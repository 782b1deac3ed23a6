"""Data parallelism for the decoder-only training step: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm, over xGMI; "gloo" in CPU tests).

The reference has NO gradient synchronisation (run/train.py:475); the design here is the
build's own:
  * parameters and gradients of the trainable decoder live in two flat fp32 arenas laid out in the order in which backward
    FINISHES the gradients (heads, query side, memory side last), so that AdamW is one fused launch over per-parameter segments
    (weight decay is a per-segment flag -- utils/train_utils.py:28-48) and all-reduce buckets are contiguous slices that become
    ready front to back,
  * buckets are all-reduced asynchronously on a communication stream as soon as every parameter in the bucket has
    its final gradient (post-accumulate hooks) -> overlapped with the rest of backward; mean over ranks,
  * the contrastive batch is gathered with ONE packed all-gather per step (embeddings + pad flags + verb/noun
    vectors + the step's three normaliser counts: hand boxes, object boxes, valid words -- the reference's scalar
    `num_boxes` all-reduces, box_utils.py:218-222, ride in the same rows); its backward returns W x the local slice,
    which under mean-reduction of parameter gradients makes W ranks equivalent to one process on the concatenated batch,
  * which parameters AdamW updates is decided on the device from per-parameter "received a gradient" flags that are
    all-reduced with the gradients (one small collective per step), so ranks whose graphs differ (a parameter unused on
    one rank only) still take identical optimizer decisions.
Collectives per step: 1 all-gather (forward) + #buckets gradient all-reduces + 1 flag all-reduce (backward).
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


class _GradSink:
    """Direct-write target of one parameter's gradient inside the arena (attached to the parameter as `_hh_sink`).

    The decoder's own autograd nodes (model/qside.py QueryStack, model/tfm_decoder.py _MemorySide) own ~120 single-use parameters.
    Handing their gradients to autograd costs one `grad += new` kernel each (AccumulateGrad into the pre-set arena view) plus the
    temporaries; instead a node whose parameters all have an ARMED sink writes its weight-gradient GEMMs / LayerNorm reductions
    straight into `view`, calls done() and returns None for that input.  Armed = FlatArena.zero_grad() has run (the arena is
    zero, so accumulating kernels may add into it) and this parameter has not received a gradient yet in this step -- a second
    backward in the same step (gradient accumulation) falls back to autograd's adds.

    PRECONDITION: a sunk parameter has exactly ONE consumer node per step (true for every parameter the decoder's nodes own: each
    nn.Parameter enters one QueryStack / _MemorySide).  done() reports the gradient FINAL -- under data parallelism its bucket may
    be all-reduced right away -- so a second direct write in the same step would land after / beside the collective and let ranks
    diverge silently; FlatArena._ready raises on it instead."""
    __slots__ = ("arena", "name", "view")

    def __init__(self, arena, name, view):
        self.arena, self.name, self.view = arena, name, view

    def armed(self):
        a = self.arena
        return a.sink_armed and self.name not in a.touched

    def claim(self):
        """A node is going to write this gradient itself: from now on only done() reports the parameter (autograd still fires the
        post-accumulate hooks for the None the node returns -- possibly before another node has written ITS part)."""
        self.arena.claimed.add(self.name)

    def unclaim(self):
        self.arena.claimed.discard(self.name)

    def done(self):
        self.arena._ready(self.name)


class FlatArena:
    """Flat fp32 parameter / gradient arenas over the trainable parameters of `module`.

    Parameters that never receive a gradient in the step (class_embed.*, vid_proj.* -- SURVEY Appendix A10) are left
    out (torch.optim.AdamW skips grad-less parameters too)."""

    SKIP_PREFIXES = ("class_embed.", "vid_proj.")

    def __init__(self, module, bucket_bytes=16 << 20, late=None):
        """late(name) -> True for parameters whose gradient is only final at the very end of backward (the decoder's memory side):
        they are laid out last, so that the buckets in front of them -- launched strictly in arena order, the same on every rank --
        can be all-reduced while the memory side is still being differentiated."""
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and not n.startswith(self.SKIP_PREFIXES)]
        rev = list(reversed(named))          # reverse registration order ~ order in which backward finishes gradients (heads first)
        late = late or (lambda n: False)
        self.entries = [e for e in rev if not late(e[0])] + [e for e in rev if late(e[0])]
        dev = named[0][1].device
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in self.entries]          # keep every view 16-byte aligned
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
        self._sizes = dict(zip((n for n, _ in self.entries), sizes))
        self.names = [n for n, _ in self.entries]
        # device-side segment table of hh_adamw_arena_step: one segment per parameter
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + sz)
        self.seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_decay = torch.tensor([0 if no_decay(n) else 1 for n in self.names], dtype=torch.int32, device=dev)
        self.seg_step = torch.zeros(len(self.names), dtype=torch.int32, device=dev)      # per-parameter AdamW step count (bias correction)
        self.seg_flag = torch.zeros(len(self.names), dtype=torch.float32, device=dev)    # this step's flags (all-reduced under DP)
        self.seg_coef = torch.zeros(2 * len(self.names), dtype=torch.float32, device=dev)
        self._flag_cache = {}
        self.grads_clean = True                                  # the gradient arena is all zero (fresh, or cleared by the last update)
        self.sink_armed = False                                  # direct gradient writes allowed (between zero_grad() and the update)
        self.claimed = set()                                     # parameters whose gradient a node writes straight into the arena this step
        self._sunk = set()                                       # ... and whose write has been reported final (_ready): at most once per step
        self.ready_callbacks = []                                # called with the parameter name when its gradient is final (comm layer)
        for n, p in self.entries:
            o, k = self.offsets[n]
            p._hh_sink = _GradSink(self, n, self.grads[o:o + k].view_as(p))
        self._touch_hooks = [p.register_post_accumulate_grad_hook(self._make_touch_hook(n)) for n, p in self.entries]

    def _make_touch_hook(self, name):
        def hook(param):
            self.grads_clean = False                 # a gradient landed (also from a diagnostic backward between two steps)
            if name not in self.claimed:
                self.touched.add(name)
        return hook

    def _ready(self, name):
        """A node wrote this parameter's gradient straight into the arena (no AccumulateGrad, so no post-accumulate hooks)."""
        if name in self._sunk:
            raise RuntimeError("FlatArena: the gradient of '%s' was written straight into the arena twice in one step; a sunk parameter must have "
                               "exactly one consumer node per step (_GradSink precondition) -- its bucket may already be in an all-reduce" % name)
        self._sunk.add(name)
        self.grads_clean = False
        self.touched.add(name)
        for cb in self.ready_callbacks:
            cb(name)

    def declare_unused(self, names):
        """The forward pass knows that these parameters get no gradient in this step ON ANY RANK (a shape-dependent branch every rank
        takes alike, e.g. the trajectory branch when the clip length differs from num_frames): report them to the communication layer
        now, so that their bucket -- and every later one, buckets launch strictly in arena order -- is not held back until finish().
        They stay un-`touched`: AdamW skips them as torch does."""
        for name in names:
            if name in self.offsets and name not in self.touched:
                for cb in self.ready_callbacks:
                    cb(name)

    @property
    def steps(self):
        """{parameter name: AdamW step count} read back from the device (synchronises: checkpointing / tests only)."""
        return dict(zip(self.names, self.seg_step.cpu().tolist()))

    def set_steps(self, steps):
        self.seg_step.copy_(torch.tensor([int(steps.get(n, 0)) for n in self.names], dtype=torch.int32))

    def local_flags(self):
        """fp32 [n_params] device tensor, 1 where this rank's backward wrote the parameter's gradient.  One tensor per distinct
        touched-set is built (a small H2D copy) and cached: in steady state this costs no transfer and no synchronisation."""
        key = frozenset(self.touched)
        t = self._flag_cache.get(key)
        if t is None:
            t = torch.tensor([1.0 if n in key else 0.0 for n in self.names], dtype=torch.float32).to(self.params.device)
            self._flag_cache[key] = t
        return t

    def zero_grad(self, force=True):
        """optimizer.zero_grad() (run/train.py:199).  force=False skips the fill when the last hh_adamw_arena_step already cleared
        the arena (TrainStep.step does that)."""
        self.touched.clear()
        self.claimed.clear()
        self._sunk.clear()
        if force or not self.grads_clean:
            self.grads.zero_()
        self.grads_clean = False                     # a backward is about to write it
        self.sink_armed = True
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
        self.active = True                                       # False: skip the gradient collectives (bench.py: exposed-communication A/B; ranks diverge)
        self.launched = 0                                        # gradient-bucket collectives issued so far (tests / diagnostics)
        self.flag_reduces = 0
        self._remaining = []
        self._bucket_of = {}
        for bi, (_, _, names) in enumerate(arena.buckets):
            for n in names:
                self._bucket_of[n] = bi
        self._hooks = []
        if self.enabled:
            for n, p in arena.entries:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))
            arena.ready_callbacks.append(self._param_ready)      # gradients written straight into the arena (no AccumulateGrad)
        self.reset()

    def reset(self):
        self._remaining = [len(names) for _, _, names in self.arena.buckets]
        self._next = 0                                           # buckets are launched strictly in arena order on every rank
        self._seen = set()
        self.pending = []

    def _make_hook(self, name):
        def hook(param):
            if not self.active or name in self.arena.claimed:     # claimed: the writing node reports it (arena._ready)
                return
            o, k = self.arena.offsets[name]
            if param.grad is not None and param.grad.data_ptr() != self.arena.grads.data_ptr() + o * 4:
                self.arena.grads[o:o + k].copy_(param.grad.reshape(-1))      # autograd produced a fresh tensor
                param.grad = self.arena.grads[o:o + k].view_as(param)
            self._param_ready(name)
        return hook

    def _param_ready(self, name):
        """The gradient of `name` is final for this step.  Idempotent: a parameter whose gradient went straight into the arena
        (_GradSink.done) is reported again by its post-accumulate hook, which autograd still fires for the None the node returned."""
        if not self.active or name in self._seen:
            return
        self._seen.add(name)
        self._remaining[self._bucket_of[name]] -= 1
        self._launch_ready()

    def _launch_ready(self):
        """Launch every leading bucket whose gradients are final.  The ORDER of collectives must be identical on all ranks even when
        their graphs differ (a parameter unused on one rank finishes its bucket only in finish()), so a bucket never overtakes an
        earlier one; the arena is laid out in backward order, so in the common case this is the order of completion anyway."""
        nb = len(self._remaining)
        while self._next < nb and self._remaining[self._next] == 0:
            self._launch(self._next)
            self._next += 1

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
        mean over ranks, and arena.seg_flag holds, per parameter, whether ANY rank's backward produced its gradient (a parameter
        unused on this rank only still receives the other ranks' averaged gradient, and every rank must then update it -- as
        torch DDP does; a rank-local decision would let weights, moments and step counts diverge silently)."""
        a = self.arena
        a.seg_flag.copy_(a.local_flags())
        if not self.enabled or not self.active:
            return
        self._remaining = [0] * len(self._remaining)
        self._launch_ready()
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                work = dist.all_reduce(a.seg_flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(a.seg_flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.flag_reduces += 1
        self.pending.append((work, None))
        for w, buf in self.pending:
            w.wait()
            if buf is not None and not self.avg and self.W > 1:
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
        x = x.contiguous()
        out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=group)         # one output buffer: no W temporaries + cat
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.b * ctx.r: ctx.b * (ctx.r + 1)] * ctx.W, None


def gather_contrastive(video_embeds, text_embeds, pad_flag, verb_vec, noun_vec, group=None, force=False, counts=None):
    """ONE packed all-gather of everything the step needs from the other ranks (run/train.py:126-140 uses 5-6 collectives for the
    contrastive batch, box_utils.py:218-222 one more per box type for `num_boxes`).

    video_embeds [b,E], text_embeds [R*b,E] (differentiable), pad_flag [R*b], verb_vec [b,V], noun_vec [b,Nn]; counts: optional fp32
    [k] of this rank's normaliser counts (TrainStep: matched hand boxes, object boxes, valid words), carried in k extra columns of
    the first local row.  Returns (ve, te, pf, vv, nv, sums) with the global tensors in rank-major order and sums [k] = the counts
    summed over ranks (None when counts is None).  force=True runs the collective in a 1-rank group too (RCCL smoke test)."""
    W, _ = world()
    if W == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return video_embeds, text_embeds, pad_flag, verb_vec, noun_vec, (None if counts is None else counts.float())
    b, E = video_embeds.shape
    Rb = text_embeds.shape[0]
    R = Rb // b
    # one row per clip: [video | R text embeds | R pad flags | verb | noun | counts (row 0 only)]
    cols = [video_embeds.float(), text_embeds.float().reshape(b, R * E), pad_flag.float().reshape(b, R), verb_vec.float(), noun_vec.float()]
    k = 0 if counts is None else counts.numel()
    if k:
        cpad = torch.zeros((b, k), dtype=torch.float32, device=video_embeds.device)
        cpad[0] = counts.detach().float()
        cols.append(cpad)
    row = torch.cat(cols, dim=1)
    g = _AllGatherScaled.apply(row, group)                                   # [W*b, ...]
    o = 0
    ve = g[:, o:o + E]; o += E
    te = g[:, o:o + R * E].reshape(-1, E); o += R * E
    pf = g[:, o:o + R].reshape(-1).detach(); o += R
    vv = g[:, o:o + verb_vec.shape[1]].detach(); o += verb_vec.shape[1]
    nv = g[:, o:o + noun_vec.shape[1]].detach(); o += noun_vec.shape[1]
    sums = g[:, o:o + k].detach().sum(0) if k else None
    return ve, te, pf, vv, nv, sums


def normaliser(global_count):
    """clamp(global count / world, 1): SetCriterion's num_boxes (box_utils.py:218-222) from an already reduced count."""
    W, _ = world()
    return torch.clamp(global_count / W, min=1)
