"""Training step and EgoMCQ forward harness: the build's counterpart of /root/reference/run/train.py:103-203
and /root/reference/run/test_EgoMCQ.py:56-83 (those scripts need datasets / checkpoints / CUDA and hard-code
4 frames -- SURVEY.md sections 0.6 and 8c -- so the composition is restated here, T-generic).

Step = frozen TimeSformer forward (HIP) + text tower + object-query decoder forward/backward (HIP) + EgoNCE +
hand/object box losses with on-device Hungarian matching (HIP) + word loss + gradient all-reduce (RCCL) +
fused AdamW (HIP).  bf16 compute, fp32 accumulation / master weights, no GradScaler.
No host synchronisation inside the step: every scalar in the returned dict is a device tensor.
"""
import torch

from . import ops
from .model import box_utils, tfm_decoder
from .model.loss import EgoNCE, WordContrastiveLoss
from .model.metric import compute_tv_accuracy, sim_matrix
from .parallel import BucketedAllReduce, FlatArena, gather_contrastive, no_decay, normaliser, world

ZEROED_NOUNS = [102, 504, 364, 321, 556]          # run/train.py:73
DP_ENC_CUS = 0                                    # default encoder-stream CU budget under data parallelism: 0 = no reservation (see TrainStep)
WEIGHT_DICT = {"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5, "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}


# HIP stream priorities of (the vision-tower stream, the text-tower stream); 0 = default, -1 = high (A/B knob: bench.py --stream-priorities)
STREAM_PRIORITIES = [0, 0]


def finishes_last(name: str) -> bool:
    """Decoder parameters whose gradient is written by the memory side's backward (model/tfm_decoder.py _MemorySide), the last node
    of the step's backward: FlatArena lays them out last so that every earlier all-reduce bucket can overlap it."""
    return (name == "proj.weight" or name.startswith(("transformer.pre_norm.", "pos_embed", "temporal_embed"))
            or ".multihead_attn.in_proj_" in name)


def build_criterion():
    """run/train.py:459-473."""
    return box_utils.SetCriterion(22047, matcher=box_utils.build_matcher(None), weight_dict=dict(WEIGHT_DICT), eos_coef=0.1,
                                  losses=["boxes", "cardinality"])


class TrainStep:
    def __init__(self, cfg, backbone, decoder, lr=3e-5, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8,
                 bucket_bytes=16 << 20, fast_heads=True, enc_cus=None, force_comm=False):
        self.cfg, self.backbone, self.decoder = cfg, backbone, decoder
        self.criterion = build_criterion().to(next(decoder.parameters()).device)
        self.nce, self.word = EgoNCE(), WordContrastiveLoss()
        # losses() applies obj_proj and txt_proj exactly once per step: their weight gradients may go straight into the gradient arena
        # (qside.LinearX3.single_use; the _GradSink precondition)
        from .model.qside import LinearX3
        for seq in (decoder.obj_proj, decoder.txt_proj):
            for m in seq:
                if isinstance(m, LinearX3):
                    m.single_use = True
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        decoder.materialize_logits = not fast_heads
        self.arena = FlatArena(decoder, bucket_bytes, late=finishes_last)
        self.m = torch.zeros_like(self.arena.params)
        self.v = torch.zeros_like(self.arena.params)
        self.force_comm = bool(force_comm)
        self.comm = BucketedAllReduce(self.arena, force=force_comm)
        self.iteration = 0
        self._text_stream = None
        self.enc_stream = None
        # CU budget of the persistent GEMMs on the pipelined encoder stream (0 = all CUs); see prefetch().  A reservation for the
        # RCCL kernels of the comm stream (e.g. 248 = one CU per XCD left free) is available through `enc_cus`, but it is NOT the
        # default: measured with the real RCCL collectives in flight (bench.py --force-comm, 1 rank) 248 CUs cost 2.1 % and 240 CUs
        # 2.9 % of the step (2048 / 6144 / 8192 tiles no longer divide into whole rounds of workgroups), while without a
        # reservation the collectives showed no cost: a bucket's all-reduce (<= 16 MB) is scheduled in the same gaps between
        # persistent GEMMs as the decoder's own kernels and has the whole backward to finish.
        self.enc_cus = (DP_ENC_CUS if self.comm.enabled else 0) if enc_cus is None else int(enc_cus)
        self._pending = None
        self._zeroed_idx = None
        self._row_base = None                          # cached arange(rows) * context_length of the caption matrix
        self.text_on_side_stream = True               # False: both towers on one stream (bench.py's kernel-alone timing steps)
        self.span_log = None                           # a list: every step appends (start, end) device events of its decoder / loss / optimizer span on the main stream (bench.py)
        backbone.eval()                               # run/train.py:89

    # ------------------------------------------------------------------ forward
    def encode(self, video, text, text_max_len=None):
        """Frozen towers (run/train.py:108-116): returns video_grid bf16 [B,T,n,D] and text feature map fp32.

        The text tower (12 k token rows) cannot fill the chip: it runs on a side HIP stream concurrently with the vision
        tower and is joined before returning.

        text_max_len (opt-in, a HOST int from the data pipeline -- batch["text_max_len"]: tokens of the longest caption incl. SOT and EOT):
        the tower runs on the first ceil16(text_max_len) positions only.  The attention mask is causal (LaviLa.py:636-642) and only the EOT
        row of each caption is read (run/train.py:124), so the positions behind the longest caption's EOT are dead work: EgoClip narrations
        are ~10 tokens of the 77.  The hint must be an UPPER BOUND: a caption longer than it loses its EOT row and `losses` turns its
        feature row (hence the loss) into NaN rather than reading another caption's row.  The feature map then has that many positions; the EOT rows are the same up to the GEMM kernels' summation
        order (tests/test_step_gpu.py).  Without the hint all 77 positions are computed, as the reference does (no host sync to find out)."""
        if text_max_len is not None:
            keep = min(text.shape[1], (int(text_max_len) + 15) // 16 * 16)
            if keep < text.shape[1]:
                text = text[:, :keep].contiguous()
        B, T = video.shape[:2]
        n = self.cfg.patches_per_frame
        cur = torch.cuda.current_stream()
        if self._text_stream is None:
            self._text_stream = torch.cuda.Stream(priority=STREAM_PRIORITIES[1])
        side = self._text_stream if self.text_on_side_stream else cur
        side.wait_stream(cur)
        with torch.no_grad():
            with torch.cuda.stream(side):
                _, tmap = self.backbone.encode_text(text, apply_project=False, want_cls=False)
            _, pat = self.backbone.visual.forward_features(video, out_dtype=torch.bfloat16, split_cls=True)      # patch rows, contiguous
        cur.wait_stream(side)
        tmap.record_stream(cur)
        return pat.view(B, T, n, pat.shape[-1]), tmap

    def prefetch(self, batch):
        """Start the frozen towers of `batch` on the encoder stream (software pipelining across steps: the frozen encoder
        of step i+1 does not depend on step i's optimizer update, so it overlaps step i's decoder forward/backward, whose
        13-row query-side kernels leave most CUs idle).  The result is picked up by the next step(batch)."""
        if self.enc_stream is None:
            self.enc_stream = torch.cuda.Stream(priority=STREAM_PRIORITIES[0])
            if self.enc_cus > 0:
                ops.set_stream_cu_budget(self.enc_stream, self.enc_cus)
        main = torch.cuda.current_stream()
        self.enc_stream.wait_stream(main)
        with torch.cuda.stream(self.enc_stream):
            grid, tmap = self.encode(batch["video"], batch["text"], batch.get("text_max_len"))
            ev = torch.cuda.Event()
            ev.record(self.enc_stream)
        self._pending = (batch, grid, tmap, ev)

    def _encoded(self, batch):
        if self._pending is not None and self._pending[0] is batch:
            _, grid, tmap, ev = self._pending
            self._pending = None
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            grid.record_stream(main)
            tmap.record_stream(main)
            return grid, tmap
        self._pending = None
        return self.encode(batch["video"], batch["text"], batch.get("text_max_len"))

    def losses(self, batch, next_batch=None):
        cfg = self.cfg
        video, text = batch["video"], batch["text"]
        B, T = video.shape[:2]
        W, _ = world()
        grid, tmap = self._encoded(batch)
        if self.span_log is not None:                  # (after the main stream's wait for this batch's towers: the span holds no tower time)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.span_log.append([ev, None])
        if next_batch is not None:
            self.prefetch(next_batch)
        det, hs, _, _ = self.decoder(grid)
        # EOT position and "caption present" flag of every caption row in one launch (run/train.py:124,144; hh_text_flags)
        eot, pad_flag = ops.text_flags(text.contiguous())
        key = (text.shape[0], tmap.shape[1], text.device)                          # (tmap holds fewer positions under text_max_len)
        if self._row_base is None or self._row_base[0] != key:
            self._row_base = (key, torch.arange(text.shape[0], device=text.device) * tmap.shape[1])
        if tmap.shape[1] < text.shape[1]:
            # text_max_len hint in use: it MUST be an upper bound of every caption's length.  A caption whose EOT lies behind the
            # trimmed positions has no EOT row in tmap (row_base + eot would read another caption's row, or run past the end): its
            # feature row is poisoned with NaN instead -- the loss then exposes the bad hint, without a host sync
            bad = eot >= tmap.shape[1]
            rows = tmap.reshape(-1, tmap.shape[-1]).index_select(0, self._row_base[1] + eot.clamp(max=tmap.shape[1] - 1))
            rows = torch.where(bad[:, None], torch.full((), float("nan"), dtype=rows.dtype, device=rows.device), rows)
        else:
            rows = tmap.reshape(-1, tmap.shape[-1]).index_select(0, self._row_base[1] + eot)
        # captions' EOT rows and the noun vocabulary through txt_proj in ONE call (run/train.py:126,186 apply it twice: the same Linear, so
        # the same numbers; one weight-gradient GEMM whose result goes straight into the gradient arena)
        text_embeds, noun_embeds = tfm_decoder._SplitRows.apply(self.decoder.txt_proj(torch.cat([rows, batch["all_nouns"].to(rows.dtype)])), rows.shape[0])
        hs_last = getattr(hs, "_hh_last", None)                                    # QueryStack's second output (== hs[-1], a tensor of its own)
        obj = self.decoder.obj_proj(hs[-1] if hs_last is None else hs_last)        # [B,Q,256]
        obj_rest, video_embeds = tfm_decoder._SplitLastQuery.apply(obj)             # obj[:, :-1], obj[:, -1]
        if self._zeroed_idx is None or self._zeroed_idx.device != text.device:
            self._zeroed_idx = torch.tensor(ZEROED_NOUNS, device=text.device)              # once: a Python-list index is an H2D copy + sync per step
        noun_vec = batch["noun_vec"].clone().index_fill_(1, self._zeroed_idx, 0)
        # matching first (it needs only pred_boxes): the matched-target counts of both box types and the valid-word count are the
        # normalisers of three loss terms (box_utils.py:218-222 all-reduces `num_boxes` per box type; the word loss is a mean over
        # valid words) -- under data parallelism they ride in the packed contrastive all-gather instead of three blocking scalar
        # all-reduces in the middle of the forward
        hand = batch["boxes"][:, :, :2].flatten(0, 1)
        objb = batch["boxes"][:, :, 2:].flatten(0, 1)
        nq = cfg.num_queries if cfg.num_queries != 0 else 10
        matcher = self.criterion.matcher
        cnt2 = torch.empty((2, hand.shape[0]), dtype=torch.int32, device=hand.device)      # both box types' per-frame target counts
        mh = matcher.match_raw(det["pred_boxes"], 0, 2, hand, count_out=cnt2[0])
        mo = matcher.match_raw(det["pred_boxes"], 2, nq - 2, objb, count_out=cnt2[1])
        counts = torch.empty(3, dtype=torch.float32, device=hand.device)          # (three launches instead of sum / ne / sum / cat / cast)
        torch.sum(cnt2, dim=1, dtype=torch.float32, out=counts[:2])
        torch.sum(batch["nouns"] != 0, dim=tuple(range(batch["nouns"].dim())), dtype=torch.float32, out=counts[2])
        ve, te, pf, vv, nv, sums = gather_contrastive(video_embeds, text_embeds, pad_flag, batch["verb_vec"], noun_vec,
                                                      force=self.force_comm, counts=counts)
        norm = normaliser(sums)                                                     # clamp(global / W, 1) x 3
        Bg = ve.shape[0]
        sim = sim_matrix(te, ve)                                                    # [5Bg, Bg]
        sim_v, sim_n = sim_matrix(vv, vv), sim_matrix(nv, nv)
        nce = self.nce.forward_rows(sim, sim_v, sim_n, pf)                           # multi_pad_mask = pf[:, None].repeat(1, Bg)
        R = te.shape[0] // Bg
        with torch.no_grad():
            acc_vt, acc_tv = compute_tv_accuracy(sim.view(Bg, R, Bg)[:, 0], te, sim_v, sim_n, Bg)
        if det.get("pred_logits") is None and "boxes" in self.criterion.losses:
            # both box types, every scalar of box_utils.py:142-173,445-461 and both cardinality metrics: one fused node (3 launches)
            lh, lo, mh, mo, dh = box_utils.step_box_losses(self.criterion, det, hand, objb, nq, norm, mh, mo)
            do = dh
        else:                                           # materialised class logits (fast_heads=False): the per-type module functions
            lh, mh, dh = box_utils.compute_box_loss("hand_boxes", self.criterion, det, hand, None, None, n_queries=nq, num_boxes=norm[0],
                                                    match=mh, return_loss_dict=True)
            lo, mo, do = box_utils.compute_box_loss("obj_boxes", self.criterion, det, objb, None, None, n_queries=nq, num_boxes=norm[1],
                                                    match=mo, return_loss_dict=True)
        word = self.word(noun_embeds, obj_rest, batch["nouns"], count=sums[2] / W if W > 1 else None)
        total = torch.add(nce + lh + lo, word, alpha=0.5)                           # nce + lh + lo + 0.5 * word, run/train.py:149,183,191
        return {"total_loss": total, "nce_loss": nce.detach(), "box_loss_hand": lh.detach(), "box_loss_obj": lo.detach(),
                "word_loss": word.detach(), "acc_vt": acc_vt, "acc_tv": acc_tv, "match_hand": mh, "match_obj": mo,
                "pred_boxes": det["pred_boxes"], "hs": hs,
                "cardinality_error_hand_boxes": dh.get("cardinality_error_hand_boxes"),
                "cardinality_error_obj_boxes": do.get("cardinality_error_obj_boxes"),
                "pred_logits": det.get("pred_logits"), "pred_logits_argmax": det.get("pred_logits_argmax")}

    # ------------------------------------------------------------------ step
    def step(self, batch, next_batch=None):
        """One optimisation step on `batch`; if `next_batch` is given its frozen-tower forward is launched concurrently."""
        self.decoder.train()
        self.backbone.eval()
        self.arena.zero_grad(force=False)                       # the previous update already cleared the gradient arena
        out = self.losses(batch, next_batch)
        out["total_loss"].backward()
        self.optimizer_step()
        out["total_loss"] = out["total_loss"].detach()
        return out

    def optimizer_step(self, zero_grads=True):
        """Gradient all-reduce completion + AdamW (run/train.py:199-203) over the whole arena in one call.  Which parameters are
        updated (torch.optim.AdamW skips parameters without a gradient entirely, per-parameter step counts included) is decided on
        the device from the flags BucketedAllReduce.finish() leaves in arena.seg_flag -- global under data parallelism -- so there is
        no host-side launch plan and no synchronisation.  zero_grads: clear the gradient arena in the same pass."""
        self.comm.finish()
        self.iteration += 1
        a = self.arena
        ops.adamw_arena_step(a.params, a.grads, self.m, self.v, a.seg_off, a.seg_decay, a.seg_step, a.seg_flag, a.seg_coef,
                             self.lr, *self.betas, self.eps, self.wd, zero_grads=zero_grads)
        a.grads_clean = bool(zero_grads)
        a.sink_armed = False                                    # until the next zero_grad()
        if self.span_log and self.span_log[-1][1] is None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.span_log[-1][1] = ev


    # ------------------------------------------------------------------ optimizer state (checkpoint exchange with the reference)
    def _reference_param_order(self):
        """Parameter numbering of the reference's optimizer: torch.optim.AdamW over optim_policy's two groups
        (utils/train_utils.py:28-48; run/train.py:519-520): group 0 = no-decay names, group 1 = the rest, each in
        named_parameters() order, requires_grad only.  Returns (names of group 0, names of group 1)."""
        named = [n for n, p in self.decoder.named_parameters() if p.requires_grad]
        return [n for n in named if no_decay(n)], [n for n in named if not no_decay(n)]

    def state_dict(self):
        """`optimizer.state_dict()` in torch.optim.AdamW's own format, so the 'optimizer' entry of a runtime checkpoint
        (run/train.py:232-237) can be exchanged with the reference in both directions: per-parameter {'step', 'exp_avg',
        'exp_avg_sq'} for every parameter that has been updated (class_embed / vid_proj never are), two param_groups.  The extra
        top-level key 'hh' (ignored by torch's loader) carries the attention-dropout seed stream and the step counter."""
        g0, g1 = self._reference_param_order()
        state = {}
        steps = self.arena.steps                                  # one device read
        for idx, n in enumerate(g0 + g1):
            if n in self.arena.offsets and steps[n] > 0:
                o, k = self.arena.offsets[n]
                state[idx] = {"step": torch.tensor(float(steps[n])),
                              "exp_avg": self.m[o:o + k].detach().clone(), "exp_avg_sq": self.v[o:o + k].detach().clone()}
        shapes = {n: p.shape for n, p in self.decoder.named_parameters()}
        for idx, n in enumerate(g0 + g1):
            if idx in state:
                state[idx]["exp_avg"] = state[idx]["exp_avg"].view(shapes[n])
                state[idx]["exp_avg_sq"] = state[idx]["exp_avg_sq"].view(shapes[n])
        common = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False, "foreach": None,
                  "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True}
        groups = [dict(common, weight_decay=0.0, params=list(range(len(g0)))),
                  dict(common, weight_decay=self.wd, params=list(range(len(g0), len(g0) + len(g1))))]
        return {"state": state, "param_groups": groups,
                "hh": {"iteration": self.iteration, "xattn_seed": self.decoder.transformer._seed}}

    def load_state_dict(self, sd):
        """Inverse of state_dict(); also accepts the reference's own `optimizer.state_dict()` (no 'hh' entry: the dropout seed
        stream restarts, the step counter is taken from the per-parameter 'step')."""
        g0, g1 = self._reference_param_order()
        names = g0 + g1
        groups = sd["param_groups"]
        if len(groups) != 2 or len(groups[0]["params"]) != len(g0) or len(groups[1]["params"]) != len(g1):
            raise ValueError("TrainStep.load_state_dict: param_groups do not match optim_policy's (no-decay, decay) split "
                             "(%s vs %d/%d parameters)" % ([len(g["params"]) for g in groups], len(g0), len(g1)))
        ids = list(groups[0]["params"]) + list(groups[1]["params"])
        self.lr, self.betas, self.eps = groups[1]["lr"], tuple(groups[1]["betas"]), groups[1]["eps"]
        self.wd = groups[1]["weight_decay"]
        self.m.zero_()
        self.v.zero_()
        steps = {}
        for pid, n in zip(ids, names):
            st = sd["state"].get(pid)
            if st is None:
                continue
            if n not in self.arena.offsets:
                raise ValueError(f"TrainStep.load_state_dict: optimizer state for '{n}', which never receives a gradient here")
            o, k = self.arena.offsets[n]
            self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps[n] = int(float(st["step"]))
        self.arena.set_steps(steps)
        extra = sd.get("hh") or {}
        self.iteration = int(extra.get("iteration", max(steps.values(), default=0)))
        # the checkpoint is written by one rank: re-mix the rank into the saved dropout-seed stream so that, as from a fresh start
        # (Cross_Attention.next_dropout_seed), ranks keep drawing different attention-dropout masks after a resume; rank 0 (and a
        # single process) continues the saved stream exactly
        seed = extra.get("xattn_seed", None)
        _, rank = world()
        self.decoder.transformer._seed = None if seed is None else (int(seed) ^ (40503 * rank)) & 0x7FFFFFFF


def batch_slice(batch, k, captions_per_clip):
    """The first k clips of a batch dict (synth.make_batch contract: `text` holds captions_per_clip rows per clip, `all_nouns` is shared)."""
    out = {}
    for name, v in batch.items():
        if not torch.is_tensor(v):
            out[name] = v                              # host-side hints (text_max_len)
        else:
            out[name] = v if name == "all_nouns" else (v[:k * captions_per_clip] if name == "text" else v[:k])
    return out


@torch.no_grad()
def first_clips_check(ts, batch, k=2):
    """Does clip i of a large batch get the result it gets in a batch of k clips?  Eval-mode forward (no dropout) of the frozen towers +
    decoder + hand-box matching on `batch` and on its first k clips; returns what differs.  Per clip every kernel of the path computes
    the same function at any batch size; the SUMMATION ORDER differs in two places only -- (1) the <= 64 token rows behind the last
    full 256-row GEMM tile (rows of the batch's LAST clip: the in-kernel row tail sums 8 K-slices, the tiles sum k-tiles in order), so
    the last clip of the small batch may differ from its large-batch result by bf16 roundings; every earlier clip must be
    BIT-IDENTICAL through the encoder; (2) the cross-attention's key slices (`ops.xattn_fwd` cuts the keys into more slices when
    B * heads cannot fill the chip), fp32 re-association ~1e-6 of `hs`.  bench.py prints this as `selfcheck`; tests/test_step_gpu.py
    asserts it at the benchmarked batch."""
    if not 1 <= k <= batch["video"].shape[0]:
        raise ValueError("first_clips_check: k=%d clips of a batch of %d" % (k, batch["video"].shape[0]))
    dec, cfg = ts.decoder, ts.cfg
    was_training = dec.training
    dec.eval()
    try:
        def fwd(b):
            grid, _ = ts.encode(b["video"], b["text"])
            det, hs, _, _ = dec(grid)
            hand = b["boxes"][:, :, :2].flatten(0, 1)
            mh = ts.criterion.matcher.match_raw(det["pred_boxes"], 0, 2, hand)
            return grid, hs, det["pred_boxes"], mh
        nq = cfg.num_queries if cfg.num_queries != 0 else 10
        T = batch["video"].shape[1]
        gb, hb, pb, mb = fwd(batch)
        small = batch_slice(batch, k, cfg.captions_per_clip)
        gs, hs_, ps, ms = fwd(small)
        # the k clips' own hand-box loss (box_utils.py:445-461 on the k clips alone), from the large batch's boxes and from the small batch's
        hand = small["boxes"][:, :, :2].flatten(0, 1)
        def hand_loss(pred):
            det = {"pred_boxes": pred.contiguous(), "pred_logits": None, "aux_outputs": []}
            return box_utils.compute_box_loss("hand_boxes", ts.criterion, det, hand, None, None, n_queries=nq)[0]
        lb, ls = hand_loss(pb[:k * T]), hand_loss(ps)
        f = lambda t: t.float()
        scale = lambda t: float(f(t).abs().max().clamp_min(1e-30))
        idx_equal = all(bool(torch.equal(mb[key][:k * T], ms[key])) for key in ("pred_idx", "tgt_idx", "n", "count"))
        same = torch.ones(k * T, dtype=torch.bool, device=pb.device)
        for key in ("pred_idx", "tgt_idx"):
            same &= (mb[key][:k * T].reshape(k * T, -1) == ms[key].reshape(k * T, -1)).all(1)
        rec = {"clips_compared": k, "batch": int(batch["video"].shape[0]),
               "encoder_bit_identical_clips_before_last": bool(k < 2 or torch.equal(gb[:k - 1], gs[:k - 1])),
               "encoder_last_clip_max_abs_diff_over_scale": float((f(gb[k - 1]) - f(gs[k - 1])).abs().max()) / scale(gs[k - 1]),
               "hs_max_abs_diff_over_scale": float((f(hb[:, :k]) - f(hs_)).abs().max()) / scale(hs_),
               "pred_boxes_max_abs_diff": float((pb[:k * T] - ps).abs().max()),
               "matched_indices_equal": idx_equal,
               # (matching is bit-exact on IDENTICAL boxes; these two runs' boxes differ by pred_boxes_max_abs_diff, which may flip a frame whose
               # two best assignments are tied to within that -- the loss below, each side with its own matching, then still agrees)
               "matched_frames_equal_fraction": float(same.float().mean()),
               "hand_box_loss_first_clips_rel_diff": abs(float(lb) - float(ls)) / max(abs(float(ls)), 1e-30)}
        return rec
    finally:
        dec.train(was_training)


_MCQ_STREAMS = {}


def _mcq_side_stream(device):
    key = (device.type, device.index)
    if key not in _MCQ_STREAMS:
        _MCQ_STREAMS[key] = torch.cuda.Stream(device=device)
    return _MCQ_STREAMS[key]


@torch.no_grad()
def mcq_forward(backbone, decoder, video, text, cfg):
    """Batched EgoMCQ scoring (run/test_EgoMCQ.py:56-83): video [q,5,T,3,H,W], text [q,77] -> scores [q,5]."""
    q = video.shape[0]
    T, n = video.shape[2], cfg.patches_per_frame
    was = decoder.materialize_logits
    decoder.materialize_logits = False
    try:
        # the text tower (77 x q tokens) cannot fill the chip: side stream beside the vision tower, as in TrainStep.encode
        cur = torch.cuda.current_stream()
        side = _mcq_side_stream(video.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _, tmap = backbone.encode_text(text, apply_project=False)
        _, fmap = backbone.visual.forward_features(video.flatten(0, 1), out_dtype=torch.bfloat16, split_cls=True)      # patch rows [q*5, T*n, D]
        cur.wait_stream(side)
        tmap.record_stream(cur)
        _, hs, _, _ = decoder(fmap.view(q * 5, T, n, fmap.shape[-1]))
        te = decoder.txt_proj(tmap[torch.arange(q, device=text.device), text.float().argmax(-1)])
        ve = decoder.obj_proj(hs[-1])[:, -1].view(q, 5, -1)
        return sim_matrix(te[:, None], ve)[:, 0]
    finally:
        decoder.materialize_logits = was


class McqScorer:
    """EgoMCQ scoring with the same software pipelining as TrainStep: the frozen towers of the NEXT item batch run on an encoder
    stream while the decoder forward + scoring of the current one runs on the main stream (the decoder's 13-row query side is
    latency-bound and leaves the chip idle; run/test_EgoMCQ.py:56-83 runs the two back to back).  Same results as mcq_forward.

        scorer = McqScorer(backbone, decoder, cfg)
        for item, nxt in zip(items, items[1:] + [None]):
            scores = scorer(item["video"], item["text"], next_item=None if nxt is None else (nxt["video"], nxt["text"]))
    """

    def __init__(self, backbone, decoder, cfg):
        self.backbone, self.decoder, self.cfg = backbone, decoder, cfg
        self.enc_stream = None
        self._pending = None

    @torch.no_grad()
    def _encode(self, video, text):
        cur = torch.cuda.current_stream()
        side = _mcq_side_stream(video.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _, tmap = self.backbone.encode_text(text, apply_project=False)
        _, fmap = self.backbone.visual.forward_features(video.flatten(0, 1), out_dtype=torch.bfloat16, split_cls=True)      # patch rows [q*5, T*n, D]
        cur.wait_stream(side)
        tmap.record_stream(cur)
        return fmap, tmap

    def prefetch(self, video, text):
        if self.enc_stream is None:
            self.enc_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        self.enc_stream.wait_stream(main)
        with torch.cuda.stream(self.enc_stream):
            fmap, tmap = self._encode(video, text)
            ev = torch.cuda.Event()
            ev.record(self.enc_stream)
        self._pending = (video, text, fmap, tmap, ev)

    @torch.no_grad()
    def __call__(self, video, text, next_item=None):
        """video [q,5,T,3,H,W], text [q,77] -> scores [q,5]; next_item = (video, text) of the following call, if known."""
        cfg, decoder = self.cfg, self.decoder
        q, T, n = video.shape[0], video.shape[2], cfg.patches_per_frame
        pend, self._pending = self._pending, None
        if pend is not None and pend[0] is video and pend[1] is text:
            fmap, tmap, ev = pend[2], pend[3], pend[4]
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            fmap.record_stream(main)
            tmap.record_stream(main)
        else:
            fmap, tmap = self._encode(video, text)
        if next_item is not None:
            self.prefetch(*next_item)
        was = decoder.materialize_logits
        decoder.materialize_logits = False
        try:
            _, hs, _, _ = decoder(fmap.view(q * 5, T, n, fmap.shape[-1]))
            te = decoder.txt_proj(tmap[torch.arange(q, device=text.device), text.float().argmax(-1)])
            ve = decoder.obj_proj(hs[-1])[:, -1].view(q, 5, -1)
            return sim_matrix(te[:, None], ve)[:, 0]
        finally:
            decoder.materialize_logits = was
