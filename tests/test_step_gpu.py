"""GPU parity of the decoder (fwd+bwd), the losses and the whole training step against the CPU oracle and the
golden fixtures emitted from the imported reference.

Stated tolerances (north_star: "decoder loss within 1e-3 rel of CPU reference, bit-exact Hungarian indices"):
  * losses: |gpu - ref| <= 1e-3 * |ref| for EVERY term (total, hand / object box loss, EgoNCE, word loss) -- the north-star bound;
  * activations (hs, pred_boxes): scale-relative 2e-2 / absolute 5e-3 (bf16 operands, fp32 accumulation);
  * gradients: (1) with the oracle fed the GPU path's OWN bf16 K/V (query side) and the GPU's own dK/dV (memory side), so that
    both sides differentiate the same function at the same point: relative L2 per tensor <= 5e-3 (heads <= 1e-2)
    (test_decoder_gradients_on_shared_kv); (2) end to end against the fp32 oracle / the reference golden, where the bf16 rounding
    of K/V flips single ReLU units of the FFN (B*Q ~ 10 rows feed a weight row): per tensor <= 9.5e-2, median <= 4.4e-2 (2x measured), and the
    fixture's gradient samples <= 1e-1 relative L2.  This loose bound is the price of storing K/V in bf16 (half the bytes of the 1.6 GB
    buffer the cross-attention streams six times per step); measured on MI355X (profiles/r3_parity_measured.json): worst tensor
    `temporal_embed` 5.7e-2 (T = 4) / `frame_index.weight` 5.5e-2 (T = 16), median 2.4e-2 -- (1) is the real evidence;
  * class head (R7 / R13): pred_logits values 2e-5 of the logit scale on the GPU's own hs, <= 6.5e-3 end to end (measured 3.2e-3);
    pred_logits_argmax of the benchmarked fast path == the reference's wherever its top-2 margin exceeds twice the logit bound;
    cardinality_error_* equal to the reference golden / the oracle on shared hs;
  * matching indices: bit-exact vs the oracle run on the SAME fp32 pred_boxes (index stability across precisions
    is not defined for untrained queries -- SURVEY.md section 8a R12 -- and is reported as a statistic only).
Parity is defined in eval mode (dropout p=0).
"""
import os

import numpy as np
import pytest
import torch

from helping_hand_for_egocentric_videos_amd import synth, TINY4, TINY16, ops
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder, box_utils
from helping_hand_for_egocentric_videos_amd.model.loss import EgoNCE, WordContrastiveLoss
from helping_hand_for_egocentric_videos_amd.model.metric import sim_matrix
from helping_hand_for_egocentric_videos_amd.step import McqScorer, TrainStep, mcq_forward
from oracle import decoder as OD, losses as OL, step as OS
from _record import record

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def scaled_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def to_dev(batch):
    return {k: v.cuda() for k, v in batch.items()}


@pytest.mark.parametrize("cfg", [TINY4, TINY16], ids=["T4", "T16"])
def test_decoder_forward_backward_vs_oracle(cfg):
    dsd = synth.decoder_state(cfg, seed=3)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    B = 2
    feats = torch.randn(B, cfg.num_frames, cfg.patches_per_frame, cfg.embed_dim, generator=torch.Generator().manual_seed(5))
    feats = feats.to(torch.bfloat16).float()                      # both sides see bf16-representable features
    out, hs, _, _ = dec(feats.cuda())
    params = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    ro, rhs = OD.objdecoder_forward(feats, params, cfg)
    assert hs.shape == rhs.shape and out["pred_boxes"].shape == ro["pred_boxes"].shape
    assert scaled_err(hs, rhs) < 2e-2
    assert float((out["pred_boxes"].cpu() - ro["pred_boxes"]).abs().max()) < 5e-3
    assert out["pred_logits"].shape == ro["pred_logits"].shape
    assert len(out["aux_outputs"]) == cfg.dec_layers - 1
    # R7 class head (tfm_decoder.py:208,216): VALUES, two ways.  (1) the head alone on the GPU's own hs -- hh_qgemm_f32x3 is
    # fp32-grade, so <= 2e-5 of the logit scale; (2) end to end vs the fp32 oracle: the error of hs (bf16 K/V) carried through.
    with torch.no_grad():
        W, b = dsd["class_embed.weight"], dsd["class_embed.bias"]
        own = torch.nn.functional.linear(hs.detach().cpu(), W, b)                                  # [L,B,Q,K]
        own = own[:, :, None].expand(-1, -1, cfg.num_frames, -1, -1).flatten(1, 2)
    e_head = scaled_err(out["pred_logits"], own[-1])
    e_e2e = scaled_err(out["pred_logits"], ro["pred_logits"])
    e_aux = max(scaled_err(out["aux_outputs"][l]["pred_logits"], own[l]) for l in range(cfg.dec_layers - 1))
    print("class head: on shared hs %.2e (aux layers %.2e), end to end %.2e of the logit scale" % (e_head, e_aux, e_e2e))
    assert e_head <= 2e-5 and e_aux <= 2e-5, (e_head, e_aux)
    assert e_e2e < 6.5e-3, e_e2e                                      # measured 3.2e-3 (T4) / 2.6e-3 (T16)
    assert torch.equal(out["pred_logits"].argmax(-1).cpu(), own[-1].argmax(-1)) or \
        float((out["pred_logits"].argmax(-1).cpu() == own[-1].argmax(-1)).float().mean()) > 0.99
    assert scaled_err(out["aux_outputs"][0]["pred_boxes"], ro["aux_outputs"][0]["pred_boxes"]) < 2e-2
    g = torch.Generator().manual_seed(1)
    w, wb = torch.randn(rhs.shape, generator=g), torch.randn(ro["pred_boxes"].shape, generator=g)
    ((hs * w.cuda()).sum() + (out["pred_boxes"] * wb.cuda()).sum()).backward()
    ((rhs * w).sum() + (ro["pred_boxes"] * wb).sum()).backward()
    rel = {}
    for name, p in dec.named_parameters():
        rg = params[name].grad
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, name
            continue
        assert p.grad is not None, name
        gg = p.grad.detach().cpu()
        if "multihead_attn.in_proj_bias" in name:      # key-bias gradient is exactly 0 in maths: compare q and v parts
            C = cfg.dec_dim
            sel = torch.cat([torch.arange(0, C), torch.arange(2 * C, 3 * C)])
            gg, rg = gg[sel], rg[sel]
        rel[name] = float((gg - rg).norm() / (rg.norm() + 1e-12))
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    print("decoder grad rel-L2: median %.2e, worst %s" % (float(np.median(list(rel.values()))), worst))
    record("decoder_fwd_bwd_T%d" % cfg.num_frames, "end-to-end gradient rel-L2 vs fp32 oracle: worst tensor (%s)" % worst[0][0], worst[0][1], 9.5e-2)
    record("decoder_fwd_bwd_T%d" % cfg.num_frames, "end-to-end gradient rel-L2 vs fp32 oracle: median", float(np.median(list(rel.values()))), 4.4e-2)
    assert len(rel) > 100
    # ReLU-gated FFN weights can flip single hidden units under bf16 noise (few query rows) -> bound on rel-L2, not max.  Round 6: both
    # bounds at 2x the measurement (4.8e-2 / 4.7e-2 worst tensor, 2.2e-2 / 1.9e-2 median at T = 4 / 16) instead of 3x
    assert max(rel.values()) < 9.5e-2, worst
    assert float(np.median(list(rel.values()))) < 4.4e-2


def test_losses_vs_oracle_on_identical_inputs():
    cfg = TINY4
    g = torch.Generator().manual_seed(3)
    for B in (2, 6):
        batch = synth.make_batch(cfg, B, seed=B)
        te, ve = torch.randn(5 * B, 256, generator=g), torch.randn(B, 256, generator=g)
        batch["noun_vec"][0] = batch["noun_vec"][1]
        batch["verb_vec"][0] = batch["verb_vec"][1]
        batch["noun_vec"][0:2, 5] = 1
        batch["verb_vec"][0:2, 3] = 1
        sv, sn = OL.sim_matrix(batch["verb_vec"], batch["verb_vec"]), OL.sim_matrix(batch["noun_vec"], batch["noun_vec"])
        pad = ((batch["text"] != 0).sum(-1) != 2).float()[:, None].repeat(1, B)
        ter, ver = te.clone().requires_grad_(True), ve.clone().requires_grad_(True)
        rl, rmask = OL.egonce(OL.sim_matrix(ter, ver), sv, sn, pad)
        rl.backward()
        tg, vg = te.cuda().requires_grad_(True), ve.cuda().requires_grad_(True)
        gl, gmask = EgoNCE()(sim_matrix(tg, vg), sv.cuda(), sn.cuda(), multi_pad_mask=pad.cuda(), strict_mask=True)
        gl.backward()
        torch.testing.assert_close(gl.cpu(), rl.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(gmask.cpu(), rmask)
        torch.testing.assert_close(tg.grad.cpu(), ter.grad, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(vg.grad.cpu(), ver.grad, rtol=1e-4, atol=1e-6)
        ne, pe = torch.randn(cfg.n_nouns, 256, generator=g), torch.randn(B, cfg.num_queries, 256, generator=g)
        per = pe.clone().requires_grad_(True)
        rw, rassign = OL.word_contrastive(ne, per, batch["nouns"], return_assign=True)
        rw.backward()
        pg = pe.cuda().requires_grad_(True)
        gw, cols = WordContrastiveLoss()(ne.cuda(), pg, batch["nouns"].cuda(), return_assignment=True)
        gw.backward()
        torch.testing.assert_close(gw.cpu(), rw.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pg.grad.cpu(), per.grad, rtol=1e-4, atol=1e-6)
        for b in range(B):                                        # bit-exact word assignment
            valid = batch["nouns"][b] != 0
            assert cols[b].cpu()[valid].tolist() == rassign[b].tolist()
    # box loss through the module API: compute_box_loss on raw boxes == oracle
    crit = box_utils.SetCriterion(22047, box_utils.build_matcher(None), {"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5,
                                  "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}, 0.1, ["boxes", "cardinality"])
    batch = synth.make_batch(TINY16, 3, seed=7)
    pred = torch.rand(3 * 16, 13, 4, generator=g) * 0.4 + 0.2
    for bt, sl in (("hand_boxes", slice(0, 2)), ("obj_boxes", slice(2, 4))):
        raw = batch["boxes"][:, :, sl].flatten(0, 1)
        pr = pred.clone().requires_grad_(True)
        rloss, ridx, _ = OL.compute_box_loss(bt, pr, raw, 12)
        rloss.backward()
        pgpu = pred.cuda().requires_grad_(True)
        gloss, match = box_utils.compute_box_loss(bt, crit, {"pred_boxes": pgpu, "pred_logits": None}, raw.cuda(), None, None, n_queries=12)
        gloss.backward()
        torch.testing.assert_close(gloss.cpu(), rloss.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pgpu.grad.cpu(), pr.grad, rtol=1e-3, atol=1e-6)
        for (a, b), (c, d) in zip(match, ridx):
            assert torch.equal(a, c) and torch.equal(b, d)
    # list-of-dicts reference API
    raw = batch["boxes"][:, :, :2].flatten(0, 1)
    tl = box_utils.prepare_targets(raw.cuda(), None, None, center_crop=False)
    ol = OL.prepare_targets(raw)
    for a, b in zip(tl, ol):
        assert torch.equal(a["boxes"].cpu(), b)
    idx = box_utils.build_matcher(None)({"pred_boxes": pred[:, :2].cuda(), "pred_logits": torch.zeros(48, 2, 3).cuda()}, tl, exclude_class=True)
    for (a, b), (c, d) in zip(idx, OL.hungarian_match(pred[:, :2], ol)):
        assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.parametrize("cfg,name", [(TINY4, "tiny4"), (TINY16, "tiny16")])
def test_full_step_vs_reference_golden(cfg, name):
    g = np.load(os.path.join(GOLD, f"step_{name}.npz"))
    B = int(g["meta_B"])
    esd = synth.encoder_state(cfg, seed=int(g["meta_seed_w"]))
    dsd = synth.decoder_state(cfg, seed=int(g["meta_seed_w"]))
    batch = synth.make_batch(cfg, B, seed=int(g["meta_seed_b"]))
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    dec.eval()
    ts.arena.zero_grad()
    res = ts.losses(to_dev(batch))
    tol = {"total_loss": 1e-3, "box_loss_hand": 1e-3, "box_loss_obj": 1e-3, "nce_loss": 1e-3, "word_loss": 1e-3}
    report = {}
    for k, t in tol.items():
        ref = float(g["loss_" + k])
        got = float(res[k])
        report[k] = abs(got - ref) / abs(ref)
        record("golden_step_" + name, k + " rel error vs reference golden", report[k], t)
        assert report[k] <= t, (k, got, ref, report[k])
    print("rel loss errors vs reference golden:", {k: f"{v:.2e}" for k, v in report.items()})
    # R17: compute_tv_accuracy of the product (model/metric.py:378-392) vs the reference's values on the same step
    assert abs(float(res["acc_vt"]) - float(g["loss_acc_vt"])) < 1e-6 and abs(float(res["acc_tv"]) - float(g["loss_acc_tv"])) < 1e-6, \
        (float(res["acc_vt"]), float(g["loss_acc_vt"]), float(res["acc_tv"]), float(g["loss_acc_tv"]))
    np.testing.assert_allclose(res["hs"].detach().cpu().numpy(), g["hs"], rtol=0, atol=2e-2 * np.abs(g["hs"]).max())
    assert np.abs(res["pred_boxes"].detach().cpu().numpy() - g["pred_boxes"]).max() < 5e-3
    # bit-exact matching on identical fp32 boxes: oracle LSAP on the GPU's own pred_boxes
    pb = res["pred_boxes"].detach().cpu()
    nq = cfg.num_queries
    for key, sl, qs in (("match_hand", slice(0, 2), slice(0, 2)), ("match_obj", slice(2, 4), slice(2, nq))):
        raw = batch["boxes"][:, :, sl].flatten(0, 1)
        ref_idx = OL.hungarian_match(pb[:, qs], OL.prepare_targets(raw))
        for (a, b), (c, d) in zip(res[key], ref_idx):
            assert torch.equal(a, c) and torch.equal(b, d)
    # R7 / R13 on the BENCHMARKED path (fast_heads=True: pred_logits_argmax instead of [6,B*T,Q,22048] logits): the argmax equals the
    # reference's wherever the reference's own top-2 margin exceeds twice the end-to-end logit error bound (LOGIT_TOL, checked on the
    # full-logits path below), and the cardinality metric (box_utils.py:142-154) equals the reference's value
    assert res["pred_logits"] is None and res["pred_logits_argmax"] is not None
    am = res["pred_logits_argmax"].cpu().numpy()
    assert am.shape == g["logits_argmax"].shape
    LOGIT_TOL = 4.5e-3 * float(g["logits_absmax"])      # measured 2.2e-3 (T4) / 1.7e-3 (T16) of the logit scale
    safe = g["logits_top2_margin"] > 2 * LOGIT_TOL
    assert safe.sum() >= 0.2 * safe.size, safe.mean()
    assert np.array_equal(am[safe], g["logits_argmax"][safe])
    print("fast-path class argmax: %d/%d entries above the margin agree, %.3f agreement over all" %
          (safe.sum(), safe.size, float((am == g["logits_argmax"]).mean())))
    for bt in ("hand_boxes", "obj_boxes"):
        assert abs(float(res["cardinality_error_" + bt]) - float(g["cardinality_error_" + bt])) < 1e-6, bt
    # the module-default path (materialised logits) on the same weights: logit VALUES vs the reference's sample, same cardinality
    dec_full = tfm_decoder.build_decoder(cfg, dsd)
    ts_full = TrainStep(cfg, backbone, dec_full, fast_heads=False)
    dec_full.eval()
    with torch.no_grad():
        rf = ts_full.losses(to_dev(batch))
    lg = rf["pred_logits"].float().cpu().numpy()
    err = np.abs(lg[:, :, ::173] - g["logits_sample"]).max()
    err_last = np.abs(lg[:, :, -1] - g["logits_last_class"]).max()
    print("pred_logits vs reference golden: max abs %.2e (no-object column %.2e) at logit scale %.2f" % (err, err_last, float(g["logits_absmax"])))
    assert err <= LOGIT_TOL and err_last <= LOGIT_TOL
    assert np.array_equal(lg.argmax(-1), am)                               # both paths run the same head kernel on the same hs
    for bt in ("hand_boxes", "obj_boxes"):
        assert abs(float(rf["cardinality_error_" + bt]) - float(g["cardinality_error_" + bt])) < 1e-6, bt
    for k_ in tol:
        assert abs(float(rf[k_]) - float(res[k_])) <= 1e-6 * abs(float(res[k_])), k_
    # statistic only: agreement with the fp32 reference's own indices
    rows = np.concatenate([a.numpy() for a, _ in res["match_obj"]])
    agree = float((rows == g["idx_obj_rows"]).mean()) if len(rows) == len(g["idx_obj_rows"]) else float("nan")
    print("end-to-end object-index agreement with the fp32 reference:", agree)
    record("golden_step_" + name, "statistic: end-to-end object-index agreement with the fp32 reference (1 = all frames)", agree, 1.0)
    res["total_loss"].backward()
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    bad = []
    for n_, p in dec.named_parameters():
        if n_ in norms:
            rel = abs(float(p.grad.norm()) - norms[n_]) / (norms[n_] + 1e-12)
            if rel > 5e-2 and "in_proj_bias" not in n_:
                bad.append((n_, rel))
        else:
            assert n_.startswith(("class_embed", "vid_proj")), n_
    assert not bad, bad[:5]
    # the fixture's gradient SAMPLES (64 evenly spaced elements per tensor, make_golden.sample): catches sign / slice errors that
    # a norm comparison cannot
    params = dict(dec.named_parameters())
    rels = {}
    for key in g.files:
        if not key.startswith("grad_sample__"):
            continue
        n_ = key[len("grad_sample__"):]
        f = params[n_].grad.detach().flatten().cpu()
        idx = torch.linspace(0, f.numel() - 1, min(64, f.numel())).long()
        got, want = f[idx].numpy(), g[key]
        if "multihead_attn.in_proj_weight" in n_ and np.abs(want).max() == 0:
            continue
        rels[n_] = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-12))
    assert len(rels) >= 15
    print("grad samples vs reference golden: worst", sorted(rels.items(), key=lambda kv: -kv[1])[:3])
    assert max(rels.values()) < 1e-1, sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    # EgoMCQ forward
    mcq = synth.make_mcq_item(cfg, 2, seed=int(g["meta_seed_b"]))
    scores = mcq_forward(backbone, dec, mcq["video"].cuda(), mcq["text"].cuda(), cfg)
    assert np.abs(scores.cpu().numpy() - g["mcq_scores"]).max() < 5e-3
    assert np.array_equal(scores.cpu().numpy().argmax(-1), g["mcq_scores"].argmax(-1))


def test_step_issues_no_host_synchronisation():
    """The step docstring's claim: after warm-up, a pipelined training step enqueues its work without a single synchronising call
    (torch's sync debug mode raises on .item(), pageable H2D copies, boolean-mask indexing, ...)."""
    cfg = TINY16
    backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device="cuda")
    decoder = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device="cuda")
    batch = {k: v.cuda() for k, v in synth.make_batch(cfg, 4, seed=1).items()}
    ts = TrainStep(cfg, backbone, decoder)
    for _ in range(2):
        ts.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = ts.step(batch, next_batch=batch)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.isfinite(out["total_loss"]).item()


def test_train_step_updates_like_oracle_adamw_and_learns():
    cfg = TINY4
    esd, dsd = synth.encoder_state(cfg, seed=4), synth.decoder_state(cfg, seed=4)
    batch = synth.make_batch(cfg, 2, seed=9)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec, lr=1e-4)
    # arena bookkeeping: params are views of the flat buffer; one AdamW segment per parameter with optim_policy's decay flag; the
    # memory side's parameters (final only at the end of backward) sit behind everything else
    from helping_hand_for_egocentric_videos_amd.step import finishes_last
    decay = dict(zip(ts.arena.names, ts.arena.seg_decay.tolist()))
    first_late = min(o for n_, (o, k) in ts.arena.offsets.items() if finishes_last(n_))
    for n_, p in dec.named_parameters():
        if n_ in ts.arena.offsets:
            o, k = ts.arena.offsets[n_]
            assert p.data_ptr() == ts.arena.params.data_ptr() + 4 * o
            assert bool(decay[n_]) == (not OS.no_decay(n_))
            assert (o >= first_late) == finishes_last(n_)
    dec.eval()                                   # deterministic (no dropout) for the comparison
    before = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    # one step by hand in eval mode
    ts.arena.zero_grad()
    out = ts.losses(to_dev(batch))
    out["total_loss"].backward()
    grads = {n_: p.grad.detach().cpu().clone() for n_, p in dec.named_parameters() if n_ in ts.arena.offsets}
    ref_params = {k: v.cpu().clone() for k, v in before.items()}
    OS.adamw_update(ref_params, grads, None, lr=1e-4, wd=1e-5)
    ts.optimizer_step(zero_grads=False)          # hh_adamw_arena_step: lr 1e-4, wd 1e-5 on the decay segments, first step
    for n_, p in dec.named_parameters():
        torch.testing.assert_close(p.detach().cpu(), ref_params[n_], rtol=1e-5, atol=1e-7)
    assert torch.equal(dec.class_embed.weight.detach().cpu(), before["class_embed.weight"].cpu())
    # train mode (dropout on): loss goes down over a few steps on a fixed batch
    l0 = float(ts.step(to_dev(batch))["total_loss"])
    for _ in range(8):
        l1 = float(ts.step(to_dev(batch))["total_loss"])
    assert np.isfinite(l1) and l1 < l0, (l0, l1)


def test_cross_attention_dropout_statistics():
    """Train-mode attention dropout (tfm_decoder.py:365, p=0.1) cannot match torch's RNG stream; validate it
    statistically: E[out] ~= out(p=0) and the backward uses the same mask (finite-difference check on dq)."""
    B, Q, M, heads = 2, 13, 2048, 8
    C = heads * 64
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, Q, C, generator=g) * 0.05).cuda()
    kv = torch.randn(B, M, 2 * C, generator=g).to(torch.bfloat16).cuda()
    k, v = kv[:, :, :C], kv[:, :, C:]
    base, _ = ops.xattn_fwd(q, k, v, heads)
    acc = torch.zeros_like(base)
    n = 64
    for s in range(n):
        o, _ = ops.xattn_fwd(q, k, v, heads, 0.1, 1000 + s)
        acc += o
    assert float(((acc / n) - base).abs().max()) < 0.05 * float(base.abs().max()) + 0.02
    o1, lse = ops.xattn_fwd(q, k, v, heads, 0.1, 5)
    o2, _ = ops.xattn_fwd(q, k, v, heads, 0.1, 5)
    assert torch.equal(o1, o2)                                 # same seed -> same mask
    dout = torch.randn(B, Q, C, generator=g).cuda()
    dkv = torch.zeros_like(kv)
    dq = ops.xattn_bwd(q, k, v, o1, lse, dout, dkv[:, :, :C], dkv[:, :, C:], heads, 0.1, 5)
    eps = 1e-2
    d = torch.randn(B, Q, C, generator=g).cuda()
    op, _ = ops.xattn_fwd(q + eps * d, k, v, heads, 0.1, 5)
    om, _ = ops.xattn_fwd(q - eps * d, k, v, heads, 0.1, 5)
    fd = float(((op - om) / (2 * eps) * dout).sum())
    an = float((dq * d).sum())
    assert abs(fd - an) <= 0.05 * abs(fd) + 1e-3, (fd, an)


@pytest.mark.parametrize("wseed,bseed", [(0, 1000), (1, 1001), (2, 1002)])
def test_headline_config_c2_losses_vs_oracle(wseed, bseed):
    """BASELINE config 2 itself (full-width TimeSformer-L, T=16, 224p, nq=12), B=2: every loss term of the GPU step vs the
    CPU oracle on identical synthetic weights/inputs; bit-exact matching on the GPU's own fp32 boxes.  Three (weight, batch) seed
    pairs (the thinnest margin -- word_loss at 0.62 of the 1e-3 bound -- had been measured on one seed only)."""
    from helping_hand_for_egocentric_videos_amd import C2
    cfg = C2
    esd, dsd = synth.encoder_state(cfg, seed=wseed), synth.decoder_state(cfg, seed=wseed)
    batch = synth.make_batch(cfg, 2, seed=bseed)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    dec.eval()
    ts.arena.zero_grad()
    res = ts.losses(to_dev(batch))
    with torch.no_grad():
        ref = OS.step_losses(esd, dsd, batch, cfg)
    rep = {}
    for k, tol in (("total_loss", 1e-3), ("box_loss_hand", 1e-3), ("box_loss_obj", 1e-3), ("nce_loss", 1e-3), ("word_loss", 1e-3)):
        got, want = float(res[k]), float(ref[k])
        rep[k] = abs(got - want) / abs(want)
        record("c2_headline_step_seed%d" % wseed, k + " rel error vs oracle", rep[k], tol)
        assert rep[k] <= tol, (k, got, want)
    print("C2 rel loss errors vs oracle:", {k: f"{v:.2e}" for k, v in rep.items()})
    pb = res["pred_boxes"].detach().cpu()
    for key, sl, qs in (("match_hand", slice(0, 2), slice(0, 2)), ("match_obj", slice(2, 4), slice(2, cfg.num_queries))):
        raw = batch["boxes"][:, :, sl].flatten(0, 1)
        for (a, b), (c, d) in zip(res[key], OL.hungarian_match(pb[:, qs], OL.prepare_targets(raw))):
            assert torch.equal(a, c) and torch.equal(b, d)
    agree = [torch.equal(a, c) for (a, _), (c, _) in zip(res["match_obj"], ref["idx_obj"])]
    print("C2 end-to-end object-index agreement with the fp32 oracle: %.3f" % (sum(agree) / len(agree)))
    record("c2_headline_step_seed%d" % wseed, "statistic: end-to-end object-index agreement with the fp32 oracle (fraction of frames)", sum(agree) / len(agree), 1.0)


def test_caption_length_hint_trims_the_text_tower_without_changing_the_step():
    """batch["text_max_len"] (opt-in, a host int from the data pipeline): the text tower runs on the first ceil16(max caption length)
    positions instead of all 77 -- causal mask + only the EOT row is read (LaviLa.py:636-642,660-670; run/train.py:124), so the rest is
    dead work.  Full-width config 2, B = 2, eval mode: EOT features equal up to the GEMM kernels' summation order, every loss term within
    1e-4 of the untrimmed step, same matching."""
    from helping_hand_for_egocentric_videos_amd import C2
    cfg = C2
    esd, dsd = synth.encoder_state(cfg, seed=0), synth.decoder_state(cfg, seed=0)
    batch = to_dev(synth.make_batch(cfg, 2, seed=1000))
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    dec.eval()
    text = batch["text"]
    longest = int((text != 0).sum(1).max())                      # (test only: the step itself never asks the device)
    assert longest <= 14
    with torch.no_grad():
        _, full = ts.encode(batch["video"], text)
        _, trim = ts.encode(batch["video"], text, text_max_len=longest)
    assert full.shape[1] == cfg.context_length and trim.shape[1] == 16
    eot = text.float().argmax(1)
    rows = torch.arange(text.shape[0], device=text.device)
    a, b = full[rows, eot], trim[rows, eot]
    assert float((a - b).norm() / a.norm()) <= 5e-3
    ts.arena.zero_grad()
    ref = ts.losses(batch)
    ts.arena.zero_grad()
    got = ts.losses(dict(batch, text_max_len=longest))
    for k in ("total_loss", "nce_loss", "box_loss_hand", "box_loss_obj", "word_loss"):
        assert abs(float(got[k]) - float(ref[k])) <= 1e-4 * abs(float(ref[k])), (k, float(got[k]), float(ref[k]))
    for key in ("match_hand", "match_obj"):
        for (p, q), (r, s) in zip(got[key], ref[key]):
            assert torch.equal(p, r) and torch.equal(q, s)


@pytest.mark.parametrize("wseed,bseed", [(0, 1004), (1, 1005), (2, 1006)])
def test_c4_full_width_step_losses_vs_oracle(wseed, bseed):
    """BASELINE config 4 ITSELF (full-width TimeSformer-L, T = 32 frames, 336 px: N = 18 433 tokens and M = 18 432 memory tokens
    per clip, nq = 12), B = 2: every loss term of the GPU step vs the CPU oracle on identical synthetic weights / inputs (north-star
    bound 1e-3), bit-exact matching of all 64 frames on the GPU's own fp32 boxes."""
    from helping_hand_for_egocentric_videos_amd import C4
    from _record import check
    cfg = C4
    esd, dsd = synth.encoder_state(cfg, seed=wseed), synth.decoder_state(cfg, seed=wseed)
    batch = synth.make_batch(cfg, 2, seed=bseed)
    tag = "c4_full_width_step_seed%d" % wseed
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    dec.eval()
    ts.arena.zero_grad()
    res = ts.losses(to_dev(batch))
    with torch.no_grad():
        ref = OS.step_losses(esd, dsd, batch, cfg)
    for k in ("total_loss", "box_loss_hand", "box_loss_obj", "nce_loss", "word_loss"):
        got, want = float(res[k]), float(ref[k])
        check(tag, k + " rel error vs oracle", abs(got - want) / abs(want), 1e-3)
    assert res["pred_boxes"].shape == (64, 13, 4)
    check(tag, "hs scaled max error vs oracle", scaled_err(res["hs"], ref["hs"]), 2e-2)
    check(tag, "pred_boxes max abs error vs oracle", float((res["pred_boxes"].detach().cpu() - ref["pred_boxes"]).abs().max()), 5e-3)
    pb = res["pred_boxes"].detach().cpu()
    for key, sl, qs in (("match_hand", slice(0, 2), slice(0, 2)), ("match_obj", slice(2, 4), slice(2, cfg.num_queries))):
        raw = batch["boxes"][:, :, sl].flatten(0, 1)
        idx = OL.hungarian_match(pb[:, qs], OL.prepare_targets(raw))
        assert len(idx) == 64
        for (a, b), (c, d) in zip(res[key], idx):
            assert torch.equal(a, c) and torch.equal(b, d)
    agree = [torch.equal(a, c) for (a, _), (c, _) in zip(res["match_obj"], ref["idx_obj"])]
    print("C4 end-to-end object-index agreement with the fp32 oracle: %.3f" % (sum(agree) / len(agree)))
    record(tag, "statistic: end-to-end object-index agreement with the fp32 oracle (fraction of frames)", sum(agree) / len(agree), 1.0)
    res["total_loss"].backward()
    for name, p in dec.named_parameters():
        if name in ts.arena.offsets:
            assert torch.isfinite(p.grad).all(), name


QUERY_SIDE_GRAD_TOL = 5e-3     # legacy K/V path: per-tensor relative L2 of the query-side gradients on shared K/V (measured: parameters ~1e-5, dK/dV 1.7e-3)
SHARED_ROWS_GRAD_TOL = 1e-4    # memory-space path: every query-side tensor on shared memory rows and a shared ReLU branch (measured: <= 2.0e-5)


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("cfg", [TINY4, TINY16], ids=["T4", "T16"])
def test_decoder_gradients_on_shared_memory_rows(cfg):
    """Gradient parity with both sides differentiating the same function at the same point, in three stages (default path: the
    cross-attention runs in memory space, csrc/mattn.hip -- no K/V projection of the memory tokens).

    Heads: the oracle's frame-conditioned box MLP runs on the GPU's OWN hs -> head gradients to 1e-2, d(hs) to 1e-3.
    Query side: the oracle's six 13-row layers -- INCLUDING every layer's key / value projection (tfm_decoder.py:438-441, evaluated by the
      oracle in fp32 on all M tokens) -- run on the GPU path's OWN bf16 memory rows (memory, memory + pos) and are back-propagated from
      the GPU's own d(hs) -> every layer / query-embedding gradient, the key and value rows of the in-projections too (<= 1e-4 per
      tensor, median <= 5e-5, measured 2e-5 / 1.2e-5: the x3 GEMMs and the hi/lo-split attention are fp32-grade), and d memory / d pos
      (through bf16 Pd^T / dS^T: <= 5e-3, measured 2.2e-3).
    Memory side: the oracle's proj -> pre_norm chain is back-propagated from the GPU's own d memory / d pos -> proj / pre_norm /
      positional-embedding gradients (<= 3e-3, measured 1.3e-3; bf16 GEMM operands)."""
    C, L = cfg.dec_dim, cfg.dec_layers
    dsd = synth.decoder_state(cfg, seed=3)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    dec.transformer.debug_keep_kv = True
    assert dec.transformer.kv_free
    B, T, n = 2, cfg.num_frames, cfg.patches_per_frame
    feats = torch.randn(B, T, n, cfg.embed_dim, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
    out, hs, _, _ = dec(feats.cuda())
    hs.retain_grad()
    g = torch.Generator().manual_seed(1)
    w, wb = torch.randn(hs.shape, generator=g), torch.randn(out["pred_boxes"].shape, generator=g)
    ((hs * w.cuda()).sum() + (out["pred_boxes"] * wb.cuda()).sum()).backward()
    gp = dict(dec.named_parameters())
    kept = dec.transformer.last_holder.kept
    M = T * n
    head_names = [k for k in dsd if k.startswith(("frame_proj.", "frame_index.", "bbox_embed."))]
    # ---- heads on the GPU's own hs
    ph = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    hs_leaf = hs.detach().cpu().clone().requires_grad_(True)
    boxes = OD.box_head(hs_leaf, ph, cfg, T)[-1]
    torch.testing.assert_close(boxes, out["pred_boxes"].detach().cpu(), rtol=1e-4, atol=1e-6)
    ((hs_leaf * w).sum() + (boxes * wb).sum()).backward()
    relh = {k: _rel(gp[k].grad, ph[k].grad) for k in head_names}
    relh["d(hs)"] = _rel(hs.grad, hs_leaf.grad)
    print("heads on shared hs: worst", sorted(relh.items(), key=lambda kv_: -kv_[1])[:3])
    # both sides are fp32-grade here; what remains are the handful of box-MLP units whose pre-activation sits within ~1e-5 of the
    # ReLU kink (a flip fraction f costs ~sqrt(2 f) in relative L2: 17 of 1.3 M entries -> 5e-3)
    assert max(relh.values()) < 1e-2 and relh["d(hs)"] < 1e-3, relh
    # ---- query side (with the key / value projections) on the GPU's own memory rows, back-propagated from the GPU's own d(hs)
    mem = kept["mem"].float().cpu().requires_grad_(True)                  # [B,M,C]
    mp = kept["mp"].float().cpu().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    # (the oracle's FFN takes the ReLU branch the GPU took: of ~3e5 hidden units a few sit within 1e-5 of the kink, and one flipped
    # unit would cost every gradient below it ~2e-3 -- measured -- although both sides are fp32-grade)
    Qn = hs.shape[2]
    masks = [m.cpu().view(B, Qn, -1) for m in kept["relu_masks"]]
    own_masks = []
    with torch.no_grad():
        _, free = OD.objdecoder_forward(feats, dsd, cfg, compute_logits=False, rows=(mem.detach(), mp.detach()), relu_trace=own_masks)
    _, rhs = OD.objdecoder_forward(feats, params, cfg, compute_logits=False, rows=(mem, mp), relu_masks=masks)
    assert _rel(rhs, free) < 1e-5                                          # the imposed branch is the oracle's own, up to units at the kink
    # ... COUNTED: the units on which the imposed masks differ from the branch the oracle takes by itself -- a handful of ~3e5
    flips = sum(int((a.bool() != b_.bool()).sum()) for a, b_ in zip(masks, own_masks))
    units = sum(a.numel() for a in masks)
    record("decoder_shared_rows_T%d" % T, "statistic: FFN units whose ReLU branch was imposed against the oracle's own (of %d)" % units, flips, 16)
    assert len(own_masks) == len(masks) == L and flips <= 16, (flips, units)
    check_hs = _rel(hs, rhs)
    record("decoder_shared_rows_T%d" % T, "hs rel-L2 vs oracle on the same memory rows", check_hs, 1e-4)
    assert check_hs < 1e-4                                                # same rows -> hs agrees to fp32-grade kernels
    rhs.backward(hs.grad.detach().cpu())
    memory_side = ("proj.weight", "transformer.pre_norm.", "pos_embed", "temporal_embed")
    rel = {}
    for name, p in dec.named_parameters():
        rg = params[name].grad
        if rg is None or name.startswith(memory_side) or name in head_names:
            continue
        gg = p.grad.detach().cpu()
        if "multihead_attn.in_proj_bias" in name:                       # the key BIAS gradient is 0 in exact arithmetic (softmax shift invariance)
            assert float(gg[C:2 * C].abs().max()) == 0.0
            sel = torch.cat([torch.arange(0, C), torch.arange(2 * C, 3 * C)])
            gg, rg = gg[sel], rg[sel]
        rel[name] = _rel(gg, rg)
    worst = sorted(rel.items(), key=lambda kv_: -kv_[1])[:4]
    med = float(np.median(list(rel.values())))
    print("query side on shared memory rows: %d tensors, median %.2e, worst %s" % (len(rel), med, worst))
    record("decoder_shared_rows_T%d" % T, "query-side gradient rel-L2 on shared rows: worst tensor (%s)" % worst[0][0], worst[0][1], SHARED_ROWS_GRAD_TOL)
    record("decoder_shared_rows_T%d" % T, "query-side gradient rel-L2 on shared rows: median", med, 5e-5)
    assert len(rel) > 100 and max(rel.values()) < SHARED_ROWS_GRAD_TOL and med < 5e-5, worst
    e_dmem = _rel(kept["dmem"].view(B, M, C), mem.grad + mp.grad)
    e_dpos = _rel(kept["dpos"], mp.grad.sum(0))
    record("decoder_shared_rows_T%d" % T, "d memory (value + key path) rel-L2", e_dmem, 5e-3)
    record("decoder_shared_rows_T%d" % T, "d pos rel-L2", e_dpos, 5e-3)
    assert e_dmem < 5e-3 and e_dpos < 5e-3, (e_dmem, e_dpos)
    # ---- memory side from the GPU's own d memory / d pos
    params2 = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    memory2, pos2 = OD.memory_rows(feats, params2, cfg)
    assert _rel(kept["mem"].float(), memory2) < 1e-2                       # the GPU's bf16 rows are the oracle's, rounded
    torch.autograd.backward([memory2, pos2], [kept["dmem"].view(B, M, C).float().cpu(), kept["dpos"].float().cpu()])
    rel2 = {}
    for name in ("proj.weight", "transformer.pre_norm.weight", "transformer.pre_norm.bias", "pos_embed", "temporal_embed"):
        gg, rg = gp[name].grad.detach().cpu(), params2[name].grad
        if name == "pos_embed":                                          # row 0 (CLS slot) is unused by construct_3d_pos_embed
            gg, rg = gg[:, 1:], rg[:, 1:]
        rel2[name] = _rel(gg, rg)
    worst2 = sorted(rel2.items(), key=lambda kv_: -kv_[1])[:4]
    print("memory side from shared d memory: median %.2e, worst %s" % (float(np.median(list(rel2.values()))), worst2))
    record("decoder_shared_rows_T%d" % T, "memory-side gradient rel-L2: worst tensor (%s)" % worst2[0][0], worst2[0][1], 3e-3)
    assert max(rel2.values()) < 3e-3, worst2


def test_decoder_gradients_on_shared_kv_legacy_path():
    """The round-1..4 path (Cross_Attention.kv_free = False: one batched K/V in-projection of the memory tokens for all layers +
    hh_xattn_*), kept for A/B measurements and for token counts that are not multiples of 128: same three stages with the oracle
    fed the GPU path's own bf16 K/V and dK/dV (TINY4)."""
    cfg = TINY4
    C, L = cfg.dec_dim, cfg.dec_layers
    dsd = synth.decoder_state(cfg, seed=3)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    dec.transformer.debug_keep_kv = True
    dec.transformer.kv_free = False
    B, T, n = 2, cfg.num_frames, cfg.patches_per_frame
    feats = torch.randn(B, T, n, cfg.embed_dim, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
    out, hs, _, _ = dec(feats.cuda())
    hs.retain_grad()
    g = torch.Generator().manual_seed(1)
    w, wb = torch.randn(hs.shape, generator=g), torch.randn(out["pred_boxes"].shape, generator=g)
    ((hs * w.cuda()).sum() + (out["pred_boxes"] * wb.cuda()).sum()).backward()
    gp = dict(dec.named_parameters())
    kv, dkv = dec.transformer.last_holder.kept
    M = T * n
    head_names = [k for k in dsd if k.startswith(("frame_proj.", "frame_index.", "bbox_embed."))]
    K = kv[:, :L * C].float().cpu().view(B, M, L, C).permute(2, 0, 1, 3).contiguous().requires_grad_(True)       # [L,B,M,C]
    V = kv[:, L * C:].float().cpu().view(B, M, L, C).permute(2, 0, 1, 3).contiguous().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    _, rhs = OD.objdecoder_forward(feats, params, cfg, compute_logits=False, kv=(K, V))
    assert _rel(hs, rhs) < 3e-3                                         # same K/V -> hs agrees to the attention kernel's own rounding
    rhs.backward(hs.grad.detach().cpu())
    memory_side = ("proj.weight", "transformer.pre_norm.", "pos_embed", "temporal_embed")
    rel = {}
    for name, p in dec.named_parameters():
        rg = params[name].grad
        if rg is None or name.startswith(memory_side) or name in head_names:
            continue
        gg = p.grad.detach().cpu()
        if "multihead_attn.in_proj" in name:                            # query rows only; key/value rows belong to the memory side
            gg, rg = gg[:C], rg[:C]
        rel[name] = _rel(gg, rg)
    dK = dkv[:, :L * C].float().cpu().view(B, M, L, C).permute(2, 0, 1, 3)
    dV = dkv[:, L * C:].float().cpu().view(B, M, L, C).permute(2, 0, 1, 3)
    for l in range(L):
        rel[f"dK[{l}]"], rel[f"dV[{l}]"] = _rel(dK[l], K.grad[l]), _rel(dV[l], V.grad[l])
    worst = sorted(rel.items(), key=lambda kv_: -kv_[1])[:4]
    assert len(rel) > 100 and max(rel.values()) < QUERY_SIDE_GRAD_TOL and float(np.median(list(rel.values()))) < 1e-3, worst
    params2 = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    K2, V2 = OD.memory_kv(feats, params2, cfg)
    torch.autograd.backward([K2, V2], [dK.contiguous(), dV.contiguous()])
    rel2 = {}
    for name in ("proj.weight", "transformer.pre_norm.weight", "transformer.pre_norm.bias", "pos_embed", "temporal_embed"):
        gg, rg = gp[name].grad.detach().cpu(), params2[name].grad
        if name == "pos_embed":
            gg, rg = gg[:, 1:], rg[:, 1:]
        rel2[name] = _rel(gg, rg)
    for l in range(L):
        name = f"transformer.decoder.layers.{l}.multihead_attn.in_proj_weight"
        gg, rg = gp[name].grad.detach().cpu(), params2[name].grad
        rel2[name + "[v]"], rel2[name + "[k]"] = _rel(gg[2 * C:], rg[2 * C:]), _rel(gg[C:2 * C], rg[C:2 * C])
    assert max(rel2.values()) < 5e-3, sorted(rel2.items(), key=lambda kv_: -kv_[1])[:4]


def test_long_clip_decoder_and_step_c4_shapes():
    """BASELINE config 4's decoder / step half at its real sequence shape (T = 32 frames, 336 px -> n = 576, M = 18 432 memory
    tokens, nq = 12) on the reduced-width tower: T-dependent positional embedding and frame conditioning
    (tfm_decoder.py:161-166,212-215), the M-dependent key-slice split of the cross-attention backward, matcher / box head with 32
    frames.  Losses within 1e-3 of the oracle, bit-exact indices on the GPU's own boxes, decoder gradients finite and close."""
    cfg = TINY4.with_(num_frames=32, img_size=336, num_queries=12)
    assert cfg.patches_per_frame == 576 and cfg.num_frames * cfg.patches_per_frame == 18432
    esd, dsd = synth.encoder_state(cfg, seed=2), synth.decoder_state(cfg, seed=2)
    batch = synth.make_batch(cfg, 2, seed=33)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    dec.eval()
    ts.arena.zero_grad()
    res = ts.losses(to_dev(batch))
    params = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    ref = OS.step_losses(esd, params, batch, cfg)
    rep = {}
    for k in ("total_loss", "box_loss_hand", "box_loss_obj", "nce_loss", "word_loss"):
        got, want = float(res[k]), float(ref[k])
        rep[k] = abs(got - want) / abs(want)
        assert rep[k] <= 1e-3, (k, got, want)
    print("C4-shape rel loss errors vs oracle:", {k: f"{v:.2e}" for k, v in rep.items()})
    assert res["pred_boxes"].shape == (2 * 32, 13, 4) and res["hs"].shape == (6, 2, 13, 512)
    assert scaled_err(res["hs"], ref["hs"]) < 2e-2
    pb = res["pred_boxes"].detach().cpu()
    for key, sl, qs in (("match_hand", slice(0, 2), slice(0, 2)), ("match_obj", slice(2, 4), slice(2, cfg.num_queries))):
        raw = batch["boxes"][:, :, sl].flatten(0, 1)
        idx = OL.hungarian_match(pb[:, qs], OL.prepare_targets(raw))
        assert len(idx) == 64
        for (a, b), (c, d) in zip(res[key], idx):
            assert torch.equal(a, c) and torch.equal(b, d)
    res["total_loss"].backward()
    ref["total_loss"].backward()
    rel = {}
    for name, p in dec.named_parameters():
        if params[name].grad is None:
            continue
        assert torch.isfinite(p.grad).all(), name
        gg, rg = p.grad.detach().cpu(), params[name].grad
        if "multihead_attn.in_proj_bias" in name:
            sel = torch.cat([torch.arange(0, cfg.dec_dim), torch.arange(2 * cfg.dec_dim, 3 * cfg.dec_dim)])
            gg, rg = gg[sel], rg[sel]
        rel[name] = _rel(gg, rg)
    worst = sorted(rel.items(), key=lambda kv_: -kv_[1])[:4]
    print("C4-shape decoder grads vs fp32 oracle: median %.2e, worst %s" % (float(np.median(list(rel.values()))), worst))
    assert max(rel.values()) < 1.5e-1 and float(np.median(list(rel.values()))) < 5e-2, worst
    # frame conditioning really is T-dependent: all 32 frame_index rows receive gradient
    assert int((dec.frame_index.weight.grad.abs().sum(1) > 0).sum()) == 32


def test_metrics_vs_oracle_r17():
    """compute_tv_accuracy / egomcq_accuracy_metrics (model/metric.py:378-392,209-225) on the GPU vs the oracle's restatement on
    identical inputs, including verb/noun-sharing and duplicate-caption positives."""
    from helping_hand_for_egocentric_videos_amd.model.metric import compute_tv_accuracy, egomcq_accuracy_metrics
    g = torch.Generator().manual_seed(7)
    for Bn in (4, 9):
        te = torch.randn(5 * Bn, 256, generator=g)
        te[5] = te[0]                                                     # clips 0 and 1 share their first caption
        ve = torch.randn(Bn, 256, generator=g)
        vv, nv = (torch.rand(Bn, 11, generator=g) < 0.3).float(), (torch.rand(Bn, 13, generator=g) < 0.3).float()
        vv[2], nv[2] = vv[3], nv[3]
        sv, sn = OL.sim_matrix(vv, vv), OL.sim_matrix(nv, nv)
        sim = OL.sim_matrix(te, ve).view(Bn, 5, Bn)[:, 0]
        want = OL.compute_tv_accuracy(sim, te, sv, sn, Bn)
        got = compute_tv_accuracy(sim.cuda(), te.cuda(), sv.cuda(), sn.cuda(), Bn)
        assert abs(float(got[0]) - float(want[0])) < 1e-6 and abs(float(got[1]) - float(want[1])) < 1e-6
    preds = torch.randn(40, 5, generator=g)
    labels = torch.randint(0, 5, (40,), generator=g)
    types = torch.randint(1, 3, (40,), generator=g)
    got = egomcq_accuracy_metrics(preds.cuda(), labels.cuda(), types.cuda())
    want = OL.egomcq_accuracy(preds, labels, types)
    assert set(got) == set(want) == {"Intra-video", "Inter-video"}
    for k in want:
        assert abs(got[k] - want[k]) < 1e-4, (k, got[k], want[k])


def test_cross_attention_reference_signature_r8():
    """Cross_Attention.forward(src, mask, query_embed, pos_embed) in the reference's own layout (tfm_decoder.py:76-93), which runs
    TransformerDecoder.forward / TransformerDecoderLayer.forward_pre with their reference signatures (every layer projects its own
    K/V): forward vs the oracle, agreement with the batched ObjDecoder route, and gradients through the per-layer path."""
    cfg = TINY4
    dsd = synth.decoder_state(cfg, seed=3)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    tfm = dec.transformer
    B, C, T, n = 2, cfg.dec_dim, cfg.num_frames, cfg.patches_per_frame
    g = torch.Generator().manual_seed(2)
    src = (torch.randn(B, C, T, n, generator=g)).to(torch.bfloat16).float()
    pos = torch.randn(1, C, T, n, generator=g) * 0.1
    mask = torch.zeros(B, T, n, dtype=torch.bool)
    srcg = src.cuda().requires_grad_(True)
    hs, memory, a, sa = tfm(srcg, mask.cuda(), dec.query_embed.weight, pos.cuda())
    assert a == [] and sa == [] and hs.shape == (cfg.dec_layers, B, cfg.dec_queries, C) and memory.shape == (B, C, T, n)
    sd = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    srcr = src.clone().requires_grad_(True)
    rhs, rmem = OD.cross_attention_forward(srcr, mask, sd["query_embed.weight"], pos, sd, cfg)
    assert scaled_err(hs, rhs) < 2e-2
    assert scaled_err(memory, rmem) < 1e-4
    w = torch.randn(rhs.shape, generator=g)
    (hs * w.cuda()).sum().backward()
    (rhs * w).sum().backward()
    assert _rel(srcg.grad, srcr.grad) < 1e-1                                # end to end through six layers (ReLU-kink noise, see header)
    rel = {}
    for name, p in dec.named_parameters():
        if name.startswith("transformer.") and sd[name].grad is not None:
            gg, rg = p.grad.detach().cpu(), sd[name].grad
            if "multihead_attn.in_proj_bias" in name:
                sel = torch.cat([torch.arange(0, C), torch.arange(2 * C, 3 * C)])
                gg, rg = gg[sel], rg[sel]
            rel[name] = _rel(gg, rg)
    worst = sorted(rel.items(), key=lambda kv_: -kv_[1])[:3]
    print("reference-signature path grads: median %.2e, worst %s" % (float(np.median(list(rel.values()))), worst))
    assert len(rel) > 100 and max(rel.values()) < 1.5e-1 and float(np.median(list(rel.values()))) < 5e-2, worst
    # layer-level signature: one layer called directly, sequence-first, returns the (tgt, attn, self_attn) triple
    layer = tfm.decoder.layers[0]
    tgt = torch.zeros(cfg.dec_queries, B, C, device="cuda")
    qpos = dec.query_embed.weight.detach()[:, None].expand(-1, B, -1)
    mem = memory.detach().flatten(2).permute(2, 0, 1).contiguous()
    out, a1, sa1 = layer(tgt, mem, memory_key_padding_mask=mask.flatten(1).cuda(), pos=pos.cuda().flatten(2).permute(2, 0, 1), query_pos=qpos)
    assert out.shape == tgt.shape and a1 is None and sa1 is None
    with pytest.raises(NotImplementedError):
        layer(tgt, mem, tgt_mask=torch.zeros(5, 5, device="cuda"))


def test_matcher_exclude_class_false():
    """HungarianMatcher / SetCriterion with exclude_class=False (box_utils.py:62,83-85): class-probability cost on the device LSAP
    path, indices bit-exact vs the oracle given the same fp32 class-cost term."""
    g = torch.Generator().manual_seed(13)
    matcher = box_utils.build_matcher(None)
    for trial in range(6):
        F_, q = 8, (2 if trial % 2 == 0 else 10)
        pred = torch.rand(F_, q, 4, generator=g) * 0.5 + 0.2
        logits = torch.randn(F_, q, 7, generator=g) * 2
        boxes = synth.make_batch(TINY4, 2, seed=100 + trial)["boxes"][:, :, :2].flatten(0, 1)
        tl = box_utils.prepare_targets(boxes.cuda(), None, None, center_crop=False)
        for t in tl:
            t["labels"] = torch.randint(0, 7, (len(t["boxes"]),), generator=g).cuda()
        outputs = {"pred_boxes": pred.cuda(), "pred_logits": logits.cuda()}
        got = matcher(outputs, tl, exclude_class=False)
        want = OL.hungarian_match(pred, [t["boxes"].cpu() for t in tl], pred_logits=logits, labels=[t["labels"].cpu() for t in tl])
        for (a, b), (c, d) in zip(got, want):
            assert torch.equal(a, c) and torch.equal(b, d)
        crit = box_utils.SetCriterion(22047, matcher, dict(loss_bbox_hand_boxes=5, loss_giou_hand_boxes=2), 0.1, ["boxes", "cardinality"])
        losses, idx = crit(outputs, tl, "hand_boxes", exclude_class=False)
        l1, giou, _ = OL.box_losses(pred, [t["boxes"].cpu() for t in tl], want)
        torch.testing.assert_close(losses["loss_bbox_hand_boxes"].cpu(), l1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(losses["loss_giou_hand_boxes"].cpu(), giou, rtol=1e-5, atol=1e-6)


def test_optimizer_state_roundtrip_and_reference_format(tmp_path):
    """TrainStep.state_dict() is torch.optim.AdamW's own format with optim_policy's two groups (utils/train_utils.py:28-48):
    (1) save -> reload into a fresh TrainStep -> the next step is identical; (2) a real torch.optim.AdamW built the reference's
    way loads it and produces the same update."""
    from helping_hand_for_egocentric_videos_amd.utils import checkpoint as CK
    cfg = TINY4
    esd, dsd = synth.encoder_state(cfg, seed=4), synth.decoder_state(cfg, seed=4)
    batch = to_dev(synth.make_batch(cfg, 2, seed=9))
    backbone = LaviLa.build_backbone(cfg, esd)

    def fresh():
        d = tfm_decoder.build_decoder(cfg, dsd)
        return TrainStep(cfg, backbone, d, lr=1e-4), d

    def eval_step(ts, d):
        d.eval()
        ts.arena.zero_grad()
        ts.losses(batch)["total_loss"].backward()
        ts.optimizer_step()

    ts, dec = fresh()
    for _ in range(3):
        eval_step(ts, dec)
    path = CK.save_runtime_checkpoint(CK.make_save_dict(dec, epoch=1, best_acc=0.5, iteration=ts.iteration, optimizer_state=ts),
                                      str(tmp_path / "runtime.pth.tar"))
    sd = ts.state_dict()
    g0 = [n for n, p in dec.named_parameters() if OS.no_decay(n)]
    assert len(sd["param_groups"]) == 2 and sd["param_groups"][0]["weight_decay"] == 0.0 and len(sd["param_groups"][0]["params"]) == len(g0)
    assert all(float(v["step"]) == 3.0 for v in sd["state"].values())
    names = g0 + [n for n, p in dec.named_parameters() if not OS.no_decay(n)]
    assert not any(names[i].startswith(("class_embed", "vid_proj")) for i in sd["state"])         # never updated -> no state, as in torch
    # (1) resume
    ts2, dec2 = fresh()
    info = CK.resume_train_step(ts2, path)
    assert info == {"epoch": 1, "best_acc": 0.5, "iteration": 3} and ts2.iteration == 3
    assert ts2.decoder.transformer._seed == ts.decoder.transformer._seed
    eval_step(ts, dec)
    eval_step(ts2, dec2)
    torch.testing.assert_close(ts2.arena.params, ts.arena.params, rtol=1e-4, atol=2e-6)      # atomics order in the LayerNorm / bias reductions
    # (2) the reference's optimizer (run/train.py:519-520 with optim_policy's groups) loads our state and agrees on the next update
    ts3, dec3 = fresh()
    CK.resume_train_step(ts3, path)
    named = [(n, p) for n, p in dec3.named_parameters() if p.requires_grad]
    opt = torch.optim.AdamW([{"params": [p for n, p in named if OS.no_decay(n)], "lr": 1e-4, "weight_decay": 0.0},
                             {"params": [p for n, p in named if not OS.no_decay(n)], "lr": 1e-4, "weight_decay": 1e-5}], lr=1e-4, weight_decay=1e-5)
    opt.load_state_dict(torch.load(path, weights_only=False)["optimizer"])
    dec3.eval()
    ts3.arena.zero_grad()
    ts3.losses(batch)["total_loss"].backward()
    for n, p in named:                                                   # torch.optim skips parameters without a gradient
        if n not in ts3.arena.touched:
            p.grad = None
    opt.step()
    torch.testing.assert_close(ts3.arena.params, ts.arena.params, rtol=1e-4, atol=2e-6)
    # and back: a state dict written by torch.optim.AdamW itself loads into a TrainStep
    ts4, dec4 = fresh()
    dec4.load_state_dict(dec3.state_dict())
    ts4.load_state_dict(opt.state_dict())
    assert ts4.iteration == 4 and all(v == 4 for n, v in ts4.arena.steps.items())
    torch.testing.assert_close(ts4.m, ts.m, rtol=1e-3, atol=1e-7)            # moments written by torch.optim.AdamW == ours after 4 steps
    torch.testing.assert_close(ts4.v, ts.v, rtol=1e-3, atol=1e-10)
    eval_step(ts, dec)
    eval_step(ts4, dec4)
    # a fifth step from (1e-5-)different weights: elements whose gradient is at rounding-noise level move by up to lr, the rest agree
    d = (ts4.arena.params - ts.arena.params).abs()
    assert float(d.max()) <= 2e-4 and float(d.mean()) < 1e-6, (float(d.max()), float(d.mean()))


def test_class_head_argmax_and_cardinality_vs_oracle_r7_r13():
    """R7 / R13 where the metric is not trivially `every query is an object`: the no-object bias is raised so that the last class
    wins for part of the queries (with the synthetic weights it never does).  Both head paths -- materialised `pred_logits`
    (module default) and `pred_logits_argmax` (TrainStep(fast_heads=True), the benchmarked one) -- against
    oracle.losses.cardinality_error (box_utils.py:142-154) and the oracle's argmax, (1) on the GPU's own hs: exact wherever the
    top-2 margin exceeds the head's 2e-5 bound, cardinality equal; (2) end to end vs the fp32 oracle's own decoder."""
    cfg = TINY16
    dsd = synth.decoder_state(cfg, seed=3)
    B, T = 3, cfg.num_frames
    feats = torch.randn(B, T, cfg.patches_per_frame, cfg.embed_dim, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
    with torch.no_grad():
        _, rhs0 = OD.objdecoder_forward(feats, dsd, cfg, compute_logits=False)
        lg0 = torch.nn.functional.linear(rhs0[-1], dsd["class_embed.weight"], dsd["class_embed.bias"])       # [B,Q,K]
        # the bias that makes `no object` win for about half of the (clip, query) pairs
        gap = lg0[..., :-1].max(-1).values - lg0[..., -1]
        dsd["class_embed.bias"] = dsd["class_embed.bias"].clone()
        dsd["class_embed.bias"][-1] += float(gap.median())
        ro, rhs = OD.objdecoder_forward(feats, dsd, cfg)
    batch = synth.make_batch(cfg, B, seed=21)
    nq = cfg.num_queries
    crit = box_utils.SetCriterion(22047, box_utils.build_matcher(None), {"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5,
                                  "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}, 0.1, ["boxes", "cardinality"]).cuda()
    results = {}
    for full in (True, False):
        dec = tfm_decoder.build_decoder(cfg, dsd).eval()
        dec.materialize_logits = full
        with torch.no_grad():
            out, hs, _, _ = dec(feats.cuda())
        own = torch.nn.functional.linear(hs[-1].cpu(), dsd["class_embed.weight"], dsd["class_embed.bias"])   # [B,Q,K] on the GPU's hs
        own_t = own[:, None].expand(-1, T, -1, -1).flatten(0, 1)                                              # [B*T,Q,K]
        am = (out["pred_logits"].argmax(-1) if full else out["pred_logits_argmax"]).cpu()
        assert am.shape == (B * T, cfg.dec_queries)
        top2 = own_t.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 2 * 2e-5 * float(own.abs().max())
        assert bool(safe.float().mean() > 0.95)
        assert torch.equal(am[safe], own_t.argmax(-1)[safe])
        frac_noobj = float((am == own.shape[-1] - 1).float().mean())
        assert 0.2 < frac_noobj < 0.8, frac_noobj                      # the test really exercises both outcomes
        for bt, sl, qs in (("hand_boxes", slice(0, 2), slice(0, 2)), ("obj_boxes", slice(2, 4), slice(2, nq))):
            raw = batch["boxes"][:, :, sl].flatten(0, 1)
            _, _, ld = box_utils.compute_box_loss(bt, crit, out, raw.cuda(), None, None, n_queries=nq, return_loss_dict=True)
            tg = OL.prepare_targets(raw)
            want_own = OL.cardinality_error(own_t[:, qs], tg)
            want_e2e = OL.cardinality_error(ro["pred_logits"][:, qs], tg)
            got = float(ld[f"cardinality_error_{bt}"])
            results[(full, bt)] = (got, float(want_own), float(want_e2e))
            if bool(safe[:, qs].all()):
                assert abs(got - float(want_own)) < 1e-6, (full, bt, got, float(want_own))
            else:                                                        # a near-tie inside the slice: at most that many counts off
                assert abs(got - float(want_own)) <= float((~safe[:, qs]).sum()) / (B * T) + 1e-6
            # end to end the hs of the two sides differ by the bf16 K/V rounding (logits to ~3e-3 of their scale, measured; bound
            # 2e-2): a (frame, query) whose no-object decision sits inside that band may flip, each flip moves the metric by 1/(B*T)
            band = 2 * 2e-2 * float(ro["pred_logits"].abs().max())
            rl = ro["pred_logits"][:, qs]
            shaky = ((rl[..., :-1].max(-1).values - rl[..., -1]).abs() < band).sum()
            assert abs(got - float(want_e2e)) <= float(shaky) / (B * T) + 1e-6, (full, bt, got, float(want_e2e), int(shaky))
    for bt in ("hand_boxes", "obj_boxes"):
        assert results[(True, bt)][0] == results[(False, bt)][0]        # the two head paths agree with each other exactly
    print("cardinality (got, oracle on shared hs, oracle end to end):", results)


def test_box_ops_vs_oracle_r15():
    """utils/box_ops.py (box_ops.py:9-61 of the reference: conversions, IoU with union + 1e-4, GIoU) on the GPU vs the oracle's
    restatement on identical boxes, incl. degenerate (zero-area), nested, disjoint and identical pairs."""
    from helping_hand_for_egocentric_videos_amd.utils import box_ops
    g = torch.Generator().manual_seed(11)
    c = torch.rand(40, 2, generator=g)
    wh = torch.rand(40, 2, generator=g) * 0.5
    wh[3] = 0                                                            # zero-area box
    a = torch.cat([c, wh], -1)
    b = torch.cat([torch.rand(17, 2, generator=g), torch.rand(17, 2, generator=g) * 0.5 + 0.01], -1)
    b[0] = a[0]                                                          # identical pair
    b[1] = torch.tensor([a[1, 0], a[1, 1], a[1, 2] * 0.5, a[1, 3] * 0.5])   # nested
    b[2] = torch.tensor([5.0, 5.0, 0.1, 0.1])                            # far away
    axy, bxy = OL.box_cxcywh_to_xyxy(a), OL.box_cxcywh_to_xyxy(b)
    gx = box_ops.box_cxcywh_to_xyxy(a.cuda())
    torch.testing.assert_close(gx.cpu(), axy, rtol=0, atol=1e-7)
    torch.testing.assert_close(box_ops.box_xyxy_to_cxcywh(gx).cpu(), OL.box_xyxy_to_cxcywh(axy), rtol=0, atol=1e-7)
    iou, union = box_ops.box_iou(gx, bxy.cuda())
    riou, runion = OL.box_iou(axy, bxy)
    torch.testing.assert_close(iou.cpu(), riou, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(union.cpu(), runion, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(box_ops.generalized_box_iou(gx, bxy.cuda()).cpu(), OL.generalized_box_iou(axy, bxy), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(box_ops.box_area(gx).cpu(), (axy[:, 2] - axy[:, 0]) * (axy[:, 3] - axy[:, 1]), rtol=1e-6, atol=1e-7)
    with pytest.raises(AssertionError):                                  # box_ops.py:51-52: malformed boxes are rejected
        box_ops.generalized_box_iou(torch.tensor([[0.5, 0.5, 0.1, 0.9]]).cuda(), bxy.cuda())


def test_arena_adamw_decides_on_the_device():
    """hh_adamw_arena_step (the step's optimizer call): per-parameter skip / step count / bias correction from device-side flags.
    Against hh_adamw_step range by range (to rounding), torch.optim.AdamW semantics for grad-less parameters (untouched: p, m, v
    and the step count stay; no weight decay), parameters with different step counts in one launch, gradient arena cleared."""
    torch.manual_seed(0)
    sizes = [4, 1024, 12, 8192 + 4, 260, 4, 40000, 16]                   # 4-aligned segments, some straddling the 8192-element chunks
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = int(offs[-1])
    dev = "cuda"
    p0, g0 = torch.randn(n, device=dev), torch.randn(n, device=dev) * 0.1
    m0, v0 = torch.randn(n, device=dev) * 0.01, torch.rand(n, device=dev) * 0.01
    seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    decay = [1, 1, 0, 1, 0, 0, 1, 0]
    steps0 = [0, 3, 3, 0, 7, 7, 1, 0]
    flags = [1.0, 1.0, 0.0, 2.0, 1.0, 0.0, 0.5, 1.0]                        # > 0 = touched on some rank (sums of rank flags)
    seg_decay = torch.tensor(decay, dtype=torch.int32, device=dev)
    seg_step = torch.tensor(steps0, dtype=torch.int32, device=dev)
    seg_flag = torch.tensor(flags, dtype=torch.float32, device=dev)
    seg_coef = torch.zeros(2 * len(sizes), device=dev)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-2
    p, g, m, v = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    ops.adamw_arena_step(p, g, m, v, seg_off, seg_decay, seg_step, seg_flag, seg_coef, lr, b1, b2, eps, wd, zero_grads=True)
    pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
    for s_ in range(len(sizes)):
        a, b = int(offs[s_]), int(offs[s_ + 1])
        if flags[s_] > 0:
            ops.adamw_step(pr[a:b], g0[a:b].contiguous(), mr[a:b], vr[a:b], lr, b1, b2, eps, wd if decay[s_] else 0.0, steps0[s_] + 1)
    # same formulas; the compiler may contract the vectorised and the scalar kernel's multiply-adds differently -> to rounding
    torch.testing.assert_close(p, pr, rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(m, mr, rtol=2e-6, atol=1e-8)           # measured 8.7e-10 on moments of magnitude 1e-2
    torch.testing.assert_close(v, vr, rtol=2e-6, atol=1e-10)
    assert seg_step.tolist() == [t + (1 if f > 0 else 0) for t, f in zip(steps0, flags)]
    assert float(g.abs().max()) == 0.0
    for s_ in (2, 5):                                                    # untouched: nothing moved, not even by weight decay
        a, b = int(offs[s_]), int(offs[s_ + 1])
        assert torch.equal(p[a:b], p0[a:b]) and torch.equal(m[a:b], m0[a:b]) and torch.equal(v[a:b], v0[a:b])
    # against torch.optim.AdamW itself on one touched segment (first step of segment 3)
    a, b = int(offs[3]), int(offs[4])
    tp = torch.nn.Parameter(p0[a:b].clone())
    opt = torch.optim.AdamW([tp], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    tp.grad = g0[a:b].clone()
    opt.state[tp] = {"step": torch.tensor(0.0), "exp_avg": m0[a:b].clone(), "exp_avg_sq": v0[a:b].clone()}
    opt.step()
    torch.testing.assert_close(p[a:b], tp.detach(), rtol=1e-5, atol=1e-7)
    # zero_grads=False keeps the gradients
    g2 = g0.clone()
    ops.adamw_arena_step(p, g2, m, v, seg_off, seg_decay, seg_step, seg_flag, seg_coef, lr, b1, b2, eps, wd, zero_grads=False)
    assert torch.equal(g2, g0)


def test_train_step_skips_gradless_parameters_like_torch_adamw():
    """A clip length != num_frames skips the trajectory branch (tfm_decoder.py:212-215): frame_index / frame_proj receive no gradient
    and torch.optim.AdamW leaves them alone (no weight decay, no step count).  TrainStep.step does the same through the device-side
    flags, and keeps per-parameter step counts when they come back."""
    cfg = TINY4
    esd, dsd = synth.encoder_state(cfg, seed=4), synth.decoder_state(cfg, seed=4)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec, lr=1e-4, weight_decay=1e-2)
    batch4 = to_dev(synth.make_batch(cfg, 2, seed=9))
    cfg2 = cfg.with_(num_frames=2)
    batch2 = to_dev(synth.make_batch(cfg2, 2, seed=9))
    ts.step(batch4)
    fi0, fp0 = dec.frame_index.weight.detach().clone(), dec.frame_proj.weight.detach().clone()
    q0 = dec.query_embed.weight.detach().clone()
    try:
        ts.step(batch2)                                                 # T = 2 != num_frames = 4: no trajectory branch
    except Exception as e:                                              # the frozen tower may not accept T != num_frames at this size
        pytest.skip("T != num_frames not runnable here: %r" % (e,))
    assert torch.equal(dec.frame_index.weight.detach(), fi0) and torch.equal(dec.frame_proj.weight.detach(), fp0)
    assert not torch.equal(dec.query_embed.weight.detach(), q0)
    steps = ts.arena.steps
    assert steps["frame_index.weight"] == 1 and steps["frame_proj.weight"] == 1 and steps["query_embed.weight"] == 2
    ts.step(batch4)
    steps = ts.arena.steps
    assert steps["frame_index.weight"] == 2 and steps["query_embed.weight"] == 3
    assert not torch.equal(dec.frame_index.weight.detach(), fi0)


def test_resume_remixes_the_rank_into_the_dropout_seed(monkeypatch):
    """A checkpoint is written by one rank; after a data-parallel resume the ranks must keep drawing DIFFERENT attention-dropout
    masks (as Cross_Attention.next_dropout_seed gives them from a fresh start), rank 0 continuing the saved stream exactly."""
    from helping_hand_for_egocentric_videos_amd import step as step_mod
    cfg = TINY4
    backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=4))
    ts = TrainStep(cfg, backbone, tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4)))
    ts.step(to_dev(synth.make_batch(cfg, 2, seed=9)))
    sd = ts.state_dict()
    saved = sd["hh"]["xattn_seed"]
    assert saved is not None
    seeds = {}
    for rank in (0, 1, 2):
        monkeypatch.setattr(step_mod, "world", lambda r=rank: (4, r))
        ts2 = TrainStep(cfg, backbone, tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4)))
        ts2.load_state_dict(sd)
        seeds[rank] = ts2.decoder.transformer._seed
        assert ts2.arena.steps == ts.arena.steps
    assert seeds[0] == saved and len(set(seeds.values())) == 3
    assert all(0 <= v <= 0x7FFFFFFF for v in seeds.values())


def test_linear_x3_accepts_sizes_that_are_not_multiples_of_four():
    """nn.Linear in the reference takes any (in, out) size (e.g. a class head with num_classes + 1 % 4 != 0); LinearX3 zero-pads
    the operands of hh_qgemm_f32x3 and slices the results: forward and all three gradients vs torch fp32."""
    from helping_hand_for_egocentric_videos_amd.model.qside import LinearX3, linear_x3
    g = torch.Generator().manual_seed(3)
    for K, N in ((30, 7), (64, 22047 % 1000 + 2), (13, 4), (512, 10)):
        lin = LinearX3(K, N).cuda()
        x = torch.randn(5, 3, K, generator=g).cuda().requires_grad_(True)
        ref_w, ref_b = lin.weight.detach().clone().requires_grad_(True), lin.bias.detach().clone().requires_grad_(True)
        xr = x.detach().clone().requires_grad_(True)
        y = lin(x)
        yr = torch.nn.functional.linear(xr, ref_w, ref_b)
        assert y.shape == yr.shape == (5, 3, N)
        w = torch.randn(y.shape, generator=g).cuda()
        (y * w).sum().backward()
        (yr * w).sum().backward()
        scale = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert scale(y, yr) < 2e-5 and scale(x.grad, xr.grad) < 2e-5, (K, N)
        assert scale(lin.weight.grad, ref_w.grad) < 2e-5 and scale(lin.bias.grad, ref_b.grad) < 2e-5, (K, N)
        yy = linear_x3(x.detach(), lin.weight.detach(), lin.bias.detach(), relu=True)
        assert scale(yy, torch.relu(yr.detach())) < 2e-5


def test_mcq_scorer_pipelined_equals_mcq_forward():
    """McqScorer (the next item batch's frozen towers prefetched on the encoder stream, bench.py's EgoMCQ leg) returns exactly what
    the back-to-back mcq_forward (run/test_EgoMCQ.py:56-83) returns, with and without a prefetched batch, for changing batches."""
    cfg = TINY16
    backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=4))
    dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4)).eval()
    items = [synth.make_mcq_item(cfg, 2, seed=s_) for s_ in (1, 2, 3)]
    items = [(it["video"].cuda(), it["text"].cuda()) for it in items]
    want = [mcq_forward(backbone, dec, v, t, cfg) for v, t in items]
    scorer = McqScorer(backbone, dec, cfg)
    for i, (v, t) in enumerate(items):
        nxt = items[i + 1] if i + 1 < len(items) else None
        got = scorer(v, t, next_item=nxt)
        assert torch.equal(got, want[i]), i
    assert scorer._pending is None
    assert torch.equal(scorer(*items[1]), want[1])                       # no prefetch pending: encodes in place
    scorer.prefetch(*items[0])
    assert torch.equal(scorer(*items[2]), want[2])                       # a stale prefetch is dropped, not used


def test_fused_box_tail_equals_the_two_compute_box_loss_calls():
    """box_utils.step_box_losses (the step's fast path: hh_box_loss_fwd x 2 + ONE hh_box_tail_fwd launch for every scalar of
    box_utils.py:142-173,445-461) == compute_box_loss('hand_boxes') + compute_box_loss('obj_boxes'): totals, the criterion's unweighted
    terms, both cardinality errors, and d(lh + 0.7 lo) / d pred_boxes."""
    g = torch.Generator().manual_seed(11)
    F_, Q, nq = 96, 13, 12
    crit = box_utils.SetCriterion(22047, box_utils.build_matcher(None), {"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5,
                                  "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}, 0.1, ["boxes", "cardinality"]).cuda()
    batch = synth.make_batch(TINY16.with_(num_queries=nq), F_ // 16, seed=3)
    hand = batch["boxes"][:, :, :2].flatten(0, 1).cuda()
    objb = batch["boxes"][:, :, 2:].flatten(0, 1).cuda()
    base = (torch.rand(F_, Q, 4, generator=g) * 0.5 + 0.2).cuda()
    am = torch.randint(22040, 22048, (F_, Q), generator=g).cuda()                      # some queries predict the no-object class 22047
    res = {}
    for fused in (True, False):
        pred = base.clone().requires_grad_(True)
        det = {"pred_boxes": pred, "pred_logits": None, "pred_logits_argmax": am, "num_classes": 22048, "aux_outputs": []}
        mh = crit.matcher.match_raw(pred, 0, 2, hand)
        mo = crit.matcher.match_raw(pred, 2, nq - 2, objb)
        nb = torch.stack([mh["count"].sum(), mo["count"].sum()]).float().clamp(min=1)
        if fused:
            lh, lo, _, _, terms = box_utils.step_box_losses(crit, det, hand, objb, nq, nb, mh, mo)
        else:
            lh, _, dh = box_utils.compute_box_loss("hand_boxes", crit, det, hand, None, None, n_queries=nq, num_boxes=nb[0], match=mh, return_loss_dict=True)
            lo, _, do = box_utils.compute_box_loss("obj_boxes", crit, det, objb, None, None, n_queries=nq, num_boxes=nb[1], match=mo, return_loss_dict=True)
            terms = dict(dh, **do)
        (lh + 0.7 * lo).backward()
        res[fused] = (lh.detach(), lo.detach(), {k: v.detach() for k, v in terms.items()}, pred.grad.clone())
    a, b = res[True], res[False]
    torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(a[1], b[1], rtol=1e-6, atol=1e-7)
    assert set(a[2]) == set(b[2]) and len(a[2]) == 6
    for k in a[2]:
        torch.testing.assert_close(a[2][k].float(), b[2][k].float(), rtol=1e-6, atol=1e-7, msg=k)
    torch.testing.assert_close(a[3], b[3], rtol=1e-5, atol=1e-8)
    assert float(a[2]["cardinality_error_hand_boxes"]) > 0 and float(a[3].abs().max()) > 0


def test_c5_egomcq_item_at_full_width_vs_oracle():
    """BASELINE config 5 at FULL width (C5 was pinned at TINY width and through a self-comparison only): one EgoMCQ item -- 5
    candidate clips of 16 x 224 px + 1 query text -- through mcq_forward against oracle.step.mcq_forward on this box's host cores
    (run/test_EgoMCQ.py:56-83): similarity scores within 5e-3 absolute, the predicted answer equal."""
    from helping_hand_for_egocentric_videos_amd import C2
    cfg = C2
    esd, dsd = synth.encoder_state(cfg, seed=3), synth.decoder_state(cfg, seed=3)
    item = synth.make_mcq_item(cfg, 1, seed=31)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    with torch.no_grad():
        got = mcq_forward(backbone, dec, item["video"].cuda(), item["text"].cuda(), cfg).float().cpu()
        want = OS.mcq_forward(esd, dsd, item["video"], item["text"], cfg)
    assert got.shape == want.shape == (1, 5)
    err = float((got - want).abs().max())
    top2 = want[0].topk(2).values
    record("c5_full_width_mcq", "max |score - oracle| (cosine similarities)", err, 5e-3)
    record("c5_full_width_mcq", "statistic: oracle's top-2 score margin", float(top2[0] - top2[1]), 0.0)
    assert err <= 5e-3, (got, want)
    if float(top2[0] - top2[1]) > 2 * 5e-3:                       # (an untrained model's five scores can tie within the bound)
        assert int(got.argmax(-1)) == int(want.argmax(-1))


@pytest.mark.parametrize("T,img", [(3, 196), (16, 224 - 28), (5, 140)], ids=["M588", "M3136", "M500"])
def test_decoder_any_memory_length_on_the_kv_free_path(T, img):
    """Round 6: the memory-space cross-attention for EVERY memory length -- M = T * n is rounded up to a multiple of 128
    with zero rows that hh_mattn_fwd / _bwd mask out of the softmax (keys_valid); no silent fall-back to the projected-K/V route.  Patch-16-like
    grids (n = 196: M = 588 and 3136) and M = 500 (not even a multiple of the 32-key chunk): forward and every parameter gradient against the fp32
    oracle within the bounds of test_decoder_forward_backward_vs_oracle, and against the legacy projected-K/V path on the same inputs."""
    cfg = TINY4.with_(num_frames=T, img_size=img)
    n = cfg.patches_per_frame
    assert (T * n) % 128 != 0
    dsd = synth.decoder_state(cfg, seed=3)
    dec = tfm_decoder.build_decoder(cfg, dsd).eval()
    assert dec.transformer.kv_free
    B = 2
    feats = torch.randn(B, T, n, cfg.embed_dim, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
    out, hs, _, _ = dec(feats.cuda())
    params = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    ro, rhs = OD.objdecoder_forward(feats, params, cfg)
    e_hs = scaled_err(hs, rhs)
    record("decoder_any_M_%d" % (T * n), "hs scaled error vs fp32 oracle", e_hs, 2e-2)
    assert e_hs < 2e-2
    assert float((out["pred_boxes"].detach().cpu() - ro["pred_boxes"]).abs().max()) < 5e-3
    g = torch.Generator().manual_seed(1)
    w, wb = torch.randn(rhs.shape, generator=g), torch.randn(ro["pred_boxes"].shape, generator=g)
    ((hs * w.cuda()).sum() + (out["pred_boxes"] * wb.cuda()).sum()).backward()
    ((rhs * w).sum() + (ro["pred_boxes"] * wb).sum()).backward()
    rel = {}
    for name, p_ in dec.named_parameters():
        rg = params[name].grad
        if rg is None:
            continue
        gg = p_.grad.detach().cpu()
        if "multihead_attn.in_proj_bias" in name:
            C = cfg.dec_dim
            sel = torch.cat([torch.arange(0, C), torch.arange(2 * C, 3 * C)])
            gg, rg = gg[sel], rg[sel]
        rel[name] = float((gg - rg).norm() / (rg.norm() + 1e-12))
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    med = float(np.median(list(rel.values())))
    record("decoder_any_M_%d" % (T * n), "end-to-end gradient rel-L2 vs fp32 oracle: worst tensor (%s)" % worst[0][0], worst[0][1], 9.5e-2)
    record("decoder_any_M_%d" % (T * n), "end-to-end gradient rel-L2 vs fp32 oracle: median", med, 4.4e-2)
    assert len(rel) > 100 and max(rel.values()) < 9.5e-2 and med < 4.4e-2, worst
    # the legacy projected-K/V route on the same inputs (an independent implementation of the same layer; hh_xattn_* needs M % 32 == 0)
    if (T * n) % 32:
        return
    dec2 = tfm_decoder.build_decoder(cfg, dsd).eval()
    dec2.transformer.kv_free = False
    out2, hs2, _, _ = dec2(feats.cuda())
    assert scaled_err(hs, hs2) < 1e-2 and float((out["pred_boxes"] - out2["pred_boxes"]).abs().max()) < 5e-3
