"""Pins the BENCHMARKED batch and the pipelined schedule (the reference step is run/train.py:103-203).

The full-width oracle comparisons run at B = 1-2 (the oracle needs ~10 s per clip on the host); bench.py runs B = 32, where M = 131 104
token rows send the persistent GEMM through its dynamic per-XCD tile counters, 16-m-tile groups and folded row tail at a scale no
oracle test reaches.  These tests tie the two together without the oracle, through a property of the path: clip i of a large batch
must get the result it gets in a small batch (every kernel is per-clip; `step.first_clips_check` states the two places where only the
fp32 summation ORDER may differ), and the frozen encoder must be bit-stable run to run while the decoder stream runs beside it (the
dynamic tile counters, opaque waits and LDS-DMA staging of DESIGN.md 4.1-4.2 were all debugged through timing-dependent failures).
"""
import pytest
import torch

from helping_hand_for_egocentric_videos_amd import C2, C4, synth
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep, first_clips_check
from _record import check

pytestmark = pytest.mark.gpu


def _build(cfg, B, seed):
    esd, dsd = synth.encoder_state(cfg, seed=0), synth.decoder_state(cfg, seed=0)
    backbone = LaviLa.build_backbone(cfg, esd)
    dec = tfm_decoder.build_decoder(cfg, dsd)
    ts = TrainStep(cfg, backbone, dec)
    batch = {k: v.cuda() for k, v in synth.make_batch(cfg, B, seed=seed).items()}
    return ts, batch


def _assert_first_clips(tag, rec):
    print(tag, rec)
    # clip 0: no row of it is in any GEMM row tail of either batch -> the encoder output must be the same BITS
    assert rec["encoder_bit_identical_clips_before_last"], rec
    # clip 1 is the small batch's last clip: its last 2 token rows (M = 2 * N = 32 * 256 + 2) go through the in-kernel row tail there and
    # through full tiles in the large batch (8 K-slices summed vs k-tiles in order: fp32 re-association before the bf16 rounding);
    # attention spreads those two rows' roundings over their frame / time column.  Bound: a few bf16 ulps of the feature-map scale.
    check(tag, "encoder last small-batch clip: max |diff| / scale (row-tail summation order)", rec["encoder_last_clip_max_abs_diff_over_scale"], 3e-2)
    # hs / boxes: the cross-attention cuts the keys into more slices when B * heads cannot fill the chip (fp32 re-association), and
    # clip 1's memory carries the roundings above
    check(tag, "hs: max |diff| / scale (key-slice count)", rec["hs_max_abs_diff_over_scale"], 5e-3)
    check(tag, "pred_boxes max |diff|", rec["pred_boxes_max_abs_diff"], 2e-3)
    # matching is bit-exact on identical boxes (tests/test_kernels_gpu.py); here the two runs' boxes differ by ~1e-4, which may flip a frame
    # whose two best assignments are tied to within that: then the loss (each side with its own matching) must still agree
    assert rec["matched_indices_equal"] or rec["matched_frames_equal_fraction"] >= 0.9, rec
    check(tag, "hand-box loss of the first clips: rel diff", rec["hand_box_loss_first_clips_rel_diff"], 1e-3)


def test_c2_benchmarked_batch_equals_small_batch():
    """Config 2 at the benchmarked B = 32 (full width): clips 0-1 of the batch == the same clips as a batch of 2."""
    ts, batch = _build(C2, 32, seed=1000)
    _assert_first_clips("c2_b32_vs_b2", first_clips_check(ts, batch, k=2))


def test_c4_benchmarked_batch_equals_small_batch():
    """Config 4 at the benchmarked B = 4 (T = 32, 336 px, full width) vs B = 2."""
    ts, batch = _build(C4, 4, seed=1004)
    _assert_first_clips("c4_b4_vs_b2", first_clips_check(ts, batch, k=2))


def test_pipelined_schedule_is_bit_stable_at_the_benchmarked_batch():
    """20 pipelined eval-mode steps at B = 32 with the decoder stream live: the frozen encoder's feature map (computed on the
    encoder stream while the previous step's decoder forward / backward / AdamW runs on the main stream, the text tower on a third)
    is bit-identical in all 20 steps and to the un-pipelined run -- a race detector for the persistent GEMM's dynamic tile counters,
    the asm-owned waits and the LDS-DMA staging of the attention kernels.  The decoder's parameters move (AdamW runs), the encoder's
    output must not."""
    ts, batch = _build(C2, 32, seed=1000)
    with torch.no_grad():
        ref_grid, ref_tmap = ts.encode(batch["video"], batch["text"])          # un-pipelined, alone on the chip
    ref_grid, ref_tmap = ref_grid.clone(), ref_tmap.clone()
    torch.cuda.synchronize()
    seen = []
    orig = ts._encoded

    def spy(b):
        grid, tmap = orig(b)
        seen.append((grid, tmap))
        return grid, tmap
    ts._encoded = spy
    ts.prefetch(batch)
    losses = []
    for i in range(20):
        out = ts.step(batch, next_batch=batch)         # train-mode decoder (dropout) + backward + AdamW beside the next encoder pass
        losses.append(out["total_loss"])
    torch.cuda.synchronize()
    assert len(seen) == 20
    for i, (grid, tmap) in enumerate(seen):
        assert torch.equal(grid, ref_grid), "step %d: encoder feature map differs from the un-pipelined run" % i
        assert torch.equal(tmap, ref_tmap), "step %d: text feature map differs from the un-pipelined run" % i
    assert all(bool(torch.isfinite(l)) for l in losses)
