"""Checkpoint compatibility helpers (CPU): inflate_positional_embeds vs the imported reference function when the
reference tree is mounted, module.-prefix stripping, the rolling runtime checkpoint window, round trip of the saved dict."""
import os
import sys

import pytest
import torch

from helping_hand_for_egocentric_videos_amd import synth, TINY4, TINY16
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.utils import checkpoint as ck


def test_inflate_matches_reference_semantics():
    g = torch.Generator().manual_seed(0)
    emb4 = torch.randn(1, 4, 32, generator=g)
    cur16 = {"visual.temporal_embed": torch.zeros(1, 16, 32)}
    out = ck.inflate_positional_embeds(cur16, {"visual.temporal_embed": emb4.clone()}, num_frames=16)
    assert out["visual.temporal_embed"].shape == (1, 16, 32)
    exp = torch.nn.functional.interpolate(emb4.unsqueeze(0), (16, 32), mode="bilinear").squeeze(0)
    assert torch.equal(out["visual.temporal_embed"], exp)
    z = ck.inflate_positional_embeds(cur16, {"visual.temporal_embed": emb4.clone()}, num_frames=16, load_temporal_fix="zeros")
    assert torch.equal(z["visual.temporal_embed"][:, :4], emb4) and z["visual.temporal_embed"][:, 4:].abs().max() == 0
    cut = ck.inflate_positional_embeds({"visual.temporal_embed": torch.zeros(1, 2, 32)}, {"visual.temporal_embed": emb4.clone()}, num_frames=2)
    assert torch.equal(cut["visual.temporal_embed"], emb4[:, :2])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _refload
    if _refload.available():                       # bit-equality with the reference's own function (run/test_egtea.py:46-96)
        src = open(os.path.join(_refload.REF_ROOT, "run", "test_egtea.py")).read()
        start, end = src.index("def inflate_positional_embeds("), src.index("@ex.main")
        ns = {"torch": torch, "F": torch.nn.functional}
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            exec(src[start:end], ns)
            for fix in ("bilinear", "interp", "zeros"):
                r = ns["inflate_positional_embeds"](cur16, {"visual.temporal_embed": emb4.clone()}, num_frames=16, load_temporal_fix=fix)
                m = ck.inflate_positional_embeds(cur16, {"visual.temporal_embed": emb4.clone()}, num_frames=16, load_temporal_fix=fix)
                assert torch.equal(r["visual.temporal_embed"], m["visual.temporal_embed"])


def test_load_4_frame_checkpoints_into_16_frame_models(tmp_path):
    sd4 = synth.encoder_state(TINY4, seed=1)
    ckpt = {"state_dict": {"module." + k: v for k, v in sd4.items()}}
    bb16 = LaviLa.build_backbone(TINY16, None, device="cpu")
    ck.load_backbone_checkpoint(bb16, ckpt)
    assert bb16.visual.temporal_embed.shape == (1, 16, TINY16.embed_dim)
    assert torch.equal(bb16.visual.pos_embed.data, sd4["visual.pos_embed"])
    d4 = synth.decoder_state(TINY4, seed=1)
    dec16 = tfm_decoder.build_decoder(TINY16.with_(num_queries=4), None, device="cpu")
    info = ck.load_decoder_checkpoint(dec16, {"state_dict": d4, "epoch": 3, "best_acc": 41.5, "iteration": 7500, "optimizer": None})
    assert info["epoch"] == 3 and info["iteration"] == 7500
    assert dec16.temporal_embed.shape == (1, 16, 512) and dec16.frame_index.weight.shape == (16, 512)
    # save / rolling window / reload
    fn = str(tmp_path / "runtime.pth.tar")
    for i in range(12):
        p = ck.save_runtime_checkpoint(ck.make_save_dict(dec16, i, 0.0, i * 10, {"step": i}), fn, keep=10)
        os.rename(p, str(tmp_path / f"runtime_2000_01_01_00_{i:02d}.pth.tar"))   # the reference's stamp has minute resolution: give each file its own minute
    assert len(list(tmp_path.glob("runtime_*.pth.tar"))) <= 11
    last = sorted(tmp_path.glob("runtime_*.pth.tar"))[-1]
    dec_b = tfm_decoder.build_decoder(TINY16.with_(num_queries=4), None, device="cpu")
    info = ck.load_decoder_checkpoint(dec_b, str(last))
    assert info["optimizer"] == {"step": 11}
    for (k1, v1), (k2, v2) in zip(dec16.state_dict().items(), dec_b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_tokenized_text_cache_equals_the_tokenizer():
    """utils/text_cache.py (run/train.py:66-75): cached token rows == tokenising the batch, only unseen
    strings reach the tokenizer, the table grows past its initial capacity, duplicates inside a batch are tokenised once."""
    import torch
    from helping_hand_for_egocentric_videos_amd.utils.text_cache import TokenizedTextCache
    calls = []

    def fake_tokenizer(texts):                                        # deterministic stand-in: ids from the characters, eot = max id
        calls.append(list(texts))
        out = torch.zeros((len(texts), 77), dtype=torch.int64)
        for i, t in enumerate(texts):
            ids = [49406] + [ord(c) + 100 for c in t][:75] + [49407]
            out[i, :len(ids)] = torch.tensor(ids)
        return out

    cache = TokenizedTextCache(fake_tokenizer, device="cpu", capacity=4)
    a = ["#C C opens the door", "", "#C C picks a cup", "#C C opens the door"]
    got = cache(a)
    assert torch.equal(got, fake_tokenizer(a)) and len(cache) == 3 and calls[0] == ["#C C opens the door", "", "#C C picks a cup"]
    n_calls = len(calls)
    b = ["#C C picks a cup", "#O a man X walks", "", "#C C cuts an onion", "#C C washes the knife", "#C C opens the door"]
    got = cache(b)
    assert calls[n_calls] == ["#O a man X walks", "#C C cuts an onion", "#C C washes the knife"]          # only the unseen ones
    assert torch.equal(got, fake_tokenizer(b)) and len(cache) == 6 and cache.table.shape[0] >= 6
    assert torch.equal(cache(a), fake_tokenizer(a)) and cache.hits >= 7
    assert got.dtype == torch.int64 and cache.table.dtype == torch.int32          # stored narrow, returned as the tokenizer returns them
    # bounded: beyond max_rows unseen captions bypass the table (tokenised + uploaded for the call only), results unchanged
    small = TokenizedTextCache(fake_tokenizer, device="cpu", capacity=2, max_rows=3)
    c = ["a b", "c d", "e f", "g h", "a b", "i j"]
    assert torch.equal(small(c), fake_tokenizer(c)) and len(small) == 3 and small.table.shape[0] == 3 and small.bypassed == 2
    assert torch.equal(small(c), fake_tokenizer(c)) and len(small) == 3 and small.bypassed == 4
    # the cap reached on a table of a realistic width: a call with bypassed captions must not copy the table (torch.cat of
    # the 0.65 GB table per step) -- no tensor as large as the table is allocated, cached and bypassed rows land in their own positions
    big = TokenizedTextCache(fake_tokenizer, device="cpu", capacity=512, max_rows=512)
    fill = ["cap %d" % i for i in range(512)]
    big(fill)
    assert len(big) == 512 and big.table.shape[0] == 512
    seen = []
    real_cat, real_empty = torch.cat, torch.empty
    try:
        torch.cat = lambda ts, *a, **k: (seen.append(sum(t.numel() for t in ts)), real_cat(ts, *a, **k))[1]
        torch.empty = lambda *a, **k: (lambda r: (seen.append(r.numel()), r)[1])(real_empty(*a, **k))
        mixed = ["new x", "cap 7", "new y", "cap 500", "new x", "cap 7"]
        got = big(mixed)
        only_new = big(["new z", "new w"])
    finally:
        torch.cat, torch.empty = real_cat, real_empty
    assert torch.equal(got, fake_tokenizer(mixed)) and torch.equal(only_new, fake_tokenizer(["new z", "new w"]))
    assert max(seen) < big.table.numel() // 8, seen
