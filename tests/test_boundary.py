"""CPU-side checks of the module-level drop-in boundary: the reference's caller scripts import their model
symbols from this package unchanged, the factory / constructors accept the scripts' own call-site arguments, and the resulting
state_dict schemas are the ones the reference checkpoints carry.

The import statements are READ from /root/reference/run/*.py at test time (dev container only; nothing of the reference is
stored here) and executed with `model` aliased to this package's model sub-package.  No compute runs: modules are built on the
meta device."""
import os
import re
import sys

import pytest
import torch

from helping_hand_for_egocentric_videos_amd import C1, synth

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (dev container only)")


class _alias_model_package:
    """sys.modules['model(.x)'] -> helping_hand_for_egocentric_videos_amd.model(.x) for the duration of the block."""

    NAMES = ("loss", "metric", "LaviLa", "tfm_decoder", "box_utils", "openai_model")

    def __enter__(self):
        import importlib
        self.saved = {k: sys.modules.get(k) for k in ["model"] + ["model." + n for n in self.NAMES]}
        pkg = importlib.import_module("helping_hand_for_egocentric_videos_amd.model")
        sys.modules["model"] = pkg
        for n in self.NAMES:
            sys.modules["model." + n] = importlib.import_module("helping_hand_for_egocentric_videos_amd.model." + n)
        return self

    def __exit__(self, *a):
        for k, v in self.saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _model_import_lines(script, first, last):
    lines = open(os.path.join(REF, "run", script)).read().splitlines()[first - 1:last]
    keep = [l for l in lines if re.match(r"\s*from model\.(?!tokenizer)", l)]        # tokenizer = host-side BPE, out of scope
    return keep


@needs_ref
@pytest.mark.parametrize("script,first,last,expect", [
    ("train.py", 19, 25, {"EgoNCE", "WordContrastiveLoss", "sim_matrix", "CLIP_OPENAI_TIMESFORMER_LARGE", "egomcq_accuracy_metrics",
                          "compute_tv_accuracy", "ObjDecoder", "Cross_Attention", "build_matcher", "SetCriterion", "compute_box_loss"}),
    ("test_EgoMCQ.py", 14, 19, {"CLIP_OPENAI_TIMESFORMER_LARGE", "egomcq_accuracy_metrics", "sim_matrix", "ObjDecoder", "Cross_Attention"}),
])
def test_reference_scripts_import_their_model_symbols_from_this_package(script, first, last, expect):
    lines = _model_import_lines(script, first, last)
    assert len(lines) >= 3, lines
    ns = {}
    with _alias_model_package():
        exec("\n".join(l.strip() for l in lines), ns)
    got = {k for k in ns if not k.startswith("__")}
    assert expect <= got, expect - got
    for k in expect:
        assert ns[k].__module__.startswith("helping_hand_for_egocentric_videos_amd."), (k, ns[k].__module__)


def _call_kwargs(script, func, first, last):
    """The literal keyword arguments of `func(...)` between two lines of a reference script (parsed, not copied)."""
    import ast
    src = "\n".join(open(os.path.join(REF, "run", script)).read().splitlines()[first - 1:last])
    tree = ast.parse("if True:\n" + "\n".join("    " + l for l in src.splitlines()) if src.startswith(" ") else src)
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == func:
            out = {}
            for kw in node.keywords:
                try:
                    out[kw.arg] = ast.literal_eval(kw.value)
                except ValueError:
                    out[kw.arg] = None
            return out
    raise AssertionError(f"{func} call not found in {script}:{first}-{last}")


@needs_ref
@pytest.mark.parametrize("script,first,last", [("train.py", 425, 431), ("test_EgoMCQ.py", 203, 219)])
def test_factory_accepts_the_scripts_call_and_yields_the_checkpoint_schema(script, first, last):
    from helping_hand_for_egocentric_videos_amd.model.LaviLa import CLIP_OPENAI_TIMESFORMER_LARGE
    kw = _call_kwargs(script, "CLIP_OPENAI_TIMESFORMER_LARGE", first, last)
    assert kw.get("num_frames") == 4 and kw.get("project_embed_dim") == 256
    with torch.device("meta"):
        backbone = CLIP_OPENAI_TIMESFORMER_LARGE(**kw)
    want = {k: tuple(v.shape) for k, v in synth.encoder_state(C1, seed=0).items()}
    got = {k: tuple(v.shape) for k, v in backbone.state_dict().items()}
    assert got == want, (set(got) ^ set(want), [k for k in got if k in want and got[k] != want[k]][:5])
    assert isinstance(backbone.visual.head, torch.nn.Identity)


@needs_ref
def test_decoder_and_criterion_accept_the_scripts_calls():
    from helping_hand_for_egocentric_videos_amd.model.tfm_decoder import Cross_Attention, ObjDecoder
    from helping_hand_for_egocentric_videos_amd.model.box_utils import SetCriterion, build_matcher
    tkw = _call_kwargs("train.py", "Cross_Attention", 440, 458)
    okw = _call_kwargs("train.py", "ObjDecoder", 440, 458)
    assert tkw == {"normalize_before": True, "return_intermediate_dec": True}
    okw.pop("transformer")
    okw["num_queries"], okw["feature_dim"] = C1.dec_queries, 1024            # `args.num_queries + 1`, `feature_dim` variables
    with torch.device("meta"):
        dec = ObjDecoder(transformer=Cross_Attention(**tkw), **okw)
    want = {k: tuple(v.shape) for k, v in synth.decoder_state(C1, seed=0).items()}
    got = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert got == want, set(got) ^ set(want)
    ckw = _call_kwargs("train.py", "SetCriterion", 459, 473)
    crit = SetCriterion(22047, matcher=build_matcher(None), weight_dict={"loss_bbox_hand_boxes": 5, "loss_bbox_obj_boxes": 5,
                        "loss_giou_hand_boxes": 2, "loss_giou_obj_boxes": 2}, eos_coef=ckw["eos_coef"], losses=ckw["losses"])
    assert crit.losses == ["boxes", "cardinality"]


def test_reference_forward_signatures_are_kept():
    """Parameter names/order of the forwards the callers (or the reference's own modules) use: tfm_decoder.py:76,183,255-264,
    420-430,463-473; LaviLa.py:575,650,660,672; loss.py:15,78; box_utils.py:43,206,250,445."""
    import inspect
    from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder, loss, box_utils, metric
    P = lambda f: list(inspect.signature(f).parameters)
    layer = ["self", "tgt", "memory", "tgt_mask", "memory_mask", "tgt_key_padding_mask", "memory_key_padding_mask", "pos", "query_pos",
             "counter", "num_frames", "seq_len"]
    assert P(tfm_decoder.TransformerDecoderLayer.forward) == layer
    assert P(tfm_decoder.TransformerDecoderLayer.forward_pre) == layer
    assert P(tfm_decoder.TransformerDecoder.forward) == [p for p in layer if p != "counter"]
    assert P(tfm_decoder.Cross_Attention.forward) == ["self", "src", "mask", "query_embed", "pos_embed"]
    assert P(tfm_decoder.ObjDecoder.forward) == ["self", "features", "use_checkpoint"]
    assert P(LaviLa.CLIP.forward) == ["self", "image", "text", "use_checkpoint", "norm_embed", "return_feature_map"]
    assert P(LaviLa.CLIP.encode_image) == ["self", "image", "use_checkpoint", "apply_project"]
    assert P(LaviLa.SpaceTimeTransformer.forward) == ["self", "x", "use_checkpoint"]
    assert P(loss.EgoNCE.forward)[:7] == ["self", "x", "mask_v", "mask_n", "multi_pad_mask", "strict_mask", "vn_threshold"]
    assert P(loss.WordContrastiveLoss.forward)[:4] == ["self", "noun_embeds", "pred_noun_embeds", "noun_gt_inds"]
    assert P(box_utils.HungarianMatcher.forward) == ["self", "outputs", "targets", "exclude_class"]
    assert P(box_utils.SetCriterion.forward) == ["self", "outputs", "targets", "box_type", "exclude_class"]
    assert P(box_utils.compute_box_loss)[:7] == ["box_type", "criterion", "detr_out", "target_boxes", "target_classes", "all_image_size", "n_queries"]
    extra = list(inspect.signature(box_utils.compute_box_loss).parameters.values())[7:]          # additions are keyword-only with defaults
    assert all(p.kind is inspect.Parameter.KEYWORD_ONLY and p.default is not inspect.Parameter.empty for p in extra)
    assert P(box_utils.prepare_targets) == ["boxes", "classes", "image_size", "center_crop"]
    assert P(metric.sim_matrix) == ["a", "b", "eps", "norm"]
