"""Loader for the upstream reference modules (dev container only).

Used ONLY by tests/golden/make_golden.py and tests/test_oracle_vs_reference.py, both of
which run in the development container where /root/reference is mounted.  Nothing here
travels to the GPU box as executable reference code: the reference is imported from where
it lies, with the three absent third-party packages (timm, torchvision, ftfy) stubbed by
the handful of trivial symbols the hot-path modules import.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("HH_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def _stub(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    sys.modules[name] = m
    return m


_loaded = False


def load():
    """Import the reference's model package; returns a namespace of its modules."""
    global _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch
    if not _loaded:
        import transformers  # noqa: F401  (must probe the real absence of torchvision first)
        if "timm" not in sys.modules:
            timm = _stub("timm")
            tm = _stub("timm.models")
            tl = _stub("timm.models.layers")
            tl.trunc_normal_ = torch.nn.init.trunc_normal_
            tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)

            class DropPath(torch.nn.Identity):
                def __init__(self, *a, **k):
                    super().__init__()
            tl.DropPath = DropPath
            timm.models = tm
            tm.layers = tl
        if "torchvision" not in sys.modules:
            tv = _stub("torchvision")
            tv.__version__ = "0.0.0"
            tt = _stub("torchvision.transforms")
            for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "InterpolationMode"):
                setattr(tt, n, type(n, (), {"BICUBIC": 3}))
            to = _stub("torchvision.ops")
            tb = _stub("torchvision.ops.boxes")
            tb.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            tv.transforms, tv.ops, to.boxes = tt, to, tb
        if "ftfy" not in sys.modules:
            ft = _stub("ftfy")
            ft.fix_text = lambda s: s
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        _loaded = True
    import model.LaviLa as LaviLa
    import model.tfm_decoder as tfm_decoder
    import model.loss as loss
    import model.box_utils as box_utils
    import model.metric as metric
    import model.openai_model as openai_model
    import utils.box_ops as box_ops
    return types.SimpleNamespace(LaviLa=LaviLa, tfm_decoder=tfm_decoder, loss=loss, box_utils=box_utils,
                                 metric=metric, openai_model=openai_model, box_ops=box_ops)
