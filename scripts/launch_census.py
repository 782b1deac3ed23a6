"""Census of the stock (non-libhh) device launches of one training step: which aten op, which shapes, called from where.
torch.profiler with stacks; the encoder is cut to 2 blocks (its launches are all libhh) so that the decoder / loss tail stands out."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0); torch.set_num_threads(16)
cfg, B = C2.with_(depth=2), 32
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
ts = TrainStep(cfg, bb, dec)
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.step(batch); torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
ours = ("gemm", "attn", "ln_", "add_ln", "xattn", "adamw", "match", "lsap", "box_loss", "transpose_kernel", "cast_f32", "im2col", "embed_ln", "qgemm", "qself", "cls_combine", "colsum", "merge", "tail",
        "rownorm", "egonce", "masked_ce", "tv_accuracy", "mattn", "text_flags", "box_tail")
stock = [k for k in kern if not any(o in k.name for o in ours)]
print("device launches in the step: %d, stock (non-libhh): %d" % (len(kern), len(stock)))
# attribute every stock kernel to the innermost aten op (CPU event) that launched it and to the first repo frame of its stack
by = collections.Counter()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
for e in cpu:
    n = sum(1 for k in e.kernels if not any(o in k.name for o in ours))
    if not n or any(c.kernels for c in e.cpu_children):
        continue
    # python stacks are not recorded for ops issued by the autograd engine: name the autograd node (or "forward") from the CPU parent chain
    where, par = "forward", e.cpu_parent
    while par is not None:
        if par.name.startswith("autograd::engine::evaluate_function: "):
            where = par.name.split(": ", 1)[1]
            break
        par = par.cpu_parent
    py = next((s for s in (e.stack or []) if REPO in s and "scripts/" not in s), "")
    if py:
        where += " @ " + py.replace(REPO + "/", "").split(": ")[0]
    by[(e.name, str(e.input_shapes)[:70], where[:90])] += n
for (name, shp, where), n in by.most_common(120):
    print("%4d  %-28s %-70s %s" % (n, name, shp, where))
