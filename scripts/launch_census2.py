"""Stock (non-libhh) device launches of one training step, attributed to the PHASE that issued them (towers / decoder forward / loss tail /
backward / optimizer) and, inside the forward, to the python call (record_function ranges pushed by a sys.setprofile hook on the package's
own functions).  Config 2 at B = 32 with a 2-block encoder (its launches are all libhh).  Prints counts per (phase, aten op, shapes)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device("cuda", 0); torch.set_num_threads(16)
cfg, B = C2.with_(depth=2), 32
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
ts = TrainStep(cfg, bb, dec)
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "helping_hand_for_egocentric_videos_amd")
stack = []
def hook(frame, event, arg):
    if event == "call" and frame.f_code.co_filename.startswith(PKG):
        r = record_function("PY:%s:%s:%d" % (os.path.basename(frame.f_code.co_filename), frame.f_code.co_name, frame.f_lineno)); r.__enter__(); stack.append((frame, r))
    elif event == "return" and stack and stack[-1][0] is frame:
        stack.pop()[1].__exit__(None, None, None)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    dec.train(); ts.arena.zero_grad(force=False)
    sys.setprofile(hook)
    with record_function("PHASE:forward"):
        out = ts.losses(batch)
    sys.setprofile(None)
    with record_function("PHASE:backward"):
        out["total_loss"].backward()
    with record_function("PHASE:optimizer"):
        ts.optimizer_step()
    torch.cuda.synchronize()
ev = prof.events()
ours = ("gemm", "attn", "ln_", "add_ln", "xattn", "adamw", "match", "lsap", "box_loss", "transpose_kernel", "cast_f32", "cast_bf16", "im2col", "embed_ln", "qgemm", "qself", "cls_combine", "colsum", "merge", "tail",
        "rownorm", "egonce", "masked_ce", "tv_accuracy", "ln_fold", "ln_rowstats")
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
ranges = [(e.time_range.start, e.time_range.end, e.name) for e in cpu if e.name.startswith(("PHASE:", "PY:"))]
by = collections.Counter(); total = 0; stock = 0
for e in cpu:
    if not e.kernels or any(c.kernels for c in e.cpu_children):
        continue
    n = sum(1 for k in e.kernels if not any(o in k.name for o in ours))
    total += len(e.kernels)
    if not n: continue
    stock += n
    t = e.time_range.start
    inside = [r for r in ranges if r[0] <= t <= r[1]]
    phase = next((r[2] for r in inside if r[2].startswith("PHASE:")), "PHASE:?")
    py = [r for r in inside if r[2].startswith("PY:")]
    where = min(py, key=lambda r: r[1] - r[0])[2][3:] if py else "-"
    by[(phase[6:], where, e.name, str(e.input_shapes)[:60])] += n
print("device launches: %d, stock (non-libhh): %d" % (total, stock))
for (ph, where, name, shp), n in sorted(by.items(), key=lambda x: (x[0][0], -x[1])):
    print("%3d  %-9s %-44s %-26s %s" % (n, ph, where[:44], name[:26], shp))
