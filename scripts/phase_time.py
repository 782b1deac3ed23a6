"""Un-pipelined step split into phases: GPU time (events) and host issue time of each."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
dev = torch.device("cuda", 0)
cfg, B = C2, int(os.environ.get("B", 32))
torch.set_num_threads(16)
backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
decoder = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1000).items()}
ts = TrainStep(cfg, backbone, decoder)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e, time.perf_counter()))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        mark(name + ":begin"); r = f(*a, **k); mark(name + ":end"); return r
    setattr(obj, attr, g)
wrap(ts, "encode", "encode")
dec_fwd = decoder.forward
def dec_wrapped(*a, **k):
    mark("decoder_fwd:begin"); r = dec_fwd(*a, **k); mark("decoder_fwd:end"); return r
decoder.forward = dec_wrapped
for it in range(4):
    marks.clear()
    torch.cuda.synchronize()
    mark("step:begin")
    decoder.train(); ts.arena.zero_grad()
    out = ts.losses(batch, None)
    mark("losses:end")
    out["total_loss"].backward()
    mark("backward:end")
    ts.optimizer_step()
    mark("opt:end")
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
print("phase                      gpu ms (since prev)   host ms (since prev)")
for (n0, e0, h0), (n1, e1, h1) in zip(marks[:-1], marks[1:]):
    print(f"{n0:>18s} -> {n1:<18s} {e0.elapsed_time(e1):8.2f}   {1e3*(h1-h0):8.2f}")
print(f"total gpu {marks[0][1].elapsed_time(marks[-1][1]):.2f} ms; host issue {1e3*(marks[-1][2]-marks[0][2]):.2f} ms; host until sync {1e3*(t_sync-marks[0][2]):.2f} ms")
