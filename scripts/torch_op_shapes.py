"""Which stock torch matmuls / reductions still run in a step, with their shapes (torch.profiler, record_shapes)."""
import os, sys, torch
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
for _ in range(2): ts.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts.step(batch); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::addmm", "aten::mm", "aten::bmm", "aten::sum", "aten::linear", "aten::matmul", "aten::copy_", "aten::argmax")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:14]:
    print(f"{e.key:14s} calls {e.count:4d} device {e.device_time_total/1e3:8.3f} ms  shapes {str(e.input_shapes)[:110]}")
