"""Vision tower alone (config 2, B = 32) for a per-launch kernel trace: run under `rocprofv3 --kernel-trace --output-format csv`;
tower_trace_report.py prints the launch sequence of one block from the trace.  argv[1] = 0: no LayerNorm fold; further NAME=V: tuning knobs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import C2, synth, ops
from helping_hand_for_egocentric_videos_amd.model import LaviLa
if len(sys.argv) > 1 and sys.argv[1] == "0":
    LaviLa.LN_FOLD = False
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ops.set_tuning(k, int(v))
cfg = C2
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0))
video = synth.make_batch(cfg, 32, seed=1)["video"].cuda()
with torch.no_grad():
    for _ in range(6):
        bb.visual.forward_features(video, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
with torch.no_grad():
    e0.record()
    for _ in range(5):
        bb.visual.forward_features(video, out_dtype=torch.bfloat16)
    e1.record()
torch.cuda.synchronize()
print("tower pass: %.2f ms (event bracket, 5 passes)" % (e0.elapsed_time(e1) / 5), flush=True)
