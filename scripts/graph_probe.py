"""Frozen vision tower, eager launches vs one captured HIP graph replay (same stream, static input buffer)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import C2, synth
from helping_hand_for_egocentric_videos_amd.model import LaviLa
cfg = C2
dev = torch.device("cuda", 0)
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
B = 32
video = synth.make_batch(cfg, B, seed=1)["video"].to(dev)
def fwd():
    with torch.no_grad():
        return bb.visual.forward_features(video, out_dtype=torch.bfloat16)[1]
def timed(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
ref = fwd().clone()
print("eager        %.2f ms" % timed(fwd))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
torch.cuda.synchronize()
print("graph replay %.2f ms   equal to eager: %s" % (timed(g.replay), torch.equal(out, ref)))
print("eager again  %.2f ms" % timed(fwd))
