"""Does a high-priority main (decoder) stream shorten the pipelined step?  (the frozen towers run on an ordinary side stream)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
dev = torch.device("cuda", 0); torch.set_num_threads(16)
cfg, B = C2, 32
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
def timed(ts, n=8, w=3):
    for _ in range(w): ts.step(batch, next_batch=batch)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): ts.step(batch, next_batch=batch)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rnd in range(2):
    ts = TrainStep(cfg, bb, dec)
    a = timed(ts)
    hp = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(hp):
        ts2 = TrainStep(cfg, bb, dec)
        b = timed(ts2)
    print(f"default main stream {a:7.2f} ms | high-priority main stream {b:7.2f} ms", flush=True)
