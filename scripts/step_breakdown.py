"""Same session: frozen towers only (TrainStep.encode in a loop) vs the full pipelined / un-pipelined step -> how much decoder-side
GPU work is NOT hidden behind the next batch's towers."""
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
ts = TrainStep(cfg, bb, dec)
def timed(f, n=8, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rnd in range(2):
    ts.text_on_side_stream = False
    e1 = timed(lambda: ts.encode(batch["video"], batch["text"]))
    p1 = timed(lambda: ts.step(batch, next_batch=batch))
    ts.text_on_side_stream = True
    print(f"text tower on the SAME stream: both towers {e1:7.2f} ms | pipelined step {p1:7.2f} ms", flush=True)
    e = timed(lambda: ts.encode(batch["video"], batch["text"]))
    v = timed(lambda: bb.visual.forward_features(batch["video"], out_dtype=torch.bfloat16))
    p = timed(lambda: ts.step(batch, next_batch=batch))
    u = timed(lambda: ts.step(batch))
    print(f"vision tower only {v:7.2f} ms | both towers {e:7.2f} ms | pipelined step {p:7.2f} ms | un-pipelined step {u:7.2f} ms | decoder side exposed in the pipelined step {p-e:6.2f} ms", flush=True)
