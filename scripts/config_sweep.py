"""BASELINE.json configs at full width on one GPU: one-line throughput per config (C1 B=2, C2 B in {8,16,32,48}, C4 B in {2,4}, C5 MCQ q in {1,8})."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C1, C2, C4
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep, mcq_forward
dev = torch.device("cuda", 0)
torch.set_num_threads(16)
def build(cfg):
    bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
    dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
    return bb, dec
def timed(f, steps, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / steps, out
only = set(sys.argv[1:])                                                   # e.g. `config_sweep.py C4`: just that config
for name, cfg, Bs in [("C1 T=4 224p nq=4", C1, [2, 32]), ("C2 T=16 224p nq=12", C2, [8, 16, 32, 48]), ("C4 T=32 336p nq=12", C4, [2, 4])]:
    if only and name.split()[0] not in only:
        continue
    bb, dec = build(cfg)
    for B in Bs:
        batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
        ts = TrainStep(cfg, bb, dec)
        dt, out = timed(lambda: ts.step(batch, next_batch=batch), 4)
        print(f"{name:22s} train B={B:3d}: {dt*1e3:8.1f} ms/step {B/dt:8.1f} clips/s  loss {float(out['total_loss']):.4f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
        del ts
    if cfg is C2:
        dec.eval()
        for q in (1, 8):
            m = synth.make_mcq_item(cfg, q, seed=2)
            v, t = m["video"].to(dev), m["text"].to(dev)
            dt, sc = timed(lambda: mcq_forward(bb, dec, v, t, cfg), 4)
            print(f"C5 EgoMCQ fwd          q={q:2d} ({5*q} clips): {dt*1e3:8.1f} ms  {5*q/dt:8.1f} clips/s  argmax {sc.argmax(-1).tolist()[:4]}", flush=True)
    del bb, dec; torch.cuda.empty_cache()
