"""BASELINE.json configurations at full width on one GPU -> profiles/<tag>_config_sweep.json (copy from gpurun_out/).
C1 B in {2, 32}; C2 B in {8, 16, 32, 64}; C4 B in {2, 4, 8}; C5 (EgoMCQ forward) q in {1, 8, 32}.  Pipelined train step on a resident
synthetic batch, per-step device events: mean / std / p50 over `steps` timed steps after `warmup`."""
import json, os, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C1, C2, C4
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep, mcq_forward
dev = torch.device("cuda", 0)
torch.set_num_threads(16)
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/config_sweep.json"
only = set(sys.argv[2:])


def build(cfg):
    bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
    dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
    return bb, dec


def timed(f, steps, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t = time.perf_counter(); ev[0].record()
    for i in range(steps):
        out = f(); ev[i + 1].record()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / steps
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return wall * 1e3, per, out


rows = []
for name, cfg, Bs, steps in [("C1: 4-frame 224p, nq=4", C1, [2, 32], 20), ("C2: 16-frame 224p, nq=12", C2, [8, 16, 32, 64], 20), ("C4: 32-frame 336p, nq=12", C4, [2, 4, 8], 10)]:
    if only and name[:2] not in only:
        continue
    bb, dec = build(cfg)
    for B in Bs:
        batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
        ts = TrainStep(cfg, bb, dec)
        torch.cuda.reset_peak_memory_stats()
        ms, per, out = timed(lambda: ts.step(batch, next_batch=batch), steps)
        rec = {"config": name, "workload": "train step (pipelined)", "clips_per_step": B, "steps": steps, "ms_per_step": round(ms, 2), "clips_per_s": round(B / ms * 1e3, 1),
               "ms_std": round(statistics.pstdev(per[1:]), 3), "ms_p50": round(statistics.median(per), 2), "loss": round(float(out["total_loss"]), 4),
               "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        rows.append(rec); print(rec, flush=True)
        del ts, batch
    if cfg is C2 and (not only or "C5" in only):
        dec.eval()
        for q in (1, 8, 32):
            m = synth.make_mcq_item(cfg, q, seed=2)
            v, t = m["video"].to(dev), m["text"].to(dev)
            ms, per, sc = timed(lambda: mcq_forward(bb, dec, v, t, cfg), 10)
            rec = {"config": "C5: EgoMCQ forward, 16-frame 224p", "workload": "forward only, q items of 5 clips + 1 query", "items": q, "clips_per_step": 5 * q, "steps": 10,
                   "ms_per_step": round(ms, 2), "clips_per_s": round(5 * q / ms * 1e3, 1), "ms_std": round(statistics.pstdev(per), 3), "ms_p50": round(statistics.median(per), 2)}
            rows.append(rec); print(rec, flush=True)
            del v, t
    del bb, dec; torch.cuda.empty_cache()
json.dump({"what": "one MI355X, full-width TimeSformer-L, synthetic resident batch; scripts/config_sweep.py", "rows": rows}, open(out_path, "w"), indent=1)
