"""Does a persistent GEMM of the pipelined step start on all CUs at once?  Timeline stamps (hh_set_tuning("gemm256_debug_ts", 1)) of the
last persistent GEMM launched in a pipelined step (the prefetched encoder's last fc2, which runs beside the decoder kernels of the
current step) vs the same GEMM in a towers-only loop: per workgroup, start of its first tile relative to the earliest workgroup."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib, synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
dev = torch.device("cuda", 0)
cfg, B = C2, 32
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
ts = TrainStep(cfg, bb, dec)
def timeline():
    buf = np.zeros((256, 8, 7), dtype=np.uint64)
    _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "timeline")
    return buf.astype(np.int64)
def report(tag, t):
    start = t[:, 0, 0] / 100.0                      # us (100 MHz counter), first tile of each workgroup
    end = t[:, :, 4].max(axis=1) / 100.0            # last recorded tile's end (<= 8 tiles per workgroup: fc2 has 8)
    s0 = start - start.min()
    late = np.nonzero(s0 > 10.0)[0]
    print("   late workgroups (> 10 us): %d, indices %s ..." % (len(late), late[:40].tolist()))
    print("%-22s first-tile start spread: p50 %.1f p90 %.1f p99 %.1f max %.1f us | kernel span (first start -> last recorded end) %.1f us | per-workgroup busy p50 %.1f max %.1f us" % (
        tag, np.percentile(s0, 50), np.percentile(s0, 90), np.percentile(s0, 99), s0.max(), end.max() - start.min(), np.median(end - start), (end - start).max()), flush=True)
for _ in range(3): ts.step(batch, next_batch=batch)
torch.cuda.synchronize()
def busy(tag, t, tiles):
    t = t[:, :tiles]
    start = t[:, 0, 0] / 100.0
    end = t[:, tiles - 1, 4] / 100.0
    b = end - start
    print("%-34s span %.1f us | per-workgroup busy over its first %d tiles: p10 %.1f p50 %.1f p90 %.1f max %.1f us | start spread p90 %.1f max %.1f" % (
        tag, end.max() - start.min(), tiles, np.percentile(b, 10), np.median(b), np.percentile(b, 90), b.max(), np.percentile(start - start.min(), 90), (start - start.min()).max()), flush=True)
# persistent launches of a step in order: per vision block qkv_t, proj_t, qkv_s, proj_s, fc1, fc2 (+ text tower / decoder launches beside them)
for nth in (26, 28, 30, 32, 34, 36, 38):
    for mode, run in (("pipelined", lambda: ts.step(batch, next_batch=batch)), ("towers only", lambda: ts.encode(batch["video"], batch["text"]))):
        torch.cuda.synchronize()
        ops.set_tuning("gemm256_debug_ts", nth)
        run(); torch.cuda.synchronize()
        ops.set_tuning("gemm256_debug_ts", 0)
        busy("launch %3d %s" % (nth, mode), timeline(), 8)
