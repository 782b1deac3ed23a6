"""Per-tile timeline of the persistent 256x256 GEMM (debug stamps of wave 0): where the time between main loops goes."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
M = 32 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, kw in [("qkv", 3072, 1024, {}), ("proj", 1024, 1024, {}), ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, {})]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    for _ in range(3): ops.gemm(a, w, bias, **kw)
    ops.set_tuning("gemm256_debug_ts", 1)
    ops.gemm(a, w, bias, **kw); torch.cuda.synchronize()
    ops.set_tuning("gemm256_debug_ts", 0)
    buf = np.zeros((256, 8, 7), dtype=np.uint64)
    _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "timeline")
    t = buf.astype(np.int64)
    ntile = min(8, (M // 256) * (N // 256) // 256)
    t = t[:, :ntile]
    us = lambda x: x / 100.0
    wait0 = us(t[:, :, 1] - t[:, :, 0]); loop = us(t[:, :, 2] - t[:, :, 1]); pro = us(t[:, :, 3] - t[:, :, 2]); st = us(t[:, :, 4] - t[:, :, 3])
    gap = us(t[:, 1:, 0] - t[:, :-1, 4]) if ntile > 1 else np.zeros((1, 1))
    per_tile = us(t[:, 1:, 0] - t[:, :-1, 0]) if ntile > 1 else np.zeros((1, 1))
    print(f"{name:5s} N={N} K={K}: tiles/block {ntile}  per-tile {per_tile.mean():6.2f} us | wait k-tile0 {wait0[:,1:].mean():5.2f} (first tile {wait0[:,0].mean():5.2f}) "
          f"main loop {loop.mean():6.2f} ({loop.mean()/(K/64):.3f}/k-tile) | bias+next prologue {pro.mean():5.2f} | epilogue math+stores {st.mean():5.2f} | loop-back {gap.mean():5.2f}", flush=True)
    mhz = (t[:, :, 6] - t[:, :, 5]) / np.maximum(us(t[:, :, 2] - t[:, :, 1]), 1e-9)
    print(f"       s_memtime ticks per us during the main loop: mean {mhz.mean():.1f} min {mhz.min():.1f} max {mhz.max():.1f}")
    span = us(t[:, -1, 4].max() - t[:, 0, 0].min()); print(f"       kernel span (first stamp -> last stamp of 8 tiles) {span:.1f} us; block start spread {us(t[:,0,0].max()-t[:,0,0].min()):.1f} us")
