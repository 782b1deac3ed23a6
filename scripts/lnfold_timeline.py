"""Per-tile timeline (hh_set_tuning("gemm256_debug_ts", 1)) of the producer side of the LayerNorm fold on the persistent 4-wave GEMM: main-loop
time, shader clock and the time between two main loops, for the plain GEMM and the fold epilogue (no write-back / write-back; 224-row tiles off / on), at the proj and fc2 shapes of the headline configuration."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 3137
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("proj", 1024, 1024), ("fc2", 1024, 4096)]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    x = torch.randn(M, N, device="cuda", generator=g)
    for tag, fn, t224 in [("plain", lambda: ops.gemm(a, w, bias), 0), ("plain 224", lambda: ops.gemm(a, w, bias), 1),
                          ("z", lambda: ops.gemm(a, w, bias, z=(x, 1e-6, False)), 0), ("z 224", lambda: ops.gemm(a, w, bias, z=(x, 1e-6, False)), 1),
                          ("z+update", lambda: ops.gemm(a, w, bias, z=(x, 1e-6, False, True)), 0), ("z+update 224", lambda: ops.gemm(a, w, bias, z=(x, 1e-6, False, True)), 1)]:
        ops.set_tuning("gemm_tile224", t224)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        ops.set_tuning("gemm256_debug_ts", 1)
        fn(); torch.cuda.synchronize()
        ops.set_tuning("gemm256_debug_ts", 0)
        buf = np.zeros((256, 8, 7), dtype=np.uint64)
        _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "timeline")
        nwg = min(256, (M // 256) * (N // 256)); ntl = max(1, min(5, (M // 256) * (N // 256) // 256))
        t = buf.astype(np.int64)[:nwg, :ntl]
        nk = K // 64
        loop_us = (t[:, :, 2] - t[:, :, 1]) / 100.0
        mhz = (t[:, :, 6] - t[:, :, 5]) / np.maximum(loop_us, 1e-9)
        epi_us = (t[:, :, 4] - t[:, :, 2]) / 100.0
        first = t[:, 0, 0].min()
        print("%-4s %-13s %7.1f us/launch: main loop %.2f us (tile 0: %.2f) clock %.0f MHz, epilogue %.2f us (p10 %.2f p90 %.2f; tile 0 %.2f, tile 3 %.2f); start spread %.1f us; end of tile-1 main loop: p10 %.1f p90 %.1f us" % (
            name, tag, us, loop_us[:, -1].mean(), loop_us[:, 0].mean(), mhz[:, -1].mean(), epi_us[:, :5].mean(), np.percentile(epi_us[:, :5], 10), np.percentile(epi_us[:, :5], 90),
            epi_us[:, 0].mean(), epi_us[:, -1].mean(), (t[:, 0, 0].max() - first) / 100.0, np.percentile(t[:, -1, 2] - first, 10) / 100.0, np.percentile(t[:, -1, 2] - first, 90) / 100.0), flush=True)
ops.set_tuning("gemm_tile224", 0)
