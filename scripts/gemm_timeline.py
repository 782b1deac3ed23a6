"""Per-tile timeline of the persistent GEMM kernels (hh_set_tuning("gemm256_debug_ts", 1)): shader-clock cycles per k-tile in the main
loop and the time between two main loops (epilogue + hand-over), for the default 8-wave kernel (mode 3) and the 4-wave one (mode 5)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
M = 32 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
modes = [3, 5]
dbgs = [0]
for name, N, K, kw in [("qkv", 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), ("proj", 1024, 1024, {}), ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, {})]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    for mode, dbg in [(3, 0)] + [(5, d) for d in dbgs]:
        ops.set_tuning("gemm256", mode)
        for _ in range(3): ops.gemm(a, w, bias, **kw)
        ops.set_tuning("gemm256_debug_ts", 1)
        ops.gemm(a, w, bias, **kw); torch.cuda.synchronize()
        ops.set_tuning("gemm256_debug_ts", 0)
        buf = np.zeros((256, 8, 7), dtype=np.uint64)
        _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "timeline")
        t = buf.astype(np.int64)
        tiles = min(8, (M // 256) * (N // 256) // 256)
        t = t[:, :tiles]
        nk = K // 64
        cyc = (t[:, :, 6] - t[:, :, 5]) / nk                              # shader cycles per k-tile in the main loop
        loop_us = (t[:, :, 2] - t[:, :, 1]) / 100.0
        mhz = (t[:, :, 6] - t[:, :, 5]) / np.maximum(loop_us, 1e-9)
        gap_us = (t[:, 1:, 1] - t[:, :-1, 2]) / 100.0                      # end of a main loop -> start of the next
        epi_us = (t[:, :, 4] - t[:, :, 3]).mean() / 100.0                  # the store part of the epilogue
        pre_us = (t[:, :, 3] - t[:, :, 2]).mean() / 100.0
        print("%-4s mode %d dbg %2d: cycles / k-tile %.0f (p10 %.0f p90 %.0f)  main loop %.2f us  clock %.0f MHz  between loops %.2f us (%.0f cycles; bias %.2f, stores %.2f us)" % (
            name, mode, dbg, cyc.mean(), np.percentile(cyc, 10), np.percentile(cyc, 90), loop_us.mean(), mhz.mean(), gap_us.mean(), gap_us.mean() * mhz.mean(), pre_us, epi_us), flush=True)
ops.set_tuning("gemm256", ops.GEMM256_DEFAULT)
