"""Does a start skew keep the epilogues of the LayerNorm-fold producer GEMM apart?  Per-tile epilogue / main-loop times (debug stamps) of the
proj GEMM with residual write-back at the headline shape, for several skews."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 4097        # config 2: 8 whole rounds for N = 1024 (100384: 6.125 rounds)
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("proj", 1024, 1024), ("fc2", 1024, 4096)]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    x = torch.randn(M, N, device="cuda", generator=g)
    fn = lambda: ops.gemm(a, w, bias, z=(x, 1e-6, False, True))
    for sk, ph in [(0, 4), (24, 4), (48, 4), (24, 8), (96, 2), (160 if K > 2000 else 64, 4)]:
        ops.set_tuning("gemm_ln_pskew", sk); ops.set_tuning("gemm_ln_phases", ph)
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
        t = buf.astype(np.int64)
        first = t[:, 0, 0].min()
        ntile = (t[:, :, 4] > first).sum(1)
        epi = (t[:, :, 4] - t[:, :, 2]) / 100.0
        loop = (t[:, :, 2] - t[:, :, 1]) / 100.0
        end = np.array([t[i, ntile[i] - 1, 4] - first for i in range(256)]) / 100.0
        print("%-4s skew %3d x %d: %7.1f us/launch; tiles per workgroup %s; epilogue us by tile index %s; main loop %s; workgroup end p10 %.0f p50 %.0f max %.0f us; first-tile start spread %.1f" % (
            name, sk, ph, us, np.bincount(ntile).tolist(), " ".join("%.1f" % epi[ntile > j, j].mean() for j in range(8) if (ntile > j).any()),
            " ".join("%.1f" % loop[ntile > j, j].mean() for j in range(8) if (ntile > j).any()), np.percentile(end, 10), np.percentile(end, 50), end.max(), (t[:, 0, 1].max() - t[:, 0, 1].min()) / 100.0), flush=True)
ops.set_tuning("gemm_ln_pskew", 0)
