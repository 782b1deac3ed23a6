import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 32 * 4096                      # exact multiple of 256: no remainder kernel
g = torch.Generator(device="cuda").manual_seed(0)
for N in (1024, 3072):
    res = {}
    for K in (512, 1024, 2048, 4096):
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        for ns in (0, 1):
            ops.set_tuning("gemm256_debug_nostore", ns)
            for _ in range(3): ops.gemm(a, w)
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(8): ops.gemm(a, w)
            e1.record(); torch.cuda.synchronize()
            res[(K, ns)] = e0.elapsed_time(e1) / 8 * 1e3
    ops.set_tuning("gemm256_debug_nostore", 0)
    rounds = (M // 256) * (N // 256) / 256
    for ns in (0, 1):
        t = [res[(K, ns)] / rounds for K in (512, 1024, 2048, 4096)]
        slope = (t[3] - t[1]) / (64 - 16)
        print(f"N={N} nostore={ns}: us/block at K=512..4096: {[round(x,1) for x in t]}  -> {slope:.2f} us per k-tile, intercept {t[1]-16*slope:.1f} us")
