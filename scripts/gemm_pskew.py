"""Start skew of the persistent GEMM kernels (hh_set_tuning("gemm256_pskew", q): workgroup i of an XCD sleeps q * i * s_sleep(8) before its walk)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 32 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, kw in [("qkv", 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), ("proj", 1024, 1024, {}), ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, {})]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = {}
    for rnd in range(4):
        for mode in (3, 5):
            for q in (0, 1, 2, 3, 5, 8):
                ops.set_tuning("gemm256", mode); ops.set_tuning("gemm256_pskew", q)
                for _ in range(2): ops.gemm(a, w, bias, **kw)
                torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(5): ops.gemm(a, w, bias, **kw)
                e1.record(); torch.cuda.synchronize()
                if rnd: res.setdefault((mode, q), []).append(e0.elapsed_time(e1) / 5)
    tf = {m: 2.0 * M * N * K / sorted(v)[len(v) // 2] / 1e9 for m, v in res.items()}
    print(name, "  ".join("m%d q%d: %.0f" % (m, q, tf[(m, q)]) for (m, q) in sorted(tf)), flush=True)
ops.set_tuning("gemm256", 3); ops.set_tuning("gemm256_pskew", 0)
