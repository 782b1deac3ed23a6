"""GEMM micro-benchmark over the hot-path shapes (M = B*4097 rows): interleaved A/B of kernel modes in ONE process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B = int(os.environ.get("B", 32))
MODES = [int(x) for x in os.environ.get("MODES", "0,2,3").split(",")]   # gemm256 mode: 0 = 128x128, 1/2 = 256x256 (no stagger / stagger), 3 = persistent
ROUNDS = int(os.environ.get("ROUNDS", 5))
M = B * 4097
shapes = [("qkv", 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), ("proj", 1024, 1024, dict()),
          ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, dict())]
g = torch.Generator(device="cuda").manual_seed(0)
tot = {m: [0.0, 0.0] for m in MODES}
for name, N, K, kw in shapes:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    kw = dict(kw)
    if kw.pop("resid", False):
        r = torch.randn(M, N, device="cuda", generator=g)
        f = lambda: ops.gemm(a, w, bias, out=r, resid=r)
    else:
        f = lambda: ops.gemm(a, w, bias, **kw)
    best = {m: [] for m in MODES}
    for rnd in range(ROUNDS + 1):
        for m in MODES:
            ops.set_tuning("gemm256", m); ops.set_tuning("gemm256_skew", 0)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            if rnd: best[m].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * M * N * K
    line = f"{name:5s} N={N:5d} K={K:5d} "
    for m in MODES:
        med = sorted(best[m])[len(best[m]) // 2]
        tot[m][0] += med; tot[m][1] += fl
        line += f" | mode{m}: {med*1e3:7.1f} us {fl/med/1e9:7.1f} TF (min {fl/max(best[m])/1e9:.0f} max {fl/min(best[m])/1e9:.0f})"
    print(line, flush=True)
print("sum over shapes: " + "  ".join(f"mode{m}: {tot[m][1]/tot[m][0]/1e9:.1f} TF/s" for m in MODES))
