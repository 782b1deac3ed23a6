"""Text-tower GEMM shapes (M = 5*B*77 = 12320 rows at B = 32, width 768): 128x128 kernel (mode 0) vs 256x256 kernels (2: one tile per
workgroup, 4: continuous persistent)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = int(os.environ.get("M", 12320))
g = torch.Generator(device="cuda").manual_seed(0)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, N, K, kw in [("qkv", 2304, 768, {}), ("proj+res", 768, 768, dict(resid=True)), ("fc1", 3072, 768, dict(act=ops.ACT_QUICKGELU)), ("fc2+res", 768, 3072, dict(resid=True))]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    kw = dict(kw)
    if kw.pop("resid", False):
        r = torch.randn(M, N, device="cuda", generator=g)
        f = lambda: ops.gemm(a, w, bias, out=r, resid=r)
    else:
        f = lambda: ops.gemm(a, w, bias, **kw)
    line = f"{name:9s} N={N:5d} K={K:5d}"
    for mode in (0, 2, 3):
        ops.set_tuning("gemm256", mode)
        ms = t(f)
        line += f" | mode{mode}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s"
    ops.set_tuning("gemm256", ops.GEMM256_DEFAULT)
    print(line, flush=True)
