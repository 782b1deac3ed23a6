"""Text-tower GEMMs with N = 768 (M = 12320 -> 144 tiles of 256x256): 128x128 kernel (default below 192 tiles) vs the persistent 4-wave
256x256 kernel ("gemm256_min_tiles" = 128), alone and as CU-time (duration x workgroups' CUs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 12320
g = torch.Generator(device="cuda").manual_seed(0)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, N, K, kw in [("proj", 768, 768, {}), ("fc2", 768, 3072, {}), ("qkv", 2304, 768, dict(colscale=0.125, colscale_cols=768)), ("fc1", 3072, 768, dict(act=ops.ACT_QUICKGELU))]:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    ops.set_tuning("gemm256_min_tiles", 192); ref = ops.gemm(a, w, bias, **kw); t0 = t(lambda: ops.gemm(a, w, bias, **kw))
    ops.set_tuning("gemm256_min_tiles", 128); got = ops.gemm(a, w, bias, **kw); t1 = t(lambda: ops.gemm(a, w, bias, **kw))
    ops.set_tuning("gemm256_min_tiles", 192)
    print("%-5s N=%4d K=%4d: min_tiles 192: %6.1f us | min_tiles 128: %6.1f us | max |diff| %.3g" % (name, N, K, t0, t1, float((ref.float() - got.float()).abs().max())), flush=True)
