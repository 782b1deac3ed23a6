import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    a = torch.randn(32, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    ref = (a.float() @ w.float().t())
    out = ops.gemm(a, w)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    for _ in range(5): ops.gemm(a, w)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(50): ops.gemm(a, w)
    e1.record(); torch.cuda.synchronize()
    print(f"M=32 N={N} K={K}: {e0.elapsed_time(e1)/50*1e3:6.1f} us/call (back-to-back)  rel err {err:.1e}")
