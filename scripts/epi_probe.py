import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M, N, K = 32 * 4097, 1024, 1024
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
r = torch.randn(M, N, device="cuda", generator=g)
r2 = torch.randn(M, N, device="cuda", generator=g)
o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
variants = {"bf16 out": lambda: ops.gemm(a, w, bias, out=o16), "fp32 out": lambda: ops.gemm(a, w, bias, out=r2),
            "fp32 out + resid in place": lambda: ops.gemm(a, w, bias, out=r, resid=r),
            "fp32 out + resid other buf": lambda: ops.gemm(a, w, bias, out=r2, resid=r),
            "bf16 out + resid": lambda: ops.gemm(a, w, bias, out=o16, resid=r)}
for rnd in range(3):
    for name, f in variants.items():
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        if rnd == 2: print(f"{name:28s} {e0.elapsed_time(e1)/5*1e3:7.1f} us  {2.0*M*N*K/(e0.elapsed_time(e1)/5)/1e9:7.1f} TF/s")
