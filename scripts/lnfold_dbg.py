"""Debug helper: one producer-side LayerNorm-fold GEMM (hh_gemm_bf16 with z_out) per process: python lnfold_dbg.py M N K keep_c [w4]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M, N, K, keep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
if len(sys.argv) > 5:
    ops.set_tuning("gemm_ln_w4", int(sys.argv[5]))
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
x = torch.randn(M, N, device="cuda", generator=g) * 2 + 0.3
torch.cuda.synchronize()
c, z, st = ops.gemm(a, w, bias, z=(x, 1e-6, keep))
torch.cuda.synchronize()
v = a.float() @ w.float().t() + bias
zr = x + v
print("M=%d N=%d K=%d keep=%d: z err %.3e  c err %s  rstd err %.3e" % (M, N, K, keep, (z.float() - zr).abs().max().item() / zr.abs().max().item(),
      "-" if c is None else "%.3e" % ((c.float() - v).abs().max().item() / v.abs().max().item()),
      ((st[:, 0] - (zr.var(1, unbiased=False) + 1e-6).rsqrt()) * zr.std(1)).abs().max().item()))
