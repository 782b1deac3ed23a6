"""Producer side of the LayerNorm fold at the headline shape (M = 32 * 4097, N = K = 1024): us per call of the proj GEMM with the residual add + z +
statistics in its epilogue, against the plain proj GEMM and the stand-alone add+LayerNorm pass it replaces; start-skew sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M, N, K = 32 * 4097, 1024, 1024
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
x = torch.randn(M, N, device="cuda", generator=g)
gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB: flush the Infinity Cache between variants
def t(fn, reps=20):
    for _ in range(3): fn()
    big.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
plain = t(lambda: ops.gemm(a, w, bias))
tb = ops.gemm(a, w, bias)
ln = t(lambda: ops.add_layernorm(x, tb, gam, bet, 1e-6, write_x=False))
print("plain proj GEMM %.1f us   add+LayerNorm pass %.1f us   sum %.1f us" % (plain, ln, plain + ln))
for keep in (False, True):
    for sk, ph in ((0, 0), (8, 2), (16, 2), (32, 2), (8, 4), (16, 4), (24, 4), (8, 8), (12, 8), (2, 0), (4, 0)):
        ops.set_tuning("gemm_ln_pskew", sk); ops.set_tuning("gemm_ln_phases", ph)
        v = t(lambda: ops.gemm(a, w, bias, z=(x, 1e-6, keep)))
        print("fold producer keep_c=%d  skew %2d x %d phases: %.1f us" % (keep, sk, ph, v))
ops.set_tuning("gemm_ln_pskew", 0)
