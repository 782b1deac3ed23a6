"""How long does the kernel that FOLLOWS a persistent GEMM take (add+LayerNorm on the GEMM's output), 8-wave vs 4-wave GEMM?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 32 * 4097
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, 4096, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(1024, 4096, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(1024, device="cuda", generator=g)
x = torch.randn(M, 1024, device="cuda", generator=g)
gam, bet = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
for rnd in range(3):
    for mode in (3, 5):
        ops.set_tuning("gemm256", mode)
        ts = []
        for it in range(12):
            e0, e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record(); y = ops.gemm(a, w, bias); e1.record(); z = ops.add_layernorm(x, y, gam, bet, 1e-6); e2.record()
            torch.cuda.synchronize()
            if it >= 2: ts.append((e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
        ts.sort(key=lambda t: t[1])
        print("mode %d: gemm %.1f us, add_ln after it %.1f us (median; min %.1f max %.1f)" % (mode, sorted(t[0] for t in ts)[len(ts) // 2], ts[len(ts) // 2][1], ts[0][1], ts[-1][1]), flush=True)
ops.set_tuning("gemm256", ops.GEMM256_DEFAULT)
# add_ln alone, back to back
ts = []
for it in range(12):
    e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True)
    e1.record(); z = ops.add_layernorm(x, y, gam, bet, 1e-6); e2.record(); torch.cuda.synchronize()
    ts.append(e1.elapsed_time(e2) * 1e3)
print("add_ln alone: %.1f us" % sorted(ts)[6])
