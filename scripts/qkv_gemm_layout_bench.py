"""QKV projection at the headline shape (M = 32 x 4097, N = 3072, K = 1024): row-major vs column-blocked (head-major planes) output."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M, N, K = 32 * 4097, 3072, 1024
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
b = torch.randn(N, device="cuda", generator=g)
def t(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rep in range(3):
    r = t(lambda: ops.gemm(a, w, b, colscale=0.125, colscale_cols=1024))
    c = t(lambda: ops.gemm(a, w, b, colscale=0.125, colscale_cols=1024, col_blocked=True))
    print("row-major %7.1f us (%6.1f TFLOP/s)   column-blocked %7.1f us (%6.1f TFLOP/s)" % (r, 2.0 * M * N * K / r / 1e6, c, 2.0 * M * N * K / c / 1e6))
