"""Micro-benchmark of the decoder query-side kernels at the headline shapes (B = 32 clips x 13 queries = 416 rows)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops

dev = "cuda"
R, C, F = 416, 512, 2048
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
shapes = [("NT qk  416x1024x512", ops.NT, (R, C), (2 * C, C)), ("NT o   416x512x512", ops.NT, (R, C), (C, C)), ("NT ff1 416x2048x512", ops.NT, (R, C), (F, C)),
          ("NT ff2 416x512x2048", ops.NT, (R, F), (C, F)), ("NT cls 416x22048x512", ops.NT, (R, C), (22048, C)),
          ("NN dz  416x2048x512", ops.NN, (R, C), (C, F)), ("NN de  416x512x2048", ops.NN, (R, F), (F, C)), ("NN do  416x512x512", ops.NN, (R, C), (C, C)),
          ("TN dW2 512x2048x416", ops.TN, (R, C), (R, F)), ("TN dW1 2048x512x416", ops.TN, (R, F), (R, C)), ("TN dWo 512x512x416", ops.TN, (R, C), (R, C)),
          ("TN box 512x512x6656", ops.TN, (6656, C), (6656, C)), ("NT box 6656x512x512", ops.NT, (6656, C), (C, C))]
for name, mode, sa, sb in shapes:
    a, b = rn(*sa), rn(*sb)
    us = t(lambda: ops.qgemm(a, b, mode))
    M = sa[0] if mode != ops.TN else sa[1]
    N = sb[0] if mode == ops.NT else sb[1]
    K = sa[1] if mode != ops.TN else sa[0]
    print(f"{name:24s} {us:8.1f} us  {2.0 * M * N * K * 3 / us / 1e6:8.1f} TFLOP/s (bf16 MFMA work, x3)")
qkv = rn(R, 3 * C)
print("self-attn fwd %.1f us, bwd %.1f us" % (t(lambda: ops.qself_attn_fwd(qkv, 32, 13, 8)), t(lambda: ops.qself_attn_bwd(qkv, rn(R, C), 32, 13, 8))))
