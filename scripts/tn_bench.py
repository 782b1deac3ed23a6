"""Weight-gradient GEMM: TN kernel on the natural layout vs transposes + NT split-K kernel (decoder K/V in-projection shapes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
K = 32 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, M, N in [("dW_k (6 layers)", 3072, 512), ("dW_proj", 512, 1024)]:
    at = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16)
    bt = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16)
    fl = 2.0 * K * M * N
    old = t(lambda: ops.gemm(ops.transpose_bf16(at), ops.transpose_bf16(bt), splitk=max(1, min(64, 512 // ((M // 128) * (N // 128)), K // 512))))
    line = f"{name:16s} M={M} N={N} K={K}: transposes + NT split-K {old*1e3:7.1f} us ({fl/old/1e9:6.1f} TF/s)"
    for sp in (None, 8, 16, 32, 64):
        ms = t(lambda: ops.gemm_tn(at, bt, splits=sp))
        line += f" | TN splits={sp}: {ms*1e3:7.1f} us ({fl/ms/1e9:6.1f} TF/s)"
    print(line, flush=True)
