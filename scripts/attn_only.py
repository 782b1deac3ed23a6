import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B, T, n, heads = int(os.environ.get("B", 32)), int(os.environ.get("T", 16)), int(os.environ.get("NP", 256)), 16
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g)
qkv[:, :D] *= 0.5
qkv = qkv.to(torch.bfloat16)
reps = int(os.environ.get("REPS", 3))
for mode in ("space", "time"):
    for fold in (True, False):
        for _ in range(2): ops.divided_attention(qkv, B, T, n, heads, mode, fold_cls=fold)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps): ops.divided_attention(qkv, B, T, n, heads, mode, fold_cls=fold)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{mode:5s} fold_cls={fold}: {ms*1e3:7.1f} us  {8.0*B*N*D/ms/1e6:7.1f} GB/s algorithmic", flush=True)
