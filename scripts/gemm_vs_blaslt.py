"""Attainable-GEMM reference on this box: torch.matmul / F.linear (hipBLASLt) against hh_gemm_bf16 (persistent 256x256) on the
four encoder shapes, same inputs, interleaved rounds.  Diagnostic only -- the product path never calls hipBLASLt here."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B = int(os.environ.get("B", 32)); M = B * 4097
if os.environ.get("PSKEW"): ops.set_tuning("gemm256_pskew", int(os.environ["PSKEW"]))
SCALE = float(os.environ.get("SCALE", 1.0))
g = torch.Generator(device="cuda").manual_seed(0)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, K in [("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    a = (torch.randn(M, K, device="cuda", generator=g) * SCALE).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    bias32 = bias.float()
    res = {"hh": [], "lt": [], "lt_nobias": []}
    for r in range(6):
        x = t(lambda: ops.gemm(a, w, bias32)); y = t(lambda: torch.nn.functional.linear(a, w, bias)); z = t(lambda: torch.matmul(a, w.t()))
        if r: res["hh"].append(x); res["lt"].append(y); res["lt_nobias"].append(z)
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K}: " + "  ".join(f"{k} {fl/sorted(v)[len(v)//2]/1e9:7.1f} TF/s" for k, v in res.items()), flush=True)
