"""One qkv / fc1 GEMM of the headline shape under hh_set_tuning("gemm256_group", G) -- run under rocprofv3 --pmc FETCH_SIZE by
scripts/gemm_group_fetch.sh to see what the XCD tile walk does to the L2-side fetch traffic (G m-tiles x 32/G n-tiles per round and XCD)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
G = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "qkv"
fold = len(sys.argv) > 3 and sys.argv[3] == "fold"          # consumer side of the LayerNorm fold (EPI 5 / 6: per-tile epilogue record by LDS-DMA)
M = 32 * 4097
N, K, kw = {"qkv": (3072, 1024, dict(colscale=0.125, colscale_cols=1024)), "fc1": (4096, 1024, dict(act=ops.ACT_QUICKGELU)),
            "proj": (1024, 1024, dict()), "fc2": (1024, 4096, dict())}[which]
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
if fold and which in ("proj", "fc2"):         # producer side: x <- x + A W^T + b in place, z and its row statistics
    xres = torch.randn(M, N, device="cuda", generator=g)
    kw["z"] = (xres, 1e-6, False, True)
elif fold:
    stats = torch.stack([torch.rand(M, device="cuda", generator=g) + 0.5, torch.randn(M, device="cuda", generator=g) * 0.1], 1).contiguous()
    kw["ln"] = (stats, torch.randn(N, device="cuda", generator=g))
    if which == "qkv":
        kw["col_blocked"] = True
ops.set_tuning("gemm256_group", G)
for _ in range(12): ops.gemm(a, w, bias, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): ops.gemm(a, w, bias, **kw)
e1.record(); torch.cuda.synchronize()
print("RESULT %s%s G=%d: %.1f us per call, %.1f TFLOP/s" % (which, " (fold consumer)" if fold else "", G, e0.elapsed_time(e1) * 100, 2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9))
