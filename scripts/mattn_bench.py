"""Stand-alone timing of the memory-space cross-attention kernels (csrc/mattn.hip) at the benchmarked decoder shapes:
    python scripts/mattn_bench.py [B] [M] [iters]          (default 32 4096 20; config 4: 4 18432)
Prints the mean device time per launch of hh_mattn_fwd / hh_mattn_bwd / the batched d-memory GEMM and the HBM rate of the row stream
(2 KB per key and clip: mp + mem, bf16).  Under rocprofv3 (--pmc ...) pass iters = 2."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/scripts", 1)[0])
from helping_hand_for_egocentric_videos_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Q, H, C, L = 13, 8, 512, 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qt = torch.randn(B * Q, H * C, device=dev, generator=g) * 0.08
mp = torch.randn(B, M, C, device=dev, generator=g).to(torch.bfloat16)
mem = torch.randn(B, M, C, device=dev, generator=g).to(torch.bfloat16)
wv = torch.randn(C, C, device=dev, generator=g) * 0.05
bv = torch.randn(C, device=dev, generator=g) * 0.1
G = torch.randn(B * Q, C, device=dev, generator=g)
pdT = torch.empty((B, L * 128, M), dtype=torch.bfloat16, device=dev); dsT = torch.empty_like(pdT)
qt16 = torch.empty((B, L * 128, C), dtype=torch.bfloat16, device=dev); dp16 = torch.empty_like(qt16)
pooled, lse2, rsum = ops.mattn_fwd(qt, mp, mem, Q)
ca = ops.head_map_out(pooled, wv, bias=bv)
dpooled = ops.head_map_in(G, wv)
for l in range(L):
    ops.mattn_bwd(qt, dpooled, lse2, G, ca, bv, mp, mem, Q, pdT, dsT, qt16, dp16, l * 128)
def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
rows_bytes = 2.0 * B * M * C * 2
for p in (0.0, 0.1):
    tf = timeit(lambda: ops.mattn_fwd(qt, mp, mem, Q, p, 5), iters)
    tb = timeit(lambda: ops.mattn_bwd(qt, dpooled, lse2, G, ca, bv, mp, mem, Q, pdT, dsT, qt16, dp16, 0, p, 5), iters)
    print("B=%d M=%d p=%.1f slices=%d: fwd %.1f us (%.2f TB/s of rows), bwd %.1f us (%.2f TB/s)" % (B, M, p, ops.mattn_slices(B, M), tf, rows_bytes / tf / 1e6, tb, rows_bytes / tb / 1e6))
tg = timeit(lambda: ops.gemm_tn_batched2(pdT, dp16, dsT, qt16), max(2, iters // 4))
print("d-memory GEMM (batched, K = 2 x %d): %.1f us = %.0f TFLOP/s" % (L * 128, tg, 2.0 * B * M * C * 2 * L * 128 / tg / 1e6))
tp = timeit(lambda: ops.gemm_tn(dsT.view(B * L * 128, M), qt16.view(B * L * 128, C)), max(2, iters // 4))
print("d-pos GEMM (split-K over %d rows): %.1f us" % (B * L * 128, tp))
