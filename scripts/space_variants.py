"""Space attention at the headline shape (B = 32, T = 16, n = 256, 16 heads, head-major planes), one variant per process argument:
1 = joint-block kernel (default), 2 = progressive staging (3 segments)  (3 = round 4's persistent variant: removed in round 5).
Runs 12 calls; meant to sit under `rocprofv3 --pmc ... --kernel-trace` for the per-variant SQ / traffic counters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B, T, n, heads = 32, 16, 256, 16
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g); qkv[:, :D] *= 0.5
planes = qkv.to(torch.bfloat16).view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
for prog in (int(v) for v in sys.argv[1:]):
    ops.set_tuning("space_prog", prog)
    for _ in range(12):
        ops.divided_attention(planes, B, T, n, heads, "space")
torch.cuda.synchronize()
