"""Achievable HBM bandwidth on this box for plain streaming kernels (ceiling for the LayerNorm / attention kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M, D = 32 * 4097, 1024
x = torch.randn(M, D, device="cuda"); y = torch.empty_like(x); xb = x.to(torch.bfloat16); yb = torch.empty_like(xb)
gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
for name, f, by in [("fp32 copy_ (r+w)", lambda: y.copy_(x), 8.0 * M * D), ("bf16 copy_ (r+w)", lambda: yb.copy_(xb), 4.0 * M * D),
                    ("fp32 sum (read only)", lambda: x.sum(), 4.0 * M * D), ("fp32 fill (write only)", lambda: y.fill_(1.0), 4.0 * M * D),
                    ("hh layernorm f32->bf16", lambda: ops.layernorm(x, gam, bet, 1e-6), 6.0 * M * D),
                    ("hh add_layernorm write_x", lambda: ops.add_layernorm(x, xb, gam, bet, 1e-6, write_x=True), 12.0 * M * D),
                    ("hh add_layernorm no write", lambda: ops.add_layernorm(x, xb, gam, bet, 1e-6, write_x=False), 8.0 * M * D)]:
    s = t(f); print(f"{name:28s} {s*1e6:8.1f} us  {by/s/1e12:6.2f} TB/s", flush=True)
