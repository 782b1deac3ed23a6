import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B, T, n, heads = 32, 16, 256, 16
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g)
qkv[:, :D] *= 0.5
qkv = qkv.to(torch.bfloat16)
out = torch.empty(B * N, D, dtype=torch.bfloat16, device="cuda")
from helping_hand_for_egocentric_videos_amd import _lib
import ctypes
L = _lib.lib()
def run(): _lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
for dbg, name in ((0, "full kernel (no cls partial)"), (1, "memory only (stage K/V, read Q, write O)"), (2, "compute only (no K/V staging)")):
    ops.set_tuning("space_debug", dbg)
    for _ in range(3): run()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print("%-45s %7.1f us" % (name, e0.elapsed_time(e1) / 10 * 1e3))
ops.set_tuning("space_debug", 0)
