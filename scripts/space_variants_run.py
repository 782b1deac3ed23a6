"""Runs hh_space_attn_fwd (head-major planes, folded CLS partial) for every hh_set_tuning("space_mfma32") value: the workload of
scripts/space_variants_pmc.sh (rocprofv3 --pmc SQ counters -> MFMA / VALU busy share per space-attention kernel)."""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
B, T, n, heads = (int(v) for v in (sys.argv[1:4] + ["16"])) if len(sys.argv) > 3 else (32, 16, 256, 16)
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g)
qkv[:, :D] *= 0.5
planes = qkv.to(torch.bfloat16).view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
out = torch.zeros(B * N, D, dtype=torch.bfloat16, device="cuda")
part = torch.zeros(B * heads * T * 68, device="cuda")
L = _lib.lib()
for mode in (0, 1, 2):
    ops.set_tuning("space_mfma32", mode)
    for _ in range(int(os.environ.get("REPS", 6))):
        _lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(planes.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(part.data_ptr()), B, T, n, heads,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "hh_space_attn_fwd")
    torch.cuda.synchronize()
