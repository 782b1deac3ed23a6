"""Space attention at the headline shape: full kernel / memory only / compute only, joint-block kernel vs 16-query kernel."""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops, _lib
B, T, n, heads = 32, 16, 256, 16
if len(sys.argv) > 1:
    B, T, n = (int(v) for v in sys.argv[1:4])
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g)
qkv[:, :D] *= 0.5
qkv = qkv.to(torch.bfloat16)
out = torch.zeros(B * N, D, dtype=torch.bfloat16, device="cuda")          # (zeroed: the CLS rows, row b*N, are written by hh_cls_combine, not by this kernel)
patch = (torch.arange(B * N, device="cuda") % N) != 0                           # rows the space kernel writes
planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()          # head-major planes [3*heads, B*N, 64]
L = _lib.lib()
LAYOUT = 0
def run(): _lib.check(L.hh_space_attn_fwd(ctypes.c_void_p((planes if LAYOUT else qkv).data_ptr()), LAYOUT, ctypes.c_void_p(out.data_ptr()), None, B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
def t(reps=10):
    for _ in range(3): run()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ops.set_tuning("space_joint", 1); a = t(); ref = out.clone()
ops.set_tuning("space_joint", 0); b = t()
print("joint-block kernel %7.1f us   16-query kernel %7.1f us   max |diff| %.3e" % (a, b, (out[patch].float() - ref[patch].float()).abs().max().item()))
ops.set_tuning("space_joint", 1); print("joint again        %7.1f us" % t())
LAYOUT = 1; c = t(); print("joint, head-major qkv planes %7.1f us   max |diff| vs token-major %.3e" % (c, (out[patch].float() - ref[patch].float()).abs().max().item()))
ops.set_tuning("space_debug", 1); print("joint, head-major, memory only %7.1f us" % t()); ops.set_tuning("space_debug", 0)
LAYOUT = 0
for dbg, name in ((1, "memory only (stage K/V, read Q, write O)"), (2, "compute only (no K/V staging)")):
    ops.set_tuning("space_debug", dbg)
    print("joint kernel,    %-45s %7.1f us" % (name, t()))
ops.set_tuning("space_debug", 0)
ops.set_tuning("space_joint", 0)
for dbg, name in ((1, "memory only (stage K/V, read Q, write O)"), (2, "compute only (no K/V staging)")):
    ops.set_tuning("space_debug", dbg)
    print("16-query kernel, %-45s %7.1f us" % (name, t()))
ops.set_tuning("space_debug", 0); ops.set_tuning("space_joint", 1)
# ---- per-workgroup timeline of the joint kernel (debug mode 3: s_memtime stamps of wave 0; 100 MHz reference clock)
stamps = torch.zeros(B * T * heads, 4, dtype=torch.int64, device="cuda")
ops.set_tuning("space_debug", 3)
_lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(planes.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stamps.data_ptr()), B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
torch.cuda.synchronize(); ops.set_tuning("space_debug", 0)
st = stamps.double().cpu()
life = (st[:, 3] - st[:, 0]).mean()
ops.set_tuning("space_joint", 1); LAYOUT = 1; dur = t(); LAYOUT = 0
rounds = B * T * heads / 512.0                       # two workgroups per CU, 256 CUs
tick = dur / rounds / float(life)                    # the stamp counter's rate is not documented: calibrate on the kernel's own duration
print("joint kernel, per-workgroup timeline of wave 0 (head-major planes; %.1f us kernel / %.0f rounds -> %.2f us mean lifetime):" % (dur, rounds, dur / rounds))
for name, a, b_ in (("launch -> K / V / Q landed + barrier", 0, 1), ("compute (all chunks)", 1, 2), ("normalise, store, stores acknowledged", 2, 3)):
    d = (st[:, b_] - st[:, a])
    print("   %-40s %5.1f %% of the lifetime  (~%5.2f us; p10 %5.2f, p90 %5.2f)" % (name, 100 * float(d.mean() / life), float(d.mean()) * tick, float(d.quantile(0.1)) * tick, float(d.quantile(0.9)) * tick))
