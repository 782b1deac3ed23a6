"""Round 6: space attention on 32x32x16 MFMAs (space_attn32_kernel, hh_set_tuning("space_mfma32", 1)) against the joint-block 16x16x32 kernel:
max |diff|, time per call alone (B = 32 and B = 2), memory-only / compute-only variants of both."""
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
planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
out = torch.zeros(B * N, D, dtype=torch.bfloat16, device="cuda")
part = torch.zeros(B * heads * T * 68, device="cuda")
patch = (torch.arange(B * N, device="cuda") % N) != 0
L = _lib.lib()
def run(): _lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(planes.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(part.data_ptr()), B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
def t(reps=200):
    for _ in range(20): run()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
if n == 576:        # config 4: 0 / 1 = the progressive 16x16x32 kernel (default), 2 = third-step pipelined + progressive staging, 3 = third-step, plain staging
    for rnd in range(3):
        for k in (0, 2, 3):
            ops.set_tuning("space_mfma32", k)
            res.setdefault(k, []).append(t())
            if rnd == 0:
                res["out%d" % k] = out.clone(); res["part%d" % k] = part.clone()
    print("n = 576: progressive 16x16x32 kernel (default): %s us    third-step pipelined + progressive staging: %s us    third-step, plain staging: %s us" % tuple(" ".join("%.1f" % v for v in res[k]) for k in (0, 2, 3)))
    for k in (2, 3):
        d = (res["out%d" % k][patch].float() - res["out0"][patch].float()).abs()
        print("mfma32=%d vs default: max |diff| rows %.3e (scale %.3f), mean %.3e; CLS partial records max |diff| %.3e" % (k, d.max().item(), res["out0"].float().abs().max().item(), d.mean().item(), (res["part%d" % k] - res["part0"]).abs().max().item()))
    print("progressive vs plain staging of the third-step kernel: rows equal %s" % torch.equal(res["out2"][patch], res["out3"][patch]))
    for k in (2, 3):
        ops.set_tuning("space_mfma32", k)
        for dbg, name in ((1, "memory only (stage K / V, read Q, write rows)"), (2, "staging + walk, no output / CLS partial")):
            ops.set_tuning("space_debug", dbg)
            print("third-step kernel, mfma32=%d  %-48s %7.1f us" % (k, name, t()))
        ops.set_tuning("space_debug", 0)
    ops.set_tuning("space_mfma32", 1)
    sys.exit(0)
for rnd in range(3):
    for k in (0, 1, 2):
        ops.set_tuning("space_mfma32", k)
        res.setdefault(k, []).append(t())
        if rnd == 0:
            res["out%d" % k] = out.clone(); res["part%d" % k] = part.clone()
print("joint 16x16x32 kernel: %s us    32x32x16 kernel, one problem per workgroup: %s us    32x32x16 persistent: %s us" % tuple(" ".join("%.1f" % v for v in res[k]) for k in (0, 1, 2)))
for k in (1, 2):
    d = (res["out%d" % k][patch].float() - res["out0"][patch].float()).abs()
    print("mfma32=%d vs joint: max |diff| rows %.3e (scale %.3f), mean %.3e; CLS partial records max |diff| %.3e" % (k, d.max().item(), res["out0"].float().abs().max().item(), d.mean().item(), (res["part%d" % k] - res["part0"]).abs().max().item()))
print("persistent vs one-per-workgroup: rows equal %s, CLS records equal %s" % (torch.equal(res["out1"][patch], res["out2"][patch]), torch.equal(res["part1"], res["part2"])))
for k in (0, 1):
    ops.set_tuning("space_mfma32", k)
    for dbg, name in ((1, "memory only"), (2, "compute only (no K/V staging)")):
        ops.set_tuning("space_debug", dbg)
        print("mfma32=%d  %-32s %7.1f us" % (k, name, t()))
    ops.set_tuning("space_debug", 0)
if n != 256:
    sys.exit(0)            # (the debug timelines below exist for the n = 256 kernels only)
ops.set_tuning("space_mfma32", 1)
# ---- per-workgroup timeline of the 32x32 kernel (debug mode 3: s_memtime stamps of wave 0)
stamps = torch.zeros(B * T * heads, 8, dtype=torch.int64, device="cuda")
ops.set_tuning("space_debug", 3)
_lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(planes.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stamps.data_ptr()), B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
torch.cuda.synchronize(); ops.set_tuning("space_debug", 0)
st = stamps.double().cpu()
life = (st[:, 5] - st[:, 0]).mean()
dur = t()
rounds = B * T * heads / 512.0
tick = dur / rounds / float(life)
print("32x32 kernel, per-workgroup timeline of wave 0 (%.1f us kernel / %.0f rounds -> %.2f us mean lifetime; %.0f stamp ticks):" % (dur, rounds, dur / rounds, float(life)))
for name, a, b_ in (("launch -> K / V / Q landed + barrier", 0, 1), ("chunk loop", 1, 2), ("redo check + CLS partial + barrier", 2, 3), ("rows through LDS, stores issued", 3, 4), ("stores acknowledged", 4, 5)):
    d = (st[:, b_] - st[:, a])
    print("   %-40s %5.1f %% of the lifetime  (~%5.2f us, %6.0f ticks; p10 %5.2f, p90 %5.2f us)" % (name, 100 * float(d.mean() / life), float(d.mean()) * tick, float(d.mean()), float(d.quantile(0.1)) * tick, float(d.quantile(0.9)) * tick))

# ---- persistent kernel: stamps of each workgroup's SECOND problem (wave 0)
if n == 256:
    ops.set_tuning("space_mfma32", 2)
    stamps.zero_()
    ops.set_tuning("space_debug", 3)
    _lib.check(L.hh_space_attn_fwd(ctypes.c_void_p(planes.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stamps.data_ptr()), B, T, n, heads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
    torch.cuda.synchronize(); ops.set_tuning("space_debug", 0)
    st = stamps[:256].double().cpu()
    st = st[st[:, 3] > 0]
    dur = t()
    cnt = float(st[:, 3].mean())
    tot = float((st[:, 0] + st[:, 1] + st[:, 2]).mean())
    span = float((st[:, 5] - st[:, 4]).mean())
    real = (st[:, 7] - st[:, 6]) / 100.0                 # us, workgroup start -> end of its last problem (100 MHz counter)
    t_first = float((st[:, 6] - st[:, 6].min()).max()) / 100.0
    t_end = float((st[:, 7] - st[:, 6].min()).max()) / 100.0
    print("persistent kernel %.1f us, %.1f problems per workgroup; compute wave 0: %.0f cycles per problem in its three phases, %.0f cycles from the first top to the last bottom;" % (dur, cnt, tot / cnt, span))
    print("   workgroup start -> end of its last problem %.1f us mean (%.1f .. %.1f); last workgroup starts %.1f us after the first; last problem ends %.1f us after the first start" % (float(real.mean()), float(real.min()), float(real.max()), t_first, t_end))
    for name, c in (("Q fragments from LDS + chunk loop", 0), ("redo check + CLS partial + barrier A (waits for the loaders' requests)", 1), ("rows -> LDS + barrier B (loaders write K / V / Q)", 2)):
        d = st[:, c] / st[:, 3]
        print("   %-72s %5.1f %%  (%6.0f cycles per problem; p10 %6.0f, p90 %6.0f over workgroups)" % (name, 100 * float(st[:, c].mean()) / tot, float(d.mean()), float(d.quantile(0.1)), float(d.quantile(0.9))))
ops.set_tuning("space_mfma32", 1)
# ---- one launch at a time with the chip idle in between (no sustained load: the clock the power management allows a lone 250-us kernel)
import time
for k, name in ((0, "joint 16x16x32"), (1, "32x32x16 one problem per workgroup"), (2, "32x32x16 persistent")):
    ops.set_tuning("space_mfma32", k)
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); time.sleep(0.02)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("single launches, 20 ms idle between: %-36s median %.1f us (min %.1f, max %.1f)" % (name, ts[len(ts) // 2], ts[0], ts[-1]))
ops.set_tuning("space_mfma32", 1)
