"""Experiment: 256x256 GEMM tile on 4 waves x 128x128 (hh_set_tuning("gemm256", 4), csrc/gemm256w4.hip) vs the 8-wave kernels:
2 = one tile per workgroup (same structure as the experiment), 3 = persistent / continuous (the default).  Correctness first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 32 * 4096
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [("qkv", 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), ("proj", 1024, 1024, {}), ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, {})]
for name, N, K, kw in shapes:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    ops.set_tuning("gemm256", 3); ref = ops.gemm(a, w, bias, **kw); ref32 = ops.gemm(a, w, bias, out_dtype=torch.float32, **kw)
    ops.set_tuning("gemm256", 4); got = ops.gemm(a, w, bias, **kw); got32 = ops.gemm(a, w, bias, out_dtype=torch.float32, **kw)
    torch.cuda.synchronize()
    same = torch.equal(ref, got), torch.equal(ref32, got32)
    d = float((ref32 - got32).abs().max())
    res = {}
    for rnd in range(4):
        for mode in (2, 3, 4, 5):
            ops.set_tuning("gemm256", mode)
            for _ in range(2): ops.gemm(a, w, bias, **kw)
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): ops.gemm(a, w, bias, **kw)
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault(mode, []).append(e0.elapsed_time(e1) / 5)
    tf = {m: 2.0 * M * N * K / sorted(v)[len(v) // 2] / 1e9 for m, v in res.items()}
    print("%-5s N=%d K=%d  identical to the default kernel (bf16, fp32 out): %s  max|diff| %.1e   TFLOP/s: 8 waves one-tile %.0f | 8 waves persistent %.0f | 4 waves one-tile %.0f | 4 waves persistent %.0f" % (
        name, N, K, same, d, tf[2], tf[3], tf[4], tf[5]), flush=True)
# shapes of the step with a row tail (M = B * 4097), head-major planes, ReLU, K = 4096 / 512 / 768 / 640
g2 = torch.Generator(device="cuda").manual_seed(1)
for name, Mx, N, K, kw in [("qkv planes + tail", 8 * 4097, 3072, 1024, dict(colscale=0.125, colscale_cols=1024, col_blocked=True)),
                           ("fc1 + tail", 8 * 4097, 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2 + tail", 8 * 4097, 1024, 4096, {}),
                           ("relu", 16384, 2048, 512, dict(act=ops.ACT_RELU)), ("k768", 16384, 3072, 768, {}), ("k640", 32768, 1024, 640, {}),
                           ("k256", 65536, 1024, 256, {})]:
    a = torch.randn(Mx, K, device="cuda", generator=g2).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g2) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g2)
    ops.set_tuning("gemm256", 3); ref = ops.gemm(a, w, bias, **kw); ref32 = ops.gemm(a, w, bias, out_dtype=torch.float32, **kw)
    out = []
    for mode in (4, 5):
        ops.set_tuning("gemm256", mode)
        for rep in range(3):
            got = ops.gemm(a, w, bias, **kw); got32 = ops.gemm(a, w, bias, out_dtype=torch.float32, **kw)
            torch.cuda.synchronize()
            out += [torch.equal(ref, got), torch.equal(ref32, got32)]
    print("%-18s M=%d N=%d K=%d identical (modes 4, 5; 3 repeats; bf16 / fp32): %s" % (name, Mx, N, K, all(out)), out if not all(out) else "", flush=True)
ops.set_tuning("gemm256", 3)
