import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
M = 32 * 4097
GROUPS = [2, 4, 8, 16, 32, 64]
shapes = [("qkv", 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), ("proj", 1024, 1024, {}), ("fc1", 4096, 1024, dict(act=ops.ACT_QUICKGELU)), ("fc2", 1024, 4096, {})]
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, kw in shapes:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    f = lambda: ops.gemm(a, w, bias, **kw)
    res = {G: [] for G in GROUPS}
    for rnd in range(4):
        for G in GROUPS:
            ops.set_tuning("gemm256_group", G)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            if rnd: res[G].append(e0.elapsed_time(e1) / 5)
    print(name, "  ".join(f"G{G}: {2.0*M*N*K/sorted(res[G])[1]/1e9:7.1f}" for G in GROUPS), flush=True)
