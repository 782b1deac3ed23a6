#!/bin/bash
# space attention variants at the headline shape: tests, kernel-trace, SQ / FETCH / WRITE counters -> gpurun_out/space_variants/ (profiles/r4_space_variants.md)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/space_variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "space_attention_persistent" 2>&1 | tail -25 > gpurun_out/space_variants/tests_space.log
tail -25 gpurun_out/space_variants/tests_space.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/space_variants/space_bench.txt
import torch, ctypes, sys
sys.path.insert(0, ".")
from helping_hand_for_egocentric_videos_amd import ops, _lib
B, T, n, heads = 32, 16, 256, 16
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g); qkv[:, :D] *= 0.5
planes = qkv.to(torch.bfloat16).view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
def t(reps=20):
    for _ in range(3): ops.divided_attention(planes, B, T, n, heads, "space")
    big.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): ops.divided_attention(planes, B, T, n, heads, "space")
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for prog in (1, 2, 1, 2):
    ops.set_tuning("space_prog", prog)
    print("space_prog=%d: %.1f us per call (space kernel + cls_combine, back to back)" % (prog, t()))
PY
