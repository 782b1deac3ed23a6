#!/bin/bash
cd $GRAFT_REPO_ROOT
export AMD_SERIALIZE_KERNEL=3
for args in "1 2 2"; do
timeout 120 python - $args <<'PY' 2>&1 | grep -v amdgpu.ids | tail -5
import torch, sys
sys.path.insert(0, ".")
from helping_hand_for_egocentric_videos_amd import ops
B, T, heads = (int(v) for v in sys.argv[1:4]); n = 256
N, D = 1 + T * n, heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
base = ops.divided_attention(qkv, B, T, n, heads, "space"); torch.cuda.synchronize(); print("base ok", flush=True)
ops.set_tuning("space_prog", 3)
pp = ops.divided_attention(qkv, B, T, n, heads, "space"); torch.cuda.synchronize()
print(B, T, heads, "ok", float((pp.float() - base.float()).abs().max()))
PY
done
