"""Which Python lines of the package issue the STOCK (non-libhh) GPU kernels of one training step: torch.profiler with stacks over one
un-pipelined step at the headline shape, stock kernels grouped by the innermost frame inside this repository.
usage (GPU box): python3 scripts/stock_census.py [B]   ->  a table on stdout"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helping_hand_for_egocentric_videos_amd import C2, synth                      # noqa: E402
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder     # noqa: E402
from helping_hand_for_egocentric_videos_amd.step import TrainStep                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
cfg = C2
backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
decoder = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1000).items()}
ts = TrainStep(cfg, backbone, decoder)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity                             # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.step(batch)
    torch.cuda.synchronize()
# CPU-side op events with their stacks; an op is "stock" when it is an aten:: op that launched at least one kernel
DEBUG = [0]
sites = collections.Counter()
ops_at = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    if any(ev2 for ev2 in (ev.cpu_children or []) if ev2.name.startswith("aten::") and ev2.kernels):
        continue                                    # count the innermost op that owns the kernels
    frame = "?"
    for fr in ev.stack or []:
        if ("helping_hand_for_egocentric_videos_amd" in fr or "bench.py" in fr) and "ops.py" not in fr:
            frame = fr.split("helping_hand_for_egocentric_videos_amd/")[-1]
            break
    if frame == "?" and DEBUG[0] < 6:
        DEBUG[0] += 1
        print("no repo frame for", ev.name, "stack:", (ev.stack or [])[:12])
    sites[frame] += len(ev.kernels)
    ops_at[frame][ev.name] += len(ev.kernels)
print("stock kernel launches of one un-pipelined step (B = %d): %d" % (B, sum(sites.values())))
for frame, n in sites.most_common():
    print("%3d  %-70s %s" % (n, frame[:70], ", ".join("%s x%d" % kv for kv in ops_at[frame].most_common())))
