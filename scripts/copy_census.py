import collections, os, sys, torch
sys.path.insert(0, "/root/repo")
from helping_hand_for_egocentric_videos_amd import synth
from helping_hand_for_egocentric_videos_amd.config import C2
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0); torch.set_num_threads(16)
cfg, B = C2.with_(depth=4), 8
bb = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=0), device=dev)
dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=0), device=dev)
batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1).items()}
ts = TrainStep(cfg, bb, dec)
for _ in range(3): ts.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.step(batch); torch.cuda.synchronize()
by = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
        n = sum(1 for k in e.kernels if "copyBuffer" in k.name or "Memcpy" in k.name)
        if n:
            st = [s for s in (e.stack or []) if "/root/repo" in s or "repo/helping" in s]
            by[(e.name, str(e.input_shapes)[:60], (st[0] if st else "?")[-90:])] += n
for k, n in by.most_common(40): print(n, k)
print("---- all device-side copy activities, whatever launched them")
ev = prof.events()
cp = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA and ("copyBuffer" in e.name or "emcpy" in e.name)]
print(len(cp), collections.Counter(e.name for e in cp).most_common(5))
cpu = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
# the python-level op that was running on the host when the copy was enqueued is unknown for ctypes calls; list copy durations
print(sorted(collections.Counter(round(e.device_time_total) for e in cp).items())[:20])
tops = collections.Counter()
for e in cpu:
    if e.cpu_parent is None:
        n = 0
        stack = [e]
        while stack:
            x = stack.pop()
            n += sum(1 for k in x.kernels if "copyBuffer" in k.name or "emcpy" in k.name)
            stack.extend(x.cpu_children)
        if n: tops[(e.name, str(e.input_shapes)[:50])] += n
for k, n in tops.most_common(25): print(n, k)
