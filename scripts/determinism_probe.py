"""Identically built TrainSteps, one eval-mode backward each on the same batch: which arena entries differ, and by how much.
Runs 0, 1: plain; run 2: gradient buckets through RCCL (world 1, force_comm); run 3: plain with a 248-CU encoder budget."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import synth, TINY16, ops
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
from helping_hand_for_egocentric_videos_amd.step import TrainStep
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = TINY16
batch = {k: v.cuda() for k, v in synth.make_batch(cfg, 4, seed=21).items()}
def build(**kw):
    backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=4))
    dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4))
    return TrainStep(cfg, backbone, dec, lr=1e-4, bucket_bytes=1 << 20, **kw), dec
grads = []
hs_grads = []
from helping_hand_for_egocentric_videos_amd import step as step_mod
_orig_sim = step_mod.sim_matrix
trace = []
def traced_sim(a, b, *args, **kw):
    rec = {"a": a.detach().clone(), "b": b.detach().clone()}
    out = _orig_sim(a, b, *args, **kw)
    rec["out"] = out.detach().clone()
    if out.requires_grad:
        out.register_hook(lambda g: rec.__setitem__("d_out", g.detach().clone()))
    for nm, t in (("a", a), ("b", b)):
        if t.requires_grad:
            t.register_hook(lambda g, nm=nm: rec.__setitem__("d_" + nm, g.detach().clone()))
    trace.append(rec)
    return out
step_mod.sim_matrix = traced_sim
traces = []
for rep, kw in enumerate(({}, {"force_comm": True}, {"force_comm": True, "only": "gather"}, {"force_comm": True, "only": "buckets"},
                          {"force_comm": True, "only": "buckets-sync"})):
    only = kw.pop("only", None)
    t, d = build(**kw)
    if only == "gather":                     # forced all-gather, no gradient buckets
        for h in t.comm._hooks: h.remove()
        t.comm.enabled = False
    elif only in ("buckets", "buckets-sync"):  # gradient buckets through RCCL, plain (un-gathered) contrastive inputs
        t.force_comm = False
        if only == "buckets-sync":
            t.comm.comm_stream = None        # collectives on the compute stream
    kw["only"] = only
    d.eval()
    cap = {}
    def fwd_hook(m, i, o):
        for k, v in enumerate(o):
            if torch.is_tensor(v) and v.requires_grad:
                v.register_hook(lambda g, k=k: cap.__setitem__(k, g.detach().clone()))
    hk = d.register_forward_hook(fwd_hook)
    t.arena.zero_grad()
    out = t.losses(batch)
    out["total_loss"].backward()
    t.comm.finish()
    torch.cuda.synchronize()
    grads.append(t.arena.grads.clone())
    hk.remove()
    hs_grads.append(cap)
    traces.append(list(trace)); trace.clear()
    if rep:
        for ci, (r1, r0) in enumerate(zip(traces[rep], traces[0])):
            for key in r0:
                if key in r1:
                    x, y = r1[key].float(), r0[key].float()
                    nd = int((x != y).sum())
                    if nd:
                        print("   sim_matrix call %d, %-5s %s: %d differing, max |d| %.3e, rows %s" % (ci, key, tuple(x.shape), nd, float((x - y).abs().max()), sorted(set((x != y).nonzero()[:, 0].tolist()))[:8]))
    if rep:
        for k in cap:
            a, b0 = cap[k], hs_grads[0][k]
            print("   d(decoder output %d) %s: max |d| %.3e of scale %.3e, differing elements %d" % (k, tuple(a.shape), float((a - b0).abs().max()), float(b0.abs().max()), int((a != b0).sum())))
            if a.dim() == 4 and int((a != b0).sum()):
                print("      rows (layer, clip, query) that differ:", sorted(set(map(tuple, (a != b0).nonzero()[:, :3].tolist())))[:12])
    print("run %d %s: loss %.8f" % (rep, kw, float(out["total_loss"])))
    if rep == 0:
        continue
    diff = (grads[rep] - grads[0]).abs()
    print("   vs run 0: max |d| %.3e, mismatching (rtol 1e-4, atol 1e-7): %d" % (float(diff.max()), int((diff > 1e-7 + 1e-4 * grads[0].abs()).sum())))
    rows = []
    for name, (off, numel) in t.arena.offsets.items():
        dd = diff[off:off + numel]
        if float(dd.max()) > 1e-6:
            rows.append((float(dd.max()) / (float(grads[0][off:off + numel].abs().max()) + 1e-30), name, float(dd.max())))
    for rel, name, mx in sorted(rows, reverse=True)[:8]:
        print("      %-60s max |d| %.3e  (%.1e of the tensor's scale)" % (name, mx, rel))
dist.destroy_process_group()
