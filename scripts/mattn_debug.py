"""Debug probes of hh_mattn_fwd (scratch)."""
import torch
from helping_hand_for_egocentric_videos_amd import ops
torch.manual_seed(0)
DEV = "cuda"
B, Q, M, H, C = 1, 13, 128, 8, 512
def run(qt, mp, mem, slices=1):
    return ops.mattn_fwd(qt.to(DEV), mp.to(DEV).to(torch.bfloat16), mem.to(DEV).to(torch.bfloat16), Q, slices=slices)
def ref(qt, mp, mem):
    q4 = qt.double().view(B, Q, H, C)
    s = torch.einsum("bqhc,bmc->bqhm", q4, mp.to(torch.bfloat16).double())
    p = torch.softmax(s, -1)
    return torch.einsum("bqhm,bmc->bqhc", p, mem.to(torch.bfloat16).double()).reshape(B * Q, H * C), s
# 1. uniform attention (qt = 0), mem = ones -> pooled = 1
qt = torch.zeros(B * Q, H * C); mp = torch.randn(B, M, C); mem = torch.ones(B, M, C)
po, l2, rs = run(qt, mp, mem)
print("1 uniform/ones: min %.4f max %.4f lse2 %.4f (want log2 M = %.4f)" % (po.min(), po.max(), l2.mean(), torch.log2(torch.tensor(float(M)))))
# 2. uniform attention, mem = column index -> pooled[., c] = c
mem = torch.arange(C).float()[None, None, :].expand(B, M, C).contiguous() / 64
po, _, _ = run(qt, mp, mem)
want = (torch.arange(C).float() / 64).to(torch.bfloat16).float()
print("2 uniform/colidx: err %.4e" % (po.cpu().view(B * Q, H, C) - want).abs().max())
print("   row0 head0 first 16:", po[0, :16].cpu().tolist())
# 3. uniform attention, mem = key index -> pooled = mean(keys)
mem = (torch.arange(M).float() / 16)[None, :, None].expand(B, M, C).contiguous()
po, _, _ = run(qt, mp, mem)
print("3 uniform/keyidx: min %.4f max %.4f want %.4f" % (po.min(), po.max(), (torch.arange(M).float() / 16).to(torch.bfloat16).float().mean()))
# 4. scores: qt one-hot on dim d for query q -> score = mp[., d]; use mem = onehot(key) in first M columns to read P out
for d in (0, 7, 8, 37, 255, 256, 300, 511):
    qt = torch.zeros(B * Q, H * C)
    qt.view(B, Q, H, C)[:, :, :, d] = 1.0
    mp = torch.randn(B, M, C)
    mem = torch.zeros(B, M, C); mem[0, torch.arange(M), torch.arange(M)] = 1.0
    po, l2, _ = run(qt, mp, mem)
    rp, s = ref(qt, mp, mem)
    print("4 onehot d=%3d: P err %.3e  lse err %.3e" % (d, (po.cpu().double() - rp).abs().max(), (l2.cpu().double() - torch.logsumexp(s, -1).reshape(B * Q, H) * 1.4426950408889634).abs().max()))
# 5. random
qt = torch.randn(B * Q, H * C) * 0.08; mp = torch.randn(B, M, C); mem = torch.randn(B, M, C)
po, l2, _ = run(qt, mp, mem)
rp, s = ref(qt, mp, mem)
e = (po.cpu().double() - rp).abs().view(B * Q, H, C)
print("5 random: err %.3e; per head" % e.max(), e.amax((0, 2)).tolist(), "per q", e.amax((1, 2)).tolist())
print("   per dim-half", e[:, :, :256].max().item(), e[:, :, 256:].max().item())
