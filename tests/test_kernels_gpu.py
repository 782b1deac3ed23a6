"""GPU parity tests of every libhh kernel (through the C ABI) against plain fp32 PyTorch / the oracle.

Tolerances: kernels consume bf16 operands and accumulate in fp32; references are computed in fp32 on the
SAME bf16-rounded inputs, so the residual is accumulation order + one bf16 rounding of the output
(relative 2^-8 = 3.9e-3).  Index work (matching) is bit-exact.
"""
import numpy as np
import pytest
import torch

from helping_hand_for_egocentric_videos_amd import ops
from _record import record

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def bf(x):
    return x.to(torch.bfloat16)


def assert_close_bf16(got, ref, rel=8e-3, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("cols,rows", [(1024, 4097), (512, 1000), (128, 37), (768, 64)])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                     (torch.bfloat16, torch.bfloat16)])
def test_layernorm(cols, rows, xdt, ydt):
    x = (rnd(rows, cols, seed=1) * 2 + 0.3).to(xdt)
    g, b = rnd(cols, seed=2) * 0.1 + 1, rnd(cols, seed=3) * 0.1
    ref = torch.nn.functional.layer_norm(x.float(), (cols,), g, b, 1e-6)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, out_dtype=ydt)
    if ydt == torch.float32:
        torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5)
    else:
        assert_close_bf16(y, ref, 5e-3, "layernorm")


@pytest.mark.parametrize("write_x", [True, False])
@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32])
def test_add_layernorm(write_x, ydt):
    rows, cols = 4097, 1024
    x, d = rnd(rows, cols, seed=1, scale=2.0), bf(rnd(rows, cols, seed=2))
    g, b = rnd(cols, seed=3) * 0.1 + 1, rnd(cols, seed=4) * 0.1
    s = x + d.float()
    ref = torch.nn.functional.layer_norm(s, (cols,), g, b, 1e-6)
    X = x.to(DEV)
    y = ops.add_layernorm(X, d.to(DEV), g.to(DEV), b.to(DEV), 1e-6, write_x=write_x, out_dtype=ydt)
    assert torch.equal(X.cpu(), s if write_x else x)
    if ydt == torch.float32:
        torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5)
    else:
        assert_close_bf16(y, ref, 5e-3, "add_layernorm")


def test_layernorm_bwd():
    rows, cols = 1031, 512
    x = rnd(rows, cols, seed=4, scale=2.0)
    g, b = rnd(cols, seed=5) * 0.1 + 1, rnd(cols, seed=6) * 0.1
    dy = rnd(rows, cols, seed=7)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5).backward(dy)
    y, mean, rstd = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, out_dtype=torch.float32, save_stats=True)
    dx, dg, db = ops.layernorm_bwd(x.to(DEV), g.to(DEV), mean, rstd, dy.to(DEV))
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(4097, 3072, 1024), (300, 128, 64), (8194, 1024, 4096), (1, 256, 128), (129, 384, 128),
                                   # row tails of the persistent kernel / short GEMMs -> split-K-in-workgroup tail kernel (K % 256 == 0)
                                   (4096 + 32, 1024, 1024), (2048 + 64, 128, 4096), (2048 + 1, 384, 768), (33, 256, 512), (64, 128, 256),
                                   (2048 + 65, 256, 256), (2048 + 37, 256, 640)])
def test_gemm_plain_and_epilogues(M, N, K):
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.05))
    bias = rnd(N, seed=3)
    ref = a.float() @ w.float().t()
    A, W, Bv = a.to(DEV), w.to(DEV), bias.to(DEV)
    assert_close_bf16(ops.gemm(A, W), ref, 8e-3, "plain")
    out32 = ops.gemm(A, W, Bv, out_dtype=torch.float32)
    torch.testing.assert_close(out32.cpu(), ref + bias, rtol=2e-3, atol=2e-3 * ref.abs().max().item())
    # qkv epilogue: bias then scale the first N/3.. columns
    cs = (N // 128 // 3 or 1) * 128
    r = ref + bias
    r[:, :cs] *= 0.125
    assert_close_bf16(ops.gemm(A, W, Bv, colscale=0.125, colscale_cols=cs), r, 8e-3, "qkv")
    # QuickGELU
    r = ref + bias
    assert_close_bf16(ops.gemm(A, W, Bv, act=ops.ACT_QUICKGELU), r * torch.sigmoid(1.702 * r), 8e-3, "gelu")
    assert_close_bf16(ops.gemm(A, W, Bv, act=ops.ACT_RELU), torch.relu(r), 8e-3, "relu")
    # fp32 residual, in place
    res = rnd(M, N, seed=5)
    R = res.to(DEV)
    out = ops.gemm(A, W, Bv, out=R, resid=R)
    assert out.data_ptr() == R.data_ptr()
    torch.testing.assert_close(R.cpu(), ref + bias + res, rtol=2e-3, atol=2e-3 * ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(4097, 3072, 1024), (300, 384, 128), (2 * 4097, 768, 256), (33, 128, 512), (256 * 6, 256, 64)])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_gemm_column_blocked_output(M, N, K, odt):
    """c_block_stride: C as N / 64 planes [M, 64] (the head-major q|k|v buffer) -- every kernel of hh_gemm_bf16 (persistent 256x256,
    128x128, row-tail) stores the same values at the blocked addresses: equal bit for bit to the row-major result."""
    a, w, bias = bf(rnd(M, K, seed=1)).to(DEV), bf(rnd(N, K, seed=2) * 0.1).to(DEV), rnd(N, seed=3).to(DEV)
    rows = ops.gemm(a, w, bias, out_dtype=odt, colscale=0.125, colscale_cols=N // 3 // 128 * 128)
    planes = ops.gemm(a, w, bias, out_dtype=odt, colscale=0.125, colscale_cols=N // 3 // 128 * 128, col_blocked=True)
    assert planes.shape == (N // 64, M, 64)
    assert torch.equal(planes.transpose(0, 1).reshape(M, N), rows)


@pytest.mark.parametrize("mode", [2, 3, 5])
@pytest.mark.parametrize("M,N,K", [(256 * 40, 2048, 256), (256 * 24, 3072, 192), (256 * 300, 256, 128), (256 * 20, 4096, 1024),
                                   (256 * 33 + 17, 1024, 512), (256 * 300, 256, 64), (256 * 40, 8192 + 256, 128)])
def test_gemm_persistent_walks_many_tiles_per_workgroup(mode, M, N, K):
    """More 256x256 tiles than CUs: every workgroup of the persistent kernel (mode 3: continuous k-tile stream across tile
    boundaries) walks several tiles, with even / odd k-tile counts (LDS buffer parity) and the 2-k-tile corner; mode 2 = the
    one-tile-per-block kernel on the same shapes."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=DEV, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = a.float() @ w.float().t() + bias
    ops.set_tuning("gemm256", mode)
    try:
        out_bf = ops.gemm(a, w, bias)
        out_f32 = ops.gemm(a, w, bias, out_dtype=torch.float32)
        out_act = ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU)
    finally:
        ops.set_tuning("gemm256", ops.GEMM256_DEFAULT)
    scale = ref.abs().max().item()
    assert (out_f32 - ref).abs().max().item() <= 2e-3 * scale
    assert (out_bf.float() - ref).abs().max().item() <= 8e-3 * scale
    r = ref * torch.sigmoid(1.702 * ref)
    assert (out_act.float() - r).abs().max().item() <= 8e-3 * r.abs().max().item()


@pytest.mark.parametrize("mode", [4, 5])
@pytest.mark.parametrize("M,N,K,kw", [(256 * 40 + 33, 3072, 1024, dict(colscale=0.125, colscale_cols=1024)), (256 * 64, 1024, 256, dict(act=ops.ACT_RELU)),
                                      (256 * 300, 256, 640, dict(act=ops.ACT_QUICKGELU)), (256 * 20, 4096, 768, {})])
def test_gemm_four_wave_kernels_match_the_eight_wave_kernel(mode, M, N, K, kw):
    """csrc/gemm256w4.hip (hh_set_tuning("gemm256", 4 / 5): the 256x256 tile on 4 waves of 128x128, one tile per workgroup /
    persistent -- 5 is the default) accumulates every output in the same order as the 8-wave persistent kernel (3): bit-identical,
    bf16 and fp32 outputs, head-major planes, row tail inside the persistent walk."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=DEV, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    ops.set_tuning("gemm256", 3)
    ref = [ops.gemm(a, w, bias, **kw), ops.gemm(a, w, bias, out_dtype=torch.float32, **kw), ops.gemm(a, w, bias, col_blocked=True, **kw)]
    ops.set_tuning("gemm256", mode)
    try:
        for rep in range(2):
            got = [ops.gemm(a, w, bias, **kw), ops.gemm(a, w, bias, out_dtype=torch.float32, **kw), ops.gemm(a, w, bias, col_blocked=True, **kw)]
            for r, o in zip(ref, got):
                assert torch.equal(r, o)
    finally:
        ops.set_tuning("gemm256", ops.GEMM256_DEFAULT)


@pytest.mark.parametrize("M,N,K", [(256 * 37 + 20, 1024, 1024), (256 * 300, 256, 512), (256 * 20, 4096, 768), (256 * 77, 3072, 384)])
def test_gemm_dynamic_tile_walk_matches_the_static_one(M, N, K):
    """hh_set_tuning("gemm256_dynamic", 1) (default): a workgroup of the persistent 4-wave kernel takes every tile after its first from
    its XCD's atomic counter; the counters are per stream and zeroed again by the launch's last workgroup.  Same tiles, same results
    as the static stride, launch after launch, on two streams at once, with a CU budget, with ragged m-tile counts per XCD."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=DEV, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    ops.set_tuning("gemm256_dynamic", 0)
    try:
        ref = ops.gemm(a, w, bias)
    finally:
        ops.set_tuning("gemm256_dynamic", 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ops.set_stream_cu_budget(s2, 64)
    for s_ in (s1, s2):
        s_.wait_stream(torch.cuda.current_stream())
    outs = []
    for rep in range(4):
        for s_ in (s1, s2):
            with torch.cuda.stream(s_):
                outs.append(ops.gemm(a, w, bias))
        outs.append(ops.gemm(a, w, bias))
    torch.cuda.synchronize()
    ops.set_stream_cu_budget(s2, 0)
    for o in outs:
        assert torch.equal(o, ref)


def test_stream_cu_budget_changes_the_grid_not_the_results():
    """hh_stream_set_cu_budget: persistent GEMMs launched on a budgeted stream walk their tiles with fewer workgroups."""
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(256 * 20, 512, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(1024, 512, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    ref = ops.gemm(a, w)
    s = torch.cuda.Stream()
    assert ops.stream_cu_budget(s) == ops.stream_cu_budget()            # default: every CU
    ops.set_stream_cu_budget(s, 64)
    assert ops.stream_cu_budget(s) == 64
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = ops.gemm(a, w)
    s.synchronize()
    assert torch.equal(out, ref)
    with pytest.raises(RuntimeError):
        ops.set_stream_cu_budget(s, 12)                                  # not a multiple of 8
    ops.set_stream_cu_budget(s, 0)
    assert ops.stream_cu_budget(s) == ops.stream_cu_budget()


@pytest.mark.parametrize("M,N,K,act", [(256 * 12 + 32, 4096, 1024, ops.ACT_QUICKGELU), (256 * 48 + 17, 1024, 4096, ops.ACT_NONE),
                                        (256 * 16 + 32, 3072, 1024, ops.ACT_NONE), (256 * 12 + 1, 4096, 512, ops.ACT_NONE),
                                        (256 * 12 + 40, 4096, 1024, ops.ACT_NONE), (256 * 48 + 64, 1024, 1024, ops.ACT_QUICKGELU)])
@pytest.mark.parametrize("budget", [0, 64])
def test_gemm_row_tail_inside_the_persistent_kernel(M, N, K, act, budget):
    """The <= 64 rows behind the last full 256-row tile are computed by the first N / 32 (x 2 beyond 32 rows) workgroups of the persistent kernel before
    their tile walk (hh_set_tuning("gemm_tail", 1), default) -- same arithmetic as the stand-alone tail kernel ("gemm_tail" 2): equal
    bit for bit, also on a stream with a 64-CU budget (two or three pieces per workgroup) and with column-blocked output."""
    a, w, bias = bf(rnd(M, K, seed=5)).to(DEV), bf(rnd(N, K, seed=6) * 0.05).to(DEV), rnd(N, seed=7).to(DEV)
    kw = dict(act=act, colscale=0.125 if act == ops.ACT_NONE else 1.0, colscale_cols=N // 4 // 128 * 128 if act == ops.ACT_NONE else 0)
    try:
        ops.set_tuning("gemm_tail", 2)
        ref = ops.gemm(a, w, bias, **kw)
    finally:
        ops.set_tuning("gemm_tail", 1)
    s = torch.cuda.Stream()
    ops.set_stream_cu_budget(s, budget)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = ops.gemm(a, w, bias, **kw)
        planes = ops.gemm(a, w, bias, col_blocked=True, **kw)
    s.synchronize()
    ops.set_stream_cu_budget(s, 0)
    assert torch.equal(out, ref)
    assert torch.equal(planes.transpose(0, 1).reshape(M, N), ref)
    tail = slice(M // 256 * 256, M)
    full = (a[tail].float() @ w.float().t() + bias)
    full[:, :kw["colscale_cols"]] *= kw["colscale"]
    if act == ops.ACT_QUICKGELU:
        full = full * torch.sigmoid(1.702 * full)
    assert_close_bf16(out[tail], full, 8e-3, "tail rows")


def test_gemm_timeline_debug_records_monotonic_stamps_and_a_plausible_clock():
    import ctypes
    import numpy as np
    from helping_hand_for_egocentric_videos_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(6)
    a = torch.randn(256 * 512, 1024, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(1024, 1024, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    ops.gemm(a, w)
    ops.set_tuning("gemm256_debug_ts", 1)
    try:
        ops.gemm(a, w)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("gemm256_debug_ts", 0)
    buf = np.zeros((256, 8, 7), dtype=np.uint64)
    _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "hh_debug_gemm_timeline")
    t = buf.astype(np.int64)
    assert (np.diff(t[:, :, :5], axis=2) >= 0).all()                     # stamps of a tile are ordered
    assert (t[:, 1:, 0] >= t[:, :-1, 4]).all()                           # tiles of a workgroup follow each other
    mhz = (t[:, :, 6] - t[:, :, 5]) / np.maximum((t[:, :, 2] - t[:, :, 1]) / 100.0, 1e-9)
    assert 500 < mhz.mean() < 3000, mhz.mean()


@pytest.mark.parametrize("K,M,N,splits", [(4097, 128, 128, None), (1000, 256, 384, 1), (64 * 37 + 5, 3072, 512, None), (300, 128, 256, 7),
                                          (131104 // 8, 512, 1024, None)])
def test_gemm_tn_weight_gradient_layout(K, M, N, splits):
    """C = At^T Bt with token-major operands (row-strided column slices of wider buffers), ragged K, explicit / automatic split-K."""
    g = torch.Generator(device=DEV).manual_seed(K + M + N)
    wide_a = torch.randn(K, M + 64, device=DEV, generator=g).to(torch.bfloat16)
    wide_b = (torch.randn(K, N + 128, device=DEV, generator=g) * 0.3).to(torch.bfloat16)
    at, bt = wide_a[:, 64:], wide_b[:, :N]
    out = ops.gemm_tn(at, bt, splits=splits)
    ref = at.float().t() @ bt.float()
    assert out.shape == (M, N)
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # column sums of At (the bias gradient when At is dY) as a by-product of the same launch
    out2, cs = ops.gemm_tn(at, bt, splits=splits, colsum=True)
    assert torch.equal(out2, out)
    cref = at.double().sum(0)
    assert cs.shape == (M,) and (cs.double() - cref).abs().max().item() <= 1e-5 * at.float().abs().sum(0).max().item() + 1e-4


@pytest.mark.parametrize("S,L,heads", [(7, 77, 12), (3, 16, 2), (2, 33, 1), (5, 80, 3), (1, 1, 2), (160, 77, 12)])
def test_text_causal_attention(S, L, heads):
    """hh_text_attn_fwd against fp32 softmax(QK^T + causal mask) V on the same bf16 qkv (q pre-scaled)."""
    W = heads * 64
    qkv = rnd(S * L, 3 * W, seed=S + L)
    qkv[:, :W] *= 0.4
    qkv[3 % (S * L), :64] += 4.0
    qkv = bf(qkv)
    out = ops.text_attention(qkv.to(DEV), S, L, heads)
    q, k, v = qkv.float().view(S, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2)
    sc = sc.masked_fill(torch.ones(L, L, dtype=torch.bool).triu(1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(S * L, W)
    assert_close_bf16(out, ref, 1.2e-2, "text-attn")
    err = (out.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2


def test_gemm_row_remap():
    B, T, n, K, N = 2, 4, 256, 640, 128
    a, w = bf(rnd(B * T * n, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.05))
    out = torch.zeros((B * (1 + T * n), N), dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out=out, remap=(T * n, 1, 1))
    ref = (a.float() @ w.float().t()).view(B, T * n, N)
    got = out.view(B, 1 + T * n, N).cpu()
    assert got[:, 0].abs().max() == 0
    torch.testing.assert_close(got[:, 1:], ref, rtol=2e-3, atol=2e-3)


def test_cast_transpose():
    x = rnd(1000, 520, seed=1)
    X = x.to(DEV)
    assert torch.equal(ops.to_bf16(X).cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.to_f32(ops.to_bf16(X)).cpu(), x.to(torch.bfloat16).float())
    t = ops.transpose_bf16(X, pad_cols_to=64)
    assert t.shape == (520, 1024)
    assert torch.equal(t[:, :1000].cpu(), x.t().to(torch.bfloat16))
    assert t[:, 1000:].abs().max() == 0


def test_patch_embed_front_end():
    from helping_hand_for_egocentric_videos_amd import synth, TINY4
    from oracle import encoder as OE
    cfg = TINY4
    sd = synth.encoder_state(cfg, seed=1, with_text=False)
    video = synth.make_batch(cfg, 2, seed=1)["video"]
    B, T = 2, cfg.num_frames
    n, D, P = cfg.patches_per_frame, cfg.embed_dim, cfg.patch_size
    patches = ops.patch_im2col(video.to(DEV), P, 640)
    ref_p = torch.nn.functional.unfold(video.flatten(0, 1), P, stride=P).transpose(1, 2).reshape(B * T * n, -1)
    assert torch.equal(patches[:, :588].cpu(), ref_p.to(torch.bfloat16))
    assert patches[:, 588:].abs().max() == 0
    w = torch.zeros(D, 640)
    w[:, :588] = sd["visual.patch_embed.proj.weight"].reshape(D, -1)
    tok = ops.gemm(patches, bf(w).to(DEV), out_dtype=torch.float32)
    x = ops.embed_ln_pre(tok, sd["visual.cls_token"].view(-1).to(DEV), sd["visual.pos_embed"][0].contiguous().to(DEV),
                         sd["visual.temporal_embed"][0].contiguous().to(DEV), sd["visual.ln_pre.weight"].to(DEV),
                         sd["visual.ln_pre.bias"].to(DEV), B, T, n)
    ref = OE.embed_tokens(video, sd, cfg)
    torch.testing.assert_close(x.cpu(), ref, rtol=2e-2, atol=2e-2)      # bf16 patch GEMM operands
    # the same pass can also emit z = bf16(x) and its row statistics for the first block's folded LayerNorm (round 5)
    x2, z, st = ops.embed_ln_pre(tok, sd["visual.cls_token"].view(-1).to(DEV), sd["visual.pos_embed"][0].contiguous().to(DEV),
                                 sd["visual.temporal_embed"][0].contiguous().to(DEV), sd["visual.ln_pre.weight"].to(DEV),
                                 sd["visual.ln_pre.bias"].to(DEV), B, T, n, z_eps=1e-6)
    assert torch.equal(x2, x) and torch.equal(z, x.view(-1, D).to(torch.bfloat16))
    torch.testing.assert_close(st, ops.ln_rowstats(z, 1e-6), rtol=1e-5, atol=1e-6)
    zf = z.float()
    rstd = (zf.var(1, unbiased=False) + 1e-6).rsqrt()
    torch.testing.assert_close(st[:, 0], rstd, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(st[:, 1], -rstd * zf.mean(1), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("clips,N,cols,xdt,ydt", [(3, 1025, 1024, torch.float32, torch.bfloat16), (2, 5, 128, torch.float32, torch.float32), (1, 2, 512, torch.bfloat16, torch.bfloat16)])
def test_layernorm_split_cls(clips, N, cols, xdt, ydt):
    """hh_layernorm_split_cls_fwd: the tower's final LayerNorm writes the CLS rows and the patch rows to two dense tensors (the decoder's
    grid needs no strided copy of the feature map): bit-identical to hh_layernorm_fwd followed by the slicing."""
    x = (rnd(clips * N, cols, seed=1) * 2 + 0.3).to(xdt).to(DEV)
    g, b = (rnd(cols, seed=2) * 0.1 + 1).to(DEV), (rnd(cols, seed=3) * 0.1).to(DEV)
    full = ops.layernorm(x, g, b, 1e-6, out_dtype=ydt).view(clips, N, cols)
    cls, pat = ops.layernorm_split_cls(x, g, b, 1e-6, clips, out_dtype=ydt)
    assert cls.shape == (clips, cols) and pat.shape == (clips, N - 1, cols) and pat.is_contiguous()
    assert torch.equal(cls, full[:, 0]) and torch.equal(pat, full[:, 1:])
    # round 5: the input as a bf16 pair (the tower's residual stream x = hi + lo): the same bits as the LayerNorm of the fp32 sum
    xf = x.float()
    hi = xf.to(torch.bfloat16)
    lo = (xf - hi.float()).to(torch.bfloat16)
    want = ops.layernorm(hi.float() + lo.float(), g, b, 1e-6, out_dtype=ydt).view(clips, N, cols)
    cls2, pat2 = ops.layernorm_split_cls(hi, g, b, 1e-6, clips, out_dtype=ydt, x_lo=lo)
    assert torch.equal(cls2, want[:, 0]) and torch.equal(pat2, want[:, 1:])


@pytest.mark.parametrize("channels_last", [False, True])
def test_uint8_front_end_equals_float_path(channels_last):
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (2, 4, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor(ops.NORM_MEAN), torch.tensor(ops.NORM_STD)
    f32 = (u8.float() / 255 - mean.view(1, 1, 3, 1, 1)) / std.view(1, 1, 3, 1, 1)
    ref = ops.patch_im2col(f32.to(DEV), 14, 640).float().cpu()
    inp = u8.permute(0, 1, 3, 4, 2).contiguous() if channels_last else u8
    got = ops.patch_im2col_u8(inp.to(DEV), 14, 640).float().cpu()
    assert float((got - ref).abs().max()) <= 2 ** -6          # <= 1 bf16 ulp at |x| < 4 (division rounding)
    assert float((got != ref).float().mean()) < 1e-3


def _ref_divided(qkv, B, T, n, heads, mode):
    """fp32 reference of the attention core on the same bf16 qkv (q pre-scaled): oracle maths, LaviLa.py:255-279.  The space kernel's
    q carries an extra log2(e) (base-2 logits, include/hh.h): taken out again here."""
    N, D = 1 + T * n, heads * 64
    q, k, v = qkv.float().view(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    if mode == "space":
        q = q / ops.LOG2E
    cls = torch.softmax(q[:, :, :1] @ k.transpose(-1, -2), -1) @ v
    ql, kl, vl = (t[:, :, 1:].reshape(B, heads, T, n, 64) for t in (q, k, v))
    if mode == "time":
        ql, kl, vl = (t.transpose(2, 3) for t in (ql, kl, vl))
    G = ql.shape[2]
    kl = torch.cat([k[:, :, None, :1].expand(B, heads, G, 1, 64), kl], 3)
    vl = torch.cat([v[:, :, None, :1].expand(B, heads, G, 1, 64), vl], 3)
    o = torch.softmax(ql @ kl.transpose(-1, -2), -1) @ vl
    if mode == "time":
        o = o.transpose(2, 3)
    o = torch.cat([cls, o.reshape(B, heads, T * n, 64)], 2)
    return o.permute(0, 2, 1, 3).reshape(B * N, D)


@pytest.mark.parametrize("mode", ["space", "time"])
@pytest.mark.parametrize("B,T,n,heads", [(2, 4, 256, 2), (1, 16, 256, 16), (1, 2, 576, 2), (1, 32, 64, 2), (1, 3, 160, 2), (1, 2, 128, 1)])
def test_divided_attention(mode, B, T, n, heads):
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=T + n)
    qkv[:, :D] *= 0.6            # realistic logits (|s| up to ~15)
    qkv[5, :64] += 6.0            # spike one query
    if mode == "space":
        qkv[:, :D] *= ops.LOG2E   # the kernel's contract: q pre-scaled by d^-1/2 * log2(e)
    qkv = bf(qkv)
    out = ops.divided_attention(qkv.to(DEV), B, T, n, heads, mode)
    ref = _ref_divided(qkv, B, T, n, heads, mode)
    # head-major planes [3*heads, B*N, 64] (what the QKV GEMM writes with col_blocked=True): same kernels, other strides
    planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous().to(DEV)
    assert torch.equal(ops.divided_attention(planes, B, T, n, heads, mode), out)
    assert torch.equal(ops.divided_attention(planes, B, T, n, heads, mode, fold_cls=False), ops.divided_attention(qkv.to(DEV), B, T, n, heads, mode, fold_cls=False))
    sep = ops.divided_attention(qkv.to(DEV), B, T, n, heads, mode, fold_cls=False)       # stand-alone CLS kernel
    assert torch.equal(sep.view(B, -1, heads * 64)[:, 1:], out.view(B, -1, heads * 64)[:, 1:])
    assert_close_bf16(sep.view(B, -1, heads * 64)[:, 0], ref.view(B, -1, heads * 64)[:, 0], 1.2e-2, "cls-separate")
    assert_close_bf16(out.view(B, -1, heads * 64)[:, 0], ref.view(B, -1, heads * 64)[:, 0], 1.2e-2, "cls-folded")
    assert_close_bf16(out, ref, 1.2e-2, f"attn-{mode}")
    # row-wise check so a wrong small-magnitude row cannot hide behind the global scale
    err = (out.float().cpu() - ref).abs().amax(1)
    scale = ref.abs().amax(1) + 1e-3
    assert (err / scale).max() < 5e-2


@pytest.mark.parametrize("n,blocks_per_wave", [(256, 4), (576, 3), (128, 2), (512, 4), (384, 3)])
def test_space_attention_joint_blocks_equal_the_16_query_kernel(n, blocks_per_wave):
    """hh_space_attn_fwd picks the joint-block kernel (a wave owns 4 / 3 / 2 query blocks at once and shares every K / V fragment
    between them) when n / 16 divides by 4 waves x blocks; the 16-query kernel is the generic path.  Same arithmetic per block
    (reference maximum from the block's first 32 keys, bf16 probabilities, MFMA row sums); only the pairing of key tiles inside the
    PV MFMAs differs (chunks of 2 / 6 tiles instead of 9), i.e. the fp32 summation order: outputs agree to one bf16 rounding."""
    B, T, heads = 2, 2, 2
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=n)
    qkv[:, :D] *= 0.6 * ops.LOG2E
    qkv = bf(qkv).to(DEV)
    try:
        ops.set_tuning("space_joint", 0)
        generic = ops.divided_attention(qkv, B, T, n, heads, "space")
        ops.set_tuning("space_joint", 1)
        ops.set_tuning("space_mfma32", 0)          # (round 6: n <= 256 defaults to the 32x32x16 kernels, tested below)
        joint = ops.divided_attention(qkv, B, T, n, heads, "space")
    finally:
        ops.set_tuning("space_joint", 1)
        ops.set_tuning("space_mfma32", 1)
    # (the CLS rows also merge 4 instead of 8 per-wave partial softmax records; at n = 128 the 16-query kernel has no fast path --
    # its first chunk would hold the CLS tile -- and runs the running-maximum softmax: probabilities round differently)
    assert_close_bf16(joint, generic, 8e-3, "joint vs 16-query")
    if n > 128:
        torch.testing.assert_close(joint.float(), generic.float(), rtol=2.0 ** -7, atol=1e-3 * float(generic.float().abs().max()))
        assert (joint != generic).float().mean() < 0.02
    assert_close_bf16(joint, _ref_divided(qkv.cpu(), B, T, n, heads, "space"), 1.2e-2, "attn-space-joint")


def test_space_attention_joint_kernel_many_problems():
    """B*T*heads = 3*6*16 = 288 (clip, frame, head) problems, folded CLS partials, against the 16-query kernel."""
    B, T, n, heads = 3, 6, 256, 16
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=77)
    qkv[:, :D] *= 0.5 * ops.LOG2E
    qkv = bf(qkv).to(DEV)
    try:
        ops.set_tuning("space_mfma32", 0)
        out = ops.divided_attention(qkv, B, T, n, heads, "space")
        ops.set_tuning("space_joint", 0)
        generic = ops.divided_attention(qkv, B, T, n, heads, "space")
    finally:
        ops.set_tuning("space_joint", 1)
        ops.set_tuning("space_mfma32", 1)
    assert_close_bf16(out, generic, 8e-3, "joint, many problems")
    torch.testing.assert_close(out.float(), generic.float(), rtol=2.0 ** -7, atol=1e-3 * float(generic.float().abs().max()))


@pytest.mark.parametrize("n", [256, 192, 128, 64])
def test_space_attention_32x32_kernels_vs_fp32_and_vs_the_16_query_kernel(n):
    """Round 6: n <= 256 (n % 64 == 0) runs on v_mfma_f32_32x32x16_bf16 with the exponentials software-pipelined under the matrix core and
    NO reference maximum (base-2 logits, P = exp2(s); csrc/attn_space32.hip) -- hh_set_tuning("space_mfma32", 1) (default) = one problem per
    workgroup, 2 = the persistent wave-specialised kernel at n = 256.  Against the fp32 reference (same bound as every space kernel), against the
    16-query kernel (different rounding of P: a reference maximum there, none here -- one bf16 ulp of the output), the two 32x32 forms
    bit-identical (same arithmetic, different data movement), and 288 problems so that persistent workgroups walk more than one."""
    B, T, heads = 3, 6, 16
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=n + 5)
    qkv[:, :D] *= 0.6 * ops.LOG2E
    qkv[7, :64] += 5.0
    qkv = bf(qkv).to(DEV)
    try:
        ops.set_tuning("space_joint", 0)
        generic = ops.divided_attention(qkv, B, T, n, heads, "space")
        ops.set_tuning("space_joint", 1)
        ops.set_tuning("space_mfma32", 1)
        one = ops.divided_attention(qkv, B, T, n, heads, "space")
        one_rev = ops.divided_attention(qkv, B, T, n, heads, "space", reverse=True)
        ops.set_tuning("space_mfma32", 2)
        per = ops.divided_attention(qkv, B, T, n, heads, "space")
        rev = ops.divided_attention(qkv, B, T, n, heads, "space", reverse=True)
        planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
        per_planes = ops.divided_attention(planes, B, T, n, heads, "space")
    finally:
        ops.set_tuning("space_joint", 1)
        ops.set_tuning("space_mfma32", 1)
    assert torch.equal(one, per) and torch.equal(per, rev) and torch.equal(per, per_planes) and torch.equal(one, one_rev)
    ref = _ref_divided(qkv.cpu(), B, T, n, heads, "space")
    assert_close_bf16(per, ref, 1.2e-2, "attn-space-32x32")
    err = (per.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2
    assert_close_bf16(per, generic, 8e-3, "32x32 vs 16-query")


def test_space_attention_third_step_kernel_n576():
    """Round 6, opt-in (hh_set_tuning("space_mfma32", 2 | 3)): config 4's n = 576 on the third-step pipelined 16x16x32 kernel -- three 16-query blocks
    per wave whose QK -> exp -> PV chains run a third of a chunk apart, no reference maximum -- against the fp32 reference and the default
    progressive kernel (one bf16 ulp: P is rounded without / with a reference maximum), CLS records bit-equal (same partial routine), and
    its running-maximum redo."""
    B, T, n, heads = 2, 3, 576, 2
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=57)
    qkv[:, :D] *= 0.6 * ops.LOG2E
    qkv[9, :64] += 5.0
    qkv[1 + n + 7, :64] *= 40.0                                    # a query whose logits leave what exp2 carries without a reference
    qkv = bf(qkv).to(DEV)
    dflt = ops.divided_attention(qkv, B, T, n, heads, "space")
    try:
        ops.set_tuning("space_mfma32", 2)                     # progressive staging: the walk starts on the first key segment while the rest is in flight
        ops.space_redo_count(reset=True)
        third = ops.divided_attention(qkv, B, T, n, heads, "space")
        rev = ops.divided_attention(qkv, B, T, n, heads, "space", reverse=True)
        redone = ops.space_redo_count(reset=True)
        planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous()
        third_planes = ops.divided_attention(planes, B, T, n, heads, "space")
        again = [ops.divided_attention(planes, B, T, n, heads, "space") for _ in range(8)]     # (a race on a segment boundary would be timing-dependent)
        ops.set_tuning("space_mfma32", 3)                     # plain staging: the same arithmetic
        plain = ops.divided_attention(qkv, B, T, n, heads, "space")
    finally:
        ops.set_tuning("space_mfma32", 1)
    assert redone >= 1 and torch.equal(third, rev) and torch.equal(third, third_planes) and torch.equal(third, plain)
    assert all(torch.equal(a, third) for a in again)
    ref = _ref_divided(qkv.cpu(), B, T, n, heads, "space")
    assert_close_bf16(third, ref, 1.2e-2, "attn-space-third-step")
    err = (third.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2
    assert_close_bf16(third, dflt, 8e-3, "third-step vs progressive")
    assert torch.equal(third.view(B, N, D)[:, 0], dflt.view(B, N, D)[:, 0])


@pytest.mark.parametrize("joint,mfma32", [(1, 1), (1, 2), (1, 0), (0, 0)])
def test_space_attention_redo_path_when_the_reference_maximum_is_exceeded(joint, mfma32):
    """The fast path of the 16x16x32 space kernels fixes one reference maximum per 16-query block (from the first 32 keys) and redoes the block
    with a running maximum when a later score exceeds it by more than 2^127; the 32x32x16 kernels use no reference and redo a 32-query block
    whose row sum leaves [2^-100, 2^100]: plant such keys / queries and compare with the fp32 reference."""
    ops.set_tuning("space_joint", joint)
    ops.set_tuning("space_mfma32", mfma32)
    B, T, n, heads = 1, 2, 256, 2
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=3)
    qkv[:, :D] *= 0.3
    q0 = qkv[1 + 40, :64].clone()                                  # query 40 of frame 0, head 0
    qkv[1 + 200, D:D + 64] = q0 * (160.0 / float(q0 @ q0))        # key 200 (beyond the first chunk): logit ~ +160 for that query
    qkv[1 + n + 7, :64] *= 30.0                                    # frame 1: a query with huge logits everywhere
    qkv[1 + n + 100, :64] *= 30.0                                  # ... and a second one, whose whole score row is then shifted to about -600:
    qkv[1 + n + 100, :64] -= 12.0 * qkv[1 + n:1 + 2 * n, D:D + 64].mean(0) / float((qkv[1 + n:1 + 2 * n, D:D + 64].mean(0) ** 2).sum()) ** 0.5
    qkv[:, :D] *= ops.LOG2E
    qkv = bf(qkv)
    out = ops.divided_attention(qkv.to(DEV), B, T, n, heads, "space")
    ref = _ref_divided(qkv, B, T, n, heads, "space")
    ops.set_tuning("space_joint", 1)
    ops.set_tuning("space_mfma32", 1)
    assert torch.isfinite(out.float()).all()
    assert_close_bf16(out, ref, 1.2e-2, "attn-space-redo")
    err = (out.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2


@pytest.mark.parametrize("B,T,n,heads", [(2, 3, 64, 2), (1, 5, 100, 1), (2, 12, 256, 2), (1, 7, 33, 3), (1, 9, 196, 2), (1, 20, 64, 2), (2, 24, 37, 1), (1, 31, 196, 2), (1, 17, 5, 1)])
def test_time_attention_any_frame_count(B, T, n, heads):
    """Round 6: hh_time_attn_fwd for every T <= 32, not only powers of two -- the tile holds the next power of two of frame
    slots, the padding rows' keys are masked, their queries never stored, the CLS query's folded partial skips them.  Against the fp32
    reference, token-major and head-major planes, forward and reversed walk, folded and stand-alone CLS row; rows the kernel must not
    write (there are none besides row 0 of each clip, which hh_cls_combine writes) are checked through a poisoned output buffer."""
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=300 + T + n)
    qkv[:, :D] *= 0.6
    qkv[3, :64] += 5.0
    qkv = bf(qkv)
    ref = _ref_divided(qkv, B, T, n, heads, "time")
    poison = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    out = ops.divided_attention(qkv.to(DEV), B, T, n, heads, "time", out=poison.clone())
    assert torch.isfinite(out.float()).all()
    assert_close_bf16(out, ref, 1.2e-2, f"attn-time-T{T}")
    err = (out.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2
    planes = qkv.view(B * N, 3 * heads, 64).transpose(0, 1).contiguous().to(DEV)
    assert torch.equal(ops.divided_attention(planes, B, T, n, heads, "time"), out)
    assert torch.equal(ops.divided_attention(qkv.to(DEV), B, T, n, heads, "time", reverse=True), out)
    sep = ops.divided_attention(qkv.to(DEV), B, T, n, heads, "time", fold_cls=False)
    assert torch.equal(sep.view(B, N, D)[:, 1:], out.view(B, N, D)[:, 1:])
    assert_close_bf16(sep.view(B, N, D)[:, 0], out.view(B, N, D)[:, 0], 8e-3, "cls folded vs stand-alone")


@pytest.mark.parametrize("B,T,n,heads", [(2, 4, 50, 2), (1, 16, 37, 1), (2, 8, 96, 3), (1, 1, 70, 2), (1, 2, 5, 1), (3, 16, 256, 2),
                                          (2, 32, 30, 2), (1, 32, 7, 1), (1, 32, 576, 2)])
def test_time_attention_ragged_patch_counts(B, T, n, heads):
    """The MFMA time kernel packs 16/T patch locations per 16-row tile and 128/T per wave: patch counts that are not
    multiples of either (partly empty tiles, clamped rows), block-diagonal masking for T < 16, several CLS records."""
    N, D = 1 + T * n, heads * 64
    qkv = rnd(B * N, 3 * D, seed=100 + T + n)
    qkv[:, :D] *= 0.6
    qkv[N - 1, :64] += 5.0
    qkv = bf(qkv)
    out = ops.divided_attention(qkv.to(DEV), B, T, n, heads, "time")
    ref = _ref_divided(qkv, B, T, n, heads, "time")
    assert_close_bf16(out, ref, 1.2e-2, "attn-time-ragged")
    err = (out.float().cpu() - ref).abs().amax(1)
    assert (err / (ref.abs().amax(1) + 1e-3)).max() < 5e-2


@pytest.mark.parametrize("B,Q,M,heads", [(2, 13, 4096, 8), (3, 5, 1024, 8), (1, 16, 96, 2), (1, 13, 18432, 8)])   # last: config 4 (32 x 576 keys)
def test_xattn_fwd_bwd(B, Q, M, heads):
    """Cross-attention core vs fp32 PyTorch on the same fp32 q / dO and bf16 K / V.  Query-side operands enter the MFMAs as bf16
    hi + lo pairs, so the result is exact with respect to the stored K / V: fp32 outputs agree to ~1e-4 of their scale (the old
    single-bf16 operands gave 1e-2); dK / dV are stored in bf16 (one rounding, 8e-3)."""
    C = heads * 64
    q = rnd(B, Q, C, seed=1, scale=0.3)
    kv = bf(rnd(B, M, 2 * C, seed=2))
    dout = rnd(B, Q, C, seed=3)
    Kd = kv.to(DEV)
    k, v = Kd[:, :, :C], Kd[:, :, C:]
    out, lse = ops.xattn_fwd(q.to(DEV), k, v, heads)
    qb = q.clone().requires_grad_(True)
    kr = kv[:, :, :C].float().requires_grad_(True)
    vr = kv[:, :, C:].float().requires_grad_(True)
    qh = qb.view(B, Q, heads, 64).transpose(1, 2)
    kh = kr.view(B, M, heads, 64).transpose(1, 2)
    vh = vr.view(B, M, heads, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Q, C)
    assert_close_bf16(out, ref.detach(), 3e-4, "xattn out")
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(s, -1).detach(), rtol=1e-4, atol=1e-4)
    ref.backward(dout)
    dkv = torch.zeros_like(Kd)
    dq = ops.xattn_bwd(q.to(DEV), k, v, out, lse, dout.to(DEV), dkv[:, :, :C], dkv[:, :, C:], heads)
    assert_close_bf16(dq, qb.grad, 1e-3, "dq")
    assert_close_bf16(dkv[:, :, :C], kr.grad, 8e-3, "dk")
    assert_close_bf16(dkv[:, :, C:], vr.grad, 8e-3, "dv")


@pytest.mark.parametrize("B,M,splits", [(2, 4096, 8), (1, 18432, 32), (3, 1184, 4), (2, 160, 64)])
def test_xattn_fwd_key_slices_equal_the_single_pass(B, M, splits):
    """hh_xattn_fwd_split (one workgroup per (clip, head, key slice) + merge) against hh_xattn_fwd on the same inputs, with and
    without dropout: the slices see the same (seed, clip, head, query, key) mask, so the outputs differ only by fp32 summation order.
    (3, 1184, 4): ragged last slice; (2, 160, 64): more slices asked for than 128-key rounds exist."""
    heads, Q = 8, 13
    C = heads * 64
    q = rnd(B, Q, C, seed=4, scale=0.3).to(DEV)
    kv = bf(rnd(B, M, 2 * C, seed=5)).to(DEV)
    k, v = kv[:, :, :C], kv[:, :, C:]
    for p, seed in ((0.0, 0), (0.1, 1234)):
        one, lse1 = ops.xattn_fwd(q, k, v, heads, dropout_p=p, seed=seed, splits=1)
        many, lse2 = ops.xattn_fwd(q, k, v, heads, dropout_p=p, seed=seed, splits=splits)
        torch.testing.assert_close(lse2, lse1, rtol=1e-5, atol=1e-5)
        assert_close_bf16(many, one.cpu(), 2e-5, "xattn split p=%g" % p)
    if M >= 1024:
        dflt, _ = ops.xattn_fwd(q, k, v, heads)                              # the default takes the sliced path at this B*heads
        assert_close_bf16(dflt, ops.xattn_fwd(q, k, v, heads, splits=1)[0], 2e-5, "xattn default split")


def _mattn_ref(qt, mp, mem, wv, bv, Q):
    """fp64 reference of the memory-space cross-attention incl. the value projection: qt [B*Q, 8*512], mp / mem [B, M, 512] ->
    (pooled [B*Q, 8*512], lse2 [B*Q, 8], ca [B*Q, 512])."""
    B, M, C = mp.shape
    H = 8
    q4 = qt.view(B, Q, H, C)
    s = torch.einsum("bqhc,bmc->bqhm", q4, mp)
    p = torch.softmax(s, -1)
    pooled = torch.einsum("bqhm,bmc->bqhc", p, mem)
    lse2 = torch.logsumexp(s, -1) * 1.4426950408889634
    ca = torch.einsum("bqhc,hnc->bqhn", pooled, wv.view(H, 64, C)) + bv.view(H, 64)
    return pooled.reshape(B * Q, H * C), lse2.reshape(B * Q, H), ca.reshape(B * Q, H * 64)


@pytest.mark.parametrize("B,Q,M,slices", [(2, 13, 4096, None), (3, 5, 1024, 1), (1, 16, 128, None), (1, 13, 18432, None), (2, 13, 1184, 3), (5, 13, 256, 2)])
def test_mattn_fwd_bwd_vs_torch(B, Q, M, slices):
    """hh_mattn_fwd / hh_mattn_bwd (cross-attention in memory space, d = 512: csrc/mattn.hip) + the batched d-memory GEMM against fp64
    PyTorch on the same fp32 query-side values and bf16 memory rows.  Query-side operands enter the MFMAs as bf16 hi + lo pairs, so
    pooled / lse2 / dqt are exact with respect to the stored rows (fp32-grade: <= 3e-5 of the scale); d mem / d mp go through bf16
    Pd^T / dS^T (one rounding: 8e-3).  (1, 13, 18432): config 4's key count; (2, 13, 1184, 3): ragged last slice; (5, 13, 256, 2):
    the clip count is not a multiple of the 8 workgroup-units a grid round holds."""
    H, C = 8, 512
    R = B * Q
    qt = rnd(R, H * C, seed=1, scale=0.08)
    mpm = bf(rnd(2, B, M, C, seed=2))
    wv, bv = rnd(C, C, seed=3, scale=0.05), rnd(C, seed=4, scale=0.1)
    G = rnd(R, C, seed=5)
    mp, mem = mpm[0].to(DEV), mpm[1].to(DEV)
    pooled, lse2, rsum = ops.mattn_fwd(qt.to(DEV), mp, mem, Q, slices=slices)
    qr = qt.double().requires_grad_(True)
    mpr, memr = mpm[0].double().requires_grad_(True), mpm[1].double().requires_grad_(True)
    rp, rl, rca = _mattn_ref(qr, mpr, memr, wv.double(), bv.double(), Q)
    assert_close_bf16(pooled, rp.detach(), 3e-5, "mattn pooled")
    torch.testing.assert_close(lse2.cpu().double(), rl.detach(), rtol=0, atol=2e-4)
    assert float((rsum.cpu() - 1).abs().max()) < 1e-5
    # value projection + bias on the pooled rows (head-batched NT), then the backward chain
    ca = ops.head_map_out(pooled, wv.to(DEV), bias=bv.to(DEV))
    assert_close_bf16(ca, rca.detach(), 3e-5, "mattn ca")
    (rca * G.double()).sum().backward()
    dca = G.to(DEV)
    dpooled = ops.head_map_in(dca, wv.to(DEV))
    L = 2
    rows = L * 128
    pdT = torch.full((B, rows, M), float("nan"), dtype=torch.bfloat16, device=DEV)
    dsT = torch.full_like(pdT, float("nan"))
    qt16 = torch.full((B, rows, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    dp16 = torch.full_like(qt16, float("nan"))
    for l in range(L):                                  # two "layers" with the same inputs: every row of the operand buffers gets written
        dqt = ops.mattn_bwd(qt.to(DEV), dpooled, lse2, dca, ca, bv.to(DEV), mp, mem, Q, pdT, dsT, qt16, dp16, l * 128, slices=slices)
    assert_close_bf16(dqt, qr.grad, 1e-4, "mattn dqt")
    for t in (pdT, dsT, qt16, dp16):
        assert torch.isfinite(t.float()).all()
    assert float(pdT.view(B, L, H, 16, M)[:, :, :, Q:].abs().max() if Q < 16 else 0.0) == 0.0           # query slots beyond Q are zero rows
    if M % 128 == 0:
        dmem = ops.gemm_tn_batched2(pdT, dp16, dsT, qt16)                                                 # [B, M, C] = d mem + d mp, L times
        assert_close_bf16(dmem, L * (mpr.grad + memr.grad), 8e-3, "mattn d(mem) + d(mp)")
        dmp = ops.gemm_tn_batched2(dsT, qt16)
        assert_close_bf16(dmp, L * mpr.grad, 8e-3, "mattn d(mp)")
        one = ops.gemm_tn_batched2(pdT[:, :128].contiguous(), dp16[:, :128].contiguous())
        assert_close_bf16(one, memr.grad, 8e-3, "mattn d(mem), one layer")


@pytest.mark.parametrize("B,Q,M,Mv,slices", [(2, 13, 640, 588, None), (1, 13, 512, 500, 16), (2, 5, 3200, 3136, None), (1, 16, 128, 97, 1)])
def test_mattn_masks_padding_keys(B, Q, M, Mv, slices):
    """Round 6: keys [keys_valid, M) are padding rows (any finite content -- here large values that would dominate the softmax if they were
    seen): pooled / lse2 / dqt equal the fp64 reference over the first keys_valid keys, the padding columns of Pd^T / dS^T are exact zeros
    (so the batched d-memory GEMM gives their rows exact zeros), also when a whole key slice is padding (M = 512, 16 slices of 32 keys)."""
    H, C = 8, 512
    R = B * Q
    qt = rnd(R, H * C, seed=31, scale=0.08)
    mpm = bf(rnd(2, B, M, C, seed=32))
    mpm[:, :, Mv:] = 3.0                                   # padding rows: must not be seen
    wv, bv = rnd(C, C, seed=33, scale=0.05), rnd(C, seed=34, scale=0.1)
    G = rnd(R, C, seed=35)
    mp, mem = mpm[0].to(DEV), mpm[1].to(DEV)
    pooled, lse2, rsum = ops.mattn_fwd(qt.to(DEV), mp, mem, Q, slices=slices, keys_valid=Mv)
    qr = qt.double().requires_grad_(True)
    mpr, memr = mpm[0][:, :Mv].double().requires_grad_(True), mpm[1][:, :Mv].double().requires_grad_(True)
    rp, rl, rca = _mattn_ref(qr, mpr, memr, wv.double(), bv.double(), Q)
    assert torch.isfinite(pooled).all() and torch.isfinite(lse2).all()
    assert_close_bf16(pooled, rp.detach(), 3e-5, "mattn pooled, masked padding")
    torch.testing.assert_close(lse2.cpu().double(), rl.detach(), rtol=0, atol=2e-4)
    ca = ops.head_map_out(pooled, wv.to(DEV), bias=bv.to(DEV))
    (rca * G.double()).sum().backward()
    dca = G.to(DEV)
    dpooled = ops.head_map_in(dca, wv.to(DEV))
    pdT = torch.full((B, 128, M), float("nan"), dtype=torch.bfloat16, device=DEV)
    dsT = torch.full_like(pdT, float("nan"))
    qt16 = torch.full((B, 128, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    dp16 = torch.full_like(qt16, float("nan"))
    dqt = ops.mattn_bwd(qt.to(DEV), dpooled, lse2, dca, ca, bv.to(DEV), mp, mem, Q, pdT, dsT, qt16, dp16, 0, slices=slices, keys_valid=Mv)
    assert_close_bf16(dqt, qr.grad, 1e-4, "mattn dqt, masked padding")
    assert float(pdT[:, :, Mv:].float().abs().max()) == 0.0 and float(dsT[:, :, Mv:].float().abs().max()) == 0.0
    if M % 128 == 0:
        dmem = ops.gemm_tn_batched2(pdT, dp16, dsT, qt16)
        assert float(dmem[:, Mv:].abs().max()) == 0.0
        assert_close_bf16(dmem[:, :Mv], mpr.grad + memr.grad, 8e-3, "mattn d(mem) + d(mp), masked padding")
    # no ticket row for the last-arriver fold (more than 32 launch streams in the process) -> one slice, not an error
    try:
        ops.set_tuning("mattn_no_ticket", 1)
        p1, l1, _ = ops.mattn_fwd(qt.to(DEV), mp, mem, Q, slices=slices, keys_valid=Mv)
        d1 = ops.mattn_bwd(qt.to(DEV), dpooled, lse2, dca, ca, bv.to(DEV), mp, mem, Q, pdT, dsT, qt16, dp16, 0, slices=slices, keys_valid=Mv)
    finally:
        ops.set_tuning("mattn_no_ticket", 0)
    assert_close_bf16(p1, pooled, 2e-5, "mattn one slice (no ticket row)")
    assert_close_bf16(d1, dqt, 2e-5, "mattn bwd one slice (no ticket row)")


def test_mattn_key_slices_and_dropout():
    """Key slices (one workgroup pair per (clip, slice) + merge) against the single pass, with and without dropout -- the slices see the
    same (seed, clip, head, query, key) mask; with dropout rsum is the kept probability mass, and the analytic d qt matches a central
    finite difference of the (deterministic, same-mask) forward."""
    B, Q, M, H, C = 2, 13, 2048, 8, 512
    qt = rnd(B * Q, H * C, seed=7, scale=0.08).to(DEV)
    mpm = bf(rnd(2, B, M, C, seed=8)).to(DEV)
    mp, mem = mpm[0], mpm[1]
    for p, seed in ((0.0, 0), (0.1, 4321)):
        one = ops.mattn_fwd(qt, mp, mem, Q, p, seed, slices=1)
        many = ops.mattn_fwd(qt, mp, mem, Q, p, seed, slices=16)
        assert_close_bf16(many[0], one[0], 2e-5, "mattn slices p=%g" % p)
        torch.testing.assert_close(many[1], one[1], rtol=0, atol=2e-5)
        torch.testing.assert_close(many[2], one[2], rtol=1e-5, atol=1e-6)
    pooled, lse2, rsum = ops.mattn_fwd(qt, mp, mem, Q, 0.1, 77)
    assert 0.8 < float(rsum.mean()) < 1.2 and float((rsum - 1).abs().max()) > 1e-3
    assert float((ops.mattn_fwd(qt, mp, mem, Q, 0.1, 78)[0] - pooled).abs().max()) > 0                    # another seed, another mask
    # backward with dropout: finite difference along a random direction (value projection with bias: the rsum term is live)
    wv, bv = rnd(C, C, seed=9, scale=0.05).to(DEV), rnd(C, seed=10, scale=0.1).to(DEV)
    G = rnd(B * Q, C, seed=11).to(DEV)

    def f(x):
        po, _, rs = ops.mattn_fwd(x, mp, mem, Q, 0.1, 77)
        return ops.head_map_out(po, wv, bias=bv, rowscale=rs)
    ca = f(qt)
    dpooled = ops.head_map_in(G, wv)
    bufs = [torch.empty((B, 128, M), dtype=torch.bfloat16, device=DEV) for _ in range(2)] + [torch.empty((B, 128, C), dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    dqt = ops.mattn_bwd(qt, dpooled, lse2, G, ca, bv, mp, mem, Q, *bufs, 0, 0.1, 77)
    d = rnd(B * Q, H * C, seed=12).to(DEV)
    eps = 2e-3
    fd = float((((f(qt + eps * d) - f(qt - eps * d)) / (2 * eps)) * G).double().sum())
    an = float((dqt * d).double().sum())
    assert abs(fd - an) <= 0.03 * abs(fd) + 1e-3, (fd, an)


def test_mattn_slice_handoff_is_bit_stable_under_uneven_load():
    """The key slices of hh_mattn_fwd / hh_mattn_bwd are folded by the LAST workgroup of a (clip, head group) to finish (agent-scope
    release -> ticket -> acquire, csrc/mattn.hip: ma_last_arriver) -- an inter-workgroup hand-off, which fails under uneven load if a
    fence is missing.  40 rounds beside a stream of large GEMMs (the tower beside the decoder, as in the pipelined step), consumer
    caches warm: every word of pooled / lse2 / dqt must equal the first round's (fixed summation order) and the single-slice result to
    fp32 re-association."""
    B, Q, M, H, C = 6, 13, 4096, 8, 512
    qt = rnd(B * Q, H * C, seed=21, scale=0.08).to(DEV)
    mpm = bf(rnd(2, B, M, C, seed=22)).to(DEV)
    mp, mem = mpm[0], mpm[1]
    wv, bv = rnd(C, C, seed=23, scale=0.05).to(DEV), rnd(C, seed=24, scale=0.1).to(DEV)
    G = rnd(B * Q, C, seed=25).to(DEV)
    bufs = [torch.empty((B, 128, M), dtype=torch.bfloat16, device=DEV) for _ in range(2)] + [torch.empty((B, 128, C), dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    a = bf(rnd(8192, 1024, seed=26)).to(DEV)
    w = bf(rnd(4096, 1024, seed=27, scale=0.05)).to(DEV)
    side = torch.cuda.Stream()
    one = ops.mattn_fwd(qt, mp, mem, Q, slices=1)
    first = None
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(1 + it % 3):
                ops.gemm(a, w)
        pooled, lse2, rsum = ops.mattn_fwd(qt, mp, mem, Q, slices=16)
        ca = ops.head_map_out(pooled, wv, bias=bv)
        dqt = ops.mattn_bwd(qt, ops.head_map_in(G, wv), lse2, G, ca, bv, mp, mem, Q, *bufs, 0, slices=8)
        cur = (pooled.clone(), lse2.clone(), dqt.clone())
        if first is None:
            first = cur
            assert_close_bf16(pooled, one[0], 2e-5, "slices vs single pass")
        else:
            assert all(torch.equal(x, y) for x, y in zip(cur, first)), it
    torch.cuda.synchronize()


def test_head_batched_qgemm_vs_torch():
    """The three head-batched hh_qgemm_f32x3 forms of the memory-space cross-attention (ops.head_map_in / _out / _wgrad) on row-strided
    views of a packed [q; k; v] in-projection, against fp64 PyTorch (fp32-grade: 2e-5)."""
    R, H, C = 37, 8, 512
    w3 = rnd(3 * C, C, seed=1, scale=0.05).to(DEV)
    b3 = rnd(3 * C, seed=2, scale=0.1).to(DEV)
    x = rnd(R, C, seed=3).to(DEV)
    y = rnd(R, H * C, seed=4).to(DEV)
    rs = (rnd(R, H, seed=5).abs() + 0.5).to(DEV)
    wk, wv = w3[C:2 * C], w3[2 * C:]
    got = ops.head_map_in(x, wk)
    ref = torch.einsum("rhn,hnc->rhc", x.double().view(R, H, 64), wk.double().view(H, 64, C)).reshape(R, H * C)
    assert_close_bf16(got, ref, 2e-5, "head_map_in")
    got = ops.head_map_out(y, wv, bias=b3[2 * C:], rowscale=rs)
    ref = torch.einsum("rhc,hnc->rhn", y.double().view(R, H, C), wv.double().view(H, 64, C)) + b3[2 * C:].double().view(H, 64) * rs.double()[:, :, None]
    assert_close_bf16(got, ref.reshape(R, C), 2e-5, "head_map_out")
    got = ops.head_map_out(y, wv)
    assert_close_bf16(got, torch.einsum("rhc,hnc->rhn", y.double().view(R, H, C), wv.double().view(H, 64, C)).reshape(R, C), 2e-5, "head_map_out, no bias")
    gw = torch.zeros((3 * C, C), dtype=torch.float32, device=DEV)
    gb = torch.zeros(3 * C, dtype=torch.float32, device=DEV)
    ops.head_map_wgrad(x, y, gw[2 * C:], colsum=gb[2 * C:], rowscale=rs)
    ref = torch.einsum("rhn,rhc->hnc", x.double().view(R, H, 64), y.double().view(R, H, C)).reshape(C, C)
    assert_close_bf16(gw[2 * C:], ref, 2e-5, "head_map_wgrad")
    assert_close_bf16(gb[2 * C:], (x.double().view(R, H, 64) * rs.double()[:, :, None]).sum(0).reshape(C), 2e-5, "head_map_wgrad colsum")
    assert float(gw[:2 * C].abs().max()) == 0 and float(gb[:2 * C].abs().max()) == 0
    ops.head_map_wgrad(x, y, gw[C:2 * C], colsum=gb[C:2 * C])
    assert_close_bf16(gb[C:2 * C], x.double().sum(0), 2e-5, "plain colsum")


def test_match_boxes_bit_exact_and_losses():
    from helping_hand_for_egocentric_videos_amd import synth, TINY4
    from oracle import losses as OL
    g = torch.Generator().manual_seed(5)
    for trial, (q0, q) in enumerate([(0, 2), (2, 10), (0, 2), (2, 2), (1, 12)]):
        Qtot = 13
        F_ = 64
        pred = torch.rand(F_, Qtot, 4, generator=g) * 0.4 + 0.2
        if trial == 2:                      # near-duplicate predictions: tie-ish costs
            pred[:, 1] = pred[:, 0] + 1e-7
        raw = synth.make_batch(TINY4.with_(num_frames=16), 4, seed=trial)["boxes"][:, :, :2].flatten(0, 1)   # [64,2,4]
        if trial == 3:
            raw = raw.repeat(1, 2, 1)[:, :3] + torch.rand(F_, 3, 4, generator=g)    # up to 3 targets vs q=2 (wide)
            raw[..., 2:] = raw[..., :2] + raw[..., 2:].abs() + 1
        tg = OL.prepare_targets(raw)
        pb = pred[:, q0:q0 + q]
        ref_idx = OL.hungarian_match(pb, tg)
        m = ops.match_boxes(pred.to(DEV), q0, q, raw.to(DEV).contiguous())
        cnt = m["count"].cpu()
        mp, mt, mn = m["pred_idx"].cpu(), m["tgt_idx"].cpu(), m["n"].cpu()
        assert mp.dtype == torch.int64 and mt.dtype == torch.int64
        for f in range(F_):
            assert int(cnt[f]) == len(tg[f])
            torch.testing.assert_close(m["tgt"][f, :len(tg[f])].cpu(), tg[f], rtol=0, atol=0)
            r, c = ref_idx[f]
            assert int(mn[f]) == len(r)
            assert torch.equal(mp[f, :len(r)], r) and torch.equal(mt[f, :len(r)], c), (trial, f)
        # losses + gradient
        pr = pred.clone().requires_grad_(True)
        l1, gi, _ = OL.box_losses(pr[:, q0:q0 + q], tg, ref_idx)
        nb = max(float(sum(len(t) for t in tg)), 1.0)
        sums = ops.box_loss_fwd(pred.to(DEV), q0, m).cpu()
        torch.testing.assert_close(sums[0] / nb, l1.detach(), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(sums[1] / nb, gi.detach(), rtol=1e-4, atol=1e-6)
        (3.75 * l1 + 1.5 * gi).backward()
        dpred = torch.zeros_like(pred, device=DEV)
        ops.box_loss_bwd(pred.to(DEV), q0, m, torch.tensor([3.75 / nb], device=DEV), torch.tensor([1.5 / nb], device=DEV), dpred)
        torch.testing.assert_close(dpred.cpu(), pr.grad, rtol=1e-3, atol=1e-5)


def test_lsap_rows_matches_scipy_golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lsap_scipy.npz"))
    co, ro = 0, 0
    for (nr, nc), n in zip(g["shapes"], g["lens"]):
        c = g["costs"][co:co + nr * nc].reshape(nr, nc)
        co += nr * nc
        rows, cols = g["rows"][ro:ro + n], g["cols"][ro:ro + n]
        ro += n
        if nr == 0 or nc == 0 or nr > 16 or nc > 16 or not np.array_equal(c.astype(np.float32).astype(np.float64), c):
            continue
        out = ops.lsap_rows(torch.tensor(c, dtype=torch.float32, device=DEV)[None], torch.ones(1, nr, dtype=torch.uint8, device=DEV)).cpu()[0]
        got = {(r, int(out[r])) for r in range(nr) if out[r] >= 0}
        assert got == set(zip(rows.tolist(), cols.tolist()))
    # masked rows
    cost = torch.rand(7, 4, 12, generator=torch.Generator().manual_seed(1))
    valid = torch.tensor([[1, 1, 0, 0], [1, 0, 1, 0], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 1, 1], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=torch.uint8)
    from oracle.lsap import linear_sum_assignment
    out = ops.lsap_rows(cost.to(DEV), valid.to(DEV)).cpu()
    for p in range(7):
        rows = [r for r in range(4) if valid[p, r]]
        if rows:
            _, cols = linear_sum_assignment(cost[p][rows].numpy())
            assert [int(out[p, r]) for r in rows] == cols.tolist()
        assert all(int(out[p, r]) == -1 for r in range(4) if not valid[p, r])


def test_adamw_step():
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2) * 1e-3
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=3e-5, weight_decay=1e-5)
    P, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        ref.grad = g * step
        opt.step()
        ops.adamw_step(P, (g * step).to(DEV), m, v, 3e-5, 0.9, 0.999, 1e-8, 1e-5, step)
    torch.testing.assert_close(P.cpu(), ref.detach(), rtol=1e-6, atol=1e-8)


def test_errors_are_loud():
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5)          # CPU tensor
    with pytest.raises(RuntimeError, match="hh_gemm_bf16"):
        ops.gemm(torch.zeros(4, 32, dtype=torch.bfloat16, device=DEV), torch.zeros(128, 32, dtype=torch.bfloat16, device=DEV))


# ---------------------------------------------------------------------------------------------------------------------
# query side of the decoder (csrc/qside.hip): fp32-grade GEMM on the bf16 matrix cores, 13 x 13 self-attention, LayerNorm variants

def _ref_mm(a, b, mode):
    a, b = a.double(), b.double()
    return (a @ b.t()) if mode == ops.NT else (a @ b) if mode == ops.NN else (a.t() @ b)


@pytest.mark.parametrize("mode", [ops.NT, ops.NN, ops.TN], ids=["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K", [(416, 512, 512), (416, 2048, 512), (416, 512, 2048), (10, 4, 512), (65, 256, 768), (26, 1536, 512),
                                   (512, 2048, 416), (4, 512, 40), (130, 68, 36)])
def test_qgemm_fp32_grade_accuracy_all_layouts(mode, M, N, K):
    """hh_qgemm_f32x3 vs an fp64 product of the same fp32 operands: error at fp32 level (1e-5 of the output scale), far below one
    bf16 rounding (4e-3) -- including ragged M / N / K tiles."""
    if mode != ops.TN and K % 4:
        pytest.skip("contiguous contraction dimension must be a multiple of 4")
    if mode == ops.TN and M % 4:
        pytest.skip("TN: M is the contiguous dimension of A")
    sa = (M, K) if mode != ops.TN else (K, M)
    sb = (N, K) if mode == ops.NT else (K, N)
    a, b = rnd(*sa, seed=1), rnd(*sb, seed=2, scale=0.05)
    ref = _ref_mm(a, b, mode)
    out = ops.qgemm(a.to(DEV), b.to(DEV), mode)
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_qgemm_prologue_epilogue_options():
    M, N, K = 72, 132, 96
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    A, W = a.to(DEV), w.to(DEV)
    ref = (a.double() @ w.double().t()).float()
    tol = dict(rtol=1e-4, atol=2e-5 * ref.abs().max().item())
    torch.testing.assert_close(ops.qgemm(A, W, bias=bias.to(DEV)).cpu(), ref + bias, **tol)
    r = ref + bias
    r2 = r.clone(); r2[:, :64] *= 0.125
    torch.testing.assert_close(ops.qgemm(A, W, bias=bias.to(DEV), scale=0.125, scale_ncols=64).cpu(), r2, **tol)
    torch.testing.assert_close(ops.qgemm(A, W, bias=bias.to(DEV), scale=0.125).cpu(), r * 0.125, **tol)
    torch.testing.assert_close(ops.qgemm(A, W, bias=bias.to(DEV), relu=True, resid=res.to(DEV)).cpu(), torch.relu(r) + res, **tol)
    mask = rnd(M, N, seed=5)
    torch.testing.assert_close(ops.qgemm(A, W, relu_mask=mask.to(DEV), mask_scale=1.5).cpu(), torch.where(mask > 0, ref * 1.5, torch.zeros(())), **tol)
    torch.testing.assert_close(ops.qgemm(A, W, a_scale=0.5).cpu(), ref * 0.5, **tol)
    # strided views: output into a column slice, operand from a row slice
    big = torch.zeros(M, 2 * N, device=DEV)
    ops.qgemm(A, W, out=big[:, N:])
    torch.testing.assert_close(big[:, N:].cpu(), ref, **tol)
    assert float(big[:, :N].abs().max()) == 0
    Wb = torch.cat([torch.zeros(3, K), w]).to(DEV)
    torch.testing.assert_close(ops.qgemm(A, Wb[3:]).cpu(), ref, **tol)
    # TN: weight gradient + bias gradient (column sums of dY) in one launch
    dy = rnd(M, N, seed=6)
    cs = torch.empty(N, device=DEV)
    dw = ops.qgemm(dy.to(DEV), A, ops.TN, colsum=cs)
    rdw = (dy.double().t() @ a.double()).float()
    torch.testing.assert_close(dw.cpu(), rdw, rtol=1e-4, atol=2e-5 * rdw.abs().max().item())
    torch.testing.assert_close(cs.cpu(), dy.sum(0), rtol=1e-5, atol=1e-5)
    # split-K (atomic accumulation into a zeroed output) for long contractions with few output tiles
    big_dy, big_x = rnd(6656, 128, seed=8), rnd(6656, 64, seed=9)
    cs2 = torch.zeros(128, device=DEV)
    dw2 = ops.qgemm(big_dy.to(DEV), big_x.to(DEV), ops.TN, colsum=cs2, splitk=13)
    r2w = (big_dy.double().t() @ big_x.double()).float()
    torch.testing.assert_close(dw2.cpu(), r2w, rtol=1e-4, atol=2e-5 * r2w.abs().max().item())
    torch.testing.assert_close(cs2.cpu(), big_dy.sum(0), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="split-K"):
        ops.qgemm(A, W, bias=bias.to(DEV), splitk=2)
    # dropout: the backward's A-prologue regenerates the forward epilogue's mask from (seed, element index)
    p, seed = 0.25, 1234
    y0, yd = ops.qgemm(A, W), ops.qgemm(A, W, drop_p=p, drop_seed=seed)
    keep = yd != 0
    assert 0.70 < float(keep.float().mean()) < 0.80
    torch.testing.assert_close(yd, torch.where(keep, y0 / (1 - p), torch.zeros((), device=DEV)), rtol=1e-5, atol=1e-6)
    assert not torch.equal(keep, ops.qgemm(A, W, drop_p=p, drop_seed=seed + 1) != 0)
    ones, eye = torch.ones(M, N, device=DEV), torch.eye(N, device=DEV)
    m_nn = ops.qgemm(ones, eye, ops.NN, a_drop_p=p, a_drop_seed=seed, a_drop_ld=N)                 # A[m, k] kept iff forward element (m, k) was
    torch.testing.assert_close(m_nn, keep.float() / (1 - p), rtol=1e-5, atol=1e-6)
    m_tn = ops.qgemm(ones, torch.eye(M, device=DEV), ops.TN, a_drop_p=p, a_drop_seed=seed, a_drop_ld=N)   # A[k, m] -> C[m, n] = mask[n, m]
    torch.testing.assert_close(m_tn, (keep.float() / (1 - p)).t().contiguous(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,Q,heads", [(3, 13, 8), (2, 5, 8), (5, 16, 2), (1, 1, 1)])
def test_query_self_attention_fwd_bwd(B, Q, heads):
    C = heads * 64
    qkv = rnd(B * Q, 3 * C, seed=1, scale=0.7)
    dout = rnd(B * Q, C, seed=2)
    x = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, Q, heads, 64).transpose(1, 2) for t in x.view(B * Q, 3, C).unbind(1)]
    ref = (torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * Q, C)
    ref.backward(dout)
    out = ops.qself_attn_fwd(qkv.to(DEV), B, Q, heads)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    dqkv = ops.qself_attn_bwd(qkv.to(DEV), dout.to(DEV), B, Q, heads)
    torch.testing.assert_close(dqkv.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    # dropout: deterministic in the seed, unbiased, and the backward uses the forward's mask (finite differences)
    X = qkv.to(DEV)
    o1, o2 = ops.qself_attn_fwd(X, B, Q, heads, 0.3, 7), ops.qself_attn_fwd(X, B, Q, heads, 0.3, 7)
    assert torch.equal(o1, o2) and (Q * heads * B < 8 or not torch.equal(o1, ops.qself_attn_fwd(X, B, Q, heads, 0.3, 8)))
    acc = sum(ops.qself_attn_fwd(X, B, Q, heads, 0.3, 100 + s) for s in range(200)) / 200
    if Q > 1:
        assert float((acc - out).abs().max()) < 0.25 * float(out.abs().max())
    d = rnd(B * Q, 3 * C, seed=3).to(DEV)
    eps = 1e-2
    fd = float(((ops.qself_attn_fwd(X + eps * d, B, Q, heads, 0.3, 7) - ops.qself_attn_fwd(X - eps * d, B, Q, heads, 0.3, 7)) / (2 * eps) * dout.to(DEV)).sum())
    an = float((ops.qself_attn_bwd(X, dout.to(DEV), B, Q, heads, 0.3, 7) * d).sum())
    assert abs(fd - an) <= 2e-2 * abs(fd) + 1e-3, (fd, an)


def test_layernorm_pos_and_backward_with_residual():
    rows, cols, Q = 39, 512, 13
    x = rnd(rows, cols, seed=1, scale=2.0)
    g, b, pos = rnd(cols, seed=2) * 0.1 + 1, rnd(cols, seed=3) * 0.1, rnd(Q, cols, seed=4)
    ref = torch.nn.functional.layer_norm(x, (cols,), g, b, 1e-5)
    y, y2, mean, rstd = ops.layernorm_pos(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, pos.to(DEV), save_stats=True)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(y2.cpu(), ref + pos.repeat(rows // Q, 1), rtol=1e-4, atol=1e-5)
    yb, yb2 = ops.layernorm_pos(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, pos.to(DEV), out_dtype=torch.bfloat16)
    assert_close_bf16(yb, ref, 5e-3, "ln_pos bf16")
    assert_close_bf16(yb2, ref + pos.repeat(rows // Q, 1), 5e-3, "ln_pos bf16 + pos")
    dy, add = rnd(rows, cols, seed=5), rnd(rows, cols, seed=6)
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-5).backward(dy)
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dx = ops.layernorm_bwd_add(x.to(DEV), g.to(DEV), mean, rstd, dy.to(DEV), add.to(DEV), dg, db)
    torch.testing.assert_close(dx.cpu(), xr.grad + add, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)
    buf = add.to(DEV)                                                      # in place (dx aliases dx_add), gradients accumulate
    ops.layernorm_bwd_add(x.to(DEV), g.to(DEV), mean, rstd, dy.to(DEV), buf, dg, db, out=buf)
    torch.testing.assert_close(buf.cpu(), xr.grad + add, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), 2 * gr.grad, rtol=1e-4, atol=2e-4)
    dx0 = ops.layernorm_bwd_add(x.to(DEV), g.to(DEV), mean, rstd, dy.to(DEV), None, dg, db)
    torch.testing.assert_close(dx0.cpu(), xr.grad, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- loss tail (csrc/loss.hip)
def test_rownorm_fwd_bwd_vs_torch():
    """hh_rownorm_fwd / bwd = x / clamp(||x||, eps) of sim_matrix (metric.py:370-373) and its gradient, incl. a row under the clamp,
    a row-strided view (bit-identical to the dense copy) and 3-D operands."""
    from helping_hand_for_egocentric_videos_amd.model.metric import _RowNorm
    g = torch.Generator().manual_seed(0)
    for rows, cols in ((5, 256), (160, 256), (33, 118), (582, 256), (7, 3)):
        x = torch.randn(rows, cols, generator=g)
        x[1] = 1e-12                                                     # ||x|| < eps: the clamp is the divisor
        xr = x.clone().requires_grad_(True)
        yr = xr / xr.norm(dim=-1, keepdim=True).clamp_min(1e-8)
        w = torch.randn(rows, cols, generator=g)
        (yr * w).sum().backward()
        xg = x.cuda().requires_grad_(True)
        yg = _RowNorm.apply(xg, 1e-8)
        (yg * w.cuda()).sum().backward()
        torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-5, atol=2e-6)
    big = torch.randn(40, 300, generator=g).cuda()
    view = big[::5, :256]                                                # row stride 1500, as the packed all-gather / text[::5] give
    assert torch.equal(_RowNorm.apply(view, 1e-8), _RowNorm.apply(view.contiguous(), 1e-8))
    x3 = torch.randn(3, 4, 64, generator=g).cuda()
    torch.testing.assert_close(_RowNorm.apply(x3, 1e-8), x3 / x3.norm(dim=-1, keepdim=True).clamp_min(1e-8), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("Bg,R", [(2, 5), (6, 5), (32, 5), (100, 5), (256, 5), (7, 3)])
def test_egonce_rows_kernel_vs_oracle(Bg, R):
    """hh_egonce_fwd (EgoNCE.forward_rows: the step's call, run/train.py:144-149) vs oracle.losses.egonce: loss and both gradients, with
    absent captions (dropped rows), verb / noun positives off the diagonal, and global batches that are not multiples of 64."""
    from helping_hand_for_egocentric_videos_amd.model.loss import EgoNCE
    from helping_hand_for_egocentric_videos_amd.model.metric import sim_matrix
    from oracle import losses as OL
    g = torch.Generator().manual_seed(Bg * 10 + R)
    te, ve = torch.randn(R * Bg, 64, generator=g), torch.randn(Bg, 64, generator=g)
    vv, nv = (torch.rand(Bg, 11, generator=g) < 0.3).float(), (torch.rand(Bg, 13, generator=g) < 0.3).float()
    if Bg > 2:
        vv[0], nv[0] = vv[1], nv[1]
        vv[0:2, 3] = 1; nv[0:2, 5] = 1                                   # clips 0 and 1 share verb and noun: mutual positives
    pf = (torch.rand(R * Bg, generator=g) > 0.3).float()
    pf[::R] = 1.0                                                        # the first caption of a clip is always there
    sv, sn = OL.sim_matrix(vv, vv), OL.sim_matrix(nv, nv)
    ter, ver = te.clone().requires_grad_(True), ve.clone().requires_grad_(True)
    if R == 5:
        rl, _ = OL.egonce(OL.sim_matrix(ter, ver), sv, sn, pf[:, None].repeat(1, Bg))
    else:                                                                # the oracle hard-codes the reference's 5 rephrases: torch module on CPU
        rl = EgoNCE().forward(OL.sim_matrix(ter, ver), sv, sn, multi_pad_mask=pf[:, None].repeat(1, Bg), strict_mask=True, return_mask=False)[0]
    rl.backward()
    tg, vg = te.cuda().requires_grad_(True), ve.cuda().requires_grad_(True)
    gl = EgoNCE().forward_rows(sim_matrix(tg, vg), sv.cuda(), sn.cuda(), pf.cuda())
    (3.0 * gl).backward()                                                # a non-unit incoming gradient
    torch.testing.assert_close(gl.detach().cpu(), rl.detach(), rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(tg.grad.cpu() / 3.0, ter.grad, rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(vg.grad.cpu() / 3.0, ver.grad, rtol=1e-4, atol=2e-6)
    # the general module path (arbitrary multi_pad_mask, stock ops) gives the same value
    gen, _ = EgoNCE()(sim_matrix(tg.detach(), vg.detach()), sv.cuda(), sn.cuda(), multi_pad_mask=pf.cuda()[:, None].repeat(1, Bg), strict_mask=True,
                      return_mask=False)
    torch.testing.assert_close(gen, gl.detach(), rtol=2e-6, atol=1e-6)


def test_masked_ce_kernel_vs_torch():
    """hh_masked_ce_fwd vs F.cross_entropy(masked_fill(sim, noun_sim[gt] > thr with a zero diagonal, -1) / T) (loss.py:95-104)."""
    g = torch.Generator().manual_seed(4)
    rows, V, T, thr = 37, 582, 0.07, 0.6
    sim = (torch.rand(rows, V, generator=g) * 2 - 1)
    ne = torch.randn(V, 16, generator=g)
    ne[5] = ne[9] * 1.01                                                 # nouns 5 and 9 are near-duplicates: masked for each other
    nsim = torch.nn.functional.normalize(ne, dim=-1) @ torch.nn.functional.normalize(ne, dim=-1).t()
    gt = torch.randint(0, V, (rows,), generator=g)
    gt[0], gt[1] = 5, 9
    valid = torch.rand(rows, generator=g) > 0.25
    valid[:2] = True
    ns0 = nsim.clone(); ns0.fill_diagonal_(0)
    sr = sim.clone().requires_grad_(True)
    ce_ref = torch.nn.functional.cross_entropy(sr.masked_fill(ns0[gt] > thr, -1) / T, gt, reduction="none")
    w = torch.randn(rows, generator=g)
    (torch.where(valid, ce_ref, torch.zeros(())) * w).sum().backward()
    from helping_hand_for_egocentric_videos_amd.model.loss import _MaskedCE
    sg = sim.cuda().requires_grad_(True)
    ce = _MaskedCE.apply(sg, nsim.cuda(), gt.cuda(), valid.cuda(), T, thr)
    (ce * w.cuda()).sum().backward()
    torch.testing.assert_close(ce.detach().cpu(), torch.where(valid, ce_ref.detach(), torch.zeros(())), rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(sg.grad.cpu(), sr.grad, rtol=1e-5, atol=1e-6)
    assert float(sg.grad[0, 9]) == 0.0 and float(sg.grad[~valid.cuda()].abs().max()) == 0.0


# ---- LayerNorm folded into the GEMMs around it (include/hh.h: hh_gemm_epilogue.ln_stats / z_out; model/LaviLa.py:372-388)
LN_FOLD_SHAPES = [(4097, 1024, 1024), (300, 128, 128), (33, 256, 512),               # generic kernels / row-tail kernel
                  (256 * 40 + 32, 1024, 1024), (256 * 36, 1024, 512),                # persistent kernel + in-kernel row tail
                  (2 * 4097, 1024, 1024)]


@pytest.mark.parametrize("M,N,K", LN_FOLD_SHAPES + [(256 * 33 + 7, 1024, 4096)])
def test_gemm_ln_fold_producer_updates_the_residual_stream_in_place(M, N, K):
    """z_update: the fp32 sum x + A W^T + b replaces x (the residual-stream update of LaviLa.py:384,388 in the GEMM's epilogue), z is its
    bf16 rounding, C is not written."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K + 7)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g) * 2.0 + 0.3
    want = x + (a.float() @ w.float().t() + bias)
    c, z, st = ops.gemm(a, w, bias, z=(x, 1e-6, False, True))
    assert c is None
    scale = want.abs().max().item()
    assert (x - want).abs().max().item() <= 2e-3 * scale                        # fp32 accumulation order only
    assert torch.equal(z, x.to(torch.bfloat16))                                 # z is the rounding of what was written back
    rstd = (want.var(1, unbiased=False) + 1e-6).rsqrt()
    assert ((st[:, 0] - rstd) / rstd).abs().max().item() <= 2e-3


@pytest.mark.parametrize("M,N,K", [(224 * 64 + 40, 1024, 1024), (224 * 75 + 20, 1024, 640), (224 * 64 + 16, 1024, 4096), (32 * 3137, 1024, 1024)])
def test_gemm_224_row_tiles_match_256_row_tiles(M, N, K):
    """hh_set_tuning("gemm_tile224", 1) (opt-in): where 224-row tiles remove a partial round of the persistent kernel (csrc/gemm256.hip:
    hh_gemm256_tile_rows -- M = 100384, N = 1024: 1792 tiles = 7 whole rounds instead of 1568 = 6.125) the bias-only bf16 GEMM and the
    producer side of the LayerNorm fold run on them.  Same arithmetic per element: everything equal bit for bit to the 256-row tiling."""
    import ctypes
    from helping_hand_for_egocentric_videos_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g) * 2.0 + 0.3
    def run():
        out = [ops.gemm(a, w, bias)]
        name = _lib.lib().hh_prof_kernel_name(0).decode()
        out += list(ops.gemm(a, w, bias, z=(x, 1e-6, True)))
        out += list(ops.gemm(a, w, bias, z=(x, 1e-6, False)))[1:]
        xu = x.clone()
        out += list(ops.gemm(a, w, bias, z=(xu, 1e-6, False, True)))[1:] + [xu]
        return out, name, _lib.lib().hh_prof_kernel_name(0).decode()
    try:
        _lib.check(_lib.lib().hh_prof_enable(1), "prof")                         # (kernel names are noted while the timers are on)
        ops.set_tuning("gemm_tile224", 1)
        got, n0, n4 = run()
        assert n0.endswith("<true, 0, 224>") and n4.endswith("<true, 4, 224>"), (n0, n4)
        ops.set_tuning("gemm_tile224", 0)
        want, m0, m4 = run()
        assert m0.endswith("<true, 0, 256>") and m4.endswith("<true, 4, 256>"), (m0, m4)
    finally:
        ops.set_tuning("gemm_tile224", 0)
        _lib.lib().hh_prof_enable(0)
    for i, (u, v) in enumerate(zip(got, want)):
        if u.dtype == torch.float32 and u.shape[1] == 2:
            # row statistics: the rows behind the last full tile (a different set in the two tilings) take theirs from z, the others from
            # the fp32 partial sums of the epilogue
            assert ((u - v).abs() <= 2e-3 * (1.0 + v.abs())).all(), i
        else:
            assert torch.equal(u, v), i
    ref = a.float() @ w.float().t() + bias
    assert (got[0].float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", LN_FOLD_SHAPES)
@pytest.mark.parametrize("keep_c", [True, False])
def test_gemm_ln_fold_producer(M, N, K, keep_c):
    """Producer side: z = bf16(x + A W^T + b) beside C = bf16(A W^T + b), and (rstd, -rstd * mean) of the rows of z."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g) * 2.0 + 0.3
    x0 = x.clone()
    eps = 1e-6
    c, z, st = ops.gemm(a, w, bias, z=(x, eps, keep_c))
    assert torch.equal(x, x0)                                                    # the residual stream is only read
    v = a.float() @ w.float().t() + bias
    zr = x + v
    scale = v.abs().max().item()
    if keep_c:
        assert (c.float() - v).abs().max().item() <= 8e-3 * scale
        assert torch.equal(c, ops.gemm(a, w, bias))                              # the branch output is the plain GEMM's, bit for bit
    else:
        assert c is None
    assert (z.float() - zr).abs().max().item() <= 8e-3 * zr.abs().max().item()
    mean, var = zr.mean(1), zr.var(1, unbiased=False)
    rstd = (var + eps).rsqrt()
    # the statistics are those of z up to its bf16 rounding (rows of 128 ... 1024 values: ~2^-9 / sqrt(N) on the moments)
    assert ((st[:, 0] - rstd) / rstd).abs().max().item() <= 2e-3
    assert (st[:, 1] + rstd * mean).abs().max().item() <= 4e-3 * (1.0 + (rstd * mean).abs().max().item())


@pytest.mark.parametrize("M,N,K", [(256 * 40 + 32, 1024, 1024), (256 * 33 + 7, 1024, 4096), (300, 128, 128), (4097, 1024, 1024)])
def test_gemm_ln_fold_producer_on_the_bf16_pair_stream(M, N, K):
    """hh_gemm_epilogue.z_resid_lo (round 5): the residual stream as a pair of bf16 rows x = hi + lo, both updated in place; hi' = bf16(x') is
    the next LayerNorm's input, lo' = bf16(x' - hi').  Against fp64: hi' is the bf16 rounding of the fp32 sum, hi' + lo' holds x' to ~2^-16,
    the row statistics are those of x'; persistent kernel (+ in-kernel row tail), generic kernels, both walk directions bit-identical."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K + 11)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g) * 2.0 + 0.3
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    x0 = hi.double() + lo.double()                                      # what the pair holds on entry
    want = x0 + a.double() @ w.double().t() + bias.double()
    h1, l1 = hi.clone(), lo.clone()
    _, z, st = ops.gemm(a, w, bias, zpair=(h1, l1, 1e-6))
    assert z.data_ptr() == h1.data_ptr()
    scale = want.abs().max().item()
    got = h1.double() + l1.double()
    assert (got - want).abs().max().item() <= 3e-5 * scale + 2e-3 * 0      # ~2^-16 of the value + the GEMM's fp32 accumulation order
    assert (h1.double() - want).abs().max().item() <= 4.1e-3 * scale      # hi alone: one bf16 rounding
    assert (l1.double().abs() <= 4.0e-3 * (h1.double().abs() + 1e-30) + 1e-30).all()      # lo is a remainder of hi
    rstd = (want.var(1, unbiased=False) + 1e-6).rsqrt()
    assert ((st[:, 0].double() - rstd) / rstd).abs().max().item() <= 2e-3
    h2, l2 = hi.clone(), lo.clone()
    ops.gemm(a, w, bias, zpair=(h2, l2, 1e-6), reverse=True)
    assert torch.equal(h1, h2) and torch.equal(l1, l2)


@pytest.mark.parametrize("offset_sigma", [10.0, 50.0])
def test_gemm_ln_fold_producer_statistics_with_a_large_row_mean(offset_sigma):
    """rows inside persistent-GEMM tiles get their statistics from one-pass fp32 partial sums of the UNROUNDED z, the row tail
    from a two-pass over the bf16 z -- measure both against fp64 statistics of the bf16 z the consumer multiplies, on rows whose mean is
    10 / 50 sigma.  What this pins: the one-pass variance loses ~(mean/sigma)^2 2^-24 relative (1.5e-4 at 50 sigma) -- far below the
    2^-9 mean/sigma that rounding z itself costs (the fold's real caveat); tile rows and tail rows agree within that."""
    M, N, K = 256 * 40 + 40, 1024, 1024                                 # 40 x 4 tiles on the persistent kernel + a 40-row tail (in-kernel)
    g = torch.Generator(device=DEV).manual_seed(int(offset_sigma))
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device=DEV)
    x = torch.randn(M, N, device=DEV, generator=g) + offset_sigma        # sigma ~ 1.2 (x + branch), mean = offset
    eps = 1e-6
    _, z, st = ops.gemm(a, w, bias, z=(x, eps, False))
    zd = z.double()
    mean, var = zd.mean(1), zd.var(1, unbiased=False)
    rstd = (var + eps).rsqrt()
    e_rstd = ((st[:, 0].double() - rstd) / rstd).abs()
    e_mean = ((-st[:, 1].double() / st[:, 0].double()) - mean).abs() / var.sqrt()        # error of the mean in units of sigma
    tile, tail = slice(0, 256 * 40), slice(256 * 40, M)
    record("ln_fold_stats_offset_%d_sigma" % int(offset_sigma), "tile rows: rstd rel error vs fp64 statistics of bf16 z", float(e_rstd[tile].max()), 4e-2)
    record("ln_fold_stats_offset_%d_sigma" % int(offset_sigma), "tail rows: rstd rel error vs fp64 statistics of bf16 z", float(e_rstd[tail].max()), 1e-5)
    record("ln_fold_stats_offset_%d_sigma" % int(offset_sigma), "tile rows: mean error / sigma", float(e_mean[tile].max()), 2e-2)
    # tile rows: statistics of the unrounded z differ from those of bf16(z) by the rounding of z (2^-9 offset per element, averaged over the row)
    assert float(e_rstd[tile].max()) <= 4e-2 and float(e_mean[tile].max()) <= 2e-2
    assert float(e_rstd[tail].max()) <= 1e-5 and float(e_mean[tail].max()) <= 1e-5
    # and against the statistics of the unrounded fp32 z (what the one-pass sums see): the cancellation itself, (mean / sigma)^2 2^-24
    zr = (x.double() + a.double() @ w.double().t())
    r2 = (zr.var(1, unbiased=False) + eps).rsqrt()
    e_cancel = ((st[tile, 0].double() - r2[tile]) / r2[tile]).abs().max().item()
    record("ln_fold_stats_offset_%d_sigma" % int(offset_sigma), "tile rows: rstd rel error vs fp64 statistics of the unrounded z (one-pass cancellation)", e_cancel, 1e-3)
    assert e_cancel <= 1e-3


@pytest.mark.parametrize("M,N,K", LN_FOLD_SHAPES + [(256 * 33 + 7, 1024, 4096)])
def test_gemm_ln_fold_producer_with_bf16_residual_rows(M, N, K):
    """z_resid_dtype = HH_BF16 (round 5): the producer adds its result to bf16 rows -- the time projection reads z3 = bf16(x), the rows its
    block's norm3 consumed, instead of the fp32 stream (LaviLa.py:372: z1 = x + t only feeds norm1).  On the persistent kernel
    (gemm256w4p_kernel<true, 7, 256>), its in-kernel row tail and the generic kernels: z == bf16(float(xb) + A W^T + b) computed by the
    fp32-residual instantiation on the widened rows, bit for bit; the bf16 rows are only read; z_update with them is refused."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K + 3)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    xb = (torch.randn(M, N, device=DEV, generator=g) * 2.0 + 0.3).to(torch.bfloat16)
    x0 = xb.clone()
    c, z, st = ops.gemm(a, w, bias, z=(xb, 1e-6, False))
    assert c is None and torch.equal(xb, x0)
    _, z_ref, st_ref = ops.gemm(a, w, bias, z=(xb.float(), 1e-6, False))
    assert torch.equal(z, z_ref)
    torch.testing.assert_close(st, st_ref, rtol=1e-6, atol=1e-7)
    c2, z2, _ = ops.gemm(a, w, bias, z=(xb, 1e-6, True))
    assert torch.equal(z2, z) and torch.equal(c2, ops.gemm(a, w, bias))
    with pytest.raises(TypeError):
        ops.gemm(a, w, bias, z=(xb, 1e-6, False, True))


@pytest.mark.parametrize("M,N,K", [(4097, 3072, 1024), (300, 384, 128), (33, 256, 512), (256 * 40 + 32, 3072, 1024), (256 * 20, 4096, 1024),
                                   (2 * 4097, 3072, 1024)])
@pytest.mark.parametrize("variant", ["qkv", "qkv_planes", "gelu"])
def test_gemm_ln_fold_consumer(M, N, K, variant):
    """Consumer side: Linear(LayerNorm(z)) with the LayerNorm applied algebraically inside the GEMM (gamma folded into the bf16 weight,
    beta into the bias, rstd / mean through the row statistics) vs fp32 torch; and vs the stand-alone LayerNorm -> GEMM route."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K + 1)
    zf = torch.randn(M, K, device=DEV, generator=g) * torch.exp(0.5 * torch.randn(K, device=DEV, generator=g)) + 0.2
    z = zf.to(torch.bfloat16)
    gamma = 1.0 + 0.2 * torch.randn(K, device=DEV, generator=g)
    beta = 0.1 * torch.randn(K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = 0.02 * torch.randn(N, device=DEV, generator=g)
    eps = 1e-6
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(z.float(), (K,), gamma, beta, eps), w, b)
    wf, cs, bf_ = ops.fold_layernorm_into_linear(w, b, gamma, beta)
    st = ops.ln_rowstats(z, eps)
    cols = N // 3 // 128 * 128
    if variant == "gelu":
        out = ops.gemm(z, wf, bf_, act=ops.ACT_QUICKGELU, ln=(st, cs)).float()
        ref = ref * torch.sigmoid(1.702 * ref)
        old = ops.gemm(ops.layernorm(z, gamma, beta, eps), ops.to_bf16(w), b, act=ops.ACT_QUICKGELU).float()
    else:
        ref[:, :cols] *= 0.125
        out = ops.gemm(z, wf, bf_, colscale=0.125, colscale_cols=cols, ln=(st, cs), col_blocked=variant == "qkv_planes")
        if variant == "qkv_planes":
            out = out.transpose(0, 1).reshape(M, N)
        out = out.float()
        old = ops.gemm(ops.layernorm(z, gamma, beta, eps), ops.to_bf16(w), b, colscale=0.125, colscale_cols=cols).float()
    scale = ref.abs().max().item()
    e_new, e_old = (out - ref).abs().max().item() / scale, (old - ref).abs().max().item() / scale
    l_new, l_old = float((out - ref).norm() / ref.norm()), float((old - ref).norm() / ref.norm())
    record("gemm_ln_fold_consumer[%s,%d,%d,%d]" % (variant, M, N, K), "rel-L2 vs fp32 (stand-alone LN -> GEMM route: %.2e)" % l_old, l_new, 8e-3)
    assert e_new <= 1.6e-2 and l_new <= 8e-3, (e_new, l_new)
    assert l_new <= 1.5 * l_old + 1e-4, (l_new, l_old)                           # the fold is as accurate as the route it replaces


def test_text_flags_vs_torch():
    """hh_text_flags: EOT position (argmax of the token ids, first index on ties) and the "caption present" flag of run/train.py:144."""
    g = torch.Generator().manual_seed(2)
    rows, L = 203, 77
    text = torch.zeros(rows, L, dtype=torch.int64)
    for r in range(rows):
        n = int(torch.randint(0, 20, (1,), generator=g))
        text[r, 0] = 49406
        text[r, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g)
        text[r, 1 + n] = 49407
    text[7] = 0                                                                   # an all-zero row: argmax 0, flag 1 (0 non-zero tokens != 2)
    text[9, :] = 5                                                                # ties: first index
    eot, pad = ops.text_flags(text.to(DEV))
    assert torch.equal(eot.cpu(), text.argmax(-1)) and eot.dtype == torch.int64
    assert torch.equal(pad.cpu(), ((text != 0).sum(-1) != 2).float())


def test_walk_direction_does_not_change_results():
    """Round 5: hh_gemm_epilogue.walk_reverse / HH_QKV_WALK_REVERSE make a kernel walk its m-tiles / (clip, ...) problems last to first, so that
    it starts on the rows its predecessor wrote last (model/LaviLa.py: SpaceTimeBlock.fused alternates the direction).  Same tiles, same
    arithmetic: every output -- C, head-major planes, z, the in-place residual update, row statistics, attention rows incl. the folded CLS
    rows -- is bit-identical, on the persistent kernel with an in-kernel row tail."""
    M, N, K = 256 * 40 + 32, 1024, 1024
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g)
    stats = torch.stack([torch.rand(M, device=DEV, generator=g) + 0.5, torch.randn(M, device=DEV, generator=g) * 0.1], 1).contiguous()
    cs = torch.randn(N, device=DEV, generator=g)
    for kw in (dict(), dict(act=ops.ACT_QUICKGELU, ln=(stats, cs)), dict(colscale=0.125, colscale_cols=256, col_blocked=True, ln=(stats, cs))):
        assert torch.equal(ops.gemm(a, w, bias, **kw), ops.gemm(a, w, bias, reverse=True, **kw)), kw
    for zargs in ((x, 1e-6, False), (x.to(torch.bfloat16), 1e-6, False)):
        f, r = ops.gemm(a, w, bias, z=zargs), ops.gemm(a, w, bias, z=zargs, reverse=True)
        assert torch.equal(f[1], r[1]) and torch.equal(f[2], r[2])
    x1, x2 = x.clone(), x.clone()
    f, r = ops.gemm(a, w, bias, z=(x1, 1e-6, False, True)), ops.gemm(a, w, bias, z=(x2, 1e-6, False, True), reverse=True)
    assert torch.equal(x1, x2) and torch.equal(f[1], r[1]) and torch.equal(f[2], r[2])
    for B, T, n, heads in ((3, 16, 256, 4), (2, 32, 576, 2), (2, 4, 256, 3)):              # config 2's kernels, config 4's, a small T
        Nn, D = 1 + T * n, heads * 64
        qkv = bf(rnd(B * Nn, 3 * D, seed=9) * 0.5).to(DEV)
        planes = qkv.view(B * Nn, 3 * heads, 64).transpose(0, 1).contiguous()
        for mode in ("space", "time"):
            for t in (qkv, planes):
                assert torch.equal(ops.divided_attention(t, B, T, n, heads, mode), ops.divided_attention(t, B, T, n, heads, mode, reverse=True)), (mode, T, n)


def test_round4_entry_points_accept_empty_inputs():
    """M = 0 / no captions: the LayerNorm-fold GEMMs, hh_ln_rowstats and hh_text_flags return empty results instead of failing on the
    null data pointer of an empty tensor."""
    z = torch.empty(0, 1024, dtype=torch.bfloat16, device=DEV)
    assert ops.ln_rowstats(z, 1e-6).shape == (0, 2)
    a = torch.empty(0, 1024, dtype=torch.bfloat16, device=DEV)
    w = torch.randn(1024, 1024, device=DEV).to(torch.bfloat16)
    b = torch.randn(1024, device=DEV)
    x = torch.empty(0, 1024, device=DEV)
    c, zz, st = ops.gemm(a, w, b, z=(x, 1e-6, True))
    assert c.shape == (0, 1024) and zz.shape == (0, 1024) and st.shape == (0, 2)
    c, zz, st = ops.gemm(a, w, b, z=(x, 1e-6, False, True))
    assert c is None and zz.shape == (0, 1024)
    assert ops.gemm(a, w, b, ln=(torch.empty(0, 2, device=DEV), torch.randn(1024, device=DEV))).shape == (0, 1024)
    eot, pad = ops.text_flags(torch.empty(0, 77, dtype=torch.int64, device=DEV))
    assert eot.shape == (0,) and pad.shape == (0,)
    g1, b1 = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
    assert ops.add_layernorm(torch.empty(0, 1024, device=DEV), z, g1, b1, 1e-5).shape == (0, 1024)
    y, yp, mean, rstd = ops.layernorm_pos(torch.empty(0, 512, device=DEV), g1[:512], b1[:512], 1e-5, torch.randn(16, 512, device=DEV), save_stats=True)
    assert y.shape == (0, 512) and yp.shape == (0, 512)
    dw, db = ops.gemm_tn(torch.empty(0, 512, dtype=torch.bfloat16, device=DEV), torch.empty(0, 256, dtype=torch.bfloat16, device=DEV), colsum=True)
    assert dw.shape == (512, 256) and float(dw.abs().max()) == 0.0 and db.shape == (512,)
    assert ops.divided_attention(torch.empty(0, 3072, dtype=torch.bfloat16, device=DEV), 0, 4, 32, 16, "space").shape == (0, 1024)
    torch.cuda.synchronize()


@pytest.mark.parametrize("splits,shape", [(1, (128, 128)), (3, (128, 256)), (7, (512,)), (64, (512, 1024))])
def test_sum_partials_vs_fp64(splits, shape):
    """hh_sum_partials (the split-K planes of the weight-gradient GEMM added up in plane order) against an fp64 sum; bit-exact against the
    same order in fp32 on the host."""
    g = torch.Generator(device="cpu").manual_seed(splits)
    part = (torch.randn((splits,) + shape, generator=g) * 3).cuda()
    got = ops.sum_partials(part)
    want = part.double().sum(0)
    assert (got.double() - want).abs().max().item() <= 2e-6 * splits * want.abs().max().item()
    seq = part[0].cpu().clone()
    for s_ in range(1, splits):
        seq += part[s_].cpu()
    assert torch.equal(got.cpu(), seq)               # plane order, left to right (four loads in flight, one chain of adds)
    out = torch.full(shape, 7.0, device="cuda")
    assert ops.sum_partials(part, out=out) is out and torch.equal(out, got)          # overwrites, does not accumulate
    with pytest.raises(ValueError):
        ops.sum_partials(part.half())


@pytest.mark.parametrize("mode", [ops.NT, ops.NN, ops.TN])
def test_qgemm_group_is_bit_identical_to_single_launches(mode):
    """hh_qgemm_f32x3_group: several independent products of one mode in one launch (the nine weight gradients of a decoder layer) -- every
    tile is computed exactly as by its own launch, so the results are bit-identical; ragged shapes, the bias-gradient column sums, the
    prologue scale / dropout and the head-batched form included.  More than HH_QGEMM_GROUP_MAX products: several launches."""
    g = torch.Generator(device="cpu").manual_seed(7 + mode)
    rnd = lambda *sh: torch.randn(*sh, generator=g).cuda()
    shapes = [(416, 512, 2048), (416, 2048, 512), (52, 512, 512), (416, 128, 64), (13, 1536, 512)] + [(416, 512, 512)] * 9     # (rows, a, b)
    grp, single, grouped = ops.QGemmGroup(), [], []
    for i, (R, a_, b_) in enumerate(shapes):
        if mode == ops.TN:                      # dY [R, a] ^T X [R, b] -> [a, b] + column sums
            A, B = rnd(R, a_), rnd(R, b_)
            kw = dict(a_scale=0.125) if i == 1 else (dict(a_drop_p=0.1, a_drop_seed=99 + i, a_drop_ld=a_) if i == 2 else {})
            cs1, cs2 = torch.empty(a_, device="cuda"), torch.empty(a_, device="cuda")
            single.append((ops.qgemm(A, B, mode, colsum=cs1, **kw), cs1))
            grouped.append((ops.qgemm(A, B, mode, colsum=cs2, defer=grp, **kw), cs2))
        elif mode == ops.NT:                    # X [R, a] W [b, a]^T + bias -> [R, b]
            A, B, bias = rnd(R, a_), rnd(b_, a_), rnd(b_)
            kw = dict(relu=True) if i == 0 else (dict(resid=rnd(R, b_), drop_p=0.1, drop_seed=5) if i == 1 else {})
            single.append((ops.qgemm(A, B, mode, bias=bias, **kw),))
            grouped.append((ops.qgemm(A, B, mode, bias=bias, defer=grp, **kw),))
        else:                                   # dY [R, a] W [a, b] -> [R, b]
            A, B = rnd(R, a_), rnd(a_, b_)
            single.append((ops.qgemm(A, B, mode),))
            grouped.append((ops.qgemm(A, B, mode, defer=grp),))
    if mode == ops.TN:                          # the head-batched weight gradient of the memory-space cross-attention + a split-K product
        x, y, rs = rnd(416, 512), rnd(416, 8 * 512), rnd(416, 8)
        o1, o2 = torch.empty(512, 512, device="cuda"), torch.empty(512, 512, device="cuda")
        c1, c2 = torch.empty(512, device="cuda"), torch.empty(512, device="cuda")
        ops.head_map_wgrad(x, y, o1, colsum=c1, rowscale=rs)
        ops.head_map_wgrad(x, y, o2, colsum=c2, rowscale=rs, defer=grp)
        single.append((o1, c1)); grouped.append((o2, c2))
        A, B = rnd(6656, 512), rnd(6656, 512)
        z1, z2 = torch.zeros(512, 512, device="cuda"), torch.zeros(512, 512, device="cuda")
        ops.qgemm(A, B, mode, out=z1, splitk=8)
        ops.qgemm(A, B, mode, out=z2, splitk=8, defer=grp)
        assert len(grp.items) > 12
    before = ops._lib.lib().hh_call_count()
    n_items = len(grp.items)
    grp.launch()
    assert ops._lib.lib().hh_call_count() - before == (n_items + 11) // 12 and not grp.items
    torch.cuda.synchronize()
    for one, many in zip(single, grouped):
        assert torch.equal(one[0], many[0])
        if len(one) > 1:                        # the column sums are folded by LDS float atomics (free order): equal to rounding, not to the bit
            assert (one[1] - many[1]).abs().max().item() <= 2e-6 * one[1].abs().max().item()
    if mode == ops.TN:                          # (atomic split-K: the order of the adds is free)
        assert (z1 - z2).abs().max().item() <= 1e-3 * z1.abs().max().item() and z2.abs().max().item() > 0
    mixed = ops.QGemmGroup()
    ops.qgemm(rnd(16, 64), rnd(64, 64), ops.NT, defer=mixed)
    with pytest.raises(ValueError):
        ops.qgemm(rnd(16, 64), rnd(64, 64), ops.NN, defer=mixed)
