"""Python-side operator wrappers over the libhh C ABI (include/hh.h).

Every function validates device / dtype / contiguity in Python (so the C side stays branch-light,
SURVEY.md section 8b "Error conventions"), allocates outputs as torch tensors, and launches on the
current torch stream.  Nothing here computes on the CPU: non-GPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from ._lib import GemmEpilogue, QGemmOpts

F32, BF16 = 0, 1
ACT_NONE, ACT_QUICKGELU, ACT_RELU = 0, 1, 2


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("libhh: unsupported dtype %s" % t.dtype)


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors (got device %s); there is no CPU fallback" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("libhh ops need contiguous tensors")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(fn_name, *dims, device):
    """fp32 scratch buffer sized by the library (include/hh.h: hh_workspace_bytes_*)."""
    nbytes = getattr(_lib.lib(), "hh_workspace_bytes_" + fn_name)(*dims)
    if nbytes < 0:
        raise ValueError("hh_workspace_bytes_%s%s: bad arguments" % (fn_name, dims))
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device)


GEMM256_DEFAULT = 5          # hh_set_tuning("gemm256", .): the library's default (csrc/gemm256.hip)


def set_tuning(name, value):
    """Performance knob for A/B measurements (include/hh.h: hh_set_tuning)."""
    _lib.check(_lib.lib().hh_set_tuning(name.encode(), int(value)), "hh_set_tuning")


def space_redo_count(reset=False):
    """Debug / tests: query blocks the space-attention kernels have redone on their running-maximum path since the last reset
    (include/hh.h: hh_debug_space_redo_count; synchronises the device)."""
    v = _lib.lib().hh_debug_space_redo_count(1 if reset else 0)
    if v < 0:
        raise RuntimeError("hh_debug_space_redo_count failed")
    return int(v)


PROF_ROLE_DECODER, PROF_ROLE_VISION, PROF_ROLE_TEXT = 0, 1, 2


class prof_role:
    """`with ops.prof_role(ops.PROF_ROLE_VISION): ...` -- names the part of the step whose kernels the host launches inside the block, for
    the library's per-kernel timers (include/hh.h: hh_prof_set_role).  Costs one relaxed store; never changes results."""

    current = PROF_ROLE_DECODER          # host-side shadow of the library's role (the only writer is this class), so that blocks nest

    def __init__(self, role):
        self.role = int(role)
        self.prev = PROF_ROLE_DECODER

    def __enter__(self):
        self.prev = prof_role.current
        prof_role.current = self.role
        _lib.lib().hh_prof_set_role(self.role)

    def __exit__(self, *exc):
        prof_role.current = self.prev
        _lib.lib().hh_prof_set_role(self.prev)
        return False


def set_stream_cu_budget(stream, n_cus):
    """Persistent one-workgroup-per-CU kernels launched on `stream` use `n_cus` workgroups (0 = all CUs); include/hh.h:
    hh_stream_set_cu_budget."""
    _lib.check(_lib.lib().hh_stream_set_cu_budget(ctypes.c_void_p(stream.cuda_stream), int(n_cus)), "hh_stream_set_cu_budget")


def stream_cu_budget(stream=None):
    s = torch.cuda.current_stream() if stream is None else stream
    n = ctypes.c_int()
    _lib.check(_lib.lib().hh_stream_get_cu_budget(ctypes.c_void_p(s.cuda_stream), ctypes.byref(n)), "hh_stream_get_cu_budget")
    return n.value


def layernorm(x, gamma, beta, eps, out_dtype=torch.bfloat16, save_stats=False):
    """LayerNorm over the last dim; x fp32/bf16 [..., cols] -> out_dtype."""
    _chk(x, gamma, beta)
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    _lib.check(L.hh_layernorm_fwd(_p(x), _dt(x), _p(gamma), _p(beta), _p(y), _dt(y), _p(mean), _p(rstd), rows, cols,
                                  float(eps), _stream()), "hh_layernorm_fwd")
    return (y, mean, rstd) if save_stats else y


def add_layernorm(x, delta, gamma, beta, eps, write_x=True, out_dtype=torch.bfloat16, delta2=None):
    """x (fp32, in place if write_x) = (x + delta) + delta2 (bf16 branches, delta2 optional); returns LN(x).  [rows, cols]."""
    _chk(x, delta, gamma, beta, delta2)
    if x.dtype != torch.float32 or delta.dtype != torch.bfloat16 or x.shape != delta.shape:
        raise TypeError("add_layernorm: x fp32 and delta bf16 of the same shape")
    if delta2 is not None and (delta2.dtype != torch.bfloat16 or delta2.shape != x.shape):
        raise TypeError("add_layernorm: delta2 must be bf16 of the shape of x")
    cols = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().hh_add_layernorm_fwd(_p(x), _p(delta), _p(delta2), int(bool(write_x)), _p(gamma), _p(beta), _p(y), _dt(y),
                                               x.numel() // cols, cols, float(eps), _stream()), "hh_add_layernorm_fwd")
    return y


def layernorm_bwd(x, gamma, mean, rstd, dy, dg=None, db=None):
    """-> (dx, dgamma, dbeta).  dg / db: fp32 [cols] buffers the weight / bias gradients are ACCUMULATED into (zero on entry for a plain
    gradient -- e.g. a parameter's slice of the zeroed gradient arena); fresh zeroed buffers by default."""
    _chk(x, gamma, mean, rstd, dy, dg, db)
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dg = torch.zeros(cols, dtype=torch.float32, device=x.device) if dg is None else dg
    db = torch.zeros(cols, dtype=torch.float32, device=x.device) if db is None else db
    L = _lib.lib()
    _lib.check(L.hh_layernorm_bwd(_p(x), _dt(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dx), _p(dg), _p(db), rows, cols,
                                  _stream()), "hh_layernorm_bwd")
    return dx, dg, db


def layernorm_pos(x, gamma, beta, eps, pos, out_dtype=torch.float32, save_stats=False):
    """(LN(x), LN(x) + pos[row % pos_rows]) in one pass; x [rows, cols] fp32/bf16, pos fp32 [pos_rows, cols]."""
    _chk(x, gamma, beta, pos)
    cols = x.shape[-1]
    rows = x.numel() // cols
    if pos.dtype != torch.float32 or pos.shape[-1] != cols:
        raise TypeError("layernorm_pos: pos must be fp32 [pos_rows, cols]")
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    y2 = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hh_layernorm_pos_fwd(_p(x), _dt(x), _p(gamma), _p(beta), _p(y), _p(y2), _dt(y), _p(pos), pos.numel() // cols,
                                               _p(mean), _p(rstd), rows, cols, float(eps), _stream()), "hh_layernorm_pos_fwd")
    return (y, y2, mean, rstd) if save_stats else (y, y2)


def layernorm_bwd_add(x, gamma, mean, rstd, dy, dx_add, dgamma, dbeta, out=None):
    """dx = dx_add + LayerNorm-backward(dy); dgamma / dbeta (fp32 [cols]) are ACCUMULATED into.  out may alias dx_add."""
    _chk(x, gamma, mean, rstd, dy, dx_add, dgamma, dbeta, out)                  # dx_add None: plain LayerNorm backward
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(_lib.lib().hh_layernorm_bwd_add(_p(x), _dt(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dx_add), _p(dx), _p(dgamma), _p(dbeta),
                                               rows, cols, _stream()), "hh_layernorm_bwd_add")
    return dx


NT, NN, TN = 0, 1, 2


class QGemmGroup:
    """Independent hh_qgemm_f32x3 products of ONE mode collected for a single launch (include/hh.h: hh_qgemm_f32x3_group):
    `qgemm(..., defer=group)` / `head_map_wgrad(..., defer=group)` record their product instead of launching it and return the output
    tensor, whose contents exist after `group.launch()`.  The group keeps every operand alive until then.  More than
    HH_QGEMM_GROUP_MAX products go out as several launches."""

    def __init__(self):
        self.items, self.refs = [], []

    def add(self, item, *tensors):
        if self.items and self.items[0].mode != item.mode:
            raise ValueError("QGemmGroup: all products of a group share one mode")
        self.items.append(item)
        self.refs.append(tensors)

    def launch(self):
        L = _lib.lib()
        for i0 in range(0, len(self.items), _lib.QGEMM_GROUP_MAX):
            chunk = self.items[i0:i0 + _lib.QGEMM_GROUP_MAX]
            arr = (_lib.QGemmItem * len(chunk))(*chunk)
            _lib.check(L.hh_qgemm_f32x3_group(arr, len(chunk), _stream()), "hh_qgemm_f32x3_group")
        self.items, self.refs = [], []


def _qgemm_item(a, lda, b, ldb, c, ldc, M, N, K, mode, o):
    it = _lib.QGemmItem()
    it.A, it.lda, it.B, it.ldb, it.C, it.ldc = a.data_ptr(), int(lda), b.data_ptr(), int(ldb), c.data_ptr(), int(ldc)
    it.M, it.N, it.K, it.mode = int(M), int(N), int(K), int(mode)
    it.opts = o
    return it


def qgemm(a, b, mode=NT, *, out=None, bias=None, scale=0.0, scale_ncols=0, relu=False, drop_p=0.0, drop_seed=0, relu_mask=None, mask_scale=1.0,
          resid=None, a_scale=0.0, a_drop_p=0.0, a_drop_seed=0, a_drop_ld=0, colsum=None, splitk=1, defer=None):
    """Query-side GEMM at fp32-grade accuracy on the bf16 matrix cores (include/hh.h: hh_qgemm_f32x3).  All operands fp32, 2-D with
    unit inner stride.  mode NT: a [M,K], b [N,K];  NN: a [M,K], b [K,N];  TN: a [K,M], b [K,N]  ->  out fp32 [M,N].
    splitk > 1: the contraction is split over workgroups that ADD into `out` / `colsum` atomically (both must be zero on entry; a fresh
    `out` is allocated zeroed here); no epilogue options."""
    for t in (a, b, out, resid, relu_mask):
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors (got device %s); there is no CPU fallback" % t.device)
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError("qgemm: operands must be fp32, 2-D, unit inner stride")
    _chk(bias, colsum)
    if mode == NT:
        (M, K), (N, Kb) = a.shape, b.shape
    elif mode == NN:
        (M, K), (Kb, N) = a.shape, b.shape
    else:
        (K, M), (Kb, N) = a.shape, b.shape
    if K != Kb:
        raise ValueError("qgemm: contraction mismatch %s vs %s (mode %d)" % (tuple(a.shape), tuple(b.shape), mode))
    if out is None:
        out = (torch.zeros if splitk > 1 else torch.empty)((M, N), dtype=torch.float32, device=a.device)
    o = QGemmOpts()
    o.splitk = int(splitk)
    o.a_scale, o.a_drop_p, o.a_drop_seed, o.a_drop_ld = float(a_scale), float(a_drop_p), int(a_drop_seed) & 0xFFFFFFFF, int(a_drop_ld)
    o.bias = bias.data_ptr() if bias is not None else None
    o.scale, o.scale_ncols, o.relu = float(scale), int(scale_ncols), int(bool(relu))
    o.drop_p, o.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    o.relu_mask = relu_mask.data_ptr() if relu_mask is not None else None
    o.ldmask = relu_mask.stride(0) if relu_mask is not None else 0
    o.mask_scale = float(mask_scale)
    o.resid = resid.data_ptr() if resid is not None else None
    o.ldr = resid.stride(0) if resid is not None else 0
    o.colsum = colsum.data_ptr() if colsum is not None else None
    if defer is not None:                              # part of a grouped launch (QGemmGroup.launch)
        defer.add(_qgemm_item(a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, mode, o), a, b, out, bias, relu_mask, resid, colsum)
        return out
    _lib.check(_lib.lib().hh_qgemm_f32x3(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, int(mode), ctypes.byref(o),
                                         _stream()), "hh_qgemm_f32x3")
    return out


def qself_attn_fwd(qkv, B, Q, heads, dropout_p=0.0, seed=0):
    """qkv fp32 [B*Q, 3*heads*64] (not pre-scaled) -> fp32 [B*Q, heads*64]."""
    _chk(qkv)
    C = heads * 64
    if qkv.dtype != torch.float32 or tuple(qkv.shape) != (B * Q, 3 * C):
        raise ValueError("qself_attn_fwd: qkv must be fp32 [B*Q, 3*heads*64]")
    out = torch.empty((B * Q, C), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().hh_qself_attn_fwd(_p(qkv), _p(out), B, Q, heads, float(dropout_p), int(seed) & 0xFFFFFFFF, _stream()), "hh_qself_attn_fwd")
    return out


def qself_attn_bwd(qkv, dout, B, Q, heads, dropout_p=0.0, seed=0):
    _chk(qkv, dout)
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.lib().hh_qself_attn_bwd(_p(qkv), _p(dout), _p(dqkv), B, Q, heads, float(dropout_p), int(seed) & 0xFFFFFFFF, _stream()),
               "hh_qself_attn_bwd")
    return dqkv


def ln_rowstats(z, eps):
    """(rstd, -rstd * mean) per row of z bf16 [rows, cols] -> fp32 [rows, 2]: the `ln=` operand of gemm() for a z that did not come
    out of gemm(..., z=...) (include/hh.h: hh_ln_rowstats)."""
    _chk(z)
    if z.dtype != torch.bfloat16 or z.dim() != 2:
        raise TypeError("ln_rowstats: z must be bf16 [rows, cols]")
    st = torch.empty((z.shape[0], 2), dtype=torch.float32, device=z.device)
    _lib.check(_lib.lib().hh_ln_rowstats(_p(z), z.stride(0), _p(st), z.shape[0], z.shape[1], float(eps), _stream()), "hh_ln_rowstats")
    return st


def fold_layernorm_into_linear(weight, bias, gamma, beta):
    """Operands of the consumer side of the LayerNorm fold for y = Linear(LayerNorm(z)) (include/hh.h): the bf16 weight gamma o W, its
    row sums over K (of the ROUNDED operand, so that rstd * (z W'^T - mean * colsum) cancels exactly what the matrix core summed), and
    the bias beta W^T + b in fp32.  Once per weight (cached by the modules)."""
    w = weight.detach().float()
    wp = to_bf16((w * gamma.detach().float()[None, :]).contiguous())
    colsum = wp.float().sum(dim=1).contiguous()
    b = (w.double() @ beta.detach().double()).float()
    if bias is not None:
        b = b + bias.detach().float()
    return wp, colsum, b.contiguous()


def gemm(a, w, bias=None, *, out=None, out_dtype=torch.bfloat16, act=ACT_NONE, resid=None, colscale=1.0,
         colscale_cols=0, remap=None, out_rows=None, splitk=1, col_blocked=False, ln=None, z=None, reverse=False, zpair=None):
    """C = epilogue(A @ W^T).  a bf16 [M,K], w bf16 [N,K] (nn.Linear weight layout), bias fp32 [N].

    LayerNorm fold (include/hh.h):  ln=(stats fp32 [M,2], colsum fp32 [N]) -- consumer side: `a` holds un-normalised rows, `w` / `bias`
    come from fold_layernorm_into_linear;  z=(x fp32 [M,N], eps, keep_c[, update]) -- producer side: returns (C or None, z = bf16(x + A W^T + bias),
    stats of z) instead of C; update=True also writes the fp32 sum back into x (the residual-stream update in the GEMM's epilogue).

    resid fp32 [rows,N] is added after the activation; `out` may alias `resid` (in-place residual update).
    remap=(group, skip, offset) scatters output row m to m + (m//group)*skip + offset (token-major scatter).
    col_blocked=True returns C as N/64 planes [N/64, M, 64] (plane j = columns 64j .. 64j+63; include/hh.h c_block_stride):
    the QKV projection written this way is the head-major buffer of divided_attention.
    reverse: the persistent kernel walks its m-tiles last to first (hh_gemm_epilogue.walk_reverse; same results).
    zpair=(hi, lo, eps): producer side on the bf16 PAIR residual stream x = hi + lo (both updated in place; returns (None, hi, stats of hi)).
    """
    _chk(bias)
    for t in (a, w, resid, out):                     # 2-D operands may be row-strided views (unit inner stride)
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors (got device %s); there is no CPU fallback" % t.device)
        if t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError("gemm: operands must be 2-D with unit inner stride")
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise TypeError("gemm: A and W must be bf16")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError("gemm: K mismatch %s vs %s" % (tuple(a.shape), tuple(w.shape)))
    if ln is not None or z is not None:
        if splitk > 1 or resid is not None or remap is not None or out is not None or out_rows is not None or (z is not None and (col_blocked or ln is not None)):
            raise ValueError("gemm: the LayerNorm fold takes no split-K / residual / remap / preallocated output (and z= no col_blocked / ln=)")
    if splitk > 1:
        if bias is not None or resid is not None or out is not None or act != ACT_NONE:
            raise ValueError("gemm: split-K takes no bias / residual / activation / preallocated output")
        part = _workspace("gemm_splitk", M, N, int(splitk), device=a.device).view(splitk, M, N)
        e = GemmEpilogue()
        e.colscale, e.c_dtype, e.splitk, e.split_stride = 1.0, F32, int(splitk), M * N
        _lib.check(_lib.lib().hh_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(part), N, M, N, K, ctypes.byref(e),
                                           _stream()), "hh_gemm_bf16")
        return sum_partials(part)
    if col_blocked:
        if out is not None or resid is not None or remap is not None or out_rows is not None:
            raise ValueError("gemm: col_blocked takes no preallocated output / residual / row remap")
        planes = torch.empty((N // 64, M, 64), dtype=out_dtype, device=a.device)
        e = GemmEpilogue()
        e.bias = bias.data_ptr() if bias is not None else None
        e.colscale, e.colscale_cols, e.act, e.c_dtype, e.c_block_stride = float(colscale), int(colscale_cols), int(act), _dt(planes), M * 64
        e.walk_reverse = int(bool(reverse))
        if ln is not None:
            _set_ln(e, ln, M, N, bias)
        _lib.check(_lib.lib().hh_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(planes), 64, M, N, K, ctypes.byref(e), _stream()),
                   "hh_gemm_bf16")
        return planes
    zt = st = None
    keep_c = True
    update = False
    if zpair is not None:
        # bf16 pair stream (hh_gemm_epilogue.z_resid_lo): x = hi + lo updated in place; hi is the next LayerNorm's input z
        xh, xl, eps = zpair
        _chk(xh, xl)
        if z is not None or ln is not None or xh.dtype != torch.bfloat16 or xl.dtype != torch.bfloat16 or tuple(xh.shape) != (M, N) or tuple(xl.shape) != (M, N) \
                or not xh.is_contiguous() or not xl.is_contiguous():
            raise TypeError("gemm: zpair=(hi, lo, eps) needs contiguous bf16 [M, N] halves and no z= / ln=")
        st = torch.empty((M, 2), dtype=torch.float32, device=a.device)
        part = _workspace("gemm_zstats", M, N, device=a.device)
        e = GemmEpilogue()
        e.walk_reverse = int(bool(reverse))
        e.z_resid, e.z_ldr, e.z_out, e.z_ldc, e.z_stats, e.z_partials = xh.data_ptr(), N, xh.data_ptr(), N, st.data_ptr(), part.data_ptr()
        e.z_eps, e.skip_c, e.z_update, e.z_resid_dtype, e.z_resid_lo = float(eps), 1, 1, BF16, xl.data_ptr()
        e.bias = bias.data_ptr() if bias is not None else None
        e.colscale, e.c_dtype = 1.0, BF16
        _lib.check(_lib.lib().hh_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(xh), N, M, N, K, ctypes.byref(e), _stream()), "hh_gemm_bf16")
        return None, xh, st
    if z is not None:
        x, eps, keep_c = z[:3]
        update = bool(z[3]) if len(z) > 3 else False
        _chk(x)
        if x.dtype not in (torch.float32, torch.bfloat16) or tuple(x.shape) != (M, N) or out_dtype != torch.bfloat16 or (update and x.dtype != torch.float32):
            raise TypeError("gemm: z=(x, eps, keep_c[, update]) needs x fp32 [M, N] (bf16 allowed without update: a branch that only feeds a LayerNorm) and a bf16 output")
        zt = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
        st = torch.empty((M, 2), dtype=torch.float32, device=a.device)
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, N), dtype=out_dtype, device=a.device) if keep_c else None
    e = GemmEpilogue()
    e.walk_reverse = int(bool(reverse))
    if ln is not None:
        _set_ln(e, ln, M, N, bias)
    if z is not None:
        part = _workspace("gemm_zstats", M, N, device=a.device)
        e.z_resid, e.z_ldr, e.z_out, e.z_ldc, e.z_stats, e.z_partials = x.data_ptr(), N, zt.data_ptr(), N, st.data_ptr(), part.data_ptr()
        e.z_eps, e.skip_c, e.z_update, e.z_resid_dtype = float(eps), int(not keep_c), int(update), _dt(x)
        e.bias = bias.data_ptr() if bias is not None else None
        e.colscale, e.c_dtype = 1.0, BF16
        cbuf = out if keep_c else zt                               # (a valid pointer even when C is skipped)
        _lib.check(_lib.lib().hh_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(cbuf), N, M, N, K, ctypes.byref(e), _stream()), "hh_gemm_bf16")
        return out, zt, st
    e.bias = bias.data_ptr() if bias is not None else None
    e.resid = resid.data_ptr() if resid is not None else None
    e.ldr = resid.stride(0) if resid is not None else 0
    e.colscale, e.colscale_cols, e.act, e.c_dtype = float(colscale), int(colscale_cols), int(act), _dt(out)
    e.remap_group, e.remap_skip, e.remap_offset = remap if remap is not None else (0, 0, 0)
    L = _lib.lib()
    _lib.check(L.hh_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K, ctypes.byref(e),
                              _stream()), "hh_gemm_bf16")
    return out


def _set_ln(e, ln, M, N, bias):
    stats, colsum = ln
    _chk(stats, colsum)
    if stats.dtype != torch.float32 or tuple(stats.shape) != (M, 2) or colsum.dtype != torch.float32 or colsum.numel() != N or bias is None:
        raise TypeError("gemm: ln=(stats fp32 [M,2], colsum fp32 [N]) and a bias are required for the LayerNorm fold")
    e.ln_stats, e.ln_colsum = stats.data_ptr(), colsum.data_ptr()


def to_bf16(x):
    _chk(x)
    if x.dtype == torch.bfloat16:
        return x
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hh_cast_f32_to_bf16(_p(x), _p(y), x.numel(), _stream()), "hh_cast_f32_to_bf16")
    return y


def to_f32(x):
    _chk(x)
    if x.dtype == torch.float32:
        return x
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hh_cast_bf16_to_f32(_p(x), _p(y), x.numel(), _stream()), "hh_cast_bf16_to_f32")
    return y


def transpose_bf16(x, pad_cols_to=1):
    """x [rows, cols] fp32/bf16 -> bf16 [cols, rows_padded] (rows padded with zeros to a multiple of pad_cols_to)."""
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError("transpose_bf16: need a 2-D GPU tensor with unit inner stride (no CPU fallback)")
    rows, cols = x.shape
    rp = (rows + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    y = (torch.zeros if rp != rows else torch.empty)((cols, rp), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().hh_transpose_to_bf16(_p(x), _dt(x), x.stride(0), _p(y), rp, rows, cols, _stream()),
               "hh_transpose_to_bf16")
    return y


def patch_im2col(video, patch, kpad):
    """video fp32 [B,T,3,H,W] -> bf16 [B*T*n, kpad] patch rows (k = c*P*P + i*P + j)."""
    _chk(video)
    B, T, C, H, W = video.shape
    if C != 3 or video.dtype != torch.float32:
        raise ValueError("patch_im2col: expected fp32 [B,T,3,H,W]")
    n = (H // patch) * (W // patch)
    out = torch.empty((B * T * n, kpad), dtype=torch.bfloat16, device=video.device)
    _lib.check(_lib.lib().hh_patch_im2col(_p(video), _p(out), B * T, H, W, patch, kpad, _stream()), "hh_patch_im2col")
    return out


# normalisation constants of the reference's training transform (run/train.py:442-445)
NORM_MEAN = (108.3272985 / 255, 116.7460125 / 255, 104.09373615000001 / 255)
NORM_STD = (68.5005327 / 255, 66.6321579 / 255, 70.32316305 / 255)


def patch_im2col_u8(video, patch, kpad, mean=NORM_MEAN, std=NORM_STD):
    """uint8 frames [B,T,3,H,W] or channels-last [B,T,H,W,3] -> normalised bf16 patch rows [B*T*n, kpad]."""
    _chk(video)
    if video.dtype != torch.uint8 or video.dim() != 5:
        raise ValueError("patch_im2col_u8: expected uint8 [B,T,3,H,W] or [B,T,H,W,3]")
    cl = video.shape[-1] == 3 and video.shape[2] != 3
    B, T = video.shape[:2]
    H, W = (video.shape[2], video.shape[3]) if cl else (video.shape[3], video.shape[4])
    n = (H // patch) * (W // patch)
    out = torch.empty((B * T * n, kpad), dtype=torch.bfloat16, device=video.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.check(_lib.lib().hh_patch_im2col_u8(_p(video), _p(out), B * T, H, W, patch, kpad, int(cl), m3, s3, _stream()),
               "hh_patch_im2col_u8")
    return out


def embed_ln_pre(tok, cls, pos, temporal, gamma, beta, B, T, n, eps=1e-5, z_eps=None, pair=False):
    """x fp32 [B, 1+T*n, D] = ln_pre(cat[cls, tok] + pos + temporal) (LaviLa.py:548-559).  z_eps (the next LayerNorm's eps): also returns
    z = bf16(x) [rows, D] and its row statistics fp32 [rows, 2] -- the operands of the first block's folded norm3 -- from the same pass.
    pair (with z_eps): the bf16 pair stream instead of x -- returns (z, z_lo = bf16(x - z), stats); the fp32 rows are not written."""
    _chk(tok, cls, pos, temporal, gamma, beta)
    D = tok.shape[-1]
    rows = B * (1 + T * n)
    if pair and z_eps is None:
        raise ValueError("embed_ln_pre: pair=True needs z_eps")
    x = None if pair else torch.empty((B, 1 + T * n, D), dtype=torch.float32, device=tok.device)
    z = st = zl = None
    if z_eps is not None:
        z = torch.empty((rows, D), dtype=torch.bfloat16, device=tok.device)
        st = torch.empty((rows, 2), dtype=torch.float32, device=tok.device)
        zl = torch.empty((rows, D), dtype=torch.bfloat16, device=tok.device) if pair else None
    _lib.check(_lib.lib().hh_embed_ln_pre(_p(tok), _p(cls), _p(pos), _p(temporal), _p(gamma), _p(beta), _p(x), B, T, n, D,
                                          float(eps), _p(z), _p(st), float(z_eps or 0.0), _p(zl), _stream()), "hh_embed_ln_pre")
    if pair:
        return z, zl, st
    return x if z_eps is None else (x, z, st)


def layernorm_split_cls(x, gamma, beta, eps, clips, out_dtype=torch.bfloat16, x_lo=None):
    """LayerNorm of `clips` clips of N = rows / clips token rows each, CLS first: -> (cls [clips, D], patches [clips, N - 1, D]) -- the
    tower's final norm written straight into the decoder's grid (include/hh.h: hh_layernorm_split_cls_fwd).  x_lo: the low halves of a bf16
    pair stream (x then holds the high halves): the rows are x + x_lo."""
    _chk(x, gamma, beta, x_lo)
    if x_lo is not None and (x.dtype != torch.bfloat16 or x_lo.dtype != torch.bfloat16 or x_lo.shape != x.shape):
        raise TypeError("layernorm_split_cls: x_lo needs bf16 x / x_lo of one shape")
    cols = x.shape[-1]
    rows = x.numel() // cols
    if clips <= 0 or rows % clips or rows // clips < 2:
        raise ValueError("layernorm_split_cls: %d rows do not split into %d clips of >= 2 tokens" % (rows, clips))
    N = rows // clips
    ycls = torch.empty((clips, cols), dtype=out_dtype, device=x.device)
    ypat = torch.empty((clips, N - 1, cols), dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().hh_layernorm_split_cls_fwd(_p(x), _dt(x), _p(gamma), _p(beta), _p(ypat), _p(ycls), _dt(ypat), clips, N, cols, float(eps), _p(x_lo), _stream()),
               "hh_layernorm_split_cls_fwd")
    return ycls, ypat


def gemm_tn(at, bt, splits=None, colsum=False, out=None):
    """Weight-gradient GEMM: at bf16 [K, M], bt bf16 [K, N] (token-major, row-strided views allowed) -> fp32 [M, N] = at^T @ bt.
    Split-K over the tokens; the partial tiles are summed here (into `out`, fp32 [M, N], when given).  colsum=True also returns sum_k at[k, :] (fp32 [M]: the bias
    gradient when `at` is dY), accumulated by the same kernel from the A fragments it already holds."""
    for t_ in (at, bt):
        if not t_.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors; there is no CPU fallback")
    K, M = at.shape
    N = bt.shape[1]
    if at.dtype != torch.bfloat16 or bt.dtype != torch.bfloat16 or bt.shape[0] != K or at.stride(1) != 1 or bt.stride(1) != 1:
        raise ValueError("gemm_tn: operands must be bf16 [K, M] / [K, N] with unit column stride")
    if K == 0:                                  # no tokens: the sum over nothing
        z = torch.zeros((M, N), dtype=torch.float32, device=at.device)
        return (z, torch.zeros((M,), dtype=torch.float32, device=at.device)) if colsum else z
    if splits is None:
        tiles = (M // 128) * (N // 128)
        splits = max(1, min(256, 1024 // max(tiles, 1), (K + 511) // 512))
    part = _workspace("gemm_tn", M, N, int(splits), device=at.device).view(splits, M, N)
    cs = torch.empty((splits, M), dtype=torch.float32, device=at.device) if colsum else None
    _lib.check(_lib.lib().hh_gemm_tn_bf16(_p(at), at.stride(0), _p(bt), bt.stride(0), _p(part), _p(cs), M, N, K, int(splits), _stream()),
               "hh_gemm_tn_bf16")
    if out is not None:                              # the partial tiles are summed (or the single one copied) into the caller's buffer
        if splits == 1:
            out.copy_(part[0])
        else:
            sum_partials(part, out)
    else:
        out = part[0] if splits == 1 else sum_partials(part)
    if colsum:
        return out, (cs[0] if splits == 1 else sum_partials(cs))
    return out


def sum_partials(part, out=None):
    """part fp32 [splits, ...] (dense) -> fp32 [...] = the planes added in order (include/hh.h: hh_sum_partials); `out`: dense fp32 of that shape."""
    if not part.is_cuda:
        raise RuntimeError("libhh ops need GPU tensors; there is no CPU fallback")
    if part.dtype != torch.float32 or not part.is_contiguous() or part.dim() < 2:
        raise ValueError("sum_partials: fp32 dense [splits, ...] planes")
    n = part[0].numel()
    if out is None:
        out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n or out.device != part.device:
        raise ValueError("sum_partials: out must be dense fp32 with one plane's element count")
    _lib.check(_lib.lib().hh_sum_partials(_p(part), _p(out), int(part.shape[0]), n, _stream()), "hh_sum_partials")
    return out


LOG2E = 1.4426950408889634


def attention_q_scale(mode, head_dim=64):
    """Scale the QKV GEMM epilogue applies to the q columns for divided_attention(mode): d^-1/2 (LaviLa.py:252) for "time"; the
    space kernel takes base-2 logits, d^-1/2 * log2(e) (include/hh.h)."""
    return head_dim ** -0.5 * (LOG2E if mode == "space" else 1.0)


QKV_TOKEN_MAJOR, QKV_HEAD_MAJOR = 0, 1                  # include/hh.h: enum hh_qkv_layout


def divided_attention(qkv, B, T, n, heads, mode, out=None, fold_cls=True, reverse=False):
    """qkv bf16 (q pre-scaled by attention_q_scale(mode)), token-major [B*N, 3D] or head-major planes [3*heads, B*N, 64]
    (gemm(..., col_blocked=True)) -> bf16 [B*N, D].  Rows 1.. come from the space/time kernel; the CLS row
    (query 0 attends all N keys) is folded into the same kernels as per-group partials + hh_cls_combine
    (fold_cls=False runs the stand-alone hh_cls_attn_fwd pass instead).  reverse: walk the problems last to first (HH_QKV_WALK_REVERSE; same
    results -- see SpaceTimeBlock.fused)."""
    _chk(qkv, out)
    N = 1 + T * n
    D = heads * 64
    if qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.shape == (3 * heads, B * N, 64) and qkv.is_contiguous():
        lay = QKV_HEAD_MAJOR
    elif qkv.dtype == torch.bfloat16 and qkv.shape == (B * N, 3 * D) and qkv.is_contiguous():
        lay = QKV_TOKEN_MAJOR
    else:
        raise ValueError("divided_attention: qkv must be contiguous bf16 [B*N, 3*heads*64] or [3*heads, B*N, 64], got %s" % (tuple(qkv.shape),))
    if mode not in ("space", "time"):
        raise ValueError(mode)
    if out is None:
        out = torch.empty((B * N, D), dtype=torch.bfloat16, device=qkv.device)
    if B == 0:
        return out
    L = _lib.lib()
    part, G = None, 0
    if fold_cls:
        tp = 1 << max(0, (T - 1).bit_length())          # the time kernel's tile holds the next power of two of frame slots
        G = T if mode == "space" else (n + (128 // tp) - 1) // (128 // tp)
        part = _workspace("attn_cls_partial", B, T, n, heads, int(mode == "time"), device=qkv.device).view(B, heads, G, 68)
    else:
        _lib.check(L.hh_cls_attn_fwd(_p(qkv), lay, _p(out), B, N, heads, int(mode == "space"), _stream()), "hh_cls_attn_fwd")
    if mode == "space":
        _lib.check(L.hh_space_attn_fwd(_p(qkv), lay | (2 if reverse else 0), _p(out), _p(part), B, T, n, heads, _stream()), "hh_space_attn_fwd")
    else:
        _lib.check(L.hh_time_attn_fwd(_p(qkv), lay | (2 if reverse else 0), _p(out), _p(part), B, T, n, heads, _stream()), "hh_time_attn_fwd")
    if fold_cls:
        _lib.check(L.hh_cls_combine(_p(part), G, _p(out), B, N, heads, _stream()), "hh_cls_combine")
    return out


def text_attention(qkv, S, L, heads, out=None):
    """Causal self-attention of the text tower: qkv bf16 [S*L, 3*heads*64] (q pre-scaled) -> bf16 [S*L, heads*64] (`out`: a dense bf16
    buffer of at least S*L rows whose first S*L rows are written -- the row-padded stream of Transformer.forward_frozen)."""
    _chk(qkv, out)
    W = heads * 64
    if qkv.dtype != torch.bfloat16 or qkv.shape != (S * L, 3 * W) or not qkv.is_contiguous():
        raise ValueError("text_attention: qkv must be contiguous bf16 [S*L, 3*heads*64], got %s" % (tuple(qkv.shape),))
    if out is None:
        out = torch.empty((S * L, W), dtype=torch.bfloat16, device=qkv.device)
    elif out.dtype != torch.bfloat16 or out.dim() != 2 or out.shape[1] != W or out.shape[0] < S * L or not out.is_contiguous():
        raise ValueError("text_attention: out must be contiguous bf16 [>= S*L, heads*64]")
    _lib.check(_lib.lib().hh_text_attn_fwd(_p(qkv), _p(out), S, L, heads, _stream()), "hh_text_attn_fwd")
    return out


# ---- decoder cross-attention without the memory-side K/V projections (include/hh.h: hh_mattn_*, csrc/mattn.hip) -------------------
MATTN_H, MATTN_C = 8, 512


def _head_qgemm(mode, a, lda, sa, b, ldb, sb, c, ldc, sc, M, N, K, *, bias=None, sbias=0, rowscale=None, ld_rs=0, s_rs=0, colsum=None, scolsum=0, defer=None):
    o = QGemmOpts()
    o.batch = MATTN_H
    o.stride_a, o.stride_b, o.stride_c = int(sa), int(sb), int(sc)
    o.bias = bias.data_ptr() if bias is not None else None
    o.stride_bias = int(sbias)
    o.rowscale = rowscale.data_ptr() if rowscale is not None else None
    o.ld_rowscale, o.stride_rowscale = int(ld_rs), int(s_rs)
    o.colsum = colsum.data_ptr() if colsum is not None else None
    o.stride_colsum = int(scolsum)
    if defer is not None:
        defer.add(_qgemm_item(a, lda, b, ldb, c, ldc, M, N, K, mode, o), a, b, c, bias, rowscale, colsum)
        return
    _lib.check(_lib.lib().hh_qgemm_f32x3(_p(a), int(lda), _p(b), int(ldb), _p(c), int(ldc), int(M), int(N), int(K), int(mode), ctypes.byref(o), _stream()),
               "hh_qgemm_f32x3(batched)")


def _f32_2d(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors (got device %s); there is no CPU fallback" % t.device)
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            raise RuntimeError("head-batched qgemm: operands must be fp32, 2-D, unit inner stride, 16-byte aligned rows")


def head_map_in(x, w, out=None):
    """out[r, h, :] = x[r, 64 h : 64 h + 64] @ w[64 h : 64 h + 64, :]  for the 8 heads in ONE launch (hh_qgemm_f32x3, batched NN):
    x fp32 [R, 512] (the head-major columns of a projected query / of a head-output gradient), w fp32 [512, 512] = the key / value rows
    of nn.MultiheadAttention.in_proj_weight (row-strided view) -> fp32 [R, 8 * 512]: the rows mapped INTO memory space."""
    _f32_2d(x, w, out)
    R, H, C = x.shape[0], MATTN_H, MATTN_C
    if x.shape[1] != H * 64 or tuple(w.shape) != (H * 64, C):
        raise ValueError("head_map_in: x [R, 512], w [512, 512]")
    if out is None:
        out = torch.empty((R, H * C), dtype=torch.float32, device=x.device)
    _head_qgemm(NN, x, x.stride(0), 64, w, w.stride(0), 64 * w.stride(0), out, out.stride(0), C, R, C, 64)
    return out


def head_map_out(y, w, bias=None, rowscale=None, out=None):
    """out[r, 64 h + n] = y[r, h, :] . w[64 h + n, :] (+ bias[64 h + n] * rowscale[r, h])  (batched NT): y fp32 [R, 8 * 512] rows in memory
    space (pooled rows / mapped-query gradients), w fp32 [512, 512] value / key rows of in_proj_weight -> fp32 [R, 512] head-major."""
    _f32_2d(y, w, out, rowscale)
    _chk(bias)
    R, H, C = y.shape[0], MATTN_H, MATTN_C
    if y.shape[1] != H * C or tuple(w.shape) != (H * 64, C):
        raise ValueError("head_map_out: y [R, 8 * 512], w [512, 512]")
    if out is None:
        out = torch.empty((R, H * 64), dtype=torch.float32, device=y.device)
    _head_qgemm(NT, y, y.stride(0), C, w, w.stride(0), 64 * w.stride(0), out, out.stride(0), 64, R, 64, C, bias=bias, sbias=64,
                rowscale=rowscale if bias is not None else None, ld_rs=rowscale.stride(0) if rowscale is not None else 0, s_rs=1)
    return out


def head_map_wgrad(x, y, out, colsum=None, rowscale=None, defer=None):
    """out[64 h + m, :] = sum_r x[r, 64 h + m] y[r, h, :]  (batched TN; the key / value rows of d in_proj_weight), optionally
    colsum[64 h + m] = sum_r x[r, 64 h + m] * rowscale[r, h] (the value-bias gradient).  x fp32 [R, 512], y fp32 [R, 8 * 512],
    out fp32 [512, 512] (row-strided view)."""
    _f32_2d(x, y, out, rowscale)
    _chk(colsum)
    R, H, C = x.shape[0], MATTN_H, MATTN_C
    if x.shape[1] != H * 64 or y.shape != (R, H * C) or tuple(out.shape) != (H * 64, C):
        raise ValueError("head_map_wgrad: x [R, 512], y [R, 8 * 512], out [512, 512]")
    _head_qgemm(TN, x, x.stride(0), 64, y, y.stride(0), C, out, out.stride(0), 64 * out.stride(0), 64, C, R, colsum=colsum, scolsum=64,
                rowscale=rowscale if colsum is not None else None, ld_rs=rowscale.stride(0) if rowscale is not None else 0, s_rs=1, defer=defer)
    return out


def mattn_slices(B, M, wanted=None):
    """Key slices of hh_mattn_fwd / hh_mattn_bwd: two workgroups (of 8 waves, one per CU) per (clip, slice) -- enough slices to give every
    CU one, at least four 32-key chunks per slice."""
    if wanted is None:
        wanted = min(max(1, (128 + B - 1) // max(B, 1)), max(1, (M // 32) // 4), 64)
    s = _lib.lib().hh_mattn_slices(int(M), int(wanted))
    if s < 1:
        raise ValueError("mattn: M must be a positive multiple of 32 (M=%d)" % M)
    return s


def _mattn_mem(mp, mem):
    for t in (mp, mem):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 3 or t.stride(2) != 1 or t.shape[2] != MATTN_C or t.stride(0) != t.shape[1] * t.stride(1):
            raise RuntimeError("mattn: mp / mem must be bf16 GPU tensors [B, M, 512] with unit column stride and dense batch stride")
    if mp.shape != mem.shape or mp.stride(1) != mem.stride(1):
        raise RuntimeError("mattn: mp and mem must share shape and row stride")


def mattn_fwd(qt, mp, mem, Q, dropout_p=0.0, seed=0, slices=None, keys_valid=0):
    """qt fp32 [B*Q, 8*512] (mapped queries), mp / mem bf16 [B, M, 512] -> (pooled fp32 [B*Q, 8*512], lse2 fp32 [B*Q, 8], rsum fp32 [B*Q, 8]).
    keys_valid: rows [keys_valid, M) of mp / mem are padding (finite) that the softmax does not see (0: none)."""
    _chk(qt)
    _mattn_mem(mp, mem)
    B, M, _ = mp.shape
    if qt.dtype != torch.float32 or tuple(qt.shape) != (B * Q, MATTN_H * MATTN_C):
        raise ValueError("mattn_fwd: qt must be fp32 [B*Q, 8*512]")
    S = mattn_slices(B, M, slices)
    pooled = torch.empty_like(qt)
    lse2 = torch.empty((B * Q, MATTN_H), dtype=torch.float32, device=qt.device)
    rsum = torch.empty_like(lse2)
    ws = _workspace("mattn_fwd", B, Q, S, device=qt.device) if S > 1 else None
    _lib.check(_lib.lib().hh_mattn_fwd(_p(qt), _p(mp), _p(mem), mp.stride(1), _p(pooled), _p(lse2), _p(rsum), _p(ws), S, B, Q, M, MATTN_H, MATTN_C,
                                       float(dropout_p), int(seed) & 0xFFFFFFFF, int(keys_valid), _stream()), "hh_mattn_fwd")
    return pooled, lse2, rsum


def mattn_bwd(qt, dpooled, lse2, dca, ca, bv, mp, mem, Q, pdT, dsT, qt16, dp16, row_off, dropout_p=0.0, seed=0, slices=None, keys_valid=0):
    """Backward of mattn_fwd for one layer: returns dqt fp32 [B*Q, 8*512]; writes this layer's 128 rows (row_off + head*16 + query) of
    pdT / dsT (bf16 [B, rows, M]) and of qt16 / dp16 (bf16 [B, rows, 512]) -- the operands of gemm_tn_batched2."""
    _chk(qt, dpooled, lse2, dca, ca, bv, pdT, dsT, qt16, dp16)
    _mattn_mem(mp, mem)
    B, M, _ = mp.shape
    rows = pdT.shape[1]
    if pdT.dtype != torch.bfloat16 or tuple(pdT.shape) != (B, rows, M) or dsT.shape != pdT.shape or tuple(qt16.shape) != (B, rows, MATTN_C) or dp16.shape != qt16.shape:
        raise ValueError("mattn_bwd: pdT / dsT bf16 [B, rows, M], qt16 / dp16 bf16 [B, rows, 512]")
    S = mattn_slices(B, M, slices)
    dqt = torch.empty_like(qt)
    ws = _workspace("mattn_bwd", B, Q, S, device=qt.device) if S > 1 else None
    _lib.check(_lib.lib().hh_mattn_bwd(_p(qt), _p(dpooled), _p(lse2), _p(dca), _p(ca), _p(bv), _p(mp), _p(mem), mp.stride(1), _p(dqt), _p(ws), S, _p(pdT), _p(dsT),
                                       _p(qt16), _p(dp16), rows, int(row_off), B, Q, M, MATTN_H, MATTN_C, float(dropout_p), int(seed) & 0xFFFFFFFF,
                                       int(keys_valid), _stream()), "hh_mattn_bwd")
    return dqt


def gemm_tn_batched2(at, bt, at2=None, bt2=None):
    """C[z] fp32 [M, N] = at[z]^T @ bt[z] (+ at2[z]^T @ bt2[z]):  at bf16 [batch, K, M], bt bf16 [batch, K, N], contiguous
    (include/hh.h: hh_gemm_tn_bf16_batched2) -> fp32 [batch, M, N]."""
    _chk(at, bt, at2, bt2)
    Bn, K, M = at.shape
    N = bt.shape[2]
    if at.dtype != torch.bfloat16 or bt.dtype != torch.bfloat16 or tuple(bt.shape) != (Bn, K, N) or (at2 is not None and (at2.shape != at.shape or bt2.shape != bt.shape)):
        raise ValueError("gemm_tn_batched2: at [batch, K, M], bt [batch, K, N] bf16 (second pair: same shapes)")
    out = torch.empty((Bn, M, N), dtype=torch.float32, device=at.device)
    _lib.check(_lib.lib().hh_gemm_tn_bf16_batched2(_p(at), _p(bt), _p(at2), _p(bt2), M, N, K * M, K * N, _p(out), M * N, M, N, K, Bn, _stream()),
               "hh_gemm_tn_bf16_batched2")
    return out


def xattn_fwd(q, k, v, heads, dropout_p=0.0, seed=0, splits=None):
    """q fp32 [B,Q,C] (pre-scaled); k, v bf16 [B,M,C] views with a common row stride -> (out fp32 [B,Q,C], lse [B,h,Q]).
    `splits` > 1 cuts the keys into slices (one workgroup per (clip, head, slice)); the default does so only when B*heads leaves
    most CUs idle (long clips at small batch)."""
    for t in (q, k, v):
        if not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors; there is no CPU fallback")
    B, Q, C = q.shape
    M = k.shape[1]
    if k.stride(1) != v.stride(1) or k.stride(2) != 1 or v.stride(2) != 1 or k.stride(0) != M * k.stride(1) or not q.is_contiguous():
        raise RuntimeError("xattn_fwd: k/v must be row-strided views [B,M,C] with dense batch stride")
    out = torch.empty_like(q)
    lse = torch.empty((B, heads, Q), dtype=torch.float32, device=q.device)
    if splits is None:
        splits = max(1, min(32, M // 512, 256 // max(B * heads, 1)))
    if splits > 1:
        ws = _workspace("xattn_fwd", B, Q, heads, int(splits), device=q.device)
        _lib.check(_lib.lib().hh_xattn_fwd_split(_p(q), _p(k), _p(v), k.stride(1), _p(out), _p(lse), _p(ws), int(splits), B, Q, M, heads,
                                                 float(dropout_p), int(seed) & 0xFFFFFFFF, _stream()), "hh_xattn_fwd_split")
    else:
        _lib.check(_lib.lib().hh_xattn_fwd(_p(q), _p(k), _p(v), k.stride(1), _p(out), _p(lse), B, Q, M, heads,
                                           float(dropout_p), int(seed) & 0xFFFFFFFF, _stream()), "hh_xattn_fwd")
    return out, lse


def xattn_bwd(q, k, v, out, lse, dout, dk, dv, heads, dropout_p=0.0, seed=0, splits=None):
    """Writes dk/dv (bf16 [B,M,C] row-strided views) in place; returns dq fp32 [B,Q,C].  The keys are cut into `splits` slices
    (one workgroup each per (clip, head)); the slices' dq partials are summed here in a fixed order."""
    B, Q, C = q.shape
    M = k.shape[1]
    _chk(q, out, lse, dout)
    if splits is None:
        # enough (clip, head, slice) workgroups to cover 256 CUs twice, slices never shorter than 512 keys
        splits = max(1, min(8, M // 1024), min(32, M // 512, -(-512 // (B * heads))))
    dq = _workspace("xattn_bwd", B, Q, heads, int(splits), device=q.device).view(splits, B, Q, C)
    _lib.check(_lib.lib().hh_xattn_bwd(_p(q), _p(k), _p(v), k.stride(1), _p(out), _p(lse), _p(dout), _p(dq), int(splits), _p(dk), _p(dv),
                                       dk.stride(1), B, Q, M, heads, float(dropout_p), int(seed) & 0xFFFFFFFF, _stream()),
               "hh_xattn_bwd")
    return dq[0] if splits == 1 else dq.sum(0)


def match_boxes(pred, q0, q, raw_boxes, img=224.0, w_l1=5.0, w_giou=2.0, given_count=None, class_cost=None, w_class=0.0, count_out=None):
    """pred fp32 [F,Qtot,4]; raw_boxes fp32 [F,k,4] -> dict(tgt, count, pred_idx, tgt_idx, n) all on device.
    given_count int32 [F]: raw_boxes are already-prepared cxcywh targets (list API of HungarianMatcher).
    class_cost fp32 [F,q,k]: -softmax(logits)[query, label of target j] (exclude_class=False, box_utils.py:83-85)."""
    _chk(pred, raw_boxes, given_count, class_cost)
    if class_cost is not None and (class_cost.dtype != torch.float32 or tuple(class_cost.shape) != (pred.shape[0], q, raw_boxes.shape[1])):
        raise ValueError("match_boxes: class_cost must be fp32 [F, q, k]")
    F_, Qtot, _ = pred.shape
    k = raw_boxes.shape[1]
    dev = pred.device
    tgt = torch.empty((F_, k, 4), dtype=torch.float32, device=dev)
    cnt = torch.empty((F_,), dtype=torch.int32, device=dev) if count_out is None else count_out      # (a row of a caller's [types, F] buffer)
    if cnt.dtype != torch.int32 or cnt.numel() != F_ or not cnt.is_contiguous():
        raise TypeError("match_boxes: count_out must be a contiguous int32 [F]")
    mp = torch.empty((F_, k), dtype=torch.int64, device=dev)
    mt = torch.empty((F_, k), dtype=torch.int64, device=dev)
    mn = torch.empty((F_,), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().hh_match_boxes(_p(pred), Qtot, q0, q, _p(raw_boxes), _p(given_count), k, float(img), float(w_l1), float(w_giou),
                                         _p(class_cost), float(w_class), _p(tgt), _p(cnt), _p(mp), _p(mt), _p(mn), F_, _stream()), "hh_match_boxes")
    return {"tgt": tgt, "count": cnt, "pred_idx": mp, "tgt_idx": mt, "n": mn}


def lsap_rows(cost, row_valid):
    """cost fp32 [P,nr,nc], row_valid uint8/bool [P,nr] -> int64 [P,nr] assigned column per valid row (-1 otherwise)."""
    _chk(cost, row_valid)
    P, nr, nc = cost.shape
    rv = row_valid.to(torch.uint8).contiguous()
    out = torch.empty((P, nr), dtype=torch.int64, device=cost.device)
    _lib.check(_lib.lib().hh_lsap_rows(_p(cost), _p(rv), _p(out), P, nr, nc, _stream()), "hh_lsap_rows")
    return out


def box_loss_fwd(pred, q0, m):
    _chk(pred)
    F_, Qtot, _ = pred.shape
    sums = torch.zeros(2, dtype=torch.float32, device=pred.device)
    _lib.check(_lib.lib().hh_box_loss_fwd(_p(pred), Qtot, q0, _p(m["tgt"]), m["tgt"].shape[1], _p(m["pred_idx"]),
                                          _p(m["tgt_idx"]), _p(m["n"]), _p(sums), F_, _stream()), "hh_box_loss_fwd")
    return sums


def box_loss_bwd(pred, q0, m, g_l1, g_giou, dpred):
    _chk(pred, g_l1, g_giou, dpred)
    F_, Qtot, _ = pred.shape
    _lib.check(_lib.lib().hh_box_loss_bwd(_p(pred), Qtot, q0, _p(m["tgt"]), m["tgt"].shape[1], _p(m["pred_idx"]),
                                          _p(m["tgt_idx"]), _p(m["n"]), _p(g_l1), _p(g_giou), _p(dpred), F_, _stream()),
               "hh_box_loss_bwd")
    return dpred


def box_tail_fwd(pred, m_h, m_o, q0_h, q_h, q0_o, q_o, num_boxes, argmax, no_object, w_l1, w_giou, denom):
    """Both box types of the step at once (include/hh.h: hh_box_loss_fwd x 2 + hh_box_tail_fwd): -> (out fp32 [8], coef fp32 [4])."""
    _chk(pred, num_boxes, argmax)
    F_, Qtot, _ = pred.shape
    sums = torch.zeros(4, dtype=torch.float32, device=pred.device)
    L = _lib.lib()
    for m, q0, off in ((m_h, q0_h, 0), (m_o, q0_o, 2)):
        _lib.check(L.hh_box_loss_fwd(_p(pred), Qtot, q0, _p(m["tgt"]), m["tgt"].shape[1], _p(m["pred_idx"]), _p(m["tgt_idx"]), _p(m["n"]),
                                     ctypes.c_void_p(sums.data_ptr() + 4 * off), F_, _stream()), "hh_box_loss_fwd")
    out = torch.empty(8, dtype=torch.float32, device=pred.device)
    coef = torch.empty(4, dtype=torch.float32, device=pred.device)
    if num_boxes.dtype != torch.float32 or num_boxes.numel() < 2 or (argmax is not None and (argmax.dtype != torch.int64 or argmax.shape[0] != F_)):
        raise TypeError("box_tail_fwd: num_boxes fp32 [>= 2], argmax int64 [F, Q]")
    _lib.check(L.hh_box_tail_fwd(_p(sums), ctypes.c_void_p(sums.data_ptr() + 8), _p(num_boxes), _p(m_h["count"]), _p(m_o["count"]), _p(argmax),
                                 0 if argmax is None else argmax.shape[1], q0_h, q_h, q0_o, q_o, int(no_object), F_, float(w_l1), float(w_giou),
                                 float(denom), _p(out), _p(coef), _stream()), "hh_box_tail_fwd")
    return out, coef


def box_tail_bwd(pred, m_h, m_o, q0_h, q0_o, g_h, g_o, coef):
    """d (g_h * total_h + g_o * total_o) / d pred: two launches into one zeroed buffer (the box types own disjoint query slices)."""
    _chk(pred, g_h, g_o, coef)
    F_, Qtot, _ = pred.shape
    dpred = torch.zeros_like(pred)
    L = _lib.lib()
    for m, q0, g, off in ((m_h, q0_h, g_h, 0), (m_o, q0_o, g_o, 2)):
        _lib.check(L.hh_box_loss_bwd_scaled(_p(pred), Qtot, q0, _p(m["tgt"]), m["tgt"].shape[1], _p(m["pred_idx"]), _p(m["tgt_idx"]), _p(m["n"]), _p(g),
                                            ctypes.c_void_p(coef.data_ptr() + 4 * off), ctypes.c_void_p(coef.data_ptr() + 4 * off + 4), _p(dpred), F_,
                                            _stream()), "hh_box_loss_bwd_scaled")
    return dpred


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    _chk(p, g, m, v)
    _lib.check(_lib.lib().hh_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                        float(eps), float(weight_decay), int(step), _stream()), "hh_adamw_step")


def adamw_arena_step(p, g, m, v, seg_off, seg_decay, seg_step, seg_flag, seg_coef, lr, beta1, beta2, eps, weight_decay, zero_grads=True):
    """hh_adamw_arena_step: AdamW over the whole flat arena, per-parameter skip / step count / bias correction decided on the device."""
    _chk(p, g, m, v)
    n_seg = seg_decay.numel()
    assert seg_off.dtype == torch.int64 and seg_off.numel() == n_seg + 1 and seg_decay.dtype == torch.int32
    assert seg_step.dtype == torch.int32 and seg_flag.dtype == torch.float32 and seg_coef.numel() >= 2 * n_seg
    _lib.check(_lib.lib().hh_adamw_arena_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_off), _p(seg_decay), _p(seg_step),
                                              _p(seg_flag), _p(seg_coef), n_seg, float(lr), float(beta1), float(beta2), float(eps),
                                              float(weight_decay), int(bool(zero_grads)), _stream()), "hh_adamw_arena_step")


# ---- loss tail (csrc/loss.hip)
def _chk_gpu(*tensors):
    """Row-strided operands are fine for these kernels (they take a leading dimension): only the device is checked."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libhh ops need GPU tensors (got device %s); there is no CPU fallback" % t.device)


def text_flags(text):
    """text int64 [rows, L] -> (eot int64 [rows] = argmax over L, pad fp32 [rows] = (#non-zero tokens != 2)); include/hh.h: hh_text_flags."""
    _chk(text)
    if text.dtype != torch.int64 or text.dim() != 2:
        raise TypeError("text_flags: int64 [rows, L]")
    rows, L = text.shape
    eot = torch.empty(rows, dtype=torch.int64, device=text.device)
    pad = torch.empty(rows, dtype=torch.float32, device=text.device)
    _lib.check(_lib.lib().hh_text_flags(_p(text), rows, L, _p(eot), _p(pad), _stream()), "hh_text_flags")
    return eot, pad


def rownorm_fwd(x, eps=1e-8):
    """x fp32 [rows, cols] (unit inner stride) -> (y = x / max(||x||, eps) dense fp32, norm fp32 [rows])."""
    _chk_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise TypeError("rownorm_fwd: fp32 [rows, cols] with unit inner stride")
    rows, cols = x.shape
    y = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hh_rownorm_fwd(_p(x), x.stride(0) if rows > 1 else cols, _p(y), _p(norm), rows, cols, float(eps), _stream()), "hh_rownorm_fwd")
    return y, norm


def rownorm_bwd(y, norm, dy, eps=1e-8):
    _chk(y, norm)
    _chk_gpu(dy)
    if dy.stride(-1) != 1 or dy.dtype != torch.float32:
        dy = dy.float().contiguous()
    rows, cols = y.shape
    dx = torch.empty_like(y)
    _lib.check(_lib.lib().hh_rownorm_bwd(_p(y), _p(norm), _p(dy), dy.stride(0) if rows > 1 else cols, _p(dx), rows, cols, float(eps), _stream()), "hh_rownorm_bwd")
    return dx


def egonce_fwd(x, sim_v, sim_n, pad, R, temperature, vn_threshold=0.0):
    """-> (loss fp32 [1], grad fp32 [R*Bg, Bg] = d loss / d x).  x fp32 [R*Bg, Bg] (unit inner stride), pad fp32 [R*Bg]."""
    _chk(sim_v, sim_n, pad)
    _chk_gpu(x)
    Rn, Bg = x.shape
    if x.dtype != torch.float32 or x.stride(1) != 1 or Rn != R * Bg or pad.numel() != Rn:
        raise TypeError("egonce_fwd: x fp32 [R*Bg, Bg], pad [R*Bg]")
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    grad = torch.empty((Rn, Bg), dtype=torch.float32, device=x.device)
    scratch = _workspace("egonce", R, Bg, device=x.device)
    _lib.check(_lib.lib().hh_egonce_fwd(_p(x), x.stride(0), _p(sim_v), _p(sim_n), _p(pad), R, Bg, float(temperature), float(vn_threshold),
                                        _p(loss), _p(grad), _p(scratch), _stream()), "hh_egonce_fwd")
    return loss, grad


def masked_ce_fwd(sim, noun_sim, gt, valid, temperature, threshold):
    """-> (ce fp32 [rows], grad fp32 [rows, V]); sim fp32 [rows, V], noun_sim fp32 [V, V], gt int64 [rows], valid bool / uint8 [rows]."""
    _chk_gpu(sim, noun_sim, gt, valid)
    if sim.dim() != 2 or noun_sim.dim() != 2 or noun_sim.shape != (sim.shape[1], sim.shape[1]):
        raise ValueError("masked_ce_fwd: sim [rows, V] and noun_sim [V, V] expected, got %s / %s" % (tuple(sim.shape), tuple(noun_sim.shape)))
    if gt.dtype.is_floating_point or gt.dtype == torch.bool or valid.dtype not in (torch.bool, torch.uint8):
        raise TypeError("masked_ce_fwd: gt must be an integer tensor, valid bool / uint8")
    # the kernel reads fp32 / int64 through raw pointers: a bf16 similarity (autocast) or int32 ids must be converted, not reinterpreted
    if sim.dtype != torch.float32 or sim.stride(1) != 1:
        sim = sim.float().contiguous()
    noun_sim = noun_sim.float().contiguous()
    gt = gt.to(torch.int64).contiguous()
    valid = valid.contiguous()
    rows, V = sim.shape
    if gt.numel() != rows or valid.numel() != rows:
        raise ValueError("masked_ce_fwd: gt / valid must have one entry per row of sim")
    ce = torch.empty(rows, dtype=torch.float32, device=sim.device)
    grad = torch.empty((rows, V), dtype=torch.float32, device=sim.device)
    v8 = valid.view(torch.uint8) if valid.dtype == torch.bool else valid
    _lib.check(_lib.lib().hh_masked_ce_fwd(_p(sim), sim.stride(0), _p(noun_sim), _p(gt), _p(v8), rows, V, float(temperature), float(threshold),
                                           _p(ce), _p(grad), _stream()), "hh_masked_ce_fwd")
    return ce, grad


def tv_accuracy(sim, text_cos, sim_v, sim_n):
    """compute_tv_accuracy on the device: sim fp32 [Bg, Bg] (row-strided view allowed) -> fp32 [2] = (acc video->text, acc text->video)."""
    _chk(text_cos, sim_v, sim_n)
    _chk_gpu(sim)
    Bg = sim.shape[0]
    if sim.stride(1) != 1:
        sim = sim.contiguous()
    out = torch.empty(2, dtype=torch.float32, device=sim.device)
    _lib.check(_lib.lib().hh_tv_accuracy(_p(sim), sim.stride(0), _p(text_cos), _p(sim_v), _p(sim_n), Bg, _p(out), _stream()), "hh_tv_accuracy")
    return out
