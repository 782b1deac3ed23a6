"""CPU-side checks of the C-ABI boundary: libhh.so builds for gfx950, loads, and exports every symbol that
include/hh.h declares (no compute calls without a GPU); the Python binding table covers the same set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hh_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from helping_hand_for_egocentric_videos_amd import build, _lib
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hh.h but not exported by libhh.so"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    assert L.hh_version() >= 100
    assert isinstance(L.hh_last_error_string(), bytes)
    # the option structs of the binding are the ones the library was compiled with (they grow by appended fields: a stale binding would hand
    # the kernels garbage; _lib.lib() raises on a mismatch at load time)
    assert L.hh_abi_sizeof(b"hh_gemm_epilogue") == ctypes.sizeof(_lib.GemmEpilogue) > 0
    assert L.hh_abi_sizeof(b"hh_qgemm_opts") == ctypes.sizeof(_lib.QGemmOpts) > 0
    assert L.hh_abi_sizeof(b"hh_qgemm_item") == ctypes.sizeof(_lib.QGemmItem) > 0
    assert L.hh_abi_sizeof(b"no_such_struct") == -1


def test_host_side_argument_validation_needs_no_gpu():
    """Shape/alignment errors are reported by the C ABI before any launch (status < 0 + message)."""
    from helping_hand_for_egocentric_videos_amd import _lib
    L = _lib.lib()
    e = _lib.GemmEpilogue()
    rc = L.hh_gemm_bf16(None, 100, None, 100, None, 100, 4, 100, 100, ctypes.byref(e), None)
    assert rc == -1 and b"hh_gemm_bf16" in L.hh_last_error_string()
    assert L.hh_space_attn_fwd(None, 0, None, None, 1, 4, 250, 2, None) == -1
    assert L.hh_space_attn_fwd(None, 7, None, None, 1, 4, 256, 2, None) == -1          # unknown qkv layout
    assert L.hh_time_attn_fwd(None, 1, None, None, 1, 33, 256, 2, None) == -3         # (round 6: every T <= 32 is supported)
    assert L.hh_xattn_fwd(None, None, None, 512, None, None, 1, 17, 4096, 8, 0.0, 0, None) == -1
    assert L.hh_adamw_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None) == -1
    # query-side entry points (round 2)
    o = _lib.QGemmOpts()
    assert L.hh_qgemm_f32x3(None, 512, None, 512, None, 512, 16, 512, 510, 0, ctypes.byref(o), None) == -1       # K % 4 != 0
    assert b"multiples of 4" in L.hh_last_error_string()
    assert L.hh_qgemm_f32x3(None, 512, None, 512, None, 512, 16, 512, 512, 3, ctypes.byref(o), None) == -1       # bad mode
    o.splitk, o.relu = 4, 1
    assert L.hh_qgemm_f32x3(None, 512, None, 512, None, 512, 16, 512, 512, 0, ctypes.byref(o), None) == -3       # split-K takes no epilogue
    assert L.hh_qself_attn_fwd(None, None, 2, 17, 8, 0.0, 0, None) == -1                                           # Q <= 16
    # grouped products (round 6): 1 .. HH_QGEMM_GROUP_MAX items of one mode, each checked like a single call
    items = (_lib.QGemmItem * 13)()
    for it in items:
        it.lda = it.ldb = it.ldc = 512
        it.M, it.N, it.K, it.mode = 16, 512, 512, 2
    assert L.hh_qgemm_f32x3_group(items, 0, None) == -1 and L.hh_qgemm_f32x3_group(items, 13, None) == -1
    assert b"products per launch" in L.hh_last_error_string()
    items[1].mode = 1
    assert L.hh_qgemm_f32x3_group(items, 2, None) == -3 and b"share one mode" in L.hh_last_error_string()
    items[1].mode, items[1].N = 2, 510
    assert L.hh_qgemm_f32x3_group(items, 2, None) == -1 and b"multiples of 4" in L.hh_last_error_string()
    assert L.hh_sum_partials(None, None, 0, 16, None) == -1 and L.hh_sum_partials(None, None, 4, 6, None) == -1
    assert L.hh_layernorm_pos_fwd(None, 0, None, None, None, None, 0, None, 13, None, None, 10, 512, 1e-5, None) == -1
    # caller-owned workspace sizes and the timing facility need no GPU
    assert L.hh_workspace_bytes_gemm_tn(512, 2048, 3) == 3 * 512 * 2048 * 4
    assert L.hh_workspace_bytes_xattn_bwd(32, 13, 8, 4) == 4 * 32 * 13 * 512 * 4
    assert L.hh_workspace_bytes_attn_cls_partial(32, 16, 256, 16, 0) == 32 * 16 * 16 * 68 * 4
    assert L.hh_workspace_bytes_attn_cls_partial(32, 16, 256, 16, 1) == 32 * 16 * 32 * 68 * 4
    assert L.hh_workspace_bytes_gemm_splitk(-1, 128, 2) < 0
    assert L.hh_set_tuning(b"gemm256", 9) == -3 and L.hh_set_tuning(b"gemm256", 5) == 0
    assert L.hh_set_tuning(b"no_such_knob", 1) == -3


def test_product_ops_refuse_cpu_tensors():
    import torch
    from helping_hand_for_egocentric_videos_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.qgemm(torch.zeros(4, 64), torch.zeros(128, 64))
    from helping_hand_for_egocentric_videos_amd.model.qside import LinearX3
    with pytest.raises(RuntimeError, match="libhh HIP kernels only"):
        LinearX3(8, 8)(torch.zeros(2, 8))


def test_header_is_plain_c(tmp_path):
    """include/hh.h is the drop-in boundary: it must compile as C99 on its own (plain pointers and sizes, no C++ in the signatures)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hh.h")
    r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
