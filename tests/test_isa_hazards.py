"""Static check of the compiled kernels for a hazard the compiler cannot see (CPU test: hipcc cross-compiles gfx950 assembly here):
an SGPR reloaded by a VALU instruction (v_readlane_b32: a spilled pointer) and used as the address of a vector-memory instruction INSIDE
an inline-asm statement fewer than 5 wait states later.  Found in round 4 as a memory access fault of the persistent GEMM's tile-counter
atomic; scripts/check_isa_hazards.py scans every file whose inline asm issues VMEM with a scalar address or loads into VGPRs (round 5: a
fragment loaded by inline asm and copied / spilled by the compiler before the wait that makes it valid)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_sgpr_reload_to_inline_asm_vmem_hazard():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not found: the check compiles the kernels to gfx950 assembly")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_isa_hazards.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gemm256w4.hip: 0 hazard(s)" in r.stdout, r.stdout
    # round 5: destinations of asm-issued vector loads (the Q fragments of the progressive space-attention kernel) are not touched -- copied,
    # spilled, overwritten -- before a vmcnt wait has made them valid
    assert "attn_space.hip: 0 hazard(s)" in r.stdout and "mattn.hip: 0 hazard(s)" in r.stdout, r.stdout
    # round 6: the loader waves of the persistent 32x32x16 space-attention kernel request two problems' K / V / Q by inline asm into two register
    # sets and wait with a counted vmcnt a whole problem later
    assert "attn_space32.hip: 0 hazard(s)" in r.stdout, r.stdout
