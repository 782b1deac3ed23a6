"""Static check of the device assembly of the hand-written kernels for a hazard hipcc cannot see: an SGPR written by a VALU instruction
(v_readlane_b32 = the reload of a spilled SGPR, v_readfirstlane_b32) and read as the ADDRESS of a vector-memory instruction that sits
inside an inline-asm statement less than 5 instructions later (gfx9: VALU writes SGPR -> VMEM reads that SGPR needs 5 wait states; the
compiler's hazard recogniser pads its own instructions but does not parse inline asm).  Round 4: the persistent GEMM's tile-counter
atomic read a stale pointer this way once SGPR pressure made the compiler spill it (memory access fault).  Also checked: (2) compiler
traffic into asm-owned accumulator registers, (3) destinations of asm-issued loads touched before a vmcnt wait (see scan()).

    python scripts/check_isa_hazards.py [file.hip ...]        (default: every csrc/*.hip that contains inline-asm VMEM with an "s" operand)
Exit code 1 and a listing if a hazard is found."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "helping_hand_for_egocentric_videos_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
WAIT_STATES = 5


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S"]


def _cache_key():
    """The cached assembly depends on the compiler and its flags as well as on the sources: hash both into the file name."""
    import hashlib
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError):
        ver = "unknown"
    return hashlib.sha1((ver + " ".join(FLAGS)).encode()).hexdigest()[:10]


def device_asm(src):
    out = os.path.join(tempfile.gettempdir(), "hh_isa_%s_%s.s" % (_cache_key(), os.path.basename(src)))
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), max(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC) if h.endswith(".h"))):
        subprocess.run([HIPCC] + FLAGS + [src, "-o", out], check=True, capture_output=True)
    return out


def scan(path):
    lines = open(path).read().split("\n")
    insts = []                      # (line number, text, inside inline asm)
    in_asm, func = False, ""
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if raw and not raw[0].isspace() and not t.startswith((".", ";")) and ":" in t:
            func = t.split(":")[0]
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        insts.append((i + 1, t, in_asm, func))
    bad = []
    # (2) kernels whose inline asm OWNS the accumulator file (gemm256w4p_kernel: a[0:255] hold the tile across asm statements; the compiler
    # only sees a clobber list) must not have compiler-generated traffic into AGPRs: under register pressure hipcc spills VGPRs into
    # "free" AGPRs (v_accvgpr_write_b32 aN, vM / loads with an a[..] destination) -- which silently corrupts the accumulators
    for ln, t, in_asm, func in insts:
        if "gemm256w4p_kernel" in func and not in_asm:
            if re.match(r"v_accvgpr_write_b32 a\d+, v\d+", t) or re.match(r"(global|buffer|ds)_(load|read)\S* a\[", t) or re.match(r"v_accvgpr_mov", t):
                bad.append((func, ln, t, ln, "compiler-generated write into the asm-owned accumulator registers (VGPR spill to AGPR)", 0))
    # (3) a vector load issued by INLINE ASM (opaque to the compiler: it believes the destination is valid when the statement "returns")
    # whose destination registers are read, written or copied by any instruction before the next s_waitcnt that names vmcnt: the register
    # allocator is free to move such a value, and a copy taken before the load has landed is garbage (round 5: whole waves of wrong space
    # attention outputs, timing-dependent -- the Q fragments, loaded long before the wait that makes them valid)
    for k, (ln, t, in_asm, func) in enumerate(insts):
        m = in_asm and re.match(r"global_load_dword(x\d)? (v\[(\d+):(\d+)\]|v(\d+)),", t)
        if not m:
            continue
        lo, hi = (int(m.group(3)), int(m.group(4))) if m.group(3) else (int(m.group(5)), int(m.group(5)))
        for ln2, t2, in_asm2, func2 in insts[k + 1:k + 1 + 2000]:
            # (a linear walk of the fall-through path: it ends at the wait, at an unconditional branch or at the end of the function)
            if func2 != func or (t2.startswith("s_waitcnt") and "vmcnt" in t2) or t2.startswith(("s_branch", "s_endpgm", "s_setpc")):
                break
            regs = set()
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t2):
                regs.update(range(int(a), int(b) + 1))
            regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t2))
            if any(lo <= r <= hi for r in regs):
                bad.append((func, ln, t, ln2, t2 + "   <- touches the destination of an asm load that no vmcnt wait has made valid", 0))
                break
    for k, (ln, t, _, func) in enumerate(insts):
        m = re.match(r"v_(readlane|readfirstlane)_b32 (s\d+)", t)
        if not m:
            continue
        sreg = int(m.group(2)[1:])
        states = 0
        for ln2, t2, in_asm2, _ in insts[k + 1:k + 1 + WAIT_STATES + 4]:
            if states >= WAIT_STATES:
                break
            mm = re.match(r"s_nop (\d+)", t2)
            if in_asm2 and re.match(r"(global|buffer|flat|scratch)_", t2):
                for a, b in re.findall(r"s\[(\d+):(\d+)\]", t2):
                    if int(a) <= sreg <= int(b):
                        bad.append((func, ln, t, ln2, t2, states))
            states += (int(mm.group(1)) + 1) if mm else 1
    return bad


def main(argv):
    files = argv or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and re.search(r'asm volatile\([^;]*(global_|buffer_)[^;]*"(s|=v)"\(', open(os.path.join(CSRC, f)).read(), re.S)]
    total = 0
    for f in files:
        bad = scan(device_asm(f))
        print("%s: %d hazard(s)" % (os.path.basename(f), len(bad)))
        for func, ln, t, ln2, t2, st in bad:
            print("   %s\n      line %d: %s\n      line %d: %s   (%d wait states between)" % (func[:90], ln, t, ln2, t2, st))
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
