"""Package power (rocm-smi) while ONE kind of kernel loops for ~4 s: which phases of the step spend the 1400 W budget."""
import os, sys, time, subprocess, threading, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helping_hand_for_egocentric_videos_amd import ops
B, T, n, heads = 32, 16, 256, 16
N = 1 + T * n; M = B * N; D = 1024
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.randn(M, D, device="cuda", generator=g)).to(torch.bfloat16)
w1 = (torch.randn(4096, D, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
h = (torch.randn(M, 4096, device="cuda", generator=g)).to(torch.bfloat16)
w2 = (torch.randn(D, 4096, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
b1 = torch.zeros(4096, device="cuda"); b2 = torch.zeros(D, device="cuda")
x = torch.randn(M, D, device="cuda", generator=g)
gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
qkv = (torch.randn(M, 3 * D, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
samples = []
def sampler(stop):
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
        m = re.search(r"Power \(W\): ([0-9.]+)", out)
        if m: samples.append(float(m.group(1)))
        time.sleep(0.25)
def run(name, f, work, unit):
    for _ in range(3): f()
    torch.cuda.synchronize()
    samples.clear(); stop = threading.Event(); th = threading.Thread(target=sampler, args=(stop,)); th.start()
    t0 = time.time(); it = 0
    while time.time() - t0 < 4.0:
        for _ in range(20): f()
        torch.cuda.synchronize(); it += 20
    dt = time.time() - t0; stop.set(); th.join()
    s = sorted(samples[2:]) or [0]
    print(f"{name:26s} {dt/it*1e6:8.1f} us/call  {work/(dt/it)/1e12:8.2f} {unit}  power median {s[len(s)//2]:6.0f} W (max {s[-1]:.0f})  energy/call {s[len(s)//2]*dt/it*1e3:7.2f} mJ", flush=True)
    time.sleep(1.0)
run("idle (sync only)", lambda: torch.cuda.synchronize(), 0, "-")
run("gemm fc1 (1024->4096)", lambda: ops.gemm(a, w1, b1, act=ops.ACT_QUICKGELU), 2.0 * M * 4096 * D, "TFLOP/s")
run("gemm fc2 (4096->1024)", lambda: ops.gemm(h, w2, b2), 2.0 * M * 4096 * D, "TFLOP/s")
run("torch matmul fc1 (hipBLASLt)", lambda: torch.nn.functional.linear(a, w1), 2.0 * M * 4096 * D, "TFLOP/s")
run("add_layernorm (x+=d, LN)", lambda: ops.add_layernorm(x, a, gam, bet, 1e-6, write_x=True), 12.0 * M * D, "TB/s")
run("space attention", lambda: ops.divided_attention(qkv, B, T, n, heads, "space"), 8.0 * M * D, "TB/s")
run("time attention", lambda: ops.divided_attention(qkv, B, T, n, heads, "time"), 8.0 * M * D, "TB/s")
