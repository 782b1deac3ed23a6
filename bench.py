"""Benchmark of the hot path: train clips/sec (16-frame 224p, nq=12) -- BASELINE.json's metric -- on N GPUs.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N --steps K --warmup W            (plain: starts its own N ranks as a child `torch.distributed.run`, self_launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = frozen TimeSformer-L forward + text tower + object-query decoder forward/backward + EgoNCE / box /
word losses (on-device Hungarian) + decoder-gradient all-reduce (RCCL) + fused AdamW over one synthetic batch
that is resident in HBM (the same batch every step).  Weak scaling: the per-GPU batch is fixed.  Prints ONE JSON line on rank 0:
`value` (+ per-step mean / std / p50 / p95 from device events), `roofline` (dominant kernel), `attention_roofline`, and on one GPU
the sub-records `mcq` (EgoMCQ forward, BASELINE config 5), `c4` (BASELINE config 4: 32-frame 336p, the HBM-bound attention
stress, with its own kernel roofline records) and `cpu_baseline` (the oracle on the host: config 2 on one clip and config 1 --
T = 4, nq = 4, B = 2 -- exactly).  --config c4|c1 benches that configuration as the main workload instead.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from helping_hand_for_egocentric_videos_amd import C1, C2, C4, ops, synth  # noqa: E402
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder  # noqa: E402
from helping_hand_for_egocentric_videos_amd.step import McqScorer, TrainStep, first_clips_check, mcq_forward  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
CONFIGS = {"c1": (C1, "C1: 4-frame 224p, nq=4"), "c2": (C2, "C2: 16-frame 224p, nq=12"), "c4": (C4, "C4: 32-frame 336p, nq=12")}
STRIDE = 5      # every 5th launch of a kernel class is timed (co-prime with the 6 GEMMs / 2 attention calls per block); bracketing every launch costs ~3 % in marker packets

PROF = {"gemm256": 0, "gemm_other": 1, "space_attn": 2, "time_attn": 3, "add_ln": 4, "gemm_tn": 5, "xattn_fwd": 6, "xattn_bwd": 7}


def step_tflop_per_clip(cfg, train=True):
    """Algorithmic work of one clip through the step in TFLOP, BASELINE.md section 3's accounting: encoder = 24*(32*N*D^2 + 4*D*[n*T*(T+1)
    + T*n*(n+1) + 2N]) + 2*T*n*588*D, text tower 66.5 GFLOP per 5 captions (2.7 GFLOP per clip for EgoMCQ's one query per 5 clips),
    decoder forward / train from the table (C1 8.7 / 24.7, C2 35.6 / 103, C4 146 / 434 GFLOP; other shapes scale with the memory
    length).  Config 2: 3.59 train, 3.45 EgoMCQ forward."""
    N, D, T, n = cfg.tokens, cfg.embed_dim, cfg.num_frames, cfg.patches_per_frame
    enc = cfg.depth * (32 * N * D * D + 4 * D * (n * T * (T + 1) + T * n * (n + 1) + 2 * N)) + 2 * T * n * 588 * D
    table = {(4, 224, 4): (8.7e9, 24.7e9), (16, 224, 12): (35.6e9, 103e9), (32, 336, 12): (146e9, 434e9)}
    dec_f, dec_t = table.get((T, cfg.img_size, cfg.num_queries), (35.6e9 * T * n / 4096, 103e9 * T * n / 4096))
    return (enc + (66.5e9 + dec_t if train else 2.7e9 + dec_f)) / 1e12


def prof_enable(stride):
    """Library-side per-kernel timing (include/hh.h: hh_prof_enable): every stride-th launch of each instrumented kernel class is
    bracketed by two HIP events on its launch stream with only that kernel between them.  stride 0 = off."""
    from helping_hand_for_egocentric_videos_amd import _lib
    _lib.check(_lib.lib().hh_prof_enable(int(stride)), "hh_prof_enable")


ROLES = {"decoder": 0, "vision_tower": 1, "text_tower": 2}      # include/hh.h: hh_prof_set_role (the part of the step the host was launching)


def prof_read(name, role=-1):
    """-> (launches timed, launches seen, total ms, total algorithmic work) of one kernel class (one role, or all: -1); synchronises on its events."""
    import ctypes
    from helping_hand_for_egocentric_videos_amd import _lib
    n, seen, ms, work = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib().hh_prof_read_role(PROF[name], int(role), ctypes.byref(n), ctypes.byref(seen), ctypes.byref(ms), ctypes.byref(work)), "hh_prof_read_role")
    return n.value, seen.value, ms.value, work.value


def prof_kernel_name(name):
    """The kernel (template instantiation as rocprofv3 prints it) the library dispatched for the class's last launch (hh_prof_kernel_name)."""
    from helping_hand_for_egocentric_videos_amd import _lib
    r = _lib.lib().hh_prof_kernel_name(PROF[name])
    return r.decode() if r else ""


def prof_snapshot():
    snap = {k: prof_read(k) for k in PROF}
    snap["names"] = {k: prof_kernel_name(k) for k in PROF}
    snap["roles"] = {r: {k: prof_read(k, i) for k in PROF} for r, i in ROLES.items()}
    return snap


VISION_GEMMS = ("gemm256w4p_kernel<true, 5,", "gemm256w4p_kernel<true, 6,", "gemm256w4p_kernel<true, 7,", "gemm256w4p_kernel<true, 8,")
# the vision tower's instantiations of the persistent GEMM with the LayerNorm fold: <5> = qkv (consumer), <6> = fc1 (consumer, QuickGELU),
# <7> = time projection (producer, reads z3), <8> = space projection and fc2 (producers on the bf16 pair stream).  <true, 0|1|2> are the text
# tower's / small shapes and are NOT part of `roofline` (their 110-170 MiB launches used to dilute the mean)


def pmc_traffic(kernel_substrs, cfg_tag="", tag_order=("r6", "r5", "r4", "r3", "r2", "r1")):
    """HBM-side bytes per launch from the COMMITTED PMC passes of the same configuration (profiles/<round>_[c4_]pmc_summary.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction); launch-weighted mean over the
    template instantiations whose name contains one of `kernel_substrs`.  NOT measured in this run.  Returns (bytes, file) or (None, None)."""
    if isinstance(kernel_substrs, str):
        kernel_substrs = (kernel_substrs,)
    for tag in tag_order:
        path = os.path.join(ROOT, "profiles", tag + "_" + cfg_tag + "pmc_summary.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rows = [v for k, v in json.load(f).items() if any(t in k for t in kernel_substrs)]
        n = sum(v["launches"] for v in rows)
        if n:
            return int(sum(v["traffic_bytes_per_launch"] * v["launches"] for v in rows) / n), os.path.basename(path)
    return None, None


def rocprof_committed(kernel_substr, cfg_tag="", pipelined=True, tag_order=("r6", "r5", "r4", "r3")):
    """Average duration per launch (us) and launch count of the kernels whose name contains `kernel_substr` in the committed rocprofv3
    --kernel-trace --stats CSV of the same bench command (profiles/<round>_[c4_]{bench,unpipelined}_kernel_stats.csv).  -> dict or None."""
    import csv
    for tag in tag_order:
        path = os.path.join(ROOT, "profiles", "%s_%s%s_kernel_stats.csv" % (tag, cfg_tag, "bench" if pipelined else "unpipelined"))
        if not os.path.exists(path):
            continue
        calls, total_ns = 0, 0.0
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row.get("Name", ""):
                    calls += int(row["Calls"]); total_ns += float(row["TotalDurationNs"])
        if calls:
            return {"avg_launch_us": round(total_ns / calls / 1e3, 1), "launches": calls, "total_ms": round(total_ns / 1e6, 2), "file": os.path.basename(path)}
    return None


def gemm_algorithmic_table(cfg, B, ln_fold=True, pair=True, z3=True):
    """Operand + result bytes of the vision tower's six GEMMs of a block, one entry per launch -- what the PMC traffic of the persistent GEMM
    is compared with.  bf16 A [M,K] and W [N,K] read once; the consumers (qkv x2, fc1) write bf16 C [M,N]; with the LayerNorm fold the three
    branch-ending GEMMs write no C: per element of [M, D] the time projection reads its residual input (z3 = bf16 image: 2 B; fp32: 4 B) and
    writes z (2 B); space projection and fc2 read and write the residual stream -- the bf16 pair hi + lo (4 B read, 4 B written: hi' IS the
    next consumer's z) or, on the fp32 stream, 4 B read, 4 B written + 2 B of z.  (Row statistics: 8 B per row and 128 columns, < 0.1 %.)"""
    M, D = B * cfg.tokens, cfg.embed_dim
    aw = lambda N, K: 2 * (M * K + N * K)
    if not ln_fold:
        return {k: aw(N, K) + 2 * M * N for k, (N, K) in {"qkv_time": (3 * D, D), "qkv_space": (3 * D, D), "proj_time": (D, D), "proj_space": (D, D),
                                                          "fc1": (4 * D, D), "fc2": (D, 4 * D)}.items()}
    upd = (4 + 4) if pair else (4 + 4 + 2)
    return {"qkv_time": aw(3 * D, D) + 2 * M * 3 * D, "qkv_space": aw(3 * D, D) + 2 * M * 3 * D,
            "proj_time": aw(D, D) + M * D * ((2 if z3 else 4) + 2), "proj_space": aw(D, D) + M * D * upd,
            "fc1": aw(4 * D, D) + 2 * M * 4 * D, "fc2": aw(D, 4 * D) + M * D * upd}


def gemm_algorithmic_bytes(cfg, B, ln_fold=True, pair=True, z3=True):
    """Launch-weighted mean of gemm_algorithmic_table over the six GEMMs of a block (C2, B = 32, pair stream: 1.30 GB)."""
    t = gemm_algorithmic_table(cfg, B, ln_fold, pair, z3)
    return int(sum(t.values()) / len(t))


def _read_timeline():
    import ctypes
    import numpy as np
    from helping_hand_for_egocentric_videos_amd import _lib
    buf = np.zeros((256, 8, 7), dtype=np.uint64)
    _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "hh_debug_gemm_timeline")
    return buf.astype(np.int64)


def in_step_clock(backbone, video, K=4096):
    """Shader clock inside the persistent GEMM's main loop, measured ON THE STEP'S OWN OPERANDS: the vision tower runs over the bench
    batch with the kernel's timeline stamps on (hh_set_tuning("gemm256_debug_ts")); the record that survives is the tower's LAST
    persistent launch -- block 23's fc2 GEMM (K = 4096), the 144th tall GEMM in a row, with the activations and weights of the step.
    s_memtime ticks per s_memrealtime microsecond, read by the kernel itself.  MI355X throttles below its 2.4 GHz nominal clock when
    the matrix cores are busy (the step sits at the 1400 W package cap), so the datasheet 2.5 PFLOP/s is not reachable by ANY bf16
    GEMM here; the clock-limited peak is 2.5 * f / 2.4."""
    import numpy as np
    with torch.no_grad():
        backbone.visual.forward_features(video, out_dtype=torch.bfloat16)          # settle the governor
        ops.set_tuning("gemm256_debug_ts", 1)
        backbone.visual.forward_features(video, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        ops.set_tuning("gemm256_debug_ts", 0)
    t = _read_timeline()
    rows = video.shape[0] * (1 + video.shape[1] * (video.shape[3] // 14) ** 2)
    ntile = max(1, min(8, (rows // 256) * (1024 // 256) // 256))
    t = t[:, :ntile]
    us = (t[:, :, 2] - t[:, :, 1]) / 100.0               # main loop, 100 MHz real-time counter
    mhz = float(((t[:, :, 6] - t[:, :, 5]) / np.maximum(us, 1e-9)).mean())
    return {"shader_mhz_in_gemm_main_loop": round(mhz, 1), "nominal_mhz": 2400,
            "where": "last persistent GEMM of a vision-tower forward over the bench batch (block 23 fc2, K = 4096): the step's own activations and weights",
            "main_loop_us_per_ktile": round(float(us.mean()) / (K // 64), 3),
            "clock_limited_peak": round(PEAK_BF16_TFLOPS * mhz / 2400.0, 1)}


class PowerSampler:
    """Package power / clock samples while the timed region runs, read by a CHILD process (`rocm-smi` every ~0.4 s; this process has
    initialised the GPU and must not exec anything itself)."""
    CODE = r"""
import subprocess, sys, time
out = open(sys.argv[1], "w")
while True:
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
    except Exception as e:
        r = ""
    p = [l.split(":")[-1].strip() for l in r.splitlines() if "Package Power" in l]
    s = [l.split("(")[-1].split("Mhz")[0] for l in r.splitlines() if "sclk" in l]
    out.write("%.2f %s %s\n" % (time.time(), p[0] if p else "nan", s[0] if s else "nan")); out.flush()
    time.sleep(0.35)
"""

    def __init__(self):
        self.path = tempfile.mktemp(prefix="hh_power_", suffix=".txt")
        self.proc = None

    def start(self):
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", self.CODE, self.path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            self.proc = None

    def stop(self, t0, t1):
        if self.proc is None:
            return None
        self.proc.terminate()
        try:
            self.proc.wait(timeout=5)
        except subprocess.TimeoutExpired:
            self.proc.kill()
        try:
            rows = [l.split() for l in open(self.path)]
            os.unlink(self.path)
        except OSError:
            return None
        w = [float(r[1]) for r in rows if t0 <= float(r[0]) <= t1 and r[1] != "nan"]
        mhz = [float(r[2]) for r in rows if t0 <= float(r[0]) <= t1 and r[2] != "nan"]
        if not w:
            return {"samples": 0, "rows_seen": len(rows), "note": "rocm-smi gave no power sample inside the timed region on this box (cap: 1400 W; other runs: profiles/r5_bench_500_steps.json)"}
        return {"package_power_w_mean": round(sum(w) / len(w), 1), "package_power_w_max": max(w), "samples": len(w),
                "sclk_mhz_mean_reported": round(sum(mhz) / len(mhz), 1) if mhz else None,
                "source": "rocm-smi --showpower --showclocks sampled by a child process during the timed region and the statistics steps behind it (cap: 1400 W)"}


def cpu_baseline(cfg, enc_sd, dec_sd, seed):
    """Oracle (CPU restatement, fp32) timed on this host: (a) one full training step on ONE clip of the headline workload (config
    2), with the thread count that is fastest on the 256-thread GPU host (16; more threads run slower) and with 8 threads (the dev
    container's core count); (b) BASELINE config 1 exactly -- the reference's own CPU-runnable case,
    run/train.py:103-203 with T = 4, num_queries = 4, batch = 2; (c) BASELINE config 5 -- one EgoMCQ item forward."""
    from oracle import step as OS

    def timed(c, esd, dsd, batch, threads, iters):
        torch.set_num_threads(threads)
        dsd = {k: v.clone() for k, v in dsd.items()}
        _, _, state = OS.train_step(esd, dsd, batch, c, None)             # warm-up (allocator, thread pool)
        t = time.time()
        for _ in range(iters):
            _, _, state = OS.train_step(esd, dsd, batch, c, state)
        return iters / (time.time() - t), time.time() - t

    ncpu = os.cpu_count() or 1
    cores = min(ncpu, int(os.environ.get("HH_CPU_BASELINE_THREADS", 16)))
    batch = synth.make_batch(cfg, 1, seed=seed)
    v16, dt16 = timed(cfg, enc_sd, dec_sd, batch, cores, 3)
    v8, dt8 = timed(cfg, enc_sd, dec_sd, batch, min(8, ncpu), 2)
    rec = {"value": round(v16, 4), "unit": "clips/s", "cores": cores, "kind": "port", "threads_8": round(v8, 4), "host_cpu_count": ncpu,
           "sample": "1 clip/step (T=%d, %dpx, nq=%d): oracle fp32 full step fwd+bwd+AdamW; %d threads: 1 warm-up + 3 timed steps (%.1f s); "
                     "8 threads: 1 warm-up + 2 timed steps (%.1f s); more than 16-32 threads run slower on this host" % (
                         cfg.num_frames, cfg.img_size, cfg.num_queries, cores, dt16, dt8)}
    c1 = C1
    esd1 = dict(enc_sd)
    if cfg.num_frames != c1.num_frames or cfg.img_size != c1.img_size:
        esd1 = synth.encoder_state(c1, seed=0)
    dsd1 = synth.decoder_state(c1, seed=0)
    s1, dt1 = timed(c1, esd1, dsd1, synth.make_batch(c1, 2, seed=seed), cores, 3)
    rec["c1"] = {"value": round(2 * s1, 4), "unit": "clips/s", "cores": cores, "kind": "port", "ms_per_step": round(1e3 / s1, 1),
                 "sample": "BASELINE config 1 exactly (4-frame 224p, num_queries=4, batch=2): oracle fp32 full step fwd+bwd+AdamW, %d threads, "
                           "1 warm-up + 3 timed steps (%.1f s)" % (cores, dt1)}
    # (c) BASELINE config 5: one EgoMCQ item (5 candidate clips + 1 query text), forward only -- run/test_EgoMCQ.py:56-83
    torch.set_num_threads(cores)
    item = synth.make_mcq_item(cfg, 1, seed=seed)
    t = time.time()
    OS.mcq_forward(enc_sd, dec_sd, item["video"], item["text"], cfg)
    dt5 = time.time() - t
    rec["c5"] = {"value": round(5 / dt5, 4), "unit": "clips/s", "cores": cores, "kind": "port", "s_per_item": round(dt5, 2),
                 "sample": "BASELINE config 5: ONE EgoMCQ item (5 clips of T=%d %dpx + 1 query), oracle fp32 forward + scoring, %d threads, 1 timed call "
                           "(%.1f s; thread pool warm from the legs above)" % (cfg.num_frames, cfg.img_size, cores, dt5)}
    return rec


def build(cfg, dev, world=1):
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 1))))      # do not oversubscribe the node when N ranks build at once
    enc_sd = synth.encoder_state(cfg, seed=0)
    dec_sd = synth.decoder_state(cfg, seed=0)
    backbone = LaviLa.build_backbone(cfg, enc_sd, device=dev)
    decoder = tfm_decoder.build_decoder(cfg, dec_sd, device=dev)
    return enc_sd, dec_sd, backbone, decoder


def visual_ln_fold(backbone):
    """What the tower's blocks actually packed (the module global can differ from it)."""
    m = backbone.visual.ln_fold_packed()
    return bool(LaviLa.LN_FOLD) if m is None else bool(m)


def selfcheck_summary(sc):
    if not sc:
        return None
    return {"clip0_encoder_bit_identical": sc.get("encoder_bit_identical_clips_before_last"), "matched_frames_equal": sc.get("matched_frames_equal_fraction"),
            "hs_diff_over_scale": round(sc.get("hs_max_abs_diff_over_scale", 0.0), 6)}


def count_collectives(fn):
    """Run fn() with torch.distributed's collective entry points wrapped: -> {name: calls} (this process)."""
    names = ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast", "reduce_scatter_tensor", "all_to_all_single", "reduce", "all_gather_object")
    n, orig = {}, {}
    for k in names:
        orig[k] = getattr(dist, k)

        def wrapped(*a, _k=k, **kw):
            n[_k] = n.get(_k, 0) + 1
            return orig[_k](*a, **kw)
        setattr(dist, k, wrapped)
    try:
        fn()
    finally:
        for k, f in orig.items():
            setattr(dist, k, f)
    return n


def barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


LAST_REGION = [0.0, 0.0, 0.0]     # wall-clock bounds (time.time()) of the last timed region [0], [1] and the end of the statistics steps that
                                  # follow it [2] (same workload, back to back): the power samples are cut to [0] + 15 % .. [2]
STATS_STEPS = 50     # per-step statistics are taken over at least this many steps: the K contract steps + extra ones run after the timed region


def timed_region(run, steps, warmup, world, timers, stats_steps=0):
    """W untimed + EXACTLY K timed calls of run(), bracketed by barrier + synchronize; one device event per step on the main stream.
    If K < stats_steps, further steps follow AFTER the timed region (not part of `value`), for the per-step statistics only.
    -> (wall seconds, per-step ms list from the events, kernel-timer snapshot, last output)."""
    for _ in range(warmup):
        run()
    barrier(world)
    if timers:
        prof_enable(STRIDE)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    LAST_REGION[0] = time.time()
    t0 = time.perf_counter()
    ev[0].record()
    out = None
    for i in range(steps):
        out = run()
        ev[i + 1].record()
    barrier(world)
    dt = time.perf_counter() - t0
    LAST_REGION[1] = time.time()
    region = prof_snapshot() if timers else None
    extra = max(0, stats_steps - steps)
    for i in range(extra):
        run()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append(e)
    if extra:
        barrier(world)
    LAST_REGION[2] = time.time() if extra else LAST_REGION[1]
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps + extra)]
    return dt, per, region, out


def step_stats(per, drop_first=False):
    """drop_first: in a pipelined region the first step's frozen towers already ran during the warm-up step before it (the step is
    ~half as long as a steady-state one): it is left out of the statistics (it IS part of `value`, whose K steps are the contract)."""
    import statistics
    if drop_first and len(per) > 2:
        per = per[1:]
    s = sorted(per)
    q = lambda f: s[min(len(s) - 1, int(round(f * (len(s) - 1))))]
    return {"steps": len(per), "ms_per_step_mean": round(sum(per) / len(per), 3), "ms_per_step_std": round(statistics.pstdev(per), 3) if len(per) > 1 else 0.0,
            "ms_per_step_p50": round(q(0.5), 3), "ms_per_step_p95": round(q(0.95), 3), "ms_per_step_min": round(s[0], 3),
            "ms_per_step_max": round(s[-1], 3), "timing": "device events on the main stream after every step (the pipelined encoder of step i+1 overlaps step i, so a step's own "
            "interval is what the steady state delivers); when --steps < 50 the statistics continue past the timed region up to 50 steps"
            + ("; the region's first step (towers prefetched during warm-up) is left out" if drop_first else "")}


def rate(rec, key, scale):
    n, seen, ms, work = rec[key]
    return (work / (ms * 1e-3) / scale, n, seen, ms) if n and ms > 0 else (None, n, seen, ms)


def _stream_rec(rec, key, steps, scale=1e12):
    """{achieved, launches, avg_launch_us, ms_per_step} of one (class, role) record set; None if nothing was timed.  ms_per_step = launches in
    the region x mean launch time / steps: the time that class occupies ITS stream per step."""
    ach, n, seen, ms = rate(rec, key, scale)
    if ach is None:
        return None
    avg_us = ms * 1e3 / n
    return {"achieved": round(ach, 1), "launches_in_region": seen, "launches_timed": n, "avg_launch_us": round(avg_us, 1),
            "ms_per_step": round(seen * avg_us / 1e3 / max(steps, 1), 2)}


def roofline_records(region, iso, dt_ms, pipelined, cfg, B, steps):
    """dt_ms: wall time of the whole timed region of `steps` steps.  `roofline` = the dominant kernel (persistent GEMM, MFMA-bound) ON THE
    VISION-TOWER STREAM: the launches the host issued inside SpaceTimeTransformer.forward_features (hh_prof_set_role) -- one serial stream, so
    launches x mean launch time <= the region's wall time by construction (`stream_ms_per_step` <= `ms_per_step`).  The text tower (side
    stream) and the decoder (main stream) run concurrently with it: their launches of the same kernel family are `by_stream`, the aggregate
    over all three `by_stream.all` (NOT comparable with a step's time).  `attention_roofline`: the HBM-bound kernels of the tower.
    Algorithmic work: 2*M*N*K of the full 256-row tiles per GEMM launch; 8*N*D bytes per clip and attention call."""
    roof, att = None, {}
    names = region.get("names", {})
    roles = region.get("roles", {})
    vis = roles.get("vision_tower", region)
    cfg_tag = "c4_" if (cfg.num_frames == 32 and cfg.img_size == 336) else ""
    have_profile = cfg_tag == "c4_" or (cfg.num_frames == 16 and cfg.img_size == 224)
    main = _stream_rec(vis, "gemm256", steps)
    if main is not None:
        default_build = bool(LaviLa.LN_FOLD and LaviLa.STREAM_PAIR and LaviLa.TIME_PROJ_READS_Z3)     # what the committed PMC passes profiled
        traffic, src = pmc_traffic(VISION_GEMMS, cfg_tag) if (have_profile and default_build) else (None, None)
        alg_bytes = gemm_algorithmic_bytes(cfg, B, LaviLa.LN_FOLD, LaviLa.STREAM_PAIR, LaviLa.TIME_PROJ_READS_Z3)
        step_ms = dt_ms / max(steps, 1)
        roof = {"kernel": "gemm256w4p_kernel, vision-tower launches (persistent 256x256x64 bf16 MFMA GEMM, LayerNorms folded in)",
                "bound": "mfma", "achieved": main["achieved"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(main["achieved"] / PEAK_BF16_TFLOPS, 4),
                "launches_in_region": main["launches_in_region"], "launches_timed": main["launches_timed"], "avg_launch_us": main["avg_launch_us"],
                "stream_ms_per_step": main["ms_per_step"], "ms_per_step": round(step_ms, 2), "stream_time_le_step": bool(main["ms_per_step"] <= step_ms),
                "traffic": traffic, "traffic_source": ("committed PMC passes, profiles/%s, vision-tower instantiations <true, 5|6|7|8> only (not measured in this run)" % src) if src else None,
                "algorithmic_bytes_per_launch": alg_bytes, "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                "ln_fold": bool(LaviLa.LN_FOLD), "last_dispatched": names.get("gemm256", ""),
                "region": "timed region, pipelined" if pipelined else "timed region, un-pipelined"}
        by = {}
        for r in ("text_tower", "decoder"):
            rec = _stream_rec(roles.get(r, {}), "gemm256", steps) if r in roles else None
            if rec is not None:
                by[r] = rec
        allr = _stream_rec(region, "gemm256", steps)
        if allr is not None:
            allr["frac"] = round(allr["achieved"] / PEAK_BF16_TFLOPS, 4)
            allr["note"] = "three concurrent streams: ms_per_step here may exceed the step's wall time"
            by["all"] = allr
        o = _stream_rec(region, "gemm_other", steps)
        if o is not None:
            o["what"] = "row tails, 128x128 kernel, one-tile-per-block kernel (all streams; not part of `achieved`)"
            by["other_gemm_kernels"] = o
        roof["by_stream"] = by
        rp = rocprof_committed("gemm256w4p_kernel", cfg_tag, pipelined) if have_profile else None
        if rp is not None:
            roof["rocprofv3_committed"] = rp
        if iso is not None:
            i2 = _stream_rec(iso.get("roles", {}).get("vision_tower", iso), "gemm256", 2)
            if i2 is not None:
                roof["isolated"] = {"achieved": i2["achieved"], "frac": round(i2["achieved"] / PEAK_BF16_TFLOPS, 4), "avg_launch_us": i2["avg_launch_us"],
                                    "note": "2 un-pipelined steps outside the timed region: the tower alone on the chip"}
                roof["isolated_frac"] = roof["isolated"]["frac"]          # (flat copy: the driver's record keeps scalars)
        roof["notes"] = {
            "timing": "library-side HIP events on the launch stream around every %dth launch of the class and role (hh_prof_enable / hh_prof_set_role); "
                      "agrees with rocprofv3 --kernel-trace of the same run (round 4: 511 vs 504 us)" % STRIDE,
            "work": "2*M*N*K of the full 256-row tiles of each launch; norm3 / norm1 / norm2 of every block run inside these launches (DESIGN.md 4.6), "
                    "their time counts here",
            "bytes": "A [M,K] + W [N,K] read, C [M,N] written, bf16, mean over the six GEMMs of a block at M = %d; the branch-ending GEMMs write no C: "
                     "time proj reads z3 (2 B) and writes z (2 B), space proj / fc2 read and write the bf16 pair stream (4 + 4 B) per element of [M, D] "
                     "(bench.py: gemm_algorithmic_table)" % (B * cfg.tokens)}
    for key in ("space_attn", "time_attn"):
        rec = _stream_rec(vis, key, steps, 1e9)
        if rec is None:
            continue
        att[key] = dict(rec, kernel=names.get(key, "") or key, bound="hbm", peak=PEAK_HBM_GBS, unit="GB/s", frac=round(rec["achieved"] / PEAK_HBM_GBS, 4))
        if key == "space_attn":
            # the same launches against the matrix cores: QK^T + PV flops of a call (n frame keys + the CLS key per query).  At n = 576 keys per
            # frame (config 4) the kernel sits right of the ridge (~290 flop per byte): the HBM fraction above understates it
            T_ = cfg.num_frames
            n_ = (cfg.tokens - 1) // T_
            heads_ = cfg.embed_dim // 64
            fl = 4.0 * B * T_ * heads_ * n_ * (n_ + 1) * 64
            att[key]["mfma_view"] = {"tflops_achieved": round(fl / (rec["avg_launch_us"] * 1e-6) / 1e12, 1),
                                     "frac_of_dense_peak": round(fl / (rec["avg_launch_us"] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                     "flop_per_algorithmic_byte": round(fl / (8.0 * B * cfg.tokens * cfg.embed_dim), 1)}
        if iso is not None:
            i2 = _stream_rec(iso.get("roles", {}).get("vision_tower", iso), key, 2, 1e9)
            if i2 is not None:
                att[key]["isolated"] = {"achieved": i2["achieved"], "frac": round(i2["achieved"] / PEAK_HBM_GBS, 4), "avg_launch_us": i2["avg_launch_us"]}
    tx = _stream_rec(roles.get("text_tower", {}), "add_ln", steps, 1e9) if "text_tower" in roles else None
    if tx is not None:
        att["text_tower_add_ln"] = dict(tx, bound="hbm", unit="GB/s", note="side-stream kernel of the TEXT tower (the vision tower has no stand-alone LayerNorm pass); stretched by CU starvation in the step")
    if att:
        att["note"] = ("vision-tower launches: algorithmic bytes (8*N*D per attention call and clip, N = %d tokens, D = %d) / event time of the kernel alone; "
                       "`isolated` = un-pipelined steps" % (cfg.tokens, cfg.embed_dim))
    return roof, (att or None)


def bench_train(cfg, backbone, decoder, B, steps, warmup, world, rank, dev, args, timers, want_iso=True):
    batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1000 + rank).items()}
    ts = TrainStep(cfg, backbone, decoder, enc_cus=args.enc_cus, force_comm=args.force_comm)
    pipelined = not args.no_pipeline
    run = lambda: ts.step(batch, next_batch=batch if pipelined else None)
    if timers:
        prof_enable(0)
    # (profiling runs pass --no-kernel-timers: no extra statistics steps under rocprofv3)
    dt, per, region, out = timed_region(run, steps, warmup, world, timers, stats_steps=STATS_STEPS if (want_iso and timers) else 0)
    bounds = tuple(LAST_REGION)
    if timers:
        prof_enable(0)          # (the extra statistics steps are not part of the kernel-timer region: the snapshot was taken before them)
    phases = None
    if timers and want_iso:
        phases = phase_times(ts, batch, run, pipelined)          # (every rank runs it -- the steps hold collectives at N > 1; rank 0's numbers are printed)
    iso = None
    if timers and pipelined and want_iso:
        # outside the timed region: two un-pipelined steps, so that every kernel is also timed alone on the chip (no decoder kernels
        # of the previous step beside the encoder's)
        prof_enable(STRIDE)
        ts.text_on_side_stream = False      # (the text tower normally runs beside the vision tower's first blocks)
        for _ in range(2):
            ts.step(batch)
        ts.text_on_side_stream = True
        barrier(world)
        iso = prof_snapshot()
    if timers:
        prof_enable(0)
    return ts, batch, dt, per, region, iso, out, phases, bounds


def phase_times(ts, batch, run, pipelined, reps=3):
    """Outside the timed region: the two halves of a step alone on the chip, and the decoder half inside the pipelined step -- so that
    "work hidden on another stream is still paid for in energy"  can be read off the bench line:
      towers_alone_ms    frozen vision + text towers of one batch (TrainStep.encode), nothing else running
      decoder_alone_ms   decoder forward / losses / backward / AdamW of one batch whose towers were computed before
      decoder_in_step_ms the same span on the main stream inside the pipelined steady state (the next batch's towers run beside it)"""
    def span(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    rec = {}
    with torch.no_grad():
        span(lambda: ts.encode(batch["video"], batch["text"]))
        rec["towers_alone_ms"] = round(min(span(lambda: ts.encode(batch["video"], batch["text"])) for _ in range(reps)), 2)
    alone = []
    for _ in range(reps + 1):
        ts.prefetch(batch)
        torch.cuda.synchronize()
        alone.append(span(lambda: ts.step(batch)))
    rec["decoder_alone_ms"] = round(min(alone[1:]), 2)
    # host-side launch cost: the time the host needs to ISSUE one un-pipelined step (device idle at the start, no synchronisation inside the
    # step, so the call returns when everything is enqueued) over the library calls it made (hh_call_count: every launching hh_* entry point)
    from helping_hand_for_egocentric_videos_amd import _lib
    issue = []
    for _ in range(reps):
        torch.cuda.synchronize()
        c0, t0 = _lib.lib().hh_call_count(), time.perf_counter()
        ts.step(batch)
        t1, c1 = time.perf_counter(), _lib.lib().hh_call_count()
        torch.cuda.synchronize()
        issue.append(((t1 - t0) * 1e3, c1 - c0))
    ms, calls = min(issue)
    rec["host_issue_ms_per_step"] = round(ms, 2)
    rec["libhh_calls_per_step"] = int(calls)
    rec["host_us_per_libhh_call"] = round(ms * 1e3 / max(calls, 1), 2)
    if pipelined:
        ts.span_log = []
        for _ in range(reps + 2):
            run()
        torch.cuda.synchronize()
        spans = [a.elapsed_time(b) for a, b in ts.span_log if b is not None][2:]
        ts.span_log = None
        if spans:
            rec["decoder_in_step_ms"] = round(sum(spans) / len(spans), 2)
    return rec


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` typed plainly (no WORLD_SIZE in the environment): this process becomes the launcher.  BEFORE any GPU
    call it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>` as a FRESH CHILD process group
    (never os.exec*: a process that has touched the GPU must not be replaced, and this one has not touched it), relays the children's
    stdout -- rank 0's one JSON line -- and returns the launcher's exit code; the whole group is killed when HH_BENCH_LAUNCH_TIMEOUT
    seconds (default 3600) pass.  Fewer than N visible devices: one line on stderr, exit code 2.  (run/train.py:372-412,579-586 of the
    reference is launched by torchrun the same way.)"""
    import signal
    import socket
    assert not torch.cuda.is_initialized(), "self_launch must run before anything touches the GPU"
    same_device = os.environ.get("HH_BENCH_SAME_DEVICE") == "1"          # test hook: every rank on cuda:0 (see main())
    ndev = torch.cuda.device_count()                                     # (counting devices does not initialise the GPU)
    need = 1 if same_device else n_gpus
    if ndev < need:
        sys.stderr.write("bench.py: --gpus %d needs %d visible GPU(s), this host shows %d\n" % (n_gpus, need, ndev))
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    limit = float(os.environ.get("HH_BENCH_LAUNCH_TIMEOUT", 3600))
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True)      # stdout / stderr inherited: the line goes straight through
    try:
        return child.wait(timeout=limit)
    except (subprocess.TimeoutExpired, KeyboardInterrupt) as e:
        try:
            os.killpg(child.pid, signal.SIGKILL)                         # the launcher's own process group (start_new_session): exact, not a pattern
        except ProcessLookupError:
            pass
        child.wait()
        sys.stderr.write("bench.py: the %d-rank run was killed (%s)\n" % (n_gpus, "timeout after %.0f s" % limit if isinstance(e, subprocess.TimeoutExpired) else "interrupted"))
        return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="BASELINE.json configuration of the main workload (the metric is quoted on c2)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 32; c4: 4; c1: 2)")
    ap.add_argument("--workload", default="train", choices=["train", "mcq"])
    ap.add_argument("--mcq-items", type=int, default=8, help="EgoMCQ items (5 clips + 1 query each) per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-mcq", action="store_true", help="skip the EgoMCQ forward sub-record (second half of BASELINE.json's metric)")
    ap.add_argument("--no-c4", action="store_true", help="skip the config-4 sub-record (32-frame 336p)")
    ap.add_argument("--no-variants", action="store_true", help="skip the opt-in variant sub-record (caption-length hint: text tower on the captions' own length)")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the first-two-clips self check after the timed region (profiling runs: its B = 2 forward would dilute per-launch averages)")
    ap.add_argument("--no-power", action="store_true", help="do not sample package power with a rocm-smi child process")
    ap.add_argument("--enc-cus", type=int, default=None, help="CU budget of the persistent GEMMs on the pipelined encoder stream (multiple of 8; 0 = all)")
    ap.add_argument("--force-comm", action="store_true", help="1 GPU only: run the RCCL collectives of the data-parallel path in a 1-rank group (A/B of the CU reservation)")
    ap.add_argument("--kv-proj", action="store_true", help="A/B: the decoder's round-1..4 cross-attention (one K/V in-projection of all memory tokens for the six layers + hh_xattn_*) "
                                                           "instead of the memory-space attention (csrc/mattn.hip)")
    ap.add_argument("--token-major-qkv", action="store_true", help="A/B: the QKV projections write nn.Linear's token-major [B*N, 3D] instead of head-major planes")
    ap.add_argument("--stream-priorities", default=None, metavar="ENC,TEXT", help="A/B: HIP stream priorities of the vision-tower and text-tower streams (0 default, -1 high)")
    ap.add_argument("--stream-fp32", action="store_true", help="A/B: the tower's residual stream in fp32 with a separate bf16 LayerNorm input (rounds 4-5) instead of the bf16 pair hi + lo (LaviLa.STREAM_PAIR)")
    ap.add_argument("--walk-forward", action="store_true", help="A/B: every kernel of a tower block walks its rows first to last (rounds 1-4) instead of opposite to its predecessor (LaviLa.WALK_ALTERNATE)")
    ap.add_argument("--time-proj-fp32", action="store_true", help="A/B: the time projection's epilogue re-reads the fp32 residual rows (round 4) instead of z3 = bf16(x)")
    ap.add_argument("--no-ln-fold", action="store_true", help="A/B: norm1 / norm2 as stand-alone fused add+LayerNorm kernels instead of folded into the GEMMs around them")
    ap.add_argument("--space-16q", action="store_true", help="A/B: space attention on the 16-query-block kernel instead of the joint-block kernel")
    ap.add_argument("--gemm-tail", type=int, default=None, help="A/B: hh_set_tuning('gemm_tail', v): 1 (default) = row tails of <= 64 rows inside the persistent kernel, 2 = always the separate tail kernel, 0 = the 128x128 kernel; same results")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=V", help="A/B: hh_set_tuning(NAME, V) before anything runs (repeatable)")
    ap.add_argument("--no-text-pad", action="store_true", help="A/B: the text tower's rows as they are (S * 77: a 160-row tail per GEMM goes to a separate launch) instead of padded to 256")
    ap.add_argument("--no-qgroup", action="store_true", help="A/B: every weight gradient of the decoder's query side as its own launch (rounds 2-5) instead of one grouped launch per layer")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the next step's frozen-encoder forward with this step's decoder")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # test hooks (tests/test_dp_gpu.py runs the N > 1 code path of this file on a ONE-GPU box): HH_BENCH_BACKEND=gloo with
    # HH_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 over gloo (RCCL refuses two ranks on one device); the defaults are what the
    # driver gets: one rank per GPU, backend "nccl" = RCCL over xGMI
    backend = os.environ.get("HH_BENCH_BACKEND", "nccl")
    same_device = os.environ.get("HH_BENCH_SAME_DEVICE") == "1"
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or plainly without WORLD_SIZE set)" % (args.gpus, world, args.gpus))
        if same_device:
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        if args.force_comm:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", local if world > 1 else 0)

    if args.no_ln_fold:
        LaviLa.LN_FOLD = False
    if args.time_proj_fp32:
        LaviLa.TIME_PROJ_READS_Z3 = False
    if args.walk_forward:
        LaviLa.WALK_ALTERNATE = False
    if args.stream_fp32:
        LaviLa.STREAM_PAIR = False
    if args.stream_priorities:
        from helping_hand_for_egocentric_videos_amd import step as _step
        _step.STREAM_PRIORITIES[:] = [int(v) for v in args.stream_priorities.split(",")]
    cfg, cfg_name = CONFIGS[args.config]
    B = args.batch or {"c2": 32, "c4": 4, "c1": 2}[args.config]
    torch.manual_seed(0)
    enc_sd, dec_sd, backbone, decoder = build(cfg, dev, world)
    decoder.transformer.kv_free = not args.kv_proj
    if args.token_major_qkv:
        from helping_hand_for_egocentric_videos_amd.model import LaviLa as _L
        _L.QKV_HEAD_MAJOR_PLANES = False
    if args.no_text_pad:
        from helping_hand_for_egocentric_videos_amd.model import openai_model as _om
        _om.ROW_PAD = 0
    if args.no_qgroup:
        from helping_hand_for_egocentric_videos_amd.model import qside as _qside
        _qside.GROUP_LAUNCHES = False
    if args.space_16q:
        ops.set_tuning("space_joint", 0)
    if args.gemm_tail is not None:
        ops.set_tuning("gemm_tail", args.gemm_tail)
    for kv in args.tune:
        k, v = kv.split("=")
        ops.set_tuning(k, int(v))
    timers = not args.no_kernel_timers

    rccl = None
    if world > 1:
        # the collectives really span `world` devices: a sum of ones over the group
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl = {"rccl_ranks": int(one.item()), "backend": dist.get_backend(), "devices_visible": torch.cuda.device_count(),
                "ranks_share_one_device": same_device}
        assert rccl["rccl_ranks"] == world

    power = PowerSampler() if (rank == 0 and not args.no_power) else None
    ts = None
    if args.workload == "train":
        clips_per_step = B
        metric = "train clips/sec (%d-frame %dp, nq=%d)" % (cfg.num_frames, cfg.img_size, cfg.num_queries)
        if power:
            power.start()
        ts, batch, dt, per, region, iso, out, phases, (w0, w1, w2) = bench_train(cfg, backbone, decoder, B, args.steps, args.warmup, world, rank, dev, args, timers)
    else:
        items = args.mcq_items
        mcq = synth.make_mcq_item(cfg, items, seed=1000 + rank)
        video, text = mcq["video"].to(dev), mcq["text"].to(dev)
        decoder.eval()
        scorer = McqScorer(backbone, decoder, cfg)
        run = (lambda: mcq_forward(backbone, decoder, video, text, cfg)) if args.no_pipeline else (lambda: scorer(video, text, next_item=(video, text)))
        clips_per_step = items * 5
        metric = "EgoMCQ fwd clips/sec (%d-frame %dp)" % (cfg.num_frames, cfg.img_size)
        if power:
            power.start()
        dt, per, region, out = timed_region(run, args.steps, args.warmup, world, timers)
        w0, w1, w2 = LAST_REGION
        iso = phases = None
        if timers:
            prof_enable(0)
    # the timed region (its first 15 % left out: the chip settles into its steady power state) and the statistics steps right behind it --
    # the same workload back to back; a 20-step region alone (2 s) holds 0-5 rocm-smi samples
    power_rec = power.stop(w0 + 0.15 * (w1 - w0), max(w1, w2)) if power else None
    dts = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank = None
    if world > 1:
        allt = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, dts)
        per_rank = [clips_per_step * args.steps / float(t) for t in allt.tolist()]
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
    dt = float(dts)
    value = clips_per_step * world * args.steps / dt

    # exposed communication at N > 1: the same region again with the gradient collectives switched off (ranks then diverge, which is
    # fine after the measurement); allreduce_exposed_ms = ms/step with - ms/step without
    comm_rec = None
    if world > 1 and args.workload == "train":
        ts.comm.active = False
        k2 = max(5, min(args.steps, 20))
        dt2, _, _, _ = timed_region(lambda: ts.step(batch, next_batch=None if args.no_pipeline else batch), k2, 2, world, False)
        ts.comm.active = True
        t2 = torch.tensor([dt2], device=dev, dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        # COUNTED, per rank: the collectives one more pipelined step issues (torch.distributed's entry points wrapped for its duration);
        # must be 1 packed all-gather + one all-reduce per gradient bucket + 1 flag all-reduce on every rank
        counted = count_collectives(lambda: ts.step(batch, next_batch=None if args.no_pipeline else batch))
        barrier(world)
        want = {"all_gather_into_tensor": 1, "all_reduce": len(ts.arena.buckets) + 1}
        mine = torch.tensor([counted.get("all_gather_into_tensor", 0), counted.get("all_reduce", 0), sum(counted.values())], device=dev, dtype=torch.int64)
        allc = torch.empty(world * 3, device=dev, dtype=torch.int64)
        dist.all_gather_into_tensor(allc, mine)
        per_rank_counts = allc.view(world, 3).tolist()
        assert all(r == [want["all_gather_into_tensor"], want["all_reduce"], want["all_gather_into_tensor"] + want["all_reduce"]] for r in per_rank_counts), \
            ("collectives per step differ from 1 + #buckets + 1", per_rank_counts, want)
        comm_rec = {"ms_per_step_without_gradient_allreduce": round(float(t2) / k2 * 1e3, 2), "steps": k2,
                    "allreduce_exposed_ms": round(dt / args.steps * 1e3 - float(t2) / k2 * 1e3, 2),
                    "collectives_per_step": {"all_gather": 1, "all_reduce_gradient_buckets": len(ts.arena.buckets), "all_reduce_flags": 1},
                    "collectives_counted_per_rank": {"[all_gather_into_tensor, all_reduce, total] per rank": per_rank_counts, "asserted": "1 + #buckets + 1 on every rank"},
                    "gradient_bytes_per_step": int(ts.arena.total * 4)}

    # the line's own parity bit: clips 0-1 of the benchmarked batch against the same two clips run as a batch of 2 (after the timed region)
    selfcheck = None
    if world == 1 and args.workload == "train" and not args.no_selfcheck and clips_per_step >= 2:
        selfcheck = first_clips_check(ts, batch, k=2)
        selfcheck["what"] = ("eval-mode forward (towers, decoder, hand-box matching) of the bench batch vs its first 2 clips alone; clip 0 must be bit-identical through "
                             "the encoder; the small batch's last clip holds its GEMM row tail (different K summation order: bf16 roundings); hs / boxes differ by the "
                             "cross-attention's key-slice count (fp32 re-association)")

    # opt-in variant, NOT part of `value`: the same step with the data pipeline's caption-length hint (batch["text_max_len"], a host int):
    # the text tower then runs on ceil16(longest caption) positions instead of the 77 the reference computes (step.py: TrainStep.encode)
    variants = None
    if world == 1 and args.workload == "train" and args.config == "c2" and not args.no_variants:
        longest = int((batch["text"] != 0).sum(1).max())          # (outside any timed region; a loader knows it from the tokenizer)
        hinted = dict(batch, text_max_len=longest)
        kv = max(5, min(args.steps, 20))
        dtv, _, _, outv = timed_region(lambda: ts.step(hinted, next_batch=None if args.no_pipeline else hinted), kv, 3, world, False)
        variants = {"caption_length_hint": {"value": round(clips_per_step * kv / dtv, 2), "unit": "clips/s", "ms_per_step": round(dtv / kv * 1e3, 2), "steps": kv,
                                            "text_positions": min(cfg.context_length, (longest + 15) // 16 * 16), "of": cfg.context_length,
                                            "loss": round(float(outv["total_loss"]), 4),
                                            "note": "opt-in: batch['text_max_len'] from the data pipeline; the causal text tower skips the positions behind the longest "
                                                    "caption's EOT (dead work: only EOT rows are read).  The headline `value` computes all 77 positions like the reference"}}

    # second half of BASELINE.json's metric in the same line: EgoMCQ forward clips/s (config 5: q = 8 items = 40 clips + 8 queries)
    mcq_rec = None
    if args.workload == "train" and not args.no_mcq and args.config == "c2":
        q = args.mcq_items
        item = synth.make_mcq_item(cfg, q, seed=2000 + rank)
        mv, mt = item["video"].to(dev), item["text"].to(dev)
        decoder.eval()
        ksteps = 20
        scorer = McqScorer(backbone, decoder, cfg)                 # the next item batch's frozen towers overlap this one's decoder forward
        mrun = (lambda: mcq_forward(backbone, decoder, mv, mt, cfg)) if args.no_pipeline else (lambda: scorer(mv, mt, next_item=(mv, mt)))
        mdt, mper, _, scores = timed_region(mrun, ksteps, 2, world, False)
        mt_ = torch.tensor([mdt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(mt_, op=dist.ReduceOp.MAX)
        mcq_rec = {"metric": "EgoMCQ fwd clips/sec (16-frame 224p)", "value": round(q * 5 * world * ksteps / float(mt_), 2), "unit": "clips/s",
                   "ms_per_step": round(float(mt_) / ksteps * 1e3, 2), "steps": ksteps, "warmup": 2, "step_stats": step_stats(mper, drop_first=not args.no_pipeline),
                   "config": {"workload": "C5: 16-frame EgoMCQ forward, q = %d items (%d clips + %d queries) per step per GPU, replicas only" % (q, 5 * q, q),
                              "pipelined_encoder": not args.no_pipeline, "resident_batch_reused": True,
                              "step_tflop_per_clip": round(step_tflop_per_clip(cfg, train=False), 2)}}
        del mv, mt, scores

    clock = None
    if rank == 0 and timers and args.workload == "train" and cfg.embed_dim == 1024:
        clock = in_step_clock(backbone, batch["video"])

    # config 4 (32-frame 336p: the HBM-bound space-time-attention stress) as a sub-record with its own kernel records
    c4_rec = None
    if world == 1 and args.workload == "train" and args.config == "c2" and not args.no_c4:
        del ts, batch
        torch.cuda.empty_cache()
        _, _, bb4, dec4 = build(C4, dev)
        B4, k4 = 4, 10
        ts4, batch4, dt4, per4, region4, iso4, out4, phases4, _ = bench_train(C4, bb4, dec4, B4, k4, 3, 1, rank, dev, args, timers)
        roof4, att4 = roofline_records(region4, iso4, dt4 * 1e3, not args.no_pipeline, C4, B4, k4) if region4 is not None else (None, None)
        c4_rec = {"metric": "train clips/sec (32-frame 336p, nq=12)", "value": round(B4 * k4 / dt4, 2), "unit": "clips/s", "ms_per_step": round(dt4 / k4 * 1e3, 2),
                  "steps": k4, "warmup": 3, "step_stats": step_stats(per4, drop_first=not args.no_pipeline),
                  "config": {"workload": "C4: 32-frame 336p (N = 18 433 tokens / clip), nq=12, frozen TimeSformer-L + object-query decoder train step",
                             "clips_per_gpu": B4, "step_tflop_per_clip": round(step_tflop_per_clip(C4), 2)},
                  "roofline": roof4, "attention_roofline": att4, "loss": round(float(out4["total_loss"]), 4)}
        del ts4, batch4, bb4, dec4
        torch.cuda.empty_cache()

    if rank == 0:
        dt_ms = dt * 1e3
        roof, att = roofline_records(region, iso, dt_ms, not args.no_pipeline, cfg, clips_per_step, args.steps) if region is not None else (None, None)
        if roof is not None and clock is not None:
            clock["frac_of_clock_limited_peak"] = round(roof["achieved"] / clock["clock_limited_peak"], 4)
            roof["sustained_clock"] = clock
        train = args.workload == "train"
        tf = step_tflop_per_clip(cfg, train)
        if roof is not None:
            # (the driver keeps `roofline` and `config` of the line: the cross-checks that must survive sit in there as plain numbers)
            roof["end_to_end_mfma_frac"] = round(value * tf / (world * PEAK_BF16_TFLOPS), 4)
            if power_rec:
                roof["power_w_mean"] = power_rec.get("package_power_w_mean")
                roof["power_cap_w"] = 1400
            if phases:
                roof.update(phases)
        line = {"metric": metric, "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "%s, frozen TimeSformer-L + object-query decoder %s" % (cfg_name, "train step" if train else "EgoMCQ forward"),
                           "clips_per_gpu": clips_per_step, "pipelined_encoder": bool(train and not args.no_pipeline),
                           "global_clips": clips_per_step * world, "parallelism": "dp%d" % world,
                           "resident_batch_reused": True, "materialize_logits": False, "ln_fold": visual_ln_fold(backbone),
                           "decoder_kv_free": bool(decoder.transformer.kv_free), "step_tflop_per_clip": round(tf, 2),
                           "selfcheck": selfcheck_summary(selfcheck)},
                "step_stats": step_stats(per, drop_first=bool(train and not args.no_pipeline)),
                "end_to_end_mfma_frac": round(value * tf / (world * PEAK_BF16_TFLOPS), 4),
                "roofline": roof}
        # flat scalars the driver's stored record keeps (it cuts nested sub-records): second half of BASELINE.json's metric, config 4, power, margins
        cfgd = line["config"]
        cfgd["end_to_end_mfma_frac"] = line["end_to_end_mfma_frac"]
        if mcq_rec is not None:
            cfgd["mcq_clips_per_s"] = mcq_rec["value"]
        if c4_rec is not None:
            cfgd["c4_clips_per_s"] = c4_rec["value"]
            if c4_rec.get("roofline"):
                cfgd["c4_roofline_frac"] = c4_rec["roofline"]["frac"]
        if power_rec and power_rec.get("package_power_w_mean") is not None:
            cfgd["power_w_mean"] = power_rec["package_power_w_mean"]
        if phases:
            for k in ("host_issue_ms_per_step", "libhh_calls_per_step", "host_us_per_libhh_call", "decoder_in_step_ms", "decoder_alone_ms", "towers_alone_ms"):
                if k in phases:
                    cfgd[k] = phases[k]
            if "decoder_in_step_ms" in phases:
                cfgd["decoder_in_step_frac"] = round(phases["decoder_in_step_ms"] / (dt / args.steps * 1e3), 3)
        if att:
            for k, short in (("space_attn", "space_attn_hbm_frac"), ("time_attn", "time_attn_hbm_frac")):
                if k in att:
                    cfgd[short] = att[k]["frac"]
        if att:
            line["attention_roofline"] = att
        if power_rec:
            line["power"] = power_rec
        if rccl:
            line["rccl"] = rccl
            line["per_rank_clips_per_s"] = {"min": round(min(per_rank), 2), "max": round(max(per_rank), 2)}
        if comm_rec:
            line["comm"] = comm_rec
        if mcq_rec is not None:
            line["mcq"] = mcq_rec
        if c4_rec is not None:
            line["c4"] = c4_rec
        if variants is not None:
            line["variants"] = variants
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, enc_sd, dec_sd, seed=1000)
        if train:
            line["loss"] = round(float(out["total_loss"]), 4)
        if selfcheck is not None:
            line["selfcheck"] = selfcheck
        print(json.dumps(line))
    if world > 1 or args.force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
