"""Benchmark of the hot path: train clips/sec (16-frame 224p, nq=12) -- BASELINE.json's metric -- on N GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = frozen TimeSformer-L forward + text tower + object-query decoder forward/backward + EgoNCE / box /
word losses (on-device Hungarian) + decoder-gradient all-reduce (RCCL) + fused AdamW over one synthetic batch
that is resident in HBM.  Weak scaling: the per-GPU batch is fixed.  Prints ONE JSON line on rank 0.
The same line carries an "mcq" sub-record: the EgoMCQ forward-only path (BASELINE config 5, q = 8 items) timed after the train
region; --workload mcq times only that path.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from helping_hand_for_egocentric_videos_amd import C2, ops, synth  # noqa: E402
from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder  # noqa: E402
from helping_hand_for_egocentric_videos_amd.step import TrainStep, mcq_forward  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


PROF = {"gemm256": 0, "gemm_other": 1, "space_attn": 2, "time_attn": 3, "add_ln": 4, "gemm_tn": 5, "xattn_fwd": 6, "xattn_bwd": 7}


def prof_enable(stride):
    """Library-side per-kernel timing (include/hh.h: hh_prof_enable): every stride-th launch of each instrumented kernel class is
    bracketed by two HIP events on its launch stream with only that kernel between them.  stride 0 = off."""
    from helping_hand_for_egocentric_videos_amd import _lib
    _lib.check(_lib.lib().hh_prof_enable(int(stride)), "hh_prof_enable")


def prof_read(name):
    """-> (launches timed, launches seen, total ms, total algorithmic work) of one kernel class; synchronises on its events."""
    import ctypes
    from helping_hand_for_egocentric_videos_amd import _lib
    n, seen, ms, work = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib().hh_prof_read(PROF[name], ctypes.byref(n), ctypes.byref(seen), ctypes.byref(ms), ctypes.byref(work)), "hh_prof_read")
    return n.value, seen.value, ms.value, work.value


def prof_snapshot():
    return {k: prof_read(k) for k in PROF}


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/r2_pmc_summary.json, else
    round 1's: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction);
    launch-weighted mean over the template instantiations whose name contains `kernel_substr`."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r2_pmc_summary.json", "r1_pmc_summary.json")) if os.path.exists(q)), None)
    if path is None:
        return None
    with open(path) as f:
        rows = [v for k, v in json.load(f).items() if kernel_substr in k]
    n = sum(v["launches"] for v in rows)
    return int(sum(v["traffic_bytes_per_launch"] * v["launches"] for v in rows) / n) if n else None


def sustained_clock(dev):
    """Shader clock inside the persistent GEMM's main loop under sustained load (s_memtime ticks per s_memrealtime us, read by
    the kernel itself: hh_debug_gemm_timeline).  MI355X throttles far below its 2.4 GHz nominal clock when the matrix cores are
    busy, so the 2.5 PFLOP/s datasheet peak is not reachable by ANY bf16 GEMM here; the clock-limited peak is 2.5 * f / 2.4."""
    import ctypes
    import numpy as np
    from helping_hand_for_egocentric_videos_amd import _lib
    M, N, K = 32 * 4096, 4096, 1024
    g = torch.Generator(device=dev).manual_seed(0)
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    for _ in range(40):                                  # ~35 ms of back-to-back GEMMs: the governor has settled
        ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU)
    ops.set_tuning("gemm256_debug_ts", 1)
    ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU)
    torch.cuda.synchronize()
    ops.set_tuning("gemm256_debug_ts", 0)
    buf = np.zeros((256, 8, 7), dtype=np.uint64)
    _lib.check(_lib.lib().hh_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p), 256), "hh_debug_gemm_timeline")
    t = buf.astype(np.int64)
    us = (t[:, :, 2] - t[:, :, 1]) / 100.0               # main loop, 100 MHz real-time counter
    mhz = float(((t[:, :, 6] - t[:, :, 5]) / np.maximum(us, 1e-9)).mean())
    return {"shader_mhz_in_gemm_main_loop": round(mhz, 1), "nominal_mhz": 2400,
            "main_loop_us_per_ktile": round(float(us.mean()) / (K // 64), 3),
            "clock_limited_peak": round(PEAK_BF16_TFLOPS * mhz / 2400.0, 1)}


def cpu_baseline(cfg, enc_sd, dec_sd, seed):
    """Oracle (CPU restatement, fp32) timed on this host: one full training step on ONE clip of the same workload, with the
    thread count that is fastest on the 256-thread GPU host (16; more threads run slower) and with 8 threads (the dev container's
    core count)."""
    from oracle import step as OS
    batch = synth.make_batch(cfg, 1, seed=seed)

    def timed(threads, iters):
        torch.set_num_threads(threads)
        dsd = {k: v.clone() for k, v in dec_sd.items()}
        _, _, state = OS.train_step(enc_sd, dsd, batch, cfg, None)       # warm-up (allocator, thread pool)
        t = time.time()
        for _ in range(iters):
            _, _, state = OS.train_step(enc_sd, dsd, batch, cfg, state)
        return iters / (time.time() - t), time.time() - t

    ncpu = os.cpu_count() or 1
    cores = min(ncpu, int(os.environ.get("HH_CPU_BASELINE_THREADS", 16)))
    v16, dt16 = timed(cores, 3)
    v8, dt8 = timed(min(8, ncpu), 2)
    return {"value": round(v16, 4), "unit": "clips/s", "cores": cores, "kind": "port", "threads_8": round(v8, 4), "host_cpu_count": ncpu,
            "sample": "1 clip/step (T=%d, %dpx, nq=%d): oracle fp32 full step fwd+bwd+AdamW; %d threads: 1 warm-up + 3 timed steps (%.1f s); "
                      "8 threads: 1 warm-up + 2 timed steps (%.1f s); more than 16-32 threads run slower on this host" % (
                          cfg.num_frames, cfg.img_size, cfg.num_queries, cores, dt16, dt8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--workload", default="train", choices=["train", "mcq"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-mcq", action="store_true", help="skip the EgoMCQ forward sub-record (second half of BASELINE.json's metric)")
    ap.add_argument("--enc-cus", type=int, default=None, help="CU budget of the persistent GEMMs on the pipelined encoder stream (multiple of 8; 0 = all)")
    ap.add_argument("--force-comm", action="store_true", help="1 GPU only: run the RCCL collectives of the data-parallel path in a 1-rank group (A/B of the CU reservation)")
    ap.add_argument("--per-op-query-side", action="store_true", help="A/B: per-op autograd query side instead of the fused QueryStack node")
    ap.add_argument("--token-major-qkv", action="store_true", help="A/B: the QKV projections write nn.Linear's token-major [B*N, 3D] instead of head-major planes")
    ap.add_argument("--space-16q", action="store_true", help="A/B: space attention on the 16-query-block kernel instead of the joint-block kernel")
    ap.add_argument("--gemm-tail", type=int, default=None, help="A/B: hh_set_tuning('gemm_tail', v): 1 (default) = row tails of <= 64 rows inside the persistent kernel, 2 = always the separate tail kernel, 0 = the 128x128 kernel; same results")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the next step's frozen-encoder forward with this step's decoder")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        if args.force_comm:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", local if world > 1 else 0)

    cfg = C2
    torch.manual_seed(0)
    # host threads for the synthetic-weight generation: do not oversubscribe the node when N ranks build models at once
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 1))))
    enc_sd = synth.encoder_state(cfg, seed=0)
    dec_sd = synth.decoder_state(cfg, seed=0)
    backbone = LaviLa.build_backbone(cfg, enc_sd, device=dev)
    decoder = tfm_decoder.build_decoder(cfg, dec_sd, device=dev)
    decoder.transformer.use_query_stack = not args.per_op_query_side
    if args.token_major_qkv:
        from helping_hand_for_egocentric_videos_amd.model import LaviLa as _L
        _L.QKV_HEAD_MAJOR_PLANES = False
    if args.space_16q:
        ops.set_tuning("space_joint", 0)
    if args.gemm_tail is not None:
        ops.set_tuning("gemm_tail", args.gemm_tail)
    B = args.batch

    STRIDE = 5      # every 5th launch of a class is timed (co-prime with the 6 GEMMs / 2 attention calls per block); bracketing every launch costs ~3 % in marker packets
    timers = not args.no_kernel_timers

    if args.workload == "train":
        batch = {k: v.to(dev) for k, v in synth.make_batch(cfg, B, seed=1000 + rank).items()}
        ts = TrainStep(cfg, backbone, decoder, enc_cus=args.enc_cus, force_comm=args.force_comm)
        run = lambda: ts.step(batch, next_batch=None if args.no_pipeline else batch)
        clips_per_step = B
        metric = "train clips/sec (16-frame 224p, nq=12)"
    else:
        items = max(1, B // 5)
        mcq = synth.make_mcq_item(cfg, items, seed=1000 + rank)
        video, text = mcq["video"].to(dev), mcq["text"].to(dev)
        decoder.eval()
        run = lambda: mcq_forward(backbone, decoder, video, text, cfg)
        clips_per_step = items * 5
        metric = "EgoMCQ fwd clips/sec (16-frame 224p)"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    if timers:
        prof_enable(STRIDE)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    barrier()
    dt = time.perf_counter() - t0
    region = prof_snapshot() if timers else None
    iso = None
    if timers and args.workload == "train" and not args.no_pipeline:
        # outside the timed region: two un-pipelined steps, so that every kernel is also timed alone on the chip (no decoder kernels
        # of the previous step beside the encoder's)
        prof_enable(STRIDE)
        for _ in range(2):
            ts.step(batch)
        barrier()
        iso = prof_snapshot()
    if timers:
        prof_enable(0)
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    value = clips_per_step * world * args.steps / dt

    # second half of BASELINE.json's metric in the same line: EgoMCQ forward clips/s (config 5: q = 8 items = 40 clips + 8 queries)
    mcq_rec = None
    if args.workload == "train" and not args.no_mcq:
        q = 8
        item = synth.make_mcq_item(cfg, q, seed=2000 + rank)
        mv, mt = item["video"].to(dev), item["text"].to(dev)
        decoder.eval()
        for _ in range(2):
            mcq_forward(backbone, decoder, mv, mt, cfg)
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            scores = mcq_forward(backbone, decoder, mv, mt, cfg)
        barrier()
        mt_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(mt_, op=dist.ReduceOp.MAX)
        mcq_rec = {"metric": "EgoMCQ fwd clips/sec (16-frame 224p)", "value": round(q * 5 * world * 5 / float(mt_), 2), "unit": "clips/s",
                   "ms_per_step": round(float(mt_) / 5 * 1e3, 2), "steps": 5, "warmup": 2,
                   "config": {"workload": "C5: 16-frame EgoMCQ forward, q = %d items (%d clips + %d queries) per step per GPU, replicas only" % (q, 5 * q, q),
                              "step_tflop_per_clip": 3.45}}
        del mv, mt, scores

    if rank == 0:
        def rate(rec, key, scale):
            n, seen, ms, work = rec[key]
            return (work / (ms * 1e-3) / scale, n, seen, ms) if n and ms > 0 else (None, n, seen, ms)

        roof = None
        if region is not None:
            ach, n, seen, ms = rate(region, "gemm256", 1e12)
            if ach is not None:
                roof = {"kernel": "gemm256d_kernel (persistent 256x256x64 bf16 MFMA GEMM, continuous k-tile stream, four barriers per k-tile) -- every template instantiation, nothing else",
                        "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                        "traffic": pmc_traffic("gemm256d_kernel<true"),
                        "traffic_note": "HBM-side bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/*_pmc_summary.json, B=32); algorithmic bytes per launch average 0.99e9",
                        "algorithmic_work": "2*M*N*K of the full 256-row tiles of each launch (DESIGN.md section 5)",
                        "launches_timed": n, "launches_in_region": seen,
                        "timing": "library-side HIP events around the kernel launch alone (hh_prof_enable), every %dth launch, on the launch stream" % STRIDE,
                        "avg_launch_us": round(ms * 1e3 / n, 1), "share_of_step": round(ms * STRIDE / (dt * 1e3), 3),
                        "region": "timed region (pipelined: decoder kernels of the previous step run beside it)" if not args.no_pipeline else "timed region (un-pipelined)"}
                if iso is not None:
                    a2, n2, _, ms2 = rate(iso, "gemm256", 1e12)
                    if a2 is not None:
                        roof["isolated"] = {"achieved": round(a2, 1), "frac": round(a2 / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(ms2 * 1e3 / n2, 1),
                                            "note": "same kernel over 2 un-pipelined steps outside the timed region (alone on the chip); compare with the un-pipelined rocprofv3 summary in profiles/"}
                o, no, _, mso = rate(region, "gemm_other", 1e12)
                if o is not None:
                    roof["other_gemm_kernels"] = {"what": "row tails (gemm_tail_kernel), 128x128 kernel, one-tile-per-block 256x256 kernel -- NOT part of `achieved`",
                                                  "achieved": round(o, 1), "unit": "TFLOP/s", "launches_timed": no, "avg_launch_us": round(mso * 1e3 / no, 1),
                                                  "share_of_step": round(mso * STRIDE / (dt * 1e3), 3)}
                sc = sustained_clock(dev)
                sc["frac_of_clock_limited_peak"] = round(roof["achieved"] / sc["clock_limited_peak"], 4)
                roof["sustained_clock"] = sc
        line = {"metric": metric, "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "C2: 16-frame 224p, nq=12, frozen TimeSformer-L + object-query decoder %s" % (
                    "train step" if args.workload == "train" else "EgoMCQ forward"), "clips_per_gpu": clips_per_step,
                    "pipelined_encoder": bool(args.workload == "train" and not args.no_pipeline),
                    "global_clips": clips_per_step * world, "parallelism": "dp%d" % world,
                    "step_tflop_per_clip": 3.59 if args.workload == "train" else 3.45},
                "end_to_end_mfma_frac": round(value * (3.59 if args.workload == "train" else 3.45) / (world * PEAK_BF16_TFLOPS), 4),
                "roofline": roof}
        if region is not None:
            att = {}
            for key, kname in (("space_attn", "space_attnj_kernel<4, 2> (joint-block kernel, head-major q|k|v planes)"), ("time_attn", "time_attn_mfma_kernel<16>"), ("add_ln", "add_ln_kernel")):
                r, n, _, ms = rate(region, key, 1e9)
                if r is None:
                    continue
                att[key] = {"kernel": kname, "bound": "hbm", "achieved": round(r, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(r / PEAK_HBM_GBS, 4),
                            "launches_timed": n, "avg_launch_us": round(ms * 1e3 / n, 1), "share_of_step": round(ms * STRIDE / (dt * 1e3), 3)}
                if iso is not None:
                    r2, n2, _, ms2 = rate(iso, key, 1e9)
                    if r2 is not None:
                        att[key]["isolated"] = {"achieved": round(r2, 1), "frac": round(r2 / PEAK_HBM_GBS, 4), "avg_launch_us": round(ms2 * 1e3 / n2, 1)}
            if att:
                att["note"] = "algorithmic bytes (8*N*D per attention call and clip; bytes read + written for add+LayerNorm) / event time of the kernel alone; `isolated` = un-pipelined steps"
                line["attention_roofline"] = att
        if mcq_rec is not None:
            line["mcq"] = mcq_rec
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, enc_sd, dec_sd, seed=1000)
        if args.workload == "train":
            line["loss"] = round(float(out["total_loss"]), 4)
        print(json.dumps(line))
    if world > 1 or args.force_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
