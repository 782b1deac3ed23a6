"""The real TrainStep under data parallelism, world_size 2 on ONE GPU (gloo backend, both ranks on cuda:0).

RCCL refuses two ranks on one device, so the collective transport here is gloo; everything else (arena buckets,
post-accumulate hooks, packed contrastive all-gather with the W x slice backward, num_boxes all-reduce, fused
AdamW) is the code the 8-GPU run uses.  Checks: ranks end with identical parameters, and the 2-rank step equals the
1-rank step on the concatenated batch (eval mode -> no dropout; tolerance = bf16 K/V rounding differences only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(cfg):
    from helping_hand_for_egocentric_videos_amd import synth
    from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
    from helping_hand_for_egocentric_videos_amd.step import TrainStep
    backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=4))
    dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4))
    return TrainStep(cfg, backbone, dec, lr=1e-4), dec


def _one_step(ts, dec, batch):
    """One optimisation step with dropout off (decoder.eval()) so that ranks / world sizes are comparable."""
    from helping_hand_for_egocentric_videos_amd import ops
    dec.eval()
    ts.arena.zero_grad()
    out = ts.losses(batch)
    out["total_loss"].backward()
    ts.optimizer_step()
    return out


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helping_hand_for_egocentric_videos_amd import synth, TINY4
        cfg = TINY4
        ts, dec = _build(cfg)
        assert ts.comm.enabled and len(ts.arena.buckets) >= 2
        full = synth.make_batch(cfg, 4, seed=21)
        b = 2
        local = {}
        for k, v in full.items():
            if k == "all_nouns":
                local[k] = v.cuda()
            elif v.shape[0] == 4:
                local[k] = v[rank * b:(rank + 1) * b].cuda()
            else:
                local[k] = v[rank * b * 5:(rank + 1) * b * 5].cuda()
        losses = []
        for _ in range(2):
            losses.append(float(_one_step(ts, dec, local)["total_loss"]))
        flat = ts.arena.params.detach().cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
        nce = float(ts.losses(local)["nce_loss"])            # collective: every rank must call it
        # the exact call bench.py makes at N > 1: train mode (dropout on), next batch's frozen towers prefetched on the
        # encoder stream while backward + bucketed all-reduce run -> ranks must still hold identical parameters
        for _ in range(3):
            out = ts.step(local, next_batch=local)
        assert torch.isfinite(out["total_loss"]).item()
        flat2 = ts.arena.params.detach().cpu()
        gathered2 = [torch.empty_like(flat2) for _ in range(world)]
        dist.all_gather(gathered2, flat2)
        assert all(torch.equal(gathered2[0], g) for g in gathered2), "ranks diverged in pipelined train mode"
        assert not torch.equal(flat2, flat)
        # the terms computed on the GATHERED batch (EgoNCE and the retrieval accuracies) are the same numbers on every rank -- each rank
        # contributed its own clips (with its own dropout masks) to the one packed all-gather; the box / word terms are rank-local shares
        glob = torch.stack([out["nce_loss"].float(), out["acc_vt"].float(), out["acc_tv"].float()]).cpu()
        allg = [torch.empty_like(glob) for _ in range(world)]
        dist.all_gather(allg, glob)
        assert all(torch.equal(allg[0], g) for g in allg), ("global loss terms differ between ranks", allg)
        for k in ("total_loss", "box_loss_hand", "box_loss_obj", "word_loss", "nce_loss"):
            assert torch.isfinite(out[k]).item(), k
        if rank == 0:
            ret["params"] = flat
            ret["nce"] = nce
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank_on_concatenated_batch():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    from helping_hand_for_egocentric_videos_amd import synth, TINY4
    ts, dec = _build(TINY4)
    full = {k: v.cuda() for k, v in synth.make_batch(TINY4, 4, seed=21).items()}
    for _ in range(2):
        _one_step(ts, dec, full)
    ref = ts.arena.params.detach().cpu()
    got = ret["params"]
    # every loss term is normalised by its all-reduced count / W (num_boxes, valid words), so W ranks are one process on the
    # concatenated batch up to bf16 K/V rounding; Adam's sign-like early steps amplify tiny gradient differences of near-zero
    # gradients -> the max is compared at lr scale, the mean tightly.
    diff = (got - ref).abs()
    print("DP 2 ranks vs 1 rank: max |dp| %.3e, mean %.3e (lr 1e-4, 2 steps)" % (float(diff.max()), float(diff.mean())))
    assert float(diff.max()) <= 2.5 * 1e-4 * 2, float(diff.max())          # <= ~2 steps x lr
    assert float(diff.mean()) < 1e-5, float(diff.mean())
    nce_ref = float(ts.losses(full)["nce_loss"])
    assert abs(ret["nce"] - nce_ref) < 5e-3 * abs(nce_ref)                  # identical global contrastive loss on every rank


def _rccl_worker(rank, world, port, ret):
    """One rank, backend "nccl" (= RCCL): the real collectives -- bucketed gradient all-reduce (ReduceOp.AVG) on the comm stream,
    packed contrastive all-gather, num_boxes all-reduce -- run beside the pipelined encoder stream's persistent GEMMs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from helping_hand_for_egocentric_videos_amd import synth, TINY16, ops
        from helping_hand_for_egocentric_videos_amd.model import LaviLa, tfm_decoder
        from helping_hand_for_egocentric_videos_amd.step import TrainStep, DP_ENC_CUS
        cfg = TINY16
        batch = {k: v.cuda() for k, v in synth.make_batch(cfg, 4, seed=21).items()}

        def build(force):
            backbone = LaviLa.build_backbone(cfg, synth.encoder_state(cfg, seed=4))
            dec = tfm_decoder.build_decoder(cfg, synth.decoder_state(cfg, seed=4))
            return TrainStep(cfg, backbone, dec, lr=1e-4, bucket_bytes=1 << 20, force_comm=force, enc_cus=248 if force else None), dec

        ts, dec = build(True)
        assert ts.comm.enabled and ts.comm.avg and ts.comm.comm_stream is not None and len(ts.arena.buckets) >= 4
        assert ts.enc_cus == 248                      # explicit reservation: exercised here, not the default (DP_ENC_CUS)
        ref, rdec = build(False)
        assert not ref.comm.enabled
        # (a) one eval-mode backward: gradients through the RCCL path == gradients without it (AVG over one rank is the identity)
        for t, d in ((ts, dec), (ref, rdec)):
            d.eval()
            t.arena.zero_grad()
            t.losses(batch)["total_loss"].backward()
            t.comm.finish()
        assert ts.comm.launched == len(ts.arena.buckets) and ts.comm.flag_reduces == 1
        torch.testing.assert_close(ts.arena.grads, ref.arena.grads, rtol=1e-4, atol=1e-7)
        assert torch.equal(ts.arena.seg_flag, ref.arena.seg_flag) and float(ts.arena.seg_flag.min()) == 1.0
        # (b) the call bench.py makes at N > 1: pipelined train-mode steps, collectives overlapping the next batch's encoder
        for _ in range(3):
            out = ts.step(batch, next_batch=batch)
        torch.cuda.synchronize()
        assert ops.stream_cu_budget(ts.enc_stream) == 248
        assert ts.comm.launched == 4 * len(ts.arena.buckets) and ts.comm.flag_reduces == 4
        ret["loss"] = float(out["total_loss"])
        ret["finite"] = bool(torch.isfinite(ts.arena.params).all())
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_run_on_the_comm_stream_beside_the_pipelined_encoder():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret["finite"] and ret["loss"] == ret["loss"]


def test_bench_multi_rank_code_path_on_one_gpu():
    """bench.py's N > 1 branch (what the driver runs on an 8-GPU node: rank discovery from the environment, a real collective that
    proves the group size, barrier + max-over-ranks timing, per-rank rates, the second region without gradient collectives for
    `allreduce_exposed_ms`, the EgoMCQ sub-record) executed end to end with two ranks sharing cuda:0 over gloo -- RCCL refuses two
    ranks on one device, so the transport differs from production, the code path does not."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HH_BENCH_BACKEND="gloo", HH_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c1",
           "--batch", "2", "--no-power"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["config"]["global_clips"] == 4
    assert line["rccl"]["rccl_ranks"] == 2 and line["rccl"]["ranks_share_one_device"]
    assert line["value"] > 0 and line["per_rank_clips_per_s"]["min"] <= line["per_rank_clips_per_s"]["max"]
    comm = line["comm"]
    assert comm["collectives_per_step"]["all_gather"] == 1 and comm["collectives_per_step"]["all_reduce_flags"] == 1
    import math
    assert comm["ms_per_step_without_gradient_allreduce"] > 0 and math.isfinite(float(comm["allreduce_exposed_ms"]))
    assert comm["collectives_per_step"]["all_reduce_gradient_buckets"] >= 1 and comm["gradient_bytes_per_step"] > 0
    # counted, not stated: every rank issued 1 packed all-gather + one all-reduce per bucket + the flag all-reduce in a step
    counted = list(comm["collectives_counted_per_rank"].values())[0]
    nb = comm["collectives_per_step"]["all_reduce_gradient_buckets"]
    assert counted == [[1, nb + 1, nb + 2]] * 2, counted
    assert "cpu_baseline" not in line and "c4" not in line          # rank-0-at-N=1-only records


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` typed PLAINLY, as the driver types the N = 1 line (no torch.distributed.run around it, no WORLD_SIZE): the
    parent must start the two ranks itself as a fresh child process group before touching the GPU (bench.py: self_launch), relay rank 0's
    JSON line and the exit code.  Same gloo / same-device hooks as above (one GPU on this box); run/train.py:372-412,579-586 is what the
    ranks replace."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HH_BENCH_BACKEND="gloo", HH_BENCH_SAME_DEVICE="1", HH_BENCH_LAUNCH_TIMEOUT="850")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c1", "--batch", "2", "--no-power"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["global_clips"] == 4 and line["config"]["parallelism"] == "dp2"
    assert line["rccl"]["rccl_ranks"] == 2
    comm = line["comm"]
    nb = comm["collectives_per_step"]["all_reduce_gradient_buckets"]
    counted = list(comm["collectives_counted_per_rank"].values())[0]
    assert counted == [[1, nb + 1, nb + 2]] * 2, counted
    # the per-rank margins the N > 1 line carries: host issue cost per library call and the decoder stream's share of the step
    c = line["config"]
    assert c["libhh_calls_per_step"] > 100 and 0 < c["host_us_per_libhh_call"] < 1000 and c["host_issue_ms_per_step"] > 0
    assert 0 < c["decoder_in_step_frac"]
