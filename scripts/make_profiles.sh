#!/bin/bash
# Regenerates the committed profile summaries on the GPU box (run through gpurun; copy gpurun_out/profiles_new/* to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default pipelined bench      -> <tag>_bench_kernel_stats.csv, <tag>_summary.md
#      (round 4: 8 warm-up + 12 timed steps instead of 2 + 3 -- a 5-step run ends before the chip reaches its steady power / clock state and
#       reads the GEMMs ~8 % faster than the bench's own 20-step region: 468 vs 504-511 us per launch in the same session)
#   2. same for the un-pipelined step (durations not inflated by overlap)   -> <tag>_unpipelined_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)            -> <tag>_pmc_summary.{md,json}
#   usage: make_profiles.sh <tag> [extra bench.py arguments, e.g. --config c4 --batch 4]   (default: config 2, B = 32)
TAG=${1:-r1_final}
shift
EXTRA="$*"
export HH_PROFILE_EXTRA="$EXTRA"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
rm -rf $R/gpurun_out/prof_p $R/gpurun_out/prof_u $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/pmc_SQ
BENCH="python3 $R/bench.py --no-cpu-baseline --no-kernel-timers --no-mcq --no-c4 --no-variants --no-power --no-selfcheck $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p -- $BENCH --steps 12 --warmup 8 > $OUT/${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_u -- $BENCH --steps 12 --warmup 8 --no-pipeline > $OUT/${TAG}_bench_unpipelined.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- $BENCH --steps 1 --warmup 1 --no-pipeline > $OUT/${TAG}_pmc_$c.log 2>&1
done
#   4. SQ pass: matrix-core / VALU busy cycles per kernel                   -> <tag>_sq_summary.md
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_SQ -- $BENCH --steps 1 --warmup 1 --no-pipeline > $OUT/${TAG}_pmc_SQ.log 2>&1
TAG=$TAG python3 - <<'PY'
import csv, glob, json, os, collections
R, TAG = os.environ["GRAFT_REPO_ROOT"], os.environ["TAG"]
EXTRA = os.environ.get("HH_PROFILE_EXTRA", "").strip()
FLAGS = "--no-cpu-baseline --no-kernel-timers --no-mcq --no-c4 --no-variants --no-power --no-selfcheck" + (" " + EXTRA if EXTRA else "")
WHAT = ("config 4 (32-frame 336p, nq=12), B = 4 clips" if "c4" in EXTRA else "config 2 (16-frame 224p, nq=12), B = 32 clips") if "--batch" not in EXTRA or "c4" in EXTRA else "bench.py " + EXTRA
OUT = R + "/gpurun_out/profiles_new/"
STEPS = 20
def stats(d, dst, title, cmd, note):
    f = glob.glob(R + "/gpurun_out/%s/**/*kernel_stats.csv" % d, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    open(OUT + dst + "_kernel_stats.csv", "w").write(open(f).read())
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    L = ["# %s -- rocprofv3 --kernel-trace --stats of `%s`" % (title, cmd), "", note % (tot / 1e6, tot / (STEPS * 1e6)), "",
         "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---|---|---|---|"]
    for r in rows[:24]:
        L.append("| %s | %s | %.2f | %.1f | %.1f |" % (r["Name"][:96].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                    float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    return L
L = stats("prof_p", TAG + "_bench", TAG + " build", "python3 bench.py --steps 12 --warmup 8 " + FLAGS,
          "MI355X, " + WHAT + ", software-pipelined step, 20 steps profiled (8 warm-up + 12 timed).\nSum of kernel durations %.1f ms over 20 steps = %.1f ms/step (kernels of the encoder stream and of the decoder stream overlap, so this sum exceeds the wall time per step).")
L += [""] + stats("prof_u", TAG + "_unpipelined", "Same build, un-pipelined step", "python3 bench.py --steps 12 --warmup 8 " + FLAGS + " --no-pipeline",
                  "Every kernel runs alone on the chip here, so the averages are the isolated kernel durations.\nSum of kernel durations %.1f ms over 20 steps = %.1f ms/step.")
# ---- steady state of the un-pipelined run: the dispatches between the first and the last optimizer launch (adamw_arena_kernel = one per step), so
# that the process's start-up (weight packing, copies, fills) does not count as "per step"
LIBHH = ("gemm", "attn", "ln_", "add_ln", "embed", "im2col", "cast_", "transpose", "qgemm", "qself", "xattn", "lsap", "match_boxes", "box_", "rownorm", "egonce",
         "masked_ce", "tv_accuracy", "adamw", "cls_combine", "text_flags", "_Z13ln_fwd", "accuracy", "gather_rows", "word_", "dropout_", "mattn", "sum_partials")
tf = glob.glob(R + "/gpurun_out/prof_u/**/*kernel_trace.csv", recursive=True)
if tf:
    tr = sorted(csv.DictReader(open(tf[0])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(tr) if "adamw_arena_kernel" in r["Kernel_Name"]]
    if len(marks) >= 3:
        seg = tr[marks[0] + 1:marks[-1] + 1]
        nsteps = len(marks) - 1
        cnt, dur = collections.Counter(), collections.Counter()
        for r in seg:
            cnt[r["Kernel_Name"]] += 1; dur[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        is_lib = lambda n: any(t in n for t in LIBHH) and "at::native" not in n and "rocclr" not in n and "rocblas" not in n
        lib = sum(c for n, c in cnt.items() if is_lib(n)); oth = sum(c for n, c in cnt.items() if not is_lib(n))
        with open(OUT + TAG + "_unpipelined_steady_kernel_stats.csv", "w") as f:
            w = csv.writer(f); w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "libhh"])
            for n, c in sorted(cnt.items(), key=lambda x: -dur[x[0]]):
                w.writerow([n, c, "%.2f" % (c / nsteps), dur[n], "%.0f" % (dur[n] / c), int(is_lib(n))])
        L += ["", "## Steady state of the un-pipelined run (" + TAG + "_unpipelined_steady_kernel_stats.csv)", "",
              "Dispatches between the first and the last `adamw_arena_kernel` launch (one per step): %d steps, **%.1f launches per step: %.1f libhh + %.1f other** "
              "(stock torch ops, blits); the whole-process table above also counts the start-up (weight packing, copies, fills)." % (nsteps, (lib + oth) / nsteps, lib / nsteps, oth / nsteps)]
open(OUT + TAG + "_summary.md", "w").write("\n".join(L) + "\n")
# ---- PMC
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(R + "/gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                agg[row["Kernel_Name"]] += float(row["Counter_Value"]); cnt[row["Kernel_Name"]] += 1
    val[c] = (agg, cnt)
names = [k for k in val["FETCH_SIZE"][0] if any(t in k for t in ("gemm", "attn", "ln_", "add_ln", "xattn", "mattn", "transpose", "embed", "im2col"))]
js, rows = {}, []
for k in names:
    n = val["FETCH_SIZE"][1][k]
    fetch = 2.0 * val["FETCH_SIZE"][0][k] * 1024.0            # KiB, x2: gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md)
    write = val["WRITE_SIZE"][0].get(k, 0.0) * 1024.0
    js[k] = {"launches": n, "fetch_bytes_corrected": int(fetch / n), "write_bytes": int(write / n), "traffic_bytes_per_launch": int((fetch + write) / n)}
    rows.append((fetch + write, "| %s | %d | %.1f | %.1f | %.1f |" % (k[:80].replace("|", "/"), n, fetch / n / 2**20, write / n / 2**20, (fetch + write) / n / 2**20)))
json.dump(js, open(OUT + TAG + "_pmc_summary.json", "w"), indent=1)
M = ["# " + TAG + " -- HBM-side traffic from PMC counters (rocprofv3 --pmc, separate passes)", "",
     "Commands (one counter per pass, as MI355X_MICROARCH.md prescribes):", "",
     "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py --steps 1 --warmup 1 " + FLAGS + " --no-pipeline",
     "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -- python3 bench.py --steps 1 --warmup 1 " + FLAGS + " --no-pipeline", "",
     "Units/corrections: counter values are KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of wide coalesced streaming reads -> doubled below; WRITE_SIZE is exact.",
     "The counters sit on the L2's fabric side, so Infinity-Cache hits are included.  Per-launch averages, " + WHAT + ".", "",
     "| kernel | launches | fetch (corrected) MiB | write MiB | traffic / launch MiB |", "|---|---|---|---|---|"]
M += [r for _, r in sorted(rows, reverse=True)]
# ---- per-GEMM table of the vision tower: the persistent kernel's instantiations <true, 5> = qkv (time and space), <6> = fc1,
# <7> = time projection, <8> = space projection AND fc2 -- told apart by dispatch order (they alternate: proj, fc2 per block)
try:
    import re as _re
    c4 = "c4" in EXTRA
    Mrows, Dm = (4 * 18433, 1024) if c4 else (32 * 4097, 1024)
    aw = lambda N, K: 2 * (Mrows * K + N * K)
    alg = {"qkv (time + space)": aw(3 * Dm, Dm) + 2 * Mrows * 3 * Dm, "time proj": aw(Dm, Dm) + Mrows * Dm * 4, "space proj": aw(Dm, Dm) + Mrows * Dm * 8,
           "fc1": aw(4 * Dm, Dm) + 2 * Mrows * 4 * Dm, "fc2": aw(Dm, 4 * Dm) + Mrows * Dm * 8}
    per = {k: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]} for k in alg}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        rowsc = []
        for f in glob.glob(R + "/gpurun_out/pmc_%s/**/*counter_collection.csv" % cname, recursive=True):
            rowsc += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == cname and "gemm256w4p_kernel<true," in r["Kernel_Name"]]
        rowsc.sort(key=lambda r: int(r["Dispatch_Id"]))
        n8 = 0
        for r in rowsc:
            m = _re.search(r"gemm256w4p_kernel<true, (\d+),", r["Kernel_Name"])
            e = int(m.group(1)) if m else -1
            key = {5: "qkv (time + space)", 6: "fc1", 7: "time proj"}.get(e)
            if e == 8:
                key = "space proj" if n8 % 2 == 0 else "fc2"
                n8 += 1
            if key:
                per[key][cname][0] += float(r["Counter_Value"]) * 1024.0 * (2.0 if cname == "FETCH_SIZE" else 1.0)
                per[key][cname][1] += 1
    M += ["", "## Vision-tower GEMMs one by one (bytes per launch; algorithmic = bench.py: gemm_algorithmic_table, the bf16 pair stream)", "",
          "| GEMM | launches | fetch (corrected) MiB | write MiB | traffic MiB | algorithmic MiB | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
    tot_t = tot_a = tot_n = 0
    for k in alg:
        nf, nw = per[k]["FETCH_SIZE"][1], per[k]["WRITE_SIZE"][1]
        if not nf or not nw: continue
        fe, wr = per[k]["FETCH_SIZE"][0] / nf, per[k]["WRITE_SIZE"][0] / nw
        M.append("| %s | %d | %.0f | %.0f | %.0f | %.0f | %.2f |" % (k, nf, fe / 2**20, wr / 2**20, (fe + wr) / 2**20, alg[k] / 2**20, (fe + wr) / alg[k]))
        tot_t += (fe + wr) * nf; tot_a += alg[k] * nf; tot_n += nf
    if tot_n:
        M.append("| launch-weighted mean | %d | | | %.0f | %.0f | **%.2f** |" % (tot_n, tot_t / tot_n / 2**20, tot_a / tot_n / 2**20, tot_t / tot_a))
        M += ["", "(`roofline.traffic` / `roofline.algorithmic_bytes_per_launch` of bench.py's line are these two means; the counters sit on L2's fabric side and include Infinity-Cache hits.)"]
except Exception as _e:
    M += ["", "(per-GEMM table not produced: %r)" % (_e,)]
open(OUT + TAG + "_pmc_summary.md", "w").write("\n".join(M) + "\n")
print("\n".join(M[-len(rows):]))
# ---- SQ counters: MFMA / VALU busy share per kernel
agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(int)
for f in glob.glob(R + "/gpurun_out/pmc_SQ/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_BUSY_CU_CYCLES": cnt[row["Kernel_Name"]] += 1
S = ["# " + TAG + " -- matrix-core and VALU utilisation per kernel (rocprofv3 --pmc, SQ counters, one pass)", "",
     "    rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 " + FLAGS + " --no-pipeline", "",
     "MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) (the rocprof-compute definition); VALU busy % = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (4 x SQ_BUSY_CU_CYCLES);",
     "MFMA flop = 512 x SQ_INSTS_VALU_MFMA_MOPS_BF16.  Per-launch averages, " + WHAT + ".", "",
     "| kernel | launches | MFMA busy % | VALU busy % | MFMA GFLOP / launch | SQ_INSTS_VALU / launch (M) |", "|---|---|---|---|---|---|"]
rows = []
for k, c in agg.items():
    if not any(t in k for t in ("gemm", "attn", "ln_", "add_ln", "xattn", "mattn")) or c["SQ_BUSY_CU_CYCLES"] <= 0: continue
    busy = 4.0 * c["SQ_BUSY_CU_CYCLES"]
    rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"], "| %s | %d | %.1f | %.1f | %.1f | %.1f |" % (k[:80].replace("|", "/"), cnt[k], 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy,
                 100 * 4.0 * c["SQ_ACTIVE_INST_VALU"] / busy, 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / max(cnt[k], 1) / 1e9, c["SQ_INSTS_VALU"] / max(cnt[k], 1) / 1e6)))
S += [r for _, r in sorted(rows, reverse=True)]
open(OUT + TAG + "_sq_summary.md", "w").write("\n".join(S) + "\n")
print("\n".join(S[-len(rows):]))
PY
ls -la $OUT
