#!/bin/bash
# MFMA / VALU busy share and duration of the space-attention kernel variants (round 6): one rocprofv3 --pmc pass over scripts/space_variants_run.py
#   usage: space_variants_pmc.sh [B T n]      (default 32 16 256; config 4: 4 32 576)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/space_pmc
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/space_pmc -- python3 $R/scripts/space_variants_run.py $* > $R/gpurun_out/space_pmc.log 2>&1
python3 - $* <<'PY'
import csv, glob, os, sys, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(R + "/gpurun_out/space_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "space_attn" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_BUSY_CU_CYCLES": cnt[k] += 1
dur = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/space_pmc/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "space_attn" in row["Kernel_Name"]: dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("| kernel (B T n = %s) | launches | us / launch under the counters (median) | MFMA busy %% | VALU busy %% | MFMA GFLOP / launch | VALU instructions / launch (M) |" % (" ".join(sys.argv[1:]) or "32 16 256"))
print("|---|---|---|---|---|---|---|")
for k, c in agg.items():
    busy = 4.0 * c["SQ_BUSY_CU_CYCLES"]
    d = sorted(dur[k])
    print("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f |" % (k[:44], cnt[k], d[len(d) // 2] if d else 0.0, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, 100 * 4.0 * c["SQ_ACTIVE_INST_VALU"] / busy,
                                                 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / max(cnt[k], 1) / 1e9, c["SQ_INSTS_VALU"] / max(cnt[k], 1) / 1e6))
PY
