#!/bin/bash
# SQ counters of the two attention kernels (scripts/attn_only.py), one rocprofv3 --pmc pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ATTN_ITERS=2 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc/g$i -- python3 $R/scripts/attn_only.py > $R/gpurun_out/attn_pmc_g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(R + "/gpurun_out/attn_pmc/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:28]
        if "attn" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]): print(f"   {c:32s} {agg[k][c]/cnt[k][c]:16.0f}  per launch ({cnt[k][c]} launches)")
PY
