#!/bin/bash
# SQ / traffic counters of the memory-space cross-attention kernels, stand-alone (B = 32, M = 4096): separate --pmc passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/mattn_bench.py 32 4096 20
python3 $R/scripts/mattn_bench.py 4 18432 20
for c in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf $R/gpurun_out/mattn_pmc_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/mattn_pmc_$tag -- python3 $R/scripts/mattn_bench.py 32 4096 2 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/mattn_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "mattn" not in k and "gemm_tn" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] in ("SQ_BUSY_CU_CYCLES", "FETCH_SIZE", "WRITE_SIZE"): cnt[(k, row["Counter_Name"])] += 1
for k, c in agg.items():
    n = max(cnt[(k, "SQ_BUSY_CU_CYCLES")], 1); nf = max(cnt[(k, "FETCH_SIZE")], 1); nw = max(cnt[(k, "WRITE_SIZE")], 1)
    busy = 4.0 * c["SQ_BUSY_CU_CYCLES"]
    print("%-28s launches %d: MFMA busy %.1f %%  VALU busy %.1f %%  wave wait %.1f %%  LDS bank-conflict cycles / LDS active %.3f  fetch %.1f MiB (corrected x2)  write %.1f MiB" % (
        k, n, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(busy, 1), 100 * 4 * c["SQ_ACTIVE_INST_VALU"] / max(busy, 1), 100 * c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
        c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1), 2 * c["FETCH_SIZE"] / nf / 1024, c["WRITE_SIZE"] / nw / 1024))
PY
