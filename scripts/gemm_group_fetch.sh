#!/bin/bash
# FETCH_SIZE (L2-side read traffic) of the persistent GEMM for XCD tile-walk shapes: G m-tiles x (32 / G) n-tiles per round and XCD.
# -> gpurun_out/gemm_group_fetch.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gemm_group_fetch.md
echo "# Persistent GEMM (${1:-plain} epilogue): XCD tile walk vs L2-side fetch traffic (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction), M = 32 x 4097 rows" > $OUT
echo "" >> $OUT
echo "| GEMM | G (m-tiles x n-tiles per round and XCD) | FETCH MiB / launch | x operand bytes | TFLOP/s (same process, timers off) |" >> $OUT
echo "|---|---|---|---|---|" >> $OUT
MODE=${1:-plain}          # "fold": the consumer side of the LayerNorm fold (what the step runs since round 4)
for which in ${2:-qkv fc1}; do
for G in ${3:-4 8 16 32}; do
  rm -rf $R/gpurun_out/pmc_g
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g -- python3 $R/scripts/gemm_group_fetch.py $G $which $MODE > $R/gpurun_out/pmc_g.log 2>&1
  TF=$(python3 $R/scripts/gemm_group_fetch.py $G $which $MODE 2>/dev/null | grep RESULT | sed -e 's/.*call, //')
  python3 - "$which" "$G" "$TF" >> $OUT <<'PY'
import csv, glob, os, sys
which, G, tf = sys.argv[1], int(sys.argv[2]), sys.argv[3]
R = os.environ["GRAFT_REPO_ROOT"]
tot, n = 0.0, 0
for f in glob.glob(R + "/gpurun_out/pmc_g/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and ("gemm256d_kernel" in row["Kernel_Name"] or "gemm256w4p_kernel" in row["Kernel_Name"]):
            tot += float(row["Counter_Value"]); n += 1
M = 32 * 4097
N, K = {"qkv": (3072, 1024), "fc1": (4096, 1024), "proj": (1024, 1024), "fc2": (1024, 4096)}[which]
ops = (M * K + N * K) * 2 / 2**20
fetch = 2.0 * tot * 1024 / max(n, 1) / 2**20
print("| %s | %d (%d x %d) | %.0f | %.2f | %s |" % (which, G, G, 32 // G, fetch, fetch / ops, tf))
PY
done
done
cat $OUT
