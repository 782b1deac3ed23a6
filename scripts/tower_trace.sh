#!/bin/bash
# per-kernel durations of one vision-tower pass, LayerNorm fold on / off (rocprofv3 --kernel-trace) -> gpurun_out/tower_trace/report_fold{1,0}.txt (profiles/r4_tower_trace_*.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tower_trace
for v in 1 0; do
  rm -rf /tmp/tt_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_$v -- python3 $R/scripts/tower_trace.py $v > $R/gpurun_out/tower_trace/trace_$v.log 2>&1
  python3 $R/scripts/tower_trace_report.py /tmp/tt_$v > $R/gpurun_out/tower_trace/report_fold$v.txt 2>&1
  cat $R/gpurun_out/tower_trace/report_fold$v.txt
done
