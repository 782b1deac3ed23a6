#!/bin/bash
# rocprofv3 kernel trace of the un-pipelined step: per-kernel totals (the decoder / loss / optimizer side is what the pipelined
# step has to hide behind the next batch's frozen towers).  usage: prof_step.sh [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_np
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_np -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-pipeline --no-mcq --no-c4 --no-variants --no-selfcheck --no-power $* > $R/gpurun_out/prof_np.log 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_np/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 5 steps: %.1f" % (tot / 1e6))
for r in rows[:60]:
    print("%-90s %6s calls %9.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
