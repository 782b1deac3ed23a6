cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mcq --no-kernel-timers | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused   ', d['value'], d['ms_per_step'])"
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mcq --no-kernel-timers --per-op-query-side | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-op  ', d['value'], d['ms_per_step'])"
done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mcq --no-kernel-timers --no-pipeline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused nopipe ', d['value'], d['ms_per_step'])"
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mcq --no-kernel-timers --no-pipeline --per-op-query-side | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-op nopipe', d['value'], d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timers --no-mcq --steps 3 --warmup 2 --no-pipeline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/prof_q/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", tot/5e6, "launches/step", sum(int(r["Calls"]) for r in rows)/5)
for r in rows[:45]:
    print("%-90s %6s %9.2f %8.1f %5.1f"%(r["Name"][:90],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
