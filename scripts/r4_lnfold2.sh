#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power"
for sk in 0 2 4 8; do
  $B --tune gemm256_pskew=$sk 2>&1 | tail -1 > gpurun_out/r4d/bench_fold_pskew$sk.log
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in fold nofold; do
  rm -rf $R/gpurun_out/r4d/prof_$v
  X=""; [ $v = nofold ] && X="--no-ln-fold"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4d/prof_$v -- python3 $R/bench.py --steps 6 --warmup 4 --no-mcq --no-c4 --no-cpu-baseline --no-power --no-kernel-timers --no-pipeline $X > $R/gpurun_out/r4d/prof_$v.log 2>&1
  cp $(find $R/gpurun_out/r4d/prof_$v -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4d/kernel_stats_$v.csv
  rm -rf $R/gpurun_out/r4d/prof_$v
done
cd $R
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r4d/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']; a=d['attention_roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], r['stream_time_over_step'], 'ln', a['add_ln']['avg_launch_us'], a['add_ln']['stream_time_over_step'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
for v in ('fold','nofold'):
    print('==',v)
    for r in list(csv.DictReader(open('gpurun_out/r4d/kernel_stats_%s.csv'%v)))[:14]:
        print("%-70s %6s %9.1f us  %5.1f%%"%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY
