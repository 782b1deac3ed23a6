#!/bin/bash
# GPU-box verification of a build (run through gpurun): full -m gpu suite, smoke, the default bench line, LayerNorm-fold A/B, tower traces -> gpurun_out/verify/
cd $GRAFT_REPO_ROOT
O=gpurun_out/verify; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/tests_gpu.log; cat $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 > $O/bench_default.json
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power"
for rep in 1 2; do
  $B 2>&1 | tail -1 > $O/bench_fold_$rep.log
  $B --no-ln-fold 2>&1 | tail -1 > $O/bench_nofold_$rep.log
done
python - <<'PY'
import json,glob
for f in ['gpurun_out/verify/bench_default.json']+sorted(glob.glob('gpurun_out/verify/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], r['stream_time_over_step'], 'iso', r['isolated']['achieved'], 'loss', d['loss'], d['selfcheck'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf /tmp/tt_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_$v -- python3 $R/scripts/tower_trace.py $v > $R/$O/trace_$v.log 2>&1
  python3 $R/scripts/tower_trace_report.py /tmp/tt_$v > $R/$O/report_fold$v.txt 2>&1
  head -14 $R/$O/report_fold$v.txt
done
