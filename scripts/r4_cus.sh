#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power --no-variants --no-selfcheck"
for c in 0 248 240 0 248; do
  $B --enc-cus $c 2>&1 | tail -1 > gpurun_out/r4s/bench_cus${c}_$RANDOM.log
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r4s/bench_*.log'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
