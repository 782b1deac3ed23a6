#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power --no-selfcheck"
for cfg in "0:4" "32:2" "64:2" "24:4" "48:4" "12:8" "0:4"; do
  sk=${cfg%%:*}; ph=${cfg##*:}
  $B --tune gemm_ln_pskew=$sk --tune gemm_ln_phases=$ph 2>&1 | tail -1 > gpurun_out/r4m/bench_sk${sk}_ph${ph}_$RANDOM.log
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r4m/bench_*.log'), key=os.path.getmtime):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], 'iso', r['isolated']['achieved'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
