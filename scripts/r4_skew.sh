#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power"
for rep in 1 2; do
for cfg in "0:4" "16:4" "32:2" "8:8"; do
  sk=${cfg%%:*}; ph=${cfg##*:}
  $B --tune gemm_ln_pskew=$sk --tune gemm_ln_phases=$ph 2>&1 | tail -1 > gpurun_out/r4e/bench_sk${sk}_ph${ph}_$rep.log
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
