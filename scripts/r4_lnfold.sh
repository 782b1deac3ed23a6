#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ln_fold or gemm" 2>&1 | tail -15 > gpurun_out/r4b/tests_kernels.log
python -m pytest tests/test_encoder_gpu.py tests/test_batch_pin_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r4b/tests_enc.log
tail -5 gpurun_out/r4b/tests_kernels.log; tail -8 gpurun_out/r4b/tests_enc.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power 2>&1 | tail -1 > gpurun_out/r4b/bench_fold_$i.log
python bench.py --no-ln-fold --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power 2>&1 | tail -1 > gpurun_out/r4b/bench_nofold_$i.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']; a=d['attention_roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], r['stream_time_over_step'], 'ln', a['add_ln']['avg_launch_us'], a['add_ln']['stream_time_over_step'], a['add_ln']['launches_timed'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
