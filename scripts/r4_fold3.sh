#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ln_fold or gemm" 2>&1 | tail -6 > gpurun_out/r4g/tests_kernels.log
python -m pytest tests/test_encoder_gpu.py tests/test_batch_pin_gpu.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4g/tests_enc.log
cat gpurun_out/r4g/tests_kernels.log gpurun_out/r4g/tests_enc.log | tail -20
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power"
for rep in 1 2; do
  $B 2>&1 | tail -1 > gpurun_out/r4g/bench_fold_$rep.log
  $B --no-ln-fold 2>&1 | tail -1 > gpurun_out/r4g/bench_nofold_$rep.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], r['stream_time_over_step'], 'loss', d['loss'], d['selfcheck']['encoder_bit_identical_clips_before_last'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
bash scripts/r4_trace.sh 2>&1 | grep -v "^  _Z\|cls_combine\|embed_ln" | head -30
