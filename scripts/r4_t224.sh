#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "224 or ln_fold or gemm" 2>&1 | tail -15 > gpurun_out/r4p/tests_kernels.log
cat gpurun_out/r4p/tests_kernels.log
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_batch_pin_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4p/tests_enc.log
cat gpurun_out/r4p/tests_enc.log
B="python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline --no-power"
for rep in 1 2; do
  $B 2>&1 | tail -1 > gpurun_out/r4p/bench_t224_$rep.log
  $B --tune gemm_tile224=0 2>&1 | tail -1 > gpurun_out/r4p/bench_t256_$rep.log
done
$B --no-ln-fold 2>&1 | tail -1 > gpurun_out/r4p/bench_nofold_t224.log
$B --no-ln-fold --tune gemm_tile224=0 2>&1 | tail -1 > gpurun_out/r4p/bench_nofold_t256.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4p/bench_*.log')):
    try:
        d=json.loads(open(f).read()); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['avg_launch_us'], r['stream_time_over_step'], 'iso', r['isolated']['achieved'], 'loss', d['loss'], d['selfcheck']['encoder_bit_identical_clips_before_last'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-300:])
PY
