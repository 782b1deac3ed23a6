#!/bin/bash
# round-4 first GPU contact: new pinning tests, bracket-vs-rocprof reconciliation in ONE run, a full bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_batch_pin_gpu.py tests/test_encoder_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4a/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4a/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-mcq --no-c4 --no-cpu-baseline --no-power > $GRAFT_REPO_ROOT/gpurun_out/r4a/bench_under_rocprof.log 2>&1
cp $(find $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r4a/kernel_stats_timers_on.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4a/prof
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.log 2>&1
tail -c 600 gpurun_out/r4a/tests.log
