cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "qgemm or sum_partials" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_group.json 2> gpurun_out/bench_group.err
tail -c 900 gpurun_out/bench_group.json
