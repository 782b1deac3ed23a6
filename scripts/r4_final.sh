#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
python bench.py --steps 20 --warmup 5 > gpurun_out/r4k/bench_line.json 2> gpurun_out/r4k/bench_err.log
tail -c 400 gpurun_out/r4k/bench_line.json
