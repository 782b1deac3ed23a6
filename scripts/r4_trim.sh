#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "caption_length or no_host_sync or golden" 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 --no-mcq --no-c4 --no-cpu-baseline 2>gpurun_out/r4r/err.log | tail -1 > gpurun_out/r4r/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4r/bench.json').read())
print(d['value'], d['ms_per_step'], d['loss'], d.get('variants'), d['roofline']['last_dispatched'])
PY
tail -3 gpurun_out/r4r/err.log
