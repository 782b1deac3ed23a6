#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/tests_gpu.log; cat $O/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 > $O/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4final2/bench_default.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['last_dispatched'], d['mcq']['value'], d['c4']['value'], d['variants'], d['power'], d['selfcheck']['encoder_bit_identical_clips_before_last'])
PY
