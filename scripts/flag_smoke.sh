# every A/B flag of bench.py still runs (3 timed steps each); prints clips/s and the loss, which must agree across flags to ~1e-3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for F in "" "--no-ln-fold" "--stream-fp32" "--kv-proj" "--token-major-qkv" "--time-proj-fp32" "--walk-forward" "--space-16q" "--no-pipeline" "--config c1" "--config c4 --batch 2" "--workload mcq"; do
python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-kernel-timers --no-power $F 2>&1 | python3 -c "import json,sys; t=sys.stdin.read(); l=[x for x in t.splitlines() if x.startswith(chr(123))]; d=json.loads(l[-1]) if l else None; print('%-28s' % sys.argv[1], (d['value'], d.get('loss')) if d else 'FAILED: ' + t[-300:])" "[$F]"
done
