# same-session sweep of bench.py --tune settings: usage  bash scripts/ab_tunes.sh <rounds> <steps> "<flags 1>" "<flags 2>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=$1; S=$2; shift; shift
for i in $(seq $N); do
for F in "$@"; do
python3 $R/bench.py --steps $S --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-kernel-timers $F 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('%-30s' % sys.argv[1], d['value'], d['ms_per_step'], d.get('loss'))" "[$F]"
done
done
