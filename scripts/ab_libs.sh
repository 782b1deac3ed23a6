# A/B of two builds of libhh.so in ONE session: usage  bash scripts/ab_libs.sh <libA.so> <libB.so> [rounds] [steps] [bench flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${3:-3}
S=${4:-20}
for i in $(seq $N); do
for L in "$1" "$2"; do
HH_LIBHH_PATH=$R/$L python3 $R/bench.py --steps $S --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-kernel-timers $5 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('%-60s' % sys.argv[1], d['value'], d['ms_per_step'])" "[$L]"
done
done
