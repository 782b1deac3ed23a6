# same-session A/B of the row-padded text stream (2464 -> 2560 rows: no 160-row GEMM tails as separate launches) against the un-padded one
# (--no-text-pad): clips/s, decoder stream, library calls per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for F in "" "--no-text-pad" "" "--no-text-pad" "" "--no-text-pad"; do
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-power $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1])
c=d['config']
print('%-12s %.2f clips/s  %.2f ms/step | decoder stream alone %.2f ms, in the step %.1f ms (%.2f of the step) | %d library calls, host issue %.2f ms per step | GEMM stream %.1f ms' % (sys.argv[1] if len(sys.argv) > 1 else 'padded', d['value'], d['ms_per_step'], c['decoder_alone_ms'], c['decoder_in_step_ms'], c['decoder_in_step_frac'], c['libhh_calls_per_step'], c['host_issue_ms_per_step'], d['roofline']['stream_ms_per_step']))" $F
done
