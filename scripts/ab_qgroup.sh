# same-session A/B of the grouped query-side launches (hh_qgemm_f32x3_group: a decoder layer's nine weight gradients + the two halves of its
# self-attention in-projection in one launch each) against one launch per product (--no-qgroup, rounds 2-5): clips/s, the decoder stream
# alone and its span inside the pipelined step, library calls and host issue time per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for F in "" "--no-qgroup" "" "--no-qgroup" "" "--no-qgroup"; do
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-power $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1])
c=d['config']
print('%-12s %.2f clips/s  %.2f ms/step | decoder stream alone %.2f ms, in the step %.1f ms (%.2f of the step) | %d library calls, host issue %.2f ms per step | GEMM stream %.1f ms' % (sys.argv[1] if len(sys.argv) > 1 else 'grouped', d['value'], d['ms_per_step'], c['decoder_alone_ms'], c['decoder_in_step_ms'], c['decoder_in_step_frac'], c['libhh_calls_per_step'], c['host_issue_ms_per_step'], d['roofline']['stream_ms_per_step']))" $F
done
