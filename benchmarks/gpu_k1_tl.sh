#!/bin/bash
# K1 forward (fp32, B8192 C64 L250): strip length and queue depth sweep
for tl in 32 64 84 125 250; do for qd in 4 8; do
echo -n "TLMAX=$tl QD=$qd: "; MM_K1_TLMAX=$tl MM_K1_QD=$qd timeout -k 10 100 python -u benchmarks/kernel_bench.py --what quick1 2>/dev/null | python -c "
import sys, json
out=[]
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        if r['kernel']=='dwconv7_ln_fwd' and r['C']==64: out.append('%s %.1f us %.0f GB/s' % (r['dtype'][6:], r['us'], r['GBps']))
print(' | '.join(out))
" || exit 1
done; done
