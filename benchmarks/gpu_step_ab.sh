#!/bin/bash
# train-step A/B (headline config, no extras): default vs variants given as env assignments in $VARIANTS (space-separated)
set -o pipefail
for rep in 1 2; do
for v in "X=0" $VARIANTS; do
  echo -n "$v: "
  env $v timeout -k 10 200 python -u bench.py --no-extras --steps 30 --warmup 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%.2f ms/step  %.0f spectra/s' % (r['ms_per_step'], r['value']))
" || exit 1
done
done
