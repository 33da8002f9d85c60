#!/bin/bash
# optimizer-into-next-forward experiment: HOOK=model starts the deferred updates at the top of the step (beside the encoder forward)
for cfg in "model 0" "model 1" "decoder 1"; do set -- $cfg; echo "== hook=$1 first=$2"; HOOK=$1 timeout -k 10 150 python benchmarks/opt_overlap_exp.py $2 2>&1 | grep "ms/step" || exit 1; done
