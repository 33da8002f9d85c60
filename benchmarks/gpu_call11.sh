#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> gpurun_out/r2/bench_train.err | tee gpurun_out/r2/bench_train.json | cut -c1-300
timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2/train_trace -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > gpurun_out/r2/train_trace.log 2>&1
python3 benchmarks/trace_groups.py gpurun_out/r2/train_trace/t_kernel_trace.csv gpurun_out/r2/train_groups.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2/train_trace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 5 steps: find og_adamw launches (one per step? count) and take the window between the 6th-last and last
idx=[i for i,r in enumerate(rows) if 'og_adamw_k' in r['Kernel_Name']]
per=len(idx)//13
print('adamw launches', len(idx), 'per step', per)
s=idx[-5*per-1]+1; e=idx[-1]
seg=rows[s:e+1]
t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
# union busy time
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in seg)
busy=0; cs,ce=iv[0]
for a,b in iv[1:]:
    if a>ce: busy+=ce-cs; cs,ce=a,b
    else: ce=max(ce,b)
busy+=ce-cs
print('5 steps: wall %.2f ms/step, any-kernel-busy %.2f ms/step, sum of kernel time %.2f ms/step, kernels/step %d' % ((t1-t0)/5e6, busy/5e6, sum(b-a for a,b in iv)/5e6, len(seg)//5))
PY
rm -rf gpurun_out/r2/train_trace
