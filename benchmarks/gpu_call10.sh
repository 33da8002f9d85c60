#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
cd benchmarks
MM_DECODE_GEMM=lib timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d ../gpurun_out/r2/decode_trace -o t -- python3 decode_bench.py > ../gpurun_out/r2/decode_trace.log 2>&1
cd ..
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r2/decode_trace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last bf16 generate: find decode_attn_k launches; bf16 run is first (2 generates), fp32 second
da=[i for i,r in enumerate(rows) if 'decode_attn_k' in r['Kernel_Name']]
print('decode_attn launches', len(da))
# bf16: first 2*79*4 launches; use second generate of bf16: launches [79*4, 2*79*4)
n=79*4
seg=rows[da[n]:da[2*n-1]+1]
t0=int(seg[0]['Start_Timestamp']); t1=int(seg[-1]['End_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
print('bf16 decode loop: wall %.1f ms, kernel busy %.1f ms, kernels %d, per step %.3f ms' % ((t1-t0)/1e6, busy/1e6, len(seg), (t1-t0)/1e6/79))
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    k=r['Kernel_Name'][:70]; agg[k][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); agg[k][1]+=1
for k,(t,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:16]:
    print('%8.1f us/step %4d calls/step %7.1f us avg  %s' % (t/1e3/79, c/79, t/1e3/c, k))
PY
rm -rf gpurun_out/r2/decode_trace
