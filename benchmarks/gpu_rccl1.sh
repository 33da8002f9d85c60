#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2/rccl1; rm -rf $O; mkdir -p $O
STEPS=6 timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/t -o t -- python3 benchmarks/rccl1_step.py > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
grep ms_per_step $O/log.txt
python3 benchmarks/trace_step.py $O/t/t_kernel_trace.csv > $O/timeline.txt; head -14 $O/timeline.txt
python3 benchmarks/trace_groups.py $O/t/t_kernel_trace.csv $O/groups.csv; grep -i -E "ccl|rccl|AllReduce|copyBuffer|multi_tensor" $O/groups.csv | cut -c1-200 | head
rm -rf $O/t
