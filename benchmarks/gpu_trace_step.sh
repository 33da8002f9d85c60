#!/bin/bash
# the train step alone under the kernel trace -> per-shape groups and the step timeline (profiles/r02_trainstep_*)
export TMPDIR=/tmp
O=gpurun_out/r2/ev; mkdir -p $O; rm -rf $O/step
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/step -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > $O/step.log 2>&1 || { tail -5 $O/step.log; exit 1; }
python3 benchmarks/trace_groups.py $O/step/t_kernel_trace.csv $O/r02_trainstep_kernel_groups.csv
python3 benchmarks/trace_step.py $O/step/t_kernel_trace.csv > $O/r02_trainstep_timeline.txt
grep '"metric"' $O/step.log > $O/r02_trainstep_under_rocprof.json
rm -rf $O/step
head -14 $O/r02_trainstep_timeline.txt
