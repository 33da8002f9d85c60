#!/bin/bash
# the train step alone under the kernel trace -> per-shape groups and the step timeline (profiles/r03_trainstep_*)
export TMPDIR=/tmp
O=gpurun_out/r3/ev; mkdir -p $O; rm -rf $O/step
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/step -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > $O/step.log 2>&1 || { tail -5 $O/step.log; exit 1; }
python3 benchmarks/trace_groups.py $O/step/t_kernel_trace.csv $O/r03_trainstep_kernel_groups.csv
python3 benchmarks/trace_step.py $O/step/t_kernel_trace.csv > $O/r03_trainstep_timeline.txt
grep '"metric"' $O/step.log > $O/r03_trainstep_under_rocprof.json
rm -rf $O/step
head -14 $O/r03_trainstep_timeline.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r3/ev/r03_trainstep_kernel_groups.csv")))
tot = sum(int(r["TotalNs"]) for r in rows)
lib = sum(int(r["TotalNs"]) for r in rows if r["Kernel_Name"].startswith("Cijk"))
print(f"kernel time total {tot/1e6/13:.2f} ms per step (13 steps traced); Cijk (hipBLASLt) {lib/tot*100:.1f} %")
for r in rows[:28]:
    print(f'{int(r["TotalNs"])/tot*100:5.1f}%  {int(r["Calls"]):5d} x {float(r["AverageNs"])/1e3:8.1f} us  {r["Kernel_Name"][:90]}')
PY
