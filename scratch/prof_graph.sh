#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_graph -- python3 scratch/graph_try.py local_config > gpurun_out/prof_graph.log 2>&1 || exit 1
f=$(ls -t gpurun_out/prof_graph/*/*_kernel_trace.csv | head -1)
python3 scratch/timeline.py $f dump > gpurun_out/prof_graph_timeline.txt
head -12 gpurun_out/prof_graph_timeline.txt
