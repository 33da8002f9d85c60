#!/bin/bash
# usage: scratch/prof.sh <tag>  -- profile bench and print per-step kernel table
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 5 --warmup 2 --no-extras > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 scratch/profsum.py gpurun_out/prof_$tag 7
