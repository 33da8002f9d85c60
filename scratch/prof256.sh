#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MM_DEFER_DW_LAYERS=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b256 -- python3 bench.py --batch 256 --steps 2 --warmup 1 --no-extras > gpurun_out/prof_b256.log 2>&1 || exit 1
python3 scratch/profsum.py gpurun_out/prof_b256 3 30
