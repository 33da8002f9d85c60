cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_greedy -- python3 scratch/greedy_ab.py > gpurun_out/prof_greedy.log 2>&1 || exit 1
python3 scratch/profsum.py gpurun_out/prof_greedy 320 22
