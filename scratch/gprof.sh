cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_greedy -- python3 scratch/greedy_prof.py > gpurun_out/prof_greedy.log 2>&1 || exit 1
grep sec gpurun_out/prof_greedy.log
python3 scratch/profsum.py gpurun_out/prof_greedy 80 30
